"""Micro-benchmark of the weight-stationary pyramid GEMMs (value projection -> pair layout, G projection) at
cfg-2 size over the persistent-grid knob of wreg_gemm.hip.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd import _lib, ops  # noqa: E402

lib = _lib.load()
n_img, S = 5, 40320
feat = torch.randn(n_img, S, 256, device="cuda").to(torch.bfloat16)
Wv = (torch.randn(256, 256, device="cuda") / 16).to(torch.bfloat16)
Wf = ops.swizzle_weight(Wv)
bv = torch.randn(256, device="cuda")
vp = torch.empty((n_img, 8, S, 32), dtype=torch.bfloat16, device="cuda")


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for grid in (256, 512, 768, 1024):
    lib.mvg_set_tuning(b"wreg_grid", grid)
    a = t(lambda: ops.value_proj_planes_ws(feat, Wf, bv, vp))
    b = t(lambda: ops.feat_linear_ws(feat, Wf, 192))
    print("grid %4d  value->planes %7.1f us (%5.0f GB/s)   G %7.1f us (%5.0f GB/s)"
          % (grid, a, 206e6 / a / 1e3, b, 180e6 / b / 1e3))
lib.mvg_set_tuning(b"wreg_grid", 512)
