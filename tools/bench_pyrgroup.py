"""Micro-benchmark of the grouped pyramid products (mvg_pyramid_group_ws) against one launch per product, at a configuration's
pyramid size.  tools/bench_pyrgroup.py [n_img] [layers]   GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd import _lib, ops  # noqa: E402

lib = _lib.load()
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 5
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = 40320
feat = torch.randn(n_img, S, 256, device="cuda").to(torch.bfloat16)
jobs = []
for l in range(layers):
    W = ops.swizzle_weight((torch.randn(256, 256, device="cuda") / 16).to(torch.bfloat16))
    Wg = ops.swizzle_weight((torch.randn(256, 256, device="cuda") / 16).to(torch.bfloat16))
    jobs += [(W, torch.randn(256, device="cuda"), torch.empty((n_img, 8, S, 32), dtype=torch.bfloat16, device="cuda"), True),
             (Wg, None, torch.empty((n_img * S, 192), dtype=torch.bfloat16, device="cuda"), False)]


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


def single():
    for w, b, o, planes in jobs:
        if planes:
            ops.value_proj_planes_ws(feat, w, b, o)
        else:
            ops.feat_linear_ws(feat, w, 192, out=o)


mb = lambda nj: (n_img * S * 512 + nj // 2 * n_img * S * (512 + 384)) / 1e6
print("%d images x %d pixels, %d layers" % (n_img, S, layers))
a = t(single)
print("one launch per product (%d launches): %7.1f us" % (len(jobs), a))
line = "grouped:"
for nl in (1, 2, 3, 4):
    if nl > layers:
        break
    us = t(lambda: ops.pyramid_group_ws(feat, jobs[:2 * nl]))
    line += "  %d layer%s %6.1f us (%4.0f GB/s min traffic)" % (nl, "s" if nl > 1 else " ", us, mb(2 * nl) / us * 1e3 / 1e3)
print(line)
for grid in (384, 448, 512):
    lib.mvg_set_tuning(b"wreg_grid", grid)
    us1, us3 = t(lambda: ops.pyramid_group_ws(feat, jobs[:2])), t(lambda: ops.pyramid_group_ws(feat, jobs[2:8]))
    print("wreg_grid %d: 1 layer %6.1f us, 3 layers %6.1f us" % (grid, us1, us3))
lib.mvg_set_tuning(b"wreg_grid", 512)
