"""times the grouped pyramid products of whatever library MVG_LIB names (knock-out builds of tools/probes/ko_wreg.py). GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mvgformer_amd import _lib, ops
lib = _lib.load()
n_img, S = 5, 40320
feat = torch.randn(n_img, S, 256, device="cuda").to(torch.bfloat16)
jobs = []
for l in range(4):
    W = ops.swizzle_weight((torch.randn(256, 256, device="cuda") / 16).to(torch.bfloat16))
    Wg = ops.swizzle_weight((torch.randn(256, 256, device="cuda") / 16).to(torch.bfloat16))
    jobs += [(W, torch.randn(256, device="cuda"), torch.empty((n_img, 8, S, 32), dtype=torch.bfloat16, device="cuda"), True),
             (Wg, None, torch.empty((n_img * S, 192), dtype=torch.bfloat16, device="cuda"), False)]
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
v1 = t(lambda: ops.value_proj_planes_ws(feat, *jobs[0][:3]))
g1 = t(lambda: ops.feat_linear_ws(feat, jobs[1][0], 192, out=jobs[1][2]))
a = t(lambda: ops.pyramid_group_ws(feat, jobs[:2]))
b = t(lambda: ops.pyramid_group_ws(feat, jobs[2:8]))
print("%-22s value %6.1f  G %6.1f  group1 %6.1f  group3 %6.1f us" % (os.path.basename(os.environ.get("MVG_LIB", "product")), v1, g1, a, b))
