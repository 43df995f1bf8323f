"""Phase timing of linear_kernel's split form (experiment build with -DMVG_EXP_TIME=1 only): per-workgroup s_memtime sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mvgformer_amd import _lib, ops
M, N, K = 201600, 256, 256
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda")
nwg = ((M + 127) // 128) * ((N + 127) // 128)
buf = torch.zeros((M + (nwg * 16 + N - 1) // N + 1, N), device="cuda")
_lib.check(_lib.load().mvg_set_tuning(b"f32_split", 1), "k")
for _ in range(3):
    ops.linear(A, W, b, out=buf[:M])
torch.cuda.synchronize()
d = buf[M:].reshape(-1)[: nwg * 16].reshape(nwg, 16).cpu()
names = ["gload issue", "compute (MFMA issue)", "barrier 1", "wait vmcnt(0)", "split + ds_write", "barrier 2", "prologue", "start->loop end", "tstart"]
print("workgroups", nwg, " (cycles, summed over the 8 slabs of a workgroup; mean / median / p90 over workgroups)")
for i, n in enumerate(names[:8]):
    c = d[:, i]
    print("%-22s %9.0f %9.0f %9.0f" % (n, c.mean(), c.median(), c.quantile(0.9)))
