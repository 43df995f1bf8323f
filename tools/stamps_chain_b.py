"""phase times of chain_b_f32s_kernel from s_memtime stamps (variant build: tools/ab_f32s.sh build stamps)"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from mvgformer_amd import ops, _lib
dev = "cuda:0"
torch.manual_seed(0)
B, NQ, J, V = 1, 1024, 15, 5
rows = B * NQ * J
mk = lambda n, k: (torch.randn(n, k, device=dev) / k ** 0.5, torch.randn(n, device=dev) * 0.1)
attn, tgt, qpos = torch.randn(V * rows, 256, device=dev), torch.randn(rows, 256, device=dev), torch.randn(rows, 256, device=dev)
(Wu, bu), (Wf1, bf1), (Wf2, bf2), (Wc, bc), (Wn, bn) = mk(256, 256), mk(1024, 256), mk(256, 1024), mk(2, 256), mk(192, 256)
g2, be2, g3, be3 = (1 + 0.1 * torch.randn(256, device=dev) for _ in range(4))
args = (ops.split_swizzle_weight(Wu), bu, g2, be2, ops.split_swizzle_weight(Wf1), bf1, ops.split_swizzle_weight(Wf2), bf2, g3, be3, Wc.contiguous(), bc)
nxt = (qpos, ops.split_swizzle_weight(Wn), torch.cat([bn, bn.new_zeros(64)]), 192)
for _ in range(5):
    ops.chain_update_ffn_class_f32s(attn, V, tgt, *args, 0.5, B, NQ, J, next_query_proj=nxt)
torch.cuda.synchronize()
lib = _lib.load()
nb = 256
buf = (C.c_ulonglong * (64 * nb))()
lib.mvg_f32s_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mvg_f32s_read_stamps(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 64).astype(np.int64)
d = np.diff(t[:, :42], axis=1)
names = ["mean->planes", "skew+upd GEMM", "LN2+planes"] + sum([["gap", "FFN1 c%d" % c, "epi+bar", "FFN2 c%d" % c] for c in range(8)], [])[1:] + ["gap", "y=..+t1", "LN3", "class head", "xw planes", "xw GEMM", "xw store"]
print("cycles per phase: median over 256 workgroups (min .. max); s_memtime ticks = 100 MHz?" )
for i in range(41):
    print("%2d %-16s %8.0f  (%6.0f .. %6.0f)" % (i, names[i] if i < len(names) else "?", np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
print("total %.0f (median)" % np.median(t[:, 41] - t[:, 0]), " start spread %d, end spread %d" % (t[:, 0].max() - t[:, 0].min(), t[:, 41].max() - t[:, 41].min()))
