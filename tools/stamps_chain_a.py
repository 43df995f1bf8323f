"""phase times of chain_a_f32s_kernel from s_memtime stamps, second tile of every workgroup, all rows inside (tools/ab_f32s.sh stamps)"""
import ctypes as C, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from mvgformer_amd import ops, _lib
dev = "cuda:0"
torch.manual_seed(0)
R = 76800
samp = torch.randn(R, 256, device=dev)
inside = torch.ones(R, device=dev, dtype=torch.uint8)
mk = lambda n, k: (torch.randn(n, k, device=dev) / k ** 0.5, torch.randn(n, device=dev) * 0.1)
(Wp, bp), (W0, b0), (W1, b1), (W2, b2) = mk(256, 256), mk(256, 256), mk(256, 256), mk(3, 256)
wts = (ops.split_swizzle_weight(Wp), bp, ops.split_swizzle_weight(W0), b0, ops.split_swizzle_weight(W1), b1, W2.contiguous(), b2)
for _ in range(3):
    ops.chain_attn_pose_f32s(samp, inside, *wts)
torch.cuda.synchronize()
lib = _lib.load()
nb = 256
buf = (C.c_ulonglong * (64 * nb))()
lib.mvg_f32s_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mvg_f32s_read_stamps(buf, nb) == 0
tall = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 64).astype(np.int64)
e = tall[:, [6, 20, 21, 22, 7]]
de = np.diff(e, axis=1)
for i, nm in enumerate(["  epilogue 2: bias + ring prefetch issue", "  epilogue 2: barrier (all stages done)", "  epilogue 2: bias wait + split + LDS writes", "  epilogue 2: barrier"]):
    print("%-44s %8.0f  (%6.0f .. %6.0f)" % (nm, np.median(de[:, i]), de[:, i].min(), de[:, i].max()))
t = tall[:, :11]
d = np.diff(t, axis=1)
names = ["order / inside -> rid, barrier", "samp rows -> planes, barrier", "stage 1 (output_proj)", "epilogue 1 + barriers", "attn rows -> global",
         "stage 2", "epilogue 2 + barriers", "stage 3", "epilogue 3 + barriers", "last layer (3 outputs)"]
for i in range(10):
    print("%-34s %8.0f  (%6.0f .. %6.0f)" % (names[i], np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
print("one tile: %.0f cycles (median); MFMA floor 3 x 12288 per SIMD" % np.median(t[:, 10] - t[:, 0]))

w = tall[:, 40:48] - tall[:, 32:40]
st = tall[:, 32:40] - tall[:, 32:33]
en = tall[:, 40:48] - tall[:, 32:33]
print("stage 2 per wavefront (median over workgroups): start offset / duration / end offset, cycles relative to wavefront 0's start")
for k in range(8):
    print("  wavefront %d: %6.0f %6.0f %6.0f" % (k, np.median(st[:, k]), np.median(w[:, k]), np.median(en[:, k])))
