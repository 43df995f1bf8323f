"""phase times of pyramid_f32s_kernel (tiled) from s_memtime stamps, third tile of every workgroup (variant build: tools/ab_f32s.sh stamps)"""
import ctypes as C, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from mvgformer_amd import ops, _lib
dev = "cuda:0"
torch.manual_seed(0)
rows = 201600
feat = torch.randn(1, rows, 256, device=dev)
Wv, bv, Wg = torch.randn(256, 256, device=dev) / 16, torch.randn(256, device=dev), torch.randn(192, 256, device=dev) / 16
Wv_p, Wg_p = ops.split_swizzle_weight(Wv), ops.split_swizzle_weight(Wg)
lib = _lib.load()
lib.mvg_set_tuning(b"f32s_pyr_ws", 0)
for _ in range(3):
    ops.pyramid_f32s(feat, Wv_p, bv, Wg_p, 192)
torch.cuda.synchronize()
nb = 256
buf = (C.c_ulonglong * (64 * nb))()
lib.mvg_f32s_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mvg_f32s_read_stamps(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 64).astype(np.int64)
d = np.diff(t[:, :9], axis=1)
names = ["wait barrier (others finish stage G)", "split -> planes", "barrier", "stage value (16 k-steps)", "x loads + value stores", "stage G", "G stores", "loop back"]
for i in range(8):
    print("%-40s %8.0f  (%6.0f .. %6.0f)" % (names[i], np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
print("one tile: %.0f cycles (median), MFMA floor 2 x 6144 per SIMD" % np.median(t[:, 8] - t[:, 0]))
