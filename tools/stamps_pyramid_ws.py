"""phase times of pyramid_ws_f32s_kernel from s_memtime stamps, fourth tile of every workgroup (tools/ab_f32s.sh stamps)"""
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
lib.mvg_set_tuning(b"f32s_pyr_ws", 1)
for _ in range(3):
    ops.pyramid_f32s(feat, Wv_p, bv, Wg_p, 192)
torch.cuda.synchronize()
nb = 256
buf = (C.c_ulonglong * (64 * nb))()
lib.mvg_f32s_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mvg_f32s_read_stamps(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 64).astype(np.int64)[:, 16:23]
d = np.diff(t, axis=1)
names = ["barrier", "k-steps 0-7", "k-steps 8-11 (split + DMA)", "k-steps 12-15", "stores", "loop back"]
for role in (0, 1):
    sel = ((np.arange(nb) >> 3) & 1) == role
    print("role", "G" if role else "value")
    for i in range(6):
        print("  %-30s %8.0f  (%6.0f .. %6.0f)" % (names[i], np.median(d[sel, i]), d[sel, i].min(), d[sel, i].max()))
    print("  one tile: %.0f cycles (median); MFMA floor 6144 per SIMD" % np.median(t[sel, 6] - t[sel, 0]))
