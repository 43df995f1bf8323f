"""Phase times inside pyramid_ws_f32h_kernel (csrc/f32s.hip) from s_memtime stamps of one steady-state tile per wavefront.
  tools/probes/stamps_pyr_ws.py build   (here)      tools/probes/stamps_pyr_ws.py   (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "mvgformer_amd", "csrc")
OUT = os.path.join(ROOT, "build", "ko_wreg")
LIB = os.path.join(OUT, "lib_pyrws_stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-ffp-contract=fast", "-fno-slp-vectorize"]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DPYRWS_STAMPS=20", "-c", os.path.join(CSRC, "f32s.hip"), "-o", os.path.join(OUT, "f32s_stamps.o")])
    objs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "f32s.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", LIB, os.path.join(OUT, "f32s_stamps.o")] + objs)
    sys.exit(0)
os.environ["MVG_LIB"] = LIB
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from mvgformer_amd import _lib, ops
lib = _lib.load()
feat = torch.randn(5, 40320, 256, device="cuda")
Wv, bv, Wg = torch.randn(256, 256, device="cuda") / 16, torch.randn(256, device="cuda"), torch.randn(192, 256, device="cuda") / 16
(Wv_h, sv), (Wg_h, sg) = ops.split_swizzle_weight_h2(Wv), ops.split_swizzle_weight_h2(Wg)
value = torch.empty(5, 40320, 256, device="cuda"); G = torch.empty(5 * 40320, 192, device="cuda")
import time
for _ in range(3):
    ops.pyramid_f32h(feat, Wv_h, sv, bv, Wg_h, sg, 192, value=value, G=G)
torch.cuda.synchronize(); time.sleep(0.3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.pyramid_f32h(feat, Wv_h, sv, bv, Wg_h, sg, 192, value=value, G=G); e1.record(); torch.cuda.synchronize()
print("one launch on an idle chip: %.1f us" % (e0.elapsed_time(e1) * 1e3))
nb = 256
buf = (C.c_ulonglong * (128 * nb))()
lib.mvg_pyrws_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mvg_pyrws_read_stamps(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nb * 8, 16).astype(np.int64)
t = t[t[:, 7] > t[:, 0]]
d = np.diff(t[:, :8], axis=1)
names = ["barrier", "k steps 0-3 (+ stage reads, stores)", "k step 4", "k steps 5-9 (+ split of the next tile)", "k steps 10-11 (+ row loads)",
         "k steps 12-15", "epilogue -> staging"]
print("%d wavefronts with columns; cycles median (p10 .. p90)" % len(t))
for i in range(7):
    print("  %-42s %7.0f (%6.0f .. %6.0f)" % (names[i], np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("  one tile: %.0f" % np.median(t[:, 7] - t[:, 0]))
