"""Phase times inside wreg2_body (csrc/wreg_gemm.hip) from s_memtime stamps of one steady-state tile per wavefront.
  tools/probes/stamps_wreg.py build     (here: hipcc cross-compiles build/ko_wreg/lib_stamps.so)
  tools/probes/stamps_wreg.py           (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "mvgformer_amd", "csrc")
OUT = os.path.join(ROOT, "build", "ko_wreg")
LIB = os.path.join(OUT, "lib_stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-ffp-contract=fast", "-fno-slp-vectorize"]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DWREG_STAMPS=%s" % (sys.argv[2] if len(sys.argv) > 2 else "20"), "-c", os.path.join(CSRC, "wreg_gemm.hip"), "-o", os.path.join(OUT, "wreg_stamps.o")])
    objs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "wreg_gemm.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", LIB, os.path.join(OUT, "wreg_stamps.o")] + objs)
    sys.exit(0)
os.environ["MVG_LIB"] = LIB
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from mvgformer_amd import _lib, ops
lib = _lib.load()
n_img, S = 5, 40320
feat = torch.randn(n_img, S, 256, device="cuda").to(torch.bfloat16)
jobs = []
for l in range(3):
    W = ops.swizzle_weight((torch.randn(256, 256, device="cuda") / 16).to(torch.bfloat16))
    Wg = ops.swizzle_weight((torch.randn(256, 256, device="cuda") / 16).to(torch.bfloat16))
    jobs += [(W, torch.randn(256, device="cuda"), torch.empty((n_img, 8, S, 32), dtype=torch.bfloat16, device="cuda"), True),
             (Wg, None, torch.empty((n_img * S, 192), dtype=torch.bfloat16, device="cuda"), False)]
names = ["vmcnt wait", "barrier", "-", "-", "k loop (32 MFMA + stores + DMA + bias reads)", "epilogue -> staging", "loop back"]
for label, fn in (("group of 1 layer", lambda: ops.pyramid_group_ws(feat, jobs[:2])), ("group of 3 layers", lambda: ops.pyramid_group_ws(feat, jobs))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    import time; time.sleep(0.2)
    fn(); torch.cuda.synchronize()        # (a launch on an idle chip: clocks as the forward would find them after a pause)
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    nb = 512
    buf = (C.c_ulonglong * (64 * nb))()
    lib.mvg_wreg_read_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.mvg_wreg_read_stamps(buf, nb) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(nb * 4, 16).astype(np.int64)
    live = t[(t[:, 12] > 0) & (t[:, 14] > t[:, 13])]
    real_ns = (live[:, 14] - live[:, 13]) * 10.0          # s_memrealtime: 100 MHz
    ticks = live[:, 11] - live[:, 8]
    print("== %s: %d wavefronts; lifetime %d ticks = %.1f us by the 100-MHz counter -> %.3f ticks per ns (s_memtime rate)" % (
        label, len(live), np.median(ticks), np.median(real_ns) / 1e3, np.median(ticks / real_ns)))
    print("  weights in regs +%d | loop %d ticks for %.1f tiles = %d per tile | drain +%d; span of all wavefronts: %.1f us" % (
        np.median(live[:, 9] - live[:, 8]), np.median(live[:, 10] - live[:, 9]),
        np.median(live[:, 12]), np.median((live[:, 10] - live[:, 9]) / live[:, 12]), np.median(live[:, 11] - live[:, 10]),
        (live[:, 14].max() - live[:, 13].min()) / 100.0))
    ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev[0].record(); fn(); ev[1].record(); torch.cuda.synchronize()
    print("  kernel (events, one launch): %.1f us" % (ev[0].elapsed_time(ev[1]) * 1e3))
    t = t[t[:, 6] > 0]                      # wavefronts with columns
    t[:, 3] = t[:, 2]; t[:, 4] = t[:, 2]
    d = np.diff(t[:, :7], axis=1)
    print("== %s: %d wavefronts, cycles (s_memtime ticks = 100 MHz? see below) median (p10 .. p90)" % (label, len(t)))
    for i in range(6):
        print("  %-30s %8.0f  (%6.0f .. %6.0f)" % (names[i], np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
    print("  one tile, top to end of epilogue: %.0f (median)" % np.median(t[:, 6] - t[:, 0]))
