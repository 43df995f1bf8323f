"""Workgroup timeline and phase times of triangulate_kernel INSIDE the graph-replayed forward (cfg-2 bf16), where the launches of
layers 0-2 run next to the next layer's pyramid products: the -DTRI_STAMPS build of csrc/geom.hip.  GPU only.
    cd mvgformer_amd/csrc && hipcc <CXXFLAGS> -DTRI_STAMPS -c geom.hip -o ../../build/stamps/geom.o && hipcc --offload-arch=gfx950 -shared \
      -o ../../build/stamps/lib_tri_stamps.so api.o msda.o ../../build/stamps/geom.o gemm.o chain.o wreg_gemm.o msda_bwd.o f32s.o"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVG_LIB"] = os.path.join(ROOT, "build", "stamps", "lib_tri_stamps.so")
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from mvgformer_amd import _lib
from mvgformer_amd.decoder import DecoderContext
from mvgformer_amd.factory import build_decoder_for_case, case_to_device
from mvgformer_amd.synthetic import build_case
lib = _lib.load()
lib.mvg_tri_read_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.mvg_tri_read_stamps.restype = C.c_int
dev = torch.device("cuda", 0)
case = build_case("cfg2", B=1, seed=0, valid_fraction=(float(os.environ["VALID_FRACTION"]) if "VALID_FRACTION" in os.environ else None))
dec = build_decoder_for_case(case, dev, torch.bfloat16)
g = case_to_device(case, dev)
ctx = DecoderContext.prepare(g.spatial_shapes, g.level_start_index, g.meta, case.img_size, torch.bfloat16, 1, dev)
def forward():
    ctx.feat = None
    return dec(g.tgt, g.reference_points, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None, query_pos=g.query_pos,
               threshold=0.1, context=ctx)
with torch.no_grad():
    for _ in range(3):
        forward()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        forward()
    for _ in range(50):
        graph.replay()
    torch.cuda.synchronize()
    assert lib.mvg_tri_read_stamps(None, 0, 1) >= 0
    graph.replay()
    torch.cuda.synchronize()
buf = (C.c_ulonglong * (12 * 4096))()
n = lib.mvg_tri_read_stamps(buf, 4096, 0)
assert n > 0, n
t = np.frombuffer(buf, dtype=np.uint64)[: n * 12].reshape(n, 12).astype(np.int64)
order = np.argsort(t[:, 1])
cuts = np.nonzero(np.diff(t[order, 1]) > 5000)[0] + 1
print("%d records, %d launches" % (n, len(cuts) + 1))
names = ["loads + undistortion + DLT rows + Gram partials", "barrier", "Jacobi + divide + stores", "next layer's projection"]
for li, idx in enumerate(np.split(order, cuts)):
    s0 = t[idx, 1].min()
    st, en = (t[idx, 1] - s0) / 100.0, (t[idx, 2] - s0) / 100.0
    d = np.diff(t[idx][:, 4:9], axis=1)
    last = 8
    if (t[idx, 8] == 0).all():          # last layer: no next projection
        d, last = d[:, :3], 7
    clk = (t[idx, last] - t[idx, 4]) / np.maximum((en - st) * 1e3, 1)
    print("launch %d: %d workgroups; span %.1f us; start median %.1f max %.1f; life median %.1f max %.1f us; clock %.2f GHz" % (
        li, len(idx), en.max(), np.median(st), st.max(), np.median(en - st), (en - st).max(), np.median(clk)))
    print("   cycles (thread 0 = the solving wavefront): " + " | ".join("%s %d" % (nm, np.median(d[:, i])) for i, nm in enumerate(names[: d.shape[1]])))
    fb = t[idx, 9]
    print("   inverse iteration: %d cycles (median, barrier -> vector); wavefronts with a Jacobi fallback lane: %d of %d, fallback lanes %d of %d"
          % (np.median(t[idx, 10] - t[idx, 6]), int((fb > 0).sum()), len(idx), int(fb.sum()), 64 * len(idx)))
    print("   fallback lanes with unsafe pivots: %d" % int(t[idx, 11].sum()))
