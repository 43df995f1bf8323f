"""Shader clock and phase times of the grouped pyramid kernel INSIDE the graph-replayed forward (stamps build of
tools/probes/stamps_wreg.py): s_memtime (shader cycles) against s_memrealtime (100 MHz) over a workgroup's life.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVG_LIB"] = os.path.join(ROOT, "build", "ko_wreg", "lib_stamps.so")
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from mvgformer_amd import _lib
from mvgformer_amd.decoder import DecoderContext
from mvgformer_amd.factory import build_decoder_for_case, case_to_device
from mvgformer_amd.synthetic import build_case
lib = _lib.load()
dev = torch.device("cuda", 0)
case = build_case("cfg2", B=1, seed=0)
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
    for _ in range(200):
        graph.replay()
    torch.cuda.synchronize()
nb = 256
buf = (C.c_ulonglong * (64 * nb))()
lib.mvg_wreg_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mvg_wreg_read_stamps(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nb * 4, 16).astype(np.int64)
live = t[(t[:, 12] > 0) & (t[:, 14] > t[:, 13])]
real_ns = (live[:, 14] - live[:, 13]) * 10.0
ticks = live[:, 11] - live[:, 8]
print("last product launch of the forward (just-in-time schedule: one layer, one workgroup per CU; records of earlier launches with more workgroups may be mixed in): %d wavefronts, %.1f tiles each; lifetime %d ticks = %.1f us -> %.3f GHz; span %.1f us" % (
    len(live), np.median(live[:, 12]), np.median(ticks), np.median(real_ns) / 1e3, np.median(ticks / real_ns),
    (live[:, 14].max() - live[:, 13].min()) / 100.0))
print("per tile: %d ticks" % np.median((live[:, 10] - live[:, 9]) / live[:, 12]))
t = t[: nb * 4]
w = t[(t[:, 6] > 0) & (t[:, 12] > 0)]
d = np.diff(w[:, [0, 1, 2, 5, 6]], axis=1)
for i, nm in enumerate(("vmcnt wait", "barrier", "k loop", "epilogue")):
    print("  %-12s %6.0f (%6.0f .. %6.0f)" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
kq = w[(w[:, 3] > 0) & (w[:, 4] > 0) & (w[:, 7] > 0)]
if len(kq):
    dq = np.diff(kq[:, [2, 3, 4, 7, 5]], axis=1)
    print("  k loop in quarters (4 k-steps = 8 MFMAs = 256 cycles of issue each): " + " | ".join("%d" % np.median(dq[:, i]) for i in range(4)))
print("  tiles per wavefront: %s" % dict(zip(*np.unique(live[:, 12], return_counts=True))))
