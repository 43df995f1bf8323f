"""Workgroup timeline of the bf16 sampler INSIDE the graph-replayed forward (cfg-2): the -DGSAMP_STAMPS build of csrc/msda.hip
records, for every workgroup whose first wavefront sampled, start / end (s_memrealtime) and its CU.  GPU only.
    cd mvgformer_amd/csrc && hipcc <CXXFLAGS> -DGSAMP_STAMPS -c msda.hip -o ../../build/stamps/msda.o && hipcc --offload-arch=gfx950 -shared \
      -o ../../build/stamps/lib_gsamp_stamps.so api.o ../../build/stamps/msda.o geom.o gemm.o chain.o wreg_gemm.o msda_bwd.o f32s.o"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVG_LIB"] = os.path.join(ROOT, "build", "stamps", "lib_gsamp_stamps.so")
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from mvgformer_amd import _lib
from mvgformer_amd.decoder import DecoderContext
from mvgformer_amd.factory import build_decoder_for_case, case_to_device
from mvgformer_amd.synthetic import build_case
lib = _lib.load()
lib.mvg_gsamp_read_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.mvg_gsamp_read_stamps.restype = C.c_int
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
    for _ in range(50):
        graph.replay()
    torch.cuda.synchronize()
    assert lib.mvg_gsamp_read_stamps(None, 0, 1) >= 0
    graph.replay()
    torch.cuda.synchronize()
buf = (C.c_ulonglong * (4 * 65536))()
n = lib.mvg_gsamp_read_stamps(buf, 65536, 0)
assert n > 0, n
t = np.frombuffer(buf, dtype=np.uint64)[: n * 4].reshape(n, 4).astype(np.int64)
blk, act = t[:, 0] & 0xffffffff, t[:, 0] >> 32
hw, xcc = t[:, 3] & 0xffffffff, (t[:, 3] >> 32) & 0xf
cu = xcc * 256 + (((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5))
order = np.argsort(t[:, 1])
cuts = np.nonzero(np.diff(t[order, 1]) > 5000)[0] + 1
print("%d records, %d launches" % (n, len(cuts) + 1))
for li, idx in enumerate(np.split(order, cuts)):
    s0 = t[idx, 1].min()
    st, en = (t[idx, 1] - s0) / 100.0, (t[idx, 2] - s0) / 100.0
    span = en.max()
    print("launch %d: %d workgroups recorded on %d CUs; span %.1f us; workgroup life median %.1f p90 %.1f max %.1f us" % (
        li, len(idx), len(np.unique(cu[idx])), span, np.median(en - st), np.percentile(en - st, 90), (en - st).max()))
    # workgroups resident over time (10 bins) and per XCD end times
    edges = np.linspace(0, span, 14)
    res = [int(((st <= e) & (en > e)).sum()) for e in edges[:-1]]
    print("   resident recorded workgroups at %s us: %s" % (" ".join("%.0f" % e for e in edges[:-1]), res))
    print("   last end per XCD (us): %s" % " ".join("%.1f" % en[xcc[idx] == x].max() for x in range(8)))
    print("   starts after 90 %% of the span: %d; ends in the last 10 %%: %d" % ((st > 0.9 * span).sum(), (en > 0.9 * span).sum()))
