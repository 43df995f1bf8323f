"""Workgroup timeline of chain A / chain B INSIDE the graph-replayed forward (cfg-2 bf16): every workgroup of the -DCHAIN_STAMPS build of
csrc/chain.hip records start / end (s_memrealtime), its CU and phase stamps (s_memtime).  GPU only.

    cd mvgformer_amd/csrc && mkdir -p ../../build/stamps && hipcc <CXXFLAGS of the Makefile> -DCHAIN_STAMPS -c chain.hip -o ../../build/stamps/chain.o \
      && hipcc --offload-arch=gfx950 -shared -o ../../build/stamps/lib_chain_stamps.so api.o msda.o geom.o gemm.o ../../build/stamps/chain.o \
         wreg_gemm.o msda_bwd.o f32s.o"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVG_LIB"] = os.path.join(ROOT, "build", "stamps", "lib_chain_stamps.so")
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from mvgformer_amd import _lib
from mvgformer_amd.decoder import DecoderContext
from mvgformer_amd.factory import build_decoder_for_case, case_to_device
from mvgformer_amd.synthetic import build_case
lib = _lib.load()
lib.mvg_chain_read_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.mvg_chain_read_stamps.restype = C.c_int
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
    assert lib.mvg_chain_read_stamps(None, 0, 1) >= 0
    graph.replay()
    torch.cuda.synchronize()
buf = (C.c_ulonglong * (16 * 8192))()
n = lib.mvg_chain_read_stamps(buf, 8192, 0)
assert n > 0, n
t = np.frombuffer(buf, dtype=np.uint64)[: n * 16].reshape(n, 16).astype(np.int64)
kid, live, blk = t[:, 0] >> 48, (t[:, 0] >> 32) & 0xffff, t[:, 0] & 0xffffffff
hw, xcc = t[:, 3] & 0xffffffff, (t[:, 3] >> 32) & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)       # cu_id | sh_id | se_id
cuid = xcc * 256 + cu
print("%d records" % n)
for k, name in ((1, "chain A"), (2, "chain B")):
    rows = np.nonzero(kid == k)[0]
    if not len(rows):
        continue
    # launches: split at gaps of the start times
    order = rows[np.argsort(t[rows, 1])]
    st = t[order, 1]
    cuts = np.nonzero(np.diff(st) > 5000)[0] + 1          # > 50 us apart
    for li, idx in enumerate(np.split(order, cuts)):
        s0 = t[idx, 1].min()
        start = (t[idx, 1] - s0) / 100.0
        end = (t[idx, 2] - s0) / 100.0
        lv = live[idx] == 1
        life = end - start
        ncu = len(np.unique(cuid[idx][lv]))
        per_cu = np.bincount(np.unique(cuid[idx][lv], return_inverse=True)[1])
        print("%s launch %d: %d workgroups (%d computing) on %d CUs (computing tiles per CU: %s); span %.1f us" % (
            name, li, len(idx), lv.sum(), ncu, dict(zip(*np.unique(per_cu, return_counts=True))), end.max()))
        print("   computing: start median %.1f max %.1f us | life median %.1f p90 %.1f max %.1f us | end median %.1f p90 %.1f max %.1f" % (
            np.median(start[lv]), start[lv].max(), np.median(life[lv]), np.quantile(life[lv], 0.9), life[lv].max(),
            np.median(end[lv]), np.quantile(end[lv], 0.9), end[lv].max()))
        if (~lv).any():
            print("   skipped tiles: start median %.1f max %.1f, life median %.2f us" % (np.median(start[~lv]), start[~lv].max(), np.median(life[~lv])))
        if k == 1:
            ph = t[idx][lv][:, 4:13]
            d = np.diff(ph, axis=1)
            clk = (ph[:, 8] - ph[:, 0]) / np.maximum(life[lv] * 1e3, 1)
            names = ["tile load", "stage 1", "epilogue 1 + attn stores", "stage 2", "epilogue 2", "stage 3", "epilogue 3", "last layer + o"]
            # tiles alone on their CU against tiles that share it with another computing tile
            cl = cuid[idx][lv]
            cnt = dict(zip(*np.unique(cl, return_counts=True)))
            alone = np.array([cnt[c] == 1 for c in cl])
            for tag, m in (("alone on the CU", alone), ("two per CU", ~alone)):
                if m.any():
                    print("   %-16s (%3d tiles, life %.1f us) cycles: " % (tag, m.sum(), np.median(life[lv][m])) +
                          " | ".join("%s %d" % (nm, np.median(d[m, i])) for i, nm in enumerate(names)) + " | clock %.2f GHz" % np.median(clk[m]))
        else:
            ph = t[idx][lv][:, 4:16]
            d = np.diff(ph, axis=1)
            clk = (ph[:, 11] - ph[:, 0]) / np.maximum(life[lv] * 1e3, 1)
            names = ["load + view mean", "update GEMM", "LN2", "FFN 0", "FFN 1", "FFN 2", "FFN 3", "residual + barrier", "LN3 + class head",
                     "query term", "validity"]
            print("   cycles: " + " | ".join("%s %d" % (nm, np.median(d[:, i])) for i, nm in enumerate(names)) + " | clock %.2f GHz" % np.median(clk))
        # by life histogram of starts
        hist, edges = np.histogram(start[lv], bins=8)
        print("   start histogram (us): " + " ".join("%.0f-%.0f:%d" % (edges[i], edges[i + 1], hist[i]) for i in range(8)))
# dispatch detail of the last chain A launch: who started late, and how many workgroups every XCD took in the first 2 us
if os.environ.get("STAMPS_DETAIL"):
    rows = np.nonzero(kid == 1)[0]
    order = rows[np.argsort(t[rows, 1])]
    cuts = np.nonzero(np.diff(t[order, 1]) > 5000)[0] + 1
    idx = np.split(order, cuts)[-1]
    s0 = t[idx, 1].min()
    start = (t[idx, 1] - s0) / 100.0
    end = (t[idx, 2] - s0) / 100.0
    early = start < 2.0
    print("per XCD: workgroups started < 2 us / all; distinct CUs; max blockIdx started early")
    for x in range(8):
        m = xcc[idx] == x
        print("  xcd %d: %d / %d, %d CUs, early blockIdx max %d, late live blockIdx %s" % (
            x, (m & early).sum(), m.sum(), len(np.unique(cu[idx][m])), blk[idx][m & early].max(),
            sorted(blk[idx][m & ~early & (live[idx] == 1)].tolist())))
    m = (xcc[idx] == 0)
    o = np.argsort(start[m])
    print("xcd 0 in start order: (blockIdx, cu, live, start, end)")
    print([(int(b), int(c), int(l), round(float(s), 1), round(float(e), 1)) for b, c, l, s, e in
           zip(blk[idx][m][o], cu[idx][m][o], live[idx][m][o], start[m][o], end[m][o])])
