"""Marginal cost of every kernel of the graph-replayed forward: the captured cfg-2 bf16 forward is re-captured with ONE library
entry point at a time replaced by a no-op (its outputs keep the bytes of the previous forward, so everything downstream sees the
same data and does the same work) and timed against the full forward in alternation.  What a kernel costs the forward is what
the forward loses when it is gone -- not its stand-alone duration: kernels that share the chip with others hide part of it.
GPU only.    tools/ablate_forward.py [config] [dtype] [replays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mvgformer_amd import _lib
from mvgformer_amd.decoder import DecoderContext
from mvgformer_amd.factory import build_decoder_for_case, case_to_device
from mvgformer_amd.synthetic import build_case

config = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lib = _lib.load()
dev = torch.device("cuda", 0)
case = build_case(config, B=1, seed=0)
dec = build_decoder_for_case(case, dev, dtype)
g = case_to_device(case, dev)
ctx = DecoderContext.prepare(g.spatial_shapes, g.level_start_index, g.meta, case.img_size, dtype, 1, dev)


def forward():
    ctx.feat = None
    return dec(g.tgt, g.reference_points, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None, query_pos=g.query_pos,
               threshold=0.1, context=ctx)


def capture():
    with torch.no_grad():
        forward(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            forward()
        gr.replay(); torch.cuda.synchronize()
    return gr


def time_graph(gr, n=N):
    for _ in range(10):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    for _ in range(3):
        forward()
    torch.cuda.synchronize()
full = capture()
calls = {}
names = [n for n in _lib.SIGNATURES if n not in ("mvg_device_info", "mvg_set_tuning", "mvg_bin_pairs_workspace", "mvg_msda_backward_det_workspace")]
# which entry points does the forward use, and how often?
orig = {n: getattr(lib, n) for n in names}
for n in names:
    def counted(*a, _n=n, **k):
        calls[_n] = calls.get(_n, 0) + 1
        return orig[_n](*a, **k)
    setattr(lib, n, counted)
with torch.no_grad():
    forward(); torch.cuda.synchronize()
for n in names:
    setattr(lib, n, orig[n])
print("entry points of one forward:", calls)
base = time_graph(full)
print("%-34s %9.1f us" % ("full forward", base))
rows = []
variants = [(n,) for n in calls] + [("mvg_pyramid_group_ws", "mvg_pack_pyramid"), ("mvg_chain_attn_pose", "mvg_chain_update_ffn_class"),
                                    ("mvg_bin_pairs", "mvg_triangulate", "mvg_triangulate_project", "mvg_project")]
for ko in variants:
    if not all(k in calls for k in ko):
        continue
    for k in ko:
        setattr(lib, k, lambda *a, **kw: 0)
    try:
        gr = capture()
    finally:
        for k in ko:
            setattr(lib, k, orig[k])
    a1, b1, a2, b2 = time_graph(full), time_graph(gr), time_graph(full), time_graph(gr)
    label = " + ".join(k.replace("mvg_", "") for k in ko)
    print("without %-26s %9.1f us   (full %7.1f)   marginal cost %7.1f us   [%s launches]" % (
        label, (b1 + b2) / 2, (a1 + a2) / 2, (a1 + a2) / 2 - (b1 + b2) / 2, "+".join(str(calls[k]) for k in ko)), flush=True)
    del gr
