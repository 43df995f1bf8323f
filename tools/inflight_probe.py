"""Throughput with K samples in flight: K independent decoder instances (own vh/G buffers), each forward captured as a
HIP graph on its own stream, replayed concurrently.  python tools/inflight_probe.py [K] [steps]  (GPU only)"""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd.decoder import DecoderContext  # noqa: E402
from mvgformer_amd.factory import build_decoder_for_case, case_to_device  # noqa: E402
from mvgformer_amd.synthetic import build_case  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dt = torch.bfloat16
slots = []
for k in range(K):
    case = build_case("cfg2", seed=k)
    dec = build_decoder_for_case(case, "cuda", dt)
    g = case_to_device(case, "cuda")
    ctx = DecoderContext.prepare(g.spatial_shapes, g.level_start_index, g.meta, case.img_size, dt, 1, "cuda")

    def fwd(dec=dec, g=g, ctx=ctx):
        ctx.feat = None
        return dec(g.tgt, g.reference_points, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None,
                   query_pos=g.query_pos, threshold=0.1, context=ctx)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(stream):
        for _ in range(3):
            fwd()
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            out = fwd()
    slots.append((stream, graph, out, (dec, g, ctx, case)))      # the graph holds raw pointers: keep their owners alive
    torch.cuda.synchronize()
    print("slot", k, "captured", flush=True)
torch.cuda.synchronize()


def step():
    for stream, graph, _, _ in slots:
        with torch.cuda.stream(stream):
            graph.replay()


for stream, graph, _, _ in slots:       # one at a time first
    with torch.cuda.stream(stream):
        graph.replay()
    torch.cuda.synchronize()
print("serial replays ok", flush=True)
for _ in range(10):
    step()
torch.cuda.synchronize()
print("concurrent replays ok", flush=True)
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("in flight %d: %.4f ms per step of %d samples = %.1f samples/s" % (K, el / steps * 1e3, K, K * steps / el))
