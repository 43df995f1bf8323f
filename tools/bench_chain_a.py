"""Chain A (output projection x mask + pose MLP) against its row count: is a launch one "round" of tiles or two?  GPU only.
Every call is timed inside a HIP graph of 20 launches (a Python-driven loop is host-bound for kernels this short)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"
bf = torch.bfloat16
mk = lambda n, k: ops.swizzle_weight((torch.randn(n, k, device=dev) / 16).to(bf))
vec = lambda n: torch.randn(n, device=dev) * 0.1
Wp, W0, W1 = mk(256, 256), mk(256, 256), mk(256, 256)
W2 = torch.randn(3, 256, device=dev) / 16
bp, b0, b1, b2 = vec(256), vec(256), vec(256), vec(3)


def graph_time(fn, n=20, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3


tiles = [int(a) for a in sys.argv[1:]] or [128, 256, 384, 448, 512, 513, 520, 540, 600, 640, 768, 1024]
print("# chain A, every row live; 128-row tiles, 2 workgroups per CU -> 512 resident tiles")
for nt in tiles:
    rows = nt * 128
    samp = torch.randn(rows, 256, device=dev).to(bf)
    inside = torch.ones(rows, dtype=torch.uint8, device=dev)
    t = graph_time(lambda: ops.chain_attn_pose(samp, inside, Wp, bp, W0, b0, W1, b1, W2, b2))
    print("%5d tiles (%6d rows): %6.1f us   %5.1f ns per row" % (nt, rows, t, t * 1e3 / rows))
