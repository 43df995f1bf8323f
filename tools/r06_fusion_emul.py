"""VERDICT r5 item 3, measured without building the fused kernel: what would a token-major chain A + B workgroup (64 tokens, its 5 views'
chain-A stages one after the other, then chain B, one workgroup per CU) cost?  Its chain-A part is the existing 64-row chain-A body run
with ONE workgroup per CU (probe knob chain_a_lds_pad), 5 (tile, view) units per CU for a token tile seen by all 5 views; chain B is
today's kernel (already one workgroup per CU).  python tools/r06_fusion_emul.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev, bf = "cuda", torch.bfloat16
mk = lambda n, k: ops.swizzle_weight((torch.randn(n, k, device=dev) / 16).to(bf))
vec = lambda n: torch.randn(n, device=dev) * 0.1
Wp, W0, W1 = mk(256, 256), mk(256, 256), mk(256, 256)
W2 = torch.randn(3, 256, device=dev) / 16
bp, b0, b1, b2 = vec(256), vec(256), vec(256), vec(3)


def graph_time(fn, n=10, reps=10):
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


def set_knobs(rm, pad):
    assert lib.mvg_set_tuning(b"chain_rm", rm) == 0 and lib.mvg_set_tuning(b"chain_a_lds_pad", pad) == 0


for label, units_per_cu in (("5 units per CU (a token tile seen by all 5 views: the workgroup that ends the launch)", 5),
                            ("3.14 units per CU (cfg-2's 804 computing (tile, view) units spread evenly)", 3.14)):
    rows = int(256 * units_per_cu) * 64
    samp = torch.randn(rows, 256, device=dev).to(bf)
    inside = torch.ones(rows, dtype=torch.uint8, device=dev)
    run = lambda: ops.chain_attn_pose(samp, inside, Wp, bp, W0, b0, W1, b1, W2, b2)
    print("== " + label + ": %d rows" % rows)
    for rm, pad, what in ((128, 0, "128-row tiles, 2 workgroups per CU (today's launch geometry)"), (64, 0, "64-row tiles, 2 workgroups per CU"),
                          (64, 46 * 1024, "64-row tiles, ONE workgroup per CU (the fused kernel's chain-A part)")):
        set_knobs(rm, pad)
        print("   %-70s %6.1f us" % (what, graph_time(run)))
    set_knobs(128, 0)
