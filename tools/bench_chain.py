"""Micro-benchmark of the fused chains at cfg-2 size (random data): which phase of chain B costs what.  GPU only.
NOTE: a Python-driven loop issues a call every ~23 us -- shorter kernels are host-bound here; read their GPU
durations from `rocprofv3 --kernel-trace -- python tools/bench_chain.py` instead of the printed numbers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"
V, B, NQ, J = 5, 1, 1024, 15
rows = B * NQ * J
bf = torch.bfloat16
attn = torch.randn(V * rows, 256, device=dev).to(bf)
tgt = torch.randn(rows, 256, device=dev)
qpos = torch.randn(rows, 256, device=dev)
mk = lambda n, k: ops.swizzle_weight((torch.randn(n, k, device=dev) / 16).to(bf))
Wu, W1, W2, Wn = mk(256, 256), mk(1024, 256), mk(256, 1024), mk(256, 256)
vec = lambda n: torch.randn(n, device=dev) * 0.1
bu, b1, b2, bn = vec(256), vec(1024), vec(256), vec(256)
g2, be2, g3, be3 = vec(256) + 1, vec(256), vec(256) + 1, vec(256)
Wc, bc = torch.randn(2, 256, device=dev) / 16, vec(2)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def run(v=V, ffn=True, nxt=True):
    a = attn[: v * rows]
    return ops.chain_update_ffn_class(a, v, tgt, Wu, bu, g2, be2, W1, b1, W2, b2, g3, be3, Wc, bc, 0.1, B, NQ, J, None, ffn,
                                      next_query_proj=(qpos, Wn, bn, 192) if nxt else None)


print("chain B full                 %7.1f us" % t(lambda: run()))
print("chain B without next-xw tail %7.1f us" % t(lambda: run(nxt=False)))
print("chain B without FFN          %7.1f us" % t(lambda: run(ffn=False, nxt=False)))
print("chain B 1 view, no FFN       %7.1f us" % t(lambda: run(v=1, ffn=False, nxt=False)))
