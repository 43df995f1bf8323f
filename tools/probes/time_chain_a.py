"""reads the stamps of the instrumented chain A (tools/probes/instr_chain_a.py); 600 tiles of 128 rows, every row inside"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mvgformer_amd import ops
dev, bf = "cuda", torch.bfloat16
mk = lambda n, k: ops.swizzle_weight((torch.randn(n, k, device=dev) / 16).to(bf))
vec = lambda n: torch.randn(n, device=dev) * 0.1
Wp, W0, W1 = mk(256, 256), mk(256, 256), mk(256, 256)
W2 = torch.randn(3, 256, device=dev) / 16
bp, b0, b1, b2 = vec(256), vec(256), vec(256), vec(3)
rows = 600 * 128
samp = torch.randn(rows, 256, device=dev).to(bf)
inside = torch.ones(rows, dtype=torch.uint8, device=dev)
for _ in range(3):
    out = ops.chain_attn_pose(samp, inside, Wp, bp, W0, b0, W1, b1, W2, b2)
torch.cuda.synchronize()
attn = out[0]
d = attn.view(rows, 256)[::128].contiguous().view(torch.float32)[:, :11].cpu()
assert (d[:, 0] == -12345).all()
names = ["", "w2s + order + samp tile -> LDS (+ barrier inside)", "stage 1 GEMM (output_proj)", "bias, next weights, barrier, epilogue, barrier", "attn stores (issue)",
         "stage 2 GEMM (pose layer 0)", "bias, prefetch, barrier, epilogue, barrier", "stage 3 GEMM (pose layer 1) + barrier", "epilogue + barrier", "last layer (3 outputs) + DPP", "o store"]
tot = 0
for i in range(1, 11):
    c = d[:, i]; tot += c.mean().item()
    print("%-52s %8.0f %8.0f %8.0f" % (names[i], c.mean(), c.median(), c.quantile(0.9)))
print("total", tot)
