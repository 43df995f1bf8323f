"""Exact-fp32 and bf16 MFMA GEMM (csrc/gemm.hip: linear_kernel) at the decoder's shapes: TFLOP/s.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd import ops  # noqa: E402

dev = "cuda"


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


for M, N, K in ((201600, 256, 256), (201600, 192, 256), (76800, 256, 256), (15360, 1024, 256), (15360, 256, 1024), (230400, 192, 256)):
    for dt in (torch.float32, torch.bfloat16):
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, K, device=dev) / 16).to(dt)
        b = torch.randn(N, device=dev)
        sec = t(lambda: ops.linear(A, W, b, out_dtype=dt))
        ref = t(lambda: torch.nn.functional.linear(A, W, b.to(dt)))
        print("%7d x %4d x %4d %-8s  linear_kernel %7.1f us = %6.1f TFLOP/s    torch (rocBLAS / hipBLASLt) %7.1f us = %6.1f TFLOP/s"
              % (M, N, K, str(dt).split(".")[1], sec * 1e6, 2.0 * M * N * K / sec / 1e12, ref * 1e6, 2.0 * M * N * K / ref / 1e12))
