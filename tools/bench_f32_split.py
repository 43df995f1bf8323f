"""fp32 GEMM of the decoder (csrc/gemm.hip) in its two forms -- exact (v_mfma_f32_32x32x2_f32) and split (three bf16 parts per
operand value, six v_mfma_f32_32x32x16_bf16 per 16 k) -- against an fp64 product of the same fp32 operands: error and time.  GPU only.

    python tools/bench_f32_split.py            # decoder shapes, three operand distributions
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd import _lib, ops  # noqa: E402

dev = "cuda"


def timed(fn, n=10):
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


def knob(v):
    _lib.check(_lib.load().mvg_set_tuning(b"f32_split", int(v)), "f32_split")


def operands(kind, M, N, K, g):
    A = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) / 16
    if kind == "wide":           # 2^-20 .. 2^20 per value: the parts of a value span three binades of bf16 each
        A = A * torch.exp2(torch.randint(-20, 21, (M, K), device=dev, generator=g).float())
        W = W * torch.exp2(torch.randint(-20, 21, (N, K), device=dev, generator=g).float())
    elif kind == "positive":     # no cancellation: the sum's own magnitude is the yardstick
        A, W = A.abs(), W.abs()
    return A, W


def main():
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    rows = []
    shapes = ((201600, 256, 256), (201600, 192, 256), (76800, 256, 256), (15360, 1024, 256), (15360, 256, 1024))
    kinds = ("normal", "positive", "wide")
    if len(sys.argv) == 4:                 # one shape, timing only (for counter passes): python tools/bench_f32_split.py M N K
        shapes, kinds = (tuple(int(v) for v in sys.argv[1:4]),), ("normal",)
    for M, N, K in shapes:
        for kind in kinds:
            A, W = operands(kind, M, N, K, g)
            b = torch.randn(N, device=dev, generator=g)
            sub = slice(0, 16384)                      # error on the first 16 384 rows (fp64 product of 16 384 x N x K)
            ref = A[sub].double() @ W.double().t() + b.double()
            scale = A[sub].double().abs() @ W.double().abs().t() + b.double().abs()     # sum |a||w|: the fp32 error unit
            res = {}
            for name, v in (("exact", 0), ("split", 1)):
                knob(v)
                out = ops.linear(A, W, b, out_dtype=torch.float32)
                err = (out[sub].double() - ref).abs() / scale
                sec = timed(lambda: ops.linear(A, W, b, out_dtype=torch.float32)) if kind == "normal" else float("nan")
                res[name] = (err.max().item(), err.mean().item(), sec)
            knob(0)
            tf = lambda s: 2.0 * M * N * K / s / 1e12
            print("%7d x %4d x %4d %-8s  exact: max %.2e mean %.2e %7.1f us %6.1f TF   split: max %.2e mean %.2e %7.1f us %6.1f TF(eq)"
                  % (M, N, K, kind, res["exact"][0], res["exact"][1], res["exact"][2] * 1e6, tf(res["exact"][2]),
                     res["split"][0], res["split"][1], res["split"][2] * 1e6, tf(res["split"][2])))
            rows.append((M, N, K, kind, res))
    print("error unit: |out - fp64| / (sum_k |a||w| + |b|); 2^-24 = %.2e" % 2.0 ** -24)


if __name__ == "__main__":
    main()
