"""Achievable streaming rates on the box (torch elementwise kernels): read-only, write-only, copy.  GPU only."""
import torch

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
    return s.elapsed_time(e) / n * 1e-3

for mb in (64, 206, 412, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    r = t(lambda: a.sum())
    w = t(lambda: b.fill_(1.0))
    c = t(lambda: b.copy_(a))
    m = t(lambda: torch.mul(a, 2.0, out=b))
    print("%5d MB: read %6.0f GB/s   fill %6.0f GB/s   copy %6.0f GB/s (r+w)  mul-out %6.0f GB/s (r+w)"
          % (mb, n * 4 / r / 1e9, n * 4 / w / 1e9, 2 * n * 4 / c / 1e9, 2 * n * 4 / m / 1e9))
