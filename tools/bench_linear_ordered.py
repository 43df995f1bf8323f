import sys, torch
sys.path.insert(0, '/root/repo')
from mvgformer_amd import ops
dev = 'cuda'
M, N, K = 76800, 256, 256
a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / 16; b = torch.randn(N, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
zero = torch.zeros(N, device=dev)
print("plain            %.1f us" % t(lambda: ops.linear(a, w, b)))
for frac, shuffled in ((0.0, False), (0.33, False), (0.33, True), (0.66, False), (1.0, False)):
    inside = torch.ones(M, dtype=torch.uint8, device=dev)
    nm = int(M * frac)
    perm = torch.randperm(M, device=dev) if shuffled else torch.arange(M, device=dev)
    order = perm.to(torch.int32)
    if nm: inside[perm[M - nm:]] = 0            # the LAST nm slots of the order are masked
    us = t(lambda: ops.linear_ordered(a, w, b, order, inside, zero, rowmask=inside))
    got = ops.linear_ordered(a, w, b, order, inside, zero, rowmask=inside)
    want = ops.linear(a, w, b, rowmask=inside)
    print("ordered masked %.2f shuffled=%s  %.1f us   equal: %s" % (frac, shuffled, us, bool(torch.equal(got, want))))
