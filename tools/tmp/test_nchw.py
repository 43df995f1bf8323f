import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mvgformer_amd import ops
torch.manual_seed(0)
dev = "cuda"
def run(n_img, shapes):
    lv = ops.Levels([list(s) for s in shapes], [0] + list(torch.tensor([h * w for h, w in shapes]).cumsum(0)[:-1].tolist()))
    S = lv.S
    src = [torch.randn(n_img, 256, h, w, device=dev) for h, w in shapes]
    Wv = ops.swizzle_weight((torch.randn(256, 256, device=dev) / 16).to(torch.bfloat16))
    Wg = ops.swizzle_weight((torch.randn(256, 256, device=dev) / 16).to(torch.bfloat16))
    bv = torch.randn(256, device=dev)
    mk = lambda: [(Wv, bv, torch.zeros((n_img, 8, S, 32), dtype=torch.bfloat16, device=dev), True),
                  (Wg, None, torch.zeros((n_img * S, 192), dtype=torch.bfloat16, device=dev), False)]
    feat = ops.pack_pyramid(src, lv, torch.bfloat16)
    jobs_a = mk(); ops.pyramid_group_ws(feat, jobs_a, slots=32)
    assert ops.pyramid_group_ws_nchw_ok(src, lv, torch.bfloat16, 32)
    feat_b = torch.zeros_like(feat); jobs_b = mk()
    ops.pyramid_group_ws_nchw(src, lv, feat_b, jobs_b, 32)
    torch.cuda.synchronize()
    ok = torch.equal(feat.view(torch.int16), feat_b.view(torch.int16)), torch.equal(jobs_a[0][2].view(torch.int16), jobs_b[0][2].view(torch.int16)), torch.equal(jobs_a[1][2].view(torch.int16), jobs_b[1][2].view(torch.int16))
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    ta = t(lambda: (ops.pack_pyramid(src, lv, torch.bfloat16, out=feat), ops.pyramid_group_ws(feat, jobs_a, slots=32)))
    tb = t(lambda: ops.pyramid_group_ws_nchw(src, lv, feat_b, jobs_b, 32))
    print("n_img %d shapes %s: feat / planes / G identical: %s | pack + products %.1f us, fused %.1f us" % (n_img, shapes, ok, ta, tb))
    if not all(ok):
        d = (feat.float() - feat_b.float()).abs()
        print("   feat mismatches:", int((d > 0).sum()), "of", d.numel(), "first rows:", (d > 0).any(-1).nonzero()[:8].flatten().tolist())
        d2 = (jobs_a[1][2].float() - jobs_b[1][2].float()).abs()
        print("   G mismatching rows:", (d2 > 0).any(-1).nonzero()[:8].flatten().tolist(), int((d2 > 0).any(-1).sum()))
run(5, [(128, 240), (64, 120), (32, 60)])
run(3, [(152, 200), (76, 100), (40, 48)])
run(1, [(8, 8), (4, 4)])
run(2, [(16, 24), (8, 12), (4, 4)])
