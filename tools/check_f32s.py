"""fp32 fused kernels (csrc/f32s.hip) against fp64 torch on random operands + timing at cfg-2's shapes."""
import sys, time
import torch
sys.path.insert(0, ".")
from mvgformer_amd import ops

dev = "cuda:0"
torch.manual_seed(0)


def rel(a, b, scale):
    return float(((a.double() - b).abs() / scale).max())


def timeit(fn, n=20):
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


def lin64(x, W, b=None):
    y = x.double() @ W.double().t()
    return y if b is None else y + b.double()


def scale_of(x, W, b=None):
    s = x.double().abs() @ W.double().abs().t()
    return s + (0 if b is None else b.double().abs()) + 1e-30


full = len(sys.argv) > 1 and sys.argv[1] == "full"
# ---------------- pyramid
rows = 201600 if full else 5 * 777 + 13
feat = torch.randn(1, rows, 256, device=dev)
Wv, bv = torch.randn(256, 256, device=dev) / 16, torch.randn(256, device=dev)
Wg = torch.randn(192, 256, device=dev) / 16
Wv_p, Wg_p = ops.split_swizzle_weight(Wv), ops.split_swizzle_weight(Wg)
value, G = ops.pyramid_f32s(feat, Wv_p, bv, Wg_p, 192)
torch.cuda.synchronize()
n = min(rows, 20000)
f = feat[0, :n]
print("pyramid: value err %.2e  G err %.2e (units of sum|a||w| + |b|)" % (
    rel(value[0, :n], lin64(f, Wv, bv), scale_of(f, Wv, bv)), rel(G[:n], lin64(f, Wg), scale_of(f, Wg))))
f = feat[0, -n:]
print("pyramid tail: value err %.2e  G err %.2e" % (
    rel(value[0, -n:], lin64(f, Wv, bv), scale_of(f, Wv, bv)), rel(G[-n:], lin64(f, Wg), scale_of(f, Wg))))
v32 = ops.linear(feat.view(rows, 256), Wv.contiguous(), bv)
print("  vs mvg_linear split form: max abs diff %.2e" % float((v32 - value.view(rows, 256)).abs().max()))
if full:
    print("pyramid_f32s %.1f us  (mvg_linear x2: %.1f us)" % (
        timeit(lambda: ops.pyramid_f32s(feat, Wv_p, bv, Wg_p, 192, value, G)),
        timeit(lambda: (ops.linear(feat.view(rows, 256), Wv, bv), ops.linear(feat.view(rows, 256), Wg.contiguous(), None)))))

# ---------------- pyramid, two-part fp16 operands (three products)
(Wv_h, sv), (Wg_h, sg) = ops.split_swizzle_weight_h2(Wv), ops.split_swizzle_weight_h2(Wg)
for name, ft in (("randn", feat), ("rows over 12 decades", feat * torch.exp(torch.randn(1, rows, 1, device=dev) * 5)),
                 ("entries over 6 decades + zero rows", feat * torch.exp(torch.randn(1, rows, 256, device=dev) * 3) * (torch.rand(1, rows, 1, device=dev) > 0.1))):
    vh, Gh = ops.pyramid_f32h(ft, Wv_h, sv, bv, Wg_h, sg, 192)
    v6, G6 = ops.pyramid_f32s(ft, Wv_p, bv, Wg_p, 192)
    torch.cuda.synchronize()
    f = ft[0, :n]
    print("pyramid f32h [%s]: value err %.2e  G err %.2e   (six-product form on the same input: %.2e / %.2e)" % (
        name, rel(vh[0, :n], lin64(f, Wv, bv), scale_of(f, Wv, bv)), rel(Gh[:n], lin64(f, Wg), scale_of(f, Wg)),
        rel(v6[0, :n], lin64(f, Wv, bv), scale_of(f, Wv, bv)), rel(G6[:n], lin64(f, Wg), scale_of(f, Wg))))
    f = ft[0, -n:]
    print("   tail: value err %.2e  G err %.2e" % (rel(vh[0, -n:], lin64(f, Wv, bv), scale_of(f, Wv, bv)), rel(Gh[-n:], lin64(f, Wg), scale_of(f, Wg))))
if full:
    print("pyramid_f32h %.1f us" % timeit(lambda: ops.pyramid_f32h(feat, Wv_h, sv, bv, Wg_h, sg, 192, value, G)))

# ---------------- chain A
R = 76800 if full else 1000
samp = torch.randn(R, 256, device=dev)
inside = (torch.rand(R, device=dev) < 0.67).to(torch.uint8)
mk = lambda n, k: (torch.randn(n, k, device=dev) / k ** 0.5, torch.randn(n, device=dev) * 0.1)
(Wp, bp), (W0, b0), (W1, b1), (W2, b2) = mk(256, 256), mk(256, 256), mk(256, 256), mk(3, 256)
wts = (ops.split_swizzle_weight(Wp), bp, ops.split_swizzle_weight(W0), b0, ops.split_swizzle_weight(W1), b1, W2.contiguous(), b2)
o_masked = ops.chain_masked_row_output_f32s(*wts)
order = torch.argsort(1 - inside.int(), stable=True).to(torch.int32)       # masked rows last
for od, om in ((None, None), (order, o_masked)):
    attn, o = ops.chain_attn_pose_f32s(samp, inside, *wts, order=od, o_masked=om)
    torch.cuda.synchronize()
    a64 = lin64(samp, Wp, bp) * inside.double()[:, None]
    h = torch.relu(lin64(a64, W0, b0))
    h = torch.relu(lin64(h, W1, b1))
    o64 = lin64(h, W2, b2)
    print("chain A (order %s): attn err %.2e (abs %.2e)  o abs err %.2e  masked o equal %s" % (
        od is not None, rel(attn, a64, scale_of(samp, Wp, bp)), float((attn.double() - a64).abs().max()),
        float((o.double() - o64).abs().max()), bool((o[inside == 0] == o_masked).all())))
if full:
    print("chain_a_f32s %.1f us (ordered, skipping) / %.1f us (no order)" % (
        timeit(lambda: ops.chain_attn_pose_f32s(samp, inside, *wts, order=order, o_masked=o_masked)),
        timeit(lambda: ops.chain_attn_pose_f32s(samp, inside, *wts))))

# ---------------- chain A, two-part fp16 operands
sp2 = ops.split_swizzle_weight_h2
(Wp_h, swp), (W0_h, sw0), (W1_h, sw1) = sp2(Wp), sp2(W0), sp2(W1)
wts_h = (Wp_h, swp, bp, W0_h, sw0, b0, W1_h, sw1, b1, W2.contiguous(), b2)
o_masked_h = ops.chain_masked_row_output_f32h(*wts_h)
for od, om in ((None, None), (order, o_masked_h)):
    attn_h, o_h = ops.chain_attn_pose_f32h(samp, inside, *wts_h, order=od, o_masked=om)
    torch.cuda.synchronize()
    print("chain A f32h (order %s): attn err %.2e (abs %.2e)  o abs err %.2e  masked o equal %s" % (
        od is not None, rel(attn_h, a64, scale_of(samp, Wp, bp)), float((attn_h.double() - a64).abs().max()),
        float((o_h.double() - o64).abs().max()), bool((o_h[inside == 0] == o_masked_h).all())))
if full:
    print("chain_a_f32h %.1f us (ordered, skipping) / %.1f us (no order)" % (
        timeit(lambda: ops.chain_attn_pose_f32h(samp, inside, *wts_h, order=order, o_masked=o_masked_h)),
        timeit(lambda: ops.chain_attn_pose_f32h(samp, inside, *wts_h))))

# ---------------- chain B
B, NQ, J, V = (1, 1024, 15, 5) if full else (1, 37, 15, 3)
rows = B * NQ * J
attn = torch.randn(V * rows, 256, device=dev)
tgt = torch.randn(rows, 256, device=dev)
qpos = torch.randn(rows, 256, device=dev)
(Wu, bu), (Wf1, bf1), (Wf2, bf2), (Wc, bc) = mk(256, 256), mk(1024, 256), mk(256, 1024), mk(2, 256)
Wn, bn = mk(192, 256)
g2, be2, g3, be3 = (1 + 0.1 * torch.randn(256, device=dev) for _ in range(4))
bn_pad = torch.cat([bn, bn.new_zeros(64)])
args = (ops.split_swizzle_weight(Wu), bu, g2, be2, ops.split_swizzle_weight(Wf1), bf1, ops.split_swizzle_weight(Wf2), bf2, g3, be3,
        Wc.contiguous(), bc)
nxt = (qpos, ops.split_swizzle_weight(Wn), bn_pad, 192)
res = ops.chain_update_ffn_class_f32s(attn, V, tgt, *args, 0.5, B, NQ, J, next_query_proj=nxt)
torch.cuda.synchronize()
ln = lambda x, g, b: torch.nn.functional.layer_norm(x, (256,), g.double(), b.double(), 1e-5)
mean = attn.double().view(V, rows, 256).mean(0)
t1 = ln(tgt.double() + lin64(mean, Wu, bu), g2, be2)
y = ln(t1 + lin64(torch.relu(lin64(t1, Wf1, bf1)), Wf2, bf2), g3, be3)
pr = torch.sigmoid(lin64(y, Wc, bc)).view(B, NQ, J, 2).mean(2)
xw = lin64(y + qpos.double(), Wn, bn)
print("chain B: tgt' abs err %.2e  prob err %.2e  xw abs err %.2e  valid equal %s" % (
    float((res[0].double() - y).abs().max()), float((res[1].double() - pr).abs().max()), float((res[4].double() - xw).abs().max()),
    bool((res[2].bool() == (pr[..., 1] > 0.5)).all())))
# ---------------- chain B, two-part fp16 operands
(Wu_h, su), (Wf1_h, s1), (Wf2_h, s2), (Wn_h, sn) = sp2(Wu), sp2(Wf1), sp2(Wf2), sp2(Wn)
args_h = (Wu_h, su, bu, g2, be2, Wf1_h, s1, bf1, Wf2_h, s2, bf2, g3, be3, Wc.contiguous(), bc)
res_h = ops.chain_update_ffn_class_f32h(attn, V, tgt, *args_h, 0.5, B, NQ, J, next_query_proj=(qpos, Wn_h, sn, bn_pad, 192))
torch.cuda.synchronize()
print("chain B f32h: tgt' abs err %.2e  prob err %.2e  xw abs err %.2e  valid equal %s" % (
    float((res_h[0].double() - y).abs().max()), float((res_h[1].double() - pr).abs().max()), float((res_h[4].double() - xw).abs().max()),
    bool((res_h[2].bool() == (pr[..., 1] > 0.5)).all())))
if full:
    print("chain_b_f32h %.1f us" % timeit(lambda: ops.chain_update_ffn_class_f32h(attn, V, tgt, *args_h, 0.5, B, NQ, J, next_query_proj=(qpos, Wn_h, sn, bn_pad, 192))))
if full:
    print("chain_b_f32s %.1f us" % timeit(lambda: ops.chain_update_ffn_class_f32s(attn, V, tgt, *args, 0.5, B, NQ, J, next_query_proj=nxt)))
