"""Micro-benchmark of msda_fused on the cfg-2 layer-0 inputs (real offsets/logits from the
synthetic weights), A/B over the tuning knobs.  GPU only.  python tools/bench_msda.py [dtype]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd import _lib, ops  # noqa: E402
from mvgformer_amd.decoder import DecoderContext  # noqa: E402
from mvgformer_amd.factory import build_decoder_for_case, case_to_device  # noqa: E402
from mvgformer_amd.synthetic import build_case  # noqa: E402

dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
case = build_case("cfg2", seed=0, layers=1)
dec = build_decoder_for_case(case, "cuda", dtype)
g = case_to_device(case, "cuda")
layer = dec.layers[0]
pa = layer.proj_attn
lib = _lib.load()
with torch.no_grad():
    ctx = DecoderContext.build(g.src_views, g.spatial_shapes, g.level_start_index, g.meta, case.img_size, dtype, 1)
    r, ref_lvl, inside = ops.project(g.reference_points, ctx.cams, ctx.levels, ctx.V, 1)
    x = (g.tgt + g.query_pos).contiguous()
    Wv, bv, Woa, boa, Wp, bp = pa.weights(dtype)
    ain = ops.gather_ref(ctx.feat, ref_lvl, x, ctx.levels, ctx.V, 1)
    value = ops.linear(ctx.feat.view(-1, 256), Wv, bv, out_dtype=dtype).view(ctx.V, -1, 256)
    oa = ops.linear(ain, Woa, boa, out_dtype=torch.float32)
    elem = 2 if dtype == torch.bfloat16 else 4
    Lq, S = 15360, ctx.levels.S
    bytes_launch = ctx.V * (S * 256 * elem + Lq * 8 * 3 * 8 * 3 * elem + Lq * 256 * elem)

    def run(tag, n=20):
        for _ in range(3):
            ops.msda_fused(value, oa, ref_lvl, ctx.levels)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            out = ops.msda_fused(value, oa, ref_lvl, ctx.levels)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / n * 1e3
        print("%-28s %8.1f us  %7.1f GB/s algorithmic" % (tag, us, bytes_launch / us / 1e3))
        return out

    if dtype == torch.bfloat16:
        vp = pa.project_values(ctx.feat)
        Wq, bq = pa._fast_query_weights(dtype)
        xw = ops.linear(x.reshape(-1, 256), Wq, bq, out_dtype=torch.float32)
        Woa_f = pa.query_term_weights(dtype)[0]
        G = ops.feat_linear_ws(ctx.feat, Woa_f, 192)
        for _ in range(3):
            ops.msda_gsamp(vp, G, xw, ref_lvl, ctx.levels, 1)
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(20):
            outp = ops.msda_gsamp(vp, G, xw, ref_lvl, ctx.levels, 1)
        e_.record()
        torch.cuda.synchronize()
        us = s_.elapsed_time(e_) / 20 * 1e3
        print("%-28s %8.1f us  %7.1f GB/s algorithmic" % ("G-sampling (fast path)", us, bytes_launch / us / 1e3))
        msk = inside.view(-1)
        print("    in-image fraction of the pairs: %.3f" % float(msk.float().mean()))
        order = ops.bin_pairs(ref_lvl, msk, ctx.levels)
        order_all = ops.bin_pairs(ref_lvl, None, ctx.levels)
        for tag, kw in (("  + pair mask", dict(pair_mask=msk)), ("  + Morton order", dict(order=order_all)),
                        ("  + mask + Morton order", dict(pair_mask=msk, order=order))):
            for _ in range(3):
                ops.msda_gsamp(vp, G, xw, ref_lvl, ctx.levels, 1, **kw)
            torch.cuda.synchronize()
            s_.record()
            for _ in range(20):
                o2 = ops.msda_gsamp(vp, G, xw, ref_lvl, ctx.levels, 1, **kw)
            e_.record()
            torch.cuda.synchronize()
            us = s_.elapsed_time(e_) / 20 * 1e3
            keep = msk.bool() if "pair_mask" in kw else torch.ones_like(msk).bool()
            same = bool((o2[keep] == outp[keep]).all()) and bool((o2[~keep] == 0).all())
            print("%-28s %8.1f us  %7.1f GB/s algorithmic   identical rows: %s" % (tag, us, bytes_launch / us / 1e3, same))
        # locality probes: where does the time go when (a) every pair samples the same place (all L1 hits), (b) the
        # reference points are uniformly random (sorted / unsorted)?
        def timed(tag, rr, **kw):
            for _ in range(3):
                ops.msda_gsamp(vp, G, xw, rr, ctx.levels, 1, **kw)
            torch.cuda.synchronize()
            s_.record()
            for _ in range(20):
                ops.msda_gsamp(vp, G, xw, rr, ctx.levels, 1, **kw)
            e_.record()
            torch.cuda.synchronize()
            print("%-28s %8.1f us" % (tag, s_.elapsed_time(e_) / 20 * 1e3))
        same = torch.full_like(ref_lvl, 0.5)
        timed("  probe: all pairs at (.5,.5)", same)
        rnd = torch.rand_like(ref_lvl[:, :, :1, :]).expand_as(ref_lvl).contiguous()
        timed("  probe: random refs", rnd)
        timed("  probe: random refs, sorted", rnd, order=ops.bin_pairs(rnd, None, ctx.levels))
        half = ref_lvl.clone(); half[:, :, :, :] = ref_lvl[:, :1, :, :] * 0 + torch.rand_like(ref_lvl[:, :, :1, :]) * 0.25 + 0.3
        timed("  probe: refs in a 1/16 window", half.contiguous())
        s_.record()
        for _ in range(20):
            ops.bin_pairs(ref_lvl, msk, ctx.levels)
        e_.record()
        torch.cuda.synchronize()
        print("%-28s %8.1f us" % ("  bin_pairs", s_.elapsed_time(e_) / 20 * 1e3))
        ref = ops.msda_fused(value, oa, ref_lvl, ctx.levels)
        d = (outp.float() - ref.float()).abs()
        print("    vs generic fused kernel: max |diff| %.3e  mean %.3e  (|ref| max %.2f)" % (float(d.max()), float(d.mean()), float(ref.float().abs().max())))
    run("generic fused kernel")
