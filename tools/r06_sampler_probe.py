"""Round-6 sampler experiments on the cfg-2 layer-0 inputs (isolated launches, HIP events):
  E1 occupancy caps through unused dynamic LDS (16 / 12 / 8 / 4 wavefronts per CU)
  E2 all-hit variant: G = 0 and xw = 0 -> every sample of a level sits on the reference point (4 lines per level and pair)
python tools/r06_sampler_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd import _lib, ops  # noqa: E402
from mvgformer_amd.decoder import DecoderContext  # noqa: E402
from mvgformer_amd.factory import build_decoder_for_case, case_to_device  # noqa: E402
from mvgformer_amd.synthetic import build_case  # noqa: E402

case = build_case(os.environ.get("AB_CONFIG", "cfg2"), seed=0, layers=1)
dec = build_decoder_for_case(case, "cuda", torch.bfloat16)
g = case_to_device(case, "cuda")
pa = dec.layers[0].proj_attn
lib = _lib.load()


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(n):
        fn()
    e_.record()
    torch.cuda.synchronize()
    return s_.elapsed_time(e_) / n * 1e3


with torch.no_grad():
    ctx = DecoderContext.build(g.src_views, g.spatial_shapes, g.level_start_index, g.meta, case.img_size, torch.bfloat16, 1)
    r, ref_lvl, inside = ops.project(g.reference_points, ctx.cams, ctx.levels, ctx.V, 1)
    x = (g.tgt + g.query_pos).contiguous()
    vp = pa.project_values(ctx.feat)
    Wq, bq = pa._fast_query_weights(torch.bfloat16)
    xw = ops.linear(x.reshape(-1, 256), Wq, bq, out_dtype=torch.float32)
    G = ops.feat_linear_ws(ctx.feat, pa.query_term_weights(torch.bfloat16)[0], 192)
    msk = inside.view(-1)
    order = ops.bin_pairs(ref_lvl, msk, ctx.levels)
    off = (xw.view(-1, 8, 24)[:, :, :16]).abs()
    print("pairs %d, in-image %.3f; |xw offsets| mean %.2f max %.2f px" % (msk.numel(), float(msk.float().mean()), float(off.mean()), float(off.max())))
    run = lambda G_=G, xw_=xw, m_=msk, o_=order: ops.msda_gsamp(vp, G_, xw_, ref_lvl, ctx.levels, 1, pair_mask=m_, order=o_)
    for rep in range(2):
        for pad in (0, 22 * 1024, 35 * 1024, 62 * 1024):   # 4 / 3 / 2 / 1 workgroups of 4 wavefronts per CU (20 KB static each)
            assert lib.mvg_set_tuning(b"gsamp_lds_pad", pad) == 0
            print("E1 lds pad %6d B: %7.1f us" % (pad, timed(run)))
        lib.mvg_set_tuning(b"gsamp_lds_pad", 0)
    G0, xw0 = torch.zeros_like(G), torch.zeros_like(xw)
    print("E2 all samples on the reference point (G = 0, xw = 0): %7.1f us  (real offsets: %.1f us)" % (timed(lambda: run(G0, xw0)), timed(run)))
    xw1 = xw.clone()
    xw1.view(-1, 8, 24)[:, :, :16] *= 0.25
    print("E2b offsets x 0.25: %7.1f us" % timed(lambda: run(G, xw1)))
    ones = torch.ones_like(msk)
    print("E3 no pair mask (all %d pairs sampled), binned order: %7.1f us" % (msk.numel(), timed(lambda: run(G, xw, ones, order))))
    print("E3b in-image pairs only, no order (pair order = query order): %7.1f us" % timed(lambda: run(G, xw, msk, None)))

# ---- E4: which share of the samples of in-image pairs lies outside its map (all four corner weights zero)?
with torch.no_grad():
    shapes = [(int(h), int(w)) for h, w in ctx.levels.shapes]
    starts = [int(v) for v in ctx.levels.starts]
    Lq = ref_lvl.shape[1]
    tot = out = border = 0
    for n in range(ctx.V):
        Gn = G.view(ctx.V, -1, 192)[n].float()
        keep = inside[n].bool().view(-1)
        ref = ref_lvl[n][keep]
        xwq = xw.view(1, Lq, 192)[0][keep]
        for m in range(8):
            offs = []
            for t in range(3):
                fg = m * 3 + t
                l, g_ = fg >> 3, fg & 7
                H, W = shapes[l]
                gx = (ref[:, l, 0] * 2 - 1).clamp(-1.1, 1.1)
                gy = (ref[:, l, 1] * 2 - 1).clamp(-1.1, 1.1)
                px, py = ((gx + 1) * W - 1) * 0.5, ((gy + 1) * H - 1) * 0.5
                x0, y0 = torch.floor(px), torch.floor(py)
                acc = 0
                for dy in (0, 1):
                    for dx in (0, 1):
                        xi, yi = x0 + dx, y0 + dy
                        w = (1 - (px - xi).abs()) * (1 - (py - yi).abs())
                        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
                        idx = starts[l] + yi.clamp(0, H - 1).long() * W + xi.clamp(0, W - 1).long()
                        acc = acc + (w * ok)[:, None] * Gn[idx][:, 24 * g_:24 * g_ + 16]
                offs.append(acc + xwq[:, 24 * g_:24 * g_ + 16])
            offs = torch.cat(offs, 1).view(-1, 24, 2)
            for i in range(24):
                l2 = i // 8
                H, W = shapes[l2]
                w_im = (ref[:, l2, 0] + offs[:, i, 0] / W) * W - 0.5
                h_im = (ref[:, l2, 1] + offs[:, i, 1] / H) * H - 0.5
                ins = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
                full = (h_im >= 0) & (w_im >= 0) & (h_im <= H - 1) & (w_im <= W - 1)
                tot += ins.numel()
                out += int((~ins).sum())
                border += int((ins & ~full).sum())
    print("E4 samples of in-image pairs: %d; outside their map (4 zero corners) %.2f %%; on the border (1-2 zero rows / columns) %.2f %%"
          % (tot, 100.0 * out / tot, 100.0 * border / tot))
