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
