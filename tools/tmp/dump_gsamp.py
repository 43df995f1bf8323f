import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mvgformer_amd import _lib, ops
from mvgformer_amd.decoder import DecoderContext
from mvgformer_amd.factory import build_decoder_for_case, case_to_device
from mvgformer_amd.synthetic import build_case
case = build_case("cfg2", seed=0, layers=1)
dec = build_decoder_for_case(case, "cuda", torch.bfloat16)
g = case_to_device(case, "cuda")
pa = dec.layers[0].proj_attn
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
    out = ops.msda_gsamp(vp, G, xw, ref_lvl, ctx.levels, 1, pair_mask=msk, order=order)
    out2 = ops.msda_gsamp(vp, G, xw, ref_lvl, ctx.levels, 1, pair_mask=None, order=None)
    torch.save({"out": out.cpu(), "out2": out2.cpu(), "msk": msk.cpu()}, sys.argv[1])
