"""sha256 + time of the fp32 G-sampling kernel on cfg-2's layer-0 inputs (old / new builds through MVG_LIB)"""
import hashlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mvgformer_amd import ops
from mvgformer_amd.decoder import DecoderContext
from mvgformer_amd.factory import build_decoder_for_case, case_to_device
from mvgformer_amd.synthetic import build_case
cfg = os.environ.get("AB_CONFIG", "cfg2")
case = build_case(cfg, seed=0, layers=1)
dec = build_decoder_for_case(case, "cuda", torch.float32)
g = case_to_device(case, "cuda")
pa = dec.layers[0].proj_attn
with torch.no_grad():
    ctx = DecoderContext.build(g.src_views, g.spatial_shapes, g.level_start_index, g.meta, case.img_size, torch.float32, 1)
    r, ref_lvl, inside = ops.project(g.reference_points, ctx.cams, ctx.levels, ctx.V, 1)
    x = (g.tgt + g.query_pos).contiguous()
    Wq, bq = pa._fast_query_weights(torch.float32)
    xw = ops.linear(x.reshape(-1, 256), Wq, bq, out_dtype=torch.float32)
    value, G = pa.project_pyramid(ctx.feat)
    msk = inside.view(-1)
    order = ops.bin_pairs(ref_lvl, msk, ctx.levels)
    for tag, kw in (("masked + ordered", dict(pair_mask=msk, order=order)), ("plain", dict())):
        for _ in range(3):
            out = ops.msda_gfused_f32(value, G, xw, ref_lvl, ctx.levels, 1, **kw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            out = ops.msda_gfused_f32(value, G, xw, ref_lvl, ctx.levels, 1, **kw)
        b.record(); torch.cuda.synchronize()
        print("%s %-18s %7.1f us  sha %s" % (cfg, tag, a.elapsed_time(b) / 20 * 1e3, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]))
