import sys; sys.path.insert(0, "/root/repo")
import torch
from mvgformer_amd import ops
from mvgformer_amd.factory import build_decoder_for_case, case_to_device
from mvgformer_amd.synthetic import build_case
vf = float(sys.argv[1]) if len(sys.argv) > 1 else None
case = build_case("cfg2", seed=0, valid_fraction=vf)
dec = build_decoder_for_case(case, "cuda", torch.bfloat16)
g = case_to_device(case, "cuda")
orig = ops.project
def proj(*a, **k):
    r, ref_lvl, inside = orig(*a, **k)
    print("inside fraction %.3f" % float(inside.float().mean()), "per view", [round(float(x), 2) for x in inside.float().mean(1)])
    return r, ref_lvl, inside
ops.project = proj
import mvgformer_amd.decoder as D
ops.PROFILE = {} if hasattr(ops, "PROFILE") else None
with torch.no_grad():
    out = dec(g.tgt, g.reference_points, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None, query_pos=g.query_pos)
