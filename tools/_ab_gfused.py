import os, sys, torch
sys.path.insert(0, "/root/repo")
from mvgformer_amd import _lib
import subprocess, json
lib = _lib.load()
from mvgformer_amd.factory import build_decoder_for_case, case_to_device
from mvgformer_amd.synthetic import build_case
for cfg in ("cfg2", "cfg4"):
    case = build_case(cfg, seed=0)
    dec = build_decoder_for_case(case, "cuda", torch.float32).eval()
    g = case_to_device(case, "cuda")
    outs = {}
    for mp in (0, 1):
        assert lib.mvg_set_tuning(b"gfused_map", mp) == 0
        with torch.no_grad():
            o = dec(g.tgt, g.reference_points, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None, query_pos=g.query_pos, threshold=0.1)
        outs[mp] = [t.clone() for t in o[:2]] + [c.clone() for c in o[4]]
    print(cfg, "bit-identical:", all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])))
