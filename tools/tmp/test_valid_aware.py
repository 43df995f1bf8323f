import os, sys, subprocess, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mvgformer_amd.factory import build_decoder_for_case, case_to_device
from mvgformer_amd.synthetic import build_case
mode = os.environ.get("MVG_VALID_AWARE", "1")
for name, kw in (("cfg2 valid10", dict(valid_fraction=0.1)), ("cfg2 all valid", dict()), ("cfg2 valid50", dict(valid_fraction=0.5)), ("mini5 valid30", dict(valid_fraction=0.3))):
    cfgname = "mini5" if name.startswith("mini5") else "cfg2"
    case = build_case(cfgname, seed=0, **kw)
    dec = build_decoder_for_case(case, "cuda", torch.bfloat16)
    g = case_to_device(case, "cuda")
    with torch.no_grad():
        out = dec(g.tgt, g.reference_points, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None, query_pos=g.query_pos, threshold=0.1)
    h = hashlib.sha256()
    for t in out[:4]:
        h.update(t.float().cpu().numpy().tobytes())
    for c in out[4]:
        h.update(c.float().cpu().numpy().tobytes())
    share = [float((c[..., 1] > 0.1).float().mean()) for c in out[4]]
    print("VALID_AWARE=%s %-16s sha %s  valid share per layer %s" % (mode, name, h.hexdigest()[:16], ["%.3f" % v for v in share]))
