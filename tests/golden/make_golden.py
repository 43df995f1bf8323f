"""Generate the golden vectors under tests/golden/*.npz by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python -m tests.golden.make_golden

The reference (lib/models/dq_decoder.py, lib/models/ops/modules/projattn.py,
lib/models/ops/functions/deform_func.py, lib/utils/cameras.py,
lib/mvn/utils/multiview.py) is imported read-only via ``ref_harness`` and run on
the seeded synthetic cases of ``mvgformer_amd.synthetic``.  Inputs are re-created
from the seed by the tests; the fixtures hold the reference's OUTPUTS (and a few
small inputs), all float32.  The reference has no tests of its own for this path
 -- these vectors are the pin for ``oracle/``.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from mvgformer_amd.synthetic import build_case  # noqa: E402
from tests.golden.cases import LAYER_CASES, msda_case, threshold_margin  # noqa: E402
from tests.golden.ref_harness import build_reference_decoder, load_reference  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy()


def gen_msda(ref):
    """(i) deform_core_pytorch in/out (deform_func.py:68-99) incl. out-of-range points."""
    out = {}
    for name in ("small_f32", "ragged_f32", "edge_f32"):
        c = msda_case(name)
        y = ref.deform_core_pytorch(c["value"], c["shapes"], c["loc"], c["weight"])
        out[name + "/out"] = npy(y)
        y64 = ref.deform_core_pytorch(c["value"].double(), c["shapes"], c["loc"].double(), c["weight"].double())
        out[name + "/out_f64"] = npy(y64)
    np.savez_compressed(os.path.join(HERE, "msda.npz"), **out)
    print("msda.npz", {k: v.shape for k, v in out.items()})


def run_layers(ref, cname, spec):
    case = build_case(spec["config"], B=spec.get("B", 1), seed=spec["seed"], NQ=spec.get("NQ"),
                      layers=spec.get("layers"), valid_fraction=spec.get("valid_fraction"))
    dec = build_reference_decoder(case)
    thr = spec.get("threshold", 0.1)
    cap = {}

    # capture ProjAttn intermediates of layer 0 / view 0 through hooks (no reference edits)
    pa0 = dec.layers[0].proj_attn
    calls = {"n": 0}

    def hook_off(mod, inp, outp):
        if calls["n"] == 0:
            cap["pa_x"] = inp[0].detach().clone()          # ref-point feats + query
            cap["pa_off"] = outp.detach().clone()
    h1 = pa0.sampling_offsets.register_forward_hook(hook_off)

    orig_apply = ref.projattn.DeformFunction.apply

    def spy(value, shapes, starts, loc, w, step):
        y = orig_apply(value, shapes, starts, loc, w, step)
        if calls["n"] == 0:
            cap.update(pa_value=value.detach().clone(), pa_loc=loc.detach().clone(),
                       pa_w=w.detach().clone(), pa_samp=y.detach().clone())
        calls["n"] += 1
        return y
    ref.projattn.DeformFunction.apply = staticmethod(spy)

    def hook_out(mod, inp, outp):
        if "pa_out" not in cap:
            cap["pa_query"] = inp[0].detach().clone()
            cap["pa_ref"] = inp[1].detach().clone()
            cap["pa_out"] = outp.detach().clone()
    h2 = pa0.register_forward_hook(hook_out)

    # projection of the layer-0 reference points, per view (dq_decoder.py:331-397)
    layer0 = dec.layers[0]
    proj_r, proj_in = [], []
    with torch.no_grad():
        for v in range(case.V):
            r, inside = layer0.project_ref_points(case.reference_points[:, :, None], case.meta[v], 1, case.B,
                                                  case.NQ * 15, torch.device("cpu"))
            proj_r.append(r.reshape(case.B, -1, 2))
            proj_in.append(inside.reshape(case.B, -1))
        hs, refs, r2d, p2d, cls = dec(case.tgt, case.reference_points, case.src_views, case.meta,
                                      case.spatial_shapes, case.level_start_index, None,
                                      query_pos=case.query_pos,
                                      src_padding_mask=[torch.zeros(1, 1, dtype=torch.bool)], threshold=thr)
    h1.remove(); h2.remove()
    ref.projattn.DeformFunction.apply = orig_apply

    margin = threshold_margin(cls, thr)
    assert margin > 1e-3, "class prob within %.2e of the threshold in %s" % (margin, cname)
    valid_counts = [int((c[..., 1] > thr).sum()) for c in cls]
    inside_frac = float(torch.stack(proj_in).float().mean())
    print(cname, "valid/layer", valid_counts, "of", case.B * case.NQ, "margin %.3e" % margin,
          "inside-image frac %.3f" % inside_frac)

    out = dict(hs=npy(hs), refs=npy(refs), refs2d=npy(r2d), projs2d=npy(p2d),
               cls=npy(torch.stack(cls)), proj_r=npy(torch.stack(proj_r)),
               proj_inside=npy(torch.stack(proj_in)), threshold=np.float32(thr))
    if spec.get("projattn", False):
        S = cap["pa_value"].shape[1]
        rows = np.arange(0, S, 40)
        out.update(pa_out=npy(cap["pa_out"]), pa_ref=npy(cap["pa_ref"]),
                   pa_off=npy(cap["pa_off"]), pa_w=npy(cap["pa_w"]), pa_loc_q40=npy(cap["pa_loc"])[:, :40],
                   pa_samp=npy(cap["pa_samp"]), pa_x_q20=npy(cap["pa_x"])[:, :20],
                   pa_value_rows=rows, pa_value_sub=npy(cap["pa_value"])[:, rows],
                   pa_value_sum=np.float64(cap["pa_value"].double().sum().item()))
    return case, out


def gen_triangulation(ref, case):
    """(iv) undistort + get_proj_matricies_batch + DLT on a seeded set of 2D points
    (dq_decoder.py:119-246, multiview.py:170-269)."""
    rs = np.random.RandomState(4242)
    n, V, J = 7, case.V, 15
    w, h = case.cfg["orig_wh"]
    kp = torch.from_numpy((rs.rand(n, V, J, 2) * [w, h]).astype(np.float32))
    conf = torch.softmax(torch.from_numpy(rs.standard_normal((n, V, J)).astype(np.float32)), 1)
    meta_b = []
    for m in case.meta:
        cam = {k: v[:1].expand(n, *v.shape[1:]).contiguous() for k, v in m["camera"].items()}
        meta_b.append(dict(camera=cam))
    with torch.no_grad():
        ud = ref.dq_decoder.undistort(kp, meta_b, iter_num=5)
        Pm = ref.dq_decoder.get_proj_matricies_batch(meta_b, V, torch.device("cpu"), inv_trans=True)
        X = ref.multiview.triangulate_batch_of_points_batch_version(Pm, ud, confidences_batch=conf, solver="linalg")
    return dict(kp=npy(kp), conf=npy(conf), undist=npy(ud), proj_mats=npy(Pm), points3d=npy(X))


def main():
    ref = load_reference()
    torch.manual_seed(0)
    only = sys.argv[1:]                  # optional: names of the layer cases to (re)generate; default everything
    if not only:
        gen_msda(ref)
    for cname, spec in LAYER_CASES.items():
        if only and cname not in only:
            continue
        case, out = run_layers(ref, cname, spec)
        if spec.get("triangulation", False):
            out.update({"tri_" + k: v for k, v in gen_triangulation(ref, case).items()})
        path = os.path.join(HERE, cname + ".npz")
        np.savez_compressed(path, **out)
        print("  ->", path, "%.0f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
