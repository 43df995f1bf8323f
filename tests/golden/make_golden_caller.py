"""Caller-glue fixtures produced by RUNNING THE REFERENCE's own caller code.

Build container only (needs /root/reference):

    python -m tests.golden.make_golden_caller

What runs, unmodified, from the reference tree:

  * ``DyanmicQueryTransformer.forward`` (lib/models/dq_transformer.py:335-755) as an unbound method on an
    instance created without its CUDA-only constructor (``object.__new__`` + the attributes the method reads):
    backbone hand-off and level tables (:352-388), the person/joint query embedding sum and split (:394-432),
    ``initialize_reference_points('sample_space')`` + ``generate_T_pose`` + ``norm2absolute``
    (:250-330, :225-236, multi_view_pose_transformer.py:292-297) with the reference's own ``tpose.pt``,
    the reference ``DQDecoder`` (CPU twin of the sampling op, as in make_golden.py), ``inverse_sigmoid``
    and the out dict incl. the Shelf/Campus joint permutation (:569-603);
  * ``validate_3d`` (lib/core/function.py:329-585) around it with a two-batch list as the loader: the
    ``[x, y, z, (score > thr) - 1, score]`` packing (:386-396).

The "backbone" is a function that returns the synthetic pyramid (the backbone is out of scope).
Fixture = the decoder's inputs as the reference's glue built them, the decoder's raw outputs, the out dict
and the packed predictions, for the Panoptic joint format and for the Shelf/Campus permutation.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from mvgformer_amd.synthetic import build_case  # noqa: E402
from tests.golden.cases import CALLER_CASE, caller_embeddings  # noqa: E402
from tests.golden.ref_harness import REF_ROOT, build_reference_decoder, load_reference  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy()


def reference_model(dqt, case, dec, tpose, convert):
    m = object.__new__(dqt.DyanmicQueryTransformer)
    nn.Module.__init__(m)
    je, ie = caller_embeddings(case.NQ)
    m.joint_embedding = nn.Embedding(15, 512)
    m.instance_embedding = nn.Embedding(case.NQ, 512)
    with torch.no_grad():
        m.joint_embedding.weight.copy_(je)
        m.instance_embedding.weight.copy_(ie)
    m.decoder = dec
    m.backbone = None                                    # set per batch (returns that batch's pyramid)
    m.use_feat_level = [0, 1, 2]
    m.num_instance, m.num_joints = case.NQ, 15
    m.query_embed_type = "person_joint"
    m.init_ref_method, m.init_ref_method_value = "sample_space", None
    m.close_pose_embedding = False
    m.t_pose_origin = tpose
    m.grid_size = torch.tensor(list(case.space_size))
    m.grid_center = torch.tensor(list(case.space_center))
    m.gt_match, m.gt_match_test = False, False
    m.convert_joint_format_indices = convert
    m.aux_loss = False
    m.visualization_jump_num = -1
    m.log_val_loss = False
    m.eval()
    return m


def run(ref, dqt, fn, convert, tag, out):
    spec = CALLER_CASE
    cases = [build_case(spec["config"], B=spec["B"], seed=spec["seed"] + i, layers=spec["layers"],
                        valid_fraction=spec["valid_fraction"])
             for i in range(spec["batches"])]
    case = cases[0]
    dec = build_reference_decoder(case)
    tpose = torch.load(os.path.join(REF_ROOT, "tpose.pt"), map_location="cpu")
    model = reference_model(dqt, case, dec, tpose, convert)

    seen = []
    orig_fwd = dec.forward

    def spy(tgt, reference_points, src_views, **kw):
        res = orig_fwd(tgt, reference_points, src_views, **kw)
        seen.append(dict(tgt=tgt, ref=reference_points, pos=kw["query_pos"], shapes=kw["src_spatial_shapes"],
                         starts=kw["src_level_start_index"], res=res))
        return res
    dec.forward = spy

    nj = 15 if convert is None else len(convert)
    loader = []
    for c in cases:
        views = [torch.zeros(c.B, 3, c.img_size[1], c.img_size[0]) for _ in range(c.V)]
        meta = [dict(m) for m in c.meta]
        meta[0].update(image=["frame"] * c.B, joints_3d=torch.zeros(c.B, 10, nj, 3),
                       num_person=torch.zeros(c.B, dtype=torch.long),
                       joints_3d_voxelpose_pred=torch.zeros(c.B, 10, nj, 5))
        loader.append((views, meta))
    it = iter(cases)

    def backbone(x, levels):
        c = next(it)
        assert x.shape[0] == c.V * c.B
        return list(reversed(c.src_views))                # the reference reverses the backbone's list (:353)
    model.backbone = backbone

    config = SimpleNamespace(DEBUG=SimpleNamespace(LOG_VAL_LOSS=False), PRINT_FREQ=100)
    thr = spec["threshold"]
    preds, _ = fn.validate_3d(config, model, loader, "/tmp", thr, num_views=case.V, device="cpu")
    assert len(preds) == spec["B"] * spec["batches"] and len(seen) == spec["batches"]

    for i, s in enumerate(seen):
        hs, refs, r2d, p2d, cls = s["res"]
        pre = "%s/b%d/" % (tag, i)
        if convert is None:                               # decoder inputs do not depend on the joint format
            out[pre + "tgt"] = npy(s["tgt"][:1])          # expanded over the batch: every item identical
            out[pre + "query_pos"] = npy(s["pos"][:1])
            assert all(torch.equal(s["tgt"][b], s["tgt"][0]) and torch.equal(s["pos"][b], s["pos"][0])
                       for b in range(case.B))
            out[pre + "reference_points"] = npy(s["ref"])
            out[pre + "spatial_shapes"] = npy(s["shapes"])
            out[pre + "level_start_index"] = npy(s["starts"])
            out[pre + "dec_refs"] = npy(refs)
            out[pre + "dec_refs2d"] = npy(r2d)
            out[pre + "dec_projs2d"] = npy(p2d)
            out[pre + "dec_cls"] = npy(torch.stack(cls))
            out[pre + "hs_shape"] = np.array(hs.shape)
        out[pre + "pred"] = np.stack(preds[i * case.B:(i + 1) * case.B])
    # the out dict of the LAST batch, straight from the model (validate_3d only keeps the packed array)
    last = cases[-1]
    it = iter([last])
    with torch.no_grad():
        o = model(views=loader[-1][0], meta=loader[-1][1], threshold=thr)
    out[tag + "/out/pred_logits"] = npy(o["pred_logits"])
    out[tag + "/out/pred_poses"] = npy(o["pred_poses"]["outputs_coord"])
    out[tag + "/out/pred_poses_2d"] = npy(o["pred_poses_2d"]["outputs_coord_2d"])
    out[tag + "/out/pred_poses_2d_proj"] = npy(o["pred_poses_2d_proj"]["outputs_coord_2d_proj"])
    dec.forward = orig_fwd
    return tpose


def main():
    ref = load_reference()
    import models.dq_transformer as dqt
    import lib.core.function as fn
    out = {}
    tpose = run(ref, dqt, fn, None, "panoptic", out)
    shelf = [14, 13, 12, 6, 7, 8, 11, 10, 9, 3, 4, 5, 0, 1]   # configs/shelf_campus/shelf_knn5-lr4-q1024.yaml:143
    run(ref, dqt, fn, shelf, "shelf", out)
    out["tpose"] = npy(tpose)
    out["convert_joint_format_indices"] = np.array(shelf)
    out["threshold"] = np.float32(CALLER_CASE["threshold"])
    # inverse_sigmoid (lib/models/util/misc.py:608-612) on edge values
    from models.util.misc import inverse_sigmoid
    x = torch.tensor([0.0, 1e-7, 1e-5, 0.1, 0.5, 0.9, 1.0 - 1e-6, 1.0, 1.5, -0.2])
    out["inverse_sigmoid/x"] = npy(x)
    out["inverse_sigmoid/y"] = npy(inverse_sigmoid(x))
    np.savez_compressed(os.path.join(HERE, "caller.npz"), **out)
    print("caller.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
