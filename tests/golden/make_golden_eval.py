"""Golden vectors for the evaluation post-processing, produced by the REFERENCE's own
functions (build container only; /root/reference is imported through tests/golden/ref_harness.py, nothing is copied):

  lib/core/nms.py::nearby_joints_nms, lib/dataset/panoptic.py::Panoptic.evaluate (+ the three _eval_list_* helpers),
  lib/dataset/shelf.py::Shelf.evaluate (identical to Campus.evaluate).

    python tests/golden/make_golden_eval.py        -> tests/golden/eval.npz

The inputs are seeded synthetic scenes (tests/golden/eval_cases.py, shared with the tests); only the reference's
OUTPUTS are stored.
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.golden import ref_harness  # noqa: E402
from tests.golden.eval_cases import NMS_CASES, panoptic_scene, pcp_scene  # noqa: E402


def main():
    ref_harness.load_reference()
    nms = importlib.import_module("core.nms")
    pan = importlib.import_module("dataset.panoptic")
    shelf = importlib.import_module("dataset.shelf")
    out = {}

    # ---- NMS
    for name, spec in NMS_CASES.items():
        preds, _, _ = panoptic_scene(spec["seed"], frames=spec["frames"])
        for f, p in enumerate(preds):
            keep = nms.nearby_joints_nms(p.copy(), spec["dist_thr"], spec["num_nearby"], max_dets=spec.get("max_dets", -1))
            out["nms_%s_f%d" % (name, f)] = np.asarray(keep, dtype=np.int64)

    # ---- Panoptic.evaluate on a stub dataset object (the method only touches these attributes)
    for seed in (11, 12):
        preds, gts, vis = panoptic_scene(seed, frames=6)
        V = 5
        stub = types.SimpleNamespace()
        stub.num_views = V
        stub.db = []
        for g, v in zip(gts, vis):
            rec = {"joints_3d": list(g), "joints_3d_vis": list(v)}
            stub.db.extend([rec] * V)
        stub.db_size = len(stub.db)
        stub.show_camera_detail = False
        stub._eval_list_to_ap = pan.Panoptic._eval_list_to_ap
        stub._eval_list_to_mpjpe = pan.Panoptic._eval_list_to_mpjpe
        stub._eval_list_to_recall = pan.Panoptic._eval_list_to_recall
        for method in ("score_sort", "mpjpe_sort"):
            aps, recs, mpjpe, rec500 = pan.Panoptic.evaluate(stub, [p.copy() for p in preds], method=method)
            out["pan_%d_%s" % (seed, method)] = np.asarray(list(aps) + list(recs) + [mpjpe, rec500], dtype=np.float64)
        # the validate_3d.py:228-236 pipeline: classification filter + NMS(0.3, 7), then evaluate
        pn = []
        for p in preds:
            p = p[p[:, 0, 3] >= 0]
            pn.append(p[nms.nearby_joints_nms(p, 0.3, 7)].copy())
        aps, recs, mpjpe, rec500 = pan.Panoptic.evaluate(stub, pn)
        out["pan_%d_nms" % seed] = np.asarray(list(aps) + list(recs) + [mpjpe, rec500], dtype=np.float64)

    # ---- Shelf.evaluate with the ground truth served from memory instead of actorsGT.mat
    for seed in (21, 22):
        preds, actors, frame_range, n_frames_total = pcp_scene(seed)
        P = len(actors)
        actor3d = np.empty((1, P), dtype=object)
        for a in range(P):
            col = np.empty((n_frames_total, 1), dtype=object)
            for fi in range(n_frames_total):
                col[fi, 0] = np.zeros((1, 0))
            for i, fi in enumerate(frame_range):
                g = actors[a][i]
                col[fi, 0] = np.zeros((1, 0)) if g is None else np.asarray(g) / 1000.0     # the .mat is in metres
            actor3d[0, a] = col
        shelf.scio = types.SimpleNamespace(loadmat=lambda _f, _d=actor3d: {"actor3D": _d})
        stub = types.SimpleNamespace(dataset_root="/nonexistent", frame_range=list(frame_range))
        actor_pcp, avg_pcp, bone, recall = shelf.Shelf.evaluate(stub, [p.copy() for p in preds])
        out["pcp_%d" % seed] = np.concatenate([np.asarray(actor_pcp, np.float64), [avg_pcp, recall],
                                               np.concatenate([np.asarray(v, np.float64) for v in bone.values()])])
    path = os.path.join(HERE, "eval.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "(%d arrays)" % len(out))


if __name__ == "__main__":
    main()
