"""Evaluation post-processing of the decoder's predictions.

What ``run/validate_3d.py:204-284`` does with the packed predictions ``[x, y, z, (score > thr) - 1, score]``
(``mvgformer_amd.caller.pack_predictions``): classification filter + nearby-joints NMS
(``lib/core/nms.py:210-283``), then AP@25..150 mm / recall / MPJPE on Panoptic
(``lib/dataset/panoptic.py:493-574,711-764``) or PCP on Shelf / Campus (``lib/dataset/shelf.py:255-330``,
``lib/dataset/campus.py:250-320``).

The reference evaluates in nested Python loops over numpy arrays on the host (and its NMS builds an
(N, N, J, 3) fp64 temporary: 377 MB for 1024 candidates).  Here the two dense parts -- the pose-to-pose joint
distances of the NMS and the prediction-to-ground-truth MPJPE matrix -- are batched tensor ops that run on
whatever device the predictions live on (the MI355X when they come straight from the decoder), and only the
inherently sequential greedy steps (suppression order, first-match-wins AP bookkeeping) run on the host over
the small boolean / index results.  Numerics: distances are computed in fp64 like the reference's numpy code:
keep-lists and matches are identical, metrics agree to 1e-12 (summation order of the joint means differs;
tests/test_evaluate.py against golden vectors produced by the reference's own functions).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

__all__ = ["nearby_joints_nms", "filter_and_nms", "match_predictions", "eval_list_to_ap", "eval_list_to_mpjpe",
           "eval_list_to_recall", "evaluate_panoptic", "evaluate_pcp", "MPJPE_THRESHOLDS", "PCP_LIMBS"]

MPJPE_THRESHOLDS = tuple(range(25, 155, 25))            # panoptic.py:559
PCP_LIMBS = ((0, 1), (1, 2), (3, 4), (4, 5), (6, 7), (7, 8), (9, 10), (10, 11), (12, 13))   # shelf.py:270-271
PCP_BONE_GROUPS = OrderedDict([("Head", [8]), ("Torso", [9]), ("Upper arms", [5, 6]), ("Lower arms", [4, 7]),
                               ("Upper legs", [1, 2]), ("Lower legs", [0, 3])])              # shelf.py:316-319


def _as_tensor(x, device=None):
    if isinstance(x, torch.Tensor):
        return x.detach().to(torch.float64)
    return torch.as_tensor(np.asarray(x), dtype=torch.float64, device=device)


# --------------------------------------------------------------------------------------- NMS
def nearby_joints_nms(kpts_db, dist_thr, num_nearby_joints_thr=None, max_dets=-1):
    """lib/core/nms.py:210-283 for the combined input format used by validate_3d.py:231:
    ``kpts_db`` (N, J, >=5) rows ``[x, y, z, flag, score]`` (numpy or torch, any device).  Two poses are "close"
    when more than ``num_nearby_joints_thr`` of their joints are nearer than ``dist_thr`` x the diagonal of the
    first pose's bounding box; in descending score order every not-yet-ignored pose hands its slot to the best
    scored member of its neighbourhood and the whole neighbourhood is ignored from then on.
    Returns the list of kept indices in keep order."""
    if not dist_thr > 0:
        raise AssertionError("`dist_thr` must be greater than 0.")                   # nms.py:233
    if len(kpts_db) == 0:
        return []
    db = _as_tensor(kpts_db)
    scores = db[:, 0, 4]
    kpts = db[:, :, :3]
    n, J, _ = kpts.shape
    if num_nearby_joints_thr is None:
        num_nearby_joints_thr = J // 2
    if not num_nearby_joints_thr < J:
        raise AssertionError("`num_nearby_joints_thr` must be less than the number of joints.")
    # distance threshold per (row) pose: dist_thr x bounding-box diagonal (nms.py:255-260)
    area = torch.sqrt(((kpts.max(1).values - kpts.min(1).values) ** 2).sum(1))       # (N,)
    thr = area * dist_thr
    # joint-wise distances between all pose pairs, one joint at a time: (N, N) per joint instead of the
    # reference's (N, N, J, 3) temporary (nms.py:263-266)
    close_cnt = torch.zeros((n, n), dtype=torch.int32, device=kpts.device)
    for j in range(J):
        d = kpts[:, None, j, :] - kpts[None, :, j, :]
        close_cnt += (torch.sqrt((d ** 2).sum(-1)) < thr[:, None]).to(torch.int32)
    close = (close_cnt > num_nearby_joints_thr).cpu().numpy()
    sc = scores.cpu().numpy()
    # greedy suppression (nms.py:268-277); np.argsort(...)[::-1] as in the reference (tie order included)
    ignored, keep = set(), []
    for i in np.argsort(sc)[::-1]:
        if i in ignored:
            continue
        nb = close[i].nonzero()[0]
        best = nb[np.argmax(sc[nb])]
        if best not in ignored:
            keep.append(int(best))
            ignored.update(int(k) for k in nb)
    if max_dets > 0 and len(keep) > max_dets:                                         # nms.py:280-282
        sub = np.argsort(sc[keep])[-1:-max_dets - 1:-1]
        keep = [keep[i] for i in sub]
    return keep


def filter_and_nms(pred, dist_thr=0.3, num_nearby_joints_thr=7):
    """validate_3d.py:228-234: drop candidates below the classification threshold (flag column < 0), then NMS
    with the reference's default thresholds.  pred (N, J, 5) numpy or torch; returns the surviving rows."""
    keep_cls = pred[:, 0, 3] >= 0
    pred = pred[keep_cls]
    idx = nearby_joints_nms(pred, dist_thr, num_nearby_joints_thr)
    return pred[idx]


# --------------------------------------------------------------------------------------- AP / MPJPE / recall
def match_predictions(preds, gts, gts_vis, method="score_sort"):
    """panoptic.py:497-556: for every frame, every (classification-filtered) prediction is aligned with the
    ground-truth person of smallest MPJPE over the visible joints.
    preds: list over frames of (N_f, J, 5); gts / gts_vis: lists over frames of (G_f, J, 3) / (G_f, J, >=1)
    (frames with G_f == 0 are skipped like panoptic.py:508).  Returns (eval_list, total_gt) with
    eval_list = dict(mpjpe, score, gt_id) of equal-length float64 / int64 arrays in the reference's append
    order."""
    mp, sc, gid = [], [], []
    total_gt = 0
    for pred, gt, vis in zip(preds, gts, gts_vis):
        if len(gt) == 0:
            continue
        p = _as_tensor(pred)
        g = _as_tensor(gt, p.device)[..., :3]
        v = (_as_tensor(vis, p.device)[..., 0] > 0)                                   # (G, J)
        if method != "mpjpe_sort":
            p = p[p[:, 0, 3] >= 0]                                                    # panoptic.py:544
        if len(p):
            d = torch.sqrt(((p[:, None, :, :3] - g[None]) ** 2).sum(-1))              # (N, G, J)
            w = v.to(d.dtype)[None]
            # mean over the visible joints only (np.mean of the masked selection, panoptic.py:548-550); a person
            # without visible joints gives nan there -- same here (0/0)
            m = (d * w).sum(-1) / w.sum(-1)
            best = torch.min(_nan_last(m), dim=1)
            m_np, b_np = m.cpu().numpy(), best.indices.cpu().numpy()
            s_np = p[:, 0, 4].cpu().numpy()
            seen = set()
            for i in range(len(p)):
                gt_id = int(total_gt + b_np[i])
                if method == "mpjpe_sort":                                            # one prediction per gt, first wins
                    if gt_id in seen:
                        continue
                    seen.add(gt_id)
                mp.append(float(m_np[i, b_np[i]]))
                sc.append(float(s_np[i]))
                gid.append(gt_id)
        total_gt += len(gt)
    return dict(mpjpe=np.asarray(mp, np.float64), score=np.asarray(sc, np.float64), gt_id=np.asarray(gid, np.int64)), total_gt


def _nan_last(m):
    """np.argmin returns the first nan if there is one; persons without visible joints do not occur in the
    datasets, keep torch.min's behaviour well defined anyway."""
    return torch.where(torch.isnan(m), torch.full_like(m, float("inf")), m)


def _order(ev, method):
    # list.sort is stable, also with reverse=True (panoptic.py:713-717)
    if method == "score_sort":
        return np.argsort(-ev["score"], kind="stable")
    if method == "mpjpe_sort":
        return np.argsort(ev["mpjpe"], kind="stable")
    return np.arange(len(ev["score"]))


def _first_hits(ev, order, threshold):
    """true positives in `order`: mpjpe below the threshold and the gt not matched before (panoptic.py:723-729)."""
    mp, gid = ev["mpjpe"][order], ev["gt_id"][order]
    ok = mp < threshold
    first = np.zeros(len(order), dtype=bool)
    if ok.any():
        idx = np.flatnonzero(ok)
        _, pos = np.unique(gid[idx], return_index=True)          # first qualifying occurrence of every gt
        first[idx[pos]] = True
    return first


def eval_list_to_ap(ev, total_gt, threshold, method="score_sort"):
    """panoptic.py:711-741 -> (ap, recall)."""
    order = _order(ev, method)
    tp = _first_hits(ev, order, threshold)
    n = len(order)
    tpc = np.cumsum(tp.astype(np.float64))
    fpc = np.cumsum((~tp).astype(np.float64))
    recall = tpc / (total_gt + 1e-5)
    precise = tpc / (tpc + fpc + 1e-5)
    if n:
        precise = np.maximum.accumulate(precise[::-1])[::-1]       # panoptic.py:733-734
    precise = np.concatenate(([0.0], precise, [0.0]))
    recall = np.concatenate(([0.0], recall, [1.0]))
    index = np.where(recall[1:] != recall[:-1])[0]
    ap = np.sum((recall[index + 1] - recall[index]) * precise[index + 1])
    return float(ap), float(recall[-2])


def eval_list_to_mpjpe(ev, threshold=500, method="score_sort"):
    """panoptic.py:743-758: mean MPJPE of the true positives at `threshold` (inf when there are none)."""
    order = _order(ev, method)
    tp = _first_hits(ev, order, threshold)
    return float(np.mean(ev["mpjpe"][order][tp])) if tp.any() else float("inf")


def eval_list_to_recall(ev, total_gt, threshold=500):
    """panoptic.py:760-764."""
    return len(np.unique(ev["gt_id"][ev["mpjpe"] < threshold])) / total_gt


def evaluate_panoptic(preds, gts, gts_vis, method="score_sort"):
    """Panoptic.evaluate (panoptic.py:493-709) -> (aps, recalls, mpjpe, recall500) at 25..150 mm."""
    ev, total_gt = match_predictions(preds, gts, gts_vis, method)
    aps, recs = [], []
    for t in MPJPE_THRESHOLDS:
        ap, rec = eval_list_to_ap(ev, total_gt, t, method)
        aps.append(ap)
        recs.append(rec)
    return aps, recs, eval_list_to_mpjpe(ev, method=method), eval_list_to_recall(ev, total_gt)


# --------------------------------------------------------------------------------------- PCP (Shelf / Campus)
def evaluate_pcp(preds, actor_gts, recall_threshold=500, alpha=0.5):
    """Shelf.evaluate / Campus.evaluate (shelf.py:255-330) on in-memory ground truth.
    preds: list over evaluated frames of (N_f, J, 5); actor_gts: list over actors of lists over the same frames of
    (14, 3) arrays in mm, or None / empty where the actor is not annotated (the reference reads them from
    actorsGT.mat and skips empty entries, shelf.py:288-290).
    Returns (actor_pcp, avg_pcp, bone_person_pcp, recall)."""
    num_person = len(actor_gts)
    correct = np.zeros(num_person)
    total = np.zeros(num_person)
    bone_correct = np.zeros((num_person, 10))
    total_gt = match_gt = 0
    li = torch.as_tensor([k[0] for k in PCP_LIMBS])
    lj = torch.as_tensor([k[1] for k in PCP_LIMBS])
    for f, pred in enumerate(preds):
        p = _as_tensor(pred)
        p = p[p[:, 0, 3] >= 0][:, :, :3]                                              # shelf.py:281
        for person in range(num_person):
            gt = actor_gts[person][f]
            if gt is None or len(gt) == 0:
                continue
            if len(p) == 0:
                raise ValueError("frame %d has ground truth but no prediction passed the classification filter "
                                 "(the reference fails in np.stack here, shelf.py:284)" % f)
            g = _as_tensor(gt, p.device)
            mp = torch.sqrt(((g[None] - p) ** 2).sum(-1)).mean(-1)                    # shelf.py:292-294
            n = int(torch.argmin(mp))
            if float(mp[n]) < recall_threshold:
                match_gt += 1
            total_gt += 1
            best = p[n]
            e_s = torch.linalg.norm(best[li] - g[li], dim=-1)
            e_e = torch.linalg.norm(best[lj] - g[lj], dim=-1)
            length = torch.linalg.norm(g[li] - g[lj], dim=-1)
            ok = ((e_s + e_e) / 2.0 <= alpha * length).cpu().numpy()
            total[person] += len(PCP_LIMBS)
            correct[person] += ok.sum()
            bone_correct[person, :len(PCP_LIMBS)] += ok
            p_hip, g_hip = (best[2] + best[3]) / 2.0, (g[2] + g[3]) / 2.0             # shelf.py:308-315
            total[person] += 1
            if float((torch.linalg.norm(p_hip - g_hip) + torch.linalg.norm(best[12] - g[12])) / 2.0) <= \
                    alpha * float(torch.linalg.norm(g_hip - g[12])):
                correct[person] += 1
                bone_correct[person, 9] += 1
    actor_pcp = correct / (total + 1e-8)
    avg_pcp = float(np.mean(actor_pcp[:3]))
    bone_person_pcp = OrderedDict((k, np.sum(bone_correct[:, v], axis=-1) / (total / 10 * len(v) + 1e-8))
                                  for k, v in PCP_BONE_GROUPS.items())
    return actor_pcp, avg_pcp, bone_person_pcp, match_gt / (total_gt + 1e-8)
