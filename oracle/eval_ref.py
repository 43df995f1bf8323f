"""TEST INFRASTRUCTURE -- CPU restatement (plain numpy loops) of the reference's evaluation post-processing
(SURVEY.md section 8 f4).  Imported only by tests/; the product code is mvgformer_amd/evaluate.py.

Pinned by tests/golden/eval.npz, produced by the reference's own functions (tests/golden/make_golden_eval.py):
lib/core/nms.py:210-283, lib/dataset/panoptic.py:493-574,711-764, lib/dataset/shelf.py:255-330.
"""
import numpy as np

MPJPE_THRESHOLDS = list(range(25, 155, 25))          # panoptic.py:559


def nearby_joints_nms(db, dist_thr, num_nearby_joints_thr=None, max_dets=-1):
    """nms.py:210-283 (combined input): db (N, J, 5) rows [x, y, z, flag, score]."""
    if len(db) == 0:
        return []
    db = np.asarray(db, dtype=np.float64)
    N, J = db.shape[:2]
    scores = db[:, 0, 4]
    if num_nearby_joints_thr is None:
        num_nearby_joints_thr = J // 2
    close = np.zeros((N, N), dtype=bool)
    for a in range(N):
        span = db[a, :, :3].max(0) - db[a, :, :3].min(0)                 # nms.py:255-256
        limit = np.sqrt((span ** 2).sum()) * dist_thr                    # nms.py:257-260
        for b in range(N):
            near = 0
            for j in range(J):
                diff = db[a, j, :3] - db[b, j, :3]                       # nms.py:263-264
                if np.sqrt((diff ** 2).sum()) < limit:
                    near += 1
            close[a, b] = near > num_nearby_joints_thr                    # nms.py:265-266
    ignored, keep = set(), []
    for a in np.argsort(scores)[::-1]:                                    # nms.py:270
        if a in ignored:
            continue
        members = [b for b in range(N) if close[a, b]]
        best = members[int(np.argmax(scores[members]))]
        if best not in ignored:                                           # nms.py:275-277
            keep.append(int(best))
            ignored |= set(members)
    if max_dets > 0 and len(keep) > max_dets:                             # nms.py:280-282
        top = np.argsort(scores[keep])[-1:-max_dets - 1:-1]
        keep = [keep[i] for i in top]
    return keep


def match_predictions(preds, gts, gts_vis, method="score_sort"):
    """panoptic.py:497-556 -> (list of dict(mpjpe, score, gt_id), total_gt)."""
    out, total_gt = [], 0
    for pred, gt, vis in zip(preds, gts, gts_vis):
        if len(gt) == 0:                                                  # panoptic.py:508
            continue
        pred = np.asarray(pred, dtype=np.float64)
        used = []
        if method != "mpjpe_sort":
            pred = pred[pred[:, 0, 3] >= 0]                               # panoptic.py:544
        for pose in pred:
            errs = []
            for g, gv in zip(gt, vis):
                sel = np.asarray(gv)[:, 0] > 0
                errs.append(np.mean(np.sqrt(np.sum((pose[sel, 0:3] - np.asarray(g)[sel]) ** 2, axis=-1))))
            k = int(np.argmin(errs))
            gt_id = total_gt + k
            if method == "mpjpe_sort":                                    # panoptic.py:533-541: first prediction per gt
                if gt_id in used:
                    continue
                used.append(gt_id)
            out.append({"mpjpe": float(errs[k]), "score": float(pose[0, 4]), "gt_id": int(gt_id)})
        total_gt += len(gt)
    return out, total_gt


def _sorted(eval_list, method):
    if method == "score_sort":                                            # panoptic.py:713-717 (stable sorts)
        return sorted(eval_list, key=lambda k: k["score"], reverse=True)
    if method == "mpjpe_sort":
        return sorted(eval_list, key=lambda k: k["mpjpe"])
    return list(eval_list)


def eval_list_to_ap(eval_list, total_gt, threshold, method="score_sort"):
    """panoptic.py:711-741."""
    items = _sorted(eval_list, method)
    n = len(items)
    tp, fp, seen = np.zeros(n), np.zeros(n), []
    for i, it in enumerate(items):
        if it["mpjpe"] < threshold and it["gt_id"] not in seen:
            tp[i] = 1
            seen.append(it["gt_id"])
        else:
            fp[i] = 1
    tp, fp = np.cumsum(tp), np.cumsum(fp)
    recall = tp / (total_gt + 1e-5)
    precise = tp / (tp + fp + 1e-5)
    for k in range(n - 2, -1, -1):
        precise[k] = max(precise[k], precise[k + 1])
    precise = np.concatenate(([0], precise, [0]))
    recall = np.concatenate(([0], recall, [1]))
    idx = np.where(recall[1:] != recall[:-1])[0]
    return float(np.sum((recall[idx + 1] - recall[idx]) * precise[idx + 1])), float(recall[-2])


def eval_list_to_mpjpe(eval_list, threshold=500, method="score_sort"):
    """panoptic.py:743-758."""
    seen, vals = [], []
    for it in _sorted(eval_list, method):
        if it["mpjpe"] < threshold and it["gt_id"] not in seen:
            vals.append(it["mpjpe"])
            seen.append(it["gt_id"])
    return float(np.mean(vals)) if vals else float("inf")


def eval_list_to_recall(eval_list, total_gt, threshold=500):
    """panoptic.py:760-764."""
    return len({it["gt_id"] for it in eval_list if it["mpjpe"] < threshold}) / total_gt


def evaluate_panoptic(preds, gts, gts_vis, method="score_sort"):
    ev, total_gt = match_predictions(preds, gts, gts_vis, method)
    aps, recs = [], []
    for t in MPJPE_THRESHOLDS:
        a, r = eval_list_to_ap(ev, total_gt, t, method)
        aps.append(a)
        recs.append(r)
    return aps, recs, eval_list_to_mpjpe(ev, method=method), eval_list_to_recall(ev, total_gt)


PCP_LIMBS = [[0, 1], [1, 2], [3, 4], [4, 5], [6, 7], [7, 8], [9, 10], [10, 11], [12, 13]]   # shelf.py:270-271


def evaluate_pcp(preds, actor_gts, recall_threshold=500, alpha=0.5):
    """shelf.py:255-330 on in-memory ground truth (actor_gts[person][frame] = (14,3) mm or empty)."""
    P = len(actor_gts)
    correct, total = np.zeros(P), np.zeros(P)
    bone = np.zeros((P, 10))
    total_gt = match_gt = 0
    for f, pred in enumerate(preds):
        pred = np.asarray(pred, dtype=np.float64)
        pred = pred[pred[:, 0, 3] >= 0, :, :3]
        for person in range(P):
            gt = actor_gts[person][f]
            if gt is None or len(gt) == 0:
                continue
            gt = np.asarray(gt, dtype=np.float64)
            errs = [np.mean(np.sqrt(np.sum((gt - p) ** 2, axis=-1))) for p in pred]
            n = int(np.argmin(errs))
            if errs[n] < recall_threshold:
                match_gt += 1
            total_gt += 1
            for j, (a, b) in enumerate(PCP_LIMBS):
                total[person] += 1
                es = np.linalg.norm(pred[n, a] - gt[a])
                ee = np.linalg.norm(pred[n, b] - gt[b])
                if (es + ee) / 2.0 <= alpha * np.linalg.norm(gt[a] - gt[b]):
                    correct[person] += 1
                    bone[person, j] += 1
            ph, gh = (pred[n, 2] + pred[n, 3]) / 2.0, (gt[2] + gt[3]) / 2.0
            total[person] += 1
            if (np.linalg.norm(ph - gh) + np.linalg.norm(pred[n, 12] - gt[12])) / 2.0 <= alpha * np.linalg.norm(gh - gt[12]):
                correct[person] += 1
                bone[person, 9] += 1
    actor_pcp = correct / (total + 1e-8)
    groups = [("Head", [8]), ("Torso", [9]), ("Upper arms", [5, 6]), ("Lower arms", [4, 7]), ("Upper legs", [1, 2]),
              ("Lower legs", [0, 3])]
    bone_pcp = {k: np.sum(bone[:, v], axis=-1) / (total / 10 * len(v) + 1e-8) for k, v in groups}
    return actor_pcp, float(np.mean(actor_pcp[:3])), bone_pcp, match_gt / (total_gt + 1e-8)
