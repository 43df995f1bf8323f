"""Seeded synthetic scenes for the evaluation post-processing tests (inputs only; the expected outputs in
tests/golden/eval.npz come from the reference's functions, make_golden_eval.py)."""
import numpy as np

from mvgformer_amd.synthetic import TPOSE_MM

NMS_CASES = {
    "default": dict(seed=1, frames=4, dist_thr=0.3, num_nearby=7),
    "tight": dict(seed=2, frames=3, dist_thr=0.05, num_nearby=3),
    "loose_maxdets": dict(seed=3, frames=3, dist_thr=0.8, num_nearby=10, max_dets=3),
    "none_thr": dict(seed=4, frames=2, dist_thr=0.3, num_nearby=None),
}


def _person(rng, J=15):
    """a T-pose skeleton at a random place / heading in an 8 x 8 m space (mm)."""
    base = np.asarray(TPOSE_MM, dtype=np.float64)[:J]
    base = base - base.mean(0)
    a = rng.uniform(0, 2 * np.pi)
    R = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
    return base @ R.T + np.array([rng.uniform(-3000, 3000), rng.uniform(-3000, 3000), rng.uniform(800, 1200)])


def panoptic_scene(seed, frames=6, J=15):
    """Returns (preds, gts, gts_vis): per frame (N, J, 5) candidates [x, y, z, flag, score] -- noisy copies of
    the ground truth, near-duplicates (what the NMS is for), far false positives, candidates below the
    classification threshold (flag = -1) -- and the ground truth (G, J, 3) / visibility (G, J, 3)."""
    rng = np.random.default_rng(seed)
    preds, gts, vis = [], [], []
    for f in range(frames):
        G = int(rng.integers(0, 5)) if f == 2 else int(rng.integers(1, 6))
        if f == 2:
            G = 0                                               # a frame without ground truth (skipped by evaluate)
        gt = np.stack([_person(rng, J) for _ in range(G)]) if G else np.zeros((0, J, 3))
        v = (rng.uniform(size=(G, J, 1)) > 0.15).astype(np.float64).repeat(3, axis=2)
        if G:
            v[:, 0] = 1.0                                       # every person keeps at least one visible joint
        cand = []
        for g in gt:
            for _ in range(int(rng.integers(1, 4))):            # 1-3 detections of the same person
                noise = rng.normal(0, rng.choice([8.0, 30.0, 90.0]), size=(J, 3))
                cand.append(g + noise)
        for _ in range(int(rng.integers(1, 4))):                # false positives
            cand.append(_person(rng, J) + rng.normal(0, 200.0, size=(J, 3)))
        cand = np.stack(cand)
        N = len(cand)
        score = rng.permutation(N).astype(np.float64) / N * 0.9 + 0.05 + rng.uniform(0, 1e-3, size=N)   # distinct
        flag = np.where(score > 0.2, 0.0, -1.0)
        p = np.zeros((N, J, 5))
        p[:, :, :3] = cand
        p[:, :, 3] = flag[:, None]
        p[:, :, 4] = score[:, None]
        preds.append(p)
        gts.append(gt)
        vis.append(v)
    return preds, gts, vis


def pcp_scene(seed, J=14, actors=4, frames=7):
    """Shelf / Campus style: `actors` annotated people (actor k missing in some frames), evaluated on a subset of
    the recording's frames.  Returns (preds, actor_gts, frame_range, n_frames_total); actor_gts[a][i] is the
    (14, 3) ground truth in mm of actor a in evaluated frame i, or None."""
    rng = np.random.default_rng(seed)
    n_total = 3 * frames
    frame_range = sorted(rng.choice(n_total, size=frames, replace=False).tolist())
    actor_gts = [[None] * frames for _ in range(actors)]
    preds = []
    for i in range(frames):
        cand = []
        for a in range(actors):
            if rng.uniform() < 0.3 and not (a == 0 and i == 0):
                continue
            g = _person(rng, J)
            actor_gts[a][i] = g
            cand.append(g + rng.normal(0, rng.choice([15.0, 60.0, 150.0]), size=(J, 3)))
        cand.append(_person(rng, J))                             # a false positive (also keeps every frame non-empty)
        cand = np.stack(cand)
        N = len(cand)
        p = np.zeros((N, J, 5))
        p[:, :, :3] = cand
        score = rng.uniform(0.3, 1.0, size=N)
        p[:, :, 4] = score[:, None]
        p[-1, :, 3] = -1.0 if rng.uniform() < 0.5 and N > 1 else 0.0
        preds.append(p)
    return preds, actor_gts, frame_range, n_total
