"""Evaluation post-processing: the loop restatement (oracle/eval_ref.py) and the batched
product code (mvgformer_amd/evaluate.py) against golden vectors produced by the reference's own functions
(tests/golden/make_golden_eval.py -> tests/golden/eval.npz)."""
import os

import numpy as np
import pytest
import torch

from mvgformer_amd import evaluate as E
from oracle import eval_ref as O
from tests.golden.eval_cases import NMS_CASES, panoptic_scene, pcp_scene

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval.npz"))
IMPLS = [pytest.param(O, id="oracle"), pytest.param(E, id="product")]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("name", sorted(NMS_CASES))
def test_nearby_joints_nms_matches_reference(impl, name):
    spec = NMS_CASES[name]
    preds, _, _ = panoptic_scene(spec["seed"], frames=spec["frames"])
    for f, p in enumerate(preds):
        keep = impl.nearby_joints_nms(p, spec["dist_thr"], spec["num_nearby"], max_dets=spec.get("max_dets", -1))
        assert list(keep) == GOLD["nms_%s_f%d" % (name, f)].tolist(), (name, f)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("seed", [11, 12])
@pytest.mark.parametrize("method", ["score_sort", "mpjpe_sort"])
def test_panoptic_metrics_match_reference(impl, seed, method):
    preds, gts, vis = panoptic_scene(seed, frames=6)
    aps, recs, mpjpe, rec500 = impl.evaluate_panoptic(preds, gts, vis, method=method)
    got = np.asarray(list(aps) + list(recs) + [mpjpe, rec500])
    np.testing.assert_allclose(got, GOLD["pan_%d_%s" % (seed, method)], rtol=1e-12, atol=1e-12)
    assert 0 < max(aps) <= 1.0 + 1e-9 and np.isfinite(mpjpe)


@pytest.mark.parametrize("seed", [11, 12])
def test_filter_nms_evaluate_pipeline_matches_reference(seed):
    """validate_3d.py:228-236: classification filter -> NMS(0.3, 7) -> Panoptic.evaluate."""
    preds, gts, vis = panoptic_scene(seed, frames=6)
    kept = [E.filter_and_nms(p) for p in preds]
    assert sum(len(k) for k in kept) < sum(len(p) for p in preds)
    aps, recs, mpjpe, rec500 = E.evaluate_panoptic(kept, gts, vis)
    np.testing.assert_allclose(np.asarray(list(aps) + list(recs) + [mpjpe, rec500]), GOLD["pan_%d_nms" % seed],
                               rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("seed", [21, 22])
def test_pcp_matches_reference(impl, seed):
    preds, actors, _, _ = pcp_scene(seed)
    actor_pcp, avg_pcp, bone, recall = impl.evaluate_pcp(preds, actors)
    got = np.concatenate([np.asarray(actor_pcp), [avg_pcp, recall], np.concatenate([np.asarray(v) for v in bone.values()])])
    np.testing.assert_allclose(got, GOLD["pcp_%d" % seed], rtol=1e-12, atol=1e-12)


def test_edge_cases():
    assert E.nearby_joints_nms(np.zeros((0, 15, 5)), 0.3, 7) == [] and O.nearby_joints_nms(np.zeros((0, 15, 5)), 0.3, 7) == []
    with pytest.raises(AssertionError):
        E.nearby_joints_nms(np.zeros((2, 15, 5)), 0.0, 7)
    with pytest.raises(AssertionError):
        E.nearby_joints_nms(np.zeros((2, 15, 5)), 0.3, 15)
    # a single candidate survives; two identical candidates collapse onto the better scored one
    one = np.random.default_rng(0).normal(size=(1, 15, 5))
    assert E.nearby_joints_nms(one, 0.3, 7) == [0]
    two = np.concatenate([one, one])
    two[1, :, 4] = two[0, 0, 4] + 1.0
    assert E.nearby_joints_nms(two, 0.3, 7) == [1] == O.nearby_joints_nms(two, 0.3, 7)
    # no prediction passes the classification filter: empty eval list -> AP 0, MPJPE inf, recall 0
    preds, gts, vis = panoptic_scene(11, frames=3)
    for p in preds:
        p[:, :, 3] = -1.0
    aps, recs, mpjpe, rec500 = E.evaluate_panoptic(preds, gts, vis)
    assert aps == [0.0] * 6 and recs == [0.0] * 6 and mpjpe == float("inf") and rec500 == 0.0
    assert O.evaluate_panoptic(preds, gts, vis)[2] == float("inf")


def test_decoder_predictions_flow_into_the_metrics():
    """caller.pack_predictions rows ([x, y, z, (score > thr) - 1, score]) are the evaluate input format."""
    from mvgformer_amd.caller import pack_predictions
    preds, gts, vis = panoptic_scene(12, frames=2)
    p = torch.as_tensor(preds[0])
    N, J = p.shape[:2]
    score = p[:, 0, 4].clamp(1e-6, 1 - 1e-6)
    logits = torch.stack([torch.zeros_like(score), torch.log(score / (1 - score))], -1)[None]      # sigmoid -> score
    out = {"pred_logits": logits, "pred_poses": {"outputs_coord": p[None, :, :, :3].reshape(1, N * J, 3)}}
    packed = pack_predictions(out, 0.2)                                                     # (B, NQ, J, 5)
    assert tuple(packed.shape) == (1, N, J, 5)
    row = packed[0].double().numpy()
    assert np.array_equal(row[:, 0, 3] >= 0, preds[0][:, 0, 4] > 0.2)
    kept = E.filter_and_nms(row)
    assert 0 < len(kept) <= int((row[:, 0, 3] >= 0).sum())


@pytest.mark.gpu
def test_evaluate_on_device_tensors_matches_reference():
    """same golden vectors with the predictions resident on the MI355X (the dense parts run there)."""
    dev = torch.device("cuda")
    spec = NMS_CASES["default"]
    preds, _, _ = panoptic_scene(spec["seed"], frames=spec["frames"])
    for f, p in enumerate(preds):
        keep = E.nearby_joints_nms(torch.as_tensor(p, device=dev), spec["dist_thr"], spec["num_nearby"])
        assert keep == GOLD["nms_default_f%d" % f].tolist()
    preds, gts, vis = panoptic_scene(11, frames=6)
    aps, recs, mpjpe, rec500 = E.evaluate_panoptic([torch.as_tensor(p, device=dev) for p in preds], gts, vis)
    np.testing.assert_allclose(np.asarray(list(aps) + list(recs) + [mpjpe, rec500]), GOLD["pan_11_score_sort"],
                               rtol=1e-10, atol=1e-10)
    # many candidates: 1024 poses (the full query set) -- no (N, N, J, 3) temporary, identical keep list as the loops
    rng = np.random.default_rng(5)
    big = rng.normal(0, 1500.0, size=(1024, 1, 3)) + rng.normal(0, 300.0, size=(1024, 15, 3))
    db = np.zeros((1024, 15, 5))
    db[:, :, :3] = big
    db[:, :, 4] = rng.permutation(1024)[:, None] / 1024.0
    k_dev = E.nearby_joints_nms(torch.as_tensor(db, device=dev), 0.3, 7)
    k_cpu = E.nearby_joints_nms(db, 0.3, 7)
    assert k_dev == k_cpu and 0 < len(k_dev) < 1024
