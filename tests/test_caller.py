"""CPU tests of the caller-side glue (mvgformer_amd.caller), pinned to tests/golden/caller.npz: arrays produced by
RUNNING the reference's own DyanmicQueryTransformer.forward (lib/models/dq_transformer.py:335-755) and validate_3d
(lib/core/function.py:329-585) in the build container (tests/golden/make_golden_caller.py)."""
import os

import numpy as np
import pytest
import torch

from tests.golden.cases import CALLER_CASE, caller_embeddings

GOLD = os.path.join(os.path.dirname(__file__), "golden", "caller.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_inverse_sigmoid_equals_reference_values(gold):
    """lib/models/util/misc.py:608-612 on edge values (0, eps, 1, out-of-range)."""
    from mvgformer_amd import caller
    y = caller.inverse_sigmoid(_t(gold["inverse_sigmoid/x"]))
    assert torch.equal(y, _t(gold["inverse_sigmoid/y"]))


def test_query_embedding_sum_and_split_equal_reference(gold):
    """dq_transformer.py:394-432: the decoder's tgt / query_pos as the reference model built them."""
    from mvgformer_amd import caller
    je, ie = caller_embeddings(12)
    qpos, tgt = caller.person_joint_queries(je, ie, CALLER_CASE["B"])
    for b in range(CALLER_CASE["batches"]):
        for i in range(CALLER_CASE["B"]):
            assert torch.equal(tgt[i], _t(gold["panoptic/b%d/tgt" % b])[0])
            assert torch.equal(qpos[i], _t(gold["panoptic/b%d/query_pos" % b])[0])


def test_sample_space_reference_points_equal_reference(gold):
    """dq_transformer.py:298-323 + generate_T_pose :225-236 + norm2absolute, NQ = 12 (4 x 4 grid, first 12 cells)."""
    from mvgformer_amd import caller
    from mvgformer_amd.synthetic import CONFIGS
    c = CONFIGS[CALLER_CASE["config"]]
    want = _t(gold["panoptic/b0/reference_points"])
    got = caller.sample_space_reference_points(12, c["space_size"], c["space_center"], CALLER_CASE["B"], "cpu",
                                               t_pose=_t(gold["tpose"]))
    assert got.dtype == torch.float32 and torch.equal(got, want)
    # the built-in T-pose constant is the reference's tpose.pt to 1e-12 mm -> same fp32 points up to one ulp
    dflt = caller.sample_space_reference_points(12, c["space_size"], c["space_center"], CALLER_CASE["B"], "cpu")
    assert float((dflt - want).abs().max()) <= 1.3e-4
    assert float(np.abs(gold["tpose"] - caller.TPOSE_MM).max()) < 1e-12


@pytest.mark.parametrize("fmt", ["panoptic", "shelf"])
def test_out_dict_and_packed_predictions_equal_reference(gold, fmt):
    """dq_transformer.py:569-603 (incl. the Shelf/Campus permutation) and function.py:386-396, fed with the
    REFERENCE decoder's raw outputs: every array of the reference's out dict and validate_3d's packed preds, exactly."""
    from mvgformer_amd import caller
    conv = None if fmt == "panoptic" else [int(i) for i in gold["convert_joint_format_indices"]]
    thr = float(gold["threshold"])
    for b in range(CALLER_CASE["batches"]):
        pre = "panoptic/b%d/" % b
        hs = torch.zeros(*[int(n) for n in gold[pre + "hs_shape"]][:3], 1)
        cls = [c for c in _t(gold[pre + "dec_cls"])]
        out = caller.decoder_outputs_to_dict(hs, _t(gold[pre + "dec_refs"]), _t(gold[pre + "dec_refs2d"]),
                                             _t(gold[pre + "dec_projs2d"]), cls, 12, 15, conv)
        pred = caller.pack_predictions(out, thr)
        want = _t(gold["%s/b%d/pred" % (fmt, b)])
        assert pred.shape == want.shape and torch.equal(pred, want)
        assert set(np.unique(want[..., 3].numpy())) == {-1.0, 0.0}          # both outcomes of (score > thr) - 1
        if b == CALLER_CASE["batches"] - 1:
            assert torch.equal(out["pred_logits"], _t(gold[fmt + "/out/pred_logits"]))
            assert torch.equal(out["pred_poses"]["outputs_coord"], _t(gold[fmt + "/out/pred_poses"]))
            assert torch.equal(out["pred_poses_2d"]["outputs_coord_2d"], _t(gold[fmt + "/out/pred_poses_2d"]))
            assert torch.equal(out["pred_poses_2d_proj"]["outputs_coord_2d_proj"], _t(gold[fmt + "/out/pred_poses_2d_proj"]))


def test_level_tables_built_by_the_head_equal_reference(gold):
    """dq_transformer.py:360-388: spatial shapes / level starts derived from the backbone's maps."""
    from mvgformer_amd import caller
    from mvgformer_amd.synthetic import build_case
    case = build_case(CALLER_CASE["config"], B=CALLER_CASE["B"], seed=CALLER_CASE["seed"], layers=1)
    shapes, starts = caller.level_tables(case.src_views)
    assert torch.equal(shapes, _t(gold["panoptic/b0/spatial_shapes"]))
    assert torch.equal(starts, _t(gold["panoptic/b0/level_start_index"]))

from mvgformer_amd import caller
from mvgformer_amd.synthetic import CONFIGS, init_reference_points


def test_inverse_sigmoid_roundtrip_and_clamps():
    p = torch.tensor([0.0, 1e-7, 0.1, 0.5, 0.9, 1.0, 1.5, -0.2])
    y = caller.inverse_sigmoid(p)
    assert torch.allclose(torch.sigmoid(y)[2:5], p[2:5], atol=1e-6)
    assert torch.isfinite(y).all()
    assert float(y[0]) == float(torch.log(torch.tensor(1e-5 / 1.0)))         # eps clamp (util/misc.py:608-612)


def test_person_joint_queries_layout():
    J, NQ, C = 15, 7, 8
    je, ie = torch.randn(J, 2 * C), torch.randn(NQ, 2 * C)
    qpos, tgt = caller.person_joint_queries(je, ie, batch=3)
    assert qpos.shape == tgt.shape == (3, NQ * J, C)
    i, j = 4, 9                                                               # token q = i*J + j
    assert torch.equal(qpos[1, i * J + j], (je[j] + ie[i])[:C])
    assert torch.equal(tgt[2, i * J + j], (je[j] + ie[i])[C:])


def test_sample_space_reference_points_match_synthetic_generator():
    c = CONFIGS["cfg2"]
    for NQ in (1024, 100, 7):
        a = caller.sample_space_reference_points(NQ, c["space_size"], c["space_center"], 2, "cpu")
        b = init_reference_points(2, NQ, c["space_size"], c["space_center"], jitter=0.0)
        assert a.shape == (2, NQ * 15, 3)
        assert float((a - b).abs().max()) < 1e-3                              # fp32 vs fp64 linspace


def test_output_dict_and_prediction_packing():
    Ly, B, V, NQ, J = 2, 2, 3, 4, 15
    hs = torch.randn(Ly, B, NQ * J, 8)
    refs = torch.randn(Ly, B, NQ * J, 3)
    r2d = torch.randn(Ly, B, V, NQ * J, 2)
    p2d = torch.randn(Ly, B, V, NQ * J, 2)
    cls = [torch.rand(B, NQ, 2) for _ in range(Ly)]
    out = caller.decoder_outputs_to_dict(hs, refs, r2d, p2d, cls, NQ, J)
    assert torch.allclose(out["pred_logits"].sigmoid(), cls[-1], atol=1e-5)
    assert torch.equal(out["pred_poses"]["outputs_coord"], refs[-1])
    pred = caller.pack_predictions(out, 0.5)
    assert pred.shape == (B, NQ, J, 5)
    score = cls[-1][:, :, 1]
    assert torch.allclose(pred[..., 4], score[:, :, None].expand(-1, -1, J), atol=1e-5)
    assert torch.equal(pred[..., 3], (pred[..., 4] > 0.5).float() - 1)
    assert torch.equal(pred[..., :3], refs[-1].view(B, NQ, J, 3))
    # Shelf/Campus: 14-joint permutation (dq_transformer.py:584-597)
    idx = [14, 13, 12, 6, 7, 8, 11, 10, 9, 3, 4, 5, 0, 1]
    out14 = caller.decoder_outputs_to_dict(hs, refs, r2d, p2d, cls, NQ, J, idx)
    assert out14["pred_poses"]["outputs_coord"].shape == (B, NQ * 14, 3)
    assert torch.equal(out14["pred_poses"]["outputs_coord"].view(B, NQ, 14, 3)[:, :, 0], refs[-1].view(B, NQ, J, 3)[:, :, 14])
    assert caller.pack_predictions(out14, 0.5).shape == (B, NQ, 14, 5)


def test_decoder_head_state_dict_names():
    from mvgformer_amd.factory import build_decoder_for_case
    from mvgformer_amd.synthetic import build_case
    case = build_case("mini5", with_features=False)
    head = caller.DecoderHead(build_decoder_for_case(case, "cpu"), case.NQ, 15, 256, case.space_size, case.space_center)
    keys = set(head.state_dict())
    assert "joint_embedding.weight" in keys and "instance_embedding.weight" in keys
    assert "decoder.layers.0.proj_attn.rayconv.weight" in keys and "decoder.layers.1.class_embed.bias" in keys
    assert head.joint_embedding.weight.shape == (15, 512) and head.instance_embedding.weight.shape == (case.NQ, 512)
