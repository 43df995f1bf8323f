"""CPU tests of the caller-side glue (mvgformer_amd.caller) against the formulas of the reference's
DyanmicQueryTransformer.forward / validate_3d."""
import numpy as np
import torch

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
