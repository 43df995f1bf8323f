"""Build the decoder for a synthetic case or from the reference's YAML config files."""
from __future__ import annotations

from types import SimpleNamespace

import torch
import yaml

from .decoder import DQDecoder, DQDecoderLayer
from .synthetic import decoder_cfg, to_torch_state

# defaults of lib/core/config.py for the keys the decoder reads (config.py:303-304 etc.)
_DECODER_DEFAULTS = dict(pose_embed_layer=3, d_model=256, dim_feedforward=1024, dropout=0.1, activation="relu",
                         num_feature_levels=1, nhead=8, dec_n_points=8, num_decoder_layers=4,
                         return_intermediate_dec=True, num_instance=1024, num_keypoints=15,
                         detach_refpoints_cameraprj_firstlayer=True, fuse_view_feats="cat_proj",
                         projattn_posembed_mode="ablation_not_use_rayconv", feature_update_method="MLP",
                         init_self_attention=False, open_forward_ffn=True, query_filter_method="threshold",
                         bayesian_update=False, triangulation_method="linalg", filter_query=True,
                         share_layer_weights=False, inference_conf_thr=[0.1])


def load_yaml_config(path):
    """Read the hot-path hyper-parameters from one of the reference's YAML entry points
    (e.g. configs/panoptic/knn5-lr4-q1024-g8.yaml; keys per lib/models/dq_transformer.py:129-157)."""
    with open(path) as f:
        raw = yaml.safe_load(f)
    dec = dict(_DECODER_DEFAULTS)
    dec.update(raw.get("DECODER", {}))
    return SimpleNamespace(
        DECODER=SimpleNamespace(**dec),
        NETWORK=SimpleNamespace(IMAGE_SIZE=list(raw["NETWORK"]["IMAGE_SIZE"])),
        MULTI_PERSON=SimpleNamespace(SPACE_SIZE=list(raw["MULTI_PERSON"]["SPACE_SIZE"]),
                                     SPACE_CENTER=list(raw["MULTI_PERSON"]["SPACE_CENTER"])),
        DATASET=SimpleNamespace(CAMERA_NUM=int(raw["DATASET"]["CAMERA_NUM"])),
        DEBUG=SimpleNamespace(VISUALIZATION_JUMP_NUM=-1),
    )


def build_decoder_from_cfg(cfg):
    """Same construction as lib/models/dq_transformer.py:129-157."""
    d = cfg.DECODER
    layer = DQDecoderLayer(cfg.MULTI_PERSON.SPACE_SIZE, cfg.MULTI_PERSON.SPACE_CENTER, cfg.NETWORK.IMAGE_SIZE,
                           d.pose_embed_layer, d.d_model, d.dim_feedforward, d.dropout, d.activation,
                           d.num_feature_levels, d.nhead, d.dec_n_points, d.detach_refpoints_cameraprj_firstlayer,
                           d.fuse_view_feats, cfg.DATASET.CAMERA_NUM, d.projattn_posembed_mode,
                           d.feature_update_method, d.init_self_attention, d.open_forward_ffn, d.query_filter_method,
                           visualization_jump_num=-1, bayesian_update=d.bayesian_update,
                           triangulation_method=d.triangulation_method, filter_query=d.filter_query,
                           num_joints=d.num_keypoints)
    return DQDecoder(cfg, layer, d.num_decoder_layers, d.return_intermediate_dec)


def build_decoder_for_case(case, device="cuda", dtype=torch.float32):
    """Decoder with the panoptic hyper-parameters and the case's seeded weights."""
    layer = DQDecoderLayer(list(case.space_size), list(case.space_center), list(case.img_size), 3,
                           256, 1024, 0.1, "relu", 1, 8, 8, True, "cat_proj", case.V,
                           "ablation_not_use_rayconv", "MLP", False, True, "threshold",
                           visualization_jump_num=-1, bayesian_update=False, triangulation_method="linalg",
                           filter_query=True, num_joints=15)
    dec = DQDecoder(decoder_cfg(case.space_size, case.space_center), layer, case.layers, True)
    missing, unexpected = dec.load_state_dict(to_torch_state(case.weights), strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    dec = dec.to(device).eval()
    dec.set_compute_dtype(dtype)
    return dec


def case_to_device(case, device="cuda"):
    """Move a synthetic case's tensors to the GPU (meta schema preserved)."""
    mv = lambda t: t.to(device)
    case.tgt, case.query_pos, case.reference_points = mv(case.tgt), mv(case.query_pos), mv(case.reference_points)
    if case.src_views is not None:
        case.src_views = [mv(s) for s in case.src_views]
    case.spatial_shapes, case.level_start_index = mv(case.spatial_shapes), mv(case.level_start_index)
    case.meta = [{k: ({kk: mv(vv) for kk, vv in v.items()} if isinstance(v, dict) else mv(v)) for k, v in m.items()}
                 for m in case.meta]
    return case
