"""CPU-only checks of the C-ABI boundary: the shared library loads, exports every symbol
include/mvg_decoder.h declares, the ctypes table covers them all, and the host-side mirror
has the reference's class surface.  No kernels are launched."""
import ctypes
import os
import re

import pytest
import torch

import mvgformer_amd
from mvgformer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "mvg_decoder.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mvg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), "missing export %s" % s


def test_ctypes_table_matches_header():
    declared = set(_header_symbols()) - {"mvg_version"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().mvg_version().startswith(b"mvgformer_amd")


def test_camera_stride_matches_header():
    src = open(os.path.join(ROOT, "include", "mvg_decoder.h")).read()
    assert int(re.search(r"#define MVG_CAM_STRIDE (\d+)", src).group(1)) == _lib.CAM_STRIDE


def test_state_dict_surface_matches_reference():
    """32 entries / 1 169 861 parameters per layer, names per SURVEY.md section 8(b)."""
    from mvgformer_amd.factory import build_decoder_for_case
    from mvgformer_amd.synthetic import build_case
    case = build_case("mini5", with_features=False)
    dec = build_decoder_for_case(case, device="cpu")
    sd = dec.state_dict()
    per_layer = [k for k in sd if k.startswith("layers.0.")]
    assert len(per_layer) == 32
    assert sum(p.numel() for p in dec.layers[0].parameters()) == 1169861
    assert sd["layers.0.proj_attn.sampling_offsets.weight"].shape == (128, 256)
    assert sd["layers.0.proj_attn.attention_weights.weight"].shape == (64, 256)
    assert sd["layers.0.pose_embed.MLP.layers.2.weight"].shape == (3, 256)
    assert sd["layers.0.self_attn.in_proj_weight"].shape == (768, 256)
    assert sd["layers.0.class_embed.weight"].shape == (2, 256)
    assert mvgformer_amd.MSDeformAttn is mvgformer_amd.ProjAttn
    assert mvgformer_amd.MultiViewDecoderLayer is mvgformer_amd.DQDecoderLayer


def test_cpu_tensors_raise_like_the_reference_stub():
    # lib/models/ops/src/deform.h:49 -> "Not implemented on the CPU"
    from mvgformer_amd import deformable
    from tests.golden.cases import msda_case
    c = msda_case("edge_f32")
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        deformable.deform_forward(c["value"], c["shapes"], c["starts"], c["loc"], c["weight"], 64)


def test_reference_yaml_entry_point_is_readable(tmp_path):
    """the panoptic YAML keys the decoder needs (configs/panoptic/knn5-lr4-q1024-g8.yaml)."""
    from mvgformer_amd.factory import build_decoder_from_cfg, load_yaml_config
    y = tmp_path / "cfg.yaml"
    y.write_text("""
DATASET: {CAMERA_NUM: 5}
NETWORK: {IMAGE_SIZE: [960, 512]}
MULTI_PERSON: {SPACE_SIZE: [8000.0, 8000.0, 2000.0], SPACE_CENTER: [0.0, -500.0, 800.0]}
DECODER: {d_model: 256, nhead: 8, dim_feedforward: 1024, num_feature_levels: 1, dec_n_points: 8,
          num_decoder_layers: 4, num_instance: 1024, num_keypoints: 15, feature_update_method: MLP,
          init_self_attention: false, open_forward_ffn: true, projattn_posembed_mode: ablation_not_use_rayconv}
""")
    cfg = load_yaml_config(str(y))
    dec = build_decoder_from_cfg(cfg)
    assert len(dec.layers) == 4 and dec.layers[0].proj_attn.n_points == 8


def test_every_reference_yaml_entry_point_builds_a_decoder():
    """The reference's REAL YAML entry points (configs/panoptic/knn5-lr4-q1024-g8.yaml and the 11 others).  In the build
    container they are opened where they lie under /root/reference and must (a) load through factory.load_yaml_config, (b)
    build a decoder on the supported hot path and (c) equal the committed extract of their values
    (mvgformer_amd/data/yaml_extract.json, made by tests/golden/make_yaml_extract.py); on the GPU box, where the reference does
    not exist, the decoders are built from the extract alone."""
    import json
    import os
    from types import SimpleNamespace
    from mvgformer_amd.factory import build_decoder_from_cfg, load_yaml_config
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(os.path.dirname(here), "mvgformer_amd", "data", "yaml_extract.json")) as f:
        extract = json.load(f)
    assert "configs/panoptic/knn5-lr4-q1024-g8.yaml" in extract and len(extract) == 12
    ref = os.environ.get("MVG_REFERENCE", "/root/reference")
    for rel, val in extract.items():
        if os.path.isdir(ref):
            from tests.golden.make_yaml_extract import flatten
            assert flatten(load_yaml_config(os.path.join(ref, rel))) == val, rel
        cfg = SimpleNamespace(DECODER=SimpleNamespace(**val["DECODER"]), NETWORK=SimpleNamespace(IMAGE_SIZE=val["IMAGE_SIZE"]),
                              MULTI_PERSON=SimpleNamespace(SPACE_SIZE=val["SPACE_SIZE"], SPACE_CENTER=val["SPACE_CENTER"]),
                              DATASET=SimpleNamespace(CAMERA_NUM=val["CAMERA_NUM"]))
        dec = build_decoder_from_cfg(cfg)
        assert len(dec.layers) == val["DECODER"]["num_decoder_layers"] == 4
        layer = dec.layers[0]
        layer._check_supported()                                  # every shipped YAML is on the built hot path
        assert layer.proj_attn.n_points == 8 and layer.proj_attn.n_heads == 8 and layer.num_joints == 15
        assert len(layer.state_dict()) == 32


def test_host_level_tables_are_cached_per_tensor_and_follow_in_place_updates():
    """ops.host_levels keeps the host copy of (spatial_shapes, level_start_index) on the tensor object (so the deterministic
    backward sizes its workspace without a D2H sync per call) and drops it when the tensor is modified in place."""
    import torch
    from mvgformer_amd import ops
    shapes = torch.tensor([[8, 12], [4, 6]], dtype=torch.long)
    starts = torch.tensor([0, 96], dtype=torch.long)
    a = ops.host_levels(shapes, starts)
    b = ops.host_levels(shapes, starts)
    assert a[0] is b[0] and a[1] is b[1]                       # second lookup: the cached ctypes arrays
    assert list(a[0]) == [8, 12, 4, 6] and list(a[1]) == [0, 96]
    shapes[1, 0] = 5                                           # bumps the version counter
    c = ops.host_levels(shapes, starts)
    assert c[0] is not a[0] and list(c[0]) == [8, 12, 5, 6] and c[1] is a[1]
    other = ops.host_levels(shapes.clone(), starts)            # a different tensor object never sees somebody else's copy
    assert other[0] is not c[0] and list(other[0]) == [8, 12, 5, 6]


def test_only_the_product_and_checker_libraries_ship():
    """The tree that travels to the GPU box holds exactly two kinds of shared objects: the product (libmvgformer_hip.so) and the
    oracle's C checker (libmsda_ref.so, oracle/_ref/*).  Experiment builds left in the package directory would ship with every
    push and could be loaded by accident."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    allowed = {os.path.join("mvgformer_amd", "libmvgformer_hip.so"), os.path.join("oracle", "libmsda_ref.so")}
    found = set()
    for base, dirs, files in os.walk(root):
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "__pycache__", ".pytest_cache", "_ref", "build")]
        for name in files:
            if name.endswith(".so") or ".so." in name:
                found.add(os.path.relpath(os.path.join(base, name), root))
    assert found <= allowed, sorted(found - allowed)


def test_deepcopy_of_a_decoder_does_not_copy_run_time_buffers():
    """copy.deepcopy(decoder) (EMA / checkpoint copies): parameters are copied, the pooled fp32 (value, G) buffers, weight caches and
    side-stream hand-offs are not; the copy's layers share ONE fresh pool again."""
    import copy
    from mvgformer_amd.factory import build_decoder_for_case
    from mvgformer_amd.synthetic import build_case
    dec = build_decoder_for_case(build_case("cfg1", seed=0, layers=2), "cpu", torch.float32)
    pa0 = dec.layers[0].proj_attn
    pa0._f32_pool["slot"] = (torch.zeros(4), torch.zeros(4))
    pa0._vp, pa0._G = torch.zeros(3), torch.zeros(3)
    dec.layers[0]._after_chain_b = None
    cp = copy.deepcopy(dec)
    assert cp.layers[0].proj_attn._f32_pool == {} and cp.layers[0].proj_attn._f32_pool is cp.layers[1].proj_attn._f32_pool
    assert cp.layers[0].proj_attn._f32_pool is not pa0._f32_pool and cp.layers[0].proj_attn._vp is None
    a, b = dec.state_dict(), cp.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) and a[k].data_ptr() != b[k].data_ptr() for k in a)
    single = copy.deepcopy(pa0)
    assert single._f32_pool == {} and single._f32_pool is not pa0._f32_pool
