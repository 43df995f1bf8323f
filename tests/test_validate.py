"""mvgformer_amd.validate -- the decoder-side equivalent of the reference's run/validate_3d.py entry point (SURVEY.md
section 8 b): same YAML files (or their committed extract where the reference tree does not exist), checkpoint loading
with strict=False, frames -> predictions -> classification filter + NMS."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PANOPTIC = "configs/panoptic/knn5-lr4-q1024-g8.yaml"
SHELF = "configs/shelf_campus/shelf_knn5-lr4-q1024.yaml"


def test_extract_config_equals_the_yaml_file():
    from mvgformer_amd import validate
    from tests.golden.make_yaml_extract import flatten
    ref = os.environ.get("MVG_REFERENCE", "/root/reference")
    for rel, thr in ((PANOPTIC, [0.1]), (SHELF, [0.0])):
        cfg = validate.load_config("extract:" + rel)
        assert cfg.DECODER.num_instance == 1024 and cfg.DECODER.inference_conf_thr == thr
        if os.path.isdir(ref):
            assert flatten(validate.load_config(os.path.join(ref, rel))) == flatten(cfg)
    assert validate.load_config("extract:" + SHELF).DECODER.convert_joint_format_indices[:3] == [14, 13, 12]


def test_checkpoint_loading_takes_the_decoder_keys_and_ignores_the_rest(tmp_path):
    """validate_3d.py:160-166: load_state_dict(torch.load(path), strict=False) of a full-model checkpoint (DDP prefix,
    backbone / criterion keys present)."""
    from types import SimpleNamespace
    from mvgformer_amd import validate
    cfg = validate.load_config("extract:" + PANOPTIC)
    cfg.DECODER = SimpleNamespace(**dict(vars(cfg.DECODER), num_instance=6, num_decoder_layers=2))
    head = validate.build_head(cfg, "cpu", torch.float32)
    sd = {"module." + k: torch.full_like(v, 0.5) for k, v in head.state_dict().items()}
    sd["module.backbone.conv1.weight"] = torch.zeros(4)
    sd["module.criterion.empty_weight"] = torch.zeros(2)
    del sd["module.decoder.layers.1.norm1.weight"]
    path = str(tmp_path / "model_best.pth.tar")
    torch.save(sd, path)
    missing, ignored = validate.load_checkpoint(head, path)
    assert missing == ["decoder.layers.1.norm1.weight"] and len(ignored) == 2
    assert float(head.joint_embedding.weight.mean()) == 0.5
    assert float(head.decoder.layers[0].proj_attn.sampling_offsets.weight.mean()) == 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("rel,joints,thr", [(PANOPTIC, 15, 0.1), (SHELF, 14, 0.0)])
def test_validate_entry_point_runs_the_yaml_configuration(rel, joints, thr, tmp_path):
    out = str(tmp_path / "pred")
    p = subprocess.run([sys.executable, "-m", "mvgformer_amd.validate", "--cfg", "extract:" + rel, "--frames", "2",
                        "--pred-out", out], cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), timeout=900,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    rep = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["num_instance"] == 1024 and rep["layers"] == 4 and rep["views"] == 5
    row = rep["results"][0]
    assert row["inference_conf_thr"] == thr and row["frames"] == 2
    assert 0 <= row["poses_after_nms"] <= row["candidates_above_thr"] <= 2 * 1024
    import numpy as np
    pred = np.load("%s-%s.npy" % (out, thr))
    assert pred.shape == (2, 1024, joints, 5) and np.isfinite(pred).all()
    assert set(np.unique(pred[..., 3])) <= {-1.0, 0.0}
