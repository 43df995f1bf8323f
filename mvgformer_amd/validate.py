"""Decoder-side equivalent of the reference's inference entry point ``run/validate_3d.py`` (SURVEY.md section 8 b, "Entry
points"): reads the SAME YAML files, builds the decoder head behind the reference's class surface, optionally fills it from
a published checkpoint, runs the frames through the HIP path and post-processes the predictions like
``validate_3d.py:185-284`` (classification filter at every ``DECODER.inference_conf_thr``, nearby-joints NMS with the
default 0.3 m / 7 joints, AP / recall / MPJPE or PCP when ground truth is given).

    python -m mvgformer_amd.validate --cfg configs/panoptic/knn5-lr4-q1024-g8.yaml --model_path model_best.pth.tar
    python -m mvgformer_amd.validate --cfg extract:configs/panoptic/knn5-lr4-q1024-g8.yaml --frames 8        # GPU box

The backbone (PoseResNet-50) and the dataset loaders are out of scope: the frames are either a
``--frames-npz`` file holding what the backbone hands to the decoder (``feat0..feat{L-1}`` (F, V, C, H_l, W_l), the per-view
camera arrays of ``meta`` and optionally ``joints_3d`` / ``joints_3d_vis``), or seeded synthetic frames of the YAML's
geometry (``mvgformer_amd.synthetic``).  ``--cfg extract:<relative path>`` reads the values of the reference's YAML from
``mvgformer_amd/data/yaml_extract.json`` where the reference tree does not exist.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_config(spec):
    """YAML path -> cfg namespace; ``extract:<rel>`` -> the same namespace from the committed extract of that file."""
    from .factory import load_yaml_config
    if not spec.startswith("extract:"):
        return load_yaml_config(spec)
    with open(os.path.join(ROOT, "mvgformer_amd", "data", "yaml_extract.json")) as f:
        val = json.load(f)[spec[len("extract:"):]]
    return SimpleNamespace(DECODER=SimpleNamespace(**val["DECODER"]), NETWORK=SimpleNamespace(IMAGE_SIZE=val["IMAGE_SIZE"]),
                           MULTI_PERSON=SimpleNamespace(SPACE_SIZE=val["SPACE_SIZE"], SPACE_CENTER=val["SPACE_CENTER"]),
                           DATASET=SimpleNamespace(CAMERA_NUM=val["CAMERA_NUM"]),
                           DEBUG=SimpleNamespace(VISUALIZATION_JUMP_NUM=-1))


def build_head(cfg, device, dtype):
    """DecoderHead with the YAML's hyper-parameters (dq_transformer.py:129-205 without backbone / criterion)."""
    from .caller import DecoderHead
    from .factory import build_decoder_from_cfg
    d = cfg.DECODER
    conv = getattr(d, "convert_joint_format_indices", None)
    head = DecoderHead(build_decoder_from_cfg(cfg), d.num_instance, d.num_keypoints, d.d_model,
                       cfg.MULTI_PERSON.SPACE_SIZE, cfg.MULTI_PERSON.SPACE_CENTER, conv)
    head = head.to(device).eval()
    head.decoder.set_compute_dtype(dtype)
    return head


def load_checkpoint(head, path):
    """``model.load_state_dict(torch.load(path), strict=False)`` (validate_3d.py:160-166): the decoder head takes the keys it
    owns (``decoder.*``, ``joint_embedding.*``, ``instance_embedding.*``; a DDP ``module.`` prefix is dropped)."""
    sd = torch.load(path, map_location="cpu")
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    own = set(head.state_dict())
    missing, unexpected = head.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    return sorted(missing), sorted(k for k in sd if k not in own)


def synthetic_frames(cfg, n_frames, seed=0):
    """seeded stand-ins for (backbone features, meta) of the YAML's geometry: ring cameras around the space centre"""
    from .synthetic import make_meta, make_pyramid, pyramid_shapes, ring_cameras
    V = cfg.DATASET.CAMERA_NUM
    img_wh = tuple(cfg.NETWORK.IMAGE_SIZE)
    orig_wh = (1920, 1080) if img_wh[0] >= 900 else (int(img_wh[0] * 1.29), int(img_wh[1] * 1.2763))
    shapes = pyramid_shapes(img_wh)
    for f in range(n_frames):
        cams = ring_cameras(V, orig_wh, 1400.0 * orig_wh[0] / 1920.0, 4000.0, cfg.MULTI_PERSON.SPACE_CENTER,
                            (-0.1, 0.05, 0.0), (1e-3, -1e-3), seed + f)
        yield make_pyramid(V, 1, shapes, seed=seed + f), make_meta(cams, 1, orig_wh, img_wh), None


def npz_frames(path):
    z = np.load(path)
    L = len([k for k in z.files if k.startswith("feat")])
    F, V = z["feat0"].shape[:2]
    for f in range(F):
        src = [torch.from_numpy(z["feat%d" % l][f]) for l in range(L)]
        meta = []
        for v in range(V):
            cam = {k: torch.from_numpy(z["camera_" + k][f, v])[None] for k in ("R", "T", "fx", "fy", "cx", "cy", "k", "p")}
            meta.append(dict(camera=cam, center=torch.from_numpy(z["center"][f, v])[None],
                             scale=torch.from_numpy(z["scale"][f, v])[None],
                             inv_affine_trans=torch.from_numpy(z["inv_affine_trans"][f, v])[None]))
        gt = (z["joints_3d"][f], z["joints_3d_vis"][f]) if "joints_3d" in z.files else None
        yield src, meta, gt


def to_device(meta, device):
    mv = lambda t: t.to(device)
    return [{k: ({kk: mv(vv) for kk, vv in v.items()} if isinstance(v, dict) else mv(v)) for k, v in m.items()} for m in meta]


def main(argv=None):
    ap = argparse.ArgumentParser(description="decoder-side validate_3d")
    ap.add_argument("--cfg", required=True, help="YAML entry point of the reference, or extract:<path relative to the reference>")
    ap.add_argument("--model_path", default=None, help="checkpoint (validate_3d.py --model_path)")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--frames", type=int, default=4, help="synthetic frames when no --frames-npz is given")
    ap.add_argument("--frames-npz", default=None)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--pred-out", default=None, help="save the packed predictions per threshold (TEST.PRED_FILE)")
    ap.add_argument("--graph", type=int, default=1, help="1: the decoder forward captured once as a HIP graph and replayed per "
                                                         "frame (mvgformer_amd.serving.GraphedDecoder); 0: eager launches")
    args = ap.parse_args(argv)

    from . import evaluate as E
    cfg = load_config(args.cfg)
    dev = torch.device(args.device)
    if dev.type != "cuda":
        raise SystemExit("Not implemented on the CPU")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(args.seed)
    head = build_head(cfg, dev, dtype)
    report = {"cfg": args.cfg, "dtype": args.dtype, "num_instance": cfg.DECODER.num_instance,
              "views": cfg.DATASET.CAMERA_NUM, "layers": cfg.DECODER.num_decoder_layers}
    if args.model_path:
        missing, ignored = load_checkpoint(head, args.model_path)
        report["checkpoint"] = {"path": args.model_path, "missing_keys": missing[:8], "n_missing": len(missing),
                                "n_ignored_keys_of_other_modules": len(ignored)}
    frames = list(npz_frames(args.frames_npz) if args.frames_npz else synthetic_frames(cfg, args.frames, args.seed))
    results = []
    from . import caller
    from .serving import GraphedDecoder
    for thr in cfg.DECODER.inference_conf_thr:                                   # validate_3d.py:185
        preds, gts, gts_vis, t_dec, n_timed = [], [], [], 0.0, 0
        runner = None
        for fi, (src, meta, gt) in enumerate(frames):
            src = [s.to(dev) for s in src]
            meta = to_device(meta, dev)
            if args.graph and runner is None:
                # queries and initial poses do not depend on the frame (dq_transformer.py:394-432, 298-323): loaded once
                shapes, starts = caller.level_tables(src)
                runner = GraphedDecoder(head.decoder, meta, shapes, starts, 1, head.num_instance, thr)
                qpos, tgt = caller.person_joint_queries(head.joint_embedding.weight, head.instance_embedding.weight, 1)
                ref0 = caller.sample_space_reference_points(head.num_instance, head.space_size, head.space_center, 1, dev,
                                                            t_pose=head.t_pose)
                runner.load(tgt=tgt, query_pos=qpos, reference_points=ref0).capture()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if runner is not None:
                runner.set_cameras(meta).load(src_views=src)
                hs, refs, r2d, p2d, cls = runner.replay()
                out = caller.decoder_outputs_to_dict(hs, refs, r2d, p2d, cls, head.num_instance, head.num_joints,
                                                     head.convert_joint_format_indices)
                pred = caller.pack_predictions(out, thr)                         # function.py:386-396
            else:
                _, pred = head(src, meta, threshold=thr)                         # function.py:372-396
            torch.cuda.synchronize()
            if fi > 0 or len(frames) == 1:          # the first frame builds the weight caches / captures the graph (one-time)
                t_dec += time.perf_counter() - t0
                n_timed += 1
            preds.extend(p for p in pred)
            if gt is not None:
                gts.append(gt[0])
                gts_vis.append(gt[1])
        kept = [E.filter_and_nms(p) for p in preds]                              # validate_3d.py:228-234 (0.3 m, 7 joints)
        row = {"inference_conf_thr": thr, "frames": len(preds),
               "candidates_above_thr": int(sum(int((p[:, 0, 3] >= 0).sum()) for p in preds)),
               "poses_after_nms": int(sum(len(k) for k in kept)),
               "decoder_ms_per_frame": round(1e3 * t_dec / max(n_timed, 1), 3),    # host-timed, incl. the camera H2D copy
               "hip_graph": runner is not None}
        if gts and getattr(cfg.DECODER, "convert_joint_format_indices", None) is None:
            aps, recs, mpjpe, recall500 = E.evaluate_panoptic(kept, gts, gts_vis)   # panoptic.py:493-574
            row.update(AP={str(t): round(100 * a, 2) for t, a in zip(E.MPJPE_THRESHOLDS, aps)},
                       recall={str(t): round(100 * r, 2) for t, r in zip(E.MPJPE_THRESHOLDS, recs)},
                       MPJPE=round(float(mpjpe), 2), recall500=round(100 * float(recall500), 2))
        if args.pred_out:
            np.save("%s-%s.npy" % (args.pred_out, thr), np.stack([p.cpu().numpy() for p in preds]))
        results.append(row)
    report["results"] = results
    print(json.dumps(report))
    return report


if __name__ == "__main__":
    sys.exit(0 if main() else 1)
