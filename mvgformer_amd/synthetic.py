"""Seeded synthetic inputs for the MVGFormer decoder hot path.

Everything here is generated with ``numpy.random.RandomState`` so that the same
seed yields bit-identical inputs in the build container (where the golden
fixtures are made against the imported reference) and on the GPU box (where the
reference tree does not exist).  Used by ``tests/``, ``bench.py`` and
``__graft_entry__.smoke()``.

Shapes / schema follow the reference:
  * ``meta`` per-view dict schema: lib/dataset/JointsDataset.py:197-220 (collated
    to a batch dim), camera dict lib/dataset/panoptic.py:395-407.
  * feature pyramid: list of L tensors ``(V*B, C, H_l, W_l)`` view-major
    (lib/models/dq_decoder.py:560).
  * initial reference points "sample_space": lib/models/dq_transformer.py:298-323.
  * ``get_scale``: lib/utils/transforms.py:170-181.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch

# T-pose joint offsets (mm) relative to the root joint -- the data held in the
# reference's ``tpose.pt`` (15x3 float64), loaded at lib/models/dq_transformer.py:182.
TPOSE_MM = np.array([
    [-8.0720, -32.5520, 571.1680],
    [60.7220, 149.9020, 738.6780],
    [0.0, 0.0, 0.0],
    [-164.4850, 40.3180, 565.1980],
    [-240.8650, 30.6990, 320.6780],
    [-50.1000, 157.6670, 396.0380],
    [-84.9300, 59.3810, -4.9090],
    [-85.8840, 12.7500, -397.5280],
    [-74.3460, -30.8230, -712.4110],
    [143.1300, -101.2810, 584.0480],
    [249.9620, -103.4240, 364.7480],
    [192.5020, 82.1720, 451.7580],
    [84.9310, -59.3810, 4.9090],
    [142.1820, -112.1650, -360.1260],
    [177.2020, -227.3750, -712.7630],
], dtype=np.float64)


# --------------------------------------------------------------------------- configs
# The five BASELINE.json configurations (SURVEY.md section 8d "synthetic inputs").
CONFIGS = {
    # CPU reference case: 1 sample, 2 synthetic 256x256 views, 64 queries, 1 layer
    "cfg1": dict(V=2, NQ=64, layers=1, img_wh=(256, 256), orig_wh=(256, 256), focal=300.0,
                 space_size=(2000.0, 2000.0, 1500.0), space_center=(0.0, 0.0, 800.0),
                 k=(-0.1, 0.05, 0.0), p=(1e-3, -1e-3), radius=3500.0),
    # Panoptic CMU0 5 views, 1024 queries, 4 layers (configs/panoptic/knn5-lr4-q1024-g8.yaml)
    "cfg2": dict(V=5, NQ=1024, layers=4, img_wh=(960, 512), orig_wh=(1920, 1080), focal=1400.0,
                 space_size=(8000.0, 8000.0, 2000.0), space_center=(0.0, -500.0, 800.0),
                 k=(-0.1, 0.05, 0.0), p=(1e-3, -1e-3), radius=4000.0),
    # Shelf 5 views, 512 queries, 4 layers, k=p=0 (configs/shelf_campus/shelf_knn5-lr4-q1024.yaml)
    "cfg4": dict(V=5, NQ=512, layers=4, img_wh=(800, 608), orig_wh=(1032, 776), focal=1060.0,
                 space_size=(8000.0, 8000.0, 2000.0), space_center=(450.0, -320.0, 800.0),
                 k=(0.0, 0.0, 0.0), p=(0.0, 0.0), radius=4000.0),
    # stress: 31 views, 2048 queries, 6 layers
    "cfg5": dict(V=31, NQ=2048, layers=6, img_wh=(960, 512), orig_wh=(1920, 1080), focal=1400.0,
                 space_size=(8000.0, 8000.0, 2000.0), space_center=(0.0, -500.0, 800.0),
                 k=(-0.1, 0.05, 0.0), p=(1e-3, -1e-3), radius=4000.0),
    # tiny cases for fixtures / smoke
    "mini5": dict(V=5, NQ=12, layers=2, img_wh=(320, 192), orig_wh=(640, 360), focal=480.0,
                  space_size=(4000.0, 4000.0, 2000.0), space_center=(0.0, -200.0, 800.0),
                  k=(-0.12, 0.06, 0.01), p=(2e-3, -1.5e-3), radius=3200.0),
    # Shelf-like: 3 views, no distortion (data/Shelf/calibration_shelf.json), 4 : 3.04 network image
    "mini3s": dict(V=3, NQ=10, layers=2, img_wh=(400, 304), orig_wh=(516, 388), focal=530.0,
                   space_size=(4000.0, 4000.0, 2000.0), space_center=(450.0, -320.0, 800.0),
                   k=(0.0, 0.0, 0.0), p=(0.0, 0.0), radius=3600.0),
    # 9 views: more than the 8 lanes a (query, joint) problem has in the view softmax / triangulation / next projection
    "mini9": dict(V=9, NQ=8, layers=2, img_wh=(320, 192), orig_wh=(640, 360), focal=480.0,
                  space_size=(4000.0, 4000.0, 2000.0), space_center=(0.0, -200.0, 800.0),
                  k=(-0.12, 0.06, 0.01), p=(2e-3, -1.5e-3), radius=3200.0),
}
CONFIGS["cfg3"] = CONFIGS["cfg2"]  # same workload, queries sharded over 8 GPUs


def pyramid_shapes(img_wh):
    """PoseResNet deconv outputs at 1/4, 1/8, 1/16 of the network image
    (lib/models/pose_resnet.py:198-216; comment lib/models/dq_transformer.py:442-444).
    Returns [(H, W)] * 3."""
    w, h = img_wh
    return [(h // 4, w // 4), (h // 8, w // 8), (h // 16, w // 16)]


def get_scale(image_size, resized_size):
    """Padded-original size / 200 (lib/utils/transforms.py:170-181)."""
    w, h = image_size
    wr, hr = resized_size
    if w / wr < h / hr:
        w_pad, h_pad = h / hr * wr, h
    else:
        w_pad, h_pad = w, w / wr * hr
    return np.array([w_pad / 200.0, h_pad / 200.0], dtype=np.float32)


def crop_affine(center, scale, output_size, inv=False):
    """Closed form of get_affine_transform(center, scale, rot=0, output_size)
    (lib/utils/transforms.py:72-112): for rot=0 the three-point construction is
    a uniform scale about the centre.  Returns a float64 2x3 matrix."""
    center = np.asarray(center, dtype=np.float32).astype(np.float64)
    scale_tmp = (np.asarray(scale, dtype=np.float32) * np.float32(200.0)).astype(np.float64)
    dst_w, dst_h = float(output_size[0]), float(output_size[1])
    if scale_tmp[0] >= scale_tmp[1]:
        s = dst_w / scale_tmp[0]
    else:
        s = dst_h / scale_tmp[1]
    fwd = np.array([[s, 0.0, dst_w * 0.5 - s * center[0]],
                    [0.0, s, dst_h * 0.5 - s * center[1]]], dtype=np.float64)
    if not inv:
        return fwd
    return np.array([[1.0 / s, 0.0, center[0] - dst_w * 0.5 / s],
                     [0.0, 1.0 / s, center[1] - dst_h * 0.5 / s]], dtype=np.float64)


def ring_cameras(V, orig_wh, focal, radius, target, k, p, seed=0):
    """V pinhole cameras on a ring looking at ``target`` (mm).  Convention of the
    reference (lib/utils/cameras.py:188): x_cam = R (x - T), T = camera centre."""
    rs = np.random.RandomState(seed + 7919)
    cams = []
    w, h = orig_wh
    for v in range(V):
        ang = 2.0 * math.pi * (v + 0.13 * rs.rand()) / V
        hgt = 1000.0 + 1500.0 * rs.rand()
        rad = radius * (0.85 + 0.3 * rs.rand())
        C = np.array([target[0] + rad * math.cos(ang), target[1] + rad * math.sin(ang), hgt])
        fwd = np.asarray(target, dtype=np.float64) - C
        fwd /= np.linalg.norm(fwd)
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], 0)  # rows: camera x (right), y (down), z (forward)
        f = focal * (0.97 + 0.06 * rs.rand())
        cams.append(dict(
            R=R.astype(np.float32), T=C.reshape(3, 1).astype(np.float32),
            fx=np.float32(f), fy=np.float32(f * 1.002),
            cx=np.float32(w / 2.0 + 3.0 * rs.randn()), cy=np.float32(h / 2.0 + 3.0 * rs.randn()),
            k=np.asarray(k, dtype=np.float32).reshape(3, 1) * np.float32(1.0 + 0.05 * v / max(V, 1)),
            p=np.asarray(p, dtype=np.float32).reshape(2, 1),
        ))
    return cams


def make_meta(cams, B, orig_wh, img_wh, device="cpu"):
    """list[V] of per-view dicts with batch-collated tensors (the schema consumed
    by DQDecoderLayer: lib/models/dq_decoder.py:341-392,411-428)."""
    w, h = orig_wh
    center = np.array([w / 2.0, h / 2.0], dtype=np.float64)
    scale = get_scale((w, h), img_wh)
    inv = np.eye(3)
    inv[:2] = crop_affine(center, scale, img_wh, inv=True)
    fwd = np.eye(3)
    fwd[:2] = crop_affine(center, scale, img_wh, inv=False)
    meta = []
    for cam in cams:
        cd = {}
        for key in ("R", "T", "k", "p"):
            cd[key] = torch.from_numpy(np.stack([cam[key]] * B)).to(device)
        for key in ("fx", "fy", "cx", "cy"):
            cd[key] = torch.from_numpy(np.stack([np.asarray(cam[key], dtype=np.float32)] * B)).to(device)
        meta.append(dict(
            camera=cd,
            center=torch.from_numpy(np.stack([center] * B)).to(device),          # float64, as the loader
            scale=torch.from_numpy(np.stack([scale] * B)).to(device),            # float32
            affine_trans=torch.from_numpy(np.stack([fwd] * B)).to(device),       # float64 (3,3)
            inv_affine_trans=torch.from_numpy(np.stack([inv] * B)).to(device),   # float64 (3,3)
        ))
    return meta


def make_pyramid(V, B, shapes, C=256, seed=0, dtype=torch.float32, device="cpu", smooth_sigma0=None):
    """list of L tensors (V*B, C, H_l, W_l) ~ N(0,1), view-major.  smooth_sigma0: band-limit the maps -- white noise
    low-passed with a Gaussian of sigma = smooth_sigma0 cells at level 0 (halved per level: the same length in the image),
    rescaled to unit variance: feature maps with the spatial coherence of a backbone's, on which a sub-pixel shift of a
    sampling location changes the sampled value a little instead of completely."""
    rs = np.random.RandomState(seed + 104729)
    out = []
    for l, (H, W) in enumerate(shapes):
        a = rs.standard_normal((V * B, C, H, W)).astype(np.float32)
        if smooth_sigma0:
            sig = max(smooth_sigma0 / (2 ** l), 0.75)
            fy = np.fft.fftfreq(H)[:, None]
            fx = np.fft.rfftfreq(W)[None, :]
            gain = np.exp(-2.0 * (math.pi * sig) ** 2 * (fy ** 2 + fx ** 2)).astype(np.float32)     # periodic Gaussian blur
            for i in range(a.shape[0]):
                b = np.fft.irfft2(np.fft.rfft2(a[i], axes=(-2, -1)) * gain, s=(H, W), axes=(-2, -1))
                a[i] = (b / b.std()).astype(np.float32)
        out.append(torch.from_numpy(a).to(device=device, dtype=dtype))
    return out


def make_queries(B, NQ, J=15, C=256, seed=0, device="cpu"):
    rs = np.random.RandomState(seed + 1299709)
    tgt = torch.from_numpy(rs.standard_normal((B, NQ * J, C)).astype(np.float32)).to(device)
    pos = torch.from_numpy(rs.standard_normal((B, NQ * J, C)).astype(np.float32)).to(device)
    return tgt, pos


def init_reference_points(B, NQ, space_size, space_center, J=15, jitter=0.0, seed=0, device="cpu"):
    """'sample_space' initialisation (lib/models/dq_transformer.py:298-323):
    ceil(sqrt(NQ))^2 xy grid over the space at z-centre, first NQ cells, norm2absolute,
    plus the T-pose offsets.  ``jitter`` (mm) optionally perturbs the points."""
    N = int(math.ceil(math.sqrt(NQ)))
    xs = np.linspace(0.0, 1.0, N)
    gx, gy = np.meshgrid(xs, xs, indexing="ij")
    root = np.stack([gx, gy, np.full_like(gx, 0.5)], -1).reshape(-1, 3)[:NQ]
    size = np.asarray(space_size, dtype=np.float64)
    cen = np.asarray(space_center, dtype=np.float64)
    root_abs = root * size + cen - size / 2.0
    pts = root_abs[:, None, :] + TPOSE_MM[None]
    if jitter:
        pts = pts + np.random.RandomState(seed + 15485863).standard_normal(pts.shape) * jitter
    pts = np.broadcast_to(pts.reshape(1, NQ * J, 3), (B, NQ * J, 3)).astype(np.float32).copy()
    return torch.from_numpy(pts).to(device)


# ----------------------------------------------------------------------- layer weights
def layer_state_dict(seed, d_model=256, d_ffn=1024, n_heads=8, n_levels=1, n_points=8,
                     pose_embed_layer=3, valid_fraction=None, pose_scale=1.0):
    """Deterministic (numpy) weights for ONE DQDecoderLayer, keyed exactly like the
    reference state dict (probe of DQDecoderLayer.state_dict()).
    Scales are chosen so every term matters: sampling offsets of a few feature
    cells, non-uniform attention logits, 2D refinements of a few pixels.

    ``valid_fraction``: if given, the class head bias is set so that roughly that
    fraction of queries pass ``inference_conf_thr`` = 0.1 (the caller re-checks
    the margin to the threshold).
    """
    rs = np.random.RandomState(seed)
    C = d_model

    def lin(out_f, in_f, wscale=None, bscale=0.02):
        ws = (1.0 / math.sqrt(in_f)) if wscale is None else wscale
        return (rs.standard_normal((out_f, in_f)) * ws).astype(np.float32), \
               (rs.standard_normal((out_f,)) * bscale).astype(np.float32)

    sd = {}
    n_off = n_heads * n_levels * n_points * 2
    n_att = n_heads * n_levels * n_points
    w, _ = lin(n_off, C, wscale=0.02)
    # bias: the reference's directional grid init (projattn.py:98-108) -- same construction
    thetas = np.arange(n_heads, dtype=np.float32) * (2.0 * math.pi / n_heads)
    grid = np.stack([np.cos(thetas), np.sin(thetas)], -1)
    grid = grid / np.abs(grid).max(-1, keepdims=True)
    grid = np.tile(grid.reshape(n_heads, 1, 1, 2), (1, n_levels, n_points, 1))
    for i in range(n_points):
        grid[:, :, i, :] *= i + 1
    sd["proj_attn.sampling_offsets.weight"] = w
    sd["proj_attn.sampling_offsets.bias"] = grid.reshape(-1).astype(np.float32)
    sd["proj_attn.attention_weights.weight"], sd["proj_attn.attention_weights.bias"] = lin(n_att, C, wscale=0.03, bscale=0.3)
    sd["proj_attn.rayconv.weight"], sd["proj_attn.rayconv.bias"] = lin(C, C)
    sd["proj_attn.output_proj.weight"], sd["proj_attn.output_proj.bias"] = lin(C, C)
    for nm in ("norm1", "norm2", "norm3"):
        sd[nm + ".weight"] = (1.0 + 0.1 * rs.standard_normal(C)).astype(np.float32)
        sd[nm + ".bias"] = (0.05 * rs.standard_normal(C)).astype(np.float32)
    sd["self_attn.in_proj_weight"] = (rs.standard_normal((3 * C, C)) / math.sqrt(C)).astype(np.float32)
    sd["self_attn.in_proj_bias"] = np.zeros(3 * C, np.float32)
    sd["self_attn.out_proj.weight"], sd["self_attn.out_proj.bias"] = lin(C, C)
    sd["feature_update_mlp.weight"], sd["feature_update_mlp.bias"] = lin(C, C)
    sd["linear1.weight"], sd["linear1.bias"] = lin(d_ffn, C)
    sd["linear2.weight"], sd["linear2.bias"] = lin(C, d_ffn)
    dims = [C] + [C] * (pose_embed_layer - 1) + [3]
    for i in range(pose_embed_layer):
        last = i == pose_embed_layer - 1
        # last layer: pixel offsets of a few px, view-confidence logits O(1)
        w, b = lin(dims[i + 1], dims[i], wscale=(2.0 / math.sqrt(dims[i])) if last else None)
        if last and pose_scale != 1.0:
            w[:2] *= pose_scale          # the two pixel-offset rows (the view-confidence logit row keeps its scale)
        sd["pose_embed.MLP.layers.%d.weight" % i] = w
        sd["pose_embed.MLP.layers.%d.bias" % i] = b
    w, b = lin(2, C, wscale=4.0 / math.sqrt(C), bscale=0.0)
    if valid_fraction is not None:
        # prob = mean_j sigmoid(b + 4 z_j); E[sigmoid(b + s z)] ~ sigmoid(b / sqrt(1 + pi s^2 / 8)), so
        # b = logit(0.1) * sqrt(1 + 2 pi) centres the per-query probability on the 0.1 threshold;
        # the per-query spread (~0.05) then lets roughly ``valid_fraction`` of the queries pass.
        from scipy.stats import norm as _norm
        b[1] = -2.1972246 * math.sqrt(1.0 + 2.0 * math.pi) + 1.5 * _norm.ppf(valid_fraction)
    else:
        b[1] = 4.0  # everything valid
    sd["class_embed.weight"], sd["class_embed.bias"] = w, b.astype(np.float32)
    return sd


def decoder_state_dict(seed, num_layers, **kw):
    """``layers.{i}.<key>`` for an unshared-weights decoder (mvp_decoder.py:272-275)."""
    out = {}
    for i in range(num_layers):
        for k_, v_ in layer_state_dict(seed * 1000 + i, **kw).items():
            out["layers.%d.%s" % (i, k_)] = v_
    return out


def to_torch_state(sd, device="cpu", dtype=torch.float32):
    return {k_: torch.from_numpy(np.ascontiguousarray(v_)).to(device=device, dtype=dtype) for k_, v_ in sd.items()}


def decoder_cfg(space_size, space_center, share_layer_weights=False):
    """Minimal cfg namespace that DQDecoder reads (mvp_decoder.py:268-282)."""
    return SimpleNamespace(
        DECODER=SimpleNamespace(share_layer_weights=share_layer_weights),
        MULTI_PERSON=SimpleNamespace(SPACE_SIZE=list(space_size), SPACE_CENTER=list(space_center)),
    )


def build_case(name, B=1, seed=0, NQ=None, layers=None, V=None, feat_dtype=torch.float32,
               device="cpu", jitter=25.0, valid_fraction=None, with_features=True, ref_extent=1.0, smooth_sigma0=None,
               pose_scale=1.0):
    """Assemble all decoder inputs for a named configuration.  ref_extent: fraction of the space's x / y extent the initial
    query grid covers (1.0 = the model's own 'sample_space' initialisation over the whole space, where ~40 % of the (view,
    query) pairs of cfg-2 project outside their image; 0.3 puts > 99 % of them inside every view -- the regime of a trained
    model's later layers, whose queries sit on the people)."""
    c = dict(CONFIGS[name])
    if NQ is not None:
        c["NQ"] = NQ
    if layers is not None:
        c["layers"] = layers
    if V is not None:
        c["V"] = V
    shapes = pyramid_shapes(c["img_wh"])
    cams = ring_cameras(c["V"], c["orig_wh"], c["focal"], c["radius"], c["space_center"], c["k"], c["p"], seed)
    meta = make_meta(cams, B, c["orig_wh"], c["img_wh"], device)
    tgt, pos = make_queries(B, c["NQ"], seed=seed, device=device)
    grid_size = (c["space_size"][0] * ref_extent, c["space_size"][1] * ref_extent, c["space_size"][2])
    ref = init_reference_points(B, c["NQ"], grid_size, c["space_center"], jitter=jitter, seed=seed, device=device)
    src = make_pyramid(c["V"], B, shapes, seed=seed, dtype=feat_dtype, device=device,
                       smooth_sigma0=smooth_sigma0) if with_features else None
    spatial_shapes = torch.tensor(shapes, dtype=torch.long, device=device)
    level_start = torch.cat([spatial_shapes.new_zeros(1), (spatial_shapes[:, 0] * spatial_shapes[:, 1]).cumsum(0)[:-1]])
    weights = decoder_state_dict(seed + 1, c["layers"], valid_fraction=valid_fraction, pose_scale=pose_scale)
    return SimpleNamespace(cfg=c, name=name, B=B, V=c["V"], NQ=c["NQ"], J=15, layers=c["layers"],
                           img_size=list(c["img_wh"]), shapes=shapes, meta=meta, tgt=tgt, query_pos=pos,
                           reference_points=ref, src_views=src, spatial_shapes=spatial_shapes,
                           level_start_index=level_start, weights=weights,
                           space_size=c["space_size"], space_center=c["space_center"])
