"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the MVGFormer decoder hot path.

This file restates, with plain torch CPU tensor arithmetic, the algorithm of the
reference's decoder path (project -> sample -> attend -> triangulate).  It is the
checker for the HIP kernels: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``mvgformer_amd``) never imports it and fails loudly when the HIP library is absent.

Pinning: the reference ships NO tests for this path, so the
oracle is pinned against outputs of the reference itself, generated in the build
container by ``tests/golden/make_golden.py`` (which imports /root/reference with
stubs) and committed under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors.

Every function cites the reference file:line it follows (paths relative to the
reference root).  ``dtype`` selects float32 (the reference's arithmetic) or float64
(a high-precision run used to measure how much fp32 rounding the reference itself
carries, e.g. in the DLT/SVD step).

Notation: B batch, V views, Lq = NQ*J joint tokens, C=256 channels, M heads,
D=C/M, L feature levels, P points, S = sum_l H_l*W_l.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- sampling
def msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    """Multi-scale deformable sampling forward.

    Restates the CUDA forward kernel lib/models/ops/src/cuda/deform_im2col_cuda.cuh:248-309
    (bilinear helper :44-94); its CPU twin is deform_core_pytorch,
    lib/models/ops/functions/deform_func.py:68-99.

    value (N,S,M,D); spatial_shapes (L,2) int64 (H,W); level_start_index (L,);
    sampling_loc (N,Lq,M,L,P,2) as (x,y) in [0,1]; attn_weight (N,Lq,M,L,P).
    Returns (N, Lq, M*D).
    """
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    out = value.new_zeros((N, Lq, M, D))
    bidx = torch.arange(N).view(N, 1, 1, 1)
    midx = torch.arange(M).view(1, 1, M, 1)
    for l in range(L):
        H = int(spatial_shapes[l, 0])
        W = int(spatial_shapes[l, 1])
        start = int(level_start_index[l])
        loc = sampling_loc[:, :, :, l]                     # (N,Lq,M,P,2)
        wgt = attn_weight[:, :, :, l]                      # (N,Lq,M,P)
        h_im = loc[..., 1] * H - 0.5                        # cuh:295
        w_im = loc[..., 0] * W - 0.5                        # cuh:296
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)   # cuh:298
        h_low = torch.floor(h_im)
        w_low = torch.floor(w_im)
        lh = h_im - h_low
        lw = w_im - w_low
        hh = 1 - lh
        hw = 1 - lw
        h_low = h_low.long()
        w_low = w_low.long()
        h_high = h_low + 1
        w_high = w_low + 1
        acc = None
        for (hi, wi, cw) in ((h_low, w_low, hh * hw), (h_low, w_high, hh * lw),
                             (h_high, w_low, lh * hw), (h_high, w_high, lh * lw)):
            ok = (hi >= 0) & (hi <= H - 1) & (wi >= 0) & (wi <= W - 1)      # cuh:66-88
            idx = start + hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)      # (N,Lq,M,P)
            v = value[bidx, idx, midx]                                      # (N,Lq,M,P,D)
            term = (cw * ok.to(value.dtype)).unsqueeze(-1) * v
            acc = term if acc is None else acc + term
        contrib = acc * (wgt * inside.to(value.dtype)).unsqueeze(-1)
        out = out + contrib.sum(3)
    return out.reshape(N, Lq, M * D)


def bilinear_zeros(src, grid):
    """F.grid_sample(src, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    restated with explicit gathers (used at lib/models/ops/modules/projattn.py:148-153).
    src (B,C,H,W); grid (B,Lq,2) in [-1,1] as (x,y).  Returns (B,Lq,C)."""
    B, C, H, W = src.shape
    ix = ((grid[..., 0] + 1) * W - 1) / 2
    iy = ((grid[..., 1] + 1) * H - 1) / 2
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    tx = ix - x0
    ty = iy - y0
    x0 = x0.long()
    y0 = y0.long()
    flat = src.flatten(2)                                   # (B,C,H*W)
    out = src.new_zeros((B, grid.shape[1], C))
    for (yy, xx, cw) in ((y0, x0, (1 - tx) * (1 - ty)), (y0, x0 + 1, tx * (1 - ty)),
                         (y0 + 1, x0, (1 - tx) * ty), (y0 + 1, x0 + 1, tx * ty)):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        idx = yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)   # (B,Lq)
        v = torch.gather(flat, 2, idx.unsqueeze(1).expand(-1, C, -1)).transpose(1, 2)
        out = out + (cw * ok.to(src.dtype)).unsqueeze(-1) * v
    return out


# ------------------------------------------------------------------------- projection
def crop_affine_matrix(center, scale, img_size, dtype):
    """get_affine_transform(center, scale, 0, img_size) for rot=0
    (lib/utils/transforms.py:72-112) in closed form: uniform scale about the centre.
    center (B,2), scale (B,2) -> (B,2,3) float32-rounded like dq_decoder.py:361-372."""
    c = center.to(torch.float32).to(torch.float64)
    st = (scale.to(torch.float32) * 200.0).to(torch.float64)
    dw, dh = float(img_size[0]), float(img_size[1])
    s = torch.where(st[:, 0] >= st[:, 1], dw / st[:, 0], dh / st[:, 1])
    A = torch.zeros((c.shape[0], 2, 3), dtype=torch.float64)
    A[:, 0, 0] = s
    A[:, 1, 1] = s
    A[:, 0, 2] = dw * 0.5 - s * c[:, 0]
    A[:, 1, 2] = dh * 0.5 - s * c[:, 1]
    return A.to(torch.float32).to(dtype)


def project_ref_points(X, cam, center, scale, img_size, dtype=torch.float32):
    """A.1: DQDecoderLayer.project_ref_points (lib/models/dq_decoder.py:331-397) with
    cameras.project_pose_batch / project_point_radial_batch (lib/utils/cameras.py:167-217).

    X (B,Lq,3) mm; cam dict of (B,...) tensors.  Returns r (B,Lq,2) normalised network
    image coordinates and inside (B,Lq) bool."""
    X = X.to(dtype)
    R = cam["R"].to(dtype)
    T = cam["T"].to(dtype).reshape(-1, 3, 1)
    xc = torch.matmul(R, X.transpose(1, 2) - T)                 # (B,3,Lq)      cameras.py:188
    y = xc[:, :2] / (xc[:, 2:] + 1e-5)                          #               cameras.py:190
    k = cam["k"].to(dtype).reshape(-1, 3, 1)
    p = cam["p"].to(dtype).reshape(-1, 2, 1)
    r2 = (y ** 2).sum(1, keepdim=True)                          # (B,1,Lq)
    radial = 1 + (k[:, 0:1] * r2 + k[:, 1:2] * r2 ** 2 + k[:, 2:3] * r2 ** 3)   # cameras.py:195-198
    tan = p[:, 0:1] * y[:, 1:2] + p[:, 1:2] * y[:, 0:1]         # cameras.py:200
    y = y * (radial + 2 * tan) + torch.cat([p[:, 1:2], p[:, 0:1]], 1) * r2      # cameras.py:201-204
    f = torch.stack([cam["fx"], cam["fy"]], 1).to(dtype).reshape(-1, 2, 1)
    c = torch.stack([cam["cx"], cam["cy"]], 1).to(dtype).reshape(-1, 2, 1)
    u = (f * y + c).transpose(1, 2)                             # (B,Lq,2)      cameras.py:206
    wh = center.unsqueeze(1) * 2                                # (B,1,2)       dq_decoder.py:374
    inside = (u[..., 0] >= 0) & (u[..., 1] >= 0) & (u[..., 0] < wh[..., 0]) & (u[..., 1] < wh[..., 1])
    u = torch.clamp(u, -1.0, float(wh.max()))                   # dq_decoder.py:382-383
    A = crop_affine_matrix(center, scale, img_size, dtype)      # (B,2,3)
    n = torch.matmul(torch.cat([u, torch.ones_like(u[..., :1])], -1), A.transpose(1, 2))  # transforms.py:135-141
    r = n / torch.tensor(img_size, dtype=dtype)                 # dq_decoder.py:390-392
    return r, inside


# --------------------------------------------------------------------------- ProjAttn
def proj_attn_forward(prm, prefix, query, ref_lvl, src_views, spatial_shapes, level_start_index,
                      n_heads=8, n_points=8, return_intermediates=False):
    """A.3: ProjAttn.forward in mode 'ablation_not_use_rayconv'
    (lib/models/ops/modules/projattn.py:115-204).

    query (B,Lq,C); ref_lvl (B,Lq,L,2); src_views: L tensors (B,C,H_l,W_l).
    NOTE the memory reinterpretation of the Linear outputs (projattn.py:180-184):
    the module is built with n_levels=1 but applied to L levels, so the (L, 128) and
    (L, 64) Linear outputs are *viewed* as (M, L, P, 2) / (M, L*P)."""
    B, Lq, C = query.shape
    L = len(src_views)
    M, P = n_heads, n_points
    Wv, bv = prm[prefix + "rayconv.weight"], prm[prefix + "rayconv.bias"]
    Wo, bo = prm[prefix + "sampling_offsets.weight"], prm[prefix + "sampling_offsets.bias"]
    Wa, ba = prm[prefix + "attention_weights.weight"], prm[prefix + "attention_weights.bias"]
    Wp, bp = prm[prefix + "output_proj.weight"], prm[prefix + "output_proj.bias"]

    grid = torch.clamp(ref_lvl * 2.0 - 1.0, -1.1, 1.1)                        # projattn.py:134
    feats = torch.stack([bilinear_zeros(src_views[l], grid[:, :, l]) for l in range(L)], 2)  # (B,Lq,L,C)
    flat = torch.cat([s.flatten(2) for s in src_views], -1).transpose(1, 2)   # (B,S,C)  projattn.py:160
    value = (flat @ Wv.t() + bv).view(B, -1, M, C // M)                       # projattn.py:169,175
    x = feats + query.unsqueeze(2)
    off = (x @ Wo.t() + bo).reshape(B, Lq, M, L, P, 2)                        # projattn.py:180 (reinterpretation)
    aw = (x @ Wa.t() + ba).reshape(B, Lq, M, L * P)                           # projattn.py:181
    aw = torch.softmax(aw, -1).view(B, Lq, M, L, P)                           # projattn.py:184
    norm = torch.stack([spatial_shapes[:, 1], spatial_shapes[:, 0]], -1).to(query.dtype)  # (L,2) = (W,H)
    loc = ref_lvl[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]      # projattn.py:186-191
    samp = msda_forward(value, spatial_shapes, level_start_index, loc, aw)    # projattn.py:200
    out = samp @ Wp.t() + bp                                                  # projattn.py:203
    if return_intermediates:
        return out, dict(ref_feats=feats, value=value, offsets=off, weights=aw, locations=loc, sampled=samp)
    return out


# ------------------------------------------------------------------------ triangulation
def undistort_points(uo, cam, dtype, iters=5):
    """undistort (lib/models/dq_decoder.py:119-204): K^-1, 5 fixed-point iterations with the
    OpenCV-ordered coefficient vector [k1,k2,p1,p2,k3,0*7], re-apply K.
    uo (n,V,J,2) original-image px; cam tensors (n,V,...)."""
    fx, fy, cx, cy = (cam[k_].to(dtype)[..., None] for k_ in ("fx", "fy", "cx", "cy"))   # (n,V,1)
    k = cam["k"].to(dtype).reshape(*cam["k"].shape[:2], 3)
    p = cam["p"].to(dtype).reshape(*cam["p"].shape[:2], 2)
    k1, k2, k3 = (k[..., i:i + 1] for i in range(3))
    p1, p2 = p[..., 0:1], p[..., 1:2]
    x0 = uo[..., 0] * (1 / fx) + (-cx / fx)            # K^-1 [u,v,1]  (dq_decoder.py:171-176)
    y0 = uo[..., 1] * (1 / fy) + (-cy / fy)
    x, y = x0, y0
    for _ in range(iters):
        r2 = x * x + y * y
        icd = 1 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)                                      # dq_decoder.py:188 (k[5..7]=0)
        dX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)                                          # :190 (k[8..11]=0)
        dY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y                                          # :192
        x = (x0 - dX) * icd
        y = (y0 - dY) * icd
    return torch.stack([fx * x + cx, fy * y + cy], -1)                                       # :198-203


def projection_matrices(cam, dtype):
    """get_proj_matricies_batch(inv_trans=True) (lib/models/dq_decoder.py:223-246):
    P = K [R | -R T].  cam tensors (n,V,...) -> (n,V,3,4)."""
    R = cam["R"].to(dtype)
    T = cam["T"].to(dtype).reshape(*R.shape[:2], 3, 1)
    K = torch.zeros(R.shape[:2] + (3, 3), dtype=dtype)
    K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2] = (cam[k_].to(dtype) for k_ in ("fx", "fy", "cx", "cy"))
    K[..., 2, 2] = 1
    RT = torch.cat([R, -R @ T], -1)
    return K @ RT


def dlt_triangulate(Pm, pts, conf):
    """triangulate_point_from_multiple_views_linear_torch_batch, solver='linalg'
    (lib/mvn/utils/multiview.py:170-228): rows conf*(x*P[2]-P[0]), conf*(y*P[2]-P[1]);
    X = -V[:,3] of the SVD; X[:3]/X[3].
    Pm (n,V,3,4); pts (n,V,J,2); conf (n,V,J) -> (n,J,3)."""
    n, V, J, _ = pts.shape
    pt = pts.permute(0, 2, 1, 3)                                   # (n,J,V,2)
    A = Pm[:, None, :, 2:3, :] * pt[..., None]                     # (n,J,V,2,4)
    A = A - Pm[:, None, :, :2, :]
    A = A * conf.permute(0, 2, 1)[..., None, None]
    A = A.reshape(n, J, 2 * V, 4)
    _, _, Vh = torch.linalg.svd(A)                                 # multiview.py:210
    Xh = -Vh[..., 3, :]
    return Xh[..., :3] / Xh[..., 3:4], A


# -------------------------------------------------------------------------- the layer
def _ln(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _stack_cam(meta, dtype):
    """per-view camera dicts -> dict of (B,V,...) tensors (dq_decoder.py:226-233)."""
    keys = ("R", "T", "fx", "fy", "cx", "cy", "k", "p")
    return {k_: torch.stack([m["camera"][k_] for m in meta], 1) for k_ in keys}


def decoder_layer_forward(prm, prefix, tgt, query_pos, reference_points, src_views, spatial_shapes,
                          level_start_index, meta, img_size, threshold=0.5, num_joints=15,
                          n_heads=8, n_points=8, dtype=torch.float32, indices=None, extras=False):
    """A.1-A.8: DQDecoderLayer.forward (lib/models/dq_decoder.py:850-1045) for the shipped
    configuration (feature_update_method='MLP', init_self_attention=False,
    open_forward_ffn=True, query_filter_method='threshold', filter_query=True,
    triangulation_method='linalg', bayesian_update=False).

    All queries are computed densely and masked at the end: every step of the path is
    per-query, so this equals the reference's gather/pad/scatter
    (dq_decoder.py:615-656,929-967,1013-1029).

    reference_points (B,Lq,3) mm; src_views L tensors (V*B,C,H,W) view-major.
    Returns (tgt_update, new_reference_points, refined_2d_abs, projs_2d_abs, class_prob)."""
    P_ = lambda n: prm[prefix + n].to(dtype)
    B, Lq, C = tgt.shape
    L = len(src_views)
    V = src_views[0].shape[0] // B
    J = num_joints
    NQ = Lq // J
    tgt = tgt.to(dtype)
    query_pos = query_pos.to(dtype)
    img = torch.tensor(img_size, dtype=dtype)
    prm_t = {k_: v_.to(dtype) for k_, v_ in prm.items() if k_.startswith(prefix + "proj_attn.")}

    WH = spatial_shapes.flip(-1).to(dtype)                                   # (L,2) (W,H)
    attn_views, r_views = [], []
    for v in range(V):                                                        # dq_decoder.py:553
        src_v = [s[v * B:(v + 1) * B].to(dtype) for s in src_views]           # :560
        r, inside = project_ref_points(reference_points, meta[v]["camera"], meta[v]["center"],
                                       meta[v]["scale"], img_size, dtype)     # :563
        ref_lvl = r.unsqueeze(2) * WH / (WH - 1)                              # :570-573
        a = proj_attn_forward(prm_t, prefix + "proj_attn.", tgt + query_pos, ref_lvl, src_v,
                              spatial_shapes, level_start_index, n_heads, n_points)   # :579
        attn_views.append(inside.unsqueeze(-1).to(dtype) * a)                 # :585-586
        r_views.append(r)

    mean = torch.stack(attn_views, 0).mean(0)                                 # :770
    t1 = _ln(tgt + (mean @ P_("feature_update_mlp.weight").t() + P_("feature_update_mlp.bias")),
             P_("norm2.weight"), P_("norm2.bias"))                            # :773-778
    h = torch.relu(t1 @ P_("linear1.weight").t() + P_("linear1.bias"))        # mvp_decoder.py:94-98
    tgt_update = _ln(t1 + (h @ P_("linear2.weight").t() + P_("linear2.bias")), P_("norm3.weight"), P_("norm3.bias"))

    logits = tgt_update @ P_("class_embed.weight").t() + P_("class_embed.bias")          # :889
    prob = torch.sigmoid(logits.view(B, NQ, J, 2)).mean(2)                    # :890-893
    if indices is not None:                                                   # training: GT-matched ids (:900-901)
        valid = torch.zeros((B, NQ), dtype=torch.bool)
        for b, q in enumerate(indices):
            valid[b, torch.as_tensor(q, dtype=torch.long)] = True
    else:
        valid = prob[..., 1] > threshold                                      # :605
    if not bool(valid.any()):
        valid[0, 0] = True                                                    # :620-623

    ref2d, proj2d, logit = [], [], []
    for v in range(V):                                                        # :673-690
        hcur = attn_views[v]
        hcur = torch.relu(hcur @ P_("pose_embed.MLP.layers.0.weight").t() + P_("pose_embed.MLP.layers.0.bias"))
        hcur = torch.relu(hcur @ P_("pose_embed.MLP.layers.1.weight").t() + P_("pose_embed.MLP.layers.1.bias"))
        o = hcur @ P_("pose_embed.MLP.layers.2.weight").t() + P_("pose_embed.MLP.layers.2.bias")
        ref2d.append((r_views[v] + o[..., :2] / img) * img)                   # :679-685,696
        proj2d.append(r_views[v] * img)                                       # :699
        logit.append(o[..., 2])
    ref2d = torch.stack(ref2d, 1)                                             # (B,V,Lq,2) network-image px
    proj2d = torch.stack(proj2d, 1)
    conf = torch.softmax(torch.stack(logit, 1), 1)                            # softmax over views  :706-707

    cam = _stack_cam(meta, dtype)
    cam_q = {k_: v_.repeat_interleave(NQ, 0) for k_, v_ in cam.items()}       # one copy per query (:953-967)
    kp = ref2d.view(B, V, NQ, J, 2).permute(0, 2, 1, 3, 4).reshape(B * NQ, V, J, 2)
    cf = conf.view(B, V, NQ, J).permute(0, 2, 1, 3).reshape(B * NQ, V, J)
    Ainv = torch.stack([m["inv_affine_trans"][:, :2, :] for m in meta], 1).float().to(dtype)   # (B,V,2,3)  :414-419
    Ainv = Ainv.repeat_interleave(NQ, 0)
    uo = torch.matmul(torch.cat([kp, torch.ones_like(kp[..., :1])], -1), Ainv.transpose(2, 3))  # :420
    ud = undistort_points(uo, cam_q, dtype)                                   # :422
    Pm = projection_matrices(cam_q, dtype)                                    # :428
    X3, Amat = dlt_triangulate(Pm, ud, cf)                                    # :457
    X3 = X3.view(B, NQ, J, 3)

    vm = valid.view(B, NQ, 1, 1)
    new_ref = torch.where(vm, X3, torch.zeros_like(X3)).reshape(B, Lq, 3)     # :1013-1029
    vm2 = valid.view(B, 1, NQ, 1, 1)
    ref2d_o = torch.where(vm2, ref2d.view(B, V, NQ, J, 2), torch.zeros(())).reshape(B, V, Lq, 2).to(dtype)
    proj2d_o = torch.where(vm2, proj2d.view(B, V, NQ, J, 2), torch.zeros(())).reshape(B, V, Lq, 2).to(dtype)
    out = (tgt_update, new_ref, ref2d_o, proj2d_o, prob)
    if extras:
        return out, dict(valid=valid, attn_views=attn_views, r_views=r_views, conf=conf, undist=ud,
                         proj_mats=Pm, dlt_rows=Amat, dense_points=X3, ref2d_dense=ref2d)
    return out


def decoder_forward(prm, num_layers, tgt, reference_points, src_views, meta, spatial_shapes,
                    level_start_index, query_pos, img_size, threshold=0.5, dtype=torch.float32,
                    share_layer_weights=False, **kw):
    """A: DQDecoder.forward with return_intermediate=True (lib/models/dq_decoder.py:1107-1172).
    Returns (hs (layers,B,Lq,C), refs (layers,B,Lq,3), refs2d (layers,B,V,Lq,2),
    projs2d (layers,B,V,Lq,2), [class_prob (B,NQ,2)] * layers)."""
    out, ref = tgt, reference_points
    hs, refs, r2d, p2d, cls = [], [], [], [], []
    for lid in range(num_layers):
        prefix = "layers.%d." % (0 if share_layer_weights else lid)
        out, ref, a, b, c = decoder_layer_forward(prm, prefix, out, query_pos, ref, src_views, spatial_shapes,
                                                  level_start_index, meta, img_size, threshold, dtype=dtype, **kw)
        hs.append(out); refs.append(ref); r2d.append(a); p2d.append(b); cls.append(c)
    return torch.stack(hs), torch.stack(refs), torch.stack(r2d), torch.stack(p2d), cls
