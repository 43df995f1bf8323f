"""Differentiable torch formulation of the decoder's camera geometry, used ONLY by the training
(autograd) path of DQDecoderLayer; the inference path runs the HIP kernels of csrc/geom.hip.

  project_points   lib/models/dq_decoder.py:331-397 + lib/utils/cameras.py:167-217
  undistort        lib/models/dq_decoder.py:119-204 (5 fixed-point iterations)
  proj_matrices    lib/models/dq_decoder.py:223-246 (P = K [R | -R T])
  dlt              lib/mvn/utils/multiview.py:170-228 (torch.linalg.svd, differentiable)
All tensors live on the caller's device; nothing here synchronises with the host except the tiny
crop-affine construction from (center, scale), done once per forward.
"""
from __future__ import annotations

import torch


def crop_affine(center, scale, img_size, device):
    """closed form of get_affine_transform(center, scale, 0, img_size) (lib/utils/transforms.py:72-112) -> (B,2,3)."""
    c = center.detach().to("cpu", torch.float32).double()
    st = (scale.detach().to("cpu", torch.float32) * 200.0).double()
    dw, dh = float(img_size[0]), float(img_size[1])
    s = torch.where(st[:, 0] >= st[:, 1], dw / st[:, 0], dh / st[:, 1])
    A = torch.zeros((c.shape[0], 2, 3), dtype=torch.float64)
    A[:, 0, 0] = s
    A[:, 1, 1] = s
    A[:, 0, 2] = dw * 0.5 - s * c[:, 0]
    A[:, 1, 2] = dh * 0.5 - s * c[:, 1]
    return A.float().to(device)


def project_points(X, cam, center, A_crop, img_size, views=1):
    """X (B,Lq,3) mm -> r (B,Lq,2) normalised network-image coords, inside (B,Lq) bool.  views > 1: the batch is `views`
    stacked views (image v*B' + b); the clamp bound below is then taken per view, as a per-view call would."""
    f32 = torch.float32
    R = cam["R"].to(f32)
    T = cam["T"].to(f32).reshape(-1, 3, 1)
    xc = torch.matmul(R, X.transpose(1, 2) - T)
    y = xc[:, :2] / (xc[:, 2:] + 1e-5)
    k = cam["k"].to(f32).reshape(-1, 3, 1)
    p = cam["p"].to(f32).reshape(-1, 2, 1)
    r2 = (y ** 2).sum(1, keepdim=True)
    radial = 1 + (k[:, 0:1] * r2 + k[:, 1:2] * r2 ** 2 + k[:, 2:3] * r2 ** 3)
    tan = p[:, 0:1] * y[:, 1:2] + p[:, 1:2] * y[:, 0:1]
    y = y * (radial + 2 * tan) + torch.cat([p[:, 1:2], p[:, 0:1]], 1) * r2
    f = torch.stack([cam["fx"], cam["fy"]], 1).to(f32).reshape(-1, 2, 1)
    c = torch.stack([cam["cx"], cam["cy"]], 1).to(f32).reshape(-1, 2, 1)
    u = (f * y + c).transpose(1, 2)
    wh = center.to(u.device).unsqueeze(1) * 2
    inside = (u[..., 0] >= 0) & (u[..., 1] >= 0) & (u[..., 0] < wh[..., 0]) & (u[..., 1] < wh[..., 1])
    bound = wh.reshape(views, -1).amax(1).repeat_interleave(wh.shape[0] // views).view(-1, 1, 1)   # dq_decoder.py:382-383
    u = torch.minimum(torch.clamp(u, min=-1.0), bound.to(u.dtype))
    n = torch.matmul(torch.cat([u, torch.ones_like(u[..., :1])], -1), A_crop.transpose(1, 2))
    return n / torch.tensor(img_size, dtype=f32, device=u.device), inside


def undistort(uo, cam, iters=5):
    """uo (B,V,N,2) original-image px; cam tensors (B,V,...)."""
    fx, fy, cx, cy = (cam[k_].float()[..., None] for k_ in ("fx", "fy", "cx", "cy"))
    k = cam["k"].float().reshape(*cam["k"].shape[:2], 3)
    p = cam["p"].float().reshape(*cam["p"].shape[:2], 2)
    k1, k2, k3 = (k[..., i:i + 1] for i in range(3))
    p1, p2 = p[..., 0:1], p[..., 1:2]
    x0 = uo[..., 0] * (1 / fx) + (-cx / fx)
    y0 = uo[..., 1] * (1 / fy) + (-cy / fy)
    x, y = x0, y0
    for _ in range(iters):
        r2 = x * x + y * y
        icd = 1 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dX) * icd
        y = (y0 - dY) * icd
    return torch.stack([fx * x + cx, fy * y + cy], -1)


class UncropUndistort(torch.autograd.Function):
    """ref2d (B,V,Lq,2) network-image px -> undistorted original-image px: the inverse crop affine + undistort() above as one HIP
    launch that also returns every point's 2 x 2 Jacobian (csrc/geom.hip: uncrop_undistort_jac_kernel); backward = J^T g."""

    @staticmethod
    def forward(ctx, ref2d, cams, V, B):
        from . import ops
        ud, jac = ops.uncrop_undistort_jac(ref2d.detach(), cams, V, B)
        ctx.save_for_backward(jac)
        return ud

    @staticmethod
    def backward(ctx, g):
        (jac,) = ctx.saved_tensors
        # grad_ref[j] = sum_i jac[i][j] g[i]  (elementwise: a batched 2 x 2 matmul went to a 625-us rocBLAS kernel)
        return (jac * g.unsqueeze(-1)).sum(-2), None, None, None


class DenseDLT(torch.autograd.Function):
    """new reference points (B, Lq, 3) of all tokens from ud (B,V,Lq,2), conf (B,V,Lq), Pm (B,V,3,4): dlt() below for the tokens of the
    queries flagged in valid (B, NQ) uint8, zeros elsewhere -- one HIP launch forward, one backward (csrc/geom.hip: dlt_fwd_kernel /
    dlt_bwd_kernel) instead of nonzero + index + ~20 torch launches forward and ~60 backward per decoder layer."""

    @staticmethod
    def forward(ctx, ud, conf, Pm, valid, J):
        from . import ops
        ud, conf = ud.detach().contiguous(), conf.detach().contiguous()
        ctx.save_for_backward(ud, conf, Pm, valid)
        ctx.J = J
        return ops.dlt_forward(ud, conf, Pm, valid, J)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        ud, conf, Pm, valid = ctx.saved_tensors
        g_ud, g_conf = ops.dlt_backward(ud, conf, Pm, valid, ctx.J, g.contiguous().float())
        return g_ud, g_conf, None, None, None


def proj_matrices_from_records(cams, V, B):
    """(B, V, 3, 4) projection matrices K [R | -R T] from the packed camera records (ops.pack_cameras: image n = v * B + b; R at 0:9,
    T at 9:12, fx fy cx cy at 12:16) -- the same fp32 products as proj_matrices, without touching the per-view meta dicts again."""
    rec = cams.view(V, B, -1).transpose(0, 1)
    R = rec[..., 0:9].reshape(B, V, 3, 3)
    T = rec[..., 9:12].reshape(B, V, 3, 1)
    K = torch.zeros((B, V, 3, 3), dtype=torch.float32, device=cams.device)
    K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2] = rec[..., 12], rec[..., 13], rec[..., 14], rec[..., 15]
    K[..., 2, 2] = 1
    return K @ torch.cat([R, -R @ T], -1)


def dlt(Pm, pts, conf):
    """Pm (B,V,3,4); pts (B,V,N,2); conf (B,V,N) -> (B,N,3) via the smallest right singular vector."""
    pt = pts.permute(0, 2, 1, 3)                                     # (B,N,V,2)
    A = Pm[:, None, :, 2:3, :] * pt[..., None] - Pm[:, None, :, :2, :]
    A = A * conf.permute(0, 2, 1)[..., None, None]
    A = A.reshape(A.shape[0], A.shape[1], -1, 4)
    if A.is_cuda:
        # smallest right singular vector of A = eigenvector of the smallest eigenvalue of A^T A (Gram matrix in fp64,
        # like the inference kernel): mvg_sym4_eigh + an analytic backward instead of rocSOLVER's batched SVD, which
        # took 70 % of a training step at cfg-2 (223 of 333 ms)
        Ad = A.double()
        Xh = SmallestEigvec4.apply(Ad.transpose(-1, -2) @ Ad).to(A.dtype)
    else:
        # host tensors: plain torch reference of the same quantity (used by the gradient-parity test as the thing to
        # compare with; DQDecoderLayer never gets here -- it raises "Not implemented on the CPU" like the reference op)
        _, _, Vh = torch.linalg.svd(A)
        Xh = -Vh[..., 3, :]
    return Xh[..., :3] / Xh[..., 3:4]


class SmallestEigvec4(torch.autograd.Function):
    """v0(G): unit eigenvector of the smallest eigenvalue of a symmetric 4x4 matrix (sign arbitrary: the caller divides
    by a component).  d v0 = sum_{i != 0} v_i (v_i^T dG v0) / (l0 - l_i)  =>  dL/dG = sym(m v0^T),
    m = sum_{i != 0} v_i (v_i^T g) / (l0 - l_i)."""

    @staticmethod
    def forward(ctx, G):
        from . import ops
        w, V = ops.sym4_eigh(G.detach())
        k = w.argmin(-1)
        v0 = torch.gather(V, -1, k[..., None, None].expand(*k.shape, 4, 1)).squeeze(-1)
        ctx.save_for_backward(w, V, k)
        return v0

    @staticmethod
    def backward(ctx, g):
        w, V, k = ctx.saved_tensors
        l0 = torch.gather(w, -1, k[..., None])                                   # (..., 1)
        v0 = torch.gather(V, -1, k[..., None, None].expand(*k.shape, 4, 1))      # (..., 4, 1)
        den = l0 - w                                                             # 0 at i = k
        scale = w.abs().amax(-1, keepdim=True).clamp_min(1e-300)
        safe = den.abs() > 1e-14 * scale
        coef = torch.where(safe, (V.transpose(-1, -2) @ g[..., None]).squeeze(-1) / torch.where(safe, den, torch.ones_like(den)),
                           torch.zeros_like(den))                                 # (..., 4): (v_i^T g) / (l0 - l_i)
        m = V @ coef[..., None]                                                  # (..., 4, 1)
        M = m @ v0.transpose(-1, -2)
        return 0.5 * (M + M.transpose(-1, -2))
