"""Stage-level wrappers over the C ABI: torch tensors in, torch tensors out.

Each function validates what the C side cannot (device, dtype, contiguity), allocates the
output with torch and enqueues the HIP kernel on torch's current stream.  No host sync.
"""
from __future__ import annotations

import ctypes as C

import math

import numpy as np
import torch

from . import _lib as L

# backward of the sampling op: "det" = deterministic tile-binned form where the shape allows (D = 32), "atomic" = always the
# fp32-atomics form of round 1 (the reference's own scheme)
BACKWARD_MODE = __import__("os").environ.get("MVG_BACKWARD", "det")


# ---- optional per-launch timing with HIP events on the launch stream (bench.py roofline) -----
PROFILE = None   # None (off) or dict name -> list[(start_event, end_event)]


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if PROFILE is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()      # recorded on torch's current stream == the stream the kernel is launched on

    def __exit__(self, *a):
        if PROFILE is not None:
            self.e.record()
            PROFILE.setdefault(self.name, []).append((self.s, self.e))
        return False


def profile_summary():
    """name -> (launches, mean ms) after a torch.cuda.synchronize()."""
    out = {}
    for k, evs in (PROFILE or {}).items():
        ms = [a.elapsed_time(b) for a, b in evs]
        out[k] = (len(ms), sum(ms) / max(len(ms), 1))
    return out


def _i64_host(t):
    """small int64 table (shapes / starts) as a host ctypes array (no device sync if it is
    already a CPU tensor / list)."""
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().tolist()
    flat = np.asarray(t, dtype=np.int64).reshape(-1)
    return (C.c_int64 * len(flat))(*flat.tolist()), flat


def host_levels(spatial_shapes, level_start_index):
    """(shapes_c, starts_c) host ctypes copies of the two level tables.  A device tensor costs one D2H sync the first time it is
    seen; the copy is then kept ON THE TENSOR OBJECT (valid while its version counter is unchanged), so a training loop that
    reuses its level tensors -- and the backward of a forward that already looked them up -- never syncs again."""
    out = []
    for t in (spatial_shapes, level_start_index):
        cached = getattr(t, "_mvg_host", None) if isinstance(t, torch.Tensor) else None
        # keyed on (storage address, version): in-place writes bump the version, .data / set_() re-pointing changes the address
        if cached is not None and cached[0] == (t.data_ptr(), t._version):
            out.append(cached[1])
            continue
        arr, _ = _i64_host(t)
        if isinstance(t, torch.Tensor):
            try:
                t._mvg_host = ((t.data_ptr(), t._version), arr)
            except Exception:      # pragma: no cover - tensor subclasses without a __dict__
                pass
        out.append(arr)
    return tuple(out)


class Levels:
    """Host copy of (spatial_shapes, level_start_index): avoids a D2H sync per call."""

    def __init__(self, spatial_shapes, level_start_index):
        self.shapes_c, shapes = _i64_host(spatial_shapes)
        self.starts_c, starts = _i64_host(level_start_index)
        self.shapes = shapes.reshape(-1, 2)
        self.starts = starts
        self.L = len(starts)
        self.S = int((self.shapes[:, 0] * self.shapes[:, 1]).sum())


# --------------------------------------------------------------------------- Deformable op
def msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    """Deformable.deform_forward (lib/models/ops/src/deform.h:32-50)."""
    L.require_cuda(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    for name, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                    ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)      # deform_cuda.cu:39-43
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes / level_start_index must be int64")
    N, S, M, D = value.shape
    _, Lq, _, nl, P, _ = sampling_loc.shape
    lib = L.load()
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    if Lq == 0:
        return out
    if value.dtype == torch.float32:
        if sampling_loc.dtype != torch.float32 or attn_weight.dtype != torch.float32:
            raise RuntimeError("sampling_loc / attn_weight must be float32")
        rc = lib.mvg_msda_forward_f32(L.ptr(value), L.ptr(spatial_shapes), L.ptr(level_start_index), L.ptr(sampling_loc),
                                      L.ptr(attn_weight), L.ptr(out), N, S, M, D, nl, Lq, P, L.stream_ptr())
    elif value.dtype == torch.float64:      # AT_DISPATCH_FLOATING_TYPES' double case (deform_cuda.cu:75)
        if sampling_loc.dtype != torch.float64 or attn_weight.dtype != torch.float64:
            raise RuntimeError("sampling_loc / attn_weight must be float64")
        rc = lib.mvg_msda_forward_f64(L.ptr(value), L.ptr(spatial_shapes), L.ptr(level_start_index), L.ptr(sampling_loc),
                                      L.ptr(attn_weight), L.ptr(out), N, S, M, D, nl, Lq, P, L.stream_ptr())
    elif value.dtype == torch.bfloat16:
        rc = lib.mvg_msda_forward_bf16(L.ptr(value), L.ptr(spatial_shapes), L.ptr(level_start_index),
                                       L.ptr(sampling_loc.float().contiguous()), L.ptr(attn_weight.float().contiguous()),
                                       L.ptr(out), N, S, M, D, nl, Lq, P, L.stream_ptr())
    else:
        raise RuntimeError("deform_forward: float32 / float64 / bfloat16 only (got %s)" % value.dtype)
    L.check(rc, "mvg_msda_forward")
    return out


def msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, host=None):
    """Deformable.deform_backward (lib/models/ops/src/deform.h:53-72).  host: host_levels(...) of the two tables when the
    caller already has them (DeformFunction keeps the forward's).

    float32 with D = 32 runs the deterministic form (MVG_BACKWARD=atomic selects the reference's fp32-atomic scheme): grad_value is
    summed in 64-bit fixed point with one scale PER IMAGE, 2^30 / (max |grad_output[n]| * max |attn_weight[n]|) over the finite
    entries -- any weight magnitude and any gradient magnitude down to 2^-96 are exact int32 contributions; what is smaller than
    2^-31 of its image's largest possible contribution rounds to zero.  Non-finite grad_output entries do not enter grad_value."""
    L.require_cuda(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output)
    if value.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("deform_backward: float32 / float64 only")        # AT_DISPATCH_FLOATING_TYPES, deform_cuda.cu:145
    for t in (sampling_loc, attn_weight, grad_output):
        if t.dtype != value.dtype:
            raise RuntimeError("deform_backward: all floating tensors must have value's dtype (%s)" % value.dtype)
    grad_output = grad_output.contiguous()
    N, S, M, D = value.shape
    _, Lq, _, nl, P, _ = sampling_loc.shape
    if value.dtype == torch.float32 and BACKWARD_MODE != "atomic":
        # deterministic form (csrc/msda_bwd.hip): binned by destination tile, fixed-point accumulation in LDS
        lib = L.load()
        shapes_c, starts_c = host if host is not None else host_levels(spatial_shapes, level_start_index)
        ws_bytes = int(lib.mvg_msda_backward_det_workspace(N, S, M, D, nl, Lq, P, shapes_c))
        if ws_bytes:
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=value.device)
            gv, gl, ga = torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
            with _timed("msda_backward_det"):
              L.check(lib.mvg_msda_backward_det_f32(L.ptr(value), shapes_c, starts_c, L.ptr(sampling_loc), L.ptr(attn_weight),
                                                    L.ptr(grad_output), L.ptr(gv), L.ptr(gl), L.ptr(ga), N, S, M, D, nl, Lq, P,
                                                    L.ptr(ws), ws_bytes, L.stream_ptr()), "mvg_msda_backward_det_f32")
            return gv, gl, ga
    gv = torch.zeros_like(value)                                             # deform_cuda.cu:132-134
    gl = torch.empty_like(sampling_loc)
    ga = torch.empty_like(attn_weight)
    fn = L.load().mvg_msda_backward_f32 if value.dtype == torch.float32 else L.load().mvg_msda_backward_f64
    rc = fn(L.ptr(value), L.ptr(spatial_shapes), L.ptr(level_start_index),
                                        L.ptr(sampling_loc), L.ptr(attn_weight), L.ptr(grad_output), L.ptr(gv), L.ptr(gl),
                                        L.ptr(ga), N, S, M, D, nl, Lq, P, L.stream_ptr())
    L.check(rc, "mvg_msda_backward")
    return gv, gl, ga


# ----------------------------------------------------------------------------- stage ops
def pyramid_level_views(feat, levels):
    """The L levels of a packed pyramid (n_img, S, C) as (n_img, C, H_l, W_l) tensors in channels-last strides --
    what a producer (the backbone's last deconvolutions run in torch.channels_last) writes into so that
    pack_pyramid has nothing left to do."""
    n_img, _, Cc = feat.shape
    views = []
    for l in range(levels.L):
        H, W = int(levels.shapes[l, 0]), int(levels.shapes[l, 1])
        st = int(levels.starts[l])
        views.append(feat[:, st:st + H * W, :].view(n_img, H, W, Cc).permute(0, 3, 1, 2))
    return views


def pack_pyramid(src_views, levels, dtype, out=None):
    """list of L (N_img,C,H,W) maps -> channels-last pyramid (N_img,S,C) in `dtype`.

    Per level, by layout of the producer's tensor:
      * it IS the level view of `out` (pyramid_level_views) in `dtype`: produced in place, nothing to do;
      * torch.channels_last memory format (fp32 / bf16 / fp16): one cast+copy, no transposition;
      * NCHW (what the reference's backbone emits, pose_resnet.py:198-216): LDS-tiled transpose kernel."""
    lib = L.load()
    n_img, Cc = src_views[0].shape[:2]
    feat = out if out is not None else torch.empty((n_img, levels.S, Cc), dtype=dtype, device=src_views[0].device)
    if tuple(feat.shape) != (n_img, levels.S, Cc) or feat.dtype != dtype or not feat.is_contiguous():
        raise RuntimeError("pack_pyramid: out must be a contiguous (%d,%d,%d) %s tensor" % (n_img, levels.S, Cc, dtype))
    dst_views = pyramid_level_views(feat, levels)
    if len(src_views) != levels.L:
        raise RuntimeError("pack_pyramid: %d feature maps for %d levels" % (len(src_views), levels.L))
    if (all(s.dtype == torch.float32 and s.is_contiguous() and s.is_cuda and tuple(s.shape) ==
                                (n_img, Cc, int(levels.shapes[l, 0]), int(levels.shapes[l, 1])) and s.data_ptr() != dst_views[l].data_ptr()
                                for l, s in enumerate(src_views))):
        # the reference's hand-over format on every level: one launch for the whole pyramid
        ptrs = (C.c_void_p * levels.L)(*[s.data_ptr() for s in src_views])
        with _timed("pack_level"):
            L.check(lib.mvg_pack_pyramid(ptrs, L.ptr(feat), L.dtype_code(dtype), n_img, Cc, levels.shapes_c, levels.starts_c,
                                         levels.L, levels.S, L.stream_ptr()), "mvg_pack_pyramid")
        return feat
    for l, src in enumerate(src_views):
        L.require_cuda(src)
        H, W = int(levels.shapes[l, 0]), int(levels.shapes[l, 1])
        if tuple(src.shape) != (n_img, Cc, H, W):
            raise RuntimeError("src_views[%d] has shape %s, expected %s" % (l, tuple(src.shape), (n_img, Cc, H, W)))
        dst = dst_views[l]
        if src.data_ptr() == dst.data_ptr() and src.stride() == dst.stride() and src.dtype == dtype:
            continue
        if src.is_contiguous(memory_format=torch.channels_last) and not src.is_contiguous():
            with _timed("pack_level_nhwc"):
                dst.copy_(src)
            continue
        s = src.float().contiguous()
        one = (C.c_void_p * 1)(s.data_ptr())
        with _timed("pack_level"):      # the same kernel on this level alone
          L.check(lib.mvg_pack_pyramid(one, L.ptr(feat), L.dtype_code(dtype), n_img, Cc, (C.c_int64 * 2)(H, W),
                                       (C.c_int64 * 1)(int(levels.starts[l])), 1, levels.S, L.stream_ptr()), "mvg_pack_pyramid")
    return feat


def project(X, cams, levels, V, B):
    Lq = X.shape[1]
    r = torch.empty((V * B, Lq, 2), dtype=torch.float32, device=X.device)
    ref_lvl = torch.empty((V * B, Lq, levels.L, 2), dtype=torch.float32, device=X.device)
    inside = torch.empty((V * B, Lq), dtype=torch.uint8, device=X.device)
    L.check(L.load().mvg_project(L.ptr(X), L.ptr(cams), levels.shapes_c, levels.L, L.ptr(r), L.ptr(ref_lvl),
                                 L.ptr(inside), V, B, Lq, L.stream_ptr()), "mvg_project")
    return r, ref_lvl, inside


def gather_ref(feat, r, x, levels, V, B):
    """r: per-level reference points (n_img, Lq, L, 2)."""
    n_img, S, Cc = feat.shape
    Lq = r.shape[1]
    ain = torch.empty((n_img * Lq * levels.L, Cc), dtype=feat.dtype, device=feat.device)
    with _timed("gather_ref"):
      L.check(L.load().mvg_gather_ref(L.ptr(feat), L.dtype_code(feat.dtype), L.ptr(r), L.ptr(x), levels.shapes_c,
                                    levels.starts_c, L.ptr(ain), V, B, Lq, levels.L, S, Cc, L.stream_ptr()),
            "mvg_gather_ref")
    return ain


def linear(a, w, bias, out_dtype=None, relu=False, rowmask=None, out=None, add=None):
    """out = act((a [+ add]) @ w.T + bias) (* rowmask).  a (M,K) f32|bf16; w (N,K) f32|bf16 (compute type); add: optional fp32
    tensor of a's shape and strides, summed with a on load (mvg_linear_sum)."""
    M, K = a.shape
    N = w.shape[0]
    out_dtype = out_dtype or (torch.float32 if w.dtype == torch.float32 else torch.bfloat16)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    if a.stride(1) != 1 or not w.is_contiguous() or out.stride(1) != 1:
        raise RuntimeError("mvg_linear: K-contiguous operands required")
    if add is not None and (add.dtype != torch.float32 or a.dtype != torch.float32 or add.shape != a.shape
                            or add.stride() != a.stride()):
        raise RuntimeError("mvg_linear_sum: the addend must be an fp32 tensor with a's shape and strides")
    with _timed("linear_%dx%dx%d" % (M, N, K)):
      L.check(L.load().mvg_linear_sum(L.ptr(a), L.ptr(add), L.dtype_code(a.dtype), a.stride(0), L.ptr(w),
                                    L.dtype_code(w.dtype), L.ptr(bias), L.ptr(out), L.dtype_code(out.dtype), out.stride(0),
                                    L.ptr(rowmask), 1 if relu else 0, M, N, K, L.stream_ptr()), "mvg_linear_sum")
    return out


def linear_wgrad(dy, x, splits=None):
    """dW (N, K) = dy^T x for dy (rows, N), x (rows, K) fp32 row-major: linear_wgrad_bias without the bias gradient"""
    return linear_wgrad_bias(dy, x, want_bias=False, splits=splits)[0]


def linear_wgrad_bias(dy, x, want_bias=True, splits=None):
    """(dW (N, K), db (N) | None) = (dy^T x, column sums of dy) (mvg_linear_wgrad_bias_f32): the weight-gradient launch also leaves the
    bias gradient's per-slice partials, a second small launch adds both in slice order."""
    rows, N = dy.shape
    K = x.shape[1]
    if (dy.dtype != torch.float32 or x.dtype != torch.float32 or dy.stride(1) != 1 or x.stride(1) != 1 or x.shape[0] != rows
            or N % 4 or K % 4):
        raise RuntimeError("mvg_linear_wgrad_bias: fp32 row-major (rows, N) / (rows, K) operands with N, K multiples of 4 required")
    if splits is None:
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        # ~512 workgroups (two per CU), at most 128 slices, at least 256 rows per slice (tools/bench_wgrad.py)
        splits = max(1, min((rows + 255) // 256, 128, (512 + tiles - 1) // tiles))
    partial = torch.empty((splits, N, K), dtype=torch.float32, device=dy.device)
    dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    pdb = torch.empty((splits, N), dtype=torch.float32, device=dy.device) if want_bias else None
    db = torch.empty((N,), dtype=torch.float32, device=dy.device) if want_bias else None
    with _timed("linear_wgrad_bias_%dx%dx%d" % (N, K, rows)):
      L.check(L.load().mvg_linear_wgrad_bias_f32(L.ptr(dy), dy.stride(0), L.ptr(x), x.stride(0), L.ptr(partial), L.ptr(pdb), L.ptr(dw),
                                                 L.ptr(db), rows, N, K, splits, L.stream_ptr()), "mvg_linear_wgrad_bias_f32")
    return dw, db


def linear_ordered(a, w, bias, order, inside, masked_row, relu=False, rowmask=None, out=None):
    """fp32 ``linear`` over the rows in processing order ``order`` (mvg_linear_ordered): tiles without a row of ``inside`` write
    ``masked_row`` (N,) to all their rows instead of computing them.  Rows keep their places in a / out."""
    M, K = a.shape
    N = w.shape[0]
    if a.dtype != torch.float32 or w.dtype != torch.float32 or a.stride(1) != 1 or not w.is_contiguous():
        raise RuntimeError("mvg_linear_ordered: fp32 K-contiguous operands required")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    # the kernel indexes a / out through `order` and reads `inside` / `masked_row` unchecked: a stale order (another query count)
    # would read and write out of bounds
    if out.dtype != torch.float32 or tuple(out.shape) != (M, N) or out.stride(1) != 1:
        raise RuntimeError("mvg_linear_ordered: out must be an fp32 (%d, %d) tensor with contiguous rows" % (M, N))
    if order.dtype != torch.int32 or order.numel() != M or not order.is_contiguous():
        raise RuntimeError("mvg_linear_ordered: order must be a contiguous int32 tensor with one entry per row (%d)" % M)
    if inside.dtype != torch.uint8 or inside.numel() != M or not inside.is_contiguous():
        raise RuntimeError("mvg_linear_ordered: inside must be a contiguous uint8 tensor with one entry per row (%d)" % M)
    if masked_row.dtype != torch.float32 or masked_row.numel() != N or not masked_row.is_contiguous():
        raise RuntimeError("mvg_linear_ordered: masked_row must be a contiguous fp32 tensor with %d entries" % N)
    if rowmask is not None and (rowmask.dtype != torch.uint8 or rowmask.numel() != M or not rowmask.is_contiguous()):
        raise RuntimeError("mvg_linear_ordered: rowmask must be a contiguous uint8 tensor with one entry per row (%d)" % M)
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N):
        raise RuntimeError("mvg_linear_ordered: bias must be fp32 with %d entries" % N)
    with _timed("linear_ordered_%dx%dx%d" % (M, N, K)):
      L.check(L.load().mvg_linear_ordered(L.ptr(a), a.stride(0), L.ptr(w), L.ptr(bias), L.ptr(out), out.stride(0),
                                        L.ptr(rowmask), 1 if relu else 0, M, N, K, L.ptr(order), L.ptr(inside),
                                        L.ptr(masked_row), L.stream_ptr()), "mvg_linear_ordered")
    return out


def msda_fused(value, oa, r, levels):
    """r: per-level reference points (n_img, Lq, L, 2)."""
    n_img, S, Cc = value.shape
    Lq = r.shape[1]
    samp = torch.empty((n_img * Lq, Cc), dtype=value.dtype, device=value.device)
    with _timed("msda_fused"):
      L.check(L.load().mvg_msda_fused(L.ptr(value), L.dtype_code(value.dtype), L.ptr(oa), L.ptr(r), levels.shapes_c,
                                    levels.starts_c, L.ptr(samp), n_img, Lq, levels.L, S, L.stream_ptr()),
            "mvg_msda_fused")
    return samp


def msda_gfused_f32(value, G, xw, r, levels, B, pair_mask=None, order=None):
    """fp32 G-sampling (include/mvg_decoder.h: mvg_msda_gfused_f32): value (n_img,S,256) f32, G (n_img*S,192) f32 and
    xw (B*Lq,192) f32 in gsamp_column_order, r (n_img,Lq,L,2) -> samp (n_img*Lq,256) f32."""
    n_img, S, Cc = value.shape
    Lq = r.shape[1]
    assert value.dtype == torch.float32 and G.dtype == torch.float32 and xw.dtype == torch.float32 and Cc == 256
    assert G.is_contiguous() and tuple(G.shape) == (n_img * S, 192) and xw.is_contiguous() and xw.shape[1] == 192
    samp = torch.empty((n_img * Lq, 256), dtype=torch.float32, device=value.device)
    if pair_mask is not None:
        assert pair_mask.dtype == torch.uint8 and pair_mask.numel() == n_img * Lq and pair_mask.is_contiguous()
    if order is not None:
        assert order.dtype == torch.int32 and order.numel() == n_img * Lq and order.is_contiguous()
    with _timed("msda_gfused_f32"):
      L.check(L.load().mvg_msda_gfused_f32(L.ptr(value), L.ptr(G), L.ptr(xw), L.ptr(r), levels.shapes_c, levels.starts_c,
                                           L.ptr(samp), None if pair_mask is None else L.ptr(pair_mask),
                                           None if order is None else L.ptr(order), n_img, Lq, levels.L, S, B,
                                           L.stream_ptr()), "mvg_msda_gfused_f32")
    return samp


def value_proj_planes_ws(feat, w_frag, bias, vp):
    """weight-stationary value projection (bf16 feat, swizzled weight) into head planes vh[img][head][s][32]: a one-product launch
    of mvg_pyramid_group_ws."""
    pyramid_group_ws(feat, [(w_frag, bias, vp, True)], label="value_proj_ws")
    return vp


def feat_linear_ws(feat, w_frag, N=192, out=None):
    """G = feat @ W^T, row-major bf16 (n_img*S, 192) (weight-stationary kernel, no bias): a one-product launch of
    mvg_pyramid_group_ws."""
    n_img, S, _ = feat.shape
    if N != 192:
        raise RuntimeError("feat_linear_ws: 192 columns (the [offsets | logits] Linear of the G-sampling form)")
    G = out if out is not None else torch.empty((n_img * S, N), dtype=torch.bfloat16, device=feat.device)
    pyramid_group_ws(feat, [(w_frag, None, G, False)], label="feat_linear_ws")
    return G


def pyramid_group_ws(feat, jobs, slots=0, label=None):
    """Several weight-stationary products of the same packed bf16 pyramid in one launch (mvg_pyramid_group_ws): jobs = list of
    (w_frag, bias or None, out, planes) -- planes: value projection into head planes (out = vh (n_img, 8, S, 32)), else
    G = feat @ W^T row-major (out (n_img*S, 192)).  Outputs bit-identical to value_proj_planes_ws / feat_linear_ws.
    slots: workgroups per XCD (0 = the library default, all 64 resident ones; fewer leave room for kernels running next to it)."""
    n_img, S, K = feat.shape
    n = len(jobs)
    assert 1 <= n <= 8 and K == 256 and feat.dtype == torch.bfloat16 and feat.is_contiguous()
    for w, b, out, planes in jobs:
        assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.numel() == n_img * S * (256 if planes else 192)
        assert w.dtype == torch.bfloat16 and w.numel() == 256 * 256 and (b is not None or not planes)
    vpp = C.c_void_p * n
    Wf = vpp(*[w.data_ptr() for w, _, _, _ in jobs])
    bias = vpp(*[(b.data_ptr() if b is not None else None) for _, b, _, _ in jobs])
    outs = vpp(*[o.data_ptr() for _, _, o, _ in jobs])
    Ns = (C.c_int * n)(*[256 if p else 192 for _, _, _, p in jobs])
    planes = (C.c_int * n)(*[1 if p else 0 for _, _, _, p in jobs])
    with _timed(label or "pyramid_group_ws_%d" % n):
      L.check(L.load().mvg_pyramid_group_ws(L.ptr(feat), n_img, S, n, Wf, bias, outs, Ns, planes, int(slots), L.stream_ptr()),
              "mvg_pyramid_group_ws")


def gsamp_column_order(device=None):
    """Row permutation of [sampling_offsets.weight (128); attention_weights.weight (64)] that msda_gsamp expects
    for G and xw: 8 groups of (16 offset rows | 8 logit rows).  With the reference's memory reinterpretation
    (projattn.py:180-184) head m of a level row uses whole groups, so its 72 values are contiguous in a G row."""
    j = torch.arange(192, dtype=torch.long, device=device)       # built on the device: legal during graph capture
    g, w = j // 24, j % 24
    return torch.where(w < 16, 16 * g + w, 128 + 8 * g + (w - 16))


def msda_gsamp(vp, G, xw, r, levels, B, pair_mask=None, order=None, out=None):
    """fused sampling with in-kernel gather of the offsets/logits from G (see include/mvg_decoder.h); the 192
    columns of G and xw are in gsamp_column_order().
    pair_mask (n_img*Lq) u8: rows with 0 are zero-filled, not sampled; order (n_img*Lq) i32 from bin_pairs."""
    n_img = vp.shape[0]
    Lq = r.shape[1]
    samp = out if out is not None else torch.empty((n_img * Lq, 256), dtype=torch.bfloat16, device=vp.device)
    assert samp.dtype == torch.bfloat16 and tuple(samp.shape) == (n_img * Lq, 256) and samp.is_contiguous()
    if pair_mask is not None:
        assert pair_mask.dtype == torch.uint8 and pair_mask.numel() == n_img * Lq and pair_mask.is_contiguous()
    if order is not None:
        assert order.dtype == torch.int32 and order.numel() == n_img * Lq and order.is_contiguous()
    with _timed("msda_gsamp"):
      L.check(L.load().mvg_msda_gsamp(L.ptr(vp), L.ptr(G), L.ptr(xw), L.ptr(r), levels.shapes_c, levels.starts_c,
                                      L.ptr(samp), None if pair_mask is None else L.ptr(pair_mask),
                                      None if order is None else L.ptr(order), n_img, Lq, levels.L, levels.S, B,
                                      L.stream_ptr()), "mvg_msda_gsamp")
    return samp


def bin_pairs(r, inside, levels, out=None):
    """processing order of the (image, query) pairs for msda_gsamp: Morton-sorted by level-0 cell block, pairs with
    inside == 0 last.  r (n_img, Lq, L, 2) per-level reference points; inside (n_img, Lq) u8 or None."""
    n_img, Lq = r.shape[0], r.shape[1]
    if Lq > 65536:          # the binning kernel keeps a thread's keys in registers (<= 64 per thread)
        return None
    if out is None:
        out = torch.empty((n_img * Lq,), dtype=torch.int32, device=r.device)
    order = out
    lib = L.load()
    ws_bytes = int(lib.mvg_bin_pairs_workspace(n_img, Lq))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=r.device) if ws_bytes else None     # multi-workgroup variant
    with _timed("bin_pairs"):
      L.check(lib.mvg_bin_pairs(L.ptr(r), None if inside is None else L.ptr(inside), levels.shapes_c, levels.L,
                                L.ptr(order), n_img, Lq, None if ws is None else L.ptr(ws), ws_bytes, L.stream_ptr()),
              "mvg_bin_pairs")
    return order


_SIDE_STREAMS = {}


def side_stream(device):
    """one auxiliary HIP stream per device for small kernels that overlap with the main stream (fork/join with
    wait_stream: legal inside HIP-graph capture)."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _SIDE_STREAMS:
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return _SIDE_STREAMS[idx]


def swizzle_weight(w):
    """nn.Linear weight (N, K) bf16 with N % 256 == 0 -> MFMA-fragment order of csrc/chain.hip:
    Wf[nb N/256][wn 4][ks K/16][j 2][lane 64][8], lane = h*32 + rl holds
    W[nb*256 + wn*64 + j*32 + rl][ks*16 + h*8 : +8]."""
    N, K = w.shape
    assert N % 256 == 0 and K % 16 == 0, (N, K)
    t = w.reshape(N // 256, 4, 2, 32, K // 16, 2, 8)          # nb, wn, j, rl, ks, h, e
    return t.permute(0, 1, 4, 2, 5, 3, 6).contiguous().reshape(-1)


def split_swizzle_weight_h2(w, pad_rows_to=256):
    """fp32 nn.Linear weight (N, K) -> the operand of the two-part fp16 kernels (csrc/f32s.hip, "f32h"): (planes, s) with
    w 2^s = h + l, h = fp16(w 2^s), l = fp16(w 2^s - h), s chosen so that max |w| 2^s lies in [2^13, 2^14); each plane in swizzle_weight
    order, concatenated (N zero-padded to a multiple of 256).  One host sync (the tensor's maximum) when the operand cache is built."""
    w = w.detach().float()
    N, K = w.shape
    Np = (N + pad_rows_to - 1) // pad_rows_to * pad_rows_to
    if Np != N:
        w = torch.cat([w, w.new_zeros(Np - N, K)], 0)
    mx = float(w.abs().max())
    s = 0 if not (mx > 0 and math.isfinite(mx)) else max(-100, min(100, 13 - math.frexp(mx)[1] + 1))
    ws = torch.ldexp(w, torch.tensor(s, device=w.device))
    h = ws.to(torch.float16)
    l = (ws - h.float()).to(torch.float16)
    return torch.cat([swizzle_weight(p) for p in (h, l)]), s


def pyramid_f32h(feat, Wv_planes, wv_scale, bv, Wg_planes, wg_scale, n_g, value=None, G=None):
    """value = feat Wv^T + bv (n_img, S, 256) and G = feat Wg^T (n_img*S, n_g), fp32, in one pass over the channels-last fp32
    pyramid feat (n_img, S, 256) on two-part fp16 operands (include/mvg_decoder.h: mvg_pyramid_f32h); weights from
    split_swizzle_weight_h2."""
    n_img, S, Cc = feat.shape
    if feat.dtype != torch.float32 or Cc != 256 or not feat.is_contiguous():
        raise RuntimeError("mvg_pyramid_f32h: contiguous fp32 (n_img, S, 256) pyramid required")
    rows = n_img * S
    for t in (Wv_planes, Wg_planes):
        if t.dtype != torch.float16 or t.numel() != 2 * 256 * 256 or not t.is_contiguous():
            raise RuntimeError("mvg_pyramid_f32h: weight planes from split_swizzle_weight_h2 required")
    if value is None:
        value = torch.empty((n_img, S, 256), dtype=torch.float32, device=feat.device)
    if G is None:
        G = torch.empty((rows, n_g), dtype=torch.float32, device=feat.device)
    assert value.dtype == torch.float32 and value.numel() == rows * 256 and value.is_contiguous()
    assert G.dtype == torch.float32 and tuple(G.shape) == (rows, n_g) and G.is_contiguous()
    with _timed("pyramid_f32h"):
      L.check(L.load().mvg_pyramid_f32h(L.ptr(feat), L.ptr(Wv_planes), int(wv_scale), L.ptr(bv), L.ptr(Wg_planes), int(wg_scale),
                                        L.ptr(value), L.ptr(G), rows, n_g, L.stream_ptr()), "mvg_pyramid_f32h")
    return value, G


def chain_attn_pose_f32h(samp, inside, Wp, swp, bp, W0, sw0, b0, W1, sw1, b1, W2, b2, order=None, o_masked=None):
    """fp32 chain A on two-part fp16 operands (include/mvg_decoder.h: mvg_chain_attn_pose_f32h): samp (rows, 256) f32 -> (attn f32
    (rows, 256), o f32 (rows, 3)); (Wp, swp) / (W0, sw0) / (W1, sw1) from split_swizzle_weight_h2."""
    rows = samp.shape[0]
    if samp.dtype != torch.float32 or samp.shape[1] != 256 or not samp.is_contiguous():
        raise RuntimeError("mvg_chain_attn_pose_f32h: contiguous fp32 (rows, 256) samples required")
    for t in (Wp, W0, W1):
        if t.dtype != torch.float16 or t.numel() != 2 * 256 * 256 or not t.is_contiguous():
            raise RuntimeError("mvg_chain_attn_pose_f32h: weight planes from split_swizzle_weight_h2 required")
    if inside.dtype != torch.uint8 or inside.numel() != rows or not inside.is_contiguous():
        raise RuntimeError("mvg_chain_attn_pose_f32h: inside must be a contiguous uint8 tensor with one entry per row")
    if W2.dtype != torch.float32 or tuple(W2.shape) != (3, 256) or not W2.is_contiguous():
        raise RuntimeError("mvg_chain_attn_pose_f32h: the last pose layer's (3, 256) fp32 weight required")
    if order is not None:
        assert order.dtype == torch.int32 and order.numel() == rows and order.is_contiguous()
    attn = torch.empty((rows, 256), dtype=torch.float32, device=samp.device)
    o = torch.empty((rows, 3), dtype=torch.float32, device=samp.device)
    with _timed("chain_attn_pose_f32h"):
      L.check(L.load().mvg_chain_attn_pose_f32h(L.ptr(samp), L.ptr(inside), L.ptr(Wp), int(swp), L.ptr(bp), L.ptr(W0), int(sw0), L.ptr(b0),
                                                L.ptr(W1), int(sw1), L.ptr(b1), L.ptr(W2), L.ptr(b2), L.ptr(attn), L.ptr(o),
                                                None if order is None else L.ptr(order),
                                                None if o_masked is None else L.ptr(o_masked), rows, L.stream_ptr()),
              "mvg_chain_attn_pose_f32h")
    return attn, o


def chain_masked_row_output_f32h(Wp, swp, bp, W0, sw0, b0, W1, sw1, b1, W2, b2):
    """o (3,) f32 of a row with inside == 0 as the two-part fp16 chain A computes it (the chain run on one masked row)."""
    dev = Wp.device
    samp = torch.zeros((1, 256), dtype=torch.float32, device=dev)
    inside = torch.zeros((1,), dtype=torch.uint8, device=dev)
    global PROFILE
    saved, PROFILE = PROFILE, None
    try:
        _, o = chain_attn_pose_f32h(samp, inside, Wp, swp, bp, W0, sw0, b0, W1, sw1, b1, W2, b2)
    finally:
        PROFILE = saved
    return o.reshape(3)


def chain_update_ffn_class_f32h(attn, V, tgt, Wu, su, bu, g2, be2, W1, s1, b1, W2, s2, b2, g3, be3, Wc, bc, threshold, B, NQ, J,
                                forced_valid=None, has_ffn=True, tgt_out=None, any_valid=None, next_query_proj=None):
    """fp32 chain B on two-part fp16 operands (include/mvg_decoder.h: mvg_chain_update_ffn_class_f32h); arguments and results as
    chain_update_ffn_class, attn (V * rows, 256) f32, weights (planes, scale) from split_swizzle_weight_h2, next_query_proj = (qpos, Wn, sn, bn, n_next)."""
    dev = attn.device
    rows = B * NQ * J
    if attn.dtype != torch.float32 or attn.numel() != V * rows * 256 or not attn.is_contiguous():
        raise RuntimeError("mvg_chain_update_ffn_class_f32h: contiguous fp32 (V * rows, 256) attn required")
    assert tgt.dtype == torch.float32 and tgt.numel() == rows * 256 and tgt.is_contiguous()
    assert Wu.dtype == torch.float16 and Wu.numel() == 2 * 256 * 256
    if has_ffn:
        assert W1.dtype == torch.float16 and W1.numel() == 2 * 1024 * 256 and W2.dtype == torch.float16 and W2.numel() == 2 * 256 * 1024
    if tgt_out is None:
        tgt_out = torch.empty((rows, 256), dtype=torch.float32, device=dev)
    else:
        assert tgt_out.dtype == torch.float32 and tgt_out.numel() == rows * 256 and tgt_out.is_contiguous()
        tgt_out = tgt_out.view(rows, 256)
    prob = torch.empty((B, NQ, 2), dtype=torch.float32, device=dev)
    valid = torch.empty((B, NQ), dtype=torch.uint8, device=dev)
    qpos = Wn = bn = xw_next = None
    n_next = sn = 0
    if next_query_proj is not None:
        qpos, Wn, sn, bn, n_next = next_query_proj
        assert Wn.dtype == torch.float16 and Wn.numel() == 2 * 256 * 256 and bn.numel() == 256 and bn.dtype == torch.float32
        if qpos is not None:
            assert qpos.dtype == torch.float32 and qpos.numel() == rows * 256 and qpos.is_contiguous()
        xw_next = torch.empty((rows, n_next), dtype=torch.float32, device=dev)
    if any_valid is None:
        any_valid = torch.zeros((1,), dtype=torch.int32, device=dev)
    with _timed("chain_update_ffn_class_f32h"):
      L.check(L.load().mvg_chain_update_ffn_class_f32h(
          L.ptr(attn), V, L.ptr(tgt), L.ptr(Wu), int(su), L.ptr(bu), L.ptr(g2), L.ptr(be2), L.ptr(W1), int(s1 or 0), L.ptr(b1), L.ptr(W2),
          int(s2 or 0), L.ptr(b2), L.ptr(g3), L.ptr(be3), L.ptr(Wc), L.ptr(bc), float(threshold), L.ptr(forced_valid), L.ptr(tgt_out),
          L.ptr(prob), L.ptr(valid), L.ptr(any_valid), L.ptr(qpos), L.ptr(Wn), int(sn), L.ptr(bn), L.ptr(xw_next), n_next, B, NQ, J,
          1 if has_ffn else 0, L.stream_ptr()), "mvg_chain_update_ffn_class_f32h")
    if next_query_proj is not None:
        return tgt_out, prob, valid, any_valid, xw_next
    return tgt_out, prob, valid, any_valid


def chain_attn_pose(samp, inside, Wp, bp, W0, b0, W1, b1, W2, b2, order=None, o_masked=None):
    """fused output_proj (* in-image mask) + 3-layer pose MLP; Wp/W0/W1 in swizzle_weight order.
    order (rows) i32: row processing order (bin_pairs: masked rows last); o_masked (3) f32 from
    chain_masked_row_output: lets all-masked 64-row tiles skip the chain.
    Returns (attn bf16 (rows,256), o f32 (rows,3))."""
    rows = samp.shape[0]
    attn = torch.empty((rows, 256), dtype=torch.bfloat16, device=samp.device)
    o = torch.empty((rows, 3), dtype=torch.float32, device=samp.device)
    if order is not None:
        assert order.dtype == torch.int32 and order.numel() == rows and order.is_contiguous()
    with _timed("chain_attn_pose"):
      L.check(L.load().mvg_chain_attn_pose(L.ptr(samp), L.ptr(inside), L.ptr(Wp), L.ptr(bp), L.ptr(W0), L.ptr(b0), L.ptr(W1),
                                           L.ptr(b1), L.ptr(W2), L.ptr(b2), L.ptr(attn), L.ptr(o),
                                           None if order is None else L.ptr(order),
                                           None if o_masked is None else L.ptr(o_masked), rows, L.stream_ptr()),
              "mvg_chain_attn_pose")
    return attn, o


def chain_masked_row_output(Wp, bp, W0, b0, W1, b1, W2, b2):
    """o (3,) f32 of a row with inside == 0: the chain run on one masked row (weights only -> cacheable)."""
    dev = Wp.device
    samp = torch.zeros((1, 256), dtype=torch.bfloat16, device=dev)
    inside = torch.zeros((1,), dtype=torch.uint8, device=dev)
    global PROFILE
    saved, PROFILE = PROFILE, None
    try:
        _, o = chain_attn_pose(samp, inside, Wp, bp, W0, b0, W1, b1, W2, b2)
    finally:
        PROFILE = saved
    return o.reshape(3)


def chain_update_ffn_class(attn, V, tgt, Wu, bu, g2, be2, W1, b1, W2, b2, g3, be3, Wc, bc, threshold, B, NQ, J,
                           forced_valid=None, has_ffn=True, tgt_out=None, any_valid=None, next_query_proj=None, attn_inside=None):
    """fused view-mean + update MLP + LN2 + FFN + LN3 + class head (weights in swizzle_weight order).
    attn_inside (V * rows) u8: chain A's in-image flags -- attn rows with flag 0 are zero and are not read.
    Returns (tgt_update f32 (B*NQ*J,256), prob (B,NQ,2), valid (B,NQ) u8, any_valid int32[1])."""
    dev = attn.device
    rows = B * NQ * J
    if tgt_out is None:
        tgt_out = torch.empty((rows, 256), dtype=torch.float32, device=dev)
    else:       # caller-owned destination (e.g. a slice of the stacked per-layer output)
        assert tgt_out.dtype == torch.float32 and tgt_out.numel() == rows * 256 and tgt_out.is_contiguous()
        tgt_out = tgt_out.view(rows, 256)
    prob = torch.empty((B, NQ, 2), dtype=torch.float32, device=dev)
    valid = torch.empty((B, NQ), dtype=torch.uint8, device=dev)
    # next_query_proj = (query_pos (rows,256) f32 | None, W_next fragments (256,256) bf16, b_next (256,) f32, n_next):
    # also emit xw_next = (tgt' + query_pos) @ W_next^T + b_next for the next layer's sampler
    qpos = Wn = bn = xw_next = None
    n_next = 0
    if next_query_proj is not None:
        qpos, Wn, bn, n_next = next_query_proj
        assert Wn.dtype == torch.bfloat16 and Wn.numel() == 256 * 256 and bn.numel() == 256 and bn.dtype == torch.float32
        if qpos is not None:
            assert qpos.dtype == torch.float32 and qpos.numel() == rows * 256 and qpos.is_contiguous()
        xw_next = torch.empty((rows, n_next), dtype=torch.float32, device=dev)
    if any_valid is None:       # else: a caller-owned int32[1] that is already zero
        any_valid = torch.zeros((1,), dtype=torch.int32, device=dev)
    if attn_inside is not None:
        assert attn_inside.dtype == torch.uint8 and attn_inside.numel() == V * rows and attn_inside.is_contiguous()
    with _timed("chain_update_ffn_class"):
      L.check(L.load().mvg_chain_update_ffn_class(
          L.ptr(attn), V, L.ptr(tgt), L.ptr(Wu), L.ptr(bu), L.ptr(g2), L.ptr(be2), L.ptr(W1), L.ptr(b1), L.ptr(W2), L.ptr(b2),
          L.ptr(g3), L.ptr(be3), L.ptr(Wc), L.ptr(bc), float(threshold), L.ptr(forced_valid), L.ptr(tgt_out), L.ptr(prob),
          L.ptr(valid), L.ptr(any_valid), L.ptr(qpos), L.ptr(Wn), L.ptr(bn), L.ptr(xw_next), n_next, B, NQ, J,
          1 if has_ffn else 0, L.ptr(attn_inside), L.stream_ptr()), "mvg_chain_update_ffn_class")
    if next_query_proj is not None:
        return tgt_out, prob, valid, any_valid, xw_next
    return tgt_out, prob, valid, any_valid


def mean_views(attn, V):
    rows = attn.shape[0] // V
    out = torch.empty((rows, attn.shape[1]), dtype=attn.dtype, device=attn.device)
    L.check(L.load().mvg_mean_views(L.ptr(attn), L.dtype_code(attn.dtype), L.ptr(out), V, rows, attn.shape[1],
                                    L.stream_ptr()), "mvg_mean_views")
    return out


def add_layernorm(res, h, gamma, beta):
    rows, Cc = res.shape
    y = torch.empty((rows, Cc), dtype=torch.float32, device=res.device)
    L.check(L.load().mvg_add_layernorm(L.ptr(res), L.ptr(h), L.dtype_code(h.dtype), L.ptr(gamma), L.ptr(beta), L.ptr(y),
                                       rows, Cc, L.stream_ptr()), "mvg_add_layernorm")
    return y


def class_head(tgt, Wc, bc, threshold, B, NQ, J, forced_valid=None):
    Cc = tgt.shape[-1]
    prob = torch.empty((B, NQ, 2), dtype=torch.float32, device=tgt.device)
    valid = torch.empty((B, NQ), dtype=torch.uint8, device=tgt.device)
    any_valid = torch.zeros((1,), dtype=torch.int32, device=tgt.device)
    L.check(L.load().mvg_class_head(L.ptr(tgt), L.ptr(Wc), L.ptr(bc), float(threshold), L.ptr(forced_valid), L.ptr(prob),
                                    L.ptr(valid), L.ptr(any_valid), B, NQ, J, Cc, L.stream_ptr()), "mvg_class_head")
    return prob, valid, any_valid


def rowdot3(h, W3, b3):
    rows, Cc = h.shape
    o = torch.empty((rows, 3), dtype=torch.float32, device=h.device)
    L.check(L.load().mvg_rowdot3(L.ptr(h), L.dtype_code(h.dtype), L.ptr(W3), L.ptr(b3), L.ptr(o), rows, Cc,
                                 L.stream_ptr()), "mvg_rowdot3")
    return o


def triangulate(r, o, cams, valid, any_valid, V, B, NQ, J, out=None, next_levels=None):
    """out: optional caller-owned (new_ref (B,Lq,3), ref2d (B,V,Lq,2), proj2d (B,V,Lq,2)) contiguous f32 destinations
    (e.g. slices of the stacked per-layer outputs).  next_levels (a Levels): also project the new points for the next
    layer in the same launch -> returns (new_ref, ref2d, proj2d, (r_next, ref_lvl_next, inside_next))."""
    Lq = NQ * J
    dev = r.device
    if out is not None:
        new_ref, ref2d, proj2d = out
        for t, shp in ((new_ref, (B, Lq, 3)), (ref2d, (B, V, Lq, 2)), (proj2d, (B, V, Lq, 2))):
            assert tuple(t.shape) == shp and t.dtype == torch.float32 and t.is_contiguous()
    else:
        new_ref = torch.empty((B, Lq, 3), dtype=torch.float32, device=dev)
        ref2d = torch.empty((B, V, Lq, 2), dtype=torch.float32, device=dev)
        proj2d = torch.empty((B, V, Lq, 2), dtype=torch.float32, device=dev)
    if next_levels is not None:
        lv = next_levels
        r_n = torch.empty((V * B, Lq, 2), dtype=torch.float32, device=dev)
        ref_n = torch.empty((V * B, Lq, lv.L, 2), dtype=torch.float32, device=dev)
        in_n = torch.empty((V * B, Lq), dtype=torch.uint8, device=dev)
        with _timed("triangulate_project"):
          L.check(L.load().mvg_triangulate_project(L.ptr(r), L.ptr(o), L.ptr(cams), L.ptr(valid), L.ptr(any_valid),
                                                   L.ptr(new_ref), L.ptr(ref2d), L.ptr(proj2d), V, B, NQ, J, lv.shapes_c, lv.L,
                                                   L.ptr(r_n), L.ptr(ref_n), L.ptr(in_n), L.stream_ptr()),
                  "mvg_triangulate_project")
        return new_ref, ref2d, proj2d, (r_n, ref_n, in_n)
    with _timed("triangulate"):
      L.check(L.load().mvg_triangulate(L.ptr(r), L.ptr(o), L.ptr(cams), L.ptr(valid), L.ptr(any_valid), L.ptr(new_ref),
                                     L.ptr(ref2d), L.ptr(proj2d), V, B, NQ, J, L.stream_ptr()), "mvg_triangulate")
    return new_ref, ref2d, proj2d


# ------------------------------------------------------------------------- camera packing
def pack_cameras(meta, img_size, device):
    """meta (list[V] of per-view dicts, JointsDataset.py:197-220) -> (V*B, CAM_STRIDE) fp32
    device tensor, image index n = v*B + b.  The crop affine is the closed form of
    get_affine_transform(center, scale, 0, img_size) (lib/utils/transforms.py:72-112) that the
    reference evaluates with numpy/cv2 per view per layer (dq_decoder.py:361-372); here it is
    computed once per forward."""
    V = len(meta)
    B = meta[0]["center"].shape[0]
    rec = np.zeros((V, B, L.CAM_STRIDE), dtype=np.float32)
    wh_all = []
    for v, m in enumerate(meta):
        cam = {k: t.detach().float().cpu().numpy() for k, t in m["camera"].items()
               if k in ("R", "T", "fx", "fy", "cx", "cy", "k", "p")}
        center = m["center"].detach().cpu().numpy().astype(np.float64)            # (B,2)
        scale = m["scale"].detach().cpu().numpy().astype(np.float32)
        inv = m["inv_affine_trans"].detach().cpu().numpy().astype(np.float64)[:, :2, :]
        rec[v, :, 0:9] = cam["R"].reshape(B, 9)
        rec[v, :, 9:12] = cam["T"].reshape(B, 3)
        rec[v, :, 12], rec[v, :, 13] = cam["fx"].reshape(B), cam["fy"].reshape(B)
        rec[v, :, 14], rec[v, :, 15] = cam["cx"].reshape(B), cam["cy"].reshape(B)
        rec[v, :, 16:19] = cam["k"].reshape(B, 3)
        rec[v, :, 19:21] = cam["p"].reshape(B, 2)
        c32 = center.astype(np.float32).astype(np.float64)
        st = (scale * np.float32(200.0)).astype(np.float64)
        dw, dh = float(img_size[0]), float(img_size[1])
        s = np.where(st[:, 0] >= st[:, 1], dw / st[:, 0], dh / st[:, 1])
        rec[v, :, 21], rec[v, :, 25] = s, s
        rec[v, :, 23] = dw * 0.5 - s * c32[:, 0]
        rec[v, :, 26] = dh * 0.5 - s * c32[:, 1]
        rec[v, :, 27:33] = inv.reshape(B, 6)
        wh = center * 2.0
        rec[v, :, 33:35] = wh
        rec[v, :, 35] = wh.max()                                                   # dq_decoder.py:383 (whole-batch max)
        rec[v, :, 36], rec[v, :, 37] = dw, dh
        wh_all.append(wh)
    return torch.from_numpy(rec.reshape(V * B, L.CAM_STRIDE)).to(device)


def uncrop_undistort_jac(ref2d, cams, V, B):
    """ref2d (B, V, Lq, 2) f32 -> (ud (B, V, Lq, 2), jac (B, V, Lq, 2, 2) = d ud / d ref2d) (mvg_uncrop_undistort_jac)."""
    if ref2d.dtype != torch.float32 or tuple(ref2d.shape[:2]) != (B, V) or ref2d.shape[-1] != 2:
        raise RuntimeError("mvg_uncrop_undistort_jac: (B, V, Lq, 2) float32 expected")
    ref2d = ref2d.contiguous()
    Lq = ref2d.shape[2]
    ud = torch.empty_like(ref2d)
    jac = torch.empty((B, V, Lq, 2, 2), dtype=torch.float32, device=ref2d.device)
    L.check(L.load().mvg_uncrop_undistort_jac(L.ptr(ref2d), L.ptr(cams), L.ptr(ud), L.ptr(jac), V, B, Lq, L.stream_ptr()),
            "mvg_uncrop_undistort_jac")
    return ud, jac


def _dlt_args(ud, conf, Pm, valid, J):
    B, V, Lq = conf.shape
    if (ud.dtype != torch.float32 or conf.dtype != torch.float32 or Pm.dtype != torch.float32 or valid.dtype != torch.uint8
            or tuple(ud.shape) != (B, V, Lq, 2) or tuple(Pm.shape) != (B, V, 3, 4) or Lq % J or valid.numel() != B * (Lq // J)):
        raise RuntimeError("mvg_dlt: ud (B,V,Lq,2) / conf (B,V,Lq) / Pm (B,V,3,4) float32 and valid (B,NQ) uint8 expected")
    return B, V, Lq


def dlt_forward(ud, conf, Pm, valid, J):
    """X (B, Lq, 3) of the dense differentiable triangulation (mvg_dlt_forward); zeros for the tokens of queries with valid == 0."""
    B, V, Lq = _dlt_args(ud, conf, Pm, valid, J)
    ud, conf, Pm, valid = ud.contiguous(), conf.contiguous(), Pm.contiguous(), valid.contiguous()
    X = torch.empty((B, Lq, 3), dtype=torch.float32, device=ud.device)
    L.check(L.load().mvg_dlt_forward(L.ptr(ud), L.ptr(conf), L.ptr(Pm), L.ptr(valid), L.ptr(X), V, B, Lq // J, J, L.stream_ptr()),
            "mvg_dlt_forward")
    return X


def dlt_backward(ud, conf, Pm, valid, J, gX):
    """(g_ud, g_conf) of dlt_forward for gX (B, Lq, 3) (mvg_dlt_backward)."""
    B, V, Lq = _dlt_args(ud, conf, Pm, valid, J)
    if gX.dtype != torch.float32 or tuple(gX.shape) != (B, Lq, 3):
        raise RuntimeError("mvg_dlt_backward: gX (B, Lq, 3) float32 expected")
    ud, conf, Pm, valid, gX = ud.contiguous(), conf.contiguous(), Pm.contiguous(), valid.contiguous(), gX.contiguous()
    g_ud = torch.empty_like(ud)
    g_conf = torch.empty_like(conf)
    L.check(L.load().mvg_dlt_backward(L.ptr(ud), L.ptr(conf), L.ptr(Pm), L.ptr(valid), L.ptr(gX), L.ptr(g_ud), L.ptr(g_conf), V, B,
                                      Lq // J, J, L.stream_ptr()), "mvg_dlt_backward")
    return g_ud, g_conf


def sym4_eigh(G):
    """eigen-decomposition of symmetric 4x4 matrices G (..., 4, 4) fp64 on the GPU: (evals (..., 4), evecs (..., 4, 4),
    eigenvectors as columns, no particular order)."""
    if G.dtype != torch.float64 or G.shape[-2:] != (4, 4):
        raise RuntimeError("mvg_sym4_eigh: (..., 4, 4) float64 expected")
    Gc = G.contiguous()
    n = Gc.numel() // 16
    w = torch.empty(Gc.shape[:-1], dtype=torch.float64, device=G.device)
    V = torch.empty_like(Gc)
    L.check(L.load().mvg_sym4_eigh(L.ptr(Gc), L.ptr(w), L.ptr(V), n, L.stream_ptr()), "mvg_sym4_eigh")
    return w, V

