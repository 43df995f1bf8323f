"""autograd wrapper of the sampling op -- mirror of
lib/models/ops/functions/deform_func.py:34-65 (DeformFunction)."""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import deformable as DF
from . import ops


class DeformFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        output = DF.deform_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                   attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        # host copies of the level tables for the deterministic backward (its binning workspace is sized on the host): looked
        # up here, where the tensors are the caller's own objects and usually carry the copy already -- not a sync per
        # backward.  Only when a gradient will be asked for: under no_grad (and inside a HIP-graph capture, where the D2H
        # copy of a first look-up is illegal) the forward touches nothing on the host.
        needs = value.is_cuda and any(ctx.needs_input_grad)
        ctx.host_levels = ops.host_levels(value_spatial_shapes, value_level_start_index) if needs else None
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, attn = ctx.saved_tensors
        DF._check_step(value, ctx.im2col_step)
        with DF._device_of(value):
            gv, gl, ga = ops.msda_backward(value, shapes, starts, loc, attn, grad_output.contiguous(), host=ctx.host_levels)
        return gv, None, None, gl, ga, None


# training-path GEMMs: "f32s" = this repo's mvg_linear (fp32 storage, fp32-accurate products on the bf16 matrix pipe from 3-way split
# operands -- csrc/gemm.hip) for forward, dgrad and wgrad; "torch" = nn.functional.linear (rocBLAS fp32: the f32-input MFMA runs at
# 1/16 of the bf16 rate and was 40 % of a training step's kernel time, profiles/r02_train_step.txt)
TRAIN_GEMM = os.environ.get("MVG_TRAIN_GEMM", "f32s")


class LinearF32S(Function):
    """y = act(x W^T + b) with all three GEMMs of its autograd on mvg_linear's split form:
         forward  y  = x W^T              (rows, K) x (N, K)^T, bias + ReLU in the epilogue
         dgrad    dx = dy W               = mvg_linear(dy, W^T as an (K, N) "weight")
         wgrad    dW = dy^T x, db = sum dy = mvg_linear_wgrad_bias_f32 (the reduction runs over the rows: split into slices, the
                                            slices' partials and the bias gradient summed inside the same launch)
       (the Linear + ReLU + Linear chains of lib/models/dq_decoder.py:659-717,763-778, mvp_decoder.py:94-98 and
       lib/models/ops/modules/projattn.py:169,180-181,203 under torch autograd).  Output widths that are not a multiple of 32 (the
       2- and 3-output heads) run with zero rows up to 32 outputs; the padding stays inside this Function."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        N, K = weight.shape
        Np = (N + 31) // 32 * 32
        if Np != N:
            wp = weight.new_zeros((Np, K))
            wp[:N] = weight
            bp = None
            if bias is not None:
                bp = bias.new_zeros((Np,))
                bp[:N] = bias
        else:
            wp, bp = weight.contiguous(), bias
        y = ops.linear(x2, wp, bp, out_dtype=torch.float32, relu=bool(relu))
        if Np != N:
            y = y[:, :N].contiguous()
        ctx.relu = bool(relu)
        ctx.x_shape = x.shape
        ctx.save_for_backward(x2, wp, y if relu else None)
        ctx.has_bias = bias is not None
        ctx.n_out = N
        return y.view(*x.shape[:-1], N)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        x2, wp, y = ctx.saved_tensors
        Np, K = wp.shape
        N = ctx.n_out
        dy = grad_y.reshape(-1, N)
        if ctx.relu:
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)               # dy where y > 0, one launch
        if Np != N:
            dyp = dy.new_zeros((dy.shape[0], Np))
            dyp[:, :N] = dy
            dy = dyp
        elif not dy.is_contiguous():
            dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear(dy, wp.t().contiguous(), None, out_dtype=torch.float32).view(ctx.x_shape)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw, db = ops.linear_wgrad_bias(dy, x2, want_bias=want_b)     # no dy^T / x^T copies: transposed on the way into LDS
            if Np != N:
                dw = dw[:N]
                db = None if db is None else db[:N]
        elif want_b:
            db = dy.sum(0)[:N]
        return dx, dw, db, None


def linear(x, weight, bias=None, relu=False):
    """nn.functional.linear (+ ReLU) for the training path: LinearF32S where mvg_linear's shape rules hold (fp32 on the GPU,
    in_features % 32 == 0 -- every Linear of the decoder), torch elsewhere."""
    N, K = weight.shape
    if TRAIN_GEMM == "f32s" and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and K % 32 == 0 and x.numel() > 0:
        return LinearF32S.apply(x, weight, bias, relu)
    y = torch.nn.functional.linear(x, weight, bias)
    return torch.relu(y) if relu else y


class RefGatherAdd(Function):
    """xin[n, q, l] = bilinear(feat[n], level l, reference point (n, q, l)) + query[n, q]  (projattn.py:134-141,180: grid_sample at the
    clamped reference points, zeros padding, align_corners=False) through mvg_gather_ref; differentiable in `query` only."""

    @staticmethod
    def forward(ctx, query, feat, ref_lvl, levels):
        n, Lq, C = query.shape
        ain = ops.gather_ref(feat, ref_lvl, query.detach(), levels, 1, n)
        ctx.shape = (n, Lq, levels.L, C)
        return ain.view(n, Lq, levels.L, C)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g.reshape(ctx.shape).sum(2), None, None, None
