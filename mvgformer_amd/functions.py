"""autograd wrapper of the sampling op -- mirror of
lib/models/ops/functions/deform_func.py:34-65 (DeformFunction)."""
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
