"""autograd wrapper of the sampling op -- mirror of
lib/models/ops/functions/deform_func.py:34-65 (DeformFunction)."""
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import deformable as DF


class DeformFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        output = DF.deform_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                   attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, attn = ctx.saved_tensors
        gv, gl, ga = DF.deform_backward(value, shapes, starts, loc, attn, grad_output.contiguous(), ctx.im2col_step)
        return gv, None, None, gl, ga, None
