"""Python face of the native op: same names and argument meaning as the reference's pybind
module ``Deformable`` (lib/models/ops/src/vision.cpp:24-27), backed by libmvgformer_hip.so.

    import mvgformer_amd.deformable as Deformable
    out = Deformable.deform_forward(value, spatial_shapes, level_start_index,
                                    sampling_loc, attn_weight, im2col_step)
"""
import contextlib as _contextlib

import torch as _torch

from . import ops as _ops


def _device_of(t):
    """the kernels are enqueued on the current stream of the device that owns the tensors"""
    return _torch.cuda.device(t.device) if t.is_cuda else _contextlib.nullcontext()


def _check_step(value, im2col_step):
    n = value.shape[0]
    step = min(n, int(im2col_step))
    if step <= 0 or n % step != 0:
        # deform_cuda.cu:63
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (n, step))


def deform_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """(N,S,M,D),(L,2)i64,(L,)i64,(N,Lq,M,L,P,2),(N,Lq,M,L,P) -> (N,Lq,M*D).  The reference
    splits the batch in chunks of im2col_step (deform_cuda.cu:61-86); one launch covers the
    whole batch here, the argument is validated and otherwise unused."""
    _check_step(value, im2col_step)
    with _device_of(value):
        return _ops.msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)


def deform_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] (deform_cuda.cu:94-164)."""
    _check_step(value, im2col_step)
    with _device_of(value):
        return list(_ops.msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output))
