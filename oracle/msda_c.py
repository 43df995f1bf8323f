"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes wrapper of oracle/msda_ref.c (build: make -C oracle)."""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(_HERE, "libmsda_ref.so"))
        _lib.msda_forward_ref.argtypes = [C.c_void_p] * 6 + [C.c_int] * 7
        _lib.msda_forward_ref.restype = None
        _lib.msda_backward_ref.argtypes = [C.c_void_p] * 9 + [C.c_int] * 7
        _lib.msda_backward_ref.restype = None
    return _lib


def msda_forward(value, shapes, starts, loc, wgt):
    """CPU float32 tensors; same contract as decoder_ref.msda_forward."""
    value, loc, wgt = (t.detach().float().contiguous() for t in (value, loc, wgt))
    shapes, starts = shapes.long().contiguous(), starts.long().contiguous()
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = torch.empty((N, Lq, M * D), dtype=torch.float32)
    _load().msda_forward_ref(value.data_ptr(), shapes.data_ptr(), starts.data_ptr(), loc.data_ptr(), wgt.data_ptr(),
                             out.data_ptr(), N, S, M, D, L, Lq, P)
    return out


def msda_backward(value, shapes, starts, loc, wgt, grad_out):
    """CPU float32 inputs -> (grad_value, grad_loc, grad_wgt) as float64 tensors (double accumulation)."""
    value, loc, wgt, grad_out = (t.detach().float().contiguous() for t in (value, loc, wgt, grad_out))
    shapes, starts = shapes.long().contiguous(), starts.long().contiguous()
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    gv = torch.empty((N, S, M, D), dtype=torch.float64)
    gl = torch.empty((N, Lq, M, L, P, 2), dtype=torch.float64)
    ga = torch.empty((N, Lq, M, L, P), dtype=torch.float64)
    _load().msda_backward_ref(value.data_ptr(), shapes.data_ptr(), starts.data_ptr(), loc.data_ptr(), wgt.data_ptr(),
                              grad_out.data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), N, S, M, D, L, Lq, P)
    return gv, gl, ga
