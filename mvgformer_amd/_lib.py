"""ctypes binding of libmvgformer_hip.so (the C ABI declared in include/mvg_decoder.h).

There is NO CPU fallback: if the shared library is missing or a call fails, the product
path raises.  torch is used only to own device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVG_LIB: another build of the same library (measurement variants of tools/ab_*.sh); the product never sets it
LIB_PATH = os.environ.get("MVG_LIB") or os.path.join(_HERE, "libmvgformer_hip.so")

MVG_F32, MVG_BF16 = 0, 1
CAM_STRIDE = 48

_vp, _i, _f = C.c_void_p, C.c_int, C.c_float

# name -> argtypes (restype is always int unless noted); must list EVERY symbol of the header
SIGNATURES = {
    "mvg_device_info": [C.c_char_p, _i, C.POINTER(_i)],
    "mvg_set_tuning": [C.c_char_p, _i],
    "mvg_msda_forward_f32": [_vp] * 6 + [_i] * 7 + [_vp],
    "mvg_msda_forward_bf16": [_vp] * 6 + [_i] * 7 + [_vp],
    "mvg_msda_backward_f32": [_vp] * 9 + [_i] * 7 + [_vp],
    "mvg_msda_backward_det_workspace": [_i] * 7 + [_vp],
    "mvg_msda_backward_det_f32": [_vp] * 9 + [_i] * 7 + [_vp, C.c_size_t, _vp],
    "mvg_msda_forward_f64": [_vp] * 6 + [_i] * 7 + [_vp],
    "mvg_msda_backward_f64": [_vp] * 9 + [_i] * 7 + [_vp],
    "mvg_pack_pyramid": [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp],
    "mvg_project": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mvg_gather_ref": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mvg_linear": [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp],
    "mvg_dlt_forward": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvg_dlt_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvg_linear_wgrad_bias_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvg_linear_ordered": [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "mvg_linear_sum": [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp],
    "mvg_msda_fused": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvg_msda_gfused_f32": [_vp] * 9 + [_i] * 5 + [_vp],
    "mvg_chain_attn_pose": [_vp] * 14 + [_i, _vp],
    "mvg_chain_update_ffn_class": [_vp, _i] + [_vp] * 13 + [_f] + [_vp] * 9 + [_i] * 5 + [_vp, _vp],
    "mvg_chain_update_ffn_class_f32h": [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvg_chain_attn_pose_f32h": [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "mvg_pyramid_f32h": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, C.c_int64, _i, _vp],
    "mvg_pyramid_group_ws": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "mvg_msda_gsamp": [_vp] * 9 + [_i] * 5 + [_vp],
    "mvg_bin_pairs": [_vp] * 3 + [_i] + [_vp] + [_i] * 2 + [_vp, C.c_size_t, _vp],
    "mvg_bin_pairs_workspace": [_i, _i],
    "mvg_mean_views": [_vp, _i, _vp, _i, _i, _i, _vp],
    "mvg_add_layernorm": [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "mvg_class_head": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvg_rowdot3": [_vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "mvg_triangulate": [_vp] * 8 + [_i] * 4 + [_vp],
    "mvg_triangulate_project": [_vp] * 8 + [_i] * 4 + [_vp, _i] + [_vp] * 4,
    "mvg_uncrop_undistort_jac": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mvg_sym4_eigh": [_vp] * 3 + [C.c_long, _vp],
}

_lib = None
TUNING = {}     # knob values set through mvg_set_tuning in this process (library defaults are not listed)


class MvgError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MvgError(
            "libmvgformer_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C mvgformer_amd/csrc` (there is no CPU fallback for the decoder hot path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.mvg_version.restype = C.c_char_p
    lib.mvg_bin_pairs_workspace.restype = C.c_size_t
    lib.mvg_msda_backward_det_workspace.restype = C.c_size_t
    lib.mvg_version.argtypes = []
    # every knob change goes through this wrapper, so that host-side caches that depend on a knob (DQDecoderLayer's rows of
    # all-masked tiles: computed by the GEMM form that is active) can key on its value: TUNING[key] = last value set
    raw_set = lib.mvg_set_tuning

    def set_tuning(key, value):
        rc = raw_set(key, value)
        if rc == 0:
            TUNING[key.decode() if isinstance(key, bytes) else str(key)] = int(value)
        return rc
    lib.mvg_set_tuning = set_tuning
    _lib = lib
    # A/B knobs for measurements: MVG_TUNE="chain_rm=128,gsamp_threads=256"
    for item in filter(None, os.environ.get("MVG_TUNE", "").split(",")):
        k, v = item.split("=")
        check(lib.mvg_set_tuning(k.strip().encode(), int(v)), "mvg_set_tuning(%s)" % item)
    return lib


def check(code, what):
    if code != 0:
        raise MvgError("%s failed with code %d" % (what, code))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def dtype_code(dt):
    if dt == torch.float32:
        return MVG_F32
    if dt == torch.bfloat16:
        return MVG_BF16
    raise MvgError("unsupported dtype %s (float32 / bfloat16 only)" % dt)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            # same message as the reference's CPU stub (lib/models/ops/src/deform.h:49)
            raise RuntimeError("Not implemented on the CPU")
        if t is not None and t.device.index != torch.cuda.current_device():
            # stream_ptr() hands the kernels the CURRENT device's stream: launching on another device's tensors
            # would run on the wrong GPU.  The module entry points (DQDecoder / DQDecoderLayer / ProjAttn / deformable)
            # switch to their tensors' device themselves; a direct ops.* caller has to.
            raise MvgError("tensor on %s but the current device is cuda:%d -- wrap the call in "
                           "torch.cuda.device(tensor.device)" % (t.device, torch.cuda.current_device()))


def device_info():
    lib = load()
    buf = C.create_string_buffer(64)
    cus = C.c_int(0)
    rc = lib.mvg_device_info(buf, 64, C.byref(cus))
    if rc != 0:
        raise MvgError("no usable AMD GPU (mvg_device_info -> %d)" % rc)
    return buf.value.decode(), cus.value
