"""Import the REFERENCE MVGFormer decoder (read-only, /root/reference) on CPU.

Used ONLY by ``make_golden.py`` (and optional cross-check tests) in the build
container: the reference tree does not exist on the GPU box and nothing under
``-m gpu`` / ``bench.py`` / ``smoke()`` may import this module.

Recipe = SURVEY.md section 8(c): stub the modules the reference imports but this
image lacks (Deformable CUDA extension, turtle/tkinter, torchvision, mmcv, wandb,
cv2, easydict, ...), neutralise the hard-coded ``.cuda()`` in
lib/models/dq_decoder.py:1186, and route ``DeformFunction.apply`` to the
reference's own pure-PyTorch twin ``deform_core_pytorch``
(lib/models/ops/functions/deform_func.py:68-99).  No reference file is copied.
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

REF_ROOT = os.environ.get("MVG_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "lib", "models"))


def _affine_from_3pts(src, dst):
    """numpy stand-in for cv2.getAffineTransform: exact solve of the 3-point system."""
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    A = np.concatenate([src, np.ones((3, 1))], 1)
    X = np.linalg.solve(A, dst)  # (3,2)
    return X.T.copy()


_loaded = None


def load_reference():
    """Returns a namespace with the reference symbols on the hot path."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    for p in (os.path.join(REF_ROOT, "lib"), REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []  # behave as a package
        sys.modules[name] = m
        return m

    stub("Deformable")
    stub("turtle", forward=lambda *a, **k: None)
    tv = stub("torchvision", __version__="0.15.0")
    tv.ops = stub("torchvision.ops")
    tv.ops.misc = stub("torchvision.ops.misc")
    tv.transforms = mock.MagicMock()
    sys.modules["torchvision.transforms"] = tv.transforms
    tv.utils = mock.MagicMock()
    sys.modules["torchvision.utils"] = tv.utils
    mm = stub("mmcv")
    mm.runner = stub("mmcv.runner", get_dist_info=lambda: (0, 1))
    stub("wandb")
    cv2 = mock.MagicMock()
    cv2.getAffineTransform = _affine_from_3pts
    sys.modules["cv2"] = cv2
    for name in ("easydict", "json_tricks", "prettytable", "h5py", "matplotlib", "matplotlib.pyplot",
                 "matplotlib.patches", "mpl_toolkits", "mpl_toolkits.mplot3d", "tensorboardX", "smplx",
                 "chumpy", "seaborn"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = mock.MagicMock()

    torch.Tensor.cuda = lambda self, *a, **k: self  # dq_decoder.py:1186 hard-codes .cuda()

    import models.dq_decoder as dqd  # noqa: E402  (reference module)
    import lib.models.ops.modules.projattn as pa  # noqa: E402
    from lib.models.ops.functions.deform_func import deform_core_pytorch  # noqa: E402
    import lib.utils.cameras as cameras  # noqa: E402
    from mvn.utils import multiview  # noqa: E402

    class _CPUDeform:
        @staticmethod
        def apply(value, shapes, starts, loc, w, step):
            return deform_core_pytorch(value, shapes, loc, w)

    pa.DeformFunction = _CPUDeform
    _loaded = types.SimpleNamespace(dq_decoder=dqd, projattn=pa, deform_core_pytorch=deform_core_pytorch,
                                    cameras=cameras, multiview=multiview,
                                    DQDecoderLayer=dqd.DQDecoderLayer, DQDecoder=dqd.DQDecoder,
                                    ProjAttn=pa.ProjAttn)
    return _loaded


def build_reference_decoder(case, threshold_unused=None):
    """Instantiate the reference DQDecoder for a synthetic ``case`` (mvgformer_amd.synthetic.build_case)
    with the panoptic YAML hyper-parameters and load the case's weights."""
    ref = load_reference()
    from mvgformer_amd.synthetic import decoder_cfg, to_torch_state
    layer = ref.DQDecoderLayer(
        list(case.space_size), list(case.space_center), list(case.img_size), 3,
        256, 1024, 0.1, "relu", 1, 8, 8, True, "cat_proj", case.V,
        "ablation_not_use_rayconv", "MLP", False, True, "threshold",
        visualization_jump_num=-1, bayesian_update=False, triangulation_method="linalg",
        filter_query=True, num_joints=15)
    dec = ref.DQDecoder(decoder_cfg(case.space_size, case.space_center), layer, case.layers, True)
    missing, unexpected = dec.load_state_dict(to_torch_state(case.weights), strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith(("grid_size", "grid_center")) for k in missing), missing
    dec.eval()
    return dec
