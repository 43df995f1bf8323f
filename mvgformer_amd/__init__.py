"""mvgformer_amd -- MI355X-native implementation of the MVGFormer decoder hot path.

Host-side mirror of the reference's operator interface (same names / signatures):
  Deformable op   : mvgformer_amd.deformable.deform_forward / deform_backward
  DeformFunction  : mvgformer_amd.functions.DeformFunction
  ProjAttn        : mvgformer_amd.projattn.ProjAttn            (alias MSDeformAttn)
  DQDecoderLayer  : mvgformer_amd.decoder.DQDecoderLayer       (alias MultiViewDecoderLayer)
  DQDecoder       : mvgformer_amd.decoder.DQDecoder
All compute goes through libmvgformer_hip.so (include/mvg_decoder.h); there is no CPU path.
"""
from .decoder import MLP, DQDecoder, DQDecoderLayer, MultiViewDecoder, MultiViewDecoderLayer, offset_net  # noqa: F401
from .functions import DeformFunction  # noqa: F401
from .projattn import MSDeformAttn, ProjAttn  # noqa: F401

__version__ = "0.1.0"
