"""DQDecoderLayer / DQDecoder -- the MVGFormer decoder with the reference's class surface
(lib/models/dq_decoder.py:248-328,850-1045,1101-1172; base classes lib/models/mvp_decoder.py:49-98,
267-282; MLP lib/models/multi_view_pose_transformer.py:81-102), running on libmvgformer_hip.so.

What differs from the reference by design (MI355X-first, results identical):
  * the V per-view ProjAttn calls of generate_features (dq_decoder.py:553-591) are ONE batched
    launch sequence over all (view, batch) images;
  * camera constants and the crop affine are packed once per forward instead of going through
    numpy/cv2 per view per layer (dq_decoder.py:361-372: a host sync each time);
  * no torch.where / bincount / sort / Python loops (dq_decoder.py:596-656,929-967): every
    query is processed and the validity mask is applied in the final scatter kernel -- every step
    is per-query, so the outputs are the same and the layer never synchronises with the host;
  * the per-valid-query Python loop of 15-matrix SVD launches (multiview.py:262-266) is one
    kernel (thread per joint) that solves the 4x4 normal equations of the DLT rows in fp64.
"""
from __future__ import annotations

import copy
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .projattn import ProjAttn, WeightCache



class MLP(nn.Module):
    """multi_view_pose_transformer.py:81-102"""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = F.relu(layer(x)) if i < self.num_layers - 1 else layer(x)
        return x


class offset_net(nn.Module):
    """dq_decoder.py:97-111: (dx, dy) pixel offsets + a view-confidence logit."""

    def __init__(self, in_dim, hid_dim, layer_num):
        super().__init__()
        self.MLP = MLP(in_dim, hid_dim, 3, layer_num)

    def forward(self, feature):
        out = self.MLP(feature)
        return out[..., :2], out[..., -1]


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError(F"activation should be relu/gelu, not {activation}.")


class DecoderContext:
    """Per-forward, layer-independent state: level table + packed cameras (host side, built from
    ``meta``) and the packed channels-last pyramid (device side, built from ``src_views``).

    ``prepare`` touches host memory (it reads the small camera tensors of ``meta``), ``pack`` only
    enqueues kernels -- so a HIP-graph capture can call ``pack`` with a prepared context."""

    def __init__(self, levels, cams, V, B, dtype):
        self.levels, self.cams, self.V, self.B, self.dtype = levels, cams, V, B, dtype
        self.feat = None
        self._buffer = None    # producer-owned packed pyramid (pyramid_buffers)

    @classmethod
    def prepare(cls, spatial_shapes, level_start_index, meta, img_size, dtype, batch_size, device):
        levels = ops.Levels(spatial_shapes, level_start_index)
        cams = ops.pack_cameras(meta, img_size, device)
        V = cams.shape[0] // batch_size
        return cls(levels, cams, V, batch_size, dtype)

    def pack(self, src_views):
        if src_views[0].shape[0] != self.V * self.B:
            raise RuntimeError("meta describes %d images, src_views hold %d" % (self.V * self.B, src_views[0].shape[0]))
        out = self._buffer if (self._buffer is not None and self._buffer.shape[0] == src_views[0].shape[0]) else None
        self.feat = ops.pack_pyramid(src_views, self.levels, self.dtype, out=out)
        return self

    def pyramid_buffers(self, channels=256, device=None):
        """Producer hand-off: allocates the packed pyramid and returns its L levels as
        (V*B, C, H_l, W_l) channels-last views in the compute dtype.  A backbone that writes its feature maps
        into these tensors (or hands them back as ``src_views``) makes ``pack`` free: no 206-MB NCHW->channels-last
        pass (projattn.py:160's cat + permute) per forward."""
        dev = device if device is not None else self.cams.device
        self._buffer = torch.empty((self.V * self.B, self.levels.S, channels), dtype=self.dtype, device=dev)
        return ops.pyramid_level_views(self._buffer, self.levels)

    @classmethod
    def build(cls, src_views, spatial_shapes, level_start_index, meta, img_size, dtype, batch_size):
        return cls.prepare(spatial_shapes, level_start_index, meta, img_size, dtype, batch_size,
                           src_views[0].device).pack(src_views)


class MvPDecoderLayer(nn.Module):
    """helpers shared with the MvP base class (mvp_decoder.py:49-105)."""

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        tgt2 = self.linear2(self.dropout3(self.activation(self.linear1(tgt))))
        tgt = tgt + self.dropout4(tgt2)
        return self.norm3(tgt)

    def norm2absolute(self, norm_coords):
        device = norm_coords.device
        grid_size = self.grid_size.to(device=device)
        grid_center = self.grid_center.to(device=device)
        return norm_coords * grid_size + grid_center - grid_size / 2.0


class DQDecoderLayer(MvPDecoderLayer):
    def __init__(self, space_size, space_center, img_size, pose_embed_layer, d_model=256, d_ffn=1024,
                 dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4,
                 detach_refpoints_cameraprj=True, fuse_view_feats="mean", n_views=5,
                 projattn_posembed_mode="use_rayconv", feature_update_method="MLP",
                 init_self_attention=False, open_forward_ffn=False, query_filter_method="threshold",
                 visualization_jump_num=200, bayesian_update=False, triangulation_method="linalg",
                 filter_query=True, num_joints=15):
        super().__init__()
        # --- same sub-module names / shapes as the reference => same state-dict keys
        self.proj_attn = ProjAttn(d_model, n_levels, n_heads, n_points, projattn_posembed_mode)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.feature_update_mlp = nn.Linear(d_model, d_model)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.activation_name = activation
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        self.grid_size = torch.tensor(space_size)
        self.grid_center = torch.tensor(space_center)
        self.img_size = img_size
        self.detach_refpoints_cameraprj = detach_refpoints_cameraprj
        self.fuse_view_feats = fuse_view_feats
        self.pose_embed = offset_net(d_model, d_model, pose_embed_layer)
        self.softmax_conf = nn.Softmax(dim=0)
        self.open_bayesian_update = bayesian_update
        if self.open_bayesian_update:
            self.bayesian_conf = nn.Linear(d_model, 1)
        self.use_confidences = False
        self.class_embed = nn.Linear(d_model, 2)
        self.num_joints = num_joints
        self.feature_update_method = feature_update_method
        self.init_self_attention = init_self_attention
        self.open_forward_ffn = open_forward_ffn
        self.query_filter_method = query_filter_method
        self.visualization_jump_num = visualization_jump_num
        self.triangulation_method = triangulation_method
        self.filter_query = filter_query
        self.d_model = d_model
        self.compute_dtype = torch.float32
        self.use_fused_chains = True    # bf16 inference: LDS-resident Linear chains (csrc/chain.hip)
        # fp32 inference: the same two chains on pre-split operands (csrc/f32s.hip); MVG_F32_FUSED=0: one launch per GEMM / row op
        self.use_fused_chains_f32 = os.environ.get("MVG_F32_FUSED", "1") != "0"
        self.fuse_boundary = True       # the triangulation launch also projects the new points for the next layer
        # fp32 path: output projection + pose MLP skip the tiles whose pairs are all outside their image (round 3)
        self.skip_masked_f32 = True
        self._wc = WeightCache()
        self._ctx = None   # set by DQDecoder.forward so the pyramid / cameras are packed once
        self._tgt_out = None   # set by DQDecoder.forward: this layer's slice of the stacked hidden states
        self._flag = None      # set by DQDecoder.forward: this layer's zeroed any-valid flag (int32[1])
        self._geo_out = None   # set by DQDecoder.forward: this layer's slices of the stacked 3D / 2D outputs
        self._next_layer = None   # set by DQDecoder.forward: the layer that consumes this layer's output
        self._xw_in = None        # set by the previous layer: this layer's query term (B*Lq,192) f32
        self._after_chain_b = None    # set by DQDecoder.launch_pyramid_projections (just-in-time schedule): issues the NEXT layer's pyramid products
        self._proj_in = None      # set by the previous layer's triangulation launch: (new_ref, (r, ref_lvl, inside)) of this layer
        # query-sharded runs (mvgformer_amd.dist): callable(any_valid int32[1]) that makes the
        # "no query valid anywhere -> force query (0,0)" rule (dq_decoder.py:620-623) global
        self._any_valid_hook = None

    # ------------------------------------------------------------------------------ config
    def set_compute_dtype(self, dtype):
        """torch.float32 (reference arithmetic) or torch.bfloat16 (bf16 storage + bf16 MFMA with
        fp32 accumulation; geometry stays fp32/fp64)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32 or bfloat16")
        self.compute_dtype = dtype
        self.proj_attn.compute_dtype = dtype
        return self

    def _check_supported(self):
        bad = None
        if self.feature_update_method != "MLP":
            bad = "feature_update_method=%r" % self.feature_update_method
        elif self.init_self_attention:
            bad = "init_self_attention=True"
        elif self.open_bayesian_update:
            bad = "bayesian_update=True"
        elif self.triangulation_method not in ("linalg", "batch", "default", "cpu"):
            bad = "triangulation_method=%r" % self.triangulation_method
        elif self.filter_query and self.query_filter_method not in ("threshold", "all"):
            bad = "query_filter_method=%r" % self.query_filter_method
        elif self.activation_name != "relu":
            bad = "activation=%r" % self.activation_name
        elif len(self.pose_embed.MLP.layers) < 1 or self.pose_embed.MLP.layers[-1].out_features != 3:
            bad = "pose_embed shape"
        elif not self.proj_attn.fused_supported(3):
            bad = "ProjAttn geometry (need d_model=256, nhead=8, dec_n_points=8, num_feature_levels=1)"
        if bad:
            # cross-query variants couple the queries and are not built
            raise NotImplementedError("DQDecoderLayer: %s is outside the built hot path (the shipped YAMLs use "
                                      "feature_update_method='MLP', init_self_attention=False, "
                                      "triangulation_method='linalg', bayesian_update=False)" % bad)

    def _w(self, key, params, dtype, build=None):
        return self._wc.get(key, params, dtype, build)

    # cached operands of the fused chains (csrc/chain.hip), in the order the ops take them
    def _chain_a_weights(self, dt):
        f32 = torch.float32
        pose_layers = self.pose_embed.MLP.layers
        sw = lambda w: ops.swizzle_weight(w.to(dt))
        wts = (self._w("Wp_sw", (self.proj_attn.output_proj.weight,), dt, sw),
               self._w("bp", (self.proj_attn.output_proj.bias,), f32),
               self._w("Wpe0_sw", (pose_layers[0].weight,), dt, sw), self._w("bpe0", (pose_layers[0].bias,), f32),
               self._w("Wpe1_sw", (pose_layers[1].weight,), dt, sw), self._w("bpe1", (pose_layers[1].bias,), f32),
               self._w("Wpe_last", (pose_layers[2].weight,), f32), self._w("bpe_last", (pose_layers[2].bias,), f32))
        pose_params = tuple(p for lin in pose_layers for p in (lin.weight, lin.bias))
        # o of a masked row, computed by the kernel that is going to use it (the variants reduce the last pose layer in
        # different orders; a masked row inside a mixed tile and one in an all-masked tile must agree bit for bit)
        o_masked = self._w("o_masked", pose_params, f32, lambda *_: ops.chain_masked_row_output(*wts))
        return wts, o_masked

    def _pose_masked_rows(self, dt):
        """what the hidden pose-MLP layers give for a pair outside its image (attn row = 0): layer i's output row, computed by the
        GEMM kernel itself on a one-row input so that rows of all-masked tiles (broadcast) and masked rows inside computed tiles
        agree bit for bit.  Cached per weight version."""
        f32 = torch.float32
        layers = self.pose_embed.MLP.layers[:-1]
        params = tuple(p for lin in layers for p in (lin.weight, lin.bias))

        def build(*_):
            rows, cur = [], torch.zeros((1, layers[0].weight.shape[1]), dtype=f32, device=layers[0].weight.device)
            for lin in layers:
                cur = ops.linear(cur, lin.weight.detach().to(f32).contiguous(), lin.bias.detach().to(f32).contiguous(), relu=True)
                rows.append(cur[0])
            return torch.stack(rows)
        # keyed on the active GEMM form as well: the rows must come from the form that computes the unmasked tiles
        from . import _lib
        return self._w("pose_masked_rows/f32_split=%d" % _lib.TUNING.get("f32_split", 1), params, f32, build)

    def _chain_b_weights(self, dt):
        f32 = torch.float32
        sw = lambda w: ops.swizzle_weight(w.to(dt))
        ffn = self.open_forward_ffn
        return (self._w("Wu_sw", (self.feature_update_mlp.weight,), dt, sw), self._w("bu", (self.feature_update_mlp.bias,), f32),
                self._w("g2", (self.norm2.weight,), f32), self._w("b2", (self.norm2.bias,), f32),
                self._w("W1_sw", (self.linear1.weight,), dt, sw) if ffn else None,
                self._w("b1", (self.linear1.bias,), f32) if ffn else None,
                self._w("W2_sw", (self.linear2.weight,), dt, sw) if ffn else None,
                self._w("bb2", (self.linear2.bias,), f32) if ffn else None,
                self._w("g3", (self.norm3.weight,), f32) if ffn else None,
                self._w("b3", (self.norm3.bias,), f32) if ffn else None,
                self._w("Wc", (self.class_embed.weight,), f32), self._w("bc", (self.class_embed.bias,), f32))

    def _chain_a_weights_f32h(self):
        """operands of ops.chain_attn_pose_f32h (two-part fp16 planes + their scales) and the masked-row constant of that kernel"""
        f32, f16 = torch.float32, torch.float16
        pose_layers = self.pose_embed.MLP.layers
        sp = ops.split_swizzle_weight_h2
        Wp, swp = self._w("Wp_f32h", (self.proj_attn.output_proj.weight,), f16, sp)
        W0, sw0 = self._w("Wpe0_f32h", (pose_layers[0].weight,), f16, sp)
        W1, sw1 = self._w("Wpe1_f32h", (pose_layers[1].weight,), f16, sp)
        wts = (Wp, swp, self._w("bp", (self.proj_attn.output_proj.bias,), f32), W0, sw0, self._w("bpe0", (pose_layers[0].bias,), f32),
               W1, sw1, self._w("bpe1", (pose_layers[1].bias,), f32),
               self._w("Wpe_last", (pose_layers[2].weight,), f32), self._w("bpe_last", (pose_layers[2].bias,), f32))
        pose_params = tuple(p for lin in pose_layers for p in (lin.weight, lin.bias)) + (self.proj_attn.output_proj.bias,)
        o_masked = self._w("o_masked_f32h", pose_params, f32, lambda *_: ops.chain_masked_row_output_f32h(*wts))
        return wts, o_masked

    def _chain_b_weights_f32h(self):
        """operands of ops.chain_update_ffn_class_f32h, in its argument order"""
        f32, f16 = torch.float32, torch.float16
        sp = ops.split_swizzle_weight_h2
        ffn = self.open_forward_ffn
        Wu, su = self._w("Wu_f32h", (self.feature_update_mlp.weight,), f16, sp)
        W1, s1 = self._w("W1_f32h", (self.linear1.weight,), f16, sp) if ffn else (None, 0)
        W2, s2 = self._w("W2_f32h", (self.linear2.weight,), f16, sp) if ffn else (None, 0)
        return (Wu, su, self._w("bu", (self.feature_update_mlp.bias,), f32),
                self._w("g2", (self.norm2.weight,), f32), self._w("b2", (self.norm2.bias,), f32),
                W1, s1, self._w("b1", (self.linear1.bias,), f32) if ffn else None,
                W2, s2, self._w("bb2", (self.linear2.bias,), f32) if ffn else None,
                self._w("g3", (self.norm3.weight,), f32) if ffn else None,
                self._w("b3", (self.norm3.bias,), f32) if ffn else None,
                self._w("Wc", (self.class_embed.weight,), f32), self._w("bc", (self.class_embed.bias,), f32))

    def _fuses_chains_f32(self, dt, Lq=None, levels=None):
        """(chain A, chain B) of the fp32 path run as the fused f32s kernels (csrc/f32s.hip)"""
        pose_layers = self.pose_embed.MLP.layers
        from . import _lib
        # the library's f32_split knob (bench.py --f32-gemm exact) selects the reference-arithmetic GEMMs: the fused kernels, whose
        # products are split by construction, stand down with it
        ok = (dt == torch.float32 and self.use_fused_chains_f32 and self.d_model == 256 and _lib.TUNING.get("f32_split", 1) != 0)
        fuse_a = (ok and len(pose_layers) == 3 and pose_layers[0].out_features == 256 and pose_layers[1].out_features == 256
                  and pose_layers[0].in_features == 256 and self.proj_attn.f32_fused_active()
                  and (levels is None or self.proj_attn.f32_g_form(Lq, levels.L, levels.S)))
        fuse_b = (ok and self.num_joints <= 32 and (not self.open_forward_ffn or (self.linear1.out_features == 1024 and
                                                                                   self.linear1.in_features == 256)))
        return fuse_a, fuse_b

    def _fuses_chains(self, dt):
        """(chain A, chain B) run as the fused LDS-resident kernels for this configuration."""
        pose_layers = self.pose_embed.MLP.layers
        fuse_a = (dt == torch.bfloat16 and self.use_fused_chains and len(pose_layers) == 3 and
                  pose_layers[0].out_features == 256 and pose_layers[1].out_features == 256)
        fuse_b = (dt == torch.bfloat16 and self.use_fused_chains and self.d_model == 256 and self.num_joints <= 64 and
                  (not self.open_forward_ffn or self.linear1.out_features == 1024))
        return fuse_a, fuse_b

    def prepare_caches(self, dtype=None):
        """Build every cached operand of the inference path on the CURRENT stream, outside any graph capture (one of
        them, o_masked, launches a kernel).  An entry created inside a capture would live in the graph's private pool
        and stay unwritten until the first replay -- WeightCache.get refuses to do that.  DQDecoder calls this before
        it forks its side stream, the sharded runners (mvgformer_amd.dist) before they capture."""
        dt = dtype or self.compute_dtype
        if self.proj_attn.uses_fast_path(dt):
            self.proj_attn.prepare_fast_path(dt)
        else:
            self.proj_attn.weights(dt)
            if dt == torch.float32 and self.skip_masked_f32:
                self._pose_masked_rows(dt)
                self.proj_attn._wc.get("zero_row", (self.proj_attn.output_proj.bias,), torch.float32, lambda b: torch.zeros_like(b))
            if dt == torch.float32 and self.proj_attn.g_sampling_f32 is not False and \
                    self.proj_attn.sampling_offsets.out_features + self.proj_attn.attention_weights.out_features == 192:
                self.proj_attn._fast_query_weights(dt)      # the fp32 G-sampling branch of native_sample (Woa_perm / boa_perm)
                if self.proj_attn.f32_fused_active() and self.proj_attn.rayconv.weight.shape == (256, 256):
                    self.proj_attn.query_term_weights_f32h()
                    self.proj_attn.pyramid_planes_f32h()
            fa32, fb32 = self._fuses_chains_f32(dt)
            if fa32:
                self._chain_a_weights_f32h()
            if fb32:
                # (the query-term operand indexes rows 0..191 of [offsets; logits]: only with the G form's geometry, like every
                # other operand of that form)
                g_geometry = (self.proj_attn.sampling_offsets.out_features + self.proj_attn.attention_weights.out_features == 192
                              and self.proj_attn.rayconv.weight.shape == (256, 256))
                self._chain_b_weights_f32h()
                if g_geometry:
                    self.proj_attn.query_term_weights_f32h()
        fuse_a, fuse_b = self._fuses_chains(dt)
        if fuse_a:
            self._chain_a_weights(dt)
        if fuse_b:
            self._chain_b_weights(dt)
        return self

    # ----------------------------------------------------------------------------- forward
    def forward(self, tgt, query_pos, reference_points, src_views, src_spatial_shapes, level_start_index, meta,
                src_padding_mask=None, rgb_views=None, output_dir="./", frame_id=None, indices=None,
                threshold=0.5, indices_all=None):
        """dq_decoder.py:850-1045.  tgt, query_pos (B,Lq,C); reference_points (B,Lq,1,3) or
        (B,Lq,3) in mm; src_views list of L (V*B,C,H,W) maps, view-major; meta list[V] of dicts.
        Returns (tgt_update (B,Lq,C), new_reference_points (B,Lq,3), refined_2d_abs (B,V,Lq,2),
        projs_2d_abs (B,V,Lq,2), class_prob (B,NQ,2)); 3D / 2D outputs are zero for queries that
        do not pass the filter."""
        if not tgt.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        self._check_supported()
        with torch.cuda.device(tgt.device):     # kernels go to the current stream of the TENSORS' device
            return self._forward(tgt, query_pos, reference_points, src_views, src_spatial_shapes, level_start_index, meta,
                                 indices, threshold)

    def _forward(self, tgt, query_pos, reference_points, src_views, src_spatial_shapes, level_start_index, meta, indices,
                 threshold):
        # The native path has no dropout: it is the eval()-mode layer.  Under autograd, and in train() mode with active
        # dropout (validation inside a training loop without .eval(), MC-dropout), the differentiable torch path runs,
        # which applies dropout2/3/4 like the reference (dq_decoder.py:776, mvp_decoder.py:94-98).
        dropping = self.training and max(self.dropout2.p, self.dropout3.p, self.dropout4.p) > 0
        if dropping or (torch.is_grad_enabled() and (tgt.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return self.forward_autograd(tgt, query_pos, reference_points, src_views, src_spatial_shapes,
                                         level_start_index, meta, indices, threshold)
        ctx = self._ctx
        if ctx is None:
            ctx = DecoderContext.build(src_views, src_spatial_shapes, level_start_index, meta, self.img_size,
                                       self.compute_dtype, tgt.shape[0])
        st = self.forward_features(tgt, query_pos, reference_points, ctx, threshold, indices)
        if self._any_valid_hook is not None:
            self._any_valid_hook(st["any_valid"])
        return self.forward_triangulate(st, ctx)

    def forward_autograd(self, tgt, query_pos, reference_points, src_views, src_spatial_shapes, level_start_index, meta,
                         indices=None, threshold=0.5):
        """Training path: the same layer as differentiable torch ops (fp32) with the HIP sampling op
        (DeformFunction forward + backward kernels) inside ProjAttn -- what run/train_3d.py needs
        (SURVEY.md section 8 f2).  Dense compute + masking instead of the reference's gather/pad/scatter
        (every step is per-query, so values and gradients of the kept queries are the same)."""
        from . import geometry_torch as G
        B, Lq, C = tgt.shape
        J = self.num_joints
        NQ = Lq // J
        V = len(meta)
        dev = tgt.device
        img = torch.tensor(self.img_size, dtype=torch.float32, device=dev)
        X = reference_points.reshape(B, Lq, 3)
        if self.detach_refpoints_cameraprj:
            X = X.detach()                                                    # dq_decoder.py:338-339
        WH = src_spatial_shapes.flip(-1).float()
        x = self.with_pos_embed(tgt, query_pos)
        # Per-forward constants of the geometry (packed camera records, level table, projection matrices): built once and kept on the
        # DecoderContext when the layer runs inside DQDecoder.forward -- as torch ops per layer they were ~40 launches each.
        ctx = self._ctx
        tc = getattr(ctx, "_train_cache", None) if ctx is not None else None
        if tc is not None and tc["key"] != (ctx.cams.data_ptr(), ctx.cams._version):
            tc = None           # the context was given new camera records: projection matrices and records are rebuilt
        if tc is None:
            cams = ctx.cams if ctx is not None else ops.pack_cameras(meta, self.img_size, dev)
            tc = dict(cams=cams, levels=ctx.levels if ctx is not None else ops.Levels(src_spatial_shapes, level_start_index),
                      Pm=G.proj_matrices_from_records(cams, V, B), key=(cams.data_ptr(), cams._version))
            if ctx is not None:
                ctx._train_cache = tc
        # all V views as ONE batch of V*B images (image n = v*B + b, the order of src_views): one projection, one ProjAttn
        # call and one pose MLP instead of V of each -- the training step is launch-bound (5 800 launches per step at cfg-2)
        if not X.requires_grad:
            # the reference points do not carry a gradient into the projection (detach_refpoints_cameraprj, every shipped YAML):
            # the inference kernel projects them (one launch, same arithmetic: tests/test_hip_parity.py)
            r_all, ref_lvl, inside_u8 = ops.project(X.float().contiguous(), tc["cams"], tc["levels"], V, B)
            inside_all = inside_u8.bool()
        else:
            cam_all = {k: torch.cat([meta[v]["camera"][k].to(dev) for v in range(V)], 0)
                       for k in ("R", "T", "fx", "fy", "cx", "cy", "k", "p")}
            center_all = torch.cat([meta[v]["center"].to(dev) for v in range(V)], 0)
            A_crop = torch.cat([G.crop_affine(meta[v]["center"], meta[v]["scale"], self.img_size, dev) for v in range(V)], 0)
            r_all, inside_all = G.project_points(X.repeat(V, 1, 1), cam_all, center_all, A_crop, self.img_size, views=V)
            ref_lvl = r_all.unsqueeze(2) * WH / (WH - 1)                           # dq_decoder.py:570-573
        self.proj_attn._packed_feat = ctx.feat if (ctx is not None and ctx.feat is not None) else None
        try:
            a_all = self.proj_attn(x.repeat(V, 1, 1), ref_lvl, src_views, None, src_spatial_shapes, level_start_index)
        finally:
            self.proj_attn._packed_feat = None
        a_all = inside_all.unsqueeze(-1).to(a_all.dtype) * a_all                   # dq_decoder.py:585-586
        from .functions import linear as lin
        mean = a_all.view(V, B, Lq, C).mean(0)
        tgt_update = self.norm2(tgt + self.dropout2(lin(mean, self.feature_update_mlp.weight, self.feature_update_mlp.bias)))
        if self.open_forward_ffn:           # forward_ffn (mvp_decoder.py:94-98) with the two GEMMs on mvg_linear
            h1 = lin(tgt_update, self.linear1.weight, self.linear1.bias, relu=self.activation_name == "relu")
            if self.activation_name != "relu":
                h1 = self.activation(h1)
            tgt_update = self.norm3(tgt_update + self.dropout4(lin(self.dropout3(h1), self.linear2.weight, self.linear2.bias)))
        prob = lin(tgt_update, self.class_embed.weight, self.class_embed.bias).view(B, NQ, J, 2).sigmoid().mean(2)
        if not self.filter_query or self.query_filter_method == "all":
            valid = torch.ones((B, NQ), dtype=torch.bool, device=dev)
        elif indices is not None:
            valid = torch.zeros((B, NQ), dtype=torch.bool, device=dev)
            for b, q in enumerate(indices):
                valid[b, torch.as_tensor(q, dtype=torch.long, device=dev)] = True
        else:
            valid = prob[..., 1] > threshold
        # no query kept -> the first one is (dq_decoder.py:620-623), decided on the device (no host sync in the step)
        valid[0, 0] |= ~valid.any()
        hp = a_all                                                              # offset_net (dq_decoder.py:97-111)
        pl = self.pose_embed.MLP.layers
        for i, layer_ in enumerate(pl):
            hp = lin(hp, layer_.weight, layer_.bias, relu=i < len(pl) - 1)
        off, cl = hp.split([2, 1], -1)                                          # (V*B,Lq,2), (V*B,Lq,1)
        ref2d = ((r_all + off / img) * img).view(V, B, Lq, 2).transpose(0, 1)    # (B,V,Lq,2)
        proj2d = (r_all * img).view(V, B, Lq, 2).transpose(0, 1)
        conf = torch.softmax(cl.reshape(V, B, Lq).transpose(0, 1), 1)
        # un-crop + undistortion (dq_decoder.py:414-420, 119-204): one launch forward (with every point's Jacobian), one small
        # product backward (geometry_torch.UncropUndistort) -- as torch ops ~80 launches forward and ~160 backward per layer.
        # Triangulation (dq_decoder.py:929-967): one launch each way over the dense token grid (geometry_torch.DenseDLT).  Only the
        # matched queries are triangulated, like the reference: an unmatched query with a degenerate DLT (homogeneous w == 0, an
        # Inf 2D point) would otherwise put 0 * inf = NaN into the gradients of the parameters all queries share.
        ud = G.UncropUndistort.apply(ref2d, tc["cams"], V, B)
        new_ref = G.DenseDLT.apply(ud, conf, tc["Pm"], valid.to(torch.uint8), J)
        vm2 = valid.view(B, 1, NQ, 1, 1)
        zero = torch.zeros((), device=dev)
        ref2d_o = torch.where(vm2, ref2d.view(B, V, NQ, J, 2), zero).reshape(B, V, Lq, 2)
        proj2d_o = torch.where(vm2, proj2d.view(B, V, NQ, J, 2), zero).reshape(B, V, Lq, 2)
        return tgt_update, new_ref, ref2d_o, proj2d_o, prob

    # The layer in two halves, so that a query-sharded run can put its one per-layer exchange (the global
    # "any query valid" flag, mvgformer_amd.dist) between two captured HIP-graph segments.
    def forward_features(self, tgt, query_pos, reference_points, ctx, threshold, indices=None):
        """steps 1-4 of dq_decoder.py:850-1045: projection, projective attention, feature update, class
        head + filter, 2D offsets.  Returns the state consumed by forward_triangulate."""
        B, Lq, C = tgt.shape
        J = self.num_joints
        NQ = Lq // J
        dt = self.compute_dtype
        V = ctx.V

        # 1. projective attention features of every view (generate_features, dq_decoder.py:516-593)
        X = reference_points.detach().reshape(B, Lq, 3).float().contiguous()
        proj_in, self._proj_in = self._proj_in, None
        if proj_in is not None and proj_in[0].data_ptr() == X.data_ptr() and tuple(proj_in[0].shape) == tuple(X.shape):
            r, ref_lvl, inside = proj_in[1]      # projected by the previous layer's triangulation launch (same arithmetic)
        else:
            r, ref_lvl, inside = ops.project(X, ctx.cams, ctx.levels, V, B)
        x = lambda: self.with_pos_embed(tgt.float(), None if query_pos is None else query_pos.float()).contiguous()
        if (query_pos is not None and tgt.dtype == torch.float32 and query_pos.dtype == torch.float32 and tgt.is_contiguous()
                and query_pos.is_contiguous() and query_pos.shape == tgt.shape):
            x.parts = (tgt, query_pos)          # lets the fast path fold the add into the query-term GEMM
        xw_in, self._xw_in = self._xw_in, None     # query term computed by the previous layer's chain B (or None)
        if xw_in is not None and tuple(xw_in.shape) != (B * Lq, 192):
            xw_in = None
        f32 = torch.float32
        pose_layers = self.pose_embed.MLP.layers
        fuse_a, fuse_b = self._fuses_chains(dt)
        o = None
        if self.proj_attn._vp_event is None and getattr(ctx, "_packed_event", None) is not None:
            # this layer's pyramid products run inline on THIS stream (no side-stream launch reached it) while the pyramid was packed
            # on the side stream (DQDecoder.pack_pyramid): order the read behind the pack and keep the allocator informed
            torch.cuda.current_stream().wait_event(ctx._packed_event)
            ctx.feat.record_stream(torch.cuda.current_stream())
        if fuse_a:
            # processing order of the (image, query) pairs: image-space (Morton) order, pairs outside the image last
            # (mvg_bin_pairs); shared by the sampler (L1 locality, masked pairs skipped) and chain A (all-masked
            # tiles skipped).  "first": binned once per forward from the first layer's projections.
            order = None
            mode = self.proj_attn.sort_pairs if Lq <= 65536 else False
            if mode == "first":
                if getattr(ctx, "order", None) is None or ctx.order.numel() != V * B * Lq:
                    ctx.order = ops.bin_pairs(ref_lvl, None, ctx.levels)
                order = ctx.order
            elif mode:
                order = ops.bin_pairs(ref_lvl, inside.view(-1), ctx.levels)
            samp = self.proj_attn.native_sample(x, ref_lvl, ctx.feat, ctx.levels, V, B, pair_mask=inside.view(-1),
                                                order=order, xw=xw_in)
            wts, o_masked = self._chain_a_weights(dt)
            attn, o = ops.chain_attn_pose(samp, inside.view(-1), *wts, order=order, o_masked=o_masked)
        elif self._fuses_chains_f32(dt, Lq, ctx.levels)[0] and C == 256:
            # fp32, fused: G-sampling kernel + chain A on pre-split operands (csrc/f32s.hip); pairs in processing order, masked
            # pairs last (zero-filled by the sampler, all-masked tiles skipped by the chain)
            order = ops.bin_pairs(ref_lvl, inside.view(-1), ctx.levels) if (self.proj_attn.sort_pairs and Lq <= 65536) else None
            samp = self.proj_attn.native_sample(x, ref_lvl, ctx.feat, ctx.levels, V, B, pair_mask=inside.view(-1), order=order,
                                                xw=xw_in)
            wts, o_masked = self._chain_a_weights_f32h()
            attn, o = ops.chain_attn_pose_f32h(samp, inside.view(-1), *wts, order=order, o_masked=o_masked)
        else:
            # fp32 (reference arithmetic): the per-view output projection and pose MLP run over the pairs in processing order and
            # skip the tiles whose pairs are all outside their image (their rows are zero / a cached constant either way).
            # Worth 1.7 % at cfg-2 (41 % of the tiles skipped at layer 0, but the GEMM is power-limited: tiles of zero rows were
            # cheap already); below 8192 tokens per image (cfg-4) the extra binning launch costs more than it saves.
            order32 = None
            if (dt == torch.float32 and self.skip_masked_f32 and C == 256 and 8192 <= Lq <= 65536
                    and self.proj_attn.sort_pairs):
                order32 = ops.bin_pairs(ref_lvl, inside.view(-1), ctx.levels)
            attn = self.proj_attn.native_forward(x(), ref_lvl, ctx.feat, ctx.levels, V, B, rowmask=inside.view(-1), order=order32)

        # 2.+3. update the query features (update_feature 'MLP', dq_decoder.py:763-778 + forward_ffn),
        #       class head + filter (dq_decoder.py:889-908)
        forced = None
        if not self.filter_query or self.query_filter_method == "all":
            forced = torch.ones((B, NQ), dtype=torch.uint8, device=tgt.device)
        elif indices is not None:
            forced = torch.zeros((B, NQ), dtype=torch.uint8, device=tgt.device)
            for b, q in enumerate(indices):
                forced[b, torch.as_tensor(q, dtype=torch.long, device=tgt.device)] = 1
        tgt32 = tgt.float().reshape(B * Lq, C).contiguous()
        fuse_b = fuse_b and C == 256 and J <= 64
        fuse_b32 = self._fuses_chains_f32(dt)[1] and C == 256 and not fuse_b
        if fuse_b32:
            nxt = self._next_layer[0] if self._next_layer else None
            next_proj = None
            if (nxt is not None and nxt.compute_dtype == dt and nxt._fuses_chains_f32(dt, Lq, ctx.levels)[0]
                    and (query_pos is None or query_pos.shape == tgt.shape)):
                qp = None if query_pos is None else query_pos.float().reshape(B * Lq, C).contiguous()
                (Wn, sn), bn, n_next = nxt.proj_attn.query_term_weights_f32h()
                next_proj = (qp, Wn, sn, bn, n_next)
            res = ops.chain_update_ffn_class_f32h(
                attn, V, tgt32, *self._chain_b_weights_f32h(), threshold, B, NQ, J, forced, self.open_forward_ffn,
                tgt_out=self._tgt_out, any_valid=self._flag, next_query_proj=next_proj)
            tgt_update, prob, valid, any_valid = res[:4]
            if next_proj is not None:
                nxt._xw_in = res[4]
        elif fuse_b:
            ffn = self.open_forward_ffn
            # the next layer's query term xw = (tgt' + query_pos) W^T + b rides on this chain (its rows are in LDS)
            nxt = self._next_layer[0] if self._next_layer else None     # (kept in a tuple: not a sub-module)
            next_proj = None
            if (nxt is not None and nxt.compute_dtype == dt and nxt.use_fused_chains and nxt.proj_attn.uses_fast_path(dt)
                    and (query_pos is None or query_pos.shape == tgt.shape)):
                Wn, bn, n_next = nxt.proj_attn.query_term_weights(dt)
                qp = None if query_pos is None else query_pos.float().reshape(B * Lq, C).contiguous()
                next_proj = (qp, Wn, bn, n_next)
            res = ops.chain_update_ffn_class(
                attn, V, tgt32, *self._chain_b_weights(dt), threshold, B, NQ, J, forced, ffn, tgt_out=self._tgt_out,
                any_valid=self._flag, next_query_proj=next_proj,
                attn_inside=inside.view(-1) if fuse_a else None)      # chain A wrote zeros for the pairs outside their image
            tgt_update, prob, valid, any_valid = res[:4]
            if next_proj is not None:
                nxt._xw_in = res[4]
        else:
            mean = ops.mean_views(attn, V)
            u = ops.linear(mean, self._w("Wu", (self.feature_update_mlp.weight,), dt),
                           self._w("bu", (self.feature_update_mlp.bias,), f32), out_dtype=dt)
            t1 = ops.add_layernorm(tgt32, u, self._w("g2", (self.norm2.weight,), f32), self._w("b2", (self.norm2.bias,), f32))
            if self.open_forward_ffn:
                h = ops.linear(t1, self._w("W1", (self.linear1.weight,), dt), self._w("b1", (self.linear1.bias,), f32),
                               out_dtype=dt, relu=True)
                f = ops.linear(h, self._w("W2", (self.linear2.weight,), dt), self._w("bb2", (self.linear2.bias,), f32),
                               out_dtype=dt)
                tgt_update = ops.add_layernorm(t1, f, self._w("g3", (self.norm3.weight,), f32),
                                               self._w("b3", (self.norm3.bias,), f32))
            else:
                tgt_update = t1
            prob, valid, any_valid = ops.class_head(tgt_update, self._w("Wc", (self.class_embed.weight,), f32),
                                                    self._w("bc", (self.class_embed.bias,), f32), threshold, B, NQ, J, forced)
        # 4. 2D offsets from the per-view attention features (calculate_2d_offsets, dq_decoder.py:659-717)
        if o is None:
            hcur = attn
            masked = self._pose_masked_rows(dt) if (not fuse_a and order32 is not None) else None
            for i, lin in enumerate(pose_layers[:-1]):
                Wi, bi = self._w("Wpe%d" % i, (lin.weight,), dt), self._w("bpe%d" % i, (lin.bias,), f32)
                if masked is not None:
                    hcur = ops.linear_ordered(hcur, Wi, bi, order32, inside.view(-1), masked[i], relu=True)
                else:
                    hcur = ops.linear(hcur, Wi, bi, out_dtype=dt, relu=True)
            o = ops.rowdot3(hcur, self._w("Wpe_last", (pose_layers[-1].weight,), f32),
                            self._w("bpe_last", (pose_layers[-1].bias,), f32))

        hook, self._after_chain_b = self._after_chain_b, None
        if hook is not None:
            hook()      # just-in-time schedule: the next layer's pyramid products run next to this layer's triangulation + the binning
        return dict(r=r, o=o, valid=valid, any_valid=any_valid, tgt_update=tgt_update.view(B, Lq, C), prob=prob,
                    dims=(V, B, NQ, J))

    def forward_triangulate(self, st, ctx):
        """step 5: triangulation + scatter (learnable_triangulate, dq_decoder.py:399-461,1013-1029)."""
        V, B, NQ, J = st["dims"]
        nxt = self._next_layer[0] if (self._next_layer and self.fuse_boundary) else None
        if nxt is not None:      # the next layer's projection of the new points rides on this launch
            new_ref, ref2d, proj2d, proj = ops.triangulate(st["r"], st["o"], ctx.cams, st["valid"], st["any_valid"], V, B, NQ,
                                                           J, out=self._geo_out, next_levels=ctx.levels)
            nxt._proj_in = (new_ref, proj)
        else:
            new_ref, ref2d, proj2d = ops.triangulate(st["r"], st["o"], ctx.cams, st["valid"], st["any_valid"], V, B, NQ, J,
                                                     out=self._geo_out)
        return st["tgt_update"], new_ref, ref2d, proj2d, st["prob"]


class MvPDecoder(nn.Module):
    """mvp_decoder.py:267-298 (constructor + coordinate helpers)."""

    def __init__(self, cfg, decoder_layer, num_layers, return_intermediate=False):
        super().__init__()
        if cfg.DECODER.share_layer_weights:
            self.layers = nn.ModuleList([decoder_layer for _ in range(num_layers)])
        else:
            self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.pose_embed = None
        self.class_embed = None
        self.grid_size = torch.tensor(cfg.MULTI_PERSON.SPACE_SIZE)
        self.grid_center = torch.tensor(cfg.MULTI_PERSON.SPACE_CENTER)

    def absolute2norm(self, absolute_coords):
        device = absolute_coords.device
        grid_size = self.grid_size.to(device=device)
        grid_center = self.grid_center.to(device=device)
        return (absolute_coords - grid_center + grid_size / 2.0) / grid_size

    def norm2absolute(self, norm_coords):
        device = norm_coords.device
        grid_size = self.grid_size.to(device=device)
        grid_center = self.grid_center.to(device=device)
        return norm_coords * grid_size + grid_center - grid_size / 2.0


class DQDecoder(MvPDecoder):
    """dq_decoder.py:1101-1172.  (The reference indexes a 4-entry meter list by layer id,
    dq_decoder.py:94,1142, and therefore crashes with more than 4 layers; not inherited.)"""

    def __init__(self, cfg, decoder_layer, num_layers, return_intermediate=False):
        super().__init__(cfg, decoder_layer, num_layers, return_intermediate)
        # The pyramid-side GEMMs of every layer (value planes + G, 73 us per layer) depend only on the packed
        # feature maps: issue all of them on a side stream at the start of the forward (fork/join -> a parallel
        # branch of the captured HIP graph).  They fill in next to the latency-bound query-side kernels:
        # cfg-2 bf16 1.43 -> 1.30 ms per sample on MI355X.  MVG_OVERLAP_PYRAMID=0 runs them inline.
        self.overlap_pyramid = os.environ.get("MVG_OVERLAP_PYRAMID", "1") != "0"
        self.overlap_pyramid_f32 = os.environ.get("MVG_OVERLAP_PYRAMID_F32", "1") != "0"
        self._side_stream = None
        # layer l's fused chain B also computes layer l+1's query term xw = (tgt' + query_pos) W^T + b (bf16 path)
        self.fuse_next_query_term = True
        # bf16: layers per grouped launch of the pyramid products behind layer 0's own (0 = one launch per product, rounds 1-4)
        self.pyramid_group = int(os.environ.get("MVG_PYRAMID_GROUP", "3"))
        # pack the pyramid on the side stream in front of its consumers: the first layer's query-side prologue (projection, pair
        # binning, query term) then runs next to the pack instead of behind it
        self.pack_on_side = os.environ.get("MVG_PACK_ON_SIDE", "1") != "0"
        # bf16, one sample per forward: layer l+1's products (one launch, ONE workgroup per CU) are issued behind layer l's chain B, next
        # to its triangulation and the next binning, instead of all up front -- the sampler then finds them in the 256-MB Infinity
        # Cache (137 -> 128 us) and the first sampler has the chip to itself: -2.8 ... -3.1 % at cfg-2, -2 % at cfg-5, +2 % at two samples
        # per forward (profiles/r05_experiments.txt section 10).  MVG_PYRAMID_JIT = 0 | 1 overrides the choice.
        self.pyramid_jit = os.environ.get("MVG_PYRAMID_JIT", "auto")
        self.pyramid_jit_slots = int(os.environ.get("MVG_PYRAMID_JIT_SLOTS", "32"))
        self._share_f32_pool()

    def _share_f32_pool(self):
        """the inline fp32 pyramid products share one (value, G) pair per stream -- of THIS decoder (a ProjAttn used on its own, or a
        deep copy of one, has a private, empty pool: ProjAttn.__deepcopy__)"""
        pool = {}
        for layer in self.layers:
            layer.proj_attn._f32_pool = pool

    def __deepcopy__(self, memo):
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._share_f32_pool()           # the copy's layers share ONE fresh pool again, not one each
        return new

    def set_compute_dtype(self, dtype):
        for layer in self.layers:
            layer.set_compute_dtype(dtype)
        return self

    def fork_side_stream(self, device, f32_shape=None):
        """The side stream of the forward, forked from the current stream -- or None when the query-independent work
        runs inline (generic path, share_layer_weights -- one buffer for all layers --, training, CPU,
        MVG_OVERLAP_PYRAMID=0).  f32_shape = (Lq, L, S): needed for the fp32 path, whose pyramid products exist only in its
        G-sampling form (ProjAttn.f32_g_form)."""
        distinct = len({id(l.proj_attn) for l in self.layers}) == len(self.layers)

        def ahead(l):
            if l.compute_dtype == torch.float32:
                return (self.overlap_pyramid_f32 and f32_shape is not None and l.proj_attn.rayconv.weight.shape[0] == 256
                        and l.proj_attn.f32_g_form(*f32_shape))
            return l.proj_attn.uses_fast_path(l.compute_dtype) and l.use_fused_chains
        if not (self.overlap_pyramid and distinct and device.type == "cuda" and not torch.is_grad_enabled()
                and all(ahead(l) for l in self.layers)):
            return None
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
        if not torch.cuda.is_current_stream_capturing():
            for l in self.layers:   # cached operands are built here, on the forking stream, never on the side stream
                l.prepare_caches()
        self._side_stream.wait_stream(torch.cuda.current_stream())
        return self._side_stream

    def launch_pyramid_projections(self, ctx, side=None, forked=False, jit=False):
        """Issue every layer's query-independent GEMMs (ProjAttn.project_pyramid) on the side stream, each followed
        by an event its consumer waits on.  Returns the side stream (pass it to join_pyramid_projections before
        the forward / the captured graph ends) or None when the projections run inline.  forked: `side` already waits for
        whatever produced ctx.feat (pack_pyramid).  jit: the caller runs the whole forward inside ONE fork / join of `side` (not the
        segmented graphs of mvgformer_amd.dist) -- the just-in-time schedule may be used."""
        for layer in self.layers:
            layer._after_chain_b = None       # hooks of an earlier forward that did not reach its layer (an exception in between)
        if side is None:
            side = self.fork_side_stream(ctx.feat.device)
            if side is None:
                return None
        elif not forked:
            side.wait_stream(torch.cuda.current_stream())       # the packed pyramid
        ctx.feat.record_stream(side)
        with torch.cuda.stream(side):
            groups = self._pyramid_groups(ctx)
            if groups is None:
                for layer in self.layers:
                    layer.proj_attn.project_pyramid(ctx.feat, record_event=True)
            else:
                # bf16 fast path: layer 0's value planes + G in one launch (the first sampler waits for nothing else), then the
                # remaining layers' products in launches of `pyramid_group` layers: every launch reads the pyramid through the
                # fabric once (ops.pyramid_group_ws); 8 reads of 103 MB per forward at cfg-2 become 2.  These launches fill every
                # CU (2 x 246 registers per SIMD lane) and so does the sampler: next to each other they run one after the other,
                # whatever the issue order -- issuing a group behind the sampler in front of it, or with half the workgroups,
                # measured slower (profiles/r05_experiments.txt section 2).  What does pay (section 10): one launch per layer with ONE
                # workgroup per CU, issued behind the previous layer's chain B (use_jit below).
                use_jit = self._pyramid_jit(ctx, jit)
                launches = self.pyramid_launches(ctx, jit)
                for gi, (group, slots) in enumerate(launches):
                    def issue(group=group, slots=slots):
                        jobs = []
                        for layer in group:
                            jobs += layer.proj_attn.pyramid_jobs(ctx.feat)
                        ops.pyramid_group_ws(ctx.feat, jobs, slots=slots)
                        ev = torch.cuda.Event()
                        ev.record()
                        for layer in group:
                            layer.proj_attn._vp_event = ev
                    if use_jit and gi > 0:
                        def hook(issue=issue):
                            side.wait_stream(torch.cuda.current_stream())     # behind the previous layer's chain B
                            with torch.cuda.stream(side):
                                issue()
                        self.layers[gi - 1]._after_chain_b = hook
                    else:
                        issue()
        return side

    def pack_pyramid(self, ctx, src_views, side):
        """ctx.pack(src_views) -- on the side stream (forked from the current one) when every consumer of the packed pyramid runs
        there (bf16: the pyramid products), so that the first layer's prologue does not queue behind the 46-us pack; follow it
        with launch_pyramid_projections(ctx, side, forked=True)."""
        ctx._packed_event = None
        if side is not None and self.pack_on_side and ctx.dtype == torch.bfloat16:
            with torch.cuda.stream(side):
                ctx.pack(src_views)
                # a consumer on ANOTHER stream (ProjAttn.native_sample's inline products, when a layer's just-in-time hook did not
                # fire) waits on this event; ctx.feat was allocated on `side`
                ctx._packed_event = torch.cuda.Event()
                ctx._packed_event.record()
        else:
            ctx.pack(src_views)
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())

    def _pyramid_jit(self, ctx, jit=True):
        """True when a forward of this context issues one product launch per layer, just in time (launch_pyramid_projections)"""
        return bool(jit and len(self.layers) > 1 and (self.pyramid_jit == "1" or (self.pyramid_jit == "auto" and ctx.B == 1)))

    def pyramid_launches(self, ctx, jit=True):
        """[(layers, workgroups per XCD; 0 = the kernel's default)]: the product launches of one forward in issue order
        (bf16 fast path; None otherwise)"""
        groups = self._pyramid_groups(ctx)
        if groups is None:
            return None
        if self._pyramid_jit(ctx, jit):
            # (layer 0's launch too: with two workgroups per CU it starved the first layer's binning + query-term GEMM, which run next
            # to it -- bin_scatter 57 us instead of 11, the first sampler at 142 us instead of 127: forward -0.9 %)
            return [([l], self.pyramid_jit_slots) for l in self.layers]
        return [(g, 0) for g in groups]

    def _pyramid_groups(self, ctx):
        """layers whose pyramid products share one launch (bf16 fast path), or None for one launch pair per layer"""
        if self.pyramid_group <= 0 or ctx.feat.dtype != torch.bfloat16 or ctx.feat.shape[2] != 256:
            return None
        if not all(l.proj_attn.uses_fast_path(torch.bfloat16) and l.proj_attn.rayconv.weight.shape == (256, 256) for l in self.layers):
            return None
        layers = list(self.layers)
        groups = [layers[:1]]
        rest = layers[1:]
        per = min(self.pyramid_group, 4)            # 2 jobs per layer, at most 8 per launch
        while rest:
            # equal-sized launches: 5 remaining layers at 3 per launch run as 3 + 2
            n = -(-len(rest) // -(-len(rest) // per))
            groups.append(rest[:n])
            rest = rest[n:]
        return groups

    def join_pyramid_projections(self, side, keep_results=False):
        """The current stream waits for the side stream.  keep_results: projections not consumed yet stay valid for
        the layers that follow on this stream (the segmented graphs of mvgformer_amd.dist); else they are dropped."""
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        for layer in self.layers:
            pa = layer.proj_attn
            pa._vp_event = True if (keep_results and pa._vp_event is not None) else None

    def forward(self, tgt, reference_points, src_views, meta, src_spatial_shapes, src_level_start_index,
                src_valid_ratios, query_pos=None, src_padding_mask=None, rgb_views=None, output_dir="./",
                frame_id=None, indices=None, threshold=0.5, indices_all=None, context=None):
        """Returns (hs (layers,B,Lq,C), refs (layers,B,Lq,3), refs2d (layers,B,V,Lq,2),
        projs2d (layers,B,V,Lq,2), [class_prob (B,NQ,2)] * layers) when return_intermediate,
        else (output, reference_points, ref_points_2d).  ``context`` (optional, beyond the reference
        signature): a DecoderContext.prepare(...)d context, so the host-side camera packing is
        hoisted out of a captured HIP graph."""
        if tgt.is_cuda and tgt.device.index != torch.cuda.current_device():
            with torch.cuda.device(tgt.device):     # kernels go to the current stream of the TENSORS' device
                return self.forward(tgt, reference_points, src_views, meta, src_spatial_shapes, src_level_start_index,
                                    src_valid_ratios, query_pos, src_padding_mask, rgb_views, output_dir, frame_id, indices,
                                    threshold, indices_all, context)
        output = tgt
        layer0 = self.layers[0]
        ctx = context
        side = None
        hs_buf = flags = geo_buf = None
        try:
            deferred_pack = None
            if ctx is None:
                ctx = DecoderContext.prepare(src_spatial_shapes, src_level_start_index, meta, layer0.img_size,
                                             layer0.compute_dtype, tgt.shape[0], src_views[0].device)
                deferred_pack = src_views
            elif src_views is not None:
                # a prepared context is reused across frames (static cameras): the pyramid is re-packed from THIS
                # call's src_views every time (a no-op for levels produced in place in ctx.pyramid_buffers())
                deferred_pack = src_views
            elif ctx.feat is None:
                raise RuntimeError("DQDecoder.forward: context without a packed pyramid and no src_views")
            ctx.order = None
            inter, inter_ref, inter_2d, inter_proj, classes = [], [], [], [], []
            ref_points_2d = None
            side = self.fork_side_stream(tgt.device, (tgt.shape[1], ctx.levels.L, ctx.levels.S))
            if deferred_pack is not None:
                self.pack_pyramid(ctx, deferred_pack, side)
            if side is not None:
                self.launch_pyramid_projections(ctx, side, forked=True, jit=True)
            # the fused chain writes every layer's hidden state straight into its slice of the stacked output
            hs_buf = None
            if self.return_intermediate and not torch.is_grad_enabled() and tgt.is_cuda:
                hs_buf = torch.empty((len(self.layers),) + tuple(tgt.shape), dtype=torch.float32, device=tgt.device)
            flags = torch.zeros((len(self.layers),), dtype=torch.int32, device=tgt.device) if tgt.is_cuda else None
            geo_buf = None
            if hs_buf is not None:
                nl, Bq, Lq_ = len(self.layers), tgt.shape[0], tgt.shape[1]
                Vn = ctx.V
                geo_buf = (torch.empty((nl, Bq, Lq_, 3), dtype=torch.float32, device=tgt.device),
                           torch.empty((nl, Bq, Vn, Lq_, 2), dtype=torch.float32, device=tgt.device),
                           torch.empty((nl, Bq, Vn, Lq_, 2), dtype=torch.float32, device=tgt.device))
            for lid, layer in enumerate(self.layers):
                layer._ctx = ctx
                layer._tgt_out = None if hs_buf is None else hs_buf[lid]
                layer._flag = None if flags is None else flags[lid:lid + 1]
                layer._geo_out = None if geo_buf is None else (geo_buf[0][lid], geo_buf[1][lid], geo_buf[2][lid])
                layer._next_layer = ((self.layers[lid + 1],) if (self.fuse_next_query_term and lid + 1 < len(self.layers))
                                     else None)
                output, reference_points, ref_points_2d, projs_2d_absolute, outputs_class = layer(
                    output, query_pos, reference_points[:, :, None] if reference_points.dim() == 3 else reference_points,
                    src_views, src_spatial_shapes, src_level_start_index, meta, src_padding_mask,
                    rgb_views=rgb_views, output_dir=output_dir, frame_id=frame_id, indices=indices,
                    threshold=threshold, indices_all=indices_all)
                if self.return_intermediate:
                    inter.append(output)
                    inter_ref.append(reference_points)
                    inter_2d.append(ref_points_2d)
                    inter_proj.append(projs_2d_absolute)
                    classes.append(outputs_class)
        finally:
            for layer in self.layers:
                layer._ctx = None
                layer._tgt_out = None
                layer._flag = None
                layer._geo_out = None
                layer._next_layer = None
                layer._xw_in = None
                layer._proj_in = None
                layer._after_chain_b = None
            self.join_pyramid_projections(side)
        if self.return_intermediate:
            in_place = hs_buf is not None and all(t.data_ptr() == hs_buf[i].data_ptr() and t.shape == hs_buf[i].shape
                                                  for i, t in enumerate(inter))
            hs = hs_buf if in_place else torch.stack(inter)

            def stacked(parts, buf):
                same = buf is not None and all(t.data_ptr() == buf[i].data_ptr() and t.shape == buf[i].shape
                                               for i, t in enumerate(parts))
                return buf if same else torch.stack(parts)
            return (hs, stacked(inter_ref, None if geo_buf is None else geo_buf[0]),
                    stacked(inter_2d, None if geo_buf is None else geo_buf[1]),
                    stacked(inter_proj, None if geo_buf is None else geo_buf[2]), classes)
        return output, reference_points, ref_points_2d


# north_star aliases
MultiViewDecoderLayer = DQDecoderLayer
MultiViewDecoder = DQDecoder
