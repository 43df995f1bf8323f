"""ProjAttn -- projective attention module.  Same constructor, parameters, state-dict keys and
forward signature as the reference (lib/models/ops/modules/projattn.py:42-204); the compute
runs on libmvgformer_hip.so.

Two execution paths, both GPU-only (no CPU fallback):
  * inference (autograd off): gather -> MFMA linears -> fused softmax+locations+sampling
    kernel -> MFMA output projection.  Locations / attention weights never touch HBM.
  * training (autograd on): the reference's op chain with torch autograd for the dense parts
    and DeformFunction (HIP forward + backward kernels) for the sampling.
"""
from __future__ import annotations

import math
import warnings

import os

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from . import ops
from .functions import DeformFunction


def ops_host_levels_pair(spatial_shapes, level_start_index):
    """(shapes, starts) as host lists, through ops.host_levels' per-tensor cache (no D2H sync after the first look-up)"""
    sc, st = ops.host_levels(spatial_shapes, level_start_index)
    return [[sc[2 * i], sc[2 * i + 1]] for i in range(len(st))], list(st)


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0



class WeightCache:
    """Contiguous copies of module parameters in the compute dtype, rebuilt when a parameter
    is modified in place (``_version``) or replaced."""

    def __init__(self):
        self._store = {}

    def get(self, key, params, dtype, build=None):
        stamp = tuple((id(p), p._version, p.device) for p in params) + (dtype,)
        hit = self._store.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        if any(p.is_cuda for p in params) and torch.cuda.is_current_stream_capturing():
            # An entry built inside a capture would sit in the graph's private pool, stamped valid but unwritten until
            # the first replay: an eager forward before that replay (or after the parameters were restored) would read
            # garbage without any error.  Operands are built ahead of a capture: DQDecoderLayer.prepare_caches().
            raise RuntimeError("WeightCache: %r is not cached for the current parameters and a HIP-graph capture is in "
                               "progress; call DQDecoderLayer.prepare_caches() (or run one eager forward with the same "
                               "dtype and weights) before capturing" % key)
        with torch.no_grad():
            t = build(*params) if build is not None else params[0]
            extra = ()
            if isinstance(t, tuple):      # (tensor, host-side scalars ...): e.g. an operand and its power-of-two scale
                t, extra = t[0], tuple(t[1:])
            t = t.detach().to(dtype).contiguous()
        # An entry is built on whatever stream is current and then handed out to every stream (the decoder issues
        # query-independent work on a side stream): finish the build before anybody can see the entry.  Rare (first
        # use / parameters changed).
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        out = (t,) + extra if extra else t
        self._store[key] = (stamp, out)
        return out


class ProjAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, projattn_posembed_mode="use_rayconv"):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("d_model // n_heads should be a power of 2 (the HIP kernels vectorise over head channels)")
        self.im2col_step = 64
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        if projattn_posembed_mode == "use_rayconv":
            self.rayconv = nn.Linear(d_model + 3, d_model)
        elif projattn_posembed_mode == "use_2d_coordconv":
            self.rayconv = nn.Linear(d_model + 2, d_model)
        elif projattn_posembed_mode == "ablation_not_use_rayconv":
            self.rayconv = nn.Linear(d_model, d_model)
        else:
            raise ValueError("invalid projective attention posembed mode")
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()
        self.projattn_posembed_mode = projattn_posembed_mode
        self.compute_dtype = torch.float32
        # bf16 inference: weight-stationary pyramid GEMMs + head-plane value layout + G-sampling kernel
        # (False: the generic gather -> linear -> fused-sampling kernels, also the fp32 path)
        self.use_fast_path = True
        # bf16 fast path: sample the pairs in image-space (Morton) order.  "layer"/True: binned per layer; "first":
        # DQDecoderLayer bins once per forward (first layer) and reuses the order; False: query order
        # fp32 path: G-sampling (csrc/msda.hip: msda_gfused_f32_kernel) instead of gather -> Linear -> fused sampling.
        # True / False, or "auto": wherever the offsets / logits Linear has fewer rows on the pyramid (V*S) than on the
        # gathered reference points (V*Lq*L) -- cfg-2: 40 320 vs 46 080 rows per image, 6.80 -> 6.26 ms; cfg-4 (512 queries:
        # 39 900 vs 23 040) and a rank's shard of a query-sharded run keep the gather form.  The two forms agree to fp32
        # rounding, not bit for bit: pin it to True / False where runs with different query counts must match exactly.
        self.g_sampling_f32 = {"0": False, "1": True}.get(os.environ.get("MVG_G_SAMPLING_F32", "auto"), "auto")
        # training path: reference-point features through the HIP sampling op (False: torch grid_sample per level)
        self.ref_gather_native = os.environ.get("MVG_REF_GATHER", "1") != "0"
        # fp32 path: fused kernels on pre-split operands (csrc/f32s.hip): one-pass pyramid products here, chain A / chain B in
        # DQDecoderLayer.  MVG_F32_FUSED=0: the unfused kernels of rounds 1-3 (mvg_linear per GEMM).
        self.f32_fused = os.environ.get("MVG_F32_FUSED", "1") != "0"
        self.sort_pairs = os.environ.get("MVG_SORT_PAIRS", "layer")
        if self.sort_pairs in ("0", "off", "False"):
            self.sort_pairs = False
        self._wc = WeightCache()
        self._vp = None
        self._vp_event = None
        self._G = None
        # inline fp32 pyramid products: one (value, G) pair per (device, stream) shared by the layers of ONE decoder (DQDecoder
        # hands its layers a common dict); it dies with its owner
        self._f32_pool = {}

    def __deepcopy__(self, memo):
        """copies (EMA / checkpoint copies of a decoder) get fresh, empty run-time state: the pooled fp32 (value, G) buffers
        (0.36 GB per pair at cfg-2), the weight cache and the side-stream hand-off are not part of the module's state"""
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        skip = {"_f32_pool": dict, "_wc": WeightCache}
        for k, v in self.__dict__.items():
            if k in skip:
                new.__dict__[k] = skip[k]()
            elif k in ("_vp", "_G", "_vp_event"):
                new.__dict__[k] = None
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def _reset_parameters(self):
        constant_(self.sampling_offsets.weight.data, 0.)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2) \
            .repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid_init.view(-1))
        constant_(self.attention_weights.weight.data, 0.)
        constant_(self.attention_weights.bias.data, 0.)
        xavier_uniform_(self.rayconv.weight.data)
        constant_(self.rayconv.bias.data, 0.)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.)

    # ------------------------------------------------------------------ native building blocks
    def fused_supported(self, feat_lvls):
        return (self.projattn_posembed_mode == "ablation_not_use_rayconv" and self.d_model == 256 and
                self.n_heads == 8 and self.n_points == 8 and self.n_levels == 1 and 1 <= feat_lvls <= 4)

    def weights(self, dtype):
        """(Wv, bv, W_oa, b_oa, Wp, bp): value / [offsets|logits] / output projections."""
        wc = self._wc
        cat = lambda a, b: torch.cat([a, b], 0)
        return (wc.get("Wv", (self.rayconv.weight,), dtype),
                wc.get("bv", (self.rayconv.bias,), torch.float32),
                wc.get("Woa", (self.sampling_offsets.weight, self.attention_weights.weight), dtype, cat),
                wc.get("boa", (self.sampling_offsets.bias, self.attention_weights.bias), torch.float32, cat),
                wc.get("Wp", (self.output_proj.weight,), dtype),
                wc.get("bp", (self.output_proj.bias,), torch.float32))

    def _plane_buffer(self, n_img, S, device):
        """value planes buffer (every line is fully rewritten by each projection); kept across calls."""
        shape = (n_img, 8, S, 32)
        if self._vp is None or self._vp.dtype != torch.bfloat16 or tuple(self._vp.shape) != shape or self._vp.device != device:
            self._vp = torch.empty(shape, dtype=torch.bfloat16, device=device)
        return self._vp

    def prepare_fast_path(self, dtype):
        """Fill every cached operand of the bf16 fast path on the CURRENT stream (called before work is forked onto
        a side stream, so that no cache entry is ever produced on one stream and first read on another)."""
        self.weights(dtype)
        self._wc.get("bv", (self.rayconv.bias,), torch.float32)
        self._wc.get("Wv_frag", (self.rayconv.weight,), dtype, lambda w: ops.swizzle_weight(w.to(dtype)))
        self.query_term_weights(dtype)
        self._fast_query_weights(dtype)

    def f32_fused_active(self):
        """the fused fp32 kernels on split operands (csrc/f32s.hip) are in use: MVG_F32_FUSED and the library's f32_split knob
        (bench.py --f32-gemm exact, mvg_set_tuning("f32_split", 0)) both select them; with either off the fp32 path is the
        reference-arithmetic decomposition (fmaf-chain GEMMs, one launch each)"""
        from . import _lib
        _lib.load()         # applies MVG_TUNE: before that TUNING reads as the library's defaults
        return bool(self.f32_fused) and _lib.TUNING.get("f32_split", 1) != 0

    def f32_g_form(self, Lq, L, S):
        """fp32 path: G-sampling (the offsets / logits Linear applied to the pyramid) rather than gather-then-Linear -- by
        MVG_G_SAMPLING_F32, else whenever the gathered rows (Lq * L per image) outnumber the pyramid's (S)"""
        geometry = self.sampling_offsets.out_features + self.attention_weights.out_features == 192 and self.rayconv.weight.shape == (256, 256)
        if self.g_sampling_f32 == "auto" and self.f32_fused_active() and geometry:
            # with the one-pass pyramid kernel (mvg_pyramid_f32h) the G form moves fewer bytes at every shipped shape: the gather
            # form re-fetched 2.6 x its algorithmic bytes at cfg-4 (profiles/r03_bench_cfg4_fp32.json)
            return True
        use_g = self.g_sampling_f32 if self.g_sampling_f32 != "auto" else Lq * L >= S
        return bool(use_g) and geometry

    def _wait_pyramid(self):
        """both pyramid projections were produced ahead of time on a side stream (DQDecoder.launch_pyramid_projections)."""
        if self._vp_event is not True:          # True: already joined into this stream (segmented graphs)
            torch.cuda.current_stream().wait_event(self._vp_event)
        # the event stays armed until the end of the forward (DQDecoder.join_pyramid_projections clears it)
        return self._vp, self._G

    def project_pyramid(self, feat, record_event=False):
        """The two query-independent GEMMs of a layer: value planes (projattn.py:169) and G = feat @ [Wo; Wa]^T
        (the pyramid side of projattn.py:180-181).  Neither depends on the queries, so DQDecoder runs them for
        every layer on a side stream next to the latency-bound query-side kernels (record_event=True: the
        consumer waits on the event); both land in buffers kept across calls."""
        dt = feat.dtype
        n_img, S, Cc = feat.shape
        if dt == torch.float32:      # fp32 G-sampling form: plain (rows, 256) / (rows, 192) fp32 products, kept across calls
            Wv, bv = self.weights(dt)[:2]
            Wq, _ = self._fast_query_weights(dt)
            if record_event:
                # side-stream schedule: every layer's products exist at the same time -> one pair of buffers per layer
                if (self._vp is None or self._vp.dtype != dt or tuple(self._vp.shape) != (n_img, S, Cc) or self._vp.device != feat.device
                        or any(self._vp is pair[0] for pair in self._f32_pool.values())):
                    self._vp = torch.empty((n_img, S, Cc), dtype=dt, device=feat.device)
                    self._G = torch.empty((n_img * S, 192), dtype=dt, device=feat.device)
            else:
                # inline (overlap off, or under autograd): a layer's products are dead once its sampler ran, the next layer's are
                # written behind it on the same stream -> ONE pair for all layers of this decoder on this stream (0.36 GB at cfg-2
                # instead of 0.36 GB per layer held for the model's lifetime)
                slot = (feat.device, torch.cuda.current_stream(feat.device).cuda_stream)      # streams do not share (decoders in flight)
                cur = self._f32_pool.get(slot)
                if cur is None or tuple(cur[0].shape) != (n_img, S, Cc):
                    cur = self._f32_pool[slot] = (torch.empty((n_img, S, Cc), dtype=dt, device=feat.device),
                                               torch.empty((n_img * S, 192), dtype=dt, device=feat.device))
                self._vp, self._G = cur
            if self.f32_fused_active() and Cc == 256 and feat.is_contiguous():
                # two-part fp16 operands, three products (csrc/f32s.hip: pyramid_ws_f32h_kernel)
                (Wv_pl, sv), (Wg_pl, sg) = self.pyramid_planes_f32h()
                ops.pyramid_f32h(feat, Wv_pl, sv, bv, Wg_pl, sg, 192, value=self._vp, G=self._G)   # projattn.py:169 + 180-181
            else:
                ops.linear(feat.view(n_img * S, Cc), Wv, bv, out=self._vp.view(n_img * S, Cc))       # projattn.py:169
                ops.linear(feat.view(n_img * S, Cc), Wq, None, out=self._G)
            if record_event:
                self._vp_event = torch.cuda.Event()
                self._vp_event.record()
            return self._vp, self._G
        # the 103-MB value write first, G (gathered at random by the sampler) last: G is then the fresher
        # resident of the 256-MB Infinity Cache when the sampler starts
        vp = self.project_values(feat)
        shape = (n_img * S, 192)
        if self._G is None or self._G.dtype != torch.bfloat16 or tuple(self._G.shape) != shape or self._G.device != feat.device:
            self._G = torch.empty(shape, dtype=torch.bfloat16, device=feat.device)
        ops.feat_linear_ws(feat, self.query_term_weights(dt)[0], 192, out=self._G)
        if record_event:
            self._vp_event = torch.cuda.Event()
            self._vp_event.record()
        return vp, self._G

    def pyramid_jobs(self, feat):
        """this layer's two products as jobs of ops.pyramid_group_ws (value planes, G), into the buffers project_pyramid uses"""
        dt = feat.dtype
        n_img, S, _ = feat.shape
        bv = self._wc.get("bv", (self.rayconv.bias,), torch.float32)
        Wv_f = self._wc.get("Wv_frag", (self.rayconv.weight,), dt, lambda w: ops.swizzle_weight(w.to(dt)))
        vp = self._plane_buffer(n_img, S, feat.device)
        shape = (n_img * S, 192)
        if self._G is None or self._G.dtype != torch.bfloat16 or tuple(self._G.shape) != shape or self._G.device != feat.device:
            self._G = torch.empty(shape, dtype=torch.bfloat16, device=feat.device)
        return [(Wv_f, bv, vp, True), (self.query_term_weights(dt)[0], None, self._G, False)]

    def project_values(self, feat):
        """value = rayconv(input_flatten) (projattn.py:169) as bf16 head planes vh[img][head][s][32]."""
        dt = feat.dtype
        n_img, S, _ = feat.shape
        bv = self._wc.get("bv", (self.rayconv.bias,), torch.float32)
        Wv_f = self._wc.get("Wv_frag", (self.rayconv.weight,), dt, lambda w: ops.swizzle_weight(w.to(dt)))
        vp = self._plane_buffer(n_img, S, feat.device)
        ops.value_proj_planes_ws(feat, Wv_f, bv, vp)
        return vp

    def native_forward(self, x, r, feat, levels, V, B, rowmask=None, order=None):
        """Inference path on packed inputs.  x (B,Lq,C) f32 = tgt+query_pos; r (V*B,Lq,L,2) the
        per-level reference points; feat (V*B,S,C) channels-last pyramid in the compute dtype.
        Returns (V*B*Lq, C).  order (with rowmask, fp32): the pairs' processing order (ops.bin_pairs: masked pairs last) --
        tiles of the output projection without an unmasked row are not computed (their rows are zero either way)."""
        samp = self.native_sample(x, r, feat, levels, V, B, pair_mask=rowmask, order=order)
        _, _, _, _, Wp, bp = self.weights(feat.dtype)
        if order is not None and rowmask is not None and feat.dtype == torch.float32:
            zero = self._wc.get("zero_row", (self.output_proj.bias,), torch.float32, lambda b: torch.zeros_like(b))
            return ops.linear_ordered(samp, Wp, bp, order, rowmask, zero, rowmask=rowmask)
        return ops.linear(samp, Wp, bp, out_dtype=feat.dtype, rowmask=rowmask)   # projattn.py:203 (+ dq_decoder.py:585)

    def uses_fast_path(self, dt):
        return (dt == torch.bfloat16 and self.use_fast_path and
                self.sampling_offsets.out_features + self.attention_weights.out_features == 192)

    def query_term_weights(self, dt):
        """operands of xw = (tgt + query_pos) @ [Woff; Wattn]^T + b as the fused chain B of the PREVIOUS layer takes
        them: (weight fragments (256,256) bf16 zero-padded, bias (256,) f32 zero-padded, n = 192)."""
        perm = lambda a, b: torch.cat([a, b], 0)[ops.gsamp_column_order(a.device)]
        pad = lambda a, b: ops.swizzle_weight(torch.cat([perm(a, b), a.new_zeros(64, a.shape[1])], 0).to(dt))
        Wf = self._wc.get("Woa_frag", (self.sampling_offsets.weight, self.attention_weights.weight), dt, pad)
        bpad = lambda a, b: torch.cat([perm(a, b), a.new_zeros(64)], 0)
        bn = self._wc.get("boa_pad", (self.sampling_offsets.bias, self.attention_weights.bias), torch.float32, bpad)
        return Wf, bn, 192

    def pyramid_planes_f32h(self):
        """operands of mvg_pyramid_f32h: ((value planes, scale), (G planes, scale)) -- two fp16 parts of each weight times a power of two"""
        f16 = torch.float16
        perm = lambda a, b: ops.split_swizzle_weight_h2(torch.cat([a, b], 0)[ops.gsamp_column_order(a.device)])
        return (self._wc.get("Wv_f32h", (self.rayconv.weight,), f16, ops.split_swizzle_weight_h2),
                self._wc.get("Woa_f32h", (self.sampling_offsets.weight, self.attention_weights.weight), f16, perm))

    def query_term_weights_f32h(self):
        """operands of the next layer's query term as the two-part fp16 chain B of the PREVIOUS layer takes them: ((planes, scale) of the
        (192 -> 256, 256) weight in gsamp_column_order, bias (256,) f32 zero-padded, n = 192)"""
        Wpl = self.pyramid_planes_f32h()[1]
        perm = lambda a, b: torch.cat([a, b], 0)[ops.gsamp_column_order(a.device)]
        bpad = lambda a, b: torch.cat([perm(a, b), a.new_zeros(64)], 0)
        bn = self._wc.get("boa_pad", (self.sampling_offsets.bias, self.attention_weights.bias), torch.float32, bpad)
        return Wpl, bn, 192

    def _fast_query_weights(self, dt):
        """[offsets; logits] Linear in the column order of the G-sampling kernel (ops.gsamp_column_order)."""
        perm = lambda a, b: torch.cat([a, b], 0)[ops.gsamp_column_order(a.device)]
        W = self._wc.get("Woa_perm", (self.sampling_offsets.weight, self.attention_weights.weight), dt, perm)
        b = self._wc.get("boa_perm", (self.sampling_offsets.bias, self.attention_weights.bias), torch.float32, perm)
        return W, b

    def native_sample(self, x, r, feat, levels, V, B, pair_mask=None, order=None, xw=None):
        """everything of native_forward up to (not including) output_proj: (V*B*Lq, C) sampled values.
        x (B,Lq,C) f32 = tgt + query_pos, or a callable returning it (only evaluated when needed: with xw given --
        the query term (B*Lq,192) precomputed by the previous layer -- the bf16 fast path never touches x).
        pair_mask (V*B*Lq) u8: rows the caller is going to multiply by 0 (reference point outside the image,
        dq_decoder.py:585-586); the bf16 fast path returns zeros for them instead of sampling."""
        dt = feat.dtype
        Wv, bv, Woa, boa, Wp, bp = self.weights(dt)
        n_img, S, Cc = feat.shape
        parts = getattr(x, "parts", None)        # (tgt, query_pos) when the caller has not formed tgt + query_pos yet
        f32_g = dt == torch.float32 and Cc == 256 and self.f32_g_form(r.shape[1], levels.L, S)
        if callable(x) and not (self.uses_fast_path(dt) and (xw is not None or parts is not None)) and not (f32_g and xw is not None):
            x = x()
        if self.uses_fast_path(dt):
            # Linear(bilinear(feat) + x) = bilinear(Linear(feat)) + Linear(x): project the pyramid once (G), compute
            # the query term once per layer (xw), gather offsets/logits inside the sampler (csrc/msda.hip)
            # processing order of the pairs: given by the caller (DQDecoderLayer shares it with chain A) or binned here
            if order is None and self.sort_pairs and r.shape[1] <= 65536:
                order = ops.bin_pairs(r, pair_mask, levels)
            if xw is None:      # else: already computed by the previous layer's fused chain B
                Wq, bq = self._fast_query_weights(dt)
                if parts is not None:   # first layer: tgt + query_pos formed inside the GEMM's loader (mvg_linear_sum)
                    xw = ops.linear(parts[0].reshape(-1, Cc), Wq, bq, out_dtype=torch.float32, add=parts[1].reshape(-1, Cc))
                else:
                    xw = ops.linear(x.reshape(-1, Cc), Wq, bq, out_dtype=torch.float32)
            vp, G = self.project_pyramid(feat) if self._vp_event is None else self._wait_pyramid()
            return ops.msda_gsamp(vp, G, xw, r, levels, B, pair_mask=pair_mask, order=order)   # projattn.py:148-200
        if f32_g:
            # fp32, reference arithmetic, same re-association as the bf16 fast path: the offsets / logits Linear applied to
            # the pyramid once (G) instead of to V*Lq*L gathered rows -- no `ain` (236 MB) / `oa` (177 MB) per layer
            if xw is not None:      # computed by the previous layer's fp32 chain B
                xw32 = xw
            else:
                Wq, bq = self._fast_query_weights(dt)
                if callable(x):
                    x = x()
                xw32 = ops.linear(x.reshape(-1, Cc), Wq, bq, out_dtype=dt)
            if order is None and self.sort_pairs and r.shape[1] <= 65536:
                order = ops.bin_pairs(r, pair_mask, levels)
            # the two pyramid products: computed here, or ahead of time on the side stream (DQDecoder.launch_pyramid_projections)
            value, G32 = self.project_pyramid(feat) if self._vp_event is None else self._wait_pyramid()
            return ops.msda_gfused_f32(value, G32, xw32, r, levels, B, pair_mask=pair_mask, order=order)
        ain = ops.gather_ref(feat, r, x, levels, V, B)                       # projattn.py:148-153,180 (+query)
        oa = ops.linear(ain, Woa, boa, out_dtype=torch.float32)              # projattn.py:180-181
        value = ops.linear(feat.view(n_img * S, Cc), Wv, bv, out_dtype=dt)   # projattn.py:169
        return ops.msda_fused(value.view(n_img, S, Cc), oa, r, levels)       # projattn.py:184-200

    # ------------------------------------------------------------------------------- forward
    def _step(self, n):
        """im2col_step handed to the op for a batch of n images.  The reference calls the op once per view with the per-view
        batch B (dq_decoder.py:552-593) and its kernel chunks the batch by im2col_step = 64 (deform_cuda.cu:61-86); here all
        V views go through as ONE batch of V * B images and the kernels never chunk, so the argument is only the op's
        compatibility check: a step that divides n whenever the reference's per-view call would have passed."""
        return n if n <= self.im2col_step else math.gcd(n, self.im2col_step)

    def _ref_gather(self, flat, loc, shapes, starts):
        """bilinear features of every level at its reference point: flat (n, S, C) channels-last pyramid, loc (n, Lq, L, 2)
        in [0, 1] map coordinates (align_corners=False, zero padding -- grid_sample's and the op's common convention)
        -> (n, Lq, L, C), differentiable in flat and loc through DeformFunction."""
        n, Lq, nl, _ = loc.shape
        M = self.n_heads
        C = flat.shape[-1]
        locs = loc.view(n, Lq, 1, 1, nl, 1, 2).expand(n, Lq, nl, M, nl, 1, 2).reshape(n, Lq * nl, M, nl, 1, 2)
        onehot = torch.eye(nl, dtype=flat.dtype, device=flat.device).view(1, 1, nl, 1, nl, 1)
        w = onehot.expand(n, Lq, nl, M, nl, 1).reshape(n, Lq * nl, M, nl, 1)
        out = DeformFunction.apply(flat.view(n, -1, M, C // M), shapes.contiguous(), starts.contiguous(), locs.contiguous(),
                                   w.contiguous(), self._step(n))
        return out.view(n, Lq, nl, C)

    def forward(self, query, reference_points, src_views, camera_ray_embeds, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None):
        """Reference signature (projattn.py:115-117).  query (n_views, Lq, C); reference_points
        (n_views, Lq, n_levels, 2) already rescaled per level; src_views list of (n_views,C,H,W)."""
        if not query.is_cuda:
            raise RuntimeError("Not implemented on the CPU")                  # deform.h:49
        if query.device.index != torch.cuda.current_device():
            with torch.cuda.device(query.device):   # kernels go to the current stream of the TENSORS' device
                return self.forward(query, reference_points, src_views, camera_ray_embeds, input_spatial_shapes,
                                    input_level_start_index, input_padding_mask)
        n_views, Len_q, c = query.shape
        feat_lvls = len(src_views)
        if self.projattn_posembed_mode != "ablation_not_use_rayconv":
            raise NotImplementedError("projattn_posembed_mode=%r: only 'ablation_not_use_rayconv' (every shipped "
                                      "YAML) is built" % self.projattn_posembed_mode)
        if reference_points.shape[-1] != 2:
            raise ValueError("Last dim of reference_points must be 2, but get {} instead."
                             .format(reference_points.shape[-1]))
        if not torch.is_grad_enabled() and input_padding_mask is None and self.fused_supported(feat_lvls):
            levels = ops.Levels(input_spatial_shapes, input_level_start_index)
            feat = ops.pack_pyramid(src_views, levels, self.compute_dtype)
            out = self.native_forward(query.float().contiguous(), reference_points.float().contiguous(), feat, levels,
                                      1, n_views)
            return out.view(n_views, Len_q, c).to(query.dtype)
        sample_grid = torch.clamp(reference_points * 2.0 - 1.0, -1.1, 1.1)
        packed = getattr(self, "_packed_feat", None)
        if (packed is not None and packed.dtype == torch.float32 and packed.shape[0] == n_views and packed.shape[2] == c
                and not any(s.requires_grad for s in src_views)):
            # the channels-last pyramid the decoder packed once per forward IS cat + permute of the maps (projattn.py:160); only
            # when no gradient flows into the feature maps (the packing kernel is not differentiable)
            input_flatten = packed
        else:
            input_flatten = torch.cat([s.flatten(2) for s in src_views], dim=-1).permute(0, 2, 1)
        assert int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) == input_flatten.shape[1]
        xin = None
        if (self.ref_gather_native and input_flatten.dtype == torch.float32 and not input_flatten.requires_grad
                and not reference_points.requires_grad and c % 4 == 0):
            # neither the maps nor the reference points carry a gradient (run/train_3d.py with a frozen backbone and
            # detach_refpoints_cameraprj, every shipped YAML): the inference path's gather kernel forms feats + query in one launch
            # (the sampling op with one-hot level weights sampled every level three times: 380 us against ~100); the only gradient,
            # d / d query, is the sum over the levels
            from .functions import RefGatherAdd
            levels = ops.Levels(*ops_host_levels_pair(input_spatial_shapes, input_level_start_index))
            xin = RefGatherAdd.apply(query.contiguous(), input_flatten.contiguous(), reference_points.detach().float().contiguous(), levels)
        elif self.ref_gather_native and input_flatten.dtype == torch.float32:
            # reference-point features through the sampling op itself (forward AND deterministic backward kernels) instead
            # of L grid_sample launches on the NCHW maps (projattn.py:134-141; 17 % of a training step): the channels-last
            # pyramid is the op's value with M heads of C / M channels, token (q, l) samples level l at the clamped
            # reference point with weight 1 (one point per level, the other levels' weights are 0).
            input_flatten = input_flatten.contiguous()
            feats = self._ref_gather(input_flatten, (sample_grid + 1.0) * 0.5, input_spatial_shapes, input_level_start_index)
        else:
            feats = torch.stack([F.grid_sample(src_views[l], sample_grid[:, :, l:l + 1, :], align_corners=False).squeeze(-1)
                                 .permute(0, 2, 1) for l in range(feat_lvls)], dim=2)
        from .functions import linear as lin
        value = lin(input_flatten, self.rayconv.weight, self.rayconv.bias)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(n_views, -1, self.n_heads, self.d_model // self.n_heads)
        if xin is None:
            xin = feats + query.unsqueeze(2)
        n_off = self.sampling_offsets.out_features
        oa = lin(xin, torch.cat([self.sampling_offsets.weight, self.attention_weights.weight], 0),      # one GEMM for both heads
                 torch.cat([self.sampling_offsets.bias, self.attention_weights.bias], 0))
        sampling_offsets = oa[..., :n_off].reshape(n_views, Len_q, self.n_heads, feat_lvls, self.n_points, 2)
        attention_weights = oa[..., n_off:].reshape(n_views, Len_q, self.n_heads, feat_lvls * self.n_points)
        attention_weights = F.softmax(attention_weights, -1).view(n_views, Len_q, self.n_heads, feat_lvls,
                                                                  self.n_points)
        offset_normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
        sampling_locations = reference_points[:, :, None, :, None, :] \
            + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        output = DeformFunction.apply(value.contiguous(), input_spatial_shapes.contiguous(),
                                      input_level_start_index.contiguous(), sampling_locations.contiguous(),
                                      attention_weights.contiguous(), self._step(n_views))
        return lin(output, self.output_proj.weight, self.output_proj.bias)


# north_star alias
MSDeformAttn = ProjAttn
