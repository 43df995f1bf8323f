"""Frame-by-frame inference with the decoder forward captured ONCE as a HIP graph (single GPU).

The reference runs ``model(views, meta)`` eagerly per frame (lib/core/function.py:360-396): ~25 kernel launches per decoder layer,
several host synchronisations.  Here the whole ``DQDecoder.forward`` -- pyramid packing, the side-stream GEMM train, all layers,
triangulation, output stacking -- has static shapes and no host synchronisation, so it is captured once and replayed per frame:
one graph launch on the host (what ``bench.py`` times).  The graph holds raw pointers into its input buffers, the decoder's
weight caches and its per-layer value / G buffers: ``GraphedDecoder`` owns the inputs and, from the capture on, holds a
reference to every one of those tensors (``_pinned``) -- an eager call of the same decoder with another batch, resolution or
compute dtype re-allocates ``ProjAttn._vp / _G`` and rebuilds cache entries, and without the references the graph would go on
reading and writing recycled memory.  ``replay()`` also compares the buffers' addresses with the captured ones and re-captures
when the decoder has moved on to other buffers.

    runner = GraphedDecoder(decoder, meta, spatial_shapes, level_start_index, batch=1, num_queries=1024, threshold=0.1)
    for frame in stream:
        runner.load(src_views=frame.feature_maps, tgt=..., query_pos=..., reference_points=...)   # device copies, no sync
        hs, refs, refs2d, projs2d, class_probs = runner.replay()

Static per runner: the cameras (``meta``; call ``set_cameras`` when they change -- a 960-byte host-packed record per image, no
re-capture), map shapes, batch, query count, threshold and the decoder's weights (``refresh_weights()`` after an optimizer step
or a checkpoint load re-captures).  A backbone that writes its deconvolution outputs into ``runner.pyramid_views`` (channels-last,
compute dtype) makes ``load(src_views=...)`` unnecessary.
"""
from __future__ import annotations

import torch

from .decoder import DecoderContext


class GraphedDecoder:
    def __init__(self, decoder, meta, spatial_shapes, level_start_index, batch, num_queries, threshold=0.1, device=None,
                 producer_writes_in_place=False, channels=256):
        layer0 = decoder.layers[0]
        dev = torch.device(device) if device is not None else next(decoder.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("Not implemented on the CPU")
        self.dec, self.thr, self.dev = decoder, float(threshold), dev
        self.dtype = layer0.compute_dtype
        J, C = layer0.num_joints, channels
        self.V = len(meta)
        Lq = num_queries * J
        self.spatial_shapes = spatial_shapes.to(dev)
        self.level_start_index = level_start_index.to(dev)
        self.meta = meta
        self.ctx = DecoderContext.prepare(self.spatial_shapes, self.level_start_index, meta, layer0.img_size, self.dtype, batch, dev)
        # static inputs
        self.tgt = torch.zeros((batch, Lq, C), dtype=torch.float32, device=dev)
        self.query_pos = torch.zeros((batch, Lq, C), dtype=torch.float32, device=dev)
        self.reference_points = torch.zeros((batch, Lq, 3), dtype=torch.float32, device=dev)
        self.in_place = bool(producer_writes_in_place)
        if self.in_place:       # the producer's output buffers ARE the packed pyramid (no per-frame pack kernels)
            self.pyramid_views = self.ctx.pyramid_buffers(channels=C, device=dev)
            self.src_views = self.pyramid_views
        else:                   # the reference's hand-over format: NCHW fp32 maps, view-major
            shapes = [(int(h), int(w)) for h, w in self.spatial_shapes.tolist()]
            self.src_views = [torch.zeros((self.V * batch, C, h, w), dtype=torch.float32, device=dev) for h, w in shapes]
            self.pyramid_views = None
        self.graph, self.outputs = None, None
        self._pinned, self._captured_ptrs = [], None

    # ------------------------------------------------------------------ inputs (device-side copies on the current stream)
    def load(self, src_views=None, tgt=None, query_pos=None, reference_points=None):
        with torch.no_grad():
            if src_views is not None:
                for dst, s in zip(self.src_views, src_views):
                    dst.copy_(s, non_blocking=True)
            for dst, s in ((self.tgt, tgt), (self.query_pos, query_pos), (self.reference_points, reference_points)):
                if s is not None:
                    dst.copy_(s.reshape(dst.shape), non_blocking=True)
        return self

    def set_cameras(self, meta):
        """new calibration / crop for the same number of images: refills the packed camera records in place (host packing +
        one small H2D copy); the captured graph reads them through the same pointer."""
        fresh = DecoderContext.prepare(self.spatial_shapes, self.level_start_index, meta, self.dec.layers[0].img_size, self.dtype,
                                       self.tgt.shape[0], self.dev)
        if fresh.cams.shape != self.ctx.cams.shape:
            raise RuntimeError("set_cameras: %s camera records, the runner was built for %s"
                               % (tuple(fresh.cams.shape), tuple(self.ctx.cams.shape)))
        self.ctx.cams.copy_(fresh.cams)
        self.meta = meta
        return self

    # ------------------------------------------------------------------ capture / replay
    def _forward(self):
        self.ctx.feat = None          # re-packed from the static source buffers (a no-op for levels produced in place)
        return self.dec(self.tgt, self.reference_points, self.src_views, self.meta, self.spatial_shapes, self.level_start_index,
                        None, query_pos=self.query_pos, threshold=self.thr, context=self.ctx)

    def capture(self, warmup=2):
        with torch.no_grad():
            for _ in range(max(1, warmup)):         # builds the weight caches, sizes the per-layer buffers
                self._forward()
            torch.cuda.synchronize(self.dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.outputs = self._forward()
        # everything the graph addresses by raw pointer and does not own: the per-layer pyramid products and the cached operands
        self._pinned = self._graph_operands()
        self._captured_ptrs = self._buffer_ptrs()
        return self

    def _buffer_ptrs(self):
        return tuple((None if t is None else (t.data_ptr(), t.dtype, tuple(t.shape)))
                     for l in self.dec.layers for t in (l.proj_attn._vp, l.proj_attn._G))

    def _graph_operands(self):
        keep = [self.ctx.cams, self.ctx.feat, getattr(self.ctx, "_buffer", None)]
        for l in self.dec.layers:
            keep += [l.proj_attn._vp, l.proj_attn._G]
            for wc in (l._wc, l.proj_attn._wc):
                keep += [entry[1] for entry in wc._store.values()]
        return [t for t in keep if t is not None]

    def refresh_weights(self):
        """after the decoder's parameters changed: cached operands are version-checked, the graph is re-captured"""
        self.graph = None
        return self.capture()

    def replay(self):
        """one decoder forward on the loaded inputs; returns the graph's static output tensors (overwritten by the next replay)"""
        if self.graph is None or self._buffer_ptrs() != self._captured_ptrs:
            # never captured, or the decoder was run with other shapes / another dtype since (its per-layer buffers were
            # re-allocated): the old graph is still safe to replay (its buffers are pinned) but no longer the decoder's state
            self.capture()
        self.graph.replay()
        return self.outputs

    def eager(self):
        """the same forward without the graph (reference for tests; identical results)"""
        with torch.no_grad():
            return self._forward()
