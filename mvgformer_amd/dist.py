"""Query-sharded multi-GPU execution of the decoder.

Every person-query is an independent unit of work in the shipped configuration
(init_self_attention=False, feature_update_method='MLP'): nothing in ProjAttn, the MLP/FFN,
the class head, the view-softmax, the undistortion or the DLT couples two queries.  So the
NQ person-queries are split into contiguous blocks, one per rank (one process per GPU), each
rank runs the whole decoder on its block with the feature pyramid / cameras / weights
replicated, and ONE all-gather at the end of the forward assembles the pose set.  The only other
exchange is a 4-byte MAX all-reduce per layer that keeps the reference's "no query valid anywhere ->
force query (0,0)" rule (dq_decoder.py:620-623) global (install_any_valid_sync).

The messages are small (<= 1 MB per rank at 1024 queries without the hidden states), i.e.
latency-bound: all outputs of a rank are packed into one flat buffer so the exchange is a
single RCCL all_gather_into_tensor over xGMI, not one collective per tensor per layer.

Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(NQ, world, rank):
    """contiguous, balanced block [lo, hi) of person-queries for `rank` (first NQ % world ranks
    get one more)."""
    base, rem = divmod(NQ, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_queries(tgt, query_pos, reference_points, num_joints, world, rank):
    """slice (B, NQ*J, ...) token tensors to this rank's block of person-queries."""
    NQ = tgt.shape[1] // num_joints
    lo, hi = shard_bounds(NQ, world, rank)
    sl = slice(lo * num_joints, hi * num_joints)
    cut = lambda t: None if t is None else t[:, sl].contiguous()
    return cut(tgt), cut(query_pos), cut(reference_points), (lo, hi)


def pack_outputs(outputs, NQ, num_joints, world, rank, gather_hidden=False):
    """per-person records of this rank's block in ONE flat send buffer (nq_max, Ly, B, R) + its geometry."""
    hs, refs, r2d, p2d, cls = outputs
    J = num_joints
    cls_t = torch.stack(list(cls))                                      # (Ly,B,NQ_loc,2)
    Ly, B = refs.shape[:2]
    V = r2d.shape[2]
    C = hs.shape[-1]
    nq_max = -(-NQ // world)
    lo, hi = shard_bounds(NQ, world, rank)
    nq_loc = hi - lo
    assert refs.shape[2] == nq_loc * J, (refs.shape, nq_loc)
    # per-person record: [refs J*3 | refs2d V*J*2 | projs2d V*J*2 | cls 2 | (hs J*C)] per (layer, batch)
    parts = [refs.reshape(Ly, B, nq_loc, J * 3),
             r2d.reshape(Ly, B, V, nq_loc, J * 2).permute(0, 1, 3, 2, 4).reshape(Ly, B, nq_loc, V * J * 2),
             p2d.reshape(Ly, B, V, nq_loc, J * 2).permute(0, 1, 3, 2, 4).reshape(Ly, B, nq_loc, V * J * 2),
             cls_t.reshape(Ly, B, nq_loc, 2)]
    if gather_hidden:
        parts.append(hs.reshape(Ly, B, nq_loc, J * C))
    rec = torch.cat([p.float() for p in parts], -1)                     # (Ly,B,nq_loc,R)
    send = rec.new_zeros((nq_max, Ly, B, rec.shape[-1]))
    send[:nq_loc] = rec.permute(2, 0, 1, 3)
    return send, dict(Ly=Ly, B=B, V=V, C=C, J=J, NQ=NQ, world=world, nq_max=nq_max, gather_hidden=gather_hidden)


def unpack_outputs(recv, geo):
    """inverse of pack_outputs on the all-gathered buffer (world*nq_max, Ly, B, R)."""
    Ly, B, V, C, J, NQ, world, nq_max = (geo[k] for k in ("Ly", "B", "V", "C", "J", "NQ", "world", "nq_max"))
    keep = []
    for r_ in range(world):
        a, b = shard_bounds(NQ, world, r_)
        keep.append(recv[r_ * nq_max: r_ * nq_max + (b - a)])
    full = torch.cat(keep, 0).permute(1, 2, 0, 3)                       # (Ly,B,NQ,R)
    o = 0
    refs_f = full[..., o:o + J * 3].reshape(Ly, B, NQ * J, 3); o += J * 3
    r2d_f = full[..., o:o + V * J * 2].reshape(Ly, B, NQ, V, J, 2).permute(0, 1, 3, 2, 4, 5).reshape(Ly, B, V, NQ * J, 2)
    o += V * J * 2
    p2d_f = full[..., o:o + V * J * 2].reshape(Ly, B, NQ, V, J, 2).permute(0, 1, 3, 2, 4, 5).reshape(Ly, B, V, NQ * J, 2)
    o += V * J * 2
    cls_f = full[..., o:o + 2]; o += 2
    hs_f = full[..., o:o + J * C].reshape(Ly, B, NQ * J, C) if geo["gather_hidden"] else None
    return hs_f, refs_f.contiguous(), r2d_f.contiguous(), p2d_f.contiguous(), [cls_f[i].contiguous() for i in range(Ly)]


def gather_outputs(outputs, NQ, num_joints, group=None, gather_hidden=False):
    """all-gather the sharded decoder outputs.

    outputs = (hs (Ly,B,Lq_loc,C), refs (Ly,B,Lq_loc,3), refs2d (Ly,B,V,Lq_loc,2),
               projs2d (Ly,B,V,Lq_loc,2), [cls (B,NQ_loc,2)] * Ly)  -- DQDecoder.forward's tuple.
    Returns the same tuple for all NQ queries (hs is None unless gather_hidden)."""
    world = dist.get_world_size(group)
    send, geo = pack_outputs(outputs, NQ, num_joints, world, dist.get_rank(group), gather_hidden)
    recv = send.new_empty((world * geo["nq_max"],) + tuple(send.shape[1:]))
    dist.all_gather_into_tensor(recv, send, group=group)                # the one exchange step
    return unpack_outputs(recv, geo)


def sharded_decoder_forward(decoder, tgt, reference_points, src_views, meta, spatial_shapes, level_start_index,
                            query_pos, threshold, group=None, gather_hidden=False, context=None):
    """Run DQDecoder.forward on this rank's block of queries and all-gather the pose set."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    J = decoder.layers[0].num_joints
    NQ = tgt.shape[1] // J
    t, p, r, _ = shard_queries(tgt, query_pos, reference_points, J, world, rank)
    install_any_valid_sync(decoder, group)
    try:
        out = decoder(t, r, src_views, meta, spatial_shapes, level_start_index, None, query_pos=p,
                      threshold=threshold, context=context)
    finally:
        install_any_valid_sync(decoder, None, remove=True)
    return gather_outputs(out, NQ, J, group, gather_hidden)


def install_any_valid_sync(decoder, group, remove=False):
    """The reference forces query (0,0) through triangulation when NO query of the whole batch
    passes the filter (dq_decoder.py:620-623).  With sharded queries that predicate is global: a
    4-byte MAX all-reduce per layer makes it so, and only the rank that owns global query 0 may
    apply it."""
    if remove:
        for layer in decoder.layers:
            layer._any_valid_hook = None
        return
    rank = dist.get_rank(group)

    def hook(any_valid):
        dist.all_reduce(any_valid, op=dist.ReduceOp.MAX, group=group)
        if rank != 0:
            any_valid.fill_(1)

    for layer in decoder.layers:
        layer._any_valid_hook = hook


class GraphedShardedDecoder:
    """The query-sharded forward as HIP-graph SEGMENTS with the collectives issued eagerly in between:

        [pack pyramid + layer-0 features] (all-reduce any_valid_0) [layer-0 triangulate + layer-1 features] ...
        ... (all-reduce any_valid_last) [last triangulate + stack + pack records] (all-gather) [unpack]

    so the ~25 kernels of a layer cost one graph launch on the host, and no RCCL call is captured into a
    graph (Ly+2 graph launches and Ly+1 collectives per forward).  Inputs are static tensors: refill them
    in place (tgt / query_pos / reference_points are this rank's shard) and call replay()."""

    def __init__(self, decoder, tgt, reference_points, src_views, query_pos, ctx, threshold, NQ, group=None,
                 gather_hidden=False):
        self.dec, self.ctx, self.group, self.thr, self.NQ = decoder, ctx, group, threshold, NQ
        self.tgt, self.ref, self.src, self.qpos = tgt, reference_points, src_views, query_pos
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.J = decoder.layers[0].num_joints
        self.gather_hidden = gather_hidden
        self.graphs, self.flags = [], []
        for layer in decoder.layers:        # cached operands are never built inside a capture (WeightCache.get)
            layer.prepare_caches()
        self._capture()

    def _sync_flag(self, any_valid):
        dist.all_reduce(any_valid, op=dist.ReduceOp.MAX, group=self.group)
        if self.rank != 0:
            any_valid.fill_(1)

    def _capture(self):
        layers = list(self.dec.layers)
        pool = None
        outs = []

        def segment(fn):
            nonlocal pool
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                res = fn()
            if pool is None:
                pool = g.pool()
            self.graphs.append(g)
            return res

        # same layer chaining as DQDecoder.forward: layer l's fused chain B also emits layer l+1's query term
        fuse = getattr(self.dec, "fuse_next_query_term", False)
        for l, layer in enumerate(layers):
            layer._next_layer = (layers[l + 1],) if (fuse and l + 1 < len(layers)) else None
            layer._xw_in = None
        ref = self.ref if self.ref.dim() == 4 else self.ref[:, :, None]

        def first():
            # every layer's pyramid-side GEMMs (replicated on all ranks: they bound the strong scaling) are issued on
            # the side stream in this segment, next to layer 0's query-side kernels; the later segments find them done
            side = self.dec.fork_side_stream(self.tgt.device) if hasattr(self.dec, "fork_side_stream") else None
            if side is not None:
                self.dec.pack_pyramid(self.ctx, self.src, side)
                self.dec.launch_pyramid_projections(self.ctx, side, forked=True)
            else:
                self.ctx.pack(self.src)
            st0 = layers[0].forward_features(self.tgt, self.qpos, ref, self.ctx, self.thr)
            if side is not None:
                self.dec.join_pyramid_projections(side, keep_results=True)
            return st0
        st = segment(first)
        for l, layer in enumerate(layers):
            self.flags.append(st["any_valid"])
            last = l + 1 == len(layers)

            def body(layer=layer, st=st, last=last, l=l):
                o = layer.forward_triangulate(st, self.ctx)
                outs.append(o)
                if last:
                    tup = (torch.stack([x[0] for x in outs]), torch.stack([x[1] for x in outs]),
                           torch.stack([x[2] for x in outs]), torch.stack([x[3] for x in outs]), [x[4] for x in outs])
                    return pack_outputs(tup, self.NQ, self.J, self.world, self.rank, self.gather_hidden)
                return layers[l + 1].forward_features(o[0], self.qpos, o[1][:, :, None], self.ctx, self.thr)
            res = segment(body)
            if not last:
                st = res
        for layer in layers:
            layer._next_layer = None
            layer._xw_in = None
            layer._proj_in = None
            layer._after_chain_b = None
            layer.proj_attn._vp_event = None
        self.send, self.geo = res
        self.recv = self.send.new_empty((self.world * self.geo["nq_max"],) + tuple(self.send.shape[1:]))
        self.out = segment(lambda: unpack_outputs(self.recv, self.geo))

    def replay(self):
        self.graphs[0].replay()
        for l, flag in enumerate(self.flags):
            self._sync_flag(flag)
            self.graphs[l + 1].replay()
        dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        self.graphs[-1].replay()
        return self.out


class SpeculativeShardedDecoder:
    """The query-sharded forward as ONE HIP graph and two collectives per sample.

    The only reason the segmented runner talks between the layers is the reference's "no query valid anywhere -> force
    query (0,0)" rule (dq_decoder.py:620-623), a 4-byte MAX all-reduce per layer that is true ("some query is valid")
    in every layer of every sample a trained model sees.  This runner assumes it: every rank runs all layers with the
    flag set, keeping its LOCAL flags; a MAX all-reduce of the Ly local flags rides with the final all-gather, and only
    if some layer turns out to have had no valid query on any rank is the sample redone by the exact segmented runner
    (same static inputs).  Results are identical in both cases; the common case saves Ly collectives, Ly graph
    launches and Ly points where all ranks wait for the slowest one."""

    def __init__(self, decoder, tgt, reference_points, src_views, query_pos, ctx, threshold, NQ, group=None,
                 gather_hidden=False):
        self.exact = GraphedShardedDecoder(decoder, tgt, reference_points, src_views, query_pos, ctx, threshold, NQ,
                                           group=group, gather_hidden=gather_hidden)
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        dec, layers = decoder, list(decoder.layers)
        J = layers[0].num_joints
        self.flags = torch.zeros((len(layers),), dtype=torch.int32, device=tgt.device)
        self.fallbacks = 0
        fuse = getattr(dec, "fuse_next_query_term", False)
        ref = reference_points if reference_points.dim() == 4 else reference_points[:, :, None]

        def body():
            for l, layer in enumerate(layers):
                layer._next_layer = (layers[l + 1],) if (fuse and l + 1 < len(layers)) else None
                layer._xw_in = None
            side = dec.fork_side_stream(tgt.device) if hasattr(dec, "fork_side_stream") else None
            if side is not None:
                dec.pack_pyramid(ctx, src_views, side)
                dec.launch_pyramid_projections(ctx, side, forked=True, jit=True)
            else:
                ctx.pack(src_views)
            outs = []
            st = layers[0].forward_features(tgt, query_pos, ref, ctx, threshold)
            for l, layer in enumerate(layers):
                self.flags[l:l + 1].copy_(st["any_valid"])          # this rank's own answer, checked after the forward
                st["any_valid"].fill_(1)                            # the speculation: somebody has a valid query
                o = layer.forward_triangulate(st, ctx)
                outs.append(o)
                if l + 1 < len(layers):
                    st = layers[l + 1].forward_features(o[0], query_pos, o[1][:, :, None], ctx, threshold)
            if side is not None:
                dec.join_pyramid_projections(side)
            tup = (torch.stack([x[0] for x in outs]), torch.stack([x[1] for x in outs]), torch.stack([x[2] for x in outs]),
                   torch.stack([x[3] for x in outs]), [x[4] for x in outs])
            return pack_outputs(tup, NQ, J, self.world, self.rank, gather_hidden)

        try:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, pool=self.exact.graphs[0].pool()):
                self.send, self.geo = body()
        finally:
            for layer in layers:
                layer._next_layer = None
                layer._xw_in = None
                layer._proj_in = None
                layer._after_chain_b = None     # a hook left by a body that raised must not fire in a later forward
                layer.proj_attn._vp_event = None
        self.recv = self.send.new_empty((self.world * self.geo["nq_max"],) + tuple(self.send.shape[1:]))
        self.unpack = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.unpack, pool=self.exact.graphs[0].pool()):
            self.out = unpack_outputs(self.recv, self.geo)

    def replay(self):
        self.graph.replay()
        dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        dist.all_reduce(self.flags, op=dist.ReduceOp.MAX, group=self.group)
        if int(self.flags.min()) == 0:          # some layer had no valid query on any rank: redo the sample exactly
            self.fallbacks += 1
            return self.exact.replay()
        self.unpack.replay()
        return self.out

