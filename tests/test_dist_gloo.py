"""World-size-2 tests of the query-sharding exchange (mvgformer_amd.dist) on CPU with gloo.
The decoder kernels need a GPU, so a stand-in per-query 'decoder' is used: what is tested here is
the N>1 plumbing -- shard bounds, the one-buffer all-gather, uneven shards, the global
any-valid rule -- which is backend independent (nccl == RCCL on the GPU box)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvgformer_amd import dist as mdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Layer:
    num_joints = 15
    _any_valid_hook = None


class _FakeDecoder:
    """per-query deterministic function of the inputs with DQDecoder.forward's output tuple."""

    def __init__(self, Ly=3, V=2, C=8):
        self.layers = [_Layer() for _ in range(Ly)]
        self.Ly, self.V, self.C = Ly, V, C
        self.seen_any_valid = []

    def __call__(self, tgt, ref, src_views, meta, shapes, starts, ratios, query_pos=None, threshold=0.5, context=None):
        B, Lq, _ = tgt.shape
        J = 15
        NQ = Lq // J
        hs, refs, r2d, p2d, cls = [], [], [], [], []
        for l in range(self.Ly):
            h = tgt[..., :self.C] * (l + 1) + query_pos[..., :self.C]
            hs.append(h)
            refs.append(ref * (l + 2))
            base = ref[..., :2].unsqueeze(1).expand(B, self.V, Lq, 2)
            r2d.append(base + l)
            p2d.append(base - l)
            prob = torch.sigmoid(h.view(B, NQ, J, -1).mean((2, 3)))
            cls.append(torch.stack([1 - prob, prob], -1))
            any_valid = (prob > threshold).any().to(torch.int32).reshape(1)
            if self.layers[l]._any_valid_hook is not None:
                self.layers[l]._any_valid_hook(any_valid)
            self.seen_any_valid.append(int(any_valid))
        return torch.stack(hs), torch.stack(refs), torch.stack(r2d), torch.stack(p2d), cls


def _worker(rank, world, port, NQ, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        B, J, C = 2, 15, 8
        tgt = torch.randn(B, NQ * J, C)
        pos = torch.randn(B, NQ * J, C)
        ref = torch.randn(B, NQ * J, 3)
        dec = _FakeDecoder()
        full = dec(tgt, ref, None, None, None, None, None, query_pos=pos, threshold=0.5)
        got = mdist.sharded_decoder_forward(_FakeDecoder(), tgt, ref, None, None, None, None, pos, 0.5,
                                            gather_hidden=True)
        ok = all(torch.equal(a, b) for a, b in zip(got[:4], full[:4]))
        ok = ok and all(torch.equal(a, b) for a, b in zip(got[4], full[4]))
        got2 = mdist.sharded_decoder_forward(_FakeDecoder(), tgt, ref, None, None, None, None, pos, 0.5)
        ok = ok and got2[0] is None and torch.equal(got2[1], full[1])
        # global any-valid: only rank 1's block has a valid query -> rank 0 must NOT force (0,0)
        d3 = _FakeDecoder(Ly=1)
        t3 = torch.full((1, NQ * J, C), -5.0)
        lo1, hi1 = mdist.shard_bounds(NQ, world, world - 1)
        t3[:, lo1 * J:hi1 * J] = 5.0
        mdist.sharded_decoder_forward(d3, t3, ref[:1], None, None, None, None, torch.zeros_like(t3), 0.5)
        ok = ok and d3.seen_any_valid == [1]
        # nobody valid anywhere: rank 0 sees 0 (forces query 0), the others are told 1
        d4 = _FakeDecoder(Ly=1)
        mdist.sharded_decoder_forward(d4, torch.full((1, NQ * J, C), -5.0), ref[:1], None, None, None, None,
                                      torch.zeros(1, NQ * J, C), 0.5)
        ok = ok and d4.seen_any_valid == [0 if rank == 0 else 1]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("NQ", [8, 7])   # even and uneven shards
def test_query_sharding_world2_gloo(NQ):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, NQ, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)], res


def test_shard_bounds_cover_everything():
    for NQ in (1, 5, 8, 1024, 1023):
        for world in (1, 2, 3, 8):
            spans = [mdist.shard_bounds(NQ, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == NQ
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
