"""mvgformer_amd.serving.GraphedDecoder: the decoder forward captured once as a HIP graph and replayed per frame -- bit-identical
to the eager forward, new frames / new cameras without re-capture, producer-in-place pyramid."""
import pytest
import torch

from mvgformer_amd.synthetic import build_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _eq(a, b):
    return all(torch.equal(x, y) for x, y in zip(a[:4], b[:4])) and all(torch.equal(x, y) for x, y in zip(a[4], b[4]))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
def test_graph_replay_equals_eager_and_follows_new_frames(dtype):
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    from mvgformer_amd.serving import GraphedDecoder
    cases = [case_to_device(build_case("mini5", seed=s, layers=2, valid_fraction=0.5), DEV) for s in (3, 4)]
    dec = build_decoder_for_case(cases[0], DEV, dtype=dtype)
    c = cases[0]
    run = GraphedDecoder(dec, c.meta, c.spatial_shapes, c.level_start_index, batch=1, num_queries=c.NQ, threshold=0.1)
    for i, c in enumerate(cases * 2):                                   # frames alternate: the graph must follow its inputs
        run.set_cameras(c.meta)
        run.load(src_views=c.src_views, tgt=c.tgt, query_pos=c.query_pos, reference_points=c.reference_points)
        got = run.replay()
        got = [t.clone() for t in got[:4]] + [[p.clone() for p in got[4]]]
        with torch.no_grad():
            want = dec(c.tgt, c.reference_points, c.src_views, c.meta, c.spatial_shapes, c.level_start_index, None,
                       query_pos=c.query_pos, threshold=0.1)
        torch.cuda.synchronize()
        assert _eq(got, want), (i, str(dtype))
        assert _eq(run.eager(), want)


def test_producer_in_place_pyramid_and_weight_refresh():
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    from mvgformer_amd.serving import GraphedDecoder
    c = case_to_device(build_case("mini5", seed=5, layers=2), DEV)
    dec = build_decoder_for_case(c, DEV, dtype=torch.bfloat16)
    run = GraphedDecoder(dec, c.meta, c.spatial_shapes, c.level_start_index, 1, c.NQ, 0.1, producer_writes_in_place=True)
    for dst, s in zip(run.pyramid_views, c.src_views):                  # the "backbone" writes channels-last bf16 in place
        dst.copy_(s)
    run.load(tgt=c.tgt, query_pos=c.query_pos, reference_points=c.reference_points)
    a = [t.clone() for t in run.replay()[:4]]
    with torch.no_grad():
        want = dec(c.tgt, c.reference_points, c.src_views, c.meta, c.spatial_shapes, c.level_start_index, None,
                   query_pos=c.query_pos, threshold=0.1)
    assert all(torch.equal(x, y) for x, y in zip(a, want[:4]))
    with torch.no_grad():                                               # an optimizer step / checkpoint load
        dec.layers[0].linear1.weight.mul_(1.5)
    b = [t.clone() for t in run.refresh_weights().replay()[:4]]
    with torch.no_grad():
        want2 = dec(c.tgt, c.reference_points, c.src_views, c.meta, c.spatial_shapes, c.level_start_index, None,
                    query_pos=c.query_pos, threshold=0.1)
    assert all(torch.equal(x, y) for x, y in zip(b, want2[:4])) and not torch.equal(b[0], a[0])


def test_graph_survives_an_eager_call_with_other_shapes_on_the_same_decoder():
    """the captured graph addresses ProjAttn._vp / _G and the cached operands by raw pointer.  An eager forward of
    the same decoder with another compute dtype / resolution re-allocates them; the runner pins what it captured and re-captures
    when the decoder has moved to other buffers -- results stay those of the eager forward."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    from mvgformer_amd.serving import GraphedDecoder
    c = case_to_device(build_case("mini5", seed=7, layers=2), DEV)
    dec = build_decoder_for_case(c, DEV, dtype=torch.bfloat16)
    run = GraphedDecoder(dec, c.meta, c.spatial_shapes, c.level_start_index, 1, c.NQ, 0.1)
    run.load(src_views=c.src_views, tgt=c.tgt, query_pos=c.query_pos, reference_points=c.reference_points)
    first = [t.clone() for t in run.replay()[:4]]
    held = [t.data_ptr() for t in run._pinned]
    assert held and all(l.proj_attn._vp.data_ptr() in held for l in dec.layers)
    # another scene through the same decoder object: other map shapes -> the per-layer buffers are re-allocated
    small = [s[:, :, : s.shape[2] // 2, :].contiguous() for s in c.src_views]
    shapes = torch.tensor([[s.shape[2], s.shape[3]] for s in small], dtype=torch.long, device=DEV)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    with torch.no_grad():
        dec(c.tgt, c.reference_points, small, c.meta, shapes, starts, None, query_pos=c.query_pos, threshold=0.1)
        junk = [torch.full((1 << 22,), 7.0, device=DEV) for _ in range(8)]      # recycle whatever was freed
    torch.cuda.synchronize()
    assert run._buffer_ptrs() != run._captured_ptrs
    again = [t.clone() for t in run.replay()[:4]]
    del junk
    assert all(torch.equal(x, y) for x, y in zip(first, again))


def test_product_launch_schedule_description(monkeypatch):
    """DQDecoder.pyramid_launches: which pyramid-product launches a bf16 forward issues (host logic only, no kernels).  One sample per
    forward: one launch per layer with one workgroup per CU (32 per XCD), layer l + 1's behind layer l's chain B; more samples per
    forward, or MVG_PYRAMID_JIT = 0: layer 0's launch + the remaining layers grouped; no side stream -> the same description (the
    launches then run inline)."""
    import types
    import torch
    from mvgformer_amd.factory import build_decoder_for_case
    from mvgformer_amd.synthetic import build_case
    case = build_case("cfg2", NQ=8, with_features=False)
    dec = build_decoder_for_case(case, "cpu", dtype=torch.bfloat16)
    ctx = lambda B: types.SimpleNamespace(B=B, feat=torch.zeros((case.V * B, 8, 256), dtype=torch.bfloat16))
    shape = lambda launches: [(len(g), s) for g, s in launches]
    n = len(dec.layers)
    assert n == 4 and dec.pyramid_jit == "auto" and dec.pyramid_jit_slots == 32
    assert dec._pyramid_jit(ctx(1)) and not dec._pyramid_jit(ctx(2))
    assert shape(dec.pyramid_launches(ctx(1))) == [(1, 32)] * n
    assert shape(dec.pyramid_launches(ctx(2))) == [(1, 0), (n - 1, 0)]
    assert shape(dec.pyramid_launches(ctx(1), jit=False)) == [(1, 0), (n - 1, 0)]          # segmented graphs (mvgformer_amd.dist)
    dec.pyramid_jit = "0"
    assert shape(dec.pyramid_launches(ctx(1))) == [(1, 0), (n - 1, 0)]
    dec.pyramid_jit = "1"
    assert shape(dec.pyramid_launches(ctx(2))) == [(1, 32)] * n
    dec.pyramid_jit = "auto"
    fp32 = types.SimpleNamespace(B=1, feat=torch.zeros((case.V, 8, 256), dtype=torch.float32))
    assert dec.pyramid_launches(fp32) is None                                              # fp32: one launch per layer (project_pyramid)
    # every layer of a launch is a distinct layer, in order
    seen = [l for g, _ in dec.pyramid_launches(ctx(2)) for l in g]
    assert [id(l) for l in seen] == [id(l) for l in dec.layers]
