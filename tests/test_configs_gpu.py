"""GPU parity at the REAL workloads of BASELINE.json's configurations (`pytest -m gpu`), through the C ABI.

  cfg-2  Panoptic 5 views / 1024 queries / 4 layers      : all 4 layers, fp32 and bf16, vs the fp64 oracle
  cfg-3  the same sample as 8 query shards of 128 queries : bf16, concatenation vs the single-rank run (bit-exact)
  cfg-4  Shelf 5 views / 512 queries / fp32 / 4 layers    : maps (152,200)/(76,100)/(38,50), k = p = 0, vs the fp64 oracle
  cfg-5  31 views / 2048 queries / 6 layers               : V = 31 and 6 layers against the fp64 oracle at as many queries
         as the host oracle affords (128), and the full 2048-query forward through size-independent properties

Every multi-layer comparison has two parts.

  * TEACHER-FORCED, every layer at the strict bars: layer l is run on the fp64 oracle's outputs of layer l-1, so each
    layer's kernels are held to  features 2e-4, 2D 0.05 px, 3D 0.1 mm  (fp32)  /  features 6e-2, 2D 0.1 px, 3D 6 mm  (bf16).
  * FREE-RUNNING, all layers chained on the device as in production.  This decoder is an iterated map on white-noise
    feature maps with random weights: a perturbation of the 3D points moves next layer's sampling locations on maps that
    have no spatial smoothness, so rounding differences are AMPLIFIED from layer to layer (measured x5-10 per layer at 5
    views, damped at 31 views) -- by the arithmetic of the reference itself just the same.  The free-running bars are
    therefore relative to a yardstick computed in the test: the fp32 path must stay closer to the fp64 truth than the
    oracle evaluated in fp32 (= the reference's own arithmetic, fp32 SVD included) does; the bf16 path is reported
    next to an fp64 evaluation on a bf16-rounded pyramid (the perturbation any bf16 implementation starts from).

Why the truth is the oracle in FP64: the reference's fp32 SVD of the un-normalised DLT rows carries millimetres of
conditioning noise on these scenes (tests/test_oracle_golden.py::test_reference_fp32_dlt_noise), which feeds the next
layer's projection; the fp64 evaluation of the same algorithm (oracle/decoder_ref.py, pinned to the reference's outputs
by tests/test_oracle_golden.py) does not.  Measured errors are printed (`pytest -s`) and recorded in DESIGN.md section 5."""
import os

import pytest
import torch

from mvgformer_amd.synthetic import build_case, to_torch_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_ORACLE_CACHE = {}


@pytest.fixture(scope="module")
def O():
    from oracle import decoder_ref
    return decoder_ref


def _oracle(O, case, key, dtype=torch.float64, src_views=None):
    """oracle forward of all layers, cached per (case key, variant) across the parametrisations of a test"""
    if key not in _ORACLE_CACHE:
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        prm = to_torch_state(case.weights)
        with torch.no_grad():
            out = O.decoder_forward(prm, case.layers, case.tgt.cpu(), case.reference_points.cpu(),
                                    [s.cpu() for s in (src_views or case.src_views)],
                                    [{k: ({kk: vv.cpu() for kk, vv in v.items()} if isinstance(v, dict) else v.cpu())
                                      for k, v in m.items()} for m in case.meta],
                                    case.spatial_shapes.cpu(), case.level_start_index.cpu(), case.query_pos.cpu(),
                                    case.img_size, threshold=0.1, dtype=dtype)
        _ORACLE_CACHE[key] = [t.double() if torch.is_tensor(t) else [c.double() for c in t] for t in out]
    return _ORACLE_CACHE[key]


def _run(dec, g):
    with torch.no_grad():
        out = dec(g.tgt, g.reference_points, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None,
                  query_pos=g.query_pos, threshold=0.1)
    torch.cuda.synchronize()
    return out


def _stats(got, want):
    """per layer: features (max, q99.9), 2D px (max, q99.9), 3D mm (max, q99.9, median) against `want`"""
    hs, refs, r2d = (torch.as_tensor(got[i]).cpu().double() for i in range(3))
    rows = []
    q = lambda t, p: float(torch.quantile(t.flatten()[:: max(1, t.numel() // 4_000_000)], p))
    for l in range(hs.shape[0]):
        eh = (hs[l] - want[0][l]).abs().amax(-1)
        ep = (r2d[l] - want[2][l]).abs().amax(-1)
        em = (refs[l] - want[1][l]).norm(dim=-1)
        rows.append(dict(hs=float(eh.max()), hs_q=q(eh, 0.999), px=float(ep.max()), px_q=q(ep, 0.999),
                         mm=float(em.max()), mm_q=q(em, 0.999), mm_med=q(em, 0.5)))
    return rows


def _report(tag, rows):
    for l, r in enumerate(rows):
        print("%-34s layer %d: |hs| max %.2e q99.9 %.2e | 2D px max %.2e q99.9 %.2e | 3D mm max %.4f q99.9 %.4f median %.4f"
              % (tag, l, r["hs"], r["hs_q"], r["px"], r["px_q"], r["mm"], r["mm_q"], r["mm_med"]))


def _teacher_forced(dec, g, want, tag, tol_hs, tol_px, tol_mm, tol_cls):
    """every layer on the fp64 oracle's previous-layer outputs, at the strict per-layer bars"""
    rows = []
    for l, layer in enumerate(dec.layers):
        tgt = g.tgt if l == 0 else want[0][l - 1].float().to(DEV)
        ref = g.reference_points if l == 0 else want[1][l - 1].float().to(DEV)
        with torch.no_grad():
            o = layer(tgt, g.query_pos, ref[:, :, None], g.src_views, g.spatial_shapes, g.level_start_index, g.meta,
                      threshold=0.1)
        torch.cuda.synchronize()
        assert torch.equal(o[1].cpu().abs().sum(-1) > 0, want[1][l].abs().sum(-1) > 0), "validity pattern %s layer %d" % (tag, l)
        st = _stats([o[0][None], o[1][None], o[2][None]], [want[0][l][None], want[1][l][None], want[2][l][None]])[0]
        st["cls"] = float((o[4].cpu().double() - want[4][l]).abs().max())
        rows.append(st)
    _report(tag + " teacher-forced", rows)
    print("%-34s class prob max %.2e" % (tag + " teacher-forced", max(r["cls"] for r in rows)))
    for l, r in enumerate(rows):
        assert r["hs"] < tol_hs and r["px"] < tol_px and r["mm"] < tol_mm and r["cls"] < tol_cls, (tag, l, r)
    return rows


FP32_BARS = (2e-4, 0.05, 0.1, 1e-5)      # features, 2D px, 3D mm, class prob -- per layer, teacher-forced
BF16_BARS = (6e-2, 0.1, 6.0, 2e-2)
# free-running bf16 (q99.9 features, q99.9 2D px, q99.9 3D mm, median 3D mm)
BF16_FREE_CFG2 = (0.3, 5.0, 30.0, 0.3)      # measured at layer 3: 0.175 / 2.75 px / 17.3 mm / 0.17 mm
BF16_FREE_CFG5 = (0.06, 0.7, 3.0, 0.12)     # measured at layer 5: 0.033 / 0.38 px / 1.79 mm / 0.07 mm


def _free_running_fp32(tag, got, want, yard):
    """fp32 path chained over all layers: closer to the fp64 truth than the fp32 oracle (the reference's arithmetic) is"""
    assert all(torch.isfinite(t).all() for t in got[:4])
    assert torch.equal(got[1].cpu().abs().sum(-1) > 0, want[1].abs().sum(-1) > 0), "validity pattern (%s)" % tag
    ours, ref32 = _stats(got, want), _stats(yard, want)
    _report(tag + " free-running", ours)
    _report(tag + " fp32 ORACLE vs fp64", ref32)
    for l, (a, b) in enumerate(zip(ours, ref32)):
        assert a["mm_q"] <= max(b["mm_q"], 0.02) and a["mm"] <= max(b["mm"], 0.1), (tag, l, a, b)
        assert a["hs_q"] <= max(2.0 * b["hs_q"], 2e-4) and a["px_q"] <= max(2.0 * b["px_q"], 0.05), (tag, l, a, b)
        assert a["hs"] < 5e-2 and a["px"] < 1.0 and a["mm"] < 2.0, (tag, l, a)       # absolute sanity bars
    return ours


def _free_running_bf16(tag, got, want, yard, bars):
    """bf16 path chained over all layers, reported next to the fp64 evaluation on a bf16-rounded pyramid"""
    assert all(torch.isfinite(t).all() for t in got[:4])
    assert torch.equal(got[1].cpu().abs().sum(-1) > 0, want[1].abs().sum(-1) > 0), "validity pattern (%s)" % tag
    ours, yd = _stats(got, want), _stats(yard, want)
    _report(tag + " free-running", ours)
    _report(tag + " fp64 on bf16 PYRAMID", yd)
    e_cls = max(float((c.cpu().double() - w).abs().max()) for c, w in zip(got[4], want[4]))
    print("%-34s class prob max %.2e" % (tag + " free-running", e_cls))
    # Relative bar: at most 6x the yardstick's error (measured 1.6-3.6x: this path rounds the value planes, G, the sampled
    # rows and every activation to bf16 as well, not only the pyramid), with the teacher-forced bars as the floor.
    # Absolute bars (q99.9 features / q99.9 2D px / q99.9 3D mm / median 3D mm) from the measured values, x ~1.7.
    tol_hs_q, tol_px_q, tol_mm_q, tol_mm_med = bars
    for l, (a, y) in enumerate(zip(ours, yd)):
        assert a["hs_q"] <= max(6 * y["hs_q"], 6e-2) and a["px_q"] <= max(6 * y["px_q"], 0.1), (tag, l, a, y)
        assert a["mm_q"] <= max(6 * y["mm_q"], 6.0) and a["mm_med"] <= max(6 * y["mm_med"], 0.5), (tag, l, a, y)
        assert a["hs_q"] < tol_hs_q and a["px_q"] < tol_px_q and a["mm_q"] < tol_mm_q and a["mm_med"] < tol_mm_med, (tag, l, a)
    return ours


def _bf16_pyramid(case):
    return [s.to(torch.bfloat16).float() for s in case.src_views]


def test_cfg4_shelf_full_workload_fp32_vs_fp64_oracle(O):
    """BASELINE configs[3]: Shelf geometry (800x608 network image -> maps (152,200)/(76,100)/(38,50); k = p = 0 as in
    data/Shelf/calibration_shelf.json), 5 views, 512 queries x 15 joints, fp32, all 4 layers."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg4", seed=4)
    assert case.shapes == [(152, 200), (76, 100), (38, 50)] and case.V == 5 and case.NQ == 512 and case.layers == 4
    want = _oracle(O, case, "cfg4/f64")
    yard = _oracle(O, case, "cfg4/f32", dtype=torch.float32)
    dec = build_decoder_for_case(case, DEV, dtype=torch.float32)
    g = case_to_device(case, DEV)
    _teacher_forced(dec, g, want, "cfg4 fp32", *FP32_BARS)
    _free_running_fp32("cfg4 fp32", _run(dec, g), want, yard)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_cfg2_four_layers_vs_fp64_oracle(dtype, O):
    """BASELINE configs[1], the forward bench.py times: 5 views, 1024 queries x 15 joints, maps (128,240)/(64,120)/(32,60),
    ALL 4 layers, against the fp64 oracle: teacher-forced per layer at the strict bars, then free-running."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg2", seed=1)
    assert case.V == 5 and case.NQ == 1024 and case.layers == 4
    want = _oracle(O, case, "cfg2/f64")
    dec = build_decoder_for_case(case, DEV, dtype=dtype)
    if dtype == torch.float32:
        yard = _oracle(O, case, "cfg2/f32", dtype=torch.float32)
        g = case_to_device(case, DEV)
        _teacher_forced(dec, g, want, "cfg2 fp32", *FP32_BARS)
        _free_running_fp32("cfg2 fp32", _run(dec, g), want, yard)
    else:
        yard = _oracle(O, case, "cfg2/f64-bf16pyr", src_views=_bf16_pyramid(case))
        g = case_to_device(case, DEV)
        _teacher_forced(dec, g, want, "cfg2 bf16", *BF16_BARS)
        _free_running_bf16("cfg2 bf16", _run(dec, g), want, yard, BF16_FREE_CFG2)


@pytest.mark.parametrize("features", ["smooth", "white"])
def test_cfg2_queries_inside_every_view_bf16_maximum_error_is_bounded(features, O):
    """a workload on which the bf16 path's MAXIMUM error can be bounded.  The growth of rounding
    differences over the layers in the default synthetic workload comes from its geometry, not from the arithmetic: 41 % of
    the (view, query) pairs project outside their image there (few-view, ill-conditioned triangulations; in-image masks
    that flip with a sub-millimetre move of a point).  With the initial grid over the central 30 % of the space every
    query is inside every view (> 99 % of the pairs) -- where a trained model's queries sit, on the people -- and the fp64
    oracle itself is stable (a 0.05-mm perturbation of the input stays 0.19 mm over 4 layers).  On that workload, cfg-2 full
    size, 4 layers FREE-RUNNING, bf16 against the fp64 oracle: max 3D <= 0.5 mm, max 2D <= 0.1 px (the verdict asked for
    2 mm / 0.25 px; measured on MI355X: 0.15 mm / 0.026 px on band-limited maps -- Gaussian low-pass, 8 cells at level 0 --,
    0.053 mm / 0.012 px on white noise; fp32: 0.003 mm / 5e-4 px), i.e. 1 % of the published 16.0 mm MPJPE."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg2", seed=1, ref_extent=0.3, smooth_sigma0=8.0 if features == "smooth" else None)
    want = _oracle(O, case, "cfg2-inside-%s/f64" % features)
    g = case_to_device(case, DEV)
    inside = []
    for dtype, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        dec = build_decoder_for_case(case, DEV, dtype=dtype)
        got = _run(dec, g)
        assert torch.equal(got[1].cpu().abs().sum(-1) > 0, want[1].abs().sum(-1) > 0)
        rows = _stats(got, want)
        _report("cfg2 all-inside %s %s" % (features, tag), rows)
        e_cls = max(float((c.cpu().double() - w).abs().max()) for c, w in zip(got[4], want[4]))
        print("%-34s class prob max %.2e" % ("cfg2 all-inside %s %s" % (features, tag), e_cls))
        for l, r in enumerate(rows):
            if dtype == torch.bfloat16:
                assert r["mm"] <= 0.5 and r["px"] <= 0.1 and r["hs"] <= 6e-2, (features, l, r)        # MAXIMUM over all 15 360 tokens
            else:
                assert r["mm"] <= 0.01 and r["px"] <= 2e-3 and r["hs"] <= 2e-4, (features, l, r)
        assert e_cls < (2e-2 if dtype == torch.bfloat16 else 1e-5)
    proj = got[3][0].cpu()                 # layer-0 projections (network-image px) of the initial poses: inside the image
    w, h = case.img_size
    frac = float(((proj[..., 0] >= 0) & (proj[..., 0] < w) & (proj[..., 1] >= 0) & (proj[..., 1] < h)).float().mean())
    print("in-image fraction of the layer-0 pairs: %.4f" % frac)
    assert frac > 0.98


def test_cfg3_eight_query_shards_bf16_equal_the_single_rank_run():
    """BASELINE configs[2] on one GPU: the cfg-2 sample as 8 shards of 128 person-queries (what each of 8 ranks runs:
    128-thread sampling workgroups, 32-row chain-B tiles, single-workgroup binning), bf16, 4 layers free-running; the
    concatenated shard outputs equal the single-rank run BIT FOR BIT: no kernel's arithmetic depends on where a query sits
    in the launch (chain B rotates the wavefront -> column-group assignment per tile, not the k-step order)."""
    from mvgformer_amd.decoder import DecoderContext
    from mvgformer_amd.dist import shard_queries
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg2", seed=1)
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    g = case_to_device(case, DEV)
    full = _run(dec, g)
    ctx = DecoderContext.prepare(g.spatial_shapes, g.level_start_index, g.meta, case.img_size, torch.bfloat16, 1, DEV)
    parts = []
    for rank in range(8):
        t, p, r, (lo, hi) = shard_queries(g.tgt, g.query_pos, g.reference_points, 15, 8, rank)
        assert hi - lo == 128
        with torch.no_grad():
            parts.append(dec(t, r, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None, query_pos=p,
                             threshold=0.1, context=ctx))
    torch.cuda.synchronize()
    hs = torch.cat([o[0] for o in parts], 2)
    refs = torch.cat([o[1] for o in parts], 2)
    r2d = torch.cat([o[2] for o in parts], 3)
    p2d = torch.cat([o[3] for o in parts], 3)
    cls = [torch.cat([o[4][l] for o in parts], 1) for l in range(case.layers)]
    print("cfg3 8 x 128 queries vs single rank (bf16, 4 layers): |hs| %.2e  2D %.2e px  3D %.4f mm"
          % (float((hs - full[0]).abs().max()), float((r2d - full[2]).abs().max()), float((refs - full[1]).norm(dim=-1).max())))
    assert torch.equal(hs, full[0]) and torch.equal(refs, full[1]) and torch.equal(r2d, full[2]) and torch.equal(p2d, full[3])
    assert all(torch.equal(a, b) for a, b in zip(cls, full[4]))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_cfg5_31_views_six_layers_vs_fp64_oracle(dtype, O):
    """BASELINE configs[4] geometry with ALL 31 views (four passes of the 8-lane view loops in the view mean, the view
    softmax and the DLT rows; 31 images through binning / sampling / chain A) and ALL 6 layers (the reference crashes past
    4, dq_decoder.py:94,1142), 128 queries (what the host oracle affords in well under a minute), vs the fp64 oracle."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg5", seed=2, NQ=128)
    assert case.V == 31 and case.layers == 6
    want = _oracle(O, case, "cfg5q128/f64")
    dec = build_decoder_for_case(case, DEV, dtype=dtype)
    if dtype == torch.float32:
        yard = _oracle(O, case, "cfg5q128/f32", dtype=torch.float32)
        g = case_to_device(case, DEV)
        _teacher_forced(dec, g, want, "cfg5 V=31 L=6 fp32", *FP32_BARS)
        _free_running_fp32("cfg5 V=31 L=6 fp32", _run(dec, g), want, yard)
    else:
        yard = _oracle(O, case, "cfg5q128/f64-bf16pyr", src_views=_bf16_pyramid(case))
        g = case_to_device(case, DEV)
        _teacher_forced(dec, g, want, "cfg5 V=31 L=6 bf16", *BF16_BARS)
        _free_running_bf16("cfg5 V=31 L=6 bf16", _run(dec, g), want, yard, BF16_FREE_CFG5)


def test_cfg5_full_stress_forward_properties():
    """BASELINE configs[4] at FULL size -- 31 views, 2048 queries x 15 joints (30 720 tokens per image: the 8-workgroup
    binning at 31 images, 952 320 pairs per sampling launch), 6 layers, bf16 -- through the size-independent properties
    the domain offers (person-queries are independent units), all BIT-exact over the 6 free-running
    layers: two runs agree; a permutation of the person-queries permutes the outputs; shard-concatenation equals the
    full run."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg5", seed=3)
    assert case.V == 31 and case.NQ == 2048 and case.layers == 6
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    g = case_to_device(case, DEV)
    NQ, J = case.NQ, 15
    run = lambda t, p, r: dec(t, r, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None, query_pos=p,
                              threshold=0.1)
    with torch.no_grad():
        a = run(g.tgt, g.query_pos, g.reference_points)
        b = run(g.tgt, g.query_pos, g.reference_points)
        torch.cuda.synchronize()
        assert all(torch.isfinite(t).all() for t in a[:4])
        for x, y in zip(a[:4], b[:4]):
            assert torch.equal(x, y)
        assert int((a[1].abs().sum(-1) > 0).sum()) == case.layers * NQ * J            # every query is valid in this case
        perm = torch.randperm(NQ, generator=torch.Generator().manual_seed(0)).to(DEV)
        tok = (perm[:, None] * J + torch.arange(J, device=DEV)[None]).reshape(-1)
        pm = run(g.tgt[:, tok].contiguous(), g.query_pos[:, tok].contiguous(), g.reference_points[:, tok].contiguous())
        half = NQ // 2 * J
        s0 = run(g.tgt[:, :half].contiguous(), g.query_pos[:, :half].contiguous(), g.reference_points[:, :half].contiguous())
        s1 = run(g.tgt[:, half:].contiguous(), g.query_pos[:, half:].contiguous(), g.reference_points[:, half:].contiguous())
        torch.cuda.synchronize()
    cat = [torch.cat([s0[0], s1[0]], 2), torch.cat([s0[1], s1[1]], 2), torch.cat([s0[2], s1[2]], 3)]
    for name, got, ref in (("permuted", (pm[0], pm[1], pm[2]), (a[0][:, :, tok], a[1][:, :, tok], a[2][:, :, :, tok])),
                           ("2 shards", cat, (a[0], a[1], a[2]))):
        print("cfg5 full (31 views, 2048 q, 6 layers, bf16) %s vs full run: |hs| %.2e  2D %.2e px  3D %.4f mm"
              % (name, float((got[0] - ref[0]).abs().max()), float((got[2] - ref[2]).abs().max()),
                 float((got[1] - ref[1]).norm(dim=-1).max())))
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), name


@pytest.mark.parametrize("g_form", [False, True], ids=["gather_form", "g_form_fused_chains"])
def test_cfg5_full_stress_fp32_permutation_and_shards_are_exact(g_form):
    """The fp32 path (reference arithmetic, no position-dependent rounding anywhere) at cfg-5's full size, 2 of its layers:
    query permutation and shard concatenation are BIT-exact -- in the gather form (unfused chain A + the fused fp32 chain B) and
    in the G-sampling form (one-pass pyramid products, fused fp32 chains A and B: csrc/f32s.hip)."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg5", seed=3, layers=2)
    dec = build_decoder_for_case(case, DEV, dtype=torch.float32)
    for layer in dec.layers:            # one sampling form for every query count (see ProjAttn.g_sampling_f32)
        layer.proj_attn.g_sampling_f32 = g_form
    g = case_to_device(case, DEV)
    NQ, J = case.NQ, 15
    run = lambda t, p, r: dec(t, r, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None, query_pos=p,
                              threshold=0.1)
    with torch.no_grad():
        a = run(g.tgt, g.query_pos, g.reference_points)
        perm = torch.randperm(NQ, generator=torch.Generator().manual_seed(1)).to(DEV)
        tok = (perm[:, None] * J + torch.arange(J, device=DEV)[None]).reshape(-1)
        pm = run(g.tgt[:, tok].contiguous(), g.query_pos[:, tok].contiguous(), g.reference_points[:, tok].contiguous())
        half = NQ // 2 * J
        s0 = run(g.tgt[:, :half].contiguous(), g.query_pos[:, :half].contiguous(), g.reference_points[:, :half].contiguous())
        s1 = run(g.tgt[:, half:].contiguous(), g.query_pos[:, half:].contiguous(), g.reference_points[:, half:].contiguous())
        torch.cuda.synchronize()
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()
    assert torch.equal(pm[0], a[0][:, :, tok]) and torch.equal(pm[1], a[1][:, :, tok]) and torch.equal(pm[2], a[2][:, :, :, tok])
    assert torch.equal(torch.cat([s0[0], s1[0]], 2), a[0]) and torch.equal(torch.cat([s0[1], s1[1]], 2), a[1])
    assert torch.equal(torch.cat([s0[2], s1[2]], 3), a[2])


# ------------------------------------------------------------------------------------------ round-2 boundary items
def test_deform_forward_backward_float64(O):
    """AT_DISPATCH_FLOATING_TYPES (deform_cuda.cu:75,145): the drop-in accepts double like the reference op.  Forward against
    the reference twin's own fp64 outputs (tests/golden/msda.npz "<case>/out_f64") at 1e-12, backward against fp64 autograd
    of the pinned oracle, and torch.autograd.gradcheck through DeformFunction as one would run it on the reference op."""
    import numpy as np
    from mvgformer_amd import deformable
    from mvgformer_amd.functions import DeformFunction
    from tests.golden.cases import msda_case
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msda.npz"))
    for name in ("small_f32", "ragged_f32", "edge_f32"):
        c = msda_case(name)
        value, loc, w = c["value"].double(), c["loc"].double(), c["weight"].double()
        shapes, starts = c["shapes"], c["starts"]
        dev = lambda t: t.to(DEV)
        out = deformable.deform_forward(dev(value), dev(shapes), dev(starts), dev(loc), dev(w), 64).cpu()
        assert out.dtype == torch.float64
        ref = torch.from_numpy(z[name + "/out_f64"]).double()
        assert float((out - ref).abs().max()) < 1e-12 * max(1.0, float(ref.abs().max())), name
        v, l_, w_ = value.clone().requires_grad_(), loc.clone().requires_grad_(), w.clone().requires_grad_()
        go = torch.randn(ref.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
        O.msda_forward(v, shapes, starts, l_, w_).backward(go)
        gv, gl, ga = deformable.deform_backward(dev(value), dev(shapes), dev(starts), dev(loc), dev(w), dev(go), 64)
        for got, want in ((gv, v.grad), (gl, l_.grad), (ga, w_.grad)):
            assert got.dtype == torch.float64
            assert float((got.cpu() - want).abs().max()) < 1e-11 * max(1.0, float(want.abs().max())), name
    c = msda_case("edge_f32")
    value, loc, w = (c[k].double().to(DEV) for k in ("value", "loc", "weight"))
    loc = (loc * 0.8 + 0.1).contiguous()                     # away from the cell borders, where the op is not differentiable
    shapes, starts = c["shapes"].to(DEV), c["starts"].to(DEV)
    w = w.clone().requires_grad_()
    value = value.clone().requires_grad_()
    assert torch.autograd.gradcheck(lambda vv, ww: DeformFunction.apply(vv, shapes, starts, loc, ww, 64), (value, w),
                                    eps=1e-6, atol=1e-7, nondet_tol=1e-9)
    with pytest.raises(RuntimeError):                         # mixed dtypes are refused, not converted silently
        deformable.deform_forward(value.detach(), shapes, starts, loc.float(), w.detach(), 64)


def test_prepared_context_is_repacked_for_every_frame():
    """One DecoderContext.prepare()d context reused across frames (static cameras) with NEW src_views each call: the
    second forward must sample the second frame's pyramid (round-1 advisory: it silently reused the first)."""
    from mvgformer_amd.decoder import DecoderContext
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    for dt in (torch.float32, torch.bfloat16):
        case = build_case("mini5", seed=3, layers=2)
        dec = build_decoder_for_case(case, DEV, dtype=dt)
        g = case_to_device(case, DEV)
        frame2 = [s.flip(0).contiguous() * 0.5 + 0.25 for s in g.src_views]
        ctx = DecoderContext.prepare(g.spatial_shapes, g.level_start_index, g.meta, case.img_size, dt, 1, DEV)
        fwd = lambda src, c: dec(g.tgt, g.reference_points, src, g.meta, g.spatial_shapes, g.level_start_index, None,
                                 query_pos=g.query_pos, threshold=0.1, context=c)
        with torch.no_grad():
            a1 = fwd(g.src_views, ctx)
            a2 = fwd(frame2, ctx)                      # same context, different frame
            b2 = fwd(frame2, None)                     # fresh context
        assert torch.equal(a2[0], b2[0]) and torch.equal(a2[1], b2[1])
        assert not torch.equal(a1[0], a2[0])
        # channels-last producer tensors go the same way
        nhwc = [s.to(dt).to(memory_format=torch.channels_last) for s in frame2]
        with torch.no_grad():
            c2 = fwd(nhwc, ctx)
        if dt == torch.bfloat16:
            assert torch.equal(c2[0], a2[0])


def test_weight_cache_refuses_to_build_inside_a_graph_capture():
    """An operand cache entry created during HIP-graph capture would stay unwritten until the first replay; the cache
    raises instead, and DQDecoderLayer.prepare_caches() is the way to fill it ahead of a capture."""
    from mvgformer_amd.decoder import DecoderContext
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("mini5", seed=3, layers=2)
    g = case_to_device(case, DEV)
    ctx = DecoderContext.prepare(g.spatial_shapes, g.level_start_index, g.meta, case.img_size, torch.bfloat16, 1, DEV)
    fwd = lambda dec: dec(g.tgt, g.reference_points, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None,
                          query_pos=g.query_pos, threshold=0.1, context=ctx)
    cold = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    cold.overlap_pyramid = False        # the side-stream fork prepares the caches itself; here nothing does
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="prepare_caches"):
            with torch.cuda.graph(graph):
                fwd(cold)
    torch.cuda.synchronize()
    warm = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    warm.overlap_pyramid = False
    for layer in warm.layers:
        layer.prepare_caches()
    with torch.no_grad():
        want = fwd(build_decoder_for_case(case, DEV, dtype=torch.bfloat16))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            got = fwd(warm)
        graph.replay()
        torch.cuda.synchronize()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_train_mode_without_autograd_applies_dropout():
    """Under no_grad in train() mode the reference applies dropout2/3/4; the native path is the eval()-mode layer, so a
    training-mode layer is routed to the differentiable torch path (which applies them) instead of silently skipping."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("mini5", seed=3, layers=1)
    dec = build_decoder_for_case(case, DEV, dtype=torch.float32)
    g = case_to_device(case, DEV)
    layer = dec.layers[0]
    call = lambda: layer(g.tgt, g.query_pos, g.reference_points[:, :, None], g.src_views, g.spatial_shapes,
                         g.level_start_index, g.meta, threshold=0.1)
    with torch.no_grad():
        ev = call()
        layer.train()
        torch.manual_seed(0)
        tr1 = call()
        tr2 = call()
        for m in (layer.dropout2, layer.dropout3, layer.dropout4):
            m.p = 0.0
        tr0 = call()                       # train() mode but nothing to drop: the native path again
        layer.eval()
    assert not torch.equal(tr1[0], ev[0]) and not torch.equal(tr1[0], tr2[0])
    assert torch.equal(tr0[0], ev[0])


def test_triangulation_launch_also_projects_for_the_next_layer():
    """mvg_triangulate_project: the next layer's (r, ref_lvl, inside) written by the triangulation launch are bit-identical to
    mvg_project run on the new reference points (zeros for queries that did not pass included), and the triangulated points
    equal mvg_triangulate's; the decoder with and without the fused boundary gives identical outputs."""
    from mvgformer_amd import ops
    from mvgformer_amd.decoder import DecoderContext
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    from tests.golden.cases import LAYER_CASES
    for cname in ("mini5_half", "mini5_b2"):
        sp = LAYER_CASES[cname]
        case = build_case(sp["config"], B=sp.get("B", 1), seed=sp["seed"], NQ=sp.get("NQ"), layers=sp["layers"],
                          valid_fraction=sp.get("valid_fraction"))
        g = case_to_device(case, DEV)
        B, NQ, J, V = case.B, case.NQ, 15, case.V
        ctx = DecoderContext.build(g.src_views, g.spatial_shapes, g.level_start_index, g.meta, case.img_size, torch.float32, B)
        gen = torch.Generator().manual_seed(4)
        X = g.reference_points.reshape(B, NQ * J, 3).contiguous()
        r, ref_lvl, inside = ops.project(X, ctx.cams, ctx.levels, V, B)
        o = (torch.randn((V * B * NQ * J, 3), generator=gen) * torch.tensor([3.0, 3.0, 1.0])).to(DEV)
        valid = (torch.rand((B, NQ), generator=gen) > 0.4).to(torch.uint8).to(DEV)
        flag = torch.ones((1,), dtype=torch.int32, device=DEV)
        a = ops.triangulate(r, o, ctx.cams, valid, flag, V, B, NQ, J)
        b = ops.triangulate(r, o, ctx.cams, valid, flag, V, B, NQ, J, next_levels=ctx.levels)
        assert all(torch.equal(x, y) for x, y in zip(a, b[:3]))
        want = ops.project(b[0], ctx.cams, ctx.levels, V, B)
        assert all(torch.equal(x, y) for x, y in zip(want, b[3]))
        assert int((b[0].abs().sum(-1) == 0).sum()) >= int((valid == 0).sum()) * J       # masked queries project the origin
        for dt in (torch.float32, torch.bfloat16):
            dec = build_decoder_for_case(case, DEV, dtype=dt)
            fused = _run(dec, g)
            for layer in dec.layers:
                layer.fuse_boundary = False            # separate mvg_project launch per layer
            plain = _run(dec, g)
            for k in range(4):
                assert torch.equal(fused[k], plain[k]), (cname, str(dt), k)


def test_deterministic_backward_full_size_view_layer_vs_c_oracle():
    """mvg_msda_backward_det_f32 (csrc/msda_bwd.hip) at the size of ONE cfg-2 view-layer (S = 40 320 pixels, 15 360 tokens x 8
    heads x 24 samples = 2.9 M samples, 11.8 M corner contributions of 32 channels), locations spread over and beyond the maps
    so that every bin kind occurs (interior, map border, empty, heavy): against the C restatement of the reference's backward
    accumulated in double (oracle/msda_ref.c, pinned by tests/test_oracle_golden.py), bit-reproducible run to run, and
    against the atomic form of round 1 (the reference's own scheme)."""
    import time
    from mvgformer_amd import ops
    from oracle import msda_c
    shapes = torch.tensor([(128, 240), (64, 120), (32, 60)], dtype=torch.long)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    N, M, D, Lq, P, L = 1, 8, 32, 15360, 8, 3
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    g = torch.Generator().manual_seed(21)
    value = torch.randn((N, S, M, D), generator=g)
    centre = torch.rand((N, Lq, 1, 1, 1, 2), generator=g) * 1.1 - 0.05              # some queries partly outside the maps
    loc = (centre + torch.randn((N, Lq, M, L, P, 2), generator=g) * 0.03).contiguous()
    loc[:, :64] = 0.5 + torch.randn((N, 64, M, L, P, 2), generator=g) * 1e-3          # a heavy bin: 64 queries on one spot
    wgt = torch.softmax(torch.randn((N, Lq, M, L * P), generator=g), -1).view(N, Lq, M, L, P).contiguous()
    go = torch.randn((N, Lq, M * D), generator=g)
    t0 = time.time()
    want = msda_c.msda_backward(value, shapes, starts, loc, wgt, go)
    print("C oracle backward: %.1f s" % (time.time() - t0))
    dv = lambda t: t.to(DEV)
    args = (dv(value), dv(shapes), dv(starts), dv(loc), dv(wgt), dv(go))
    saved = ops.BACKWARD_MODE
    try:
        ops.BACKWARD_MODE = "det"
        a = ops.msda_backward(*args)
        b = ops.msda_backward(*args)
        ops.BACKWARD_MODE = "atomic"
        c = ops.msda_backward(*args)
    finally:
        ops.BACKWARD_MODE = saved
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y), "deterministic backward is not bit-reproducible"
    for name, got, ref in zip(("grad_value", "grad_loc", "grad_attn"), a, want):
        scale = float(ref.abs().max())
        e_det = float((got.cpu().double() - ref).abs().max()) / scale
        print("%s: deterministic kernel vs C oracle (double accumulation) %.2e of max |g| %.3g" % (name, e_det, scale))
        # fp32 coordinates and bilinear weights (as the reference computes them) x exact fixed-point sums: 1e-5 of the
        # largest gradient, 1e-4 for the location gradient (a difference of corner values scaled by W / H)
        assert e_det < (1e-5 if name != "grad_loc" else 1e-4), (name, e_det)
    for name, got, ref in zip(("grad_value", "grad_loc", "grad_attn"), c, want):
        e_at = float((got.cpu().double() - ref).abs().max()) / float(ref.abs().max())
        print("%s: atomic kernel (round 1) vs C oracle %.2e" % (name, e_at))
        assert e_at < (1e-4 if name != "grad_loc" else 1e-3), (name, e_at)
    # speed (HIP events, same inputs)
    def timed(mode, n=5):
        ops.BACKWARD_MODE = mode
        ops.msda_backward(*args)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            ops.msda_backward(*args)
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    try:
        t_det, t_at = timed("det"), timed("atomic")
    finally:
        ops.BACKWARD_MODE = saved
    print("backward of one cfg-2 view-layer: deterministic %.0f us, atomic %.0f us" % (t_det, t_at))


def test_fp32_g_sampling_equals_the_gather_then_linear_form():
    """fp32 path: msda_gfused_f32 (offsets / logits Linear applied to the pyramid once, gathered inside the sampler) against
    the literal gather -> Linear -> fused-sampling decomposition it replaces (projattn.py:148-200): same results to fp32
    rounding -- on a golden case, on level counts 1..4 and on one full-size cfg-2 layer."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    from tests.golden.cases import LAYER_CASES
    cases = []
    for cname in ("mini5_all", "mini5_b2"):
        sp = LAYER_CASES[cname]
        cases.append(build_case(sp["config"], B=sp.get("B", 1), seed=sp["seed"], NQ=sp.get("NQ"), layers=sp["layers"],
                                valid_fraction=sp.get("valid_fraction")))
    cases.append(build_case("cfg2", seed=1, layers=1))
    for case in cases:
        dec = build_decoder_for_case(case, DEV, dtype=torch.float32)
        g = case_to_device(case, DEV)
        outs = []
        for flag in (True, False):
            for layer in dec.layers:
                layer.proj_attn.g_sampling_f32 = flag
            outs.append(_run(dec, g))
        a, b = outs
        e_hs = float((a[0][0] - b[0][0]).abs().max())
        e_px = float((a[2][0] - b[2][0]).abs().max())
        e_mm = float((a[1][0] - b[1][0]).norm(dim=-1).max())
        print("%s layer 0: G-sampling vs gather+Linear (fp32): |hs| %.2e  2D %.2e px  3D %.4f mm" % (case.name, e_hs, e_px, e_mm))
        assert e_hs < 2e-5 and e_px < 5e-3 and e_mm < 0.05
        assert torch.equal(a[1].abs().sum(-1) > 0, b[1].abs().sum(-1) > 0)
