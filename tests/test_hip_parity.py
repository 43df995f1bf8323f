"""GPU parity tests (run on a real MI355X: `pytest -m gpu`).  Every test goes through the
C ABI (libmvgformer_hip.so) and is checked against the oracle and the reference's golden
vectors.  Tolerances are written next to each assertion; fp32 target is the north_star's 1e-4
relative to the tensor's scale, the DLT output is bounded by the reference's own fp32 SVD
conditioning noise (see tests/test_oracle_golden.py)."""
import os

import numpy as np
import pytest
import torch

from mvgformer_amd.synthetic import build_case, to_torch_state
from tests.golden.cases import LAYER_CASES, msda_case

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def _relerr(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def O():
    from oracle import decoder_ref
    return decoder_ref


def _case(cname, **kw):
    spec = LAYER_CASES[cname]
    return build_case(spec["config"], B=spec.get("B", 1), seed=spec["seed"], NQ=spec.get("NQ"),
                      layers=spec.get("layers"), valid_fraction=spec.get("valid_fraction"), **kw)


# ------------------------------------------------------------------ Deformable.deform_forward
@pytest.mark.parametrize("name", ["small_f32", "ragged_f32", "edge_f32"])
def test_deform_forward_f32_vs_reference_golden(name, O):
    from mvgformer_amd import deformable
    g = _load("msda")
    c = {k: v.to(DEV) for k, v in msda_case(name).items()}
    y = deformable.deform_forward(c["value"], c["shapes"], c["starts"], c["loc"], c["weight"], 64)
    assert y.shape == g[name + "/out"].shape
    assert _relerr(y, g[name + "/out_f64"]) < 1e-5          # vs the reference twin in fp64
    assert _relerr(y, g[name + "/out"]) < 1e-5              # vs the reference twin in fp32


def test_deform_forward_known_answers():
    """texel centre -> that texel; border / outside points -> reference zero padding."""
    from mvgformer_amd import deformable
    c = msda_case("edge_f32")
    value, loc = c["value"].clone(), c["loc"].clone()
    w = torch.zeros_like(c["weight"])
    w[..., 0, 0] = 1.0                                       # only level 0 / point 0 counts
    H0, W0 = [int(x) for x in c["shapes"][0]]
    y = deformable.deform_forward(value.to(DEV), c["shapes"].to(DEV), c["starts"].to(DEV), loc.to(DEV), w.to(DEV), 64)
    y = y.cpu().view(1, 9, 4, 16)
    assert torch.allclose(y[0, 0], value[0, 2 * W0 + 3], atol=1e-6)        # exact texel (row 2, col 3)
    # query 1: x = -0.5/W -> w_im = -1 -> skipped entirely (cuh:298 strict >)
    assert float(y[0, 1].abs().max()) == 0.0
    # query 2: y = 1 + 0.5/H -> h_im = H -> skipped (strict <)
    assert float(y[0, 2].abs().max()) < 1e-5


def test_linear_ordered_skips_masked_tiles_and_equals_the_plain_gemm():
    """mvg_linear_ordered (fp32 path, round 3): rows visited in a processing order, tiles without an unmasked row broadcast the
    cached constant row instead of computing -- bit-identical to the plain GEMM + row mask, for sorted and shuffled orders,
    ragged sizes, ReLU, and a non-zero constant row (the pose MLP's hidden layers)."""
    from mvgformer_amd import ops
    rs = np.random.RandomState(11)
    for M, frac, shuffled in ((5000, 0.4, False), (5000, 0.4, True), (129, 1.0, False), (777, 0.0, True), (20000, 0.7, False)):
        a = torch.from_numpy(rs.standard_normal((M, 256)).astype(np.float32)).to(DEV)
        w = torch.from_numpy((rs.standard_normal((256, 256)) / 16).astype(np.float32)).to(DEV)
        b = torch.from_numpy(rs.standard_normal(256).astype(np.float32)).to(DEV)
        perm = torch.from_numpy(rs.permutation(M) if shuffled else np.arange(M)).to(DEV)
        inside = torch.ones(M, dtype=torch.uint8, device=DEV)
        nm = int(M * frac)
        if nm:
            inside[perm[M - nm:]] = 0
        a = a * inside[:, None].float()                                  # masked rows are zero rows (what the sampler writes)
        order = perm.to(torch.int32)
        zero = torch.zeros(256, device=DEV)
        assert torch.equal(ops.linear_ordered(a, w, b, order, inside, zero, rowmask=inside), ops.linear(a, w, b, rowmask=inside))
        const = ops.linear(torch.zeros(1, 256, device=DEV), w, b, relu=True)[0]          # a zero row through the same kernel
        assert torch.equal(ops.linear_ordered(a, w, b, order, inside, const, relu=True), ops.linear(a, w, b, relu=True))


def test_deform_forward_bf16_and_ragged_batch():
    from mvgformer_amd import deformable
    c = {k: v.to(DEV) for k, v in msda_case("small_f32").items()}
    y32 = deformable.deform_forward(c["value"], c["shapes"], c["starts"], c["loc"], c["weight"], 64)
    y16 = deformable.deform_forward(c["value"].bfloat16(), c["shapes"], c["starts"], c["loc"], c["weight"], 64)
    assert y16.dtype == torch.bfloat16
    assert _relerr(y16.float(), y32) < 2e-2                  # bf16 storage of value/out (8-bit mantissa)
    # empty query set
    e = deformable.deform_forward(c["value"], c["shapes"], c["starts"], c["loc"][:, :0].contiguous(),
                                  c["weight"][:, :0].contiguous(), 64)
    assert e.shape == (2, 0, 256)
    with pytest.raises(RuntimeError, match="contiguous"):
        deformable.deform_forward(c["value"], c["shapes"], c["starts"], c["loc"].transpose(1, 2), c["weight"], 64)
    with pytest.raises(RuntimeError, match="im2col_step"):
        deformable.deform_forward(c["value"].repeat(3, 1, 1, 1)[:3], c["shapes"], c["starts"],
                                  c["loc"].repeat(3, 1, 1, 1, 1, 1)[:3].contiguous(),
                                  c["weight"].repeat(3, 1, 1, 1, 1)[:3].contiguous(), 2)


@pytest.mark.parametrize("name", ["small_f32", "ragged_f32", "edge_f32"])
def test_deform_backward_vs_oracle_autograd(name, O):
    """gradients of the HIP backward vs torch autograd through the oracle restatement (fp64)."""
    from mvgformer_amd.functions import DeformFunction
    c = msda_case(name)
    v64 = c["value"].double().requires_grad_(True)
    l64 = c["loc"].double().requires_grad_(True)
    w64 = c["weight"].double().requires_grad_(True)
    y64 = O.msda_forward(v64, c["shapes"], c["starts"], l64, w64)
    go = torch.from_numpy(np.random.RandomState(7).standard_normal(tuple(y64.shape))).double()
    (y64 * go).sum().backward()
    v = c["value"].to(DEV).requires_grad_(True)
    lo = c["loc"].to(DEV).requires_grad_(True)
    w = c["weight"].to(DEV).requires_grad_(True)
    y = DeformFunction.apply(v, c["shapes"].to(DEV), c["starts"].to(DEV), lo, w, 64)
    (y * go.float().to(DEV)).sum().backward()
    assert _relerr(v.grad, v64.grad) < 1e-5
    assert _relerr(w.grad, w64.grad) < 1e-5
    # d/d(loc) is discontinuous at texel borders; the seeded cases stay away from them except the
    # hand-placed border points of edge_f32 (rows 0..6), which are excluded
    sl = slice(7, None) if name == "edge_f32" else slice(None)
    assert _relerr(lo.grad[:, sl], l64.grad[:, sl]) < 1e-4


@pytest.mark.parametrize("name", ["small_f32", "ragged_f32", "edge_f32"])
@pytest.mark.parametrize("mode", ["det", "atomic"])
def test_deform_backward_vs_reference_generated_gradients(name, mode, monkeypatch):
    """Deformable.deform_backward (both kernels: deterministic fixed-point and fp32-atomic) against gradients produced by the
    REFERENCE: autograd of deform_core_pytorch in fp64 (tests/golden/grad.npz, make_golden_grad.py; deform_func.py:48-65)."""
    from mvgformer_amd import deformable
    from tests.golden.cases import msda_grad_output
    from mvgformer_amd import ops
    monkeypatch.setattr(ops, "BACKWARD_MODE", mode)
    g = _load("grad")
    c = {k: v.to(DEV) for k, v in msda_case(name).items()}
    N, Lq = c["loc"].shape[:2]
    go = msda_grad_output(name, (N, Lq, c["value"].shape[2] * c["value"].shape[3])).to(DEV)
    gv, gl, ga = deformable.deform_backward(c["value"], c["shapes"], c["starts"], c["loc"], c["weight"], go, 64)
    pre = "msda/%s/" % name
    assert _relerr(gv, g[pre + "grad_value_f64"]) < 1e-5
    assert _relerr(ga, g[pre + "grad_attn_f64"]) < 1e-5
    sl = slice(7, None) if name == "edge_f32" else slice(None)       # hand-placed texel-border points: one-sided d/d(loc)
    assert _relerr(gl[:, sl], g[pre + "grad_loc_f64"][:, sl]) < 1e-4


def test_deterministic_backward_dynamic_range(O):
    """the fixed-point scale of the deterministic backward is per image and includes max |attn_weight|:
    un-normalised weights (> 1), one image of the batch with 1e6 x larger gradients, and gradients of 1e-36 all give
    the fp64 oracle's gradients to 1e-5 of EACH image's largest value (no int32 saturation, no flushed image, no NaN)."""
    from mvgformer_amd import ops
    assert ops.BACKWARD_MODE == "det"
    c = msda_case("small_f32")
    N = c["value"].shape[0]
    assert N == 2
    rs = np.random.RandomState(5)
    go = torch.from_numpy(rs.standard_normal((N, c["loc"].shape[1], 256)).astype(np.float32))
    for wscale, gscale in ((3.0, (1.0, 1e6)), (1.0, (1e-36, 1e-36)), (40.0, (1e-3, 1.0))):
        w = c["weight"] * wscale
        g = go * torch.tensor(gscale).view(N, 1, 1)
        v64 = c["value"].double().requires_grad_(True)
        l64 = c["loc"].double().requires_grad_(True)
        w64 = w.double().requires_grad_(True)
        (O.msda_forward(v64, c["shapes"], c["starts"], l64, w64) * g.double()).sum().backward()
        gv, gl, ga = ops.msda_backward(c["value"].to(DEV), c["shapes"].to(DEV), c["starts"].to(DEV), c["loc"].to(DEV),
                                       w.to(DEV), g.to(DEV))
        assert torch.isfinite(gv).all() and torch.isfinite(gl).all() and torch.isfinite(ga).all()
        for n in range(N):
            assert _relerr(gv[n], v64.grad[n]) < 1e-5, (wscale, gscale, n)
            assert _relerr(ga[n], w64.grad[n]) < 1e-5, (wscale, gscale, n)
            assert _relerr(gl[n], l64.grad[n]) < 1e-4, (wscale, gscale, n)


# ----------------------------------------------------------------------------- stage kernels
def test_linear_mfma_fp32_and_bf16():
    from mvgformer_amd import ops
    rs = np.random.RandomState(3)
    for (M, N, K) in ((300, 256, 256), (129, 192, 256), (64, 1024, 256), (257, 256, 1024), (5, 3 * 64, 64)):
        a = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(DEV)
        w = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV)
        b = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(DEV)
        mask = torch.from_numpy((rs.rand(M) > 0.3).astype(np.uint8)).to(DEV)
        want = (a.double() @ w.double().t() + b.double())
        got = ops.linear(a, w, b)
        assert _relerr(got, want) < 2e-6, (M, N, K)            # fp32-equivalent products (split or exact form), K <= 1024
        got = ops.linear(a, w, b, relu=True, rowmask=mask)
        assert _relerr(got, torch.relu(want) * mask[:, None].double()) < 2e-6
        # A + A2 formed on load (mvg_linear_sum: the first layer's Linear(tgt + query_pos)): bit-identical to the GEMM of the sum
        a2 = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(DEV)
        assert torch.equal(ops.linear(a, w, b, add=a2), ops.linear(a + a2, w, b))
        assert torch.equal(ops.linear(a, w.bfloat16(), b, out_dtype=torch.float32, add=a2),
                           ops.linear(a + a2, w.bfloat16(), b, out_dtype=torch.float32))
        # bf16 compute: compare against the same product of bf16-rounded operands
        a16, w16 = a.bfloat16(), w.bfloat16()
        want16 = a16.double() @ w16.double().t() + b.double()
        got16 = ops.linear(a16, w16, b, out_dtype=torch.float32)
        assert _relerr(got16, want16) < 1e-5, (M, N, K)       # fp32 accumulation of exact bf16 products
        got16b = ops.linear(a, w16, b, out_dtype=torch.bfloat16)   # fp32 A converted on load
        assert _relerr(got16b.float(), want16) < 1e-2


def _f32_gemm(form):
    """select the fp32 GEMM form of csrc/gemm.hip: "split" (default) or "exact" """
    from mvgformer_amd import _lib
    assert _lib.load().mvg_set_tuning(b"f32_split", 1 if form == "split" else 0) == 0


@pytest.mark.parametrize("M,N,K", [(1000, 200, 256), (128, 256, 32), (4097, 192, 1024), (77, 8, 64), (20000, 256, 256)])
def test_linear_f32_split_form_is_as_close_to_fp64_as_the_exact_form(M, N, K):
    """fp32 GEMMs run on the bf16 matrix pipe by default: every operand value as three bf16 parts (exact: 8 + 8 + 8 significand
    bits), six of the nine partial products, fp32 accumulation (csrc/gemm.hip).  Against the fp64 product of the same fp32
    operands its error must not exceed that of the exact form (v_mfma_f32_32x32x2_f32 = an fmaf chain) -- normal operands,
    all-positive operands (no cancellation) and operands spread over 40 binades; unit = sum_k |a||w| + |b|."""
    from mvgformer_amd import ops
    g = torch.Generator(device=DEV)
    g.manual_seed(M + N + K)
    try:
        for kind in ("normal", "positive", "wide"):
            a = torch.randn(M, K, device=DEV, generator=g)
            w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
            if kind == "wide":
                a = a * torch.exp2(torch.randint(-20, 21, (M, K), device=DEV, generator=g).float())
                w = w * torch.exp2(torch.randint(-20, 21, (N, K), device=DEV, generator=g).float())
            elif kind == "positive":
                a, w = a.abs(), w.abs()
            b = torch.randn(N, device=DEV, generator=g)
            ref = a.double() @ w.double().t() + b.double()
            unit = a.double().abs() @ w.double().abs().t() + b.double().abs()
            err = {}
            for form in ("exact", "split"):
                _f32_gemm(form)
                e = (ops.linear(a, w, b).double() - ref).abs() / unit
                err[form] = (float(e.max()), float(e.mean()))
            assert err["split"][0] <= 1.5 * err["exact"][0] + 2.0 ** -24, (kind, err)
            assert err["split"][1] <= 1.25 * err["exact"][1] + 2.0 ** -28, (kind, err)
            assert err["split"][0] < 2.0 ** -24 * (8 + K ** 0.5), (kind, err)        # a few units of fp32 rounding, growing like a random walk
    finally:
        _f32_gemm("split")


def test_linear_f32_random_shapes_both_forms():
    """40 random GEMM shapes (ragged M, any N % 8 == 0, K % 32 == 0 up to 1024, every tile shape of the launcher) with random
    ReLU / row mask / A + A2: both fp32 forms against the fp64 product."""
    from mvgformer_amd import ops
    rs = np.random.RandomState(2024)
    g = torch.Generator(device=DEV)
    g.manual_seed(7)
    try:
        for it in range(40):
            M = int(rs.choice([1, 7, 63, 64, 65, 127, 129, 500, 1000, 2999, 8191]))
            N = int(rs.choice([8, 24, 64, 128, 136, 192, 256, 320, 384, 576, 1024]))
            K = int(rs.choice([32, 64, 96, 256, 512, 1024]))
            a = torch.randn(M, K, device=DEV, generator=g)
            a2 = torch.randn(M, K, device=DEV, generator=g) if rs.rand() < 0.3 else None
            w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
            b = torch.randn(N, device=DEV, generator=g) if rs.rand() < 0.8 else None
            relu = bool(rs.rand() < 0.5)
            mask = (torch.rand(M, device=DEV, generator=g) > 0.3).to(torch.uint8) if rs.rand() < 0.5 else None
            x = a if a2 is None else a + a2
            ref = x.double() @ w.double().t() + (0 if b is None else b.double())
            unit = x.double().abs() @ w.double().abs().t() + (0 if b is None else b.double().abs()) + 1e-30
            if relu:
                ref = torch.relu(ref)
            if mask is not None:
                ref = ref * mask[:, None].double()
            for form in ("split", "exact"):
                _f32_gemm(form)
                got = ops.linear(a, w, b, relu=relu, rowmask=mask, add=a2)
                err = float(((got.double() - ref).abs() / unit).max())
                assert err < 2.0 ** -24 * (8 + K ** 0.5), (it, M, N, K, form, err)
                if mask is not None:
                    assert float(got[mask == 0].abs().max() if (mask == 0).any() else 0.0) == 0.0
    finally:
        _f32_gemm("split")


def test_linear_f32_split_form_edges():
    """zeros, powers of two, values that need all three parts, tiny and large magnitudes, ReLU / row mask / A + A2 on load, the
    processing-order form; a non-finite input value stays in its own row."""
    from mvgformer_amd import ops
    rs = np.random.RandomState(5)
    M, N, K = 300, 64, 64
    a = rs.standard_normal((M, K)).astype(np.float32)
    a[0] = 0.0
    a[1] = 2.0 ** rs.randint(-30, 30, K)
    a[2] = np.float32(1.0) + np.float32(2.0 ** -23)                   # 1 + ulp: hi = 1, mid = 0, lo = 2^-23
    a[3] = rs.standard_normal(K) * 1e-30
    a[4] = rs.standard_normal(K) * 1e30
    a[5] = np.float32(16777215.0)                                     # 2^24 - 1: all 24 significand bits set
    w = (rs.standard_normal((N, K)) / 8).astype(np.float32)
    w[0] = 0.0
    w[1] = 1.0
    b = rs.standard_normal(N).astype(np.float32)
    A, W, Bv = (torch.from_numpy(x).to(DEV) for x in (a, w, b))
    want = A.double() @ W.double().t() + Bv.double()
    unit = A.double().abs() @ W.double().abs().t() + Bv.double().abs()
    got = ops.linear(A, W, Bv)
    assert float(((got.double() - want).abs() / unit).max()) < 1e-6
    assert torch.equal(got[0], Bv) and torch.equal(got[:, 0], Bv[0].expand(M))             # zero row / zero weight row: the bias, exactly
    mask = torch.from_numpy((rs.rand(M) > 0.5).astype(np.uint8)).to(DEV)
    r = ops.linear(A, W, Bv, relu=True, rowmask=mask)
    assert torch.equal(r, torch.relu(got) * mask[:, None].float())
    A2 = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(DEV)
    assert torch.equal(ops.linear(A, W, Bv, add=A2), ops.linear(A + A2, W, Bv))
    bad = A.clone()
    bad[7, 3] = float("inf")
    bad[9, 0] = float("nan")
    gb = ops.linear(bad, W, Bv)
    rows = torch.ones(M, dtype=torch.bool, device=DEV)
    rows[7] = rows[9] = False
    assert torch.equal(gb[rows], got[rows]) and not torch.isfinite(gb[7]).any() and not torch.isfinite(gb[9, 1:]).any()


@pytest.mark.parametrize("cname", ["mini5_all", "mini5_b2", "mini9"])
def test_fp32_decoder_on_split_gemms_vs_exact_gemms(cname):
    """the whole fp32 decoder on the two GEMM forms: the outputs differ by fp32 rounding (the forms round differently, neither is
    the more accurate one) -- far inside the bars against the reference's own arrays."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = _case(cname)
    dec = build_decoder_for_case(case, DEV)
    gc = case_to_device(case, DEV)
    run = lambda: dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes, gc.level_start_index, None,
                      query_pos=gc.query_pos, threshold=0.1)
    try:
        with torch.no_grad():
            _f32_gemm("exact")
            hs0, refs0, r2d0, p2d0, cls0 = run()
            _f32_gemm("split")
            hs1, refs1, r2d1, p2d1, cls1 = run()
    finally:
        _f32_gemm("split")
    assert _relerr(hs1, hs0) < 2e-5
    assert float((torch.stack(cls1) - torch.stack(cls0)).abs().max()) < 2e-6
    valid = (refs0[1:].abs().sum(-1) > 0) & (refs1[1:].abs().sum(-1) > 0)
    assert torch.equal(refs0[1:].abs().sum(-1) > 0, refs1[1:].abs().sum(-1) > 0)           # same valid pattern
    d = (refs1[1:] - refs0[1:]).norm(dim=-1)[valid]
    assert d.numel() > 0 and float(d.median()) < 0.01 and float(d.max()) < 2.0             # mm; the DLT amplifies rounding


@pytest.mark.parametrize("cname", ["mini5_all", "mini5_b2", "cfg1"])
def test_project_matches_reference_golden(cname):
    from mvgformer_amd import ops
    g = _load(cname)
    case = _case(cname, with_features=False)
    levels = ops.Levels(case.spatial_shapes, case.level_start_index)
    cams = ops.pack_cameras(case.meta, case.img_size, DEV)
    r, ref_lvl, inside = ops.project(case.reference_points.to(DEV), cams, levels, case.V, case.B)
    r = r.view(case.V, case.B, -1, 2).cpu()
    assert np.array_equal(inside.view(case.V, case.B, -1).cpu().numpy().astype(bool), g["proj_inside"])
    assert float((r - torch.from_numpy(g["proj_r"])).abs().max()) < 2e-5        # normalised coords O(1)
    WH = case.spatial_shapes.flip(-1).float()
    want_lvl = r.unsqueeze(3) * WH / (WH - 1)
    assert float((ref_lvl.view(case.V, case.B, -1, 3, 2).cpu() - want_lvl).abs().max()) < 1e-6


def test_projattn_stages_match_reference_golden(O):
    """gather -> value/offset/logit linears -> fused sampling -> output projection, layer 0 / view 0."""
    from mvgformer_amd import ops
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    g = _load("mini5_all")
    case = _case("mini5_all")
    dec = build_decoder_for_case(case, DEV)
    pa = dec.layers[0].proj_attn
    gc = case_to_device(case, DEV)
    levels = ops.Levels(gc.spatial_shapes, gc.level_start_index)
    src0 = [s[0:1] for s in gc.src_views]
    feat = ops.pack_pyramid(src0, levels, torch.float32)
    # pyramid packing = cat + permute (projattn.py:160)
    want_flat = torch.cat([s.flatten(2) for s in src0], -1).transpose(1, 2)
    assert torch.equal(feat, want_flat.contiguous())
    ref_lvl = torch.from_numpy(g["pa_ref"]).to(DEV)                          # (1,Lq,3,2)
    x = (gc.tgt + gc.query_pos).contiguous()
    ain = ops.gather_ref(feat, ref_lvl, x, levels, 1, 1)
    assert _relerr(ain.view(1, -1, 3, 256)[:, :20], g["pa_x_q20"]) < 1e-5
    Wv, bv, Woa, boa, Wp, bp = pa.weights(torch.float32)
    value = ops.linear(feat.view(-1, 256), Wv, bv)
    rows = g["pa_value_rows"]
    assert _relerr(value.view(1, -1, 8, 32)[:, rows], g["pa_value_sub"]) < 1e-5
    oa = ops.linear(ain, Woa, boa, out_dtype=torch.float32)
    assert _relerr(oa[:, :128].reshape(g["pa_off"].shape), g["pa_off"]) < 1e-5
    samp = ops.msda_fused(value.view(1, -1, 256), oa, ref_lvl, levels)
    assert _relerr(samp.view(g["pa_samp"].shape), g["pa_samp"]) < 2e-5
    out = pa(x, ref_lvl, src0, None, gc.spatial_shapes, gc.level_start_index)   # reference signature, no grad
    assert _relerr(out, g["pa_out"]) < 1e-4  # north_star: 1e-4 fp32


def test_projattn_autograd_path_matches_native(O):
    """training path (torch autograd + DeformFunction) == inference path; gradients flow."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    g = _load("mini5_all")
    case = _case("mini5_all")
    dec = build_decoder_for_case(case, DEV)
    pa = dec.layers[0].proj_attn
    gc = case_to_device(case, DEV)
    src0 = [s[0:1] for s in gc.src_views]
    ref_lvl = torch.from_numpy(g["pa_ref"]).to(DEV)
    x = (gc.tgt + gc.query_pos).contiguous()
    with torch.no_grad():
        y_native = pa(x, ref_lvl, src0, None, gc.spatial_shapes, gc.level_start_index)
    xg = x.clone().requires_grad_(True)
    y_train = pa(xg, ref_lvl, src0, None, gc.spatial_shapes, gc.level_start_index)
    assert _relerr(y_train, y_native) < 1e-4
    y_train.square().sum().backward()
    assert xg.grad is not None and float(xg.grad.abs().max()) > 0
    assert pa.sampling_offsets.weight.grad is not None and pa.rayconv.weight.grad is not None


def test_triangulation_kernel_vs_reference_golden(O):
    """un-crop / undistort / DLT: HIP (fp64 normal equations) vs the reference's fp32 SVD and the
    fp64 SVD of the same rows."""
    from mvgformer_amd import ops
    g = _load("mini5_all")
    case = _case("mini5_all", with_features=False)
    cams = ops.pack_cameras(case.meta, case.img_size, DEV)
    n, V, J = g["tri_kp"].shape[:3]
    # feed the kernel original-image points through r = crop(kp)/img and zero offsets
    kp = torch.from_numpy(g["tri_kp"]).double()                              # (n,V,J,2) orig px
    A = O.crop_affine_matrix(case.meta[0]["center"], case.meta[0]["scale"], case.img_size, torch.float64)[0]
    net = kp @ A[:, :2].t() + A[:, 2]
    img = torch.tensor(case.img_size, dtype=torch.float64)
    r = (net / img).permute(1, 0, 2, 3).reshape(V, 1, n * J, 2).float().contiguous().to(DEV)   # (V*B, Lq, 2)
    conf = torch.from_numpy(g["tri_conf"])                                   # softmax over views already
    o = torch.zeros((V, 1, n * J, 3))
    o[..., 2] = torch.log(conf).permute(1, 0, 2).reshape(V, 1, n * J)
    valid = torch.ones((1, n), dtype=torch.uint8, device=DEV)
    anyv = torch.ones((1,), dtype=torch.int32, device=DEV)
    X, ref2d, proj2d = ops.triangulate(r.view(V, n * J, 2), o.to(DEV).view(V, n * J, 3).contiguous(), cams, valid, anyv,
                                       V, 1, n, J)
    X = X.cpu().view(n, J, 3)
    want32 = torch.from_numpy(g["tri_points3d"])
    # fp64 truth from the oracle on the reference's undistorted points
    cam = {k: v[:1].expand(n, *v.shape[1:]) for k, v in O._stack_cam(case.meta, torch.float64).items()}
    Pm = O.projection_matrices(cam, torch.float64)
    X64, _ = O.dlt_triangulate(Pm, torch.from_numpy(g["tri_undist"]).double(), conf.double())
    rel_ref = ((want32.double() - X64).norm(dim=-1) / X64.norm(dim=-1).clamp_min(1.0)).max()
    rel_hip = ((X.double() - X64).norm(dim=-1) / X64.norm(dim=-1).clamp_min(1.0)).max()
    rel_vs_ref = ((X - want32).norm(dim=-1) / want32.norm(dim=-1).clamp_min(1.0)).max()
    # random (non-corresponding) 2D points: ill-posed -> relative bounds
    print("DLT rel err: reference fp32 vs fp64 %.3e | HIP vs fp64 %.3e | HIP vs reference %.3e"
          % (float(rel_ref), float(rel_hip), float(rel_vs_ref)))
    assert float(rel_vs_ref) < 5e-3 + 2.0 * float(rel_ref), float(rel_vs_ref)
    assert float(rel_hip) <= max(2.0 * float(rel_ref), 1e-4), (float(rel_hip), float(rel_ref))
    assert float((proj2d.cpu().view(V, n, J, 2).permute(1, 0, 2, 3) - net.float()).abs().max()) < 1e-2


def test_triangulation_known_answer_noise_free():
    """two+ views, exact projections of known 3D points, k=p=0 -> recovered to 1e-2 mm."""
    from mvgformer_amd import ops
    from mvgformer_amd.synthetic import CONFIGS, make_meta, ring_cameras
    c = dict(CONFIGS["cfg4"])                                                # Shelf-like: k = p = 0
    for V in (2, 5):
        cams_np = ring_cameras(V, c["orig_wh"], c["focal"], c["radius"], c["space_center"], c["k"], c["p"], seed=1)
        meta = make_meta(cams_np, 1, c["orig_wh"], c["img_wh"])
        cams = ops.pack_cameras(meta, c["img_wh"], DEV)
        rs = np.random.RandomState(5)
        NQ, J = 4, 15
        X = torch.from_numpy((np.asarray(c["space_center"]) + rs.uniform(-800, 800, (1, NQ * J, 3))).astype(np.float32))
        levels = ops.Levels([[8, 8]], [0])
        r, _, inside = ops.project(X.to(DEV), cams, levels, V, 1)
        o = torch.zeros((V, NQ * J, 3), device=DEV)
        valid = torch.ones((1, NQ), dtype=torch.uint8, device=DEV)
        anyv = torch.ones((1,), dtype=torch.int32, device=DEV)
        Xr, ref2d, proj2d = ops.triangulate(r, o, cams, valid, anyv, V, 1, NQ, J)
        err = (Xr.cpu() - X).norm(dim=-1).max()
        assert float(err) < 5e-2, float(err)                                  # mm (fp32 projection round-off)
        assert torch.equal(ref2d, proj2d)                                     # zero offsets


def test_triangulation_result_does_not_depend_on_the_wavefront_neighbours():
    """The DLT's null vector is taken by inverse iteration (rounds of 4 solves, a lane stops when ITS iterates are parallel) with the
    fp64 Jacobi as the fallback for rays that do not meet: a problem's bits must not depend on which problems share its wavefront
    (query-sharded runs are compared bit for bit with single-rank runs).  Persons with exact projections, persons with 2D points
    thrown off by up to 300 px (slow convergence / fallback) and persons seen by one useful view (degenerate pivots), in two
    different orders; and every finite result is the null direction of its own fp64 row matrix."""
    from mvgformer_amd import ops
    from mvgformer_amd.synthetic import CONFIGS, make_meta, ring_cameras
    c = dict(CONFIGS["cfg4"])                                                # k = p = 0: the undistortion is exact
    V, NQ, J = 5, 64, 15
    cams_np = ring_cameras(V, c["orig_wh"], c["focal"], c["radius"], c["space_center"], c["k"], c["p"], seed=3)
    meta = make_meta(cams_np, 1, c["orig_wh"], c["img_wh"])
    cams = ops.pack_cameras(meta, c["img_wh"], DEV)
    rs = np.random.RandomState(9)
    X = torch.from_numpy((np.asarray(c["space_center"]) + rs.uniform(-800, 800, (1, NQ * J, 3))).astype(np.float32))
    r, _, _ = ops.project(X.to(DEV), cams, ops.Levels([[8, 8]], [0]), V, 1)
    o = torch.zeros((V, NQ, J, 3))
    o[:, 16:40, :, :2] = torch.from_numpy(rs.uniform(-300, 300, (V, 24, J, 2)).astype(np.float32))      # rays that do not meet
    o[1:, 40:48, :, 2] = -60.0                                                                           # one view carries all the weight
    o = o.view(V, NQ * J, 3).to(DEV)
    valid = torch.ones((1, NQ), dtype=torch.uint8, device=DEV)
    anyv = torch.ones((1,), dtype=torch.int32, device=DEV)
    Xa = ops.triangulate(r, o, cams, valid, anyv, V, 1, NQ, J)[0].view(NQ, J, 3)
    perm = torch.from_numpy(rs.permutation(NQ)).to(DEV)
    rp = r.view(V, NQ, J, 2)[:, perm].reshape(V, NQ * J, 2).contiguous()
    op = o.view(V, NQ, J, 3)[:, perm].reshape(V, NQ * J, 3).contiguous()
    Xb = ops.triangulate(rp, op, cams, valid, anyv, V, 1, NQ, J)[0].view(NQ, J, 3)
    assert torch.equal(Xa[perm].view(torch.int32), Xb.view(torch.int32))
    err = (Xa[:16].cpu().view(-1, 3) - X[0, :16 * J]).norm(dim=-1).max()
    assert float(err) < 0.2, float(err)                                       # mm: the persons with exact projections
    assert torch.isfinite(Xa[:40]).all()


# ------------------------------------------------------------------------- the decoder layer(s)
TOL = dict(hs=1e-4, px=2e-3, mm=1.5, cls=5e-6)


@pytest.mark.parametrize("cname", list(LAYER_CASES))
def test_decoder_layers_teacher_forced_vs_reference_golden(cname, O):
    """each layer gets the REFERENCE's previous-layer outputs; fp32 compute."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    g = _load(cname)
    case = _case(cname)
    dec = build_decoder_for_case(case, DEV)
    gc = case_to_device(case, DEV)
    thr = float(g["threshold"])
    tgt, ref = gc.tgt, gc.reference_points
    for l in range(case.layers):
        with torch.no_grad():
            hs, new_ref, r2d, p2d, cls = dec.layers[l](tgt, gc.query_pos, ref[:, :, None], gc.src_views,
                                                       gc.spatial_shapes, gc.level_start_index, gc.meta, threshold=thr)
        hs, new_ref, r2d, p2d, cls = (t.cpu() for t in (hs, new_ref, r2d, p2d, cls))
        assert np.array_equal((cls[..., 1] > thr).numpy(), g["cls"][l][..., 1] > thr)
        assert float((cls - torch.from_numpy(g["cls"][l])).abs().max()) < TOL["cls"]
        assert float((hs - torch.from_numpy(g["hs"][l])).abs().max()) < TOL["hs"]       # LayerNorm'd, O(1..4)
        assert float((p2d - torch.from_numpy(g["projs2d"][l])).abs().max()) < TOL["px"]
        assert float((r2d - torch.from_numpy(g["refs2d"][l])).abs().max()) < TOL["px"]
        assert float((new_ref - torch.from_numpy(g["refs"][l])).norm(dim=-1).max()) < TOL["mm"]
        # zeros exactly where the reference has zeros
        assert torch.equal(new_ref == 0, torch.from_numpy(g["refs"][l]) == 0)
        tgt = torch.from_numpy(g["hs"][l]).to(DEV)
        ref = torch.from_numpy(g["refs"][l]).to(DEV)


@pytest.mark.parametrize("cname", list(LAYER_CASES))
def test_decoder_free_running_vs_reference_golden(cname):
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    g = _load(cname)
    case = _case(cname)
    dec = build_decoder_for_case(case, DEV)
    gc = case_to_device(case, DEV)
    thr = float(g["threshold"])
    with torch.no_grad():
        hs, refs, r2d, p2d, cls = dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes,
                                      gc.level_start_index, None, query_pos=gc.query_pos, threshold=thr)
    cls = torch.stack(cls).cpu()
    assert hs.shape == g["hs"].shape and refs.shape == g["refs"].shape and r2d.shape == g["refs2d"].shape
    assert np.array_equal((cls[..., 1] > thr).numpy(), g["cls"][..., 1] > thr)
    assert float((hs.cpu() - torch.from_numpy(g["hs"])).abs().max()) < 2e-2
    assert float((r2d.cpu() - torch.from_numpy(g["refs2d"])).abs().max()) < 0.5
    assert float((refs.cpu() - torch.from_numpy(g["refs"])).norm(dim=-1).max()) < 3.0


def test_decoder_bf16_vs_fp32_oracle(O):
    """bf16 storage / bf16 MFMA with fp32 accumulation vs the fp32 oracle (all queries valid so
    no threshold flips): features 3e-2 abs (bf16 has 8 mantissa bits, values O(1..4)), 2D 0.5 px,
    3D 6 mm on a 4 m scene."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    g = _load("mini5_all")
    case = _case("mini5_all")
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    with torch.no_grad():
        hs, new_ref, r2d, p2d, cls = dec.layers[0](gc.tgt, gc.query_pos, gc.reference_points[:, :, None], gc.src_views,
                                                   gc.spatial_shapes, gc.level_start_index, gc.meta, threshold=0.1)
    assert float((hs.cpu() - torch.from_numpy(g["hs"][0])).abs().max()) < 6e-2
    assert float((p2d.cpu() - torch.from_numpy(g["projs2d"][0])).abs().max()) < 2e-3   # geometry stays fp32
    assert float((r2d.cpu() - torch.from_numpy(g["refs2d"][0])).abs().max()) < 0.5
    assert float((new_ref.cpu() - torch.from_numpy(g["refs"][0])).norm(dim=-1).max()) < 6.0


def test_decoder_full_size_properties():
    """BASELINE configs[1] geometry (5 views, 1024 queries, 960x512) at full size, 1 layer:
    size-independent properties -- query-permutation equivariance (queries are independent units,
    SURVEY.md section 8e) and shard-concatenation == full run."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg2", seed=1, layers=1)
    dec = build_decoder_for_case(case, DEV)
    # runs with different query counts are compared bit for bit: pin the sampling form ("auto" switches from G-sampling to the
    # gather form below Lq * L < S, and the two agree to fp32 rounding only)
    dec.layers[0].proj_attn.g_sampling_f32 = True
    gc = case_to_device(case, DEV)
    run = lambda t, p, r: dec.layers[0](t, p, r[:, :, None], gc.src_views, gc.spatial_shapes, gc.level_start_index,
                                        gc.meta, threshold=0.1)
    with torch.no_grad():
        full = run(gc.tgt, gc.query_pos, gc.reference_points)
        NQ, J = case.NQ, 15
        perm = torch.randperm(NQ, generator=torch.Generator().manual_seed(0)).to(DEV)
        tok = (perm[:, None] * J + torch.arange(J, device=DEV)[None]).reshape(-1)
        pm = run(gc.tgt[:, tok].contiguous(), gc.query_pos[:, tok].contiguous(), gc.reference_points[:, tok].contiguous())
        assert torch.equal(pm[0], full[0][:, tok])            # bit-identical: no cross-query coupling
        assert torch.equal(pm[1], full[1][:, tok])
        assert torch.equal(pm[4], full[4][:, perm])
        half = NQ // 2 * J
        a = run(gc.tgt[:, :half].contiguous(), gc.query_pos[:, :half].contiguous(), gc.reference_points[:, :half].contiguous())
        b = run(gc.tgt[:, half:].contiguous(), gc.query_pos[:, half:].contiguous(), gc.reference_points[:, half:].contiguous())
        assert torch.equal(torch.cat([a[0], b[0]], 1), full[0])
        assert torch.equal(torch.cat([a[1], b[1]], 1), full[1])
        assert torch.equal(torch.cat([a[2], b[2]], 2), full[2])
    assert torch.isfinite(full[0]).all() and torch.isfinite(full[1]).all()



def test_fused_chains_match_unfused_path():
    """LDS-resident chains (output_proj * mask -> pose MLP; view mean -> update -> LN -> FFN -> LN -> class
    head) vs the separate MFMA linears / LayerNorm / class-head kernels."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = _case("mini5_half")
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    layer = dec.layers[0]
    outs = []
    for fused in (False, True):
        layer.use_fused_chains = fused
        with torch.no_grad():
            outs.append(layer(gc.tgt, gc.query_pos, gc.reference_points[:, :, None], gc.src_views, gc.spatial_shapes,
                              gc.level_start_index, gc.meta, threshold=0.1))
    a, b = outs
    # same bf16 operands, fp32 accumulation in another k order -> agreement to fp32/bf16 rounding
    assert float((a[4] - b[4]).abs().max()) < 1e-3 and float((a[0] - b[0]).abs().max()) < 3e-2
    assert torch.equal(a[3], b[3])                                           # projections untouched
    assert float((a[2] - b[2]).abs().max()) < 5e-2                           # px
    assert float((a[1] - b[1]).norm(dim=-1).max()) < 1.0                     # mm



def test_decoder_head_end_to_end_vs_oracle(O):
    """caller glue (embeddings -> queries, sample_space init, decoder, output dict, pred packing) on the GPU
    vs the oracle fed with the same queries / reference points."""
    from mvgformer_amd import caller
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = _case("mini5_all")
    dec = build_decoder_for_case(case, DEV)
    head = caller.DecoderHead(dec, case.NQ, 15, 256, case.space_size, case.space_center).to(DEV)
    torch.manual_seed(0)
    with torch.no_grad():
        head.joint_embedding.weight.normal_()
        head.instance_embedding.weight.normal_()
    gc = case_to_device(_case("mini5_all"), DEV)
    out, pred = head(gc.src_views, gc.meta, threshold=0.1)
    qpos, tgt = caller.person_joint_queries(head.joint_embedding.weight.detach().cpu(),
                                            head.instance_embedding.weight.detach().cpu(), 1)
    ref = caller.sample_space_reference_points(case.NQ, case.space_size, case.space_center, 1, "cpu")
    prm = to_torch_state(case.weights)
    hs, refs, r2d, p2d, cls = O.decoder_forward(prm, case.layers, tgt.contiguous(), ref, case.src_views, case.meta,
                                                case.spatial_shapes, case.level_start_index, qpos.contiguous(),
                                                case.img_size, threshold=0.1)
    assert pred.shape == (1, case.NQ, 15, 5)
    want_valid = cls[-1][..., 1] > 0.1
    assert torch.equal((pred[..., 0, 3] == 0).cpu(), want_valid)                    # (score > thr) - 1 == 0 <=> valid
    assert float((pred[..., 4].cpu()[:, :, 0] - cls[-1][..., 1]).abs().max()) < 2e-3
    got = out["pred_poses"]["outputs_coord"].cpu()
    assert float((got - refs[-1]).norm(dim=-1).max()) < 3.0                         # mm, free-running 2 layers


@pytest.mark.parametrize("fmt", ["panoptic", "shelf"])
def test_decoder_head_vs_reference_model_forward_and_validate_3d(fmt):
    """DecoderHead on the HIP path against what the REFERENCE's DyanmicQueryTransformer.forward + validate_3d produced
    (tests/golden/caller.npz, make_golden_caller.py): two loader batches of B = 2 through one model, about a third of the
    queries valid, Panoptic joint format and the Shelf/Campus permutation.  Glue is exact; the numbers carry the decoder's
    free-running two-layer agreement with the reference (fp32 SVD noise of the reference included)."""
    from mvgformer_amd import caller
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    from tests.golden.cases import CALLER_CASE, caller_embeddings
    g = _load("caller")
    spec = CALLER_CASE
    cases = [build_case(spec["config"], B=spec["B"], seed=spec["seed"] + i, layers=spec["layers"],
                        valid_fraction=spec["valid_fraction"]) for i in range(spec["batches"])]
    conv = None if fmt == "panoptic" else [int(i) for i in g["convert_joint_format_indices"]]
    dec = build_decoder_for_case(cases[0], DEV)                       # ONE model for every batch, like the reference
    head = caller.DecoderHead(dec, cases[0].NQ, 15, 256, cases[0].space_size, cases[0].space_center, conv,
                              t_pose=torch.from_numpy(g["tpose"])).to(DEV)
    je, ie = caller_embeddings(cases[0].NQ)
    with torch.no_grad():
        head.joint_embedding.weight.copy_(je)
        head.instance_embedding.weight.copy_(ie)
    thr = float(g["threshold"])
    for b, case in enumerate(cases):
        gc = case_to_device(case, DEV)
        out, pred = head(gc.src_views, gc.meta, threshold=thr)
        want = torch.from_numpy(g["%s/b%d/pred" % (fmt, b)])
        pred = pred.cpu()
        assert pred.shape == want.shape
        assert torch.equal(pred[..., 3], want[..., 3])                                   # (score > thr) - 1: identical
        assert float((pred[..., 4] - want[..., 4]).abs().max()) < 1e-4                   # score (free-running, layer 2)
        valid = want[..., 0, 3] == 0
        assert 0 < int(valid.sum()) < valid.numel()
        assert torch.equal(pred[..., :3][~valid], want[..., :3][~valid])                 # zeros for the others
        # poses of the valid queries after two free-running layers against the reference's fp32 arithmetic: a query that failed
        # the filter in layer 1 restarts from the world origin (zeros, dq_decoder.py:1013-1029) with five inconsistent 2D
        # observations, and the reference's fp32 SVD of such DLT rows is off by centimetres from the fp64 solution of the
        # same rows (tests/test_oracle_golden.py::test_reference_fp32_dlt_noise; 589-mm outliers of the fp32 oracle in
        # DESIGN 5.1) -- so: median and 90th percentile, not the maximum
        err = (pred[..., :3] - want[..., :3]).norm(dim=-1)[valid]
        assert float(err.median()) < 0.1 and float(torch.quantile(err.flatten(), 0.9)) < 3.0, (float(err.median()), float(err.max()))
        if b == spec["batches"] - 1:
            assert float((out["pred_logits"].cpu() - torch.from_numpy(g[fmt + "/out/pred_logits"])).abs().max()) < 2e-3
            for k, kk in (("pred_poses_2d", "outputs_coord_2d"), ("pred_poses_2d_proj", "outputs_coord_2d_proj")):
                w2 = torch.from_numpy(g["%s/out/%s" % (fmt, k)])
                got = out[k][kk].cpu()
                assert got.shape == w2.shape and torch.equal(got == 0, w2 == 0)
                e2 = (got - w2).abs().amax(-1)
                assert float(e2.median()) < 0.05 and float(torch.quantile(e2.flatten(), 0.9)) < 0.5       # px


def test_decoder_layer_training_path_matches_inference_and_backprops():
    """autograd path (torch ops + HIP sampling op fwd/bwd) == native inference path (fp32); gradients reach
    every parameter group the reference trains."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = _case("mini5_half")
    dec = build_decoder_for_case(case, DEV)
    gc = case_to_device(case, DEV)
    layer = dec.layers[0]
    with torch.no_grad():
        inf = layer(gc.tgt, gc.query_pos, gc.reference_points[:, :, None], gc.src_views, gc.spatial_shapes,
                    gc.level_start_index, gc.meta, threshold=0.1)
    layer.train()
    for m in layer.modules():                       # dropout would randomise the comparison
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    tgt = gc.tgt.clone().requires_grad_(True)
    out = layer(tgt, gc.query_pos, gc.reference_points[:, :, None], gc.src_views, gc.spatial_shapes,
                gc.level_start_index, gc.meta, threshold=0.1)
    assert float((out[0] - inf[0]).abs().max()) < 1e-4
    assert float((out[4] - inf[4]).abs().max()) < 5e-6
    assert float((out[3] - inf[3]).abs().max()) < 2e-3 and float((out[2] - inf[2]).abs().max()) < 2e-3
    err3d = float((out[1] - inf[1]).norm(dim=-1).max().detach())
    print("training-path 3D vs inference-path: %.3f mm" % err3d)
    assert err3d < 0.05         # both paths: smallest eigenvector of the fp64 Gram matrix of their fp32 rows (mm, 4 m scene;
                                # 0.001 measured -- it was 1-3 mm while the training path went through rocSOLVER's fp32 SVD)
    loss = out[0].square().mean() + 1e-6 * out[1].square().mean() + out[4].sum() + 1e-4 * out[2].square().mean()
    loss.backward()
    assert tgt.grad is not None and torch.isfinite(tgt.grad).all() and float(tgt.grad.abs().max()) > 0
    for name in ("proj_attn.sampling_offsets.weight", "proj_attn.attention_weights.weight", "proj_attn.rayconv.weight",
                 "proj_attn.output_proj.weight", "feature_update_mlp.weight", "linear1.weight", "linear2.weight",
                 "pose_embed.MLP.layers.0.weight", "pose_embed.MLP.layers.2.weight", "class_embed.weight", "norm2.weight"):
        g = dict(layer.named_parameters())[name].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0, name


@pytest.mark.parametrize("cname", ["mini5_all", "mini5_half", "mini5_b2"])
def test_layer_gradients_vs_reference_autograd(cname, O):
    """f2: DQDecoderLayer.forward under autograd on the GPU (forward_autograd: HIP sampling op forward + backward, fp64
    Gram-eigenvector DLT with analytic backward) -- d(loss)/d(tgt) and d(loss)/d(every trained parameter) against
    (1) the REFERENCE layer under torch autograd in its fp32 (tests/golden/grad.npz; matched indices / class-head filter /
        batch of 2), bars as in tests/test_oracle_golden.py: 2e-4 of the tensor's largest gradient, 5e-3 for the pose head
        whose reference gradient carries the noise of the fp32 SVD backward;
    (2) the oracle under autograd in fp64, evaluated here: 1e-4 for every tensor, pose head included (measured 7e-6 on
        mini5_half / mini5_b2; the pose head of mini5_all is the documented exception below)."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    from tests.golden.cases import GRAD_CASES, layer_loss, subsample_grad
    g = _load("grad")
    pre = "layer/%s/" % cname
    case = _case(cname)
    idx = GRAD_CASES[cname]["indices"]
    thr = LAYER_CASES[cname].get("threshold", 0.1)
    # (2) fp64 oracle
    prm = {k: v.double().requires_grad_(k.startswith("layers.0.")) for k, v in to_torch_state(case.weights).items()}
    t64 = case.tgt.double().clone().requires_grad_(True)
    o64 = O.decoder_layer_forward(prm, "layers.0.", t64, case.query_pos, case.reference_points, case.src_views,
                                  case.spatial_shapes, case.level_start_index, case.meta, case.img_size, threshold=thr,
                                  dtype=torch.float64, indices=idx)
    layer_loss(o64).backward()
    want64 = {"tgt": t64.grad}
    want64.update({k[len("layers.0."):]: v.grad for k, v in prm.items() if v.grad is not None})
    # the HIP training path
    dec = build_decoder_for_case(case, DEV)
    gc = case_to_device(case, DEV)
    layer = dec.layers[0]
    layer.eval()
    tgt = gc.tgt.clone().requires_grad_(True)
    out = layer(tgt, gc.query_pos, gc.reference_points[:, :, None], gc.src_views, gc.spatial_shapes, gc.level_start_index,
                gc.meta, indices=None if idx is None else [torch.tensor(q, device=DEV) for q in idx], threshold=thr)
    loss = layer_loss(out)
    loss.backward()
    got = {"tgt": tgt.grad}
    got.update({n: p.grad for n, p in layer.named_parameters() if p.grad is not None})
    names = [str(n) for n in g[pre + "names"]]
    assert set(names) == set(got), set(names) ^ set(got)              # the same tensors receive a gradient
    assert abs(float(loss) - float(g[pre + "loss"])) < 1e-4 * abs(float(loss))
    assert np.array_equal((out[1].detach().abs().sum(-1) > 0).cpu().numpy(), g[pre + "valid"])
    for k, t in zip(("hs", "ref3d", "ref2d", "proj2d", "prob"), out):
        bar = {"hs": 1e-4, "ref3d": 1.5, "ref2d": 2e-3, "proj2d": 2e-3, "prob": 5e-6}[k]      # teacher-forced bars (a2)
        assert float((t.detach().cpu() - torch.from_numpy(g[pre + "out/" + k])).abs().max()) < bar, k
    w_ref, w_ref_dlt, w64 = 0.0, 0.0, 0.0
    for n in names:
        scale = float(g[pre + "absmax/" + n])
        a = got[n].detach().double().cpu()
        r_ref = float((subsample_grad(n, a) - torch.from_numpy(g[pre + "grad/" + n]).double()).abs().max()) / scale
        r_64 = float((a - want64[n]).abs().max()) / scale
        dlt = n.startswith("pose_embed.")
        assert r_ref < (5e-3 if dlt else 2e-4), (n, r_ref)
        # fp64 oracle: 1e-4 for every tensor.  One exception: the pose head on mini5_all, whose matched queries include
        # poorly conditioned triangulations -- fp32 and fp64 evaluations of the SAME algorithm differ there by 1.6e-3 (fp32
        # oracle vs fp64 oracle, tests/test_oracle_golden.py; no ReLU of the pose MLP changes sign, the difference is the
        # fp32 rounding of the DLT rows amplified by the triangulation's backward); this path builds its DLT rows in fp32
        # like the reference and sits with the fp32 evaluations (2.9e-3 from the fp64 oracle, inside 5e-3 of the reference)
        assert r_64 < (5e-3 if (dlt and cname == "mini5_all") else 1e-4), (n, r_64)
        w64 = max(w64, r_64)
        if dlt:
            w_ref_dlt = max(w_ref_dlt, r_ref)
        else:
            w_ref = max(w_ref, r_ref)
    print("%s gradients: vs reference fp32 %.2e (pose head %.2e), vs fp64 oracle %.2e" % (cname, w_ref, w_ref_dlt, w64))


def test_projattn_training_path_with_more_than_im2col_step_images():
    """all V views run as one batch of V * B images; 5 views x B = 16 -> 80 images > im2col_step = 64 with
    80 % 64 != 0 must work like the reference's per-view calls (batch 16 each) do -- and equal them."""
    from mvgformer_amd.projattn import ProjAttn
    torch.manual_seed(3)
    pa = ProjAttn(256, 1, 8, 8, "ablation_not_use_rayconv")
    pa._reset_parameters()
    pa = pa.to(DEV)
    V, B, Lq = 5, 16, 7
    shapes = torch.tensor([[6, 10], [3, 5]], dtype=torch.long, device=DEV)
    starts = torch.tensor([0, 60], dtype=torch.long, device=DEV)
    src = [torch.randn(V * B, 256, 6, 10, device=DEV), torch.randn(V * B, 256, 3, 5, device=DEV)]
    q = torch.randn(V * B, Lq, 256, device=DEV, requires_grad=True)
    ref = torch.rand(V * B, Lq, 2, 2, device=DEV)
    out = pa(q, ref, src, None, shapes, starts)                      # one batch of 80 images
    out.square().sum().backward()
    assert out.shape == (V * B, Lq, 256) and torch.isfinite(q.grad).all()
    q2 = q.detach().clone().requires_grad_(True)                     # the reference's call pattern: per view, batch B
    per_view = torch.cat([pa(q2[v * B:(v + 1) * B], ref[v * B:(v + 1) * B], [s[v * B:(v + 1) * B] for s in src], None,
                             shapes, starts) for v in range(V)], 0)
    assert float((out - per_view).abs().max()) < 1e-5


def test_training_path_reference_point_gather_through_the_sampling_op():
    """ProjAttn under autograd: the reference-point features come from the HIP sampling op (one point per level, forward and
    deterministic backward kernels) -- same output and same gradients (query, reference points incl. points at and beyond the
    map border, feature maps, every weight) as L torch grid_sample calls on the NCHW maps (projattn.py:134-141)."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = _case("mini5_b2")
    dec = build_decoder_for_case(case, DEV)
    gc = case_to_device(case, DEV)
    pa = dec.layers[0].proj_attn
    B = case.B
    nl = len(gc.src_views)
    gen = torch.Generator().manual_seed(3)
    Lq = 240
    ref = (torch.rand((B, Lq, nl, 2), generator=gen) * 1.3 - 0.15).to(DEV)          # some outside, some past the clamp
    ref[:, :4] = torch.tensor([0.0, 1.0], device=DEV)                                # corners
    q0 = torch.randn((B, Lq, 256), generator=gen).to(DEV)
    src0 = [s_[:B].clone() for s_ in gc.src_views]
    res = {}
    for native in (False, True):
        pa.ref_gather_native = native
        q = q0.clone().requires_grad_(True)
        r = ref.clone().requires_grad_(True)
        src = [t.clone().requires_grad_(True) for t in src0]
        for p_ in pa.parameters():
            p_.grad = None
        out = pa(q, r, src, None, gc.spatial_shapes, gc.level_start_index)
        w = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
        (out * w).sum().backward()
        res[native] = (out.detach(), q.grad, r.grad, [t.grad for t in src], {n: p_.grad.clone() for n, p_ in pa.named_parameters()})
    a, b = res[False], res[True]
    rel = lambda x, y: float((x - y).abs().max()) / max(float(x.abs().max()), 1e-12)
    assert rel(a[0], b[0]) < 2e-5, rel(a[0], b[0])
    assert rel(a[1], b[1]) < 5e-5 and rel(a[2], b[2]) < 2e-4, (rel(a[1], b[1]), rel(a[2], b[2]))
    for x, y in zip(a[3], b[3]):
        assert rel(x, y) < 5e-5, rel(x, y)
    for n in a[4]:
        assert rel(a[4][n], b[4][n]) < 1e-4, (n, rel(a[4][n], b[4][n]))
    pa.ref_gather_native = True


def test_bf16_fast_path_matches_generic_kernels():
    """bf16 fast path (weight-stationary value / G projections into the pixel-pair layout + G-sampling kernel:
    Linear and bilinear sampling commute) vs the generic bf16 kernels (ref-point gather -> per-row Linear ->
    pixel-major fused sampling): same sampled values up to bf16 rounding of G and of the blend weights.
    Batch of 2 (exercises the (b, q) indexing of the query term), reference points partly outside the maps."""
    from mvgformer_amd import ops
    from mvgformer_amd.decoder import DecoderContext
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = _case("mini5_b2")
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    pa = dec.layers[0].proj_attn
    with torch.no_grad():
        ctx = DecoderContext.build(gc.src_views, gc.spatial_shapes, gc.level_start_index, gc.meta, case.img_size,
                                   torch.bfloat16, case.B)
        r, ref_lvl, inside = ops.project(gc.reference_points, ctx.cams, ctx.levels, ctx.V, case.B)
        ref_lvl[:, 0::5] = ref_lvl[:, 0::5] * 1.5 - 0.25
        x = (gc.tgt + gc.query_pos).contiguous()
        pa.use_fast_path = False
        a = pa.native_sample(x, ref_lvl, ctx.feat, ctx.levels, ctx.V, case.B).float()
        pa.use_fast_path = True
        b = pa.native_sample(x, ref_lvl, ctx.feat, ctx.levels, ctx.V, case.B).float()
    assert torch.isfinite(b).all()
    assert float((a - b).abs().max()) < 0.08 * float(a.abs().max())
    assert float((a - b).abs().mean()) < 4e-3 * float(a.abs().max())


def _per_image_order(order, inside):
    """mvg_bin_pairs' order as (n_img, Lq) rows "image n's pairs: in-image ones in processing order, then the others", whichever of the
    two layouts the call produced: per image [in-image | outside] (single-workgroup variant), or launch-wide [in-image pairs of all
    images, image by image | outside pairs of all images, image by image] (multi-workgroup variant).  Checks the layout on the way."""
    n_img, Lq = inside.shape
    order = order.view(-1).long()
    assert torch.equal(torch.sort(order).values, torch.arange(n_img * Lq, device=order.device))          # a permutation of all pairs
    img = order // Lq
    fl = inside.view(-1)[order].bool()
    per_image = bool((img.view(n_img, Lq) == torch.arange(n_img, device=order.device)[:, None]).all())
    if not per_image:
        n_in = int(fl.sum())
        assert bool(fl[:n_in].all()) and not bool(fl[n_in:].any())                                       # in-image pairs of ALL images first
        assert bool((img[:n_in][1:] >= img[:n_in][:-1]).all()) and bool((img[n_in:][1:] >= img[n_in:][:-1]).all())   # image by image
    rows = []
    for n in range(n_img):
        mine = order[img == n]
        f = inside.view(-1)[mine].bool()
        k = int(f.sum())
        assert bool(f[:k].all()) and not bool(f[k:].any())
        rows.append(mine)
    return torch.stack(rows)


def test_pair_binning_and_masked_sampling_are_exact():
    """mvg_bin_pairs: a permutation per image, in-image pairs first in nondecreasing Morton key of the level-0
    4x4-cell block, masked pairs last.  mvg_msda_gsamp with (pair_mask, order): bit-identical rows for the kept
    pairs, zeros for the masked ones (what dq_decoder.py:585-586 multiplies by 0 anyway)."""
    from mvgformer_amd import ops
    from mvgformer_amd.decoder import DecoderContext
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = _case("mini5_b2")
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    pa = dec.layers[0].proj_attn
    with torch.no_grad():
        ctx = DecoderContext.build(gc.src_views, gc.spatial_shapes, gc.level_start_index, gc.meta, case.img_size,
                                   torch.bfloat16, case.B)
        r, ref_lvl, inside = ops.project(gc.reference_points, ctx.cams, ctx.levels, ctx.V, case.B)
        inside = inside.clone()
        inside[:, 1::3] = 0                                       # make sure both classes occur in every image
        n_img, Lq = inside.shape
        order = ops.bin_pairs(ref_lvl, inside.view(-1), ctx.levels)
        o = _per_image_order(order, inside).cpu()
        H0, W0 = [int(v) for v in ctx.levels.shapes[0]]

        def morton(cx, cy):
            k = 0
            for bit in range(6):
                k |= ((cx >> bit) & 1) << (2 * bit) | ((cy >> bit) & 1) << (2 * bit + 1)
            return k
        ins_c, ref_c = inside.cpu(), ref_lvl.cpu()
        for n in range(n_img):
            assert sorted(o[n].tolist()) == list(range(n * Lq, (n + 1) * Lq))
            keys = []
            for gp in o[n].tolist():
                q = gp - n * Lq
                if not ins_c[n, q]:
                    keys.append(4096)
                    continue
                rx, ry = np.float32(ref_c[n, q, 0, 0]), np.float32(ref_c[n, q, 0, 1])     # fp32 like the kernel
                cx = min(max(int(np.clip(rx, np.float32(0), np.float32(1)) * np.float32(W0)), 0), W0 - 1) >> 2
                cy = min(max(int(np.clip(ry, np.float32(0), np.float32(1)) * np.float32(H0)), 0), H0 - 1) >> 2
                keys.append(morton(cx, cy))
            assert keys == sorted(keys), "image %d not in Morton order" % n
        x = (gc.tgt + gc.query_pos).contiguous()
        pa.sort_pairs = False
        full = pa.native_sample(x, ref_lvl, ctx.feat, ctx.levels, ctx.V, case.B)
        pa.sort_pairs = True
        part = pa.native_sample(x, ref_lvl, ctx.feat, ctx.levels, ctx.V, case.B, pair_mask=inside.view(-1))
        resorted = pa.native_sample(x, ref_lvl, ctx.feat, ctx.levels, ctx.V, case.B)
    keep = inside.view(-1).bool()
    assert torch.equal(resorted, full)
    assert torch.equal(part[keep], full[keep])
    assert int(part[~keep].float().abs().sum()) == 0 and int((~keep).sum()) > 0


def test_chain_a_row_order_and_masked_tile_skip():
    """mvg_chain_attn_pose with (order, o_masked): rows are processed in the binned order and 64-row tiles without an
    in-image row only write attn = 0 / o = o_masked -- bit-identical outputs for every processing order (the order
    inside a bin of mvg_bin_pairs is not deterministic: a row's arithmetic must not depend on its position)."""
    from mvgformer_amd import ops
    torch.manual_seed(5)
    rows = 64 * 9 + 17
    samp = torch.randn(rows, 256, device=DEV).to(torch.bfloat16)
    inside = (torch.rand(rows, device=DEV) < 0.4).to(torch.uint8)
    inside[64:320] = 0                                            # whole masked tiles also without reordering
    mk = lambda n, k: (torch.randn(n, k, device=DEV) / 16).to(torch.bfloat16)
    Wp, W0, W1 = (ops.swizzle_weight(mk(256, 256)) for _ in range(3))
    bp, b0, b1 = (torch.randn(256, device=DEV) * 0.1 for _ in range(3))
    W2, b2 = torch.randn(3, 256, device=DEV) / 16, torch.randn(3, device=DEV)
    wts = (Wp, bp, W0, b0, W1, b1, W2, b2)
    attn0, o0 = ops.chain_attn_pose(samp, inside, *wts)
    o_masked = ops.chain_masked_row_output(*wts)
    perm = torch.argsort(1 - inside.int(), stable=True).to(torch.int32)     # in-image rows first, like mvg_bin_pairs
    shuffled = perm[torch.cat([torch.randperm(int(inside.sum()), device=DEV),
                               int(inside.sum()) + torch.randperm(int((inside == 0).sum()), device=DEV)])].contiguous()
    for order in (None, perm, shuffled):
        attn1, o1 = ops.chain_attn_pose(samp, inside, *wts, order=order, o_masked=o_masked)
        # a row's result does not depend on which tile / position it is processed in: bit-identical
        assert torch.equal(attn1, attn0)
        assert torch.equal(o1[inside != 0], o0[inside != 0])
        assert torch.equal(attn1[inside == 0], torch.zeros_like(attn1[inside == 0]))
        assert float((o1[inside == 0] - o_masked).abs().max()) <= 2e-3 * float(o0.abs().max())


def test_pyramid_producer_handoff_layouts():
    """the packed pyramid is identical whether the producer hands over NCHW fp32 maps
    (transpose kernel), channels-last maps (cast + copy) or writes into the level views of the context's own
    buffer (nothing to do) -- and the decoder output does not depend on the route."""
    from mvgformer_amd import ops
    from mvgformer_amd.decoder import DecoderContext
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = _case("mini5_all")
    gc = case_to_device(case, DEV)
    for dt in (torch.bfloat16, torch.float32):
        ctx = DecoderContext.prepare(gc.spatial_shapes, gc.level_start_index, gc.meta, case.img_size, dt, case.B, DEV)
        a = ctx.pack(gc.src_views).feat.clone()
        nhwc = [s.to(memory_format=torch.channels_last) for s in gc.src_views]
        assert not nhwc[0].is_contiguous()
        b = ctx.pack(nhwc).feat.clone()
        bufs = ctx.pyramid_buffers(channels=gc.src_views[0].shape[1])
        for dst, s in zip(bufs, gc.src_views):
            dst.copy_(s)                                          # the "producer" writes its maps in place
        ops.PROFILE = {}
        try:
            c = ctx.pack(bufs).feat
            assert ops.PROFILE == {}, "in-place produced levels must not be copied"
        finally:
            ops.PROFILE = None
        assert c.data_ptr() == ctx._buffer.data_ptr()
        assert torch.equal(a, b) and torch.equal(a, c)
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        ref = dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes, gc.level_start_index, None,
                  query_pos=gc.query_pos, threshold=0.1)
        ctx = DecoderContext.prepare(gc.spatial_shapes, gc.level_start_index, gc.meta, case.img_size, torch.bfloat16,
                                     case.B, DEV)
        bufs = ctx.pyramid_buffers(channels=gc.src_views[0].shape[1])
        for dst, s in zip(bufs, gc.src_views):
            dst.copy_(s)
        ctx.pack(bufs)
        out = dec(gc.tgt, gc.reference_points, bufs, gc.meta, gc.spatial_shapes, gc.level_start_index, None,
                  query_pos=gc.query_pos, threshold=0.1, context=ctx)
    for x, y in zip(ref[:4], out[:4]):
        assert torch.equal(x, y)


def test_next_layer_query_term_fused_into_chain_b():
    """layer l's chain B emits layer l+1's query term xw = (tgt' + query_pos) W^T + b: same decoder outputs as the
    standalone add + GEMM (identical bf16 rounding of the operand, different fp32 accumulation order)."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = _case("mini5_b2")
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    assert len(dec.layers) >= 2
    outs = []
    with torch.no_grad():
        for fuse in (True, False):
            dec.fuse_next_query_term = fuse
            outs.append(dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes, gc.level_start_index,
                            None, query_pos=gc.query_pos, threshold=0.1))
    a, b = outs
    assert float((a[0] - b[0]).abs().max()) < 2e-2 * float(b[0].abs().max())          # hidden states
    assert float((a[1] - b[1]).abs().max()) < 1.0                                      # 3D points, mm
    assert torch.equal(a[1] == 0, b[1] == 0)
    # without query_pos the fused operand is tgt' alone
    with torch.no_grad():
        dec.fuse_next_query_term = True
        c = dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes, gc.level_start_index, None,
                query_pos=None, threshold=0.1)
        dec.fuse_next_query_term = False
        d = dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes, gc.level_start_index, None,
                query_pos=None, threshold=0.1)
    assert float((c[0] - d[0]).abs().max()) < 2e-2 * float(d[0].abs().max())


@pytest.mark.parametrize("shapes", [[(24, 40)], [(24, 40), (12, 20)], [(32, 48), (16, 24), (8, 12), (4, 6)], [(5, 7), (3, 3)]])
def test_bf16_fast_path_other_level_counts(shapes):
    """The G-sampling kernel's level bookkeeping (flat group -> level row of the reinterpreted Linear outputs, pair
    line offsets, per-level reference points) for 1, 2 and 4 levels and tiny maps, against the generic bf16 kernels;
    with and without the processing order / pair mask."""
    from mvgformer_amd import ops
    from mvgformer_amd.projattn import ProjAttn
    torch.manual_seed(7)
    L_ = len(shapes)
    n_img, Bq, Lq = 4, 2, 96
    sp = torch.tensor(shapes, dtype=torch.int64)
    starts = torch.cat([sp.new_zeros(1), (sp[:, 0] * sp[:, 1]).cumsum(0)[:-1]])
    levels = ops.Levels(sp, starts)
    pa = ProjAttn(256, 1, 8, 8, "ablation_not_use_rayconv").to(DEV)
    with torch.no_grad():
        pa.sampling_offsets.weight.normal_(0, 0.02)
        pa.attention_weights.weight.normal_(0, 0.05)
    pa.compute_dtype = torch.bfloat16
    feat = torch.randn(n_img, levels.S, 256, device=DEV).to(torch.bfloat16)
    x = torch.randn(Bq, Lq, 256, device=DEV)
    r = (torch.rand(n_img, Lq, 1, 2, device=DEV) * 1.3 - 0.15).expand(n_img, Lq, L_, 2).contiguous()
    mask = (torch.rand(n_img * Lq, device=DEV) < 0.7).to(torch.uint8)
    with torch.no_grad():
        pa.use_fast_path = False
        a = pa.native_sample(x, r, feat, levels, n_img // Bq, Bq).float()
        pa.use_fast_path = True
        pa.sort_pairs = False
        b = pa.native_sample(x, r, feat, levels, n_img // Bq, Bq).float()
        pa.sort_pairs = "layer"
        c = pa.native_sample(x, r, feat, levels, n_img // Bq, Bq, pair_mask=mask)
    scale = float(a.abs().max())
    assert torch.isfinite(b).all() and scale > 0
    assert float((a - b).abs().max()) < 0.08 * scale
    assert float((a - b).abs().mean()) < 5e-3 * scale
    keep = mask.bool()
    assert torch.equal(c[keep].float(), b[keep]) and int(c[~keep].float().abs().sum()) == 0


def test_decoder_full_size_vs_oracle_one_layer(O):
    """BASELINE configs[1] at FULL size (5 views, 1024 queries x 15 joints, 960x512 maps), one decoder layer, against
    the CPU oracle run in the test (a few seconds on the host cores): fp32 path at the golden-vector tolerances,
    bf16 path (the benchmarked kernels: binning, masked pairs, G-sampling, fused chains) at the bf16 tolerances.
    3D: the oracle triangulates like the reference, with an fp32 SVD of un-normalised DLT rows; on this 8 m scene
    its own distance from the fp64 solution of ITS OWN rows reaches millimetres on the worst-conditioned of the
    15 360 problems, so the kernel (fp64 Gram + Jacobi) is held against that fp64 solution: 0.1 mm in fp32
    (measured 0.007 mm at the 99.9th percentile, 0.018 mm max; the oracle's fp32 SVD: 2.0 / 3.5 mm)."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg2", seed=1, layers=1)
    prm = to_torch_state(case.weights)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    (w_hs, w_ref, w_r2d, w_p2d, w_cls), ex = O.decoder_layer_forward(
        prm, "layers.0.", case.tgt, case.query_pos, case.reference_points, case.src_views, case.spatial_shapes,
        case.level_start_index, case.meta, case.img_size, 0.1, extras=True)
    B, Lq = w_ref.shape[:2]
    Vh = torch.linalg.svd(ex["dlt_rows"].double())[2]
    X64 = (-Vh[..., 3, :3] / -Vh[..., 3, 3:4]).reshape(B, Lq, 3).float()       # fp64 solution of the oracle's rows
    valid = ex["valid"].repeat_interleave(15, 1)                                # (B, Lq)
    X64 = torch.where(valid[..., None], X64, torch.zeros(()))
    noise = (w_ref - X64).norm(dim=-1).flatten()
    gc = None
    for dt, tol_hs, tol_px, tol_mm in ((torch.float32, 2e-4, 5e-3, 0.1), (torch.bfloat16, 6e-2, 0.05, 6.0)):
        dec = build_decoder_for_case(case, DEV, dtype=dt)
        gc = gc or case_to_device(case, DEV)
        with torch.no_grad():
            hs, refs, r2d, p2d, cls = dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes,
                                          gc.level_start_index, None, query_pos=gc.query_pos, threshold=0.1)
        assert torch.equal(refs[0].cpu().abs().sum(-1) > 0, valid & (w_ref.abs().sum(-1) > 0)), "validity pattern (%s)" % dt
        e_hs = float((hs[0].cpu() - w_hs).abs().max())
        e_px = float((r2d[0].cpu() - w_r2d).abs().max())
        d_mm = (refs[0].cpu() - X64).norm(dim=-1).flatten()
        e_mm, e_max = float(torch.quantile(d_mm, 0.999)), float(d_mm.max())
        e_cls = float((cls[0].cpu() - w_cls).abs().max())
        print("full size %s: |hs| %.2e  2D %.2e px  3D vs fp64 DLT q99.9 %.3f mm max %.3f mm (oracle fp32 SVD: %.3f / %.3f mm)"
              "  cls %.2e" % (dt, e_hs, e_px, e_mm, e_max, float(torch.quantile(noise, 0.999)), float(noise.max()), e_cls))
        assert e_hs < tol_hs and e_px < tol_px and e_mm < tol_mm and e_max < 4 * tol_mm, (str(dt), e_hs, e_px, e_mm, e_max)
        assert e_cls < (1e-5 if dt == torch.float32 else 2e-2), (str(dt), e_cls)


def test_many_views_free_running_vs_fp64_oracle(O):
    """BASELINE configs[4] geometry with 12 of its views (V > 8: second pass of the 8-lane view loops in the view mean,
    the view softmax and the DLT rows), 2 free-running layers, against the oracle evaluated in FP64.  The fp32 oracle
    (= the reference's arithmetic) is itself 1.5 mm / 4 mm / 6 px away from that after layers 1 / 2 (fp32 SVD of the
    DLT rows feeding the next layer's projection); the fp32 HIP path stays within 0.05 mm and 0.05 px of the fp64
    result (measured 0.003 mm, 7e-3 px, features 5e-6 with all 31 views)."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg5", seed=2, NQ=16, V=12, layers=2)
    prm = to_torch_state(case.weights)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    want = O.decoder_forward(prm, case.layers, case.tgt, case.reference_points, case.src_views, case.meta,
                             case.spatial_shapes, case.level_start_index, case.query_pos, case.img_size, threshold=0.1,
                             dtype=torch.float64)
    gc = case_to_device(case, DEV)
    for dt, tol_hs, tol_px, tol_mm in ((torch.float32, 1e-4, 0.05, 0.05), (torch.bfloat16, 8e-2, 1.5, 8.0)):   # bf16: 2 free-running layers
        dec = build_decoder_for_case(case, DEV, dtype=dt)
        with torch.no_grad():
            hs, refs, r2d, p2d, cls = dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes,
                                          gc.level_start_index, None, query_pos=gc.query_pos, threshold=0.1)
        assert torch.equal(refs.cpu().abs().sum(-1) > 0, want[1].abs().sum(-1) > 0), "validity pattern (%s)" % dt
        e_hs = float((hs.cpu() - want[0].float()).abs().max())
        e_px = float((r2d.cpu() - want[2].float()).abs().max())
        e_mm = float((refs.cpu() - want[1].float()).norm(dim=-1).max())
        assert e_hs < tol_hs and e_px < tol_px and e_mm < tol_mm, (str(dt), e_hs, e_px, e_mm)


def test_bf16_decoder_is_deterministic_and_order_independent():
    """The processing order of the pairs (nondeterministic inside a bin) only decides WHERE a pair is computed: the
    bf16 decoder's outputs are bit-identical run to run and with the binning switched off."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = _case("mini5_b2")
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    run = lambda: dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes, gc.level_start_index, None,
                      query_pos=gc.query_pos, threshold=0.1)
    with torch.no_grad():
        a = run()
        b = run()
        for layer in dec.layers:
            layer.proj_attn.sort_pairs = False
        c = run()
    for x, y, z in zip(a[:4], b[:4], c[:4]):
        assert torch.equal(x, y) and torch.equal(x, z)
    for x, y, z in zip(a[4], b[4], c[4]):
        assert torch.equal(x, y) and torch.equal(x, z)


def test_full_size_bf16_forward_is_bit_reproducible():
    """cfg-2 at full size (two workgroups per CU in the chain kernels, every kernel at its real occupancy): three eager
    runs of the bf16 decoder are bit-identical.  (This is the test that a small case cannot replace: a miscompiled
    packed-fp32 sequence in chain A only misbehaved with two workgroups per CU -- 2 % of the rows, run-to-run
    different, inside every tolerance.)"""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg2", seed=0)
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    run = lambda: dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes, gc.level_start_index, None,
                      query_pos=gc.query_pos, threshold=0.1)
    with torch.no_grad():
        a = [t.clone() for t in run()[:4]]
        for _ in range(2):
            b = run()
            torch.cuda.synchronize()
            for x, y in zip(a, b[:4]):
                assert torch.equal(x, y)


def test_side_stream_pyramid_projections_change_nothing():
    """DQDecoder.launch_pyramid_projections: all layers' value planes and G are produced on a side stream, each layer's
    sampler waits on its event.  Same kernels, same inputs -> the outputs must equal the inline schedule BIT FOR BIT,
    eagerly and as a replayed HIP graph with the parallel branch (a missing wait would show up here as a stale or
    half-written vh / G), and the schedule must really be in use."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg2", seed=1)
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    from mvgformer_amd.decoder import DecoderContext
    ctx = DecoderContext.prepare(gc.spatial_shapes, gc.level_start_index, gc.meta, case.img_size, torch.bfloat16, 1, DEV)

    def run():
        ctx.feat = None      # pack the pyramid inside the forward (and inside the graph)
        return dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes, gc.level_start_index, None,
                   query_pos=gc.query_pos, threshold=0.1, context=ctx)
    with torch.no_grad():
        dec.overlap_pyramid = False
        assert dec.fork_side_stream(torch.device(DEV)) is None
        ref = [t.clone() for t in run()[:4]]
        dec.overlap_pyramid = True
        assert dec.fork_side_stream(torch.device(DEV)) is not None
        torch.cuda.synchronize()
        # both issue orders of the side stream: everything up front (grouped launches), and one launch per layer issued
        # just in time behind the previous layer's chain B with one workgroup per CU (the default at one sample per forward)
        assert dec.pyramid_jit == "auto"
        assert [(len(g), s) for g, s in dec.pyramid_launches(ctx)] == [(1, dec.pyramid_jit_slots)] * len(dec.layers)
        for jit in ("auto", "0", "1"):
            dec.pyramid_jit = jit
            if jit == "0":
                assert [(len(g), s) for g, s in dec.pyramid_launches(ctx)] == [(1, 0), (len(dec.layers) - 1, 0)]
            for _ in range(3):
                got = run()
                torch.cuda.synchronize()
                for x, y in zip(ref, got[:4]):
                    assert torch.equal(x, y)
            assert all(l.proj_attn._vp_event is None for l in dec.layers)          # every event consumed / dropped
            assert all(l._after_chain_b is None for l in dec.layers)               # every hook fired / removed
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = run()
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            for x, y in zip(ref, out[:4]):
                assert torch.equal(x, y)
        dec.pyramid_jit = "auto"
        # a decoder that shares one layer (one set of vh / G buffers) must fall back to the inline schedule
        shared = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
        shared.layers = torch.nn.ModuleList([shared.layers[0]] * len(shared.layers))
        assert shared.fork_side_stream(torch.device(DEV)) is None


def test_fused_chains_match_unfused_path_full_size():
    """the same comparison at cfg-2 size: the chain kernels run with two workgroups per CU / one tile per CU there,
    which no small case reaches (a scale-dependent miscompilation hid inside the tolerances of the small test)."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg2", seed=3, layers=1)
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    layer = dec.layers[0]
    outs = []
    for fused in (False, True):
        layer.use_fused_chains = fused
        with torch.no_grad():
            outs.append(layer(gc.tgt, gc.query_pos, gc.reference_points[:, :, None], gc.src_views, gc.spatial_shapes,
                              gc.level_start_index, gc.meta, threshold=0.1))
    a, b = outs
    d2 = (a[2] - b[2]).abs().flatten()
    d3 = (a[1] - b[1]).norm(dim=-1).flatten()
    print("fused vs unfused, full size: hs %.2e  cls %.2e  2D max %.3e px (q99.9 %.3e)  3D max %.3f mm (q99.9 %.3f)"
          % (float((a[0] - b[0]).abs().max()), float((a[4] - b[4]).abs().max()), float(d2.max()),
             float(torch.quantile(d2[:2 ** 24], 0.999)), float(d3.max()), float(torch.quantile(d3, 0.999))))
    assert float((a[4] - b[4]).abs().max()) < 2e-3 and float((a[0] - b[0]).abs().max()) < 3e-2
    assert torch.equal(a[3], b[3])
    assert float(d2.max()) < 1e-2 and float(d3.max()) < 1.0      # measured 1.5e-3 px, 0.29 mm


def test_pyramid_gemms_and_binning_full_size():
    """cfg-2 size (5 x 40 320 pixels, 76 800 pairs): the weight-stationary GEMMs against torch (value as head planes
    vh[img][head][s][32]; G row-major), and the binning kernel's output is a permutation per image with the masked
    pairs last."""
    from mvgformer_amd import ops
    torch.manual_seed(11)
    n_img, S = 5, 40320
    feat = torch.randn(n_img, S, 256, device=DEV).to(torch.bfloat16)
    W = (torch.randn(256, 256, device=DEV) / 16).to(torch.bfloat16)
    bias = torch.randn(256, device=DEV) * 0.1
    vh = torch.empty((n_img, 8, S, 32), dtype=torch.bfloat16, device=DEV)
    ops.value_proj_planes_ws(feat, ops.swizzle_weight(W), bias, vh)
    ref = (feat.float().view(-1, 256) @ W.float().t() + bias).to(torch.bfloat16).view(n_img, S, 8, 32)
    got = vh.permute(0, 2, 1, 3)                                          # (img, s, head, ch)
    d = (got.float() - ref.float()).abs()
    assert float(d.max()) <= 2 ** -7 * float(ref.float().abs().max())      # one bf16 ulp of the largest value
    assert float((d > 0).float().mean()) < 0.02                            # a different fp32 summation order flips few roundings
    Wg = (torch.randn(192, 256, device=DEV) / 16).to(torch.bfloat16)
    G = ops.feat_linear_ws(feat, ops.swizzle_weight(torch.cat([Wg, Wg.new_zeros(64, 256)], 0)), 192)
    refg = (feat.float().view(-1, 256) @ Wg.float().t()).to(torch.bfloat16)
    dg = (G.view(-1, 192).float() - refg.float()).abs()
    assert float(dg.max()) <= 2 ** -7 * float(refg.float().abs().max()) and float((dg > 0).float().mean()) < 0.02
    # binning
    levels = ops.Levels(torch.tensor([[128, 240], [64, 120], [32, 60]]), torch.tensor([0, 30720, 38400]))
    Lq = 15360
    r = torch.rand(n_img, Lq, 3, 2, device=DEV)
    inside = (torch.rand(n_img, Lq, device=DEV) < 0.6).to(torch.uint8)
    order = ops.bin_pairs(r, inside.view(-1), levels).view(n_img, Lq).long()
    base = (torch.arange(n_img, device=DEV) * Lq)[:, None]
    order = _per_image_order(order, inside)         # (checks the layout: in-image pairs first, per image or launch-wide)
    assert torch.equal(torch.sort(order, 1).values, base + torch.arange(Lq, device=DEV)[None])


@pytest.mark.parametrize("n_img,S,n_layers", [(5, 40320, 1), (5, 40320, 3), (3, 1237, 4), (1, 5, 2), (2, 40, 1)])
def test_grouped_pyramid_products_equal_the_single_launches(n_img, S, n_layers):
    """mvg_pyramid_group_ws: several layers' value planes + G from one launch over the packed pyramid (the workgroups of an XCD
    divided among the products) -- every output BIT-identical to the same product launched alone (a product's bytes do not depend
    on the job mix of its launch), also for pyramids with fewer row tiles than workgroups, a ragged last tile and images of fewer
    than 32 pixels (a tile crosses several image boundaries)."""
    from mvgformer_amd import _lib, ops
    torch.manual_seed(5 + n_img)
    feat = torch.randn(n_img, S, 256, device=DEV).to(torch.bfloat16)
    jobs, want, raw = [], [], []
    for l in range(n_layers):
        raw.append((torch.randn(256, 256, device=DEV) / 16).to(torch.bfloat16))
        W = ops.swizzle_weight(raw[-1])
        bias = torch.randn(256, device=DEV) * 0.1
        Wg = (torch.randn(192, 256, device=DEV) / 16).to(torch.bfloat16)
        Wgf = ops.swizzle_weight(torch.cat([Wg, Wg.new_zeros(64, 256)], 0))
        vh = torch.full((n_img, 8, S, 32), 7.0, dtype=torch.bfloat16, device=DEV)
        G = torch.full((n_img * S, 192), 7.0, dtype=torch.bfloat16, device=DEV)
        jobs += [(W, bias, vh, True), (Wgf, None, G, False)]
        want += [ops.value_proj_planes_ws(feat, W, bias, torch.empty_like(vh)), ops.feat_linear_ws(feat, Wgf, 192)]
    ops.pyramid_group_ws(feat, jobs)
    torch.cuda.synchronize()
    for (_, _, out, _), ref in zip(jobs, want):
        assert torch.equal(out.view(-1), ref.view(-1))
    # against fp64 on the bf16 operands (fp32 accumulation, one bf16 rounding of the result)
    b0, vh0 = jobs[0][1], jobs[0][2]
    ref = (feat.double() @ raw[0].double().t() + b0.double()).view(n_img, S, 8, 32).permute(0, 2, 1, 3)
    assert float((vh0.double() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-6


def _emulate_gsamp_one_image(vp, G, xw, ref_lvl, shapes, starts, n, B):
    """torch restatement of msda_gsamp_kernel's arithmetic for image n (all its queries, 8 heads): bilinear gather of
    the head's logits / offsets from the bf16 G (+ xw), the reference's memory reinterpretation, softmax, sampling
    locations, bilinear sampling of the bf16 values with the (corner x attention) weights rounded to bf16."""
    dev = vp.device
    L_ = len(shapes)
    Lq = ref_lvl.shape[1]
    value = vp[n].float()                                                # (8, S, 32) head planes
    Gn = G.view(vp.shape[0], -1, 192)[n].float()                         # (S, 192)
    xwq = xw.view(B, Lq, 192)[n % B]                                     # (Lq, 192)
    ref = ref_lvl[n]                                                     # (Lq, L, 2)

    def bilinear_rows(src, H, W, start, px, py):
        """src rows [start, start+H*W) sampled at pixel coords (px, py) with zero padding -> (..., C) and nothing else"""
        x0, y0 = torch.floor(px), torch.floor(py)
        out = 0
        for dy in (0, 1):
            for dx in (0, 1):
                xi, yi = x0 + dx, y0 + dy
                w = (1 - (px - xi).abs()) * (1 - (py - yi).abs())
                ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
                idx = start + yi.clamp(0, H - 1).long() * W + xi.clamp(0, W - 1).long()
                out = out + (w * ok)[..., None] * src[idx]
        return out

    heads = []
    for m in range(8):
        logits, offs = [], []
        for t in range(L_):
            fg = m * L_ + t
            l, g = fg >> 3, fg & 7
            H, W = shapes[l]
            gx = (ref[:, l, 0] * 2 - 1).clamp(-1.1, 1.1)                    # projattn.py:134 (grid_sample, align_corners=False)
            gy = (ref[:, l, 1] * 2 - 1).clamp(-1.1, 1.1)
            px, py = ((gx + 1) * W - 1) * 0.5, ((gy + 1) * H - 1) * 0.5
            cols = bilinear_rows(Gn[:, 24 * g:24 * g + 24], H, W, starts[l], px, py) + xwq[:, 24 * g:24 * g + 24]
            offs.append(cols[:, :16])
            logits.append(cols[:, 16:])
        logits = torch.cat(logits, 1)                                       # (Lq, 8L)   index = l'*8 + p
        offs = torch.cat(offs, 1).view(Lq, L_ * 8, 2)
        a = torch.softmax(logits, 1)
        acc = torch.zeros((Lq, 32), device=dev)
        for i in range(L_ * 8):
            l2 = i // 8
            H, W = shapes[l2]
            lx = ref[:, l2, 0] + offs[:, i, 0] / W                          # projattn.py:186-191
            ly = ref[:, l2, 1] + offs[:, i, 1] / H
            w_im, h_im = lx * W - 0.5, ly * H - 0.5                         # cuh:295-296
            inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
            x0, y0 = torch.floor(w_im), torch.floor(h_im)
            ai = torch.where(inside, a[:, i], torch.zeros(()).to(dev))
            for dy in (0, 1):
                for dx in (0, 1):
                    xi, yi = x0 + dx, y0 + dy
                    w = (1 - (w_im - xi).abs()) * (1 - (h_im - yi).abs()) * ai
                    ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
                    wq = torch.where(ok, w, torch.zeros(()).to(dev)).to(torch.bfloat16).float()   # kernel: bf16 weights
                    idx = starts[l2] + yi.clamp(0, H - 1).long() * W + xi.clamp(0, W - 1).long()
                    acc = acc + wq[:, None] * value[m][idx]
        heads.append(acc)
    return torch.cat(heads, 1)                                              # (Lq, 256) fp32


def test_gsamp_kernel_vs_torch_emulation_full_size():
    """The G-sampling kernel at cfg-2 size against an independent torch restatement of ITS arithmetic (same bf16 G / value
    operands, same bf16 rounding of the blend weights): agreement to bf16 rounding of the output, for two of the five
    views (one of them through the binned order with masked pairs)."""
    from mvgformer_amd import ops
    from mvgformer_amd.decoder import DecoderContext
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("cfg2", seed=5, layers=1)
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    pa = dec.layers[0].proj_attn
    with torch.no_grad():
        ctx = DecoderContext.build(gc.src_views, gc.spatial_shapes, gc.level_start_index, gc.meta, case.img_size,
                                   torch.bfloat16, 1)
        r, ref_lvl, inside = ops.project(gc.reference_points, ctx.cams, ctx.levels, ctx.V, 1)
        x = (gc.tgt + gc.query_pos).contiguous()
        Wq, bq = pa._fast_query_weights(torch.bfloat16)
        xw = ops.linear(x.reshape(-1, 256), Wq, bq, out_dtype=torch.float32)
        vp = pa.project_values(ctx.feat)
        G = ops.feat_linear_ws(ctx.feat, pa.query_term_weights(torch.bfloat16)[0], 192)
        order = ops.bin_pairs(ref_lvl, inside.view(-1), ctx.levels)
        got = ops.msda_gsamp(vp, G, xw, ref_lvl, ctx.levels, 1, pair_mask=inside.view(-1), order=order).float()
        shapes = [(int(h), int(w)) for h, w in ctx.levels.shapes]
        starts = [int(v) for v in ctx.levels.starts]
        Lq = ref_lvl.shape[1]
        for n in (0, 3):
            want = _emulate_gsamp_one_image(vp, G, xw, ref_lvl, shapes, starts, n, 1)
            keep = inside[n].bool()
            d = (got[n * Lq:(n + 1) * Lq][keep] - want[keep]).abs()
            scale = float(want[keep].abs().max())
            print("gsamp vs emulation, image %d: max %.3e  mean %.3e  (|want| max %.2f)" % (n, float(d.max()), float(d.mean()), scale))
            assert float(d.max()) < 8e-3 * scale and float(d.mean()) < 4e-4 * scale      # output bf16 rounding (2^-8)
            assert int(got[n * Lq:(n + 1) * Lq][~keep].abs().sum()) == 0


def test_deform_forward_full_size_vs_c_oracle():
    """The drop-in op (mvg_msda_forward_f32 / _bf16 through Deformable.deform_forward's contract) at the size of one
    cfg-2 view-layer -- value (2, 40 320, 8, 32), 15 360 queries, 3 levels x 8 points -- against the plain-C
    restatement of the reference kernel (oracle/msda_ref.c, OpenMP on the host).  Locations reach outside [0,1]."""
    from mvgformer_amd import ops
    from oracle import msda_c
    torch.manual_seed(21)
    N, M, D, Lq, P = 2, 8, 32, 15360, 8
    shapes = torch.tensor([[128, 240], [64, 120], [32, 60]], dtype=torch.int64)
    starts = torch.tensor([0, 30720, 38400], dtype=torch.int64)
    S = 40320
    value = torch.randn(N, S, M, D)
    loc = torch.rand(N, Lq, M, 3, P, 2) * 1.2 - 0.1
    wgt = torch.softmax(torch.randn(N, Lq, M, 3 * P), -1).view(N, Lq, M, 3, P)
    want = msda_c.msda_forward(value, shapes, starts, loc, wgt)
    dv = lambda t: t.to(DEV)
    got = ops.msda_forward(dv(value), dv(shapes), dv(starts), dv(loc), dv(wgt)).cpu()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) < 2e-5 * scale
    got16 = ops.msda_forward(dv(value).to(torch.bfloat16), dv(shapes), dv(starts), dv(loc), dv(wgt)).float().cpu()
    want16 = msda_c.msda_forward(value.to(torch.bfloat16).float(), shapes, starts, loc, wgt)
    assert float((got16 - want16).abs().max()) < 6e-3 * scale            # bf16 rounding of the output


def test_non_finite_and_far_away_locations_read_nothing():
    """Sampling locations that are NaN, +-Inf or astronomically far outside the map: the reference kernel's bounds test
    (ms_deform_im2col_cuda.cuh:298, all four comparisons false for NaN) makes them contribute 0 -- and nothing may be
    read out of bounds.  Checked for the drop-in op against the C oracle, and for the bf16 sampling kernels of the
    decoder (a NaN / huge query term, as a diverged training state would produce) for memory safety + finite output
    on the untouched rows."""
    from mvgformer_amd import ops
    from oracle import msda_c
    torch.manual_seed(33)
    N, M, D, Lq, P = 1, 8, 32, 512, 8
    shapes = torch.tensor([[24, 40], [12, 20], [6, 10]], dtype=torch.int64)
    starts = torch.tensor([0, 960, 1200], dtype=torch.int64)
    S = 1260
    value = torch.randn(N, S, M, D)
    loc = torch.rand(N, Lq, M, 3, P, 2)
    bad = [float("nan"), float("inf"), float("-inf"), 1e30, -1e30, 3.4e38, -3.4e38, 2 ** 31 + 0.5, -(2 ** 31) - 0.5]
    flat = loc.view(-1)
    idx = torch.randperm(flat.numel())[:len(bad) * 200]
    flat[idx] = torch.tensor(bad).repeat(200)
    wgt = torch.softmax(torch.randn(N, Lq, M, 3 * P), -1).view(N, Lq, M, 3, P)
    want = msda_c.msda_forward(value, shapes, starts, loc, wgt)
    assert torch.isfinite(want).all()
    dv = lambda t: t.to(DEV)
    got = ops.msda_forward(dv(value), dv(shapes), dv(starts), dv(loc), dv(wgt)).cpu()
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())

    # decoder sampling kernels (generic fused + G-sampling): poison part of the query term / reference points
    from mvgformer_amd.decoder import DecoderContext
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("mini5", seed=2, layers=1)
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    pa = dec.layers[0].proj_attn
    with torch.no_grad():
        ctx = DecoderContext.build(gc.src_views, gc.spatial_shapes, gc.level_start_index, gc.meta, case.img_size,
                                   torch.bfloat16, 1)
        r, ref_lvl, inside = ops.project(gc.reference_points, ctx.cams, ctx.levels, ctx.V, 1)
        x = (gc.tgt + gc.query_pos).contiguous()
        Wq, bq = pa._fast_query_weights(torch.bfloat16)
        xw = ops.linear(x.reshape(-1, 256), Wq, bq, out_dtype=torch.float32)
        vp, G = pa.project_pyramid(ctx.feat)
        clean = ops.msda_gsamp(vp, G, xw, ref_lvl, ctx.levels, 1, pair_mask=None, order=None).float()
        Lq2 = ref_lvl.shape[1]
        poisoned_rows = torch.arange(0, Lq2, 7, device=DEV)
        xw_bad = xw.clone()
        vals = torch.tensor(bad, device=DEV)
        # offsets columns (the first 16 of each 24-column group) of every 7th query
        for k, row in enumerate(poisoned_rows.tolist()):
            xw_bad[row, (k % 8) * 24 + (k % 16)] = vals[k % len(bad)]
        ref_bad = ref_lvl.clone()
        ref_bad[0, 3::11] = float("nan")
        ref_bad[1, 5::13] = 1e30
        for order in (None, ops.bin_pairs(ref_bad, None, ctx.levels)):
            out = ops.msda_gsamp(vp, G, xw_bad, ref_bad, ctx.levels, 1, pair_mask=None, order=order).float()
            torch.cuda.synchronize()                       # a wild read would fault here
            touched = torch.zeros(ctx.V, Lq2, dtype=torch.bool, device=DEV)
            touched[:, poisoned_rows] = True
            touched[0, 3::11] = True
            touched[1, 5::13] = True
            same = out.view(ctx.V, Lq2, 256)[~touched]
            assert torch.equal(same, clean.view(ctx.V, Lq2, 256)[~touched])          # other pairs: bit-identical
        # generic fused kernel (fp32 path): same poison through `oa`
        L = ctx.levels.L
        oa = torch.randn(ctx.V * Lq2 * L, 192, device=DEV)
        oa.view(-1)[torch.randperm(oa.numel(), device=DEV)[:2000]] = vals.repeat(223)[:2000]
        value32 = torch.randn(ctx.V, ctx.feat.shape[1], 256, device=DEV)
        y = ops.msda_fused(value32, oa, ref_bad, ctx.levels)
        torch.cuda.synchronize()
        assert y.shape == (ctx.V * Lq2, 256)


def test_bf16_fast_path_batch_of_two():
    """B = 2 through the benchmarked bf16 kernels (images = V*B, the query term indexed per batch item, chain-B tiles per
    item): against the fp32 kernels of the same decoder (themselves pinned to the reference golden for B = 2,
    mini5_b2), and item by item against single-sample runs (samples are independent units)."""
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    case = build_case("mini5", B=2, seed=17, NQ=8, layers=2)                    # all queries valid: no threshold flips
    gc = case_to_device(case, DEV)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        dec = build_decoder_for_case(case, DEV, dtype=dt)
        with torch.no_grad():
            outs[dt] = [t.float().clone() for t in dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes,
                                                       gc.level_start_index, None, query_pos=gc.query_pos, threshold=0.1)[:4]]
    hs32, ref32, r2d32, p2d32 = outs[torch.float32]
    hs16, ref16, r2d16, p2d16 = outs[torch.bfloat16]
    assert hs16.shape[1] == 2 and torch.isfinite(hs16).all()
    assert float((hs16 - hs32).abs().max()) < 8e-2
    assert float((r2d16 - r2d32).abs().max()) < 0.6 and float((ref16 - ref32).norm(dim=-1).max()) < 8.0
    # the two items really differ, and each equals its single-sample run
    assert float((hs16[:, 0] - hs16[:, 1]).abs().max()) > 1e-2
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    V = case.V
    for b in range(2):
        src_b = [s.view(V, 2, *s.shape[1:])[:, b].contiguous() for s in gc.src_views]      # view-major (V*B, C, H, W)
        meta_b = [{k: ({kk: vv[b:b + 1] for kk, vv in v.items()} if isinstance(v, dict) else v[b:b + 1])
                   for k, v in m.items()} for m in gc.meta]
        with torch.no_grad():
            one = dec(gc.tgt[b:b + 1], gc.reference_points[b:b + 1], src_b, meta_b, gc.spatial_shapes, gc.level_start_index,
                      None, query_pos=gc.query_pos[b:b + 1], threshold=0.1)
        assert float((one[0][:, 0].float() - hs16[:, b]).abs().max()) < 2e-2          # tile numbering changes the fp32 sum order
        assert float((one[1][:, 0].float() - ref16[:, b]).norm(dim=-1).max()) < 1.0
        assert float((one[2][:, 0].float() - r2d16[:, b]).abs().max()) < 0.1


@pytest.mark.parametrize("n_img,Lq", [(3, 8192), (2, 8199), (1, 12345), (5, 15360), (2, 30720), (1, 65536), (4, 8191),
                                      (3, 1), (2, 2049)])
def test_bin_pairs_all_variants_against_a_stable_sort(n_img, Lq):
    """mvg_bin_pairs for sizes on both sides of the single- / multi-workgroup switch (8192 pairs per image), ragged
    slices, the largest supported image (65 536 tokens): every image's slice of `order` is a permutation of its pairs whose
    keys (Morton code of the 4x4-cell block, 4096 for masked pairs) are non-decreasing -- the multiset of keys equals the
    one of a torch sort; both variants agree on it."""
    from mvgformer_amd import _lib, ops
    torch.manual_seed(n_img * 100003 + Lq)
    levels = ops.Levels(torch.tensor([[128, 240], [64, 120], [32, 60]]), torch.tensor([0, 30720, 38400]))
    ref = torch.rand(n_img, Lq, 3, 2, device=DEV) * 1.3 - 0.15          # also outside [0, 1]: clamped keys
    ref[:, ::17] = ref[:, :1].clone()                                   # a hot key
    inside = (torch.rand(n_img, Lq, device=DEV) > 0.35).to(torch.uint8)
    H0, W0 = 128, 240
    cx = (ref[..., 0, 0].clamp(0, 1) * W0).to(torch.int64).clamp(0, W0 - 1) >> 2
    cy = (ref[..., 0, 1].clamp(0, 1) * H0).to(torch.int64).clamp(0, H0 - 1) >> 2
    key = torch.zeros_like(cx)
    for bit in range(6):
        key |= ((cx >> bit) & 1) << (2 * bit) | ((cy >> bit) & 1) << (2 * bit + 1)
    key = torch.where(inside.bool(), key, torch.full_like(key, 4096))
    lib = _lib.load()
    got = {}
    for multi in (0, 1):
        lib.mvg_set_tuning(b"bin_multi", multi)
        try:
            order = _per_image_order(ops.bin_pairs(ref, inside.view(-1), levels), inside)
        finally:
            lib.mvg_set_tuning(b"bin_multi", 1)
        base = torch.arange(n_img, device=DEV).view(-1, 1) * Lq
        local = order - base
        assert int(local.min()) >= 0 and int(local.max()) < Lq
        assert torch.equal(local.sort(dim=1).values, torch.arange(Lq, device=DEV).expand(n_img, Lq))   # permutation
        k = torch.gather(key, 1, local)
        assert bool((k[:, 1:] >= k[:, :-1]).all())                                                     # sorted by key
        got[multi] = k
    assert torch.equal(got[0], got[1]) and torch.equal(got[1], key.sort(dim=1).values)


def test_differentiable_dlt_matches_svd_autograd():
    """geometry_torch.dlt on the GPU (Gram matrix + mvg_sym4_eigh + analytic backward) against torch.linalg.svd and
    its autograd in fp64 on the CPU (what the reference differentiates through, multiview.py:206): points and the
    gradients w.r.t. 2D points, confidences and projection matrices."""
    from mvgformer_amd import geometry_torch as G
    from mvgformer_amd import ops
    torch.manual_seed(4)
    B, V, N = 2, 5, 300
    K = torch.tensor([[1400.0, 0, 960], [0, 1400.0, 540], [0, 0, 1]], dtype=torch.float64)
    Pm = []
    for v in range(V):
        a = 2 * np.pi * v / V
        R = torch.tensor([[np.cos(a), 0, -np.sin(a)], [0, 1, 0], [np.sin(a), 0, np.cos(a)]], dtype=torch.float64)
        C = torch.tensor([4000 * np.sin(a), 200.0 * v, -4000 * np.cos(a)], dtype=torch.float64)
        Pm.append(K @ torch.cat([R, (-R @ C)[:, None]], 1))
    Pm = torch.stack(Pm)[None].repeat(B, 1, 1, 1)                                    # (B,V,3,4)
    X = torch.randn(B, N, 3, dtype=torch.float64) * 500.0
    Xh = torch.cat([X, torch.ones(B, N, 1, dtype=torch.float64)], -1)
    uvw = torch.einsum("bvij,bnj->bvni", Pm, Xh)
    pts = uvw[..., :2] / uvw[..., 2:3] + torch.randn(B, V, N, 2, dtype=torch.float64) * 2.0   # noisy observations
    conf = torch.softmax(torch.randn(B, V, N, dtype=torch.float64), 1)
    wgt = torch.randn(B, N, 3, dtype=torch.float64)
    res = {}
    for dev in ("cpu", DEV):
        leaves = [t.clone().to(dev).requires_grad_(True) for t in (Pm, pts, conf)]
        out = G.dlt(*leaves)
        (out * wgt.to(dev)).sum().backward()
        res[dev] = [out.detach().cpu()] + [l.grad.cpu() for l in leaves]
    names = ("X", "dL/dP", "dL/dpts", "dL/dconf")
    for nm, a, b in zip(names, res["cpu"], res[DEV]):
        err = float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))
        print("dlt %-8s rel err %.2e" % (nm, err))
        assert err < 1e-6, nm
    assert float((res[DEV][0] - X).norm(dim=-1).max()) < 60.0                     # sanity: near the true points (mm)
    # the eigen-solver itself
    A = torch.randn(1000, 4, 4, dtype=torch.float64, device=DEV)
    S = A @ A.transpose(-1, -2)
    w, Vv = ops.sym4_eigh(S)
    assert float((S @ Vv - Vv * w[:, None, :]).abs().max()) < 1e-12 * float(S.abs().max())
    assert float((Vv.transpose(-1, -2) @ Vv - torch.eye(4, dtype=torch.float64, device=DEV)).abs().max()) < 1e-13
    assert torch.allclose(w.sort(-1).values, torch.linalg.eigvalsh(S.cpu()).to(DEV), rtol=1e-12, atol=1e-12 * float(S.abs().max()))


_KNOBS = [("gsamp_pipe", 1, True), ("linear_tiles", 0, True), ("linear_tiles", 2, True), ("gsamp_threads", 128, True), ("gsamp_threads", 512, True),
          ("gsamp_threads", 1024, True), ("gsamp_map", 0, True), ("gsamp_map", 8, True), ("bin_multi", 0, True), ("auto_small", 0, False),
          ("wreg_grid", 256, True), ("wreg_grid", 64, True), ("chain_rm", 64, True), ("chain_rm", 256, False), ("gsamp_lds_pad", 22528, True)]
_KNOB_DEFAULTS = dict(gsamp_pipe=0, linear_tiles=1, gsamp_threads=256, gsamp_map=4, bin_multi=1, auto_small=1, wreg_grid=512, chain_rm=128, gsamp_lds_pad=0)


@pytest.mark.parametrize("key,value,exact", _KNOBS, ids=["%s=%d" % (k, v) for k, v, _ in _KNOBS])
def test_every_kernel_variant_behind_a_tuning_knob(key, value, exact):
    """mvg_set_tuning selects kernel variants that the default configuration never runs (other workgroup sizes, block
    mappings, tile sizes, wave mappings): each of them must still compute the decoder -- bit-identical where the knob only
    changes WHERE an item is computed (sampler / binning / GEMM grid), to bf16 rounding where it changes the order of an
    fp32 sum (chain variants).  12 000-row case: large enough for the multi-workgroup binning and the big-tile chains."""
    from mvgformer_amd import _lib
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    lib = _lib.load()
    case = build_case("cfg2", seed=11, NQ=160, layers=2)            # Lq = 2400 per image, 12 000 pairs
    dec = build_decoder_for_case(case, DEV, dtype=torch.bfloat16)
    gc = case_to_device(case, DEV)
    run = lambda: [t.float().clone() for t in dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes,
                                                  gc.level_start_index, None, query_pos=gc.query_pos, threshold=0.1)[:4]]
    with torch.no_grad():
        ref = run()
        assert lib.mvg_set_tuning(key.encode(), value) == 0
        try:
            got = run()
        finally:
            assert lib.mvg_set_tuning(key.encode(), _KNOB_DEFAULTS[key]) == 0
        again = run()
    for a, b in zip(ref, again):
        assert torch.equal(a, b)                                     # the knob was restored
    if exact:
        for a, b in zip(ref, got):
            assert torch.equal(a, b), key
    else:
        assert float((got[0] - ref[0]).abs().max()) < 4e-2
        assert float((got[1] - ref[1]).norm(dim=-1).max()) < 6.0 and float((got[2] - ref[2]).abs().max()) < 0.5     # the bf16 bars of section 5
    assert lib.mvg_set_tuning(b"no_such_knob", 1) != 0 and lib.mvg_set_tuning(key.encode(), -7) != 0


@pytest.mark.parametrize("N,Lq,M,D,L,P,dt", [(2, 77, 8, 32, 3, 8, torch.float32), (1, 1000, 8, 32, 3, 8, torch.bfloat16), (3, 33, 4, 16, 2, 3, torch.float32),
                                             (1, 5, 2, 64, 1, 5, torch.float32), (2, 50, 3, 24, 2, 4, torch.float32)])
def test_forward_operator_mappings_agree_bit_for_bit(N, Lq, M, D, L, P, dt):
    """mvg_msda_forward (Deformable.deform_forward's drop-in): fwd_map = 1 (default: one head per workgroup, loads in batches of 4
    samples, for D = 32 every lane of a unit prepares one sample of a batch of 8; needs 256 % (D / 4) == 0, other D fall back), 2 (the
    same without sharing the per-sample arithmetic) and 0 (a wavefront takes a query's heads): identical outputs,
    with locations that leave the maps, L * P not a multiple of the batch and query counts that do not fill the last workgroup."""
    from mvgformer_amd import _lib
    from mvgformer_amd import deformable as DF
    lib = _lib.load()
    gen = torch.Generator().manual_seed(N * 1000 + Lq)
    shapes = torch.tensor([[20, 31], [11, 16], [5, 8]][:L], dtype=torch.long)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = torch.randn(N, S, M, D, generator=gen).to(DEV).to(dt)
    loc = (torch.rand(N, Lq, M, L, P, 2, generator=gen) * 1.3 - 0.15).to(DEV)
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=gen), -1).view(N, Lq, M, L, P).to(DEV)
    out = {}
    try:
        for mp in (1, 2, 0):
            assert lib.mvg_set_tuning(b"fwd_map", mp) == 0
            out[mp] = DF.deform_forward(value, shapes.to(DEV), starts.to(DEV), loc, attn, 64).clone()
    finally:
        assert lib.mvg_set_tuning(b"fwd_map", 1) == 0
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2])


@pytest.mark.parametrize("cfg,kw", [("cfg2", dict(NQ=160, layers=2)), ("cfg4", dict(layers=2)), ("cfg2", dict(NQ=3, layers=1))],
                         ids=["cfg2_160q", "cfg4", "cfg2_3q"])
def test_fp32_sampler_mappings_agree_bit_for_bit(cfg, kw):
    """mvg_msda_gfused_f32: a wavefront takes 8 neighbouring pairs of one head, one head per workgroup; gfused_chunk = which pair
    blocks an XCD takes.  Every (pair, head) unit is computed independently of the mapping: the fp32 decoder outputs are identical,
    also for a launch whose last wavefronts are partly out of range (3 queries)."""
    from mvgformer_amd import _lib
    from mvgformer_amd.factory import build_decoder_for_case, case_to_device
    lib = _lib.load()
    case = build_case(cfg, seed=5, **kw)
    dec = build_decoder_for_case(case, DEV, dtype=torch.float32)
    gc = case_to_device(case, DEV)
    run = lambda: [t.float().clone() for t in dec(gc.tgt, gc.reference_points, gc.src_views, gc.meta, gc.spatial_shapes,
                                                  gc.level_start_index, None, query_pos=gc.query_pos, threshold=0.1)[:4]]
    with torch.no_grad():
        ref = run()
        for key, value, default in ((b"gfused_chunk", 0, 4), (b"gfused_chunk", 1, 4), (b"gfused_chunk", 16, 4)):
            assert lib.mvg_set_tuning(key, value) == 0
            try:
                got = run()
            finally:
                assert lib.mvg_set_tuning(key, default) == 0
            for a, b in zip(ref, got):
                assert torch.equal(a, b), (key, value)


@pytest.mark.parametrize("rows,N,K", [(76800, 256, 256), (1000, 192, 256), (15360, 1024, 256), (777, 256, 1024), (33, 64, 32)])
def test_linear_wgrad_and_autograd_function_vs_fp64(rows, N, K):
    """mvg_linear_wgrad_f32 (dW = dY^T X from row-major operands, split over row slices) and the LinearF32S autograd Function
    (forward / dgrad / wgrad on this library's GEMMs) against fp64: errors in units of sum|a||b| like the forward GEMM's."""
    from mvgformer_amd import ops
    from mvgformer_amd.functions import LinearF32S
    gen = torch.Generator(device="cpu").manual_seed(rows + N)
    dy = torch.randn(rows, N, generator=gen).to(DEV)
    x = torch.randn(rows, K, generator=gen).to(DEV)
    dw = ops.linear_wgrad(dy, x)
    want = dy.double().t() @ x.double()
    scale = dy.double().abs().t() @ x.double().abs()
    assert float(((dw.double() - want).abs() / scale).max()) < 2e-6
    assert torch.equal(dw, ops.linear_wgrad(dy, x))                       # deterministic (slice-ordered sum)
    if N % 32 == 0 and K % 32 == 0:
        w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(DEV).requires_grad_(True)
        b = torch.randn(N, generator=gen).to(DEV).requires_grad_(True)
        xi = x.clone().requires_grad_(True)
        y = LinearF32S.apply(xi, w, b, True)
        y.backward(dy)
        w64, b64, x64 = w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True), x.double().requires_grad_(True)
        y64 = torch.relu(x64 @ w64.t() + b64)
        y64.backward(dy.double())
        for got, ref in ((y, y64), (xi.grad, x64.grad), (w.grad, w64.grad), (b.grad, b64.grad)):
            assert float((got.double() - ref.detach()).abs().max()) < 3e-6 * max(1.0, float(ref.detach().abs().max())) * (rows ** 0.5 if got is w.grad or got is b.grad else 1.0)


@pytest.mark.parametrize("rows,N,K", [(76800, 256, 256), (15360, 1024, 256), (777, 256, 1024), (4000, 32, 256), (33, 64, 32)])
def test_linear_wgrad_with_bias_gradient(rows, N, K):
    """mvg_linear_wgrad_bias_f32: dW against fp64 (the bar of mvg_linear_wgrad_f32) and bit-identical run after run (the slices'
    partials are added in slice order); db = column sums of dY against fp64."""
    from mvgformer_amd import ops
    gen = torch.Generator(device="cpu").manual_seed(rows + 3 * N)
    dy = torch.randn(rows, N, generator=gen).to(DEV)
    x = torch.randn(rows, K, generator=gen).to(DEV)
    dw, db = ops.linear_wgrad_bias(dy, x)
    want_w = dy.double().t() @ x.double()
    assert float(((dw.double() - want_w).abs() / (dy.double().abs().t() @ x.double().abs())).max()) < 2e-6
    want = dy.double().sum(0)
    assert float((db.double() - want).abs().max()) < 2e-6 * float(dy.double().abs().sum(0).max())
    for _ in range(5):
        dw2, db2 = ops.linear_wgrad_bias(dy, x)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
    dw3, none = ops.linear_wgrad_bias(dy, x, want_bias=False)
    assert none is None and torch.equal(dw, dw3)


@pytest.mark.parametrize("frac_valid", [1.0, 0.4, 0.0])
def test_dense_dlt_function_matches_the_torch_form(frac_valid):
    """geometry_torch.DenseDLT (mvg_dlt_forward / mvg_dlt_backward: one launch each way over the dense token grid) against
    geometry_torch.dlt on the tokens of the valid queries (itself pinned to the SVD's autograd above): points and the gradients
    w.r.t. the 2D points and confidences; zeros for the tokens of the other queries."""
    from mvgformer_amd import geometry_torch as G
    torch.manual_seed(9)
    B, V, NQ, J = 2, 5, 37, 15
    Lq = NQ * J
    K = torch.tensor([[1400.0, 0, 960], [0, 1400.0, 540], [0, 0, 1]], dtype=torch.float64)
    Pm = []
    for v in range(V):
        a = 2 * np.pi * v / V
        R = torch.tensor([[np.cos(a), 0, -np.sin(a)], [0, 1, 0], [np.sin(a), 0, np.cos(a)]], dtype=torch.float64)
        C = torch.tensor([4000 * np.sin(a), 200.0 * v, -4000 * np.cos(a)], dtype=torch.float64)
        Pm.append(K @ torch.cat([R, (-R @ C)[:, None]], 1))
    Pm = torch.stack(Pm)[None].repeat(B, 1, 1, 1)
    X = torch.randn(B, Lq, 3, dtype=torch.float64) * 500.0
    uvw = torch.einsum("bvij,bnj->bvni", Pm, torch.cat([X, torch.ones(B, Lq, 1, dtype=torch.float64)], -1))
    pts = (uvw[..., :2] / uvw[..., 2:3] + torch.randn(B, V, Lq, 2, dtype=torch.float64) * 2.0).float().to(DEV)
    conf = torch.softmax(torch.randn(B, V, Lq), 1).to(DEV)
    Pm = Pm.float().to(DEV)
    wgt = torch.randn(B, Lq, 3, device=DEV)
    valid = (torch.rand(B, NQ) < frac_valid).to(DEV)
    tok = valid.view(B, NQ, 1).expand(B, NQ, J).reshape(B, Lq)
    p1, c1 = pts.clone().requires_grad_(True), conf.clone().requires_grad_(True)
    out = G.DenseDLT.apply(p1, c1, Pm, valid.to(torch.uint8), J)
    (out * wgt).sum().backward()
    assert bool((out[~tok] == 0).all()) and bool((p1.grad.transpose(1, 2)[~tok] == 0).all()) and bool((c1.grad.transpose(1, 2)[~tok] == 0).all())
    if not bool(tok.any()):
        return
    # truth: the SVD form and its autograd in fp64 on the CPU, from the same fp32 inputs
    bi, ti = tok.cpu().nonzero(as_tuple=True)
    p2, c2 = pts.double().cpu().requires_grad_(True), conf.double().cpu().requires_grad_(True)
    ref = G.dlt(Pm.double().cpu()[bi], p2[bi, :, ti].unsqueeze(2), c2[bi, :, ti].unsqueeze(2))[:, 0]
    (ref * wgt.double().cpu()[bi, ti]).sum().backward()
    assert float((out.cpu()[bi, ti] - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    for nm, a, b in (("pts", p1.grad, p2.grad), ("conf", c1.grad, c2.grad)):
        err = float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))
        print("dense dlt dL/d%-5s rel err %.2e" % (nm, err))
        assert err < 1e-6, nm      # the torch form (fp32 rows) is at 1e-4 for the confidences


def test_pyramid_products_on_two_part_fp16_operands():
    """mvg_pyramid_f32h (three fp16 MFMAs per product, per-row power-of-two scales) against fp64: the error bar of the six-product
    bf16 form (3.5e-7 of sum|a||w| + |b| on well-scaled rows, 1.5e-6 with entries spread over six decades -- the fp32 accumulation's
    share in both forms), rows spread over 12 decades, zero rows, a ragged last tile; a row's result does not depend on its position
    (permutation: bit-identical); Inf / NaN stay in their rows."""
    from mvgformer_amd import _lib, ops
    lib = _lib.load()
    gen = torch.Generator().manual_seed(21)
    rows = 64 * 37 + 13
    rnd = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    Wv, bv, Wg = rnd(256, 256) / 16, rnd(256), rnd(192, 256) / 16
    (Wv_h, sv), (Wg_h, sg) = ops.split_swizzle_weight_h2(Wv), ops.split_swizzle_weight_h2(Wg)
    lin64 = lambda x, W, b=None: x.double() @ W.double().t() + (0 if b is None else b.double())
    scale = lambda x, W, b=None: x.double().abs() @ W.double().abs().t() + (0 if b is None else b.double().abs()) + 1e-300
    base = rnd(1, rows, 256)
    cases = {"randn": (base, 3.5e-7), "rows over 12 decades": (base * torch.exp(rnd(1, rows, 1) * 5), 3.5e-7),
             "entries over 6 decades, zero rows": (base * torch.exp(rnd(1, rows, 256) * 3) * (torch.rand(1, rows, 1, generator=gen).to(DEV) > 0.1), 1.5e-6)}
    for name, (feat, bar) in cases.items():
        feat = feat.contiguous()
        value, G = ops.pyramid_f32h(feat, Wv_h, sv, bv, Wg_h, sg, 192)
        ev = float(((value[0].double() - lin64(feat[0], Wv, bv)).abs() / scale(feat[0], Wv, bv)).max())
        eg = float(((G.double() - lin64(feat[0], Wg)).abs() / scale(feat[0], Wg)).max())
        print("pyramid f32h [%s]: value %.2e G %.2e" % (name, ev, eg))
        assert ev < bar and eg < bar, name
        perm = torch.randperm(rows, generator=gen).to(DEV)
        v2, G2 = ops.pyramid_f32h(feat[:, perm].contiguous(), Wv_h, sv, bv, Wg_h, sg, 192)
        assert torch.equal(v2[0], value[0][perm]) and torch.equal(G2, G[perm])
    feat = base.clone()
    feat[0, 5, 7], feat[0, 70, 0] = float("inf"), float("nan")
    value, G = ops.pyramid_f32h(feat, Wv_h, sv, bv, Wg_h, sg, 192)
    bad = ~torch.isfinite(value[0]).all(1)
    assert bad.nonzero().flatten().tolist() == [5, 70] and bool(torch.isfinite(G[[4, 6, 69, 71]]).all())


def test_fp32_chains_on_two_part_fp16_operands():
    """mvg_chain_attn_pose_f32h / mvg_chain_update_ffn_class_f32h (three fp16 MFMAs per product, per-row scales exchanged between the
    wavefronts) against fp64, with the bars tools/check_f32s.py applies to the six-product kernels; masked rows of mixed tiles equal the
    cached constant bit for bit; a row's result does not depend on its tile (rows permuted / a second launch with other neighbours)."""
    from mvgformer_amd import ops
    gen = torch.Generator().manual_seed(33)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    mk = lambda n, k: (rnd(n, k) / k ** 0.5, rnd(n) * 0.1)
    sp2 = ops.split_swizzle_weight_h2
    lin64 = lambda x, W, b=None: x.double() @ W.double().t() + (0 if b is None else b.double())
    # ---- chain A
    R = 32 * 61 + 7
    samp = rnd(R, 256) * torch.exp(rnd(R, 1))                 # rows of different magnitudes
    inside = (torch.rand(R, generator=gen) < 0.67).to(torch.uint8).to(DEV)
    (Wp, bp), (W0, b0), (W1, b1), (W2, b2) = mk(256, 256), mk(256, 256), mk(256, 256), mk(3, 256)
    (Wp_h, swp), (W0_h, sw0), (W1_h, sw1) = sp2(Wp), sp2(W0), sp2(W1)
    wts = (Wp_h, swp, bp, W0_h, sw0, b0, W1_h, sw1, b1, W2.contiguous(), b2)
    o_masked = ops.chain_masked_row_output_f32h(*wts)
    order = torch.argsort(1 - inside.int(), stable=True).to(torch.int32)
    a64 = lin64(samp, Wp, bp) * inside.double()[:, None]
    o64 = lin64(torch.relu(lin64(torch.relu(lin64(a64, W0, b0)), W1, b1)), W2, b2)
    scale = samp.double().abs() @ Wp.double().abs().t() + bp.double().abs()
    res = {}
    for key, (od, om) in {"plain": (None, None), "ordered": (order, o_masked)}.items():
        attn, o = ops.chain_attn_pose_f32h(samp, inside, *wts, order=od, o_masked=om)
        assert float(((attn.double() - a64).abs() / scale).max()) < 4e-7
        assert float((o.double() - o64).abs().max()) < 1e-6 * (1.0 + float(o64.abs().max()))
        assert bool((o[inside == 0] == o_masked).all()) and bool((attn[inside == 0] == 0).all())
        res[key] = (attn, o)
    assert torch.equal(res["plain"][0], res["ordered"][0]) and torch.equal(res["plain"][1], res["ordered"][1])   # tile membership does not matter
    from mvgformer_amd import _lib
    for r in (32, 64):                                            # nor does the tile size (the launcher picks it by the row count)
        try:
            assert _lib.load().mvg_set_tuning(b"f32h_rows", r) == 0
            attn, o = ops.chain_attn_pose_f32h(samp, inside, *wts, order=order, o_masked=o_masked)
        finally:
            assert _lib.load().mvg_set_tuning(b"f32h_rows", 0) == 0
        assert torch.equal(attn, res["ordered"][0]) and torch.equal(o, res["ordered"][1])
    # ---- chain B
    B, NQ, J, V = 1, 41, 15, 3
    rows = B * NQ * J
    attn, tgt, qpos = rnd(V * rows, 256), rnd(rows, 256), rnd(rows, 256)
    (Wu, bu), (Wf1, bf1), (Wf2, bf2), (Wc, bc), (Wn, bn) = mk(256, 256), mk(1024, 256), mk(256, 1024), mk(2, 256), mk(192, 256)
    g2, be2, g3, be3 = (1 + 0.1 * rnd(256) for _ in range(4))
    (Wu_h, su), (Wf1_h, s1), (Wf2_h, s2), (Wn_h, sn) = sp2(Wu), sp2(Wf1), sp2(Wf2), sp2(Wn)
    args = (Wu_h, su, bu, g2, be2, Wf1_h, s1, bf1, Wf2_h, s2, bf2, g3, be3, Wc.contiguous(), bc)
    nxt = (qpos, Wn_h, sn, torch.cat([bn, bn.new_zeros(64)]), 192)
    out = [t.clone() for t in ops.chain_update_ffn_class_f32h(attn, V, tgt, *args, 0.5, B, NQ, J, next_query_proj=nxt)]
    ln = lambda x, g, b: torch.nn.functional.layer_norm(x, (256,), g.double(), b.double(), 1e-5)
    t1 = ln(tgt.double() + lin64(attn.double().view(V, rows, 256).mean(0), Wu, bu), g2, be2)
    y = ln(t1 + lin64(torch.relu(lin64(t1, Wf1, bf1)), Wf2, bf2), g3, be3)
    pr = torch.sigmoid(lin64(y, Wc, bc)).view(B, NQ, J, 2).mean(2)
    assert float((out[0].double() - y).abs().max()) < 2e-5 and float((out[1].double() - pr).abs().max()) < 1e-6
    assert float((out[4].double() - lin64(y + qpos.double(), Wn, bn)).abs().max()) < 3e-5
    assert bool((out[2].bool() == (pr[..., 1] > 0.5)).all())
    # both tile sizes (the launcher picks by the row count): identical
    from mvgformer_amd import _lib as _l
    for r in (32, 64):
        try:
            assert _l.load().mvg_set_tuning(b"f32h_rows", r) == 0
            outr = ops.chain_update_ffn_class_f32h(attn, V, tgt, *args, 0.5, B, NQ, J, next_query_proj=nxt)
        finally:
            assert _l.load().mvg_set_tuning(b"f32h_rows", 0) == 0
        assert all(torch.equal(a, b) for a, b in zip(outr, out))
    # the persons in another order: every person's rows are unchanged (a tile is 2 or 4 persons)
    perm = torch.randperm(NQ, generator=gen).to(DEV)
    rperm = (perm[:, None] * J + torch.arange(J, device=DEV)[None]).reshape(-1)
    out2 = ops.chain_update_ffn_class_f32h(attn.view(V, rows, 256)[:, rperm].reshape(V * rows, 256).contiguous(), V, tgt[rperm].contiguous(), *args, 0.5, B, NQ, J,
                                           next_query_proj=(qpos[rperm].contiguous(),) + nxt[1:])
    assert torch.equal(out2[0], out[0][rperm]) and torch.equal(out2[4], out[4][rperm]) and torch.equal(out2[1][0], out[1][0][perm])
