"""Seeded case definitions shared by make_golden.py (reference side) and the tests
(oracle / HIP side).  Pure numpy + torch; no reference import here."""
import numpy as np
import torch

# name -> spec for mvgformer_amd.synthetic.build_case + what to record
LAYER_CASES = {
    # 5 views, distortion != 0, some queries outside some images, all queries valid
    "mini5_all": dict(config="mini5", seed=3, layers=2, projattn=True, triangulation=True),
    # same geometry, about half of the queries pass the 0.1 threshold
    "mini5_half": dict(config="mini5", seed=9, layers=2, valid_fraction=0.5),
    # layer 1 ends with NO query above the threshold -> the forced query (0,0) path (dq_decoder.py:620-623)
    "mini5_empty": dict(config="mini5", seed=8, layers=2, valid_fraction=0.5),
    # batch of 2 with different validity per item (exercises the pad/scatter logic)
    "mini5_b2": dict(config="mini5", seed=13, layers=2, B=2, NQ=6, valid_fraction=0.5),
    # BASELINE.json configs[0]: 1 sample, 2 views 256x256, 64 queries, 1 layer
    "cfg1": dict(config="cfg1", seed=0, layers=1),
    # round 3: Shelf-like geometry (3 views, k = p = 0, 400x304 network image), about half of the queries valid
    "mini3_shelf": dict(config="mini3s", seed=23, layers=2, valid_fraction=0.5, triangulation=True),
    # round 3: 9 views (> the 8 lanes per problem of the triangulation kernel), all queries valid
    "mini9": dict(config="mini9", seed=19, layers=2, triangulation=True),
}


def msda_case(name):
    """Inputs of the sampling op (value, shapes, starts, loc, weight) for a named case."""
    if name == "small_f32":
        rs = np.random.RandomState(101)
        N, M, D, Lq, P = 2, 8, 32, 37, 8
        shapes = [(12, 20), (6, 10), (3, 5)]
        lo, hi = -0.15, 1.15          # ~20 % of the points fall outside [0,1]
    elif name == "ragged_f32":
        rs = np.random.RandomState(102)
        N, M, D, Lq, P = 1, 8, 32, 5, 8
        shapes = [(7, 13), (1, 9), (5, 1)]   # degenerate 1-row / 1-col levels
        lo, hi = -0.3, 1.3
    elif name == "edge_f32":
        rs = np.random.RandomState(103)
        N, M, D, Lq, P = 1, 4, 16, 9, 4      # non-default head count / channels / points
        shapes = [(6, 8), (3, 4)]
        lo, hi = 0.0, 1.0
    else:
        raise KeyError(name)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = rs.standard_normal((N, S, M, D)).astype(np.float32)
    loc = (lo + (hi - lo) * rs.rand(N, Lq, M, L, P, 2)).astype(np.float32)
    if name == "edge_f32":
        # exact texel centres, the -0.5/H and 1+0.5/H borders, and just-outside points
        H0, W0 = shapes[0]
        loc[0, 0, :, 0, :, 0] = (3 + 0.5) / W0
        loc[0, 0, :, 0, :, 1] = (2 + 0.5) / H0
        loc[0, 1, :, 0, :, 0] = -0.5 / W0
        loc[0, 2, :, 0, :, 1] = 1.0 + 0.5 / H0
        loc[0, 3, :, 0, :, 0] = 0.0
        loc[0, 4, :, 0, :, 1] = 1.0
        loc[0, 5, :, :, :, :] = -0.01
        loc[0, 6, :, :, :, :] = 1.01
    wgt = rs.rand(N, Lq, M, L, P).astype(np.float32)
    wgt = wgt / wgt.reshape(N, Lq, M, -1).sum(-1)[..., None, None]
    shapes_t = torch.tensor(shapes, dtype=torch.long)
    starts = torch.cat([shapes_t.new_zeros(1), (shapes_t[:, 0] * shapes_t[:, 1]).cumsum(0)[:-1]])
    return dict(value=torch.from_numpy(value), shapes=shapes_t, starts=starts,
                loc=torch.from_numpy(loc), weight=torch.from_numpy(wgt))


def threshold_margin(cls_list, thr):
    """smallest |prob - thr| over all layers (parity tests need this >> rounding)."""
    return min(float((c[..., 1] - thr).abs().min()) for c in cls_list)


# caller glue (make_golden_caller.py): two loader batches of B=2 through the reference's model forward + validate_3d
CALLER_CASE = dict(config="mini5", seed=31, layers=2, B=2, batches=2, threshold=0.1, valid_fraction=0.5)


def caller_embeddings(NQ, C=256, J=15, seed=77):
    """joint_embedding.weight (J, 2C), instance_embedding.weight (NQ, 2C) -- seeded stand-ins for the checkpoint's."""
    rs = np.random.RandomState(seed)
    je = torch.from_numpy(rs.standard_normal((J, 2 * C)).astype(np.float32))
    ie = torch.from_numpy(rs.standard_normal((NQ, 2 * C)).astype(np.float32))
    return je, ie


# gradient fixtures (make_golden_grad.py): layer 0 of these LAYER_CASES under autograd; ``indices`` = the matched-query
# lists a training step passes (dq_decoder.py:900-901), None = the class-head filter (validation inside a training loop)
GRAD_CASES = {
    "mini5_all": dict(indices=[[1, 4, 5, 7, 10]]),
    "mini5_half": dict(indices=None),
    "mini5_b2": dict(indices=[[0, 3], [1, 2, 5]]),
}
GRAD_ROW_STRIDE = 4          # 2-D weight gradients above 16 384 elements are stored as rows [::4]


def msda_grad_output(name, shape):
    seed = {"small_f32": 7, "ragged_f32": 8, "edge_f32": 9}[name]
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(tuple(shape)).astype(np.float32))


def layer_loss(outputs, seed=5):
    """Fixed scalar of a layer's 5-tuple: seeded linear functionals, scaled so every output matters (features O(1),
    3D in mm, 2D in px, probabilities)."""
    hs, ref3d, ref2d, proj2d, prob = outputs
    rs = np.random.RandomState(seed)

    def w(t):
        return torch.from_numpy(rs.standard_normal(tuple(t.shape))).to(device=t.device, dtype=t.dtype)
    return ((hs * w(hs)).sum() + 1e-2 * (ref3d * w(ref3d)).sum() + 1e-1 * (ref2d * w(ref2d)).sum()
            + 10.0 * (prob * w(prob)).sum())


def subsample_grad(name, g):
    """what the fixture keeps of a parameter gradient (numpy / torch, any device)."""
    if g.ndim == 2 and g.shape[0] * g.shape[1] > 16384:
        return g[::GRAD_ROW_STRIDE]
    return g
