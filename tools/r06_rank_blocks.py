"""Which share of the feature pyramid can a rank's queries sample, for different ways of cutting the 32 x 32 person grid into 8 ranks?
(tile-masked pyramid products are worth building only if a balanced block shape brings the slowest rank under 60 %.)
Single-rank cfg-2 forward -> the points every layer projects -> per rank and layer the share of the pyramid's pixels in 16 x 16 tiles
within the offsets' reach (+- 9 cells) of an in-image reference point of the rank's persons (bench.py's `touched_pyramid_share`).
python tools/r06_rank_blocks.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd import ops  # noqa: E402
from mvgformer_amd.decoder import DecoderContext  # noqa: E402
from mvgformer_amd.factory import build_decoder_for_case, case_to_device  # noqa: E402
from mvgformer_amd.synthetic import build_case  # noqa: E402

dev = "cuda"
case = build_case("cfg2", seed=0)
dec = build_decoder_for_case(case, dev, torch.bfloat16)
g = case_to_device(case, dev)
ctx = DecoderContext.prepare(g.spatial_shapes, g.level_start_index, g.meta, case.img_size, torch.bfloat16, 1, dev)
NQ, J, V, Ly = case.NQ, 15, case.V, case.layers
N = int(round(NQ ** 0.5))
with torch.no_grad():
    out = dec(g.tgt, g.reference_points, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None, query_pos=g.query_pos,
              threshold=0.1, context=ctx)
    pts = [g.reference_points] + [out[1][l] for l in range(Ly - 1)]
    shp = [(int(h), int(w)) for h, w in g.spatial_shapes.tolist()]
    person = torch.arange(NQ, device=dev)
    row, col = person // N, person % N                       # 'sample_space' grid: person i at (x index i // N, y index i % N)
    schemes = {
        "8 strips of 4 rows (dist.shard_bounds today)": (row // 4),
        "2 x 4 blocks of 16 x 8 persons": (row // 16) * 4 + (col // 8),
        "4 x 2 blocks of 8 x 16 persons": (row // 8) * 2 + (col // 16),
        "interleaved (person i -> rank i % 8)": person % 8,
    }
    for name, rank_of in schemes.items():
        worst, lines = 0.0, []
        for rk in range(8):
            sel = (rank_of == rk).nonzero().view(-1)
            tok = (sel[:, None] * J + torch.arange(J, device=dev)[None]).view(-1)
            per_layer = []
            for X in pts:
                Xr = X.reshape(1, -1, 3)[:, tok].float().contiguous()
                r_, _, ins_ = ops.project(Xr, ctx.cams, ctx.levels, V, 1)
                num = den = 0.0
                for (Hl, Wl) in shp:
                    th, tw = -(-Hl // 16), -(-Wl // 16)
                    px, py = r_[..., 0] * Wl, r_[..., 1] * Hl
                    hit = torch.zeros((r_.shape[0], th, tw), dtype=torch.bool, device=dev)
                    m = ins_.view(r_.shape[0], -1).bool()
                    img = torch.arange(r_.shape[0], device=dev)[:, None].expand_as(px)
                    for dy in (-9, 0, 9):
                        for dx in (-9, 0, 9):
                            tx = ((px + dx) // 16).long().clamp(0, tw - 1)
                            ty = ((py + dy) // 16).long().clamp(0, th - 1)
                            hit[img[m], ty[m], tx[m]] = True
                    num += float(hit.float().mean()) * Hl * Wl
                    den += Hl * Wl
                per_layer.append(num / den)
            lines.append("rank %d: " % rk + " ".join("%.2f" % v for v in per_layer))
            worst = max(worst, sum(per_layer) / len(per_layer))
        print("== %s: slowest rank touches %.0f %% of the pyramid (mean over the layers)" % (name, 100 * worst))
        print("   " + " | ".join(lines))
