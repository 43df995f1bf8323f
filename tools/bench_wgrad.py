"""mvg_linear_wgrad_bias_f32 (weight-gradient launch + slice reduction) over the number of row slices, at the training step's shapes.
python tools/bench_wgrad.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mvgformer_amd import ops  # noqa: E402

shapes = [(76800, 256, 256), (230400, 192, 256), (76800, 32, 256), (15360, 256, 256), (15360, 1024, 256), (15360, 256, 1024), (15360, 32, 256)]
for rows, N, K in shapes:
    dy = torch.randn(rows, N, device="cuda")
    x = torch.randn(rows, K, device="cuda")
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    default = max(1, min((rows + 255) // 256, 128, (512 + tiles - 1) // tiles))
    line = []
    for sp in sorted({8, 16, 32, 48, 64, 96, 128, 192, 256, 384, default}):
        if sp * 32 > rows:
            continue
        for _ in range(3):
            ops.linear_wgrad_bias(dy, x, splits=sp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear_wgrad_bias(dy, x, splits=sp)
        e1.record()
        torch.cuda.synchronize()
        line.append("%d%s: %.1f" % (sp, "*" if sp == default else "", e0.elapsed_time(e1) / 20 * 1e3))
    print("rows %6d N %4d K %4d (tiles %2d)  us by splits  %s" % (rows, N, K, tiles, "  ".join(line)))
