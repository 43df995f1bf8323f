"""Backward of the sampling op at the size of one cfg-2 view-layer: deterministic (csrc/msda_bwd.hip) vs atomic form.
python tools/bench_bwd.py [det|atomic] [reps]   (run under `rocprofv3 --kernel-trace --stats` for the per-kernel split)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mvgformer_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "det"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
shapes = torch.tensor([(128, 240), (64, 120), (32, 60)], dtype=torch.long)
starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
N, M, D, Lq, P, L = 1, 8, 32, 15360, 8, 3
S = int((shapes[:, 0] * shapes[:, 1]).sum())
g = torch.Generator().manual_seed(21)
value = torch.randn((N, S, M, D), generator=g).to(dev)
centre = torch.rand((N, Lq, 1, 1, 1, 2), generator=g) * 1.1 - 0.05
loc = (centre + torch.randn((N, Lq, M, L, P, 2), generator=g) * 0.03).contiguous().to(dev)
wgt = torch.softmax(torch.randn((N, Lq, M, L * P), generator=g), -1).view(N, Lq, M, L, P).contiguous().to(dev)
go = torch.randn((N, Lq, M * D), generator=g).to(dev)
shapes, starts = shapes.to(dev), starts.to(dev)
ops.BACKWARD_MODE = mode
for _ in range(2):
    ops.msda_backward(value, shapes, starts, loc, wgt, go)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    ops.msda_backward(value, shapes, starts, loc, wgt, go)
b.record()
torch.cuda.synchronize()
print("%s backward, one cfg-2 view-layer: %.0f us" % (mode, a.elapsed_time(b) / reps * 1e3))
