"""Time of one training step of the decoder at cfg-2 size: forward under autograd (torch geometry +
ProjAttn with the HIP sampling forward / backward kernels) + backward to every parameter.  GPU only.
python tools/train_step_probe.py [config] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvgformer_amd.factory import build_decoder_for_case, case_to_device  # noqa: E402
from mvgformer_amd.synthetic import build_case  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
case = build_case(cfg, seed=0)
dec = build_decoder_for_case(case, "cuda", torch.float32)
g = case_to_device(case, "cuda")
for p in dec.parameters():
    p.requires_grad_(True)
dec.train()


def step():
    for p in dec.parameters():
        p.grad = None
    out = dec(g.tgt, g.reference_points, g.src_views, g.meta, g.spatial_shapes, g.level_start_index, None,
              query_pos=g.query_pos, threshold=0.1)
    loss = out[0].float().pow(2).mean() + 1e-6 * out[1].float().pow(2).mean() + sum(c.float().sum() for c in out[4]) * 1e-3
    loss.backward()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# one multi-tensor launch instead of five per parameter: a profile of this script counts the steps' launches, not this check's
grads = [p.grad for p in dec.parameters() if p.grad is not None]
n_grad = int(torch.isfinite(torch.stack(torch._foreach_norm(grads))).sum())
print("%s training step (fp32, forward + backward): %.1f ms; loss %.4f; %d / %d parameters with finite gradients; peak memory %.1f GB"
      % (cfg, dt * 1e3, float(loss), n_grad, sum(1 for _ in dec.parameters()), torch.cuda.max_memory_allocated() / 2 ** 30))
