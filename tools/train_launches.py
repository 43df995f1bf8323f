"""Which torch ops the launches of one training step come from: torch.profiler over ONE step of tools/train_step_probe.py's loop,
device kernels counted per top-level op (forward) and per autograd node (backward).  GPU only.
python tools/train_launches.py [config]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from mvgformer_amd.factory import build_decoder_for_case, case_to_device  # noqa: E402
from mvgformer_amd.synthetic import build_case  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
kernels = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
print("device activities in one step: %d" % len(kernels))
# top-level CPU ops (no CPU parent) and the kernels launched beneath them
count = collections.Counter()
time_us = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or e.cpu_parent is not None:
        continue
    def walk(n):
        k = list(n.kernels)
        for c in n.cpu_children:
            k += walk(c)
        return k
    ks = walk(e)
    if ks:
        count[e.name] += len(ks)
        time_us[e.name] += sum(k.duration for k in ks)
print("%-70s %7s %10s" % ("top-level op", "kernels", "device_us"))
for name, n in count.most_common(60):
    print("%-70s %7d %10.0f" % (name[:70], n, time_us[name]))
