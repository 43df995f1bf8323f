"""mvg_msda_forward (the drop-in for Deformable.deform_forward) at the decoder's shapes: the two work decompositions (tuning knob fwd_map)
timed on decoder-like sampling locations (joints of a person close together, 8 rays per head) -- outputs must be identical.
python tools/bench_fwd_op.py [fp32|bf16]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mvgformer_amd import _lib  # noqa: E402
from mvgformer_amd import deformable as DF  # noqa: E402

dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float32
lib = _lib.load()
torch.manual_seed(0)
N, NQ, J, M, D, L, P = 5, 1024, 15, 8, 32, 3, 8
shapes = torch.tensor([[128, 240], [64, 120], [32, 60]], dtype=torch.long, device="cuda")
starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
S = int((shapes[:, 0] * shapes[:, 1]).sum())
Lq = NQ * J
value = torch.randn(N, S, M, D, device="cuda").to(dt)
person = torch.rand(N, NQ, 1, 2, device="cuda") * 0.8 + 0.1
ref = (person + 0.03 * torch.randn(N, NQ, J, 2, device="cuda")).view(N, Lq, 1, 1, 1, 2)
ang = torch.arange(M, device="cuda") * (2 * math.pi / M)
ray = torch.stack([ang.cos(), ang.sin()], -1).view(1, 1, M, 1, 1, 2) * torch.arange(1, P + 1, device="cuda").view(1, 1, 1, 1, P, 1)
wh = shapes.flip(-1).float().view(1, 1, 1, L, 1, 2)
loc = (ref + ray / wh + 0.002 * torch.randn(N, Lq, M, L, P, 2, device="cuda")).contiguous()
attn = torch.softmax(torch.randn(N, Lq, M, L * P, device="cuda"), -1).view(N, Lq, M, L, P).contiguous()
res = {}
for mp in (0, 2, 1):
    assert lib.mvg_set_tuning(b"fwd_map", mp) == 0
    for _ in range(3):
        out = DF.deform_forward(value, shapes, starts, loc, attn, 64)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = DF.deform_forward(value, shapes, starts, loc, attn, 64)
    e1.record()
    torch.cuda.synchronize()
    res[mp] = out.clone()
    print("fwd_map=%d  %s  %.1f us per call (N=%d images, %d queries, %d heads x %d levels x %d points)" % (mp, str(dt)[6:], e0.elapsed_time(e1) / 20 * 1e3, N, Lq, M, L, P))
print("identical:", torch.equal(res[0], res[1]) and torch.equal(res[0], res[2]))
