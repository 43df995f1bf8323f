"""Run-to-run / order determinism of mvg_chain_attn_pose at cfg-2 size (76 800 rows, 2 workgroups per CU).  GPU only."""
import sys; sys.path.insert(0, "/root/repo")
import torch
from mvgformer_amd import ops
torch.manual_seed(0)
dev = "cuda"; bf = torch.bfloat16
rows = 76800
samp = torch.randn(rows, 256, device=dev).to(bf)
inside = (torch.rand(rows, device=dev) < 0.6).to(torch.uint8)
mk = lambda n, k: ops.swizzle_weight((torch.randn(n, k, device=dev) / 16).to(bf))
Wp, W0, W1 = mk(256, 256), mk(256, 256), mk(256, 256)
vec = lambda n: torch.randn(n, device=dev) * 0.1
bp, b0, b1 = vec(256), vec(256), vec(256)
W2, b2 = torch.randn(3, 256, device=dev) / 16, vec(3)
wts = (Wp, bp, W0, b0, W1, b1, W2, b2)
om = ops.chain_masked_row_output(*wts)
perm = torch.argsort(1 - inside.int(), stable=True).to(torch.int32)
def run(order):
    a, o = ops.chain_attn_pose(samp, inside, *wts, order=order, o_masked=om)
    torch.cuda.synchronize()
    return a.clone(), o.clone()
a0, o0 = run(perm)
for i in range(5):
    a1, o1 = run(perm)
    print("same order  run %d: attn identical %s  o identical %s  n_diff_rows %d" % (i, torch.equal(a0, a1), torch.equal(o0, o1), int(((o0 - o1).abs().amax(1) > 0).sum())))
nin = int(inside.sum())
sh = perm[torch.cat([torch.randperm(nin, device=dev), nin + torch.randperm(rows - nin, device=dev)])].contiguous()
a2, o2 = run(sh)
print("shuffled order: attn identical %s  o identical(inside rows) %s  n_diff_rows %d" % (torch.equal(a0, a2), torch.equal(o0[inside != 0], o2[inside != 0]), int(((o0 - o2).abs().amax(1) > 0).sum())))
a3, o3 = run(None)
print("no order:       attn identical %s  n_diff_rows(inside) %d" % (torch.equal(a0, a3), int(((o0 - o3).abs().amax(1)[inside != 0] > 0).sum())))
a1, o1 = run(perm)
d = (o0 - o1).abs()
rows_d = torch.nonzero(d.amax(1) > 0).flatten()
pos = torch.empty(rows, dtype=torch.long, device=dev); pos[perm.long()] = torch.arange(rows, device=dev)
tiles = (pos[rows_d] // 128)
ut, cnt = torch.unique(tiles, return_counts=True)
print("differing rows %d in %d tiles; rows per tile:" % (rows_d.numel(), ut.numel()), cnt.tolist()[:30])
print("tile ids:", ut.tolist()[:30])
comp = (d[rows_d] > 0).float().mean(0)
print("fraction of differing rows per component:", comp.tolist())
r0 = rows_d[:6]
print("pos in tile:", (pos[r0] % 128).tolist())
print(o0[r0].tolist()); print(o1[r0].tolist())
