import torch, sys
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in ("out", "out2"):
    x, y = a[k].view(-1, 8, 32).float(), b[k].view(-1, 8, 32).float()
    d = (x != y)
    rows = d.any(-1)
    print(k, "differing elements", int(d.sum()), "differing (pair, head) units", int(rows.sum()), "of", rows.numel(), "max abs", float((x - y).abs().max()))
    print("  per head:", rows.sum(0).tolist())
    idx = rows.nonzero()[:5]
    for p, m in idx.tolist():
        print("  pair", p, "head", m, "masked?", int(a["msk"][p]) == 0, x[p, m, :4].tolist(), y[p, m, :4].tolist())
