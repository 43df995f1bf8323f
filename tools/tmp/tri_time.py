import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mvgformer_amd import ops
from mvgformer_amd.synthetic import CONFIGS, make_meta, ring_cameras
DEV = "cuda"
c = dict(CONFIGS["cfg2"])
V, NQ, J = 5, 1024, 15
cams_np = ring_cameras(V, c["orig_wh"], c["focal"], c["radius"], c["space_center"], c["k"], c["p"], seed=3)
meta = make_meta(cams_np, 1, c["orig_wh"], c["img_wh"])
cams = ops.pack_cameras(meta, c["img_wh"], DEV)
rs = np.random.RandomState(9)
X = torch.from_numpy((np.asarray(c["space_center"]) + rs.uniform(-800, 800, (1, NQ * J, 3))).astype(np.float32))
r, _, _ = ops.project(X.to(DEV), cams, ops.Levels([[8, 8]], [0]), V, 1)
valid = torch.ones((1, NQ), dtype=torch.uint8, device=DEV)
anyv = torch.ones((1,), dtype=torch.int32, device=DEV)
for name, amp in (("exact rays", 0.0), ("offsets +-2 px", 2.0), ("offsets +-30 px", 30.0), ("offsets +-300 px", 300.0)):
    o = torch.zeros((V, NQ * J, 3))
    o[..., :2] = torch.from_numpy(rs.uniform(-amp, amp, (V, NQ * J, 2)).astype(np.float32))
    o = o.to(DEV)
    for _ in range(5):
        ops.triangulate(r, o, cams, valid, anyv, V, 1, NQ, J)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        ops.triangulate(r, o, cams, valid, anyv, V, 1, NQ, J)
    b.record(); torch.cuda.synchronize()
    print("%-18s %6.1f us per launch" % (name, a.elapsed_time(b) / 50 * 1e3))
import hashlib
o = torch.zeros((V, NQ * J, 3), device=DEV)
Xr = ops.triangulate(r, o, cams, valid, anyv, V, 1, NQ, J)[0]
print("exact rays: sha", hashlib.sha256(Xr.cpu().numpy().tobytes()).hexdigest()[:12], "max err mm", float((Xr.cpu().view(-1,3) - X[0]).norm(dim=-1).max()))
