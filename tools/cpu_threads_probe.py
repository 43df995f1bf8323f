import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mvgformer_amd.synthetic import build_case, to_torch_state
from oracle import decoder_ref as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
case = build_case("cfg2", seed=0, layers=1)
prm = to_torch_state(case.weights)
for nt in (16, 32, 64):
    torch.set_num_threads(nt)
    t0 = time.time()
    with torch.no_grad():
        O.decoder_layer_forward(prm, "layers.0.", case.tgt, case.query_pos, case.reference_points, case.src_views, case.spatial_shapes, case.level_start_index, case.meta, case.img_size, threshold=0.1)
    print(nt, "threads: %.1f s" % (time.time() - t0))
