"""BASELINE.md section 3.1 (build container only -- imports /root/reference through tests/golden/ref_harness.py): wall time
of the REFERENCE's DQDecoder.forward (deform_core_pytorch in place of the CUDA op) and of the oracle restatement
(oracle/decoder_ref.py) on the same seeded synthetic inputs with the same torch thread count, and their ratio.  The GPU box
can only time the oracle (the reference never travels); this ratio is what turns bench.py's cpu_baseline ("port") into an
estimate of the reference's own CPU speed.   python tests/golden/time_reference_vs_oracle.py [cfg1|cfg2-1layer ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from mvgformer_amd.synthetic import build_case, to_torch_state  # noqa: E402
from oracle import decoder_ref as O  # noqa: E402
from tests.golden.ref_harness import build_reference_decoder  # noqa: E402


def timed(fn, reps):
    fn()                                    # warm-up (first call pays allocator / thread-pool start-up)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    nthreads = os.cpu_count() or 1
    torch.set_num_threads(nthreads)
    specs = {"cfg1": dict(name="cfg1", kw={}, reps=6),
             "cfg2-1layer": dict(name="cfg2", kw=dict(layers=1), reps=1)}
    out = {"threads": nthreads, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
           "torch": torch.__version__, "cases": {}}
    for key in (sys.argv[1:] or ["cfg1", "cfg2-1layer"]):
        sp = specs[key]
        case = build_case(sp["name"], seed=0, **sp["kw"])
        dec = build_reference_decoder(case)
        prm = to_torch_state(case.weights)

        def run_ref():
            with torch.no_grad():
                return dec(case.tgt, case.reference_points, case.src_views, case.meta, case.spatial_shapes,
                           case.level_start_index, None, query_pos=case.query_pos,
                           src_padding_mask=[torch.zeros(1, 1, dtype=torch.bool)], threshold=0.1)

        def run_oracle():
            with torch.no_grad():
                return O.decoder_forward(prm, case.layers, case.tgt, case.reference_points, case.src_views, case.meta,
                                         case.spatial_shapes, case.level_start_index, case.query_pos, case.img_size,
                                         threshold=0.1)

        a, b = run_ref(), run_oracle()
        err = float((a[0] - b[0]).abs().max())
        t_ref, t_or = timed(run_ref, sp["reps"]), timed(run_oracle, sp["reps"])
        out["cases"][key] = {"layers": case.layers, "views": case.V, "queries": case.NQ, "reference_s": round(t_ref, 4),
                             "oracle_s": round(t_or, 4), "reference_over_oracle": round(t_ref / t_or, 3),
                             "max_abs_feature_difference": err}
        print(key, out["cases"][key], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
