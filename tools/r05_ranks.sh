#!/bin/bash
# round 5, VERDICT item 7: what ONE rank of the 8-GPU query-sharded job costs, measured on one GPU -- its contiguous block of the
# person grid (bench.py --emulate-rank) against 128 persons spread over the whole space (--queries 128), + touched pyramid share
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_ranks; mkdir -p $O
for r in 0 3 7; do
  python bench.py --emulate-rank $r --of 8 --cpu-baseline 0 --traffic off --steps 100 > $O/bench_rank${r}of8.json 2> $O/err_$r.txt; tail -2 $O/err_$r.txt
done
python bench.py --queries 128 --cpu-baseline 0 --traffic off --secondary 0 --steps 100 > $O/bench_q128.json 2> $O/err_q128.txt
for r in 0 1; do python bench.py --emulate-rank $r --of 2 --cpu-baseline 0 --traffic off --steps 100 > $O/bench_rank${r}of2.json 2>> $O/err_2.txt; done
for r in 0 3; do python bench.py --emulate-rank $r --of 4 --cpu-baseline 0 --traffic off --steps 100 > $O/bench_rank${r}of4.json 2>> $O/err_4.txt; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_ranks/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["ms_per_step"], d["rank_time_split"], d["config"].get("emulated_rank"), d["config"].get("touched_pyramid_share", {}).get("per_layer") if d["config"].get("touched_pyramid_share") else None, d["roofline"]["in_image_pair_fraction"])
PY
python bench.py --cpu-baseline 0 --traffic off --secondary 0 --steps 100 > $O/bench_full.json 2> $O/err_full.txt
python -c "
import json; d=json.loads(open('gpurun_out/r05_ranks/bench_full.json').read().strip().splitlines()[-1]); print('full', d['ms_per_step'])"
