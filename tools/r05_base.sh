#!/bin/bash
# round 5 re-entry: state of HEAD on a fresh box -- GPU tests, the default bench line, timeline of the graph replay
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_base; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_base/bench.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","ms_per_step_mean","ms_per_step_min_max")})
print(d["roofline"])
print(json.dumps(d.get("kernels"),indent=0)[:3000])
for k,v in d["secondary"].items(): print(k, {a:v.get(a) for a in ("ms_per_step","value","frac","sampler_us","error")})
print(d.get("rank_time_split"))
PY
tools/ktrace_graph.sh r05base --secondary 0 > /dev/null 2>&1; cp gpurun_out/ktg_r05base/timeline.txt $O/timeline.txt; cat $O/timeline.txt
