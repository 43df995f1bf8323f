#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_ab3; mkdir -p $O
P="MVG_PYRAMID_GROUP"; G="MVG_PYRAMID_GATE=0"
tools/ab.sh 3 "$P=3 $G MVG_QUERY_TERM_FIRST=0" "$P=3 $G" "$P=3 $G MVG_PYRAMID_REST_SLOTS=32" "$P=3 $G MVG_PYRAMID_REST_SLOTS=48" "$P=0 $G MVG_PACK_ON_SIDE=0 MVG_QUERY_TERM_FIRST=0" -- --secondary 0 > $O/ab.txt 2>&1; cat $O/ab.txt
MVG_PYRAMID_GATE=0 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_ab3/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","ms_per_step_mean","ms_per_step_min_max")})
print(json.dumps(d["roofline_mfma"],indent=1)[:3000])
for k,v in d["secondary"].items(): print(k, {a:v.get(a) for a in ("ms_per_step","ms_per_step_median","value","frac","sampler_us","error","valid_query_share_last_layer","device_activities_per_step","parameters_with_finite_gradients")})
print(d["rank_time_split"])
PY
