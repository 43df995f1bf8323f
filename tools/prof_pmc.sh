#!/bin/bash
# quick PMC pass:  tools/prof_pmc.sh <tag> "<counters>" [bench args]
TAG=$1; CNT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace -d "$OUT/pmc_x" -o pmc --output-format csv -- python $ROOT/bench.py --steps 4 --warmup 2 --cpu-baseline 0 --graph 0 --profile-steps 0 "$@" > "$OUT/log.txt" 2>&1
python $ROOT/tools/summarize_prof.py "$OUT" 2>&1 | grep -A12 -E "^(chain|linear|msda|gather)" 
