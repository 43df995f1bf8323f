#!/bin/bash
# kernel trace of the graph-replayed forward (what bench.py times): tools/ktrace_graph.sh <tag> [bench args...] -> timeline of the last forward
set -u
TAG=${1:-g}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ktg_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/trace" -o trace --output-format csv -- python $ROOT/bench.py --steps 6 --warmup 3 --cpu-baseline 0 --traffic off --profile-steps 0 --secondary 0 $* > "$OUT/trace.log" 2>&1
python $ROOT/tools/timeline.py "$OUT/trace" > "$OUT/timeline.txt"
cat "$OUT/timeline.txt"
