#!/bin/bash
# kernel-trace only (no PMC passes): tools/ktrace.sh <tag> [bench args...] -> per-kernel average durations
set -u
TAG=${1:-k}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline 0 --graph 0 --profile-steps 0 --traffic off $* > "$OUT/trace.log" 2>&1
python - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-60s %5s %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
