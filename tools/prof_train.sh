#!/bin/bash
# rocprofv3 kernel stats of the decoder's training step: tools/prof_train.sh -> gpurun_out/prof_train/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_train
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python $ROOT/tools/train_step_probe.py cfg2 5 2>/dev/null | tail -1 > "$OUT/summary.txt"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python $ROOT/tools/train_step_probe.py cfg2 2 > /dev/null 2>&1
python - "$OUT" >> "$OUT/summary.txt" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("rocprofv3 --kernel-trace --stats, 4 training steps (2 warm-up + 2): %.1f ms of kernels, %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
print("%-72s %7s %10s %7s" % ("kernel", "calls", "avg_us", "pct"))
for r in rows[:16]:
    print("%-72s %7s %10.1f %7.1f" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
cat "$OUT/summary.txt"
