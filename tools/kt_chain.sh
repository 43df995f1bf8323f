#!/bin/bash
# rocprofv3 kernel-trace durations of tools/bench_chain.py's chain-B variants (printed numbers there are host-bound)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/kt_chain
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/kt_chain -o trace --output-format csv -- python $ROOT/tools/bench_chain.py > /dev/null 2>&1
python - "$ROOT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/gpurun_out/kt_chain/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "chain_b" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
names = ["full", "no xw tail", "no FFN, no tail", "1 view, no FFN", "ring 4", "ring 8", "row-block split"]
for i in range(0, len(d), 23):
    seg = d[i:i + 23]
    print("%-18s avg %.1f us  min %.1f" % (names[i // 23] if i // 23 < len(names) else "?", sum(seg[3:]) / max(1, len(seg[3:])), min(seg)))
PY
