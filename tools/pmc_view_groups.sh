#!/bin/bash
# FETCH_SIZE of the sampling launches per forward at cfg-5 with and without the view-group schedule (MVG_VIEW_GROUP), side stream ON
# in both: rocprofv3 --pmc FETCH_SIZE --kernel-trace over a few eager forwards.  -> gpurun_out/pmc_view_groups.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_vg; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for vg in 0 3 6; do
  MVG_VIEW_GROUP=$vg rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/vg$vg" -o pmc --output-format csv -- \
    python $ROOT/bench.py --config cfg5 --graph 0 --steps 2 --warmup 1 --traffic off --cpu-baseline 0 --secondary 0 --profile-steps 0 > /dev/null 2>&1
  python - "$OUT/vg$vg" $vg <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
tot, n = 0.0, 0
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "msda_gsamp_kernel" in r["Kernel_Name"]:
        tot += float(r["Counter_Value"]); n += 1
fw = 6  # forwards profiled: 3 bench warm-ups + 1 warmup + 2 steps
print("MVG_VIEW_GROUP=%s: %d sampler launches, FETCH_SIZE %.1f MB per forward (x2 gfx950 correction: %.1f MB)" % (sys.argv[2], n, tot / 1024 / fw, 2 * tot / 1024 / fw))
PY
done | tee $ROOT/gpurun_out/pmc_view_groups.txt
