#!/bin/bash
# PMC counters of the sampling kernel for two builds of the library (MVG_LIB): tools/r06_pmc_lib.sh libA.so libB.so -> table per build
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for LIB in "$@"; do
  TAG=$(basename $LIB .so); OUT=$ROOT/gpurun_out/pmcl_$TAG; mkdir -p "$OUT"
  for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
              "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
    N=$(echo $PASS | cut -d' ' -f1)
    MVG_LIB=$ROOT/$LIB MVG_OVERLAP_PYRAMID=0 rocprofv3 --pmc $PASS --kernel-trace -d "$OUT/pmc_$N" -o pmc --output-format csv -- python $ROOT/bench.py --steps 4 --warmup 2 --cpu-baseline 0 --graph 0 --profile-steps 0 --traffic off --secondary 0 > "$OUT/$N.log" 2>&1
  done
  echo "== $LIB"
  python $ROOT/tools/summarize_prof.py "$OUT" 2>/dev/null | grep -A16 -E "^msda_gsamp" | grep -v "^--"
done
