#!/bin/bash
# PMC counters of the sampling kernel's variants (MVG_TUNE): tools/pmc_variants.sh "gsamp_pipe=0" "gsamp_pipe=1" ... -> table
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for V in "$@"; do
  OUT=$ROOT/gpurun_out/pmcv_$(echo $V | tr '=,' '__'); mkdir -p "$OUT"
  for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
              "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
              "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE GRBM_GUI_ACTIVE"; do
    N=$(echo $PASS | cut -d' ' -f1)
    MVG_TUNE=$V MVG_OVERLAP_PYRAMID=0 rocprofv3 --pmc $PASS --kernel-trace -d "$OUT/pmc_$N" -o pmc --output-format csv -- python $ROOT/bench.py --steps 4 --warmup 2 --cpu-baseline 0 --graph 0 --profile-steps 0 --traffic off > "$OUT/$N.log" 2>&1
  done
  echo "== $V"
  python $ROOT/tools/summarize_prof.py "$OUT" 2>/dev/null | grep -A22 -E "^msda_gsamp" | grep -v "^--"
done
