#!/bin/bash
# rocprofv3 recipe used for profiles/: kernel trace + stats, then PMC passes (separate runs).
#   tools/prof.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --cpu-baseline 0 --graph 0 --profile-steps 0 --traffic off --secondary 0 $*"
# per-kernel durations and counters are those of each kernel running alone (as bench.py's roofline pass times them):
# the pyramid GEMMs are issued inline here; the overlapped schedule is traced separately at the end
export MVG_OVERLAP_PYRAMID=0
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python $ROOT/bench.py $ARGS > "$OUT/trace.log" 2>&1
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
            "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_EA0_WRREQ_sum" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  N=$(echo $PASS | cut -d' ' -f1)
  rocprofv3 --pmc $PASS --kernel-trace -d "$OUT/pmc_$N" -o pmc --output-format csv -- python $ROOT/bench.py $ARGS > "$OUT/pmc_$N.log" 2>&1
done
python $ROOT/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
MVG_OVERLAP_PYRAMID=1 rocprofv3 --kernel-trace -d "$OUT/overlap" -o trace --output-format csv -- python $ROOT/bench.py $ARGS > "$OUT/overlap.log" 2>&1
python $ROOT/tools/timeline.py "$OUT/overlap" > "$OUT/timeline_overlap.txt" 2>&1
cat "$OUT/summary.txt"
