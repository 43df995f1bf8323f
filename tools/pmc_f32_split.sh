#!/bin/bash
# PMC counters of the fp32 GEMM forms (exact / split) at one shape: tools/pmc_f32_split.sh [M N K]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
M=${1:-201600}; N=${2:-256}; K=${3:-256}
cd /tmp; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_f32split; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python $ROOT/tools/bench_f32_split.py $M $N $K > "$OUT/trace.log" 2>&1
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_ANY"; do
  P=$(echo $PASS | cut -d' ' -f1)
  rocprofv3 --pmc $PASS --kernel-trace -d "$OUT/pmc_$P" -o pmc --output-format csv -- python $ROOT/tools/bench_f32_split.py $M $N $K > "$OUT/$P.log" 2>&1
done
python $ROOT/tools/summarize_prof.py "$OUT" 2>/dev/null | grep -A28 -E "^linear_kernel" | grep -v "^--"
