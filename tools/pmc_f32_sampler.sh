#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_f32samp; rm -rf "$OUT"; mkdir -p "$OUT"
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
            "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; do
  N=$(echo $PASS | cut -d' ' -f1)
  MVG_OVERLAP_PYRAMID=0 rocprofv3 --pmc $PASS --kernel-trace -d "$OUT/pmc_$N" -o pmc --output-format csv -- python $ROOT/bench.py --dtype fp32 --steps 3 --warmup 1 --cpu-baseline 0 --graph 0 --profile-steps 0 --traffic off > "$OUT/$N.log" 2>&1
done
python $ROOT/tools/summarize_prof.py "$OUT" 2>/dev/null | grep -A24 -E "^msda_gfused" | grep -v "^--"
