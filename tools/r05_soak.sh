#!/bin/bash
# round 5: bit-reproducibility of the new kernels over thousands of graph replays + the "bin once" A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_soak; mkdir -p $O
python tools/soak.py 2000 cfg2 bf16 2>&1 | tail -2 | tee $O/soak_bf16.txt
python tools/soak.py 1000 cfg2 fp32 2>&1 | tail -2 | tee $O/soak_fp32.txt
python tools/soak.py 300 cfg5 bf16 2>&1 | tail -2 | tee $O/soak_cfg5.txt
tools/ab.sh 2 "MVG_SORT_PAIRS=layer" "MVG_SORT_PAIRS=first" -- --secondary 0 | tee $O/ab_sort.txt
