#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_ab9; mkdir -p $O
for i in 1 2; do for v in 0 1; do
  for args in "--batch 2 --steps 50" "--batch 4 --steps 50" "--inside all --steps 50"; do
  ms=$(MVG_TUNE=gsamp_pipe=$v python bench.py --cpu-baseline 0 --profile-steps 0 --traffic off --secondary 0 $args 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$args gsamp_pipe=$v: $ms" | tee -a $O/ab.txt
  done
done; done
