#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_ab13; mkdir -p $O
J="MVG_PYRAMID_JIT=1"; S="MVG_PYRAMID_JIT_SLOTS"
for args in "--config cfg5 --steps 20" "--batch 2 --steps 50" "--batch 4 --steps 40" "--inside all --steps 50" "--valid-fraction 0.1 --steps 50" "--queries 128 --steps 100"; do
 for i in 1 2; do for e in "MVG_PYRAMID_JIT=0" "$J $S=32" "$J $S=48" "$J $S=64"; do
  ms=$(env $e python bench.py --cpu-baseline 0 --profile-steps 0 --traffic off --secondary 0 $args 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$args | $e: $ms" | tee -a $O/ab.txt
 done; done
done
