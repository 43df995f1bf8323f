#!/bin/bash
# round 5, VERDICT item 4c: the sampler's gather-loop variants (gsamp_pipe) re-measured at cfg-5 (31 views) before they are deleted
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_ab7; mkdir -p $O
for i in 1 2; do for v in 0 1 2; do
  ms=$(MVG_TUNE=gsamp_pipe=$v python bench.py --cpu-baseline 0 --profile-steps 0 --traffic off --secondary 0 --config cfg5 --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "cfg5 gsamp_pipe=$v: $ms" | tee -a $O/ab.txt
done; done
for v in 0 1 2; do
  MVG_TUNE=gsamp_pipe=$v python bench.py --cpu-baseline 0 --profile-steps 3 --traffic off --secondary 0 --config cfg5 --steps 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 gsamp_pipe=$v sampler alone', d['roofline']['avg_launch_us'])" | tee -a $O/ab.txt
done
