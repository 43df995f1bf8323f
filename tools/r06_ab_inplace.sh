#!/bin/bash
# same-box A/B of the pyramid hand-over: reference NCHW fp32 (packed per step) against the packed layout (no pack)
for i in 1 2 3; do for p in nchw inplace; do
  ms=$(python bench.py --cpu-baseline 0 --profile-steps 0 --traffic off --secondary 0 --steps 100 --producer $p 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$p: $ms"
done; done
