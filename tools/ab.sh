#!/bin/bash
# in-box A/B of environment settings: tools/ab.sh <repeats> "ENV=a" "ENV=b" ... [-- bench args]  -> ms per step
R=$1; shift
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; [ "${1:-}" == "--" ] && shift
for i in $(seq $R); do for e in "${ENVS[@]}"; do
  ms=$(env $e python bench.py --cpu-baseline 0 --profile-steps 0 --traffic off --steps 100 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$e: $ms"
done; done
