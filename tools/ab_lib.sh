#!/bin/bash
# In-box A/B of two builds of the library: tools/ab_lib.sh <repeats> <libA.so> <libB.so> [-- bench args]
#   per build: the sampler alone on cfg-2's layer-0 inputs (tools/ab_gsamp.py, output hash) and the bench headline (ms per forward)
R=$1; A=$(realpath $2); B=$(realpath $3); shift 3; [ "${1:-}" == "--" ] && shift
for lib in $A $B; do
  echo "== $lib"
  MVG_LIB=$lib AB_RESIDENCY=0 AB_HASH=1 python tools/ab_gsamp.py default 2>/dev/null | grep -v "^cfg"
done
for i in $(seq $R); do for lib in $A $B; do
  ms=$(MVG_LIB=$lib python bench.py --cpu-baseline 0 --profile-steps 0 --traffic off --secondary 0 --steps 100 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "$(basename $lib): $ms"
done; done
