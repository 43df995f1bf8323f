#!/bin/bash
# tools/isa_hist.sh <csrc file> <mangled-name regex> -- per basic-block-range VALU / memory instruction counts and the hot loop's histogram
F=$1; K=$2
S=/tmp/isa_$(basename $F .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -S --cuda-device-only /root/repo/mvgformer_amd/csrc/$F -o $S 2>/dev/null || exit 1
L=$(grep -n "^$K.*:" $S | head -1 | cut -d: -f1)
E=$(awk -v s=$L 'NR>s && /^\.Lfunc_end/ {print NR; exit}' $S)
awk -v s=$L -v e=$E 'NR>=s && NR<=e' $S > /tmp/isa_kernel.s
echo "kernel lines $L-$E; static VALU $(grep -cE '^\s+v_' /tmp/isa_kernel.s) VMEM $(grep -cE '^\s+(global|buffer)_' /tmp/isa_kernel.s) DS $(grep -cE '^\s+ds_' /tmp/isa_kernel.s)"
grep -A12 "^\s*\.name:\s*$K" $S | grep "vgpr_count\|sgpr_count\|spill\|lds_size" 
S0=$(grep -n "Inner Loop Header" /tmp/isa_kernel.s | head -1 | cut -d: -f1)
if [ -n "$S0" ]; then
  awk -v s=$S0 'NR>=s' /tmp/isa_kernel.s | awk '/s_cbranch_scc1|s_cbranch_vccnz|s_cbranch_execnz/ {print; exit} {print}' > /tmp/isa_loop.s
  echo "first inner loop: VALU $(grep -cE '^\s+v_' /tmp/isa_loop.s)"
  grep -E "^\s+[a-z]" /tmp/isa_loop.s | awk '{print $1}' | sort | uniq -c | sort -rn | head -${3:-45}
fi
