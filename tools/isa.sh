#!/bin/bash
# tools/isa.sh <csrc file> <kernel name substring> [sed range]  -- gfx950 ISA of one kernel, memory / MFMA / control instructions only
F=$1; K=$2; R=${3:-1,400}
S=/tmp/isa_$(basename $F .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -S --cuda-device-only /root/repo/mvgformer_amd/csrc/$F -o $S 2>/dev/null
grep -n "\.vgpr_count\|\.name:\|spill_count\|lds_size\|private_segment_fixed" $S | grep -A5 "name:.*$K" | grep -v "^--" | head -12
L=$(grep -n "^_Z.*$K.*:" $S | head -1 | cut -d: -f1)
E=$(awk -v s=$L 'NR>s && /s_endpgm/ {print NR; exit}' $S)
awk -v s=$L -v e=$E 'NR>=s && NR<=e' $S | grep -n "v_mfma\|ds_read\|s_waitcnt\|s_barrier\|ds_write\|global_load\|global_store\|s_cbranch\|^.LBB\|s_endpgm\|buffer_" | awk '{$1=$1};1' | sed -n "${R}p"
