#!/bin/bash
# level-2 ablation probe: full kernel vs a build that skips the last level's gathers, at 16 / 8 wavefronts per CU
for lib in mvgformer_amd/libmvgformer_hip.so build/libmvg_abl.so; do
  for pad in 0 35840; do
    echo "== $lib gsamp_lds_pad=$pad"
    MVG_LIB=$(realpath $lib) AB_RESIDENCY=1 python tools/ab_gsamp.py gsamp_lds_pad=$pad 2>/dev/null | grep -v "^cfg"
  done
done
