#!/bin/bash
# at-origin LDS window, best case: the sampler with its value gathers replaced by ds_read_b128 (results garbage) at 16 / 8 wavefronts per CU
for lib in mvgformer_amd/libmvgformer_hip.so build/libmvg_ldsemul.so; do
  echo "== $lib"
  MVG_LIB=$(realpath $lib) AB_RESIDENCY=0 python tools/ab_gsamp.py gsamp_threads=256 gsamp_threads=512 gsamp_threads=512,gsamp_lds_pad=49152 2>/dev/null | grep -v "^cfg"
done
