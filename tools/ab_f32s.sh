#!/bin/bash
# Build knock-out variants of the f32s kernels (F32S_KO bits, csrc/f32s_dev.h) into build/exp/ and time each with
# tools/check_f32s.py full.  Run the build part here (hipcc cross-compiles), the timing part on the GPU box:
#   tools/ab_f32s.sh build "0 1 2 4"      then      gpurun -- tools/ab_f32s.sh run "0 1 2 4"
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
variants=${1:-"0 1 2 4"}
mkdir -p build/exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize"
if [ "$mode" = stamps ]; then
  /opt/rocm/bin/hipcc $FLAGS -DF32S_STAMPS=1 -c mvgformer_amd/csrc/f32s.hip -o build/exp/f32s_stamps.o
  objs=$(ls mvgformer_amd/csrc/*.o | grep -v f32s.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/exp/libmvg_stamps.so $objs build/exp/f32s_stamps.o
elif [ "$mode" = build ]; then
  for v in $variants; do
    /opt/rocm/bin/hipcc $FLAGS -DF32S_KO=$v $EXTRA -c mvgformer_amd/csrc/f32s.hip -o build/exp/f32s_ko$v.o
    objs=$(ls mvgformer_amd/csrc/*.o | grep -v f32s.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/exp/libmvg_ko$v.so $objs build/exp/f32s_ko$v.o
  done
else
  for v in $variants; do
    echo "== F32S_KO=$v"
    MVG_LIB=$PWD/build/exp/libmvg_ko$v.so python tools/check_f32s.py full 2>&1 | grep " us"
  done
fi
