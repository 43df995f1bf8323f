#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_ab12; mkdir -p $O
J="MVG_PYRAMID_JIT=1"; S="MVG_PYRAMID_JIT_SLOTS"
tools/ab.sh 2 "MVG_PYRAMID_JIT=0" "$J $S=36" "$J $S=32" "$J $S=28" "$J $S=24" "$J $S=16" -- --secondary 0 | tee $O/ab.txt
MVG_PYRAMID_JIT=1 MVG_PYRAMID_JIT_SLOTS=32 tools/ktrace_graph.sh r05jit --secondary 0 > /dev/null 2>&1; cp gpurun_out/ktg_r05jit/timeline.txt $O/timeline_jit32.txt; rm -rf gpurun_out/ktg_r05jit/trace; sed -n 3,24p $O/timeline_jit32.txt
