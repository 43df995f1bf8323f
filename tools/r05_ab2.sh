#!/bin/bash
# round 5: schedule of the grouped pyramid products (gating behind the samplers, group sizes, slots)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_ab2; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_dist_gpu.py -x -q -k "grouped_pyramid or side_stream or bit_reproducible or dist" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
P="MVG_PYRAMID_GROUP"; G="MVG_PYRAMID_GATE"
tools/ab.sh 2 "$P=3 $G=0 MVG_QUERY_TERM_FIRST=0" "$P=3 $G=0" "$P=3 $G=1" "$P=1 $G=1" "$P=1 $G=1 MVG_PYRAMID_GATE_SLOTS=32" "$P=1 $G=1 MVG_PYRAMID_GATE_SLOTS=48" "$P=1 $G=1 MVG_PYRAMID_GATE_FIRST=1" "$P=3 $G=1 MVG_PYRAMID_GATE_FIRST=1" "$P=2 $G=1" -- --secondary 0 > $O/ab.txt 2>&1; cat $O/ab.txt
for v in "3 1" "1 1"; do set -- $v
  MVG_PYRAMID_GROUP=$1 MVG_PYRAMID_GATE=$2 tools/ktrace_graph.sh r05b_$1_$2 --secondary 0 > /dev/null 2>&1; cp gpurun_out/ktg_r05b_$1_$2/timeline.txt $O/timeline_group$1_gate$2.txt; cat $O/timeline_group$1_gate$2.txt
done
