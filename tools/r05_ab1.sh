#!/bin/bash
# round 5, first contact: grouped pyramid products + pack on the side stream
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_ab1; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -k "grouped_pyramid or side_stream or bit_reproducible or pyramid_gemms" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/bench_pyrgroup.py > $O/pyrgroup.txt 2>&1; cat $O/pyrgroup.txt
tools/ab.sh 2 "MVG_PYRAMID_GROUP=0 MVG_PACK_ON_SIDE=0" "MVG_PYRAMID_GROUP=0 MVG_PACK_ON_SIDE=1" "MVG_PYRAMID_GROUP=3 MVG_PACK_ON_SIDE=0" "MVG_PYRAMID_GROUP=3 MVG_PACK_ON_SIDE=1" "MVG_PYRAMID_GROUP=1 MVG_PACK_ON_SIDE=1" -- --secondary 0 > $O/ab.txt 2>&1; cat $O/ab.txt
tools/ktrace_graph.sh r05a --secondary 0 > /dev/null 2>&1; cp gpurun_out/ktg_r05a/timeline.txt $O/timeline_group3.txt; cat $O/timeline_group3.txt
