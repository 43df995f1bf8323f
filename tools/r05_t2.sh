#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_t2; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
tools/ab.sh 2 "MVG_TUNE=gsamp_pipe=0" "MVG_TUNE=gsamp_pipe=1" -- --secondary 0 | tee $O/ab.txt
