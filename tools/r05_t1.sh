#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_t1; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
