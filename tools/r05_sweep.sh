#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_sweep; mkdir -p $O
tools/ab.sh 2 "MVG_TUNE=gsamp_map=4" "MVG_TUNE=gsamp_map=2" "MVG_TUNE=gsamp_map=1" "MVG_TUNE=gsamp_map=8" "MVG_TUNE=gsamp_threads=512" "MVG_TUNE=gsamp_threads=128" "MVG_TUNE=wreg_grid=448" "MVG_TUNE=chain_rm=64" "MVG_PYRAMID_GROUP=2" "MVG_PYRAMID_GROUP=4" -- --secondary 0 | tee $O/ab.txt
