#!/bin/bash
# round 5: the committed records of the final build -- bench line, kernel stats + PMC summary, graph-replay timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_prof; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err
tools/prof.sh r06 > $O/prof.log 2>&1; tail -5 $O/prof.log
cp gpurun_out/prof_r06/summary.txt $O/rocprofv3_summary.txt; cp gpurun_out/prof_r06/timeline_overlap.txt $O/ 2>/dev/null
cp $(ls gpurun_out/prof_r06/trace/*/*kernel_stats.csv gpurun_out/prof_r06/trace/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
tools/ktrace_graph.sh r06 --secondary 0 > /dev/null 2>&1; cp gpurun_out/ktg_r06/timeline.txt $O/timeline_graph.txt
python bench.py --dtype fp32 --secondary 0 --cpu-baseline 0 > $O/bench_fp32.json 2>> $O/bench.err
# keep the merge-back under its 64-MB limit: the raw traces stay on the box
rm -rf gpurun_out/prof_r06/trace gpurun_out/prof_r06/pmc_[A-Z]* gpurun_out/prof_r06/overlap gpurun_out/ktg_r06/trace
ls -la $O
