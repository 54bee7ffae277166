#!/bin/bash
# Standard GPU-box sequence: parity tests, bench line, rocprofv3 kernel trace.
# usage (from the repo root on the GPU box): bash tools/gpu_check.sh <tag> [bench args]
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -s KILL 400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout -s KILL 400 python bench.py "$@" > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/prof -o kt -- python $R/bench.py --steps 50 --warmup 10 --only-timed > $O/prof_bench.json 2> $O/prof.err
cd $R
python tools/rocpd_summary.py $O/prof/kt_results.db 10 > $O/kernel_stats.csv 2>> $O/prof.err
rm -rf $O/prof
tail -3 $O/pytest_gpu.log; cat $O/bench.json; head -8 $O/kernel_stats.csv
