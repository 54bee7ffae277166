#!/bin/bash
# kernel trace of one bench configuration: tools/trace_one.sh <tag> <name> <skip> <bench args...>  (GPU box, repo root)
TAG=$1; NAME=$2; SKIP=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/prof_$NAME -o kt -- python $R/bench.py "$@" > $O/prof_bench_$NAME.json 2> $O/prof_$NAME.err
cd $R
python tools/rocpd_summary.py $O/prof_$NAME/kt_results.db $SKIP > $O/kernel_stats_$NAME.csv 2>> $O/prof_$NAME.err
rm -rf $O/prof_$NAME
head -14 $O/kernel_stats_$NAME.csv
