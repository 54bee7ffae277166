#!/bin/bash
# smoke + the default bench line + rocprofv3 kernel trace of the same command's timed region.  usage: bash tools/final_profile.sh <tag>
TAG=${1:-fin}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
timeout -s KILL 240 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/prof -o kt -- python $R/bench.py --steps 50 --warmup 10 --only-timed > $O/prof_bench.json 2> $O/prof.err
cd $R
python tools/rocpd_summary.py $O/prof/kt_results.db 10 > $O/kernel_stats.csv 2>> $O/prof.err
rm -rf $O/prof
cut -c1-500 $O/bench.json; head -12 $O/kernel_stats.csv
