#!/bin/bash
# One GPU-box call for a round's evidence: bench line, rocprofv3 kernel trace of the timed region, the two HBM PMC passes,
# then the GPU parity suite.  usage (repo root on the GPU box): bash tools/round_check.sh <tag> [pytest args]
TAG=${1:-round}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -s KILL 240 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/prof -o kt -- python $R/bench.py --steps 50 --warmup 10 --only-timed > $O/prof_bench.json 2> $O/prof.err
cd $R
python tools/rocpd_summary.py $O/prof/kt_results.db 10 > $O/kernel_stats.csv 2>> $O/prof.err
rm -rf $O/prof
bash tools/pmc_hbm.sh $TAG > $O/pmc.log 2>&1
timeout -s KILL ${PYTEST_LIMIT:-330} python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 "$@" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log; cat $O/bench.json | cut -c1-1500; head -12 $O/kernel_stats.csv; cat $O/pmc.log
