#!/bin/bash
# One GPU-box call for a round's evidence: bench line, rocprofv3 kernel trace of the timed region (serial headline; with
# PIPE_TRACE=1 also of the pipelined schedule, with WIRE_TRACE=1 of the wire builder), the two HBM PMC passes (PMC=1), then the
# GPU parity suite.  usage (repo root on the GPU box): [PMC=1] [WIRE_TRACE=1] [PIPE_TRACE=1] bash tools/round_check.sh <tag> [pytest args]
TAG=${1:-round}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -s KILL 240 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
trace() {  # trace <name> <skip> <bench args...>
  local name=$1 skip=$2; shift 2
  cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o kt -- python $R/bench.py "$@" > $O/prof_bench_$name.json 2> $O/prof_$name.err
  cd $R
  python tools/rocpd_summary.py $O/prof_$name/kt_results.db $skip > $O/kernel_stats_$name.csv 2>> $O/prof_$name.err
  rm -rf $O/prof_$name
}
trace serial 10 --steps 50 --warmup 10 --only-timed
[ -n "$PIPE_TRACE" ] && trace pipelined 10 --steps 50 --warmup 10 --only-timed --headline pipelined
[ -n "$WIRE_TRACE" ] && trace wire 0 --steps 8 --warmup 6 --only-timed --wire 3
[ -n "$PMC" ] && bash tools/pmc_hbm.sh $TAG > $O/pmc.log 2>&1
timeout -s KILL ${PYTEST_LIMIT:-330} python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 "$@" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -16 $O/pytest_gpu.log; cat $O/bench.json | cut -c1-700; head -12 $O/kernel_stats_serial.csv; [ -f $O/kernel_stats_wire.csv ] && head -8 $O/kernel_stats_wire.csv; [ -f $O/pmc.log ] && cat $O/pmc.log
