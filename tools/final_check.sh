#!/bin/bash
# GPU parity suite, then the partial-update A/B and the headline line.  usage: bash tools/final_check.sh <tag>
TAG=${1:-final}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -s KILL 330 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -14 $O/pytest_gpu.log
bash tools/ab_partial.sh $TAG/abp 2>&1 | tail -8
timeout -s KILL 200 python bench.py --only-timed > $O/bench_only_timed.json 2> $O/bench.err; cut -c1-400 $O/bench_only_timed.json
