#!/bin/bash
# Round-6 starting evidence at the tree as it is: the arrival-stamp path's HBM counters, SQ counters and tick timeline (on and off
# the tick grid), and the headline tick's timeline.   usage: bash tools/r07_baseline.sh <tag>
TAG=${1:-r07_base}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
HASH=$(python -c "from channeld_amd.build import source_hash; print(source_hash())")
echo "{\"source_hash\": \"$HASH\", \"tag\": \"$TAG\"}" > $O/stamp.json
timeout -s KILL 100 python bench.py --steps 100 --warmup 20 --only-timed > $O/bench_timed.json 2> $O/bench_timed.err
timeout -s KILL 100 python bench.py --steps 100 --warmup 20 --only-timed --arrival-jitter > $O/bench_aj.json 2> $O/bench_aj.err
timeout -s KILL 100 python bench.py --steps 100 --warmup 20 --only-timed --arrival-jitter --tick-jitter-us 3000 > $O/bench_aj_off.json 2> $O/bench_aj_off.err
bash tools/timeline.sh $TAG/tl_serial > $O/tl_serial.log 2>&1
bash tools/timeline.sh $TAG/tl_aj --arrival-jitter > $O/tl_aj.log 2>&1
bash tools/timeline.sh $TAG/tl_aj_off --arrival-jitter --tick-jitter-us 3000 > $O/tl_aj_off.log 2>&1
bash tools/pmc_hbm.sh $TAG/pmc_aj --arrival-jitter > $O/pmc_aj.log 2>&1
bash tools/pmc_hbm.sh $TAG/pmc_aj_off --arrival-jitter --tick-jitter-us 3000 > $O/pmc_aj_off.log 2>&1
cut -c1-600 $O/bench_timed.json $O/bench_aj.json $O/bench_aj_off.json
cat $O/tl_aj.log | tail -30
