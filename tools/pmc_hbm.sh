#!/bin/bash
# HBM traffic of the tick's kernels: FETCH_SIZE and WRITE_SIZE in two separate rocprofv3 --pmc passes
# (counters only, no tracing), summarised per kernel by tools/pmc_summary.py.
# usage (repo root on the GPU box): bash tools/pmc_hbm.sh <tag>   -> gpurun_out/<tag>/pmc_hbm_traffic.json
TAG=${1:-pmc}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --steps 20 --warmup 5 --only-timed "$@" > $O/pmc_$c.out 2> $O/pmc_$c.err
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && [ "$f" != "$O/pmc_$c/p_counter_collection.csv" ] && cp $f $O/pmc_$c/p_counter_collection.csv
done
cd $R
python tools/pmc_summary.py $O 5 > $O/pmc_hbm_traffic.json 2> $O/pmc_summary.err
python tools/pmc_summary.py $O 5 "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_hbm.sh $TAG) of: python bench.py --steps 20 --warmup 5 --only-timed; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as reported" > $O/hbm_traffic.json 2>> $O/pmc_summary.err
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
grep -A6 '"k_fanout_emit_seg"' $O/pmc_hbm_traffic.json
