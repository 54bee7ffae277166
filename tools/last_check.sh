#!/bin/bash
# The three measurements bench.py's line rests on, at the tree as it is: the HBM PMC passes, the default bench line (which then quotes
# them: same source hash), the rocprofv3 kernel trace of the timed region.  ~40 s of box time.
# usage: bash tools/last_check.sh <tag>
TAG=${1:-last}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
HASH=$(python -c "from channeld_amd.build import source_hash; print(source_hash())")
echo "{\"source_hash\": \"$HASH\", \"tag\": \"$TAG\"}" > $O/stamp.json
bash tools/pmc_hbm.sh $TAG > $O/pmc.log 2>&1; grep -q '"k_fanout_emit_seg"' $O/hbm_traffic.json 2>/dev/null && cp $O/hbm_traffic.json $R/profiles/hbm_traffic.json
cd $R
timeout -s KILL 120 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 60 rocprofv3 --kernel-trace --stats -d $O/prof_serial -o kt -- python $R/bench.py --steps 50 --warmup 10 --only-timed > $O/prof_bench_serial.json 2> $O/prof_serial.err
cd $R
(echo "# source_hash $HASH; rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 --only-timed; first 10 dispatches of every kernel skipped"; python tools/rocpd_summary.py $O/prof_serial/kt_results.db 10) > $O/kernel_stats_serial.csv 2>> $O/prof_serial.err
rm -rf $O/prof_serial
cut -c1-400 $O/bench.json; head -5 $O/kernel_stats_serial.csv; tail -7 $O/pmc.log
