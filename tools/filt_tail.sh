#!/bin/bash
# VERDICT r4 #4 ("find the 293 us launches first"): per launch of k_fanout_emit_filt_cm, how long it took and how many records it
# wrote, on ticks off the 50 ms grid.  usage (repo root on the GPU box): bash tools/filt_tail.sh <tag>
TAG=${1:-filt_tail}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
HASH=$(cd $R && python -c "from channeld_amd.build import source_hash; print(source_hash())")
for v in offgrid ongrid; do
  A=""; [ $v = offgrid ] && A="--tick-jitter-us 3000"
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o kt -- python $R/bench.py --steps 40 --warmup 10 --only-timed --arrival-jitter $A --history-out $O/history_$v.json > $O/bench_$v.json 2> $O/prof_$v.err
  (echo "# source_hash $HASH; bench.py --steps 40 --warmup 10 --only-timed --arrival-jitter $A; k_fanout_emit_filt_cm per launch"; python $R/tools/filt_tail.py $O/prof_$v/kt_results.db $O/history_$v.json) > $O/filt_launches_$v.csv 2>> $O/prof_$v.err
  (echo "# same run; k_fanout_emit_seg per launch (records = n_records - n_deferred_records)"; python - <<PY
import json
h = json.load(open("$O/history_$v.json"))
for x in h: x["n_seg"] = x["n_records"] - x["n_deferred_records"]
json.dump(h, open("$O/history_${v}_seg.json", "w"))
PY
  python $R/tools/filt_tail.py $O/prof_$v/kt_results.db $O/history_${v}_seg.json k_fanout_emit_seg n_seg) > $O/seg_launches_$v.csv 2>> $O/prof_$v.err
  rm -rf $O/prof_$v
  tail -1 $O/filt_launches_$v.csv; tail -1 $O/seg_launches_$v.csv
done
