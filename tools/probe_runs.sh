#!/bin/bash
# Unusual combinations of the bench's workload flags, one line each (timed region only): looking for cliffs, not for records.
# usage: bash tools/probe_runs.sh <tag>
TAG=${1:-probe}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
run() { local name=$1; shift; timeout -s KILL 120 python bench.py --only-timed "$@" > $O/probe_$name.json 2> $O/probe_$name.err
  python - $O/probe_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:28s} ms/tick {d['ms_per_step']:9.4f}  {d['value']/1e9:7.1f} G msgs/s  msgs/tick {d['config']['msgs_per_tick']/1e6:8.1f} M  record kernels {r.get('avg_launch_us', 0):8.1f} us frac {r['frac']:.3f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
}
run c1m_recipients --steps 20 --warmup 6 --entities 1000000 --recipients
run c1m_update_frac_0.9 --steps 20 --warmup 6 --entities 1000000 --update-frac 0.9
run c1m_update_masks --steps 20 --warmup 6 --entities 1000000 --update-masks
run subs_100k --steps 20 --warmup 6 --subs 100000
run aj_update_frac_0.9 --steps 40 --warmup 10 --arrival-jitter --update-frac 0.9
run aj_recipients --steps 40 --warmup 10 --arrival-jitter --recipients
run tick_20ms --steps 60 --warmup 10 --tick-ms 20
run aj_tick_20ms --steps 60 --warmup 10 --tick-ms 20 --arrival-jitter
run entities_10k --steps 60 --warmup 10 --entities 10000
run subs_1k --steps 60 --warmup 10 --subs 1000
for f in $O/*.err; do [ -s $f ] || rm -f $f; done
