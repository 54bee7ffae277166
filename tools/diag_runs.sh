#!/bin/bash
# Diagnostic bench configurations (timed region only), one summary line each.  usage: bash tools/diag_runs.sh <tag>
TAG=${1:-diag}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
run() { local name=$1; shift; timeout -s KILL 200 python bench.py --only-timed "$@" > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(f"{sys.argv[2]:22s} ms/tick {d['ms_per_step']:.4f}  value {d['value']/1e9:7.1f} G  msgs/tick {d['config']['msgs_per_tick']/1e6:7.1f} M  emit {r['avg_launch_us']:.1f} us frac {r['frac']:.3f}  deferred/tick {r['deferred_msgs_per_tick']:.0f}")
except Exception as e: print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace('.json','.err')).read()[-300:])
PY
}
run headline --steps 200 --warmup 20
run update_masks --steps 100 --warmup 20 --update-masks
run update_frac_0.5 --steps 100 --warmup 20 --update-frac 0.5
run recipients --steps 100 --warmup 20 --recipients
run aoi_scale_0.5 --steps 100 --warmup 20 --aoi-scale 0.5
run flat_50ms --steps 100 --warmup 20 --flat-interval-ms 50
run config_c_1M --steps 30 --warmup 10 --entities 1000000
run config_c_1M_connmajor --steps 30 --warmup 10 --entities 1000000 --emit conn-major
run config_c_1M_cellmajor_half --steps 30 --warmup 10 --entities 1000000 --update-frac 0.5
[ -n "$SQ" ] && { bash tools/sq_quick.sh $TAG/sq > $O/sq.log 2>&1; tail -5 $O/sq.log; }
