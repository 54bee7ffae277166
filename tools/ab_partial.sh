#!/bin/bash
# A/B: descriptor path (plan_seg + emit_seg + deferred) vs the first connection-major form on partially updating worlds
TAG=${1:-abp}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
run() { local name=$1; shift; env "${ENVV[@]}" timeout -s KILL 120 python bench.py --only-timed --steps 100 --warmup 20 "$@" > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(f"{sys.argv[2]:28s} ms/tick {d['ms_per_step']:.4f}  value {d['value']/1e9:7.1f} G  msgs/tick {d['config']['msgs_per_tick']/1e6:6.1f} M emit {r['avg_launch_us']:.1f} us frac {r['frac']:.3f} deferred {r['deferred_msgs_per_tick']:.0f}")
except Exception as e: print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace('.json','.err')).read()[-300:])
PY
}
for f in 0.5 0.9 0.98; do
ENVV=(A=1); run seg_frac_$f --update-frac $f
ENVV=(CHD_EMIT_PIPELINED=0); run old_frac_$f --update-frac $f
done
ENVV=(CHD_EMIT_PIPELINED=0); run old_full
