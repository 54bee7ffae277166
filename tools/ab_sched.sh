#!/bin/bash
# A/B of tick schedules on the bench workload (timed region only).  usage: bash tools/ab_sched.sh <tag>
TAG=${1:-ab}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
run() { local name=$1; shift; env "${ENVV[@]}" timeout -s KILL 120 python bench.py --only-timed --steps 200 --warmup 20 "$@" > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(f"{sys.argv[2]:28s} ms/tick {d['ms_per_step']:.4f}  value {d['value']/1e9:7.1f} G  emit {r['avg_launch_us']:.1f} us frac {r['frac']:.3f}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
ENVV=(A=1)
run serial --serial-ticks
run serial_overlap --serial-ticks --overlap-interest
run pipelined --headline pipelined
ENVV=(CHD_EMIT_WAVES_PER_CU=6);  run pipelined_w6 --headline pipelined
ENVV=(CHD_EMIT_WAVES_PER_CU=12); run pipelined_w12 --headline pipelined
ENVV=(CHD_EMIT_WAVES_PER_CU=12); run serial_overlap_w12 --serial-ticks --overlap-interest
