#!/bin/bash
# A/B of emit kernel variants on one GPU box: bench lines (no CPU baseline, no e2e legs) per variant.
# usage: bash tools/ab_emit.sh <tag>
TAG=${1:-ab}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
B="--steps 60 --warmup 10 --no-cpu --e2e-ticks 0 --latency-steps 0"
run() { name=$1; shift; env "$@" timeout -s KILL 200 python bench.py $B $EXTRA > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:28s} tick {d['ms_per_step']*1e3:7.1f} us  emit {d['stage_us_avg']['emit']:7.1f} us  frac {d['roofline']['frac']:.3f}  msgs/tick {d['config']['msgs_per_tick']:.0f}  stages {[round(v,1) for v in d['stage_us_avg'].values()]}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
EXTRA=""
run pipelined A=1
run old CHD_EMIT_PIPELINED=0
EXTRA="--update-frac 0.5"; run half A=1; run half_old CHD_EMIT_PIPELINED=0; EXTRA=""
for v in $R/channeld_amd/variants/libchd_*.so; do n=$(basename $v .so); run ${n#libchd_} CHD_SPATIAL_LIB=$v; done
