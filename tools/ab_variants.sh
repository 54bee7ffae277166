#!/bin/bash
# Timed-region bench line per prebuilt kernel variant (python -m channeld_amd.build --variant <name> <flags> -> channeld_amd/variants/).
# usage: bash tools/ab_variants.sh <tag> [variants for the pipelined schedule too ...]
TAG=${1:-abv}; shift; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
run() { local name=$1; shift; env "${ENVV[@]}" timeout -s KILL 60 python bench.py --only-timed --steps 120 --warmup 20 "$@" > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(f"{sys.argv[2]:24s} ms/tick {d['ms_per_step']:.4f}  value {d['value']/1e9:7.1f} G  emit {r['avg_launch_us']:.1f} us frac {r['frac']:.3f}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
ENVV=(A=1); run base
for v in $R/channeld_amd/variants/libchd_*.so; do n=$(basename $v .so); n=${n#libchd_}; ENVV=(CHD_SPATIAL_LIB=$v); run $n; done
ENVV=(A=1); run base_pipe --headline pipelined
for n in "$@"; do ENVV=(CHD_SPATIAL_LIB=$R/channeld_amd/variants/libchd_$n.so); run ${n}_pipe --headline pipelined; done
