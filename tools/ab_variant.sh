#!/bin/bash
# Exact-stamp workload, on and off the tick grid: alternating timed-region runs of the tree's library and channeld_amd/variants/libchd_<name>.so
# (python -m channeld_amd.build --variant <name> <flags>).  usage: bash tools/ab_variant.sh <tag> <name> [bench args]
TAG=$1; NAME=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2 3; do
  for tj in 0 3000; do
    for v in tree $NAME; do
      L=""; [ $v != tree ] && L=$R/channeld_amd/variants/libchd_$v.so
      CHD_SPATIAL_LIB=$L timeout -s KILL 100 python bench.py --steps 100 --warmup 20 --only-timed --arrival-jitter --tick-jitter-us $tj "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print('$v tick_jitter $tj', 'ms_per_step %.4f' % j['ms_per_step'], 'record_kernels_us %.1f' % r['avg_launch_us'], 'frac %.3f' % r['frac'])"
    done
  done
done | tee $O/ab.txt
