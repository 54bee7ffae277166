#!/bin/bash
# k_fanout_emit_filt_cm's work-item size (CHD_FILT_ITEMS_TARGET: items per launch filt_items_block aims for; 1 = items of up to 64
# descriptors): alternating timed-region runs of the exact-stamp workload, on and off the tick grid.  usage: bash tools/ab_filt_items.sh <tag> <targets...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2; do
  for tj in 0 3000; do
    for v in "$@"; do
      CHD_FILT_ITEMS_TARGET=$v timeout -s KILL 100 python bench.py --steps 100 --warmup 20 --only-timed --arrival-jitter --tick-jitter-us $tj 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print('target $v tick_jitter $tj', 'ms_per_step %.4f' % j['ms_per_step'], 'record_kernels_us %.1f' % r['avg_launch_us'], 'frac %.3f' % r['frac'])"
    done
  done
done | tee $O/ab.txt
