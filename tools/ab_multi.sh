#!/bin/bash
# Exact-stamp workload, on and off the tick grid: alternating timed-region runs of several libraries (tree = the tree's own, the others
# channeld_amd/variants/libchd_<name>.so), after the exact-stamp parity tests on each of the names in TEST.
# usage: [TEST="tree w8"] bash tools/ab_multi.sh <tag> <names...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for v in $TEST; do
  L=""; [ $v != tree ] && L=$R/channeld_amd/variants/libchd_$v.so
  CHD_SPATIAL_LIB=$L timeout -s KILL 400 python -m pytest tests/test_gpu_deep.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -x -k "arrival or exact or offs or cells_in" > $O/pytest_$v.log 2>&1; echo "$v: $(tail -1 $O/pytest_$v.log)"
done
for rep in 1 2; do
  for tj in 0 3000; do
    for v in "$@"; do
      L=""; [ $v != tree ] && L=$R/channeld_amd/variants/libchd_$v.so
      CHD_SPATIAL_LIB=$L timeout -s KILL 100 python bench.py --steps 100 --warmup 20 --only-timed --arrival-jitter --tick-jitter-us $tj 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print('$v tick_jitter $tj', 'ms_per_step %.4f' % j['ms_per_step'], 'record_kernels_us %.1f' % r['avg_launch_us'], 'frac %.3f' % r['frac'])"
    done
  done
done | tee $O/ab.txt
