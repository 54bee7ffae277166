#!/bin/bash
# k_fanout_emit_filt_cm A/B against channeld_amd/variants/libchd_base.so: the exact-stamp tests, then alternating timed-region runs
# on and off the tick grid, for each CHD_FILT_ITEMS_TARGET given.  usage: bash tools/ab_filt_loader.sh <tag> <targets...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -s KILL 400 python -m pytest tests/test_gpu_deep.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -x -k "arrival or exact or offs or cells_in" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
  for tj in 0 3000; do
    for v in base "$@"; do
      L=""; T=$v; [ $v = base ] && L=$R/channeld_amd/variants/libchd_base.so && T=0
      CHD_SPATIAL_LIB=$L CHD_FILT_ITEMS_TARGET=$T timeout -s KILL 100 python bench.py --steps 100 --warmup 20 --only-timed --arrival-jitter --tick-jitter-us $tj 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print('$v tick_jitter $tj', 'ms_per_step %.4f' % j['ms_per_step'], 'record_kernels_us %.1f' % r['avg_launch_us'], 'frac %.3f' % r['frac'])"
    done
  done
done | tee $O/ab.txt
