#!/bin/bash
# Same-box A/B of the tree's library against a reference build (channeld_amd/variants/libchd_base.so): alternating timed-region runs.
# usage: bash tools/ab_base.sh <tag> [bench args...]
TAG=${1:-ab}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2 3; do
  for v in base new; do
    L=""; [ $v = base ] && L=$R/channeld_amd/variants/libchd_base.so
    CHD_SPATIAL_LIB=$L timeout -s KILL 100 python bench.py --steps 200 --warmup 20 --only-timed "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$v', 'ms_per_step %.4f' % j['ms_per_step'], 'emit_us %.1f' % j['roofline']['avg_launch_us'])"
  done
done | tee $O/ab.txt
