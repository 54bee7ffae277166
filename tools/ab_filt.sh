#!/bin/bash
# A/B of the exact-stamp record path: work items of 64 / 32 / 16 descriptors (build variants) x cells in arrival order on / off
# (CHD_SORT_ARRIVALS), ticks on and off the 50 ms grid.  usage: bash tools/ab_filt.sh <tag>   (variants built beforehand:
# python -m channeld_amd.build --variant d32 -DFC_DESCS=32; ... d16 -DFC_DESCS=16)
TAG=${1:-ab_filt}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
echo "lib,sort,ticks,ms_per_step,record_kernels_frac,avg_launch_us" > $O/ab.csv
for lib in default d32 d16; do
  for so in 1 0; do
    for v in ongrid offgrid; do
      A=""; [ $v = offgrid ] && A="--tick-jitter-us 3000"
      L=""; [ $lib != default ] && L="$R/channeld_amd/variants/libchd_$lib.so"
      CHD_SPATIAL_LIB=$L CHD_SORT_ARRIVALS=$so timeout 120 python bench.py --only-timed --steps 60 --warmup 10 --arrival-jitter $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$lib,$so,$v,%.4f,%.3f,%.1f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us']))" >> $O/ab.csv
    done
  done
done
cat $O/ab.csv
