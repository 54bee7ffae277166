#!/bin/bash
# Same-box A/B of the tree's library against a reference build (channeld_amd/variants/libchd_base.so) on short-descriptor workloads, few
# connections, partial updates and the headline; alternating timed-region runs.  usage: bash tools/ab_base_small.sh <tag>
TAG=${1:-abs}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
run() { local name=$1 v=$2; shift 2; local L=""; [ $v = base ] && L=$R/channeld_amd/variants/libchd_base.so
  CHD_SPATIAL_LIB=$L timeout -s KILL 100 python bench.py --steps 100 --warmup 16 --only-timed "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print('$name $v', 'ms_per_step %.4f' % j['ms_per_step'], 'emit_us %.1f' % r['avg_launch_us'], 'frac %.3f' % r['frac'])"; }
for rep in 1 2; do
  for v in base new; do run entities_10k $v --entities 10000; done
  for v in base new; do run entities_30k $v --entities 30000; done
  for v in base new; do run aoi_0.5 $v --aoi-scale 0.5; done
  for v in base new; do run subs_1k $v --subs 1000; done
  for v in base new; do run update_frac_0.9 $v --update-frac 0.9; done
  for v in base new; do run update_masks $v --update-masks; done
  for v in base new; do run arrival_jitter $v --arrival-jitter; done
  for v in base new; do run headline $v; done
done | tee $O/ab.txt
