#!/bin/bash
# CHD_EMIT_ACTIVE_THRESHOLDS="t1,t2" (records per connection from which a tick's k_fanout_emit_seg runs 8 / 12 active waves per CU; below t2: 16)
# usage: bash tools/ab_active.sh <tag> <"t1,t2" ...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
run() { local name=$1 v=$2; shift 2
  CHD_EMIT_ACTIVE_THRESHOLDS=$v timeout -s KILL 100 python bench.py --steps 100 --warmup 16 --only-timed "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print('$name thresholds $v', 'ms_per_step %.4f' % j['ms_per_step'], 'emit_us %.1f' % r['avg_launch_us'], 'frac %.3f' % r['frac'])"; }
for rep in 1 2; do
  for v in "$@"; do run headline $v; done
  for v in "$@"; do run aoi_0.5 $v --aoi-scale 0.5; done
  for v in "$@"; do run update_frac_0.9 $v --update-frac 0.9; done
  for v in "$@"; do run entities_30k $v --entities 30000; done
  for v in "$@"; do run arrival_jitter $v --arrival-jitter; done
done | tee $O/ab.txt
