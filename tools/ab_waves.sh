#!/bin/bash
# k_fanout_emit_seg's persistent waves per CU (CHD_EMIT_WAVES_PER_CU) against workloads with short descriptors (small cells, small AOI),
# few connections, and the headline; timed-region runs.  usage: bash tools/ab_waves.sh <tag> <waves...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
run() {  # run <name> <waves> [bench args]
  local name=$1 v=$2; shift 2
  CHD_EMIT_WAVES_PER_CU=$v timeout -s KILL 100 python bench.py --steps 80 --warmup 16 --only-timed "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print('$name waves_per_cu $v', 'ms_per_step %.4f' % j['ms_per_step'], 'emit_us %.1f' % r['avg_launch_us'], 'frac %.3f' % r['frac'])"
}
for v in "$@"; do run entities_10k $v --entities 10000; done
for v in "$@"; do run entities_30k $v --entities 30000; done
for v in "$@"; do run aoi_0.5 $v --aoi-scale 0.5; done
for v in "$@"; do run subs_1k $v --subs 1000; done
for v in "$@"; do run headline $v; done
for v in "$@"; do run update_frac_0.9 $v --update-frac 0.9; done
for v in "$@"; do run arrival_jitter $v --arrival-jitter; done
