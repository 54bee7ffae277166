#!/bin/bash
# Same-box A/B of one environment setting: alternating timed-region runs.  usage: bash tools/ab_env.sh <tag> VAR=a VAR=b [bench args...]
TAG=$1; A=$2; B=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2 3; do
  for v in "$A" "$B"; do
    env $v timeout -s KILL 100 python bench.py --steps 200 --warmup 20 --only-timed "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$v', 'ms_per_step %.4f' % j['ms_per_step'], 'emit_us %.1f' % j['roofline']['avg_launch_us'])"
  done
done | tee $O/ab.txt
