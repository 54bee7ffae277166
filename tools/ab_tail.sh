#!/bin/bash
# k_fanout_emit_seg: the last connections cut into finer pieces (CHD_SEG_TAIL="<percent>,<log2 pieces>") — timed-region-only
# bench runs per setting, then one run whose latency phase checks the record digests with the setting on.
# usage: bash tools/ab_tail.sh <tag> [settings "<percent>,<log2 pieces>[:<persistent waves per CU>]" ...]
TAG=${1:-tail}; shift; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
SET=${@:-"0,2 10,2 20,2 35,2 20,3 50,3 0,2"}
for s in $SET; do
  pc=${s#*:}; [ "$pc" = "$s" ] && pc=8; s=${s%%:*}
  CHD_EMIT_WAVES_PER_CU=$pc CHD_SEG_TAIL=$s timeout -s KILL 60 python bench.py --only-timed --steps 150 --warmup 10 > $O/t_$s.json 2> $O/t_$s.err
  python - $O/t_$s.json $s $pc <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(f"tail {sys.argv[2]:6s} waves/CU {sys.argv[3]} tick {d['ms_per_step']*1e3:7.2f} us  emit_seg {r['avg_launch_us']:7.2f} us  frac {r['frac']:.4f}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
