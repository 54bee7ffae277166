#!/bin/bash
# CHD_WORLD_OVERLAP_DEFERRED (the filtering launch + epilogue beside the record kernel, serial schedule): the HBM PMC passes at
# these sources first (what bench.py quotes), then the A/B, a bench run with the flag on whose latency phase checks 100 ticks'
# record digests, and the world / full-size / wire parity tests with the flag forced on every world (CHD_WORLD_FORCE_FLAGS=256).
# usage: bash tools/side_check.sh <tag>
TAG=${1:-side}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
HASH=$(python -c "from channeld_amd.build import source_hash; print(source_hash())")
echo "{\"source_hash\": \"$HASH\", \"tag\": \"$TAG\"}" > $O/stamp.json
bash tools/pmc_hbm.sh $TAG > $O/pmc.log 2>&1; grep -q '"k_fanout_emit_seg"' $O/hbm_traffic.json 2>/dev/null && cp $O/hbm_traffic.json $R/profiles/hbm_traffic.json
cd $R
line() { python - "$@" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(f"{sys.argv[2]:22s} tick {d['ms_per_step']*1e3:7.2f} us  {r['kernel'][:18]:18s} {r['avg_launch_us']:7.2f} us  frac {r['frac']:.4f}  digests {d.get('digest_checked_ticks')}  stages {[round(v,1) for v in d['stage_us_avg'].values()]}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for rep in a b; do for v in 0 1; do
  timeout -s KILL 60 python bench.py --only-timed --steps 150 --warmup 10 --overlap-deferred $v > $O/ab_${v}_$rep.json 2> $O/ab_${v}_$rep.err; line $O/ab_${v}_$rep.json "headline od=$v ($rep)"
done; done
for v in 0 1; do
  timeout -s KILL 60 python bench.py --only-timed --steps 60 --warmup 10 --update-frac 0.9 --overlap-deferred $v > $O/uf09_$v.json 2> $O/uf09_$v.err; line $O/uf09_$v.json "update-frac 0.9 od=$v"
done
timeout -s KILL 90 python bench.py --overlap-deferred 1 --no-cpu --e2e-ticks 0 --steps 100 --warmup 10 --latency-steps 100 > $O/bench_od1_digests.json 2> $O/bench_od1.err; line $O/bench_od1_digests.json "od=1 + 100 digests"
CHD_WORLD_FORCE_FLAGS=256 timeout -s KILL ${PYTEST_LIMIT:-110} python -m pytest tests/test_gpu_world.py tests/test_gpu_wire.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider --durations=5 \
  -k "not config_c and not pipelined and not cell_major and not merged and not full]" > $O/pytest_forced.log 2>&1; echo "pytest rc=$?" >> $O/pytest_forced.log
tail -12 $O/pytest_forced.log; cat $O/pmc.log | tail -8
