#!/bin/bash
# One GPU-box call for a round's evidence, at the tree it is given: smoke, bench line, rocprofv3 kernel trace of the timed
# region (serial headline; PIPE_TRACE=1: the pipelined schedule too; WIRE_TRACE=1: the wire builder), the two HBM PMC passes
# (PMC=1), the diagnostic configurations (DIAG=1: partial updates, masks), the self-verifying sharded bench (ranks share the
# GPU, gloo), then the GPU parity suite.  Every summary is stamped with the kernel-source hash (channeld_amd.build.source_hash:
# the GPU box has no .git).
# usage (repo root on the GPU box): [PMC=1] [WIRE_TRACE=1] [PIPE_TRACE=1] [JITTER_TRACE=1] [AB_GATE=1] [TIMELINE=1] [SHARD1=1] [DIAG=1|light] bash tools/round_check.sh <tag> [pytest args]
TAG=${1:-round}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
HASH=$(python -c "from channeld_amd.build import source_hash; print(source_hash())")
echo "{\"source_hash\": \"$HASH\", \"tag\": \"$TAG\"}" > $O/stamp.json
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
# the HBM counters first: bench.py quotes roofline.traffic from profiles/hbm_traffic.json only when that file carries the hash of the
# sources it runs, so the bench line of this call quotes this call's counters
if [ -n "$PMC" ]; then bash tools/pmc_hbm.sh $TAG > $O/pmc.log 2>&1; grep -q '"k_fanout_emit_seg"' $O/hbm_traffic.json 2>/dev/null && cp $O/hbm_traffic.json $R/profiles/hbm_traffic.json; fi
timeout -s KILL 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
trace() {  # trace <name> <skip> <bench args...>
  local name=$1 skip=$2; shift 2
  cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o kt -- python $R/bench.py "$@" > $O/prof_bench_$name.json 2> $O/prof_$name.err
  cd $R
  (echo "# source_hash $HASH; rocprofv3 --kernel-trace --stats -- python bench.py $*; first $skip dispatches of every kernel skipped"; python tools/rocpd_summary.py $O/prof_$name/kt_results.db $skip) > $O/kernel_stats_$name.csv 2>> $O/prof_$name.err
  rm -rf $O/prof_$name
}
trace serial 10 --steps 50 --warmup 10 --only-timed
[ -n "$PIPE_TRACE" ] && trace pipelined 10 --steps 50 --warmup 10 --only-timed --headline pipelined
[ -n "$WIRE_TRACE" ] && trace wire 0 --steps 8 --warmup 6 --only-timed --wire 3
if [ -n "$JITTER_TRACE" ]; then  # arrival stamps at enqueue time (exact update buffers): ticks on and off the 50 ms grid
  trace arrival_jitter 10 --steps 40 --warmup 10 --only-timed --arrival-jitter
  trace arrival_jitter_offgrid 10 --steps 40 --warmup 10 --only-timed --arrival-jitter --tick-jitter-us 3000
fi
if [ -n "$AB_GATE" ]; then  # the bench's schedule (interest stream forked / joined by device-side flags) against the HIP-event form, and against one stream
  bash tools/ab_gate.sh $TAG > /dev/null 2>> $O/ab.err
  (echo '{"what": "bench.py --only-timed --steps 200 --warmup 20 --overlap-interest 0 (everything on one stream)", "source_hash": "'$HASH'", "ms_per_step": ['
   for i in 1 2; do timeout 120 python bench.py --only-timed --steps 200 --warmup 20 --overlap-interest 0 2>> $O/ab.err | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'], ',')"; done
   echo '0]}') > $O/one_stream.json
fi
[ -n "$TIMELINE" ] && bash tools/timeline.sh ${TAG}_tl > /dev/null 2>&1 && cp $R/gpurun_out/${TAG}_tl/tick_timeline.csv $O/tick_timeline.csv
[ -n "$SHARD1" ] && bash tools/shard1.sh ${TAG}_shard1 > $O/shard1.log 2>&1 && cp $R/gpurun_out/${TAG}_shard1/tick_timeline.csv $O/shard_tick_timeline_one_rank.csv && cp $R/gpurun_out/${TAG}_shard1/bench.json $O/bench_shard_one_rank_rccl.json
if [ -n "$DIAG" ]; then
  for f in 0.98 0.9 0.5; do timeout 120 python bench.py --only-timed --steps 60 --warmup 10 --update-frac $f > $O/diag_update_frac_$f.json 2>> $O/diag.err; done
  timeout 120 python bench.py --only-timed --steps 60 --warmup 10 --update-masks > $O/diag_update_masks.json 2>> $O/diag.err
  timeout 120 python bench.py --only-timed --steps 8 --warmup 6 --wire 3 > $O/diag_wire.json 2>> $O/diag.err
  if [ "$DIAG" != light ]; then
    trace uf09 10 --steps 50 --warmup 10 --only-timed --update-frac 0.9
    # the wire builder on a partially updating world and with merged updates (bench.py --wire N: "wire" in the line)
    timeout 120 python bench.py --only-timed --steps 12 --warmup 6 --wire 6 --update-frac 0.9 > $O/diag_wire_update_frac_0.9.json 2>> $O/diag.err
    timeout 120 python bench.py --only-timed --steps 12 --warmup 6 --wire 6 --update-masks > $O/diag_wire_merged_updates.json 2>> $O/diag.err
  fi
fi
CHD_BENCH_SHARE_GPU=1 CHD_DIST_BACKEND=gloo timeout 200 python bench.py --gpus 2 --steps 20 --warmup 5 --verify 3 --latency-steps 0 --no-cpu --entities 50000 --subs 5000 --max-records 400000000 > $O/bench_2ranks_shared_gpu_verified.json 2> $O/bench_2ranks.err
timeout -s KILL ${PYTEST_LIMIT:-420} python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 "$@" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -16 $O/pytest_gpu.log; cat $O/bench.json | cut -c1-700; head -14 $O/kernel_stats_serial.csv; [ -f $O/kernel_stats_wire.csv ] && head -8 $O/kernel_stats_wire.csv; [ -f $O/pmc.log ] && cat $O/pmc.log; tail -2 $O/smoke.log; cut -c1-300 $O/bench_2ranks_shared_gpu_verified.json
