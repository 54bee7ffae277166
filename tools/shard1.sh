#!/bin/bash
# The region-sharded tick as ONE rank runs it with the library's own RCCL communicator (chd_shard_tick; CHD_BENCH_FORCE_DIST makes the
# single rank of a one-GPU box take the multi-GPU path): rate, and the tick as a timeline.  usage: bash tools/shard1.sh <tag> [bench args]
# (e.g. --arrival-jitter: the update log by channel id, every update stamped at its enqueue time)
TAG=${1:-shard1}; shift; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export CHD_BENCH_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu --latency-steps 0 $*"
timeout 200 $RUN > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('sharded, one rank:', d['ms_per_step'], 'ms', d['value']/1e9, 'G msgs/s', d['config'].get('collectives_driver'))"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace -d $O/prof -o kt -- $RUN > $O/prof_bench.json 2> $O/prof.err
cd $R
DB=$(ls $O/prof/*/*_results.db $O/prof/*_results.db 2>/dev/null | head -1)
python tools/rocpd_timeline.py $DB k_ingest_by_channel 20 | tee $O/tick_timeline.csv
rm -rf $O/prof
