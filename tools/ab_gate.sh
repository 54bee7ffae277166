#!/bin/bash
# A/B of the serial tick's cross-stream dependencies: HIP events against device-side flags (CHD_WORLD_GATED_OVERLAP), alternating.
TAG=${1:-ab_gate}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
HASH=$(python -c "from channeld_amd.build import source_hash; print(source_hash())")
(echo '{"what": "bench.py --only-timed --steps 200 --warmup 20 (serial schedule, --overlap-interest 1, --prof-every 7), --gated-overlap 0 / 1 alternating", "source_hash": "'$HASH'", "runs": ['
 for i in 1 2 3; do for v in 0 1; do
   ms=$(timeout 120 python bench.py --only-timed --steps 200 --warmup 20 --gated-overlap $v 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])")
   echo "  {\"gated_overlap\": $v, \"ms_per_step\": $ms},"
 done; done
 echo '  {}]}') | tee $O/gated_overlap_ab.json
