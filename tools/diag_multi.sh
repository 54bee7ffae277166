#!/bin/bash
# BASELINE's multi-GPU configurations at their stated sizes with the ranks sharing ONE GPU (gloo), each self-verified against
# the single-world oracle (bench.py --verify), and the single-GPU side configurations (config C, R = 1.5 cells, handover
# recipients).  usage (repo root on the GPU box): bash tools/diag_multi.sh <tag>
TAG=${1:-multi}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
HASH=$(python -c "from channeld_amd.build import source_hash; print(source_hash())")
echo "{\"source_hash\": \"$HASH\", \"tag\": \"$TAG\"}" > $O/stamp.json
export CHD_BENCH_SHARE_GPU=1 CHD_DIST_BACKEND=gloo
timeout 300 python bench.py --gpus 4 --config D --verify 3 --steps 3 --warmup 3 --latency-steps 0 --no-cpu --max-records 400000000 > $O/bench_config_D_fullsize_4ranks_shared_gpu.json 2> $O/config_D.err; echo "D rc=$?"
timeout 600 python bench.py --gpus 8 --config E --verify 2 --steps 2 --warmup 2 --latency-steps 0 --no-cpu --max-records 2400000000 > $O/bench_config_E_fullsize_8ranks_shared_gpu.json 2> $O/config_E.err; echo "E rc=$?"
unset CHD_BENCH_SHARE_GPU CHD_DIST_BACKEND
timeout 200 python bench.py --only-timed --steps 20 --warmup 5 --entities 1000000 > $O/diag_config_C.json 2> $O/config_C.err; echo "C rc=$?"
timeout 200 python bench.py --only-timed --steps 20 --warmup 5 --entities 1000000 --emit conn-major > $O/diag_config_C_conn_major.json 2>> $O/config_C.err
timeout 100 python bench.py --only-timed --steps 60 --warmup 10 --aoi-scale 0.5 > $O/diag_aoi_scale_0.5.json 2> $O/diag.err
timeout 100 python bench.py --only-timed --steps 60 --warmup 10 --recipients > $O/diag_recipients.json 2>> $O/diag.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    ls = [l for l in open(f) if l.startswith("{")]
    if not ls: print(f, "NO LINE"); continue
    d = json.loads(ls[-1])
    if "value" not in d: continue
    v = d.get("verified") or {}
    print(f.split("/")[-1], "value", round(d["value"] / 1e9, 1), "G/s ms", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 3), d["roofline"]["kernel"][:28], "verified", d.get("verified_ticks"), v.get("msgs_per_verified_tick"), v.get("seconds"))
PY
