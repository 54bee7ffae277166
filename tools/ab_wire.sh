# A/B runs of the descriptor-driven wire builder's copy kernel (GPU box, repo root): tools/ab_wire.sh [pytest]
mkdir -p gpurun_out/r03p
if [ -n "$1" ]; then timeout 120 python -m pytest tests/test_gpu_wire.py -x -q -p no:cacheprovider 2>&1 | tail -3 || exit 1; fi
run() {  # run <label> <env...>
  local label=$1; shift
  env "$@" timeout 60 python bench.py --steps 8 --warmup 6 --only-timed --wire 6 > gpurun_out/r03p/wire_$label.json 2> gpurun_out/r03p/wire_$label.err || { tail -3 gpurun_out/r03p/wire_$label.err; return; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03p/wire_$label.json") if l.startswith("{")][-1])["wire"]
print("$label", d["ms_per_build_all"], d["GB_per_build_all"], d["written_GBps_all"])
PY
}
run spec CHD_X=1
bash tools/trace_one.sh r03p wire 0 --steps 8 --warmup 6 --only-timed --wire 4 | grep -i "wire\|kernel\|scan"
