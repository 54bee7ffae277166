#!/bin/bash
# The diagnostic configurations at the committed sources, one bench line each (timed region only) + the source stamp.
# usage: bash tools/diag_final.sh <tag>      (≈ 3 minutes of box time after the first import)
TAG=${1:-diagf}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python -c "from channeld_amd import build; import json; print(json.dumps({'kernel_sources': build.source_hash()}))" > $O/stamp.json
run() { local name=$1; shift; timeout -s KILL 150 python bench.py --only-timed "$@" > $O/diag_$name.json 2> $O/diag_$name.err
  python - $O/diag_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    w = d.get("wire") or {}
    print(f"{sys.argv[2]:24s} ms/tick {d['ms_per_step']:.4f}  {d['value']/1e9:7.1f} G msgs/s  record kernel {r.get('avg_launch_us', 0):.1f} us frac {r['frac']:.3f}"
          + (f"  wire build {w.get('ms_per_build')} ms frac {w.get('frac_of_hbm_peak')}" if w else ""))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-400:])
PY
}
run update_frac_0.98 --steps 60 --warmup 10 --update-frac 0.98
run update_frac_0.9 --steps 60 --warmup 10 --update-frac 0.9
run update_frac_0.5 --steps 60 --warmup 10 --update-frac 0.5
run update_masks --steps 60 --warmup 10 --update-masks
run wire --steps 8 --warmup 6 --wire 3
run recipients --steps 60 --warmup 10 --recipients
run config_C_1M --steps 30 --warmup 10 --entities 1000000
run config_C_1M_arrival_jitter --steps 30 --warmup 10 --entities 1000000 --arrival-jitter
rm -f $O/*.err.empty; for f in $O/*.err; do [ -s $f ] || rm -f $f; done
