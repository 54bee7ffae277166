#!/bin/bash
# SQ counter passes over the bench (one rocprofv3 --pmc run per group; counters only, no tracing).
# usage: bash tools/sq_pass.sh <tag> [bench args]   -> gpurun_out/<tag>/sq_<kernel>.json
TAG=${1:-sq}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --latency-steps 0 "$@" > $O/g$i.out 2> $O/g$i.err
done
cd $R
python - "$O" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
O = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(O + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v[5:]) / max(1, len(v[5:])) for c, v in d.items()} for k, d in acc.items() if k.startswith("k_")}
json.dump(out, open(O + "/sq_summary.json", "w"), indent=1)
for k in ("k_fanout_emit", "k_aoi_interest", "k_index_scatter", "k_ingest"):
    if k in out: print(k, json.dumps(out[k]))
PY
rm -rf $O/g*/
