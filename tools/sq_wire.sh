#!/bin/bash
# SQ counters (cycles / instruction mix) of the wire builder's kernels; usage: bash tools/sq_wire.sh <tag> [env assignments...]
TAG=${1:-sqw}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  env "$@" timeout 200 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o p -- python $R/bench.py --steps 8 --warmup 6 --only-timed --wire 3 > $O/g$i.out 2> $O/g$i.err
done
cd $R
python - "$O" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
O = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(O + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: max(v) for c, v in d.items()} for k, d in acc.items() if "wire" in k}
json.dump(out, open(O + "/sq_wire_summary.json", "w"), indent=1)
for k, d in out.items():
    w = max(d.get("SQ_WAVES", 1), 1)
    print(f"{k[:44]:44s} waves {w:7.0f} | per wave: cycles {4*d.get('SQ_WAVE_CYCLES',0)/w:9.0f} wait {4*d.get('SQ_WAIT_ANY',0)/w:9.0f} issue-stall {4*d.get('SQ_WAIT_INST_ANY',0)/w:8.0f} active {4*d.get('SQ_ACTIVE_INST_ANY',0)/w:8.0f} | VALU {d.get('SQ_INSTS_VALU',0)/w:7.0f} SALU {d.get('SQ_INSTS_SALU',0)/w:7.0f} LDS {d.get('SQ_INSTS_LDS',0)/w:6.0f} VMEM_RD {d.get('SQ_INSTS_VMEM_RD',0)/w:5.0f} VMEM_WR {d.get('SQ_INSTS_VMEM_WR',0)/w:5.0f} SMEM {d.get('SQ_INSTS_SMEM',0)/w:5.0f}")
PY
rm -rf $O/g*/
