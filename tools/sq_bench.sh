#!/bin/bash
# Three SQ counter passes (cycles / instruction mix) over the timed region of a bench configuration; per-wave figures for every k_* kernel.
# usage: bash tools/sq_bench.sh <tag> [bench args...]    -> gpurun_out/<tag>/sq_summary.json
TAG=${1:-sq}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o p -- python $R/bench.py --steps 12 --warmup 4 --only-timed "$@" > $O/g$i.out 2> $O/g$i.err
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
out = {k: {c: sum(v[4:]) / max(1, len(v[4:])) for c, v in d.items()} for k, d in acc.items() if k.startswith("k_")}
json.dump(out, open(O + "/sq_summary.json", "w"), indent=1)
for k, d in sorted(out.items()):
    w = max(d.get("SQ_WAVES", 1), 1)
    print(f"{k[:44]:44s} waves {w:7.0f} gui {d.get('GRBM_GUI_ACTIVE',0):8.0f} busy {d.get('SQ_BUSY_CYCLES',0):9.0f} | per wave: cycles {4*d.get('SQ_WAVE_CYCLES',0)/w:9.0f} wait {4*d.get('SQ_WAIT_ANY',0)/w:9.0f} istall {4*d.get('SQ_WAIT_INST_ANY',0)/w:8.0f} active {4*d.get('SQ_ACTIVE_INST_ANY',0)/w:8.0f} | VALU {d.get('SQ_INSTS_VALU',0)/w:7.0f} SALU {d.get('SQ_INSTS_SALU',0)/w:7.0f} LDS {d.get('SQ_INSTS_LDS',0)/w:6.0f} RD {d.get('SQ_INSTS_VMEM_RD',0)/w:5.0f} WR {d.get('SQ_INSTS_VMEM_WR',0)/w:5.0f} SMEM {d.get('SQ_INSTS_SMEM',0)/w:5.0f} | bankconf {d.get('SQ_LDS_BANK_CONFLICT',0):9.0f}")
PY
rm -rf $O/g*/
