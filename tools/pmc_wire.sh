#!/bin/bash
# L2 behaviour of the wire builder's kernels: FETCH_SIZE / WRITE_SIZE / TCC hit + miss, one rocprofv3 --pmc pass each
# (counters only).  usage (repo root on the GPU box): bash tools/pmc_wire.sh <tag> [env assignments...]
TAG=${1:-pmcw}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  env "$@" timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -o p -- python $R/bench.py --steps 8 --warmup 6 --only-timed --wire 3 > $O/pmc_$n.out 2> $O/pmc_$n.err
done
cd $R
python - "$O" <<'PY'
import csv, glob, sys
from collections import defaultdict
O = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "wire" in k:
        print(k[:50], {c: [round(x / 1e6, 2) for x in v[-3:]] for c, v in d.items()}, "(millions; FETCH_SIZE / WRITE_SIZE in KiB -> x1e6 KiB)")
PY
rm -rf $O/pmc_*/
