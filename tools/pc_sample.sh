#!/bin/bash
# Where do the waves of one kernel spend their time?  rocprofv3 PC sampling (host-trap, time based) of the bench,
# reduced to a histogram of the sampled instructions.
# usage (repo root on the GPU box): bash tools/pc_sample.sh <tag> <kernel regex> [interval_us] [bench args]
TAG=${1:-pcs}; KRE=${2:-k_fanout_emit}; IV=${3:-1}; METHOD=${PCS_METHOD:-host_trap}; UNIT=${PCS_UNIT:-time}; shift; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD \
  --pc-sampling-interval $IV --kernel-trace --output-format csv -d $O/raw -o p -- \
  python $R/bench.py --steps 12 --warmup 4 --no-cpu --latency-steps 0 "$@" > $O/pcs.out 2> $O/pcs.err
echo "rocprofv3 rc=$?" >> $O/pcs.err
cd $R
python - "$O" "$KRE" <<'PY'
import csv, glob, re, sys, collections, json
O, kre = sys.argv[1], re.compile(sys.argv[2])
files = glob.glob(O + "/raw/**/*pc_sampling*.csv", recursive=True)
print("pc sampling files:", [f.split("/")[-1] for f in files])
# dispatch ids of the wanted kernel
want = set()
for f in glob.glob(O + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kre.search(r.get("Kernel_Name", "")):
            want.add(r.get("Dispatch_Id"))
hist, n = collections.Counter(), 0
for f in files:
    rd = csv.DictReader(open(f))
    for r in rd:
        if want and r.get("Dispatch_Id") not in want:
            continue
        hist[(r.get("Instruction", "?"), r.get("Instruction_Comment", ""))] += 1
        n += 1
top = [{"instruction": k[0], "where": k[1], "samples": v, "pct": round(100.0 * v / max(n, 1), 2)} for k, v in hist.most_common(60)]
json.dump({"kernel": sys.argv[2], "samples": n, "top": top}, open(O + "/pc_hist.json", "w"), indent=1)
print("samples", n)
for t in top[:40]:
    print(f"{t['pct']:6.2f}%  {t['instruction'][:70]:70s} {t['where'][:60]}")
PY
rm -rf $O/raw
