#!/bin/bash
# The wire builder's kernels of one build as a table (rocprofv3 kernel trace of `bench.py --wire 6 --only-timed ...`): per kernel name the
# calls, total and average duration over the run, largest first.   usage: bash tools/wire_timeline.sh <tag> [bench args]
TAG=${1:-wire_tl}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace -d $O/prof -o kt -- python $R/bench.py --steps 4 --warmup 2 --no-cpu --latency-steps 0 --e2e-ticks 0 --wire 6 "$@" > $O/bench.json 2> $O/prof.err
cd $R
python - <<PY
import sqlite3, json
c=sqlite3.connect("$O/prof/kt_results.db").cursor()
rows=c.execute("select name, start, end from kernels order by start").fetchall()
from collections import defaultdict
# the last build: kernels after the last k_tick epilogue-ish marker (k_fanout_tail) up to the end
names=[r[0].split("(")[0].replace("void ","") for r in rows]
last=max(i for i,n in enumerate(names) if n.startswith("k_fanout_tail"))
agg=defaultdict(lambda:[0,0.0])
for (n,(_,s,e)) in zip(names[last+1:], rows[last+1:]):
    agg[n][0]+=1; agg[n][1]+=(e-s)/1e3
tot=sum(v[1] for v in agg.values())
print("# kernels behind the last tick of the run = one chd_wire_build; us")
for n,(k,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]): print(f"{n},{k},{t:.1f}")
print(f"sum,{sum(v[0] for v in agg.values())},{tot:.1f}")
span=(rows[-1][2]-rows[last+1][1])/1e3
print(f"span_first_to_last,,{span:.1f}")
try:
    j=json.load(open("$O/bench.json")); print("wire:", {k:j["wire"][k] for k in ("ms_per_build_all","GB_per_build_all","frac_of_hbm_peak")})
except Exception as ex: print("no bench json", ex)
PY
rm -rf $O/prof
