#!/bin/bash
# One steady-state tick of the chd_tick_segments_begin / _end loop (tools/segp_loop.py) as a timeline: every kernel and memory copy
# between two successive k_ingest launches, start offset and duration in us (rocprofv3 kernel + memory-copy trace).
# usage: bash tools/segp_timeline.sh <tag> [sync]
TAG=${1:-segp_tl}; MODE=${2:-pair}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
python $R/tools/segp_loop.py 64 $MODE > $O/loop_plain.json 2> $O/loop_plain.err; cat $O/loop_plain.json
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof -o tl -- python $R/tools/segp_loop.py 40 $MODE > $O/loop.json 2> $O/prof.err
cd $R
python - <<PY > $O/timeline.txt
import sqlite3
c=sqlite3.connect("$O/prof/tl_results.db").cursor()
tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
k=c.execute("select name, start, end from kernels order by start").fetchall()
mc_name=[t for t in tabs if t=="memory_copies"]
m=[]
if mc_name:
    cols=[r[1] for r in c.execute("pragma table_info(memory_copies)").fetchall()]
    sz="size" if "size" in cols else None
    m=c.execute(f"select name, start, end{', size' if sz else ''} from memory_copies order by start").fetchall()
ev=[(s,e,n.split("(")[0].replace("void ","")[:60],0) for (n,s,e) in k]+[(r[1],r[2],"COPY "+str(r[0])[:40],(r[3] if len(r)>3 else 0)) for r in m]
ev.sort()
ing=[i for i,x in enumerate(ev) if x[2].startswith("k_ingest")]
a,b=ing[-6],ing[-5]
t0=ev[a][0]
print(f"# one tick: {len(ev[a:b])} operations, period {(ev[b][0]-t0)/1e3:.1f} us")
print("# start_us, dur_us, op, bytes")
# everything that STARTS inside the period (copies of the previous tick included)
for s,e,n,z in ev[a:b]: print(f"{(s-t0)/1e3:8.1f} {(e-s)/1e3:8.1f}  {n}  {z or ''}")
per=[(ev[ing[i+1]][0]-ev[ing[i]][0])/1e3 for i in range(len(ing)-12,len(ing)-1)]
print("# periods of the last ticks (us):", [round(p,1) for p in per])
PY
cat $O/timeline.txt | head -120
rm -rf $O/prof
