#!/bin/bash
# The serial tick as a timeline (tools/rocpd_timeline.py) from a rocprofv3 kernel trace of the timed region.
# usage: bash tools/timeline.sh <tag> [bench args]
TAG=${1:-timeline}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace -d $O/prof -o kt -- python $R/bench.py --steps 64 --warmup 16 --only-timed "$@" > $O/bench.json 2> $O/prof.err
cd $R
python tools/rocpd_timeline.py $O/prof/kt_results.db k_ingest 16 | tee $O/tick_timeline.csv
python - <<PY
import sqlite3
c=sqlite3.connect("$O/prof/kt_results.db").cursor()
rows=c.execute("select start, end from kernels where name like '%k_fanout_emit_seg%' order by start").fetchall()
print("emit_seg durations (us) of the launches in order:", [round((e-s)/1e3,1) for s,e in rows][16:])
PY
rm -rf $O/prof
