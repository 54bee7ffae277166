#!/bin/bash
# One rank's tick of a multi-GPU configuration on one GPU (bench.py --shard-shape D|E [--arrival-jitter]): the stage times, and — from a
# rocprofv3 kernel trace of the same command — the kernels every rank runs over the WHOLE world's channels (the part of a sharded tick
# that does not divide by the number of ranks).   usage: bash tools/shard_shape_prof.sh <tag> <D|E> [bench args]
TAG=$1; SHAPE=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
python $R/bench.py --shard-shape $SHAPE "$@" > $O/shape.json 2> $O/shape.err || tail -5 $O/shape.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d $O/prof -o kt -- python $R/bench.py --shard-shape $SHAPE "$@" > $O/shape_prof.json 2> $O/prof.err
cd $R
python - <<PY > $O/replicated.json
import json, sqlite3
d = json.load(open("$O/shape.json"))
c = sqlite3.connect("$O/prof/kt_results.db").cursor()
rows = {}
for name, s, e in c.execute("select name, start, end from kernels"):
    n = name.split("(")[0].replace("void ", "")
    a = rows.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
rows = {k: (v[0], v[1] / v[0]) for k, v in rows.items()}
# kernels whose grid is the WORLD's channel count on every rank (k_ingest_by_channel's grid is the rank's own entity slots: it pulls)
rep = {k: v for k, v in rows.items() if k.startswith("k_log_push")}
us = d["us"]
whole = sum(v[1] for v in rep.values())
tick_serial = d["rank0_tick_us_with_upload"]
tick_overl = d["rank0_tick_us_upload_beside_previous_fanout"]
out = {"shape": d["shard_shape"], "arrival_stamps": d["config"]["arrival_stamps"], "rank0": d["config"]["rank0"], "stage_us": us,
       "whole_world_kernels_us": {k: {"calls": v[0], "avg_us": round(v[1], 1)} for k, v in rep.items()},
       "whole_world_kernels_us_sum": round(whole, 1), "h2d_us": us["h2d"], "h2d_bytes": d["h2d_bytes"],
       "replicated_share_upload_serial": round((whole + us["h2d"]) / tick_serial, 4),
       "replicated_share_upload_beside_previous_fanout": round(whole / tick_overl, 4),
       "fanout_us_alone_vs_beside_upload": [us["fanout"], us.get("fanout_beside_next_upload")],
       "note": "kernel averages are over all emulated ranks' launches (every rank runs them over the whole world's channels: the same work on each)"}
print(json.dumps(out, indent=1))
PY
cat $O/replicated.json
rm -rf $O/prof
