#!/usr/bin/env python
"""tests/golden/make_shard_golden.py — the single world's oracle results for `bench.py --gpus N --config D|E --verify K --verify-golden FILE`.

The sharded bench verifies its first K ticks against the SINGLE-world oracle; for BASELINE config E at its stated size (8 x 8 world,
1 M entities / 100 K subscribers, AOI x 0.5: 9 G records in the first fan-out) that oracle takes 80-100 s per tick on a few cores, which
no test suite has.  This script runs it ONCE, over exactly the frames bench.py generates (channeld_amd.dist.bench_world + synth, same
seed), through bench.py's own SingleWorldChecker, and commits per tick: {count, sum, xor} of all records, the fold of every connection's
own digest (channeld_amd.dist.conn_fold), handover / locked-abort / unsub counts.  CPU only; nothing here touches the HIP library.

    python tests/golden/make_shard_golden.py --config E --ticks 3      -> tests/golden/bench_digests_E.json
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["D", "E"], default="E")
    ap.add_argument("--ticks", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import bench  # (the checker class and nothing else: bench.py imports without a GPU)
    from channeld_amd import dist as cdist
    from channeld_amd import synth

    world = {"D": 4, "E": 8}[a.config]
    args = types.SimpleNamespace(config=a.config, entities=None, subs=None, aoi_scale=None, tick_ms=50)
    cfg, N, S, n_max, s_max, aoi, label, scaling = cdist.bench_world(args, world)
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, cdist.BENCH_SEED, tick_ms=50, aoi_scale=aoi))
    chk = bench.SingleWorldChecker()
    capq = min(int(cfg["GridCols"]) * int(cfg["GridRows"]), 256)
    chk.setup(cfg, N, S, capq, sw)
    ticks = {}
    t0 = time.perf_counter()
    slots = np.arange(S, dtype=np.uint64)
    for k in range(1, a.ticks + 1):
        sw.step()
        r = chk.step(sw.now_ns(), sw.x, sw.z, sw.queries())
        ticks[str(k)] = {"digest": [int(v) for v in r["digest"]], "conn_fold": cdist.conn_fold(slots, r["conn"][:S]), "handovers": int(r["handovers"]),
                         "locked": int(r["locked"]), "unsubs": int(r["unsubs"])}
        print(f"tick {k}: {r['digest'][0]} records, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
    out = a.out or os.path.join(ROOT, "tests", "golden", f"bench_digests_{a.config}.json")
    with open(out, "w") as f:
        json.dump({"what": f"the single world's oracle results of bench.py --config {a.config} ({label}), tick k = the k-th tick since the world began: "
                           "{count, sum, xor of mix64(conn << 32 | channel)} over all fan-out records, the fold of every connection's own digest "
                           "(channeld_amd.dist.conn_fold), handover / locked-abort / unsub counts — oracle/chd_world_oracle.c through bench.py's "
                           "SingleWorldChecker; the device never ran for this file",
                   "generator": "tests/golden/make_shard_golden.py", "source": "oracle",
                   "world": {"entities": int(N), "subs": int(S), "grid": [int(cfg["GridCols"]), int(cfg["GridRows"])], "aoi_scale": aoi, "seed": int(cdist.BENCH_SEED)},
                   "ticks": ticks}, f, indent=1)
    print(f"wrote {len(ticks)} ticks to {out} in {time.perf_counter() - t0:.0f} s")


if __name__ == "__main__":
    main()
