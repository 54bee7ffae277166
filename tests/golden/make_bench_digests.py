#!/usr/bin/env python
"""tests/golden/make_bench_digests.py — the ORACLE's per-tick record digests of bench.py's default world.

bench.py's synthetic world is seeded, so the multiset of fan-out records of its k-th tick is a constant of the workload.
This script advances the CPU oracle (oracle/chd_world_oracle.c, window formulation, digest mode: every record folded into
{count, sum, xor of mix64(conn << 32 | channel)}) over the same frames and writes the list bench.py's latency phase compares
its device digests with (tests/golden/bench_digests_B.json).  CPU only; nothing here touches the HIP library.

    python tests/golden/make_bench_digests.py [--ticks 700] [--cross 40] [--threads N]

--cross K: for the first K ticks a SECOND oracle world takes the literal forward walk of every update buffer
(window_has_update, data.go:225-269 as written) beside the sorted newest-first walk the long run uses
(window_has_update_sorted: same selection while a channel's arrival stamps do not decrease) and every digest must agree.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from channeld_amd import synth  # noqa: E402  (host-side synthetic workload; imports without a GPU)
from oracle import pyoracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "bench_digests_B.json")
SEED, N, S, TICK_MS = 0xC0FFEE01, 100_000, 10_000, 50


def new_world(cfg, sw, threads, sorted_walk):
    g = orc.grid_from_config(cfg)
    capq = min(g.cols * g.rows, 256)
    ow = orc.World(g, N, S, capq, 20, 0, literal=False)
    ow.set_threads(threads)
    ow.set_digest_only(True)
    ow.set_sorted_walk(sorted_walk)
    ow.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    for s in range(S):
        ow.add_sub(s, int(sw.sub_conn[s]))
    return ow


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=700)
    ap.add_argument("--cross", type=int, default=40)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--arrival-jitter", action="store_true",
                    help="bench.py --arrival-jitter's world: every update stamped at its enqueue time (synth.ArrivalJitter), the oracle's "
                         "buffers hold those stamps (orc_world_tick_arrivals); default output tests/golden/bench_digests_B_jitter.json")
    ap.add_argument("--tick-jitter-us", type=int, default=0, help="with --arrival-jitter: tick times off the grid (…_jitter_offgrid.json)")
    ap.add_argument("--config", choices=["B", "C"], default="B",
                    help="C: BASELINE config C (1 M entities / 10 K subscribers, seed 0xC0FFEE02) — tests/test_gpu_fullsize.py's committed list "
                         "for the reference-stamp path at that size (bench_digests_C_jitter.json; ~70 s per tick on 8 cores)")
    args = ap.parse_args()
    global SEED, N
    if args.config == "C":
        SEED, N = 0xC0FFEE02, 1_000_000
    if args.out is None:
        args.out = OUT if not args.arrival_jitter else OUT.replace(".json", "_jitter_offgrid.json" if args.tick_jitter_us else "_jitter.json")
        if args.config == "C":
            args.out = args.out.replace("bench_digests_B", "bench_digests_C")
    orc.build()
    cfg = synth.load_config("spatial_static_benchmark.json")
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, SEED, tick_ms=TICK_MS, aoi_scale=1.0))
    fast = new_world(cfg, sw, args.threads, True)
    slow = new_world(cfg, sw, args.threads, False) if args.cross > 0 else None
    ticks = {}
    t0 = time.perf_counter()
    aj = synth.ArrivalJitter(SEED, N, args.tick_jitter_us) if args.arrival_jitter else None
    for k in range(1, args.ticks + 1):
        sw.step()
        q = sw.queries()
        now, arr = aj.next(sw.now_ns()) if aj is not None else (sw.now_ns(), None)
        fast.tick(now, None, sw.x, sw.z, None, None, None, None, q, upd_arrival=arr)
        (cnt, sm, xr, _), _ = fast.digest()
        if slow is not None and k <= args.cross:
            slow.tick(now, None, sw.x, sw.z, None, None, None, None, q, upd_arrival=arr)
            (c2, s2, x2, _), _ = slow.digest()
            if (cnt, sm, xr) != (c2, s2, x2):
                raise SystemExit(f"tick {k}: sorted walk {(cnt, sm, xr)} != forward walk {(c2, s2, x2)}")
            if k == args.cross:
                slow = None
        assert not fast.unsorted()
        ticks[str(k)] = [cnt, sm, xr]
        if k % 20 == 0 or args.config == "C":
            print(f"tick {k}: {cnt} records, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
    with open(args.out, "w") as f:
        json.dump({"what": "per-tick digests {count, sum, xor of mix64(conn << 32 | channel)} of the fan-out records of bench.py's default world "
                           f"(spatial_static_benchmark.json, {N} entities / {S} subs, seed {SEED:#x}, {TICK_MS} ms ticks), tick k = the k-th tick "
                           + ("" if not args.arrival_jitter else f"ARRIVAL STAMPS AT ENQUEUE TIME (bench.py --arrival-jitter; synth.ArrivalJitter, tick jitter {args.tick_jitter_us} us), ")
                           + "since the world began, as the CPU ORACLE computes them (oracle/chd_world_oracle.c, window formulation, digest mode; "
                           "tests/golden/make_bench_digests.py — the device never ran for this file).  bench.py compares chd_tick_digest of its "
                           "latency-phase ticks with this list; tests/test_bench_digests.py recomputes a few entries",
                   "generator": "tests/golden/make_bench_digests.py", "source": "oracle",
                   "cross_checked_forward_walk_ticks": args.cross,
                   "ticks": ticks}, f)
    print(f"wrote {len(ticks)} ticks to {args.out} in {time.perf_counter() - t0:.0f} s")


if __name__ == "__main__":
    main()
