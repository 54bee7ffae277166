"""The headline world (config B) ticked through chd_tick_segments_begin / _end, two ticks in flight, inputs pre-staged in page-locked
memory: the loop tools/segp_timeline.sh traces.  usage: python tools/segp_loop.py [ticks] [sync]   (sync: chd_tick_segments instead)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import channeld_amd as A  # noqa: E402
from channeld_amd import synth  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 48
SYNC = len(sys.argv) > 2 and sys.argv[2] == "sync"
SERIAL = len(sys.argv) > 2 and sys.argv[2] == "serial"  # begin(t); end(t): one tick in flight
N, S = 100_000, 10_000
cfg = synth.load_config("spatial_static_benchmark.json")
sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE01, tick_ms=50))
ctl = A.StaticGrid2DSpatialController(device=0)
assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
world = A.SpatialWorld(ctl, N, S, flags=128)
world.set_pipelining(True)
world.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
world.add_subscribers(None, sw.sub_conn)
frames = []
for t in range(T):
    sw.step()
    px, pz, pq = world.host_array(N, np.float64), world.host_array(N, np.float64), world.host_array(S, synth.AOI_DTYPE)
    px[:], pz[:], pq[:] = sw.x, sw.z, sw.queries()
    frames.append((sw.now_ns(), px, pz, pq))
rows = []
if SYNC:
    for (now, px, pz, pq) in frames:
        a = time.perf_counter()
        res, seg = world.tick_segments(now, upd_x=px, upd_z=pz, queries=pq, pinned=True)
        rows.append((time.perf_counter() - a, 0.0, 0.0, res.n_records, 0, 0.0, 0.0))
else:
    now, px, pz, pq = frames[0]
    world.tick_segments_begin(now, upd_x=px, upd_z=pz, queries=pq)
    last = time.perf_counter()
    for t in range(T):
        enq = 0.0
        if SERIAL and t:
            now, px, pz, pq = frames[t]
            world.tick_segments_begin(now, upd_x=px, upd_z=pz, queries=pq)
        if not SERIAL and t + 1 < T:
            now, px, pz, pq = frames[t + 1]
            a = time.perf_counter()
            world.tick_segments_begin(now, upd_x=px, upd_z=pz, queries=pq)
            enq = time.perf_counter() - a
        a = time.perf_counter()
        res, seg, info = world.tick_segments_end()
        b = time.perf_counter()
        rows.append((b - last, b - a, enq, res.n_records, info["block_bytes"], info["wait_ms"], info["copy_ms"]))
        last = b
r = np.array([v[:3] for v in rows[8:]]) * 1e3
print(json.dumps(dict(form="sync" if SYNC else "pair", ticks=len(r), period_p50=float(np.percentile(r[:, 0], 50)), period_p99=float(np.percentile(r[:, 0], 99)),
                      sync_p50=float(np.percentile(r[:, 1], 50)), enqueue_p50=float(np.percentile(r[:, 2], 50)), wait_ms=float(np.median([v[5] for v in rows[8:]])), copy_ms=float(np.median([v[6] for v in rows[8:]])), block_bytes=[int(v[4]) for v in rows[8:10]],
                      msgs=[int(v[3]) for v in rows[8:12]])))
