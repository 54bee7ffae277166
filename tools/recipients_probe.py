#!/usr/bin/env python
"""Timing of the recipient planners on config B (SURVEY 8f-2 / 8f-4 decision parts)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import channeld_amd as A
from channeld_amd import synth
cfg = synth.load_config("spatial_static_benchmark.json")
N, S = 100_000, 10_000
sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE01))
ctl = A.StaticGrid2DSpatialController(device=0)
assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
w = A.SpatialWorld(ctl, N, S, flags=4)
w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender); w.add_subscribers(None, sw.sub_conn)
for _ in range(4):
    sw.step(); res = w.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=sw.queries(), want_records=False)
t0 = time.perf_counter(); off, conn, kind = w.handover_recipients(len(res.handovers)); t1 = time.perf_counter()
print(f"handover recipients: {len(res.handovers)} handovers -> {len(conn)} recipients ({np.bincount(kind, minlength=3).tolist()} by kind), fetch {1e3*(t1-t0):.2f} ms")
rng = np.random.default_rng(1)
for n in (64, 1024, 8192):
    ch = (0x10000 + rng.integers(0, 225, n)).astype(np.uint32)
    z = np.zeros(n, dtype=np.uint32)
    w.adjacent_recipients(ch, np.full(n, 64, dtype=np.uint32), z, z)
    t0 = time.perf_counter(); off, conns = w.adjacent_recipients(ch, np.full(n, 64, dtype=np.uint32), z, z); t1 = time.perf_counter()
    print(f"adjacent broadcast: {n} requests -> {len(conns)} recipients in {1e3*(t1-t0):.2f} ms through the C-ABI (host buffers)")
