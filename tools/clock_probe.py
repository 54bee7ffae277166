#!/usr/bin/env python
"""Samples the shader clock (rocm-smi) while config B ticks run back to back for ~8 s."""
import json, os, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import channeld_amd as A
from channeld_amd import synth

cfg = synth.load_config("spatial_static_benchmark.json")
N, S, T = 100_000, 10_000, 40
sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE01))
ctl = A.StaticGrid2DSpatialController(device=0)
assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
world = A.SpatialWorld(ctl, N, S)
world.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
world.add_subscribers(None, sw.sub_conn)
xs = np.empty((T, N)); zs = np.empty((T, N)); qs = np.empty((T, S), dtype=synth.AOI_DTYPE)
for t in range(T):
    sw.step(); xs[t], zs[t], qs[t] = sw.x, sw.z, sw.queries()
d_x, d_z, d_q = world.device_array(xs), world.device_array(zs), world.device_array(qs)
samples, stop = [], False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        s = [l.split(":")[-1].strip() for l in out.splitlines() if "sclk" in l or "Power (W)" in l]
        samples.append(s)
        time.sleep(0.5)
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 8.0:
    for _ in range(200):
        t = k % T
        # ping-pong through the trajectory so that positions stay continuous
        if (k // T) % 2: t = T - 1 - t
        world.tick_device((k + 1) * 50_000_000, n_updates=N, d_upd_x=d_x.at(t * N * 8), d_upd_z=d_z.at(t * N * 8),
                          n_queries=S, d_queries=d_q.at(t * S * 128))
        k += 1
    world.sync()
el = time.perf_counter() - t0
stop = True; th.join()
print(f"{k} ticks in {el:.2f} s = {1e3 * el / k:.3f} ms/tick")
for s in samples: print(s)
