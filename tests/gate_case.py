"""One process of tests/test_gpu_gates.py: config B on a CHD_WORLD_OVERLAP_INTEREST | CHD_WORLD_GATED_OVERLAP world, groups of
back-to-back chd_tick_device calls, every group's last tick digested against the oracle's committed list
(tests/golden/bench_digests_B.json).  Prints one JSON line: which schedule the world ended up with, whether a gate timed out,
the digests' verdict and the wall time of the ticks.  The environment (GPU_MAX_HW_QUEUES, CHD_TEST_DROP_GATE_RAISE) and the
arguments (--contexts: other gated worlds alive in the process; --busy-s: just keep the GPU busy for that long) are the case."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=3)
    ap.add_argument("--per", type=int, default=5)
    ap.add_argument("--contexts", type=int, default=0)
    ap.add_argument("--busy-s", type=float, default=0.0)
    a = ap.parse_args()
    import channeld_amd as amd
    from channeld_amd import _lib, synth

    amd.load()
    N, S, seed = 100_000, 10_000, 0xC0FFEE01
    cfg = synth.load_config("spatial_static_benchmark.json")
    flags = _lib.WORLD_OVERLAP_INTEREST | _lib.WORLD_GATED_OVERLAP

    def world(n, s, recs):
        ctl = amd.StaticGrid2DSpatialController()
        assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
        return ctl, amd.SpatialWorld(ctl, n, s, max_records=recs, flags=flags)

    others = [world(2000, 200, 2_000_000) for _ in range(a.contexts)]  # (their streams take hardware queues too)
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed))
    ctl, w = world(N, S, 200_000_000)
    w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    T = a.groups * a.per
    xs = np.empty((T, N)); zs = np.empty((T, N)); qs = np.empty((T, S), dtype=synth.AOI_DTYPE); now = np.empty(T, dtype=np.int64)
    for t in range(T):
        sw.step()
        xs[t], zs[t], qs[t], now[t] = sw.x, sw.z, sw.queries(), sw.now_ns()
    dx, dz, dq = w.device_array(xs), w.device_array(zs), w.device_array(qs)
    w.sync()
    if a.busy_s > 0:  # the "other process": ticks of the same world over and over (now_ns only has to not go backwards)
        t_end = time.time() + a.busy_s
        k = 0
        while time.time() < t_end:
            for t in range(T):
                w.tick_device(int(now[T - 1]) + 50_000_000 * (k * T + t + 1), n_updates=N, d_upd_x=dx.at(t * N * 8), d_upd_z=dz.at(t * N * 8),
                              n_queries=S, d_queries=dq.at(t * S * 128))
            w.sync()
            k += 1
        print(json.dumps({"busy_rounds": k}))
        return
    with open(os.path.join(ROOT, "tests", "golden", "bench_digests_B.json")) as f:
        golden = json.load(f)["ticks"]
    sched0 = w.stats()["schedule"]
    bad, ovf, slowest = [], [], 0.0
    for gi in range(a.groups):
        t0 = time.time()
        for t in range(gi * a.per, (gi + 1) * a.per):
            w.tick_device(int(now[t]), n_updates=N, d_upd_x=dx.at(t * N * 8), d_upd_z=dz.at(t * N * 8), n_queries=S, d_queries=dq.at(t * S * 128))
        (cnt, dsum, dxor, _), _ = w.digest(per_connection=False)
        slowest = max(slowest, time.time() - t0)
        if [cnt, dsum, dxor] != golden[str((gi + 1) * a.per)]:
            bad.append((gi + 1) * a.per)
        ovf += [h["overflow"] for h in w.history(a.per)]  # (per tick: chd_tick_fetch would only see the group's last one)
    st = w.stats()
    print(json.dumps({"schedule_at_start": sched0, "schedule": st["schedule"], "gate_timeouts": st["gate_timeouts"], "bad_ticks": bad,
                      "overflow": ovf, "slowest_group_s": round(slowest, 3), "contexts": a.contexts,
                      "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}))
    for c, ww in others:
        c.close()
    ctl.close()


if __name__ == "__main__":
    main()
