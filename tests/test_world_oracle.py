"""CPU-only: the tick-pipeline oracle's window formulation (literal=0, also the
cpu_baseline loop nest) must equal the literal tickData list walk (literal=1) on
every entity and spatial channel, tick after tick."""
import numpy as np
import pytest

from channeld_amd import synth
from oracle import pyoracle as orc


def canon(conn, chan):
    a = (conn.astype(np.uint64) << np.uint64(32)) | chan.astype(np.uint64)
    return np.sort(a)


def run_pair(cfg_name, N, S, ticks, tick_ms, seed, capq=64, aoi_scale=1.0, drop_every=0, pauses=None):
    cfg = synth.load_config(cfg_name)
    g = orc.grid_from_config(cfg)
    spec = synth.WorldSpec(cfg, N, S, seed, tick_ms=tick_ms, aoi_scale=aoi_scale, outside_frac=0.01, locked_frac=0.02)
    sw = synth.SynthWorld(spec)
    worlds = [orc.World(g, N, S, capq, 20, 0, literal=lit) for lit in (False, True)]
    for w in worlds:
        w.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
        for s in range(S):
            w.add_sub(s, int(sw.sub_conn[s]))
    total = 0
    late_ns = 0
    for k in range(ticks):
        sw.step()
        late_ns += (pauses or {}).get(k, 0) * 1_000_000
        now = sw.now_ns() + late_ns
        q = sw.queries()
        # only part of the entities update on some ticks (sparse idx path)
        if drop_every and k % drop_every == 1:
            idx = np.arange(0, N, 3, dtype=np.uint32)
        else:
            idx = np.arange(N, dtype=np.uint32)
        outs = []
        for w in worlds:
            w.tick(now, idx, sw.x[idx], sw.z[idx], None, None, None, None, q)
            outs.append((canon(*w.records()), [a.copy() for a in w.handovers()], [a.copy() for a in w.unsubs()],
                         w.query_status().copy()))
        (r0, h0, u0, s0), (r1, h1, u1, s1) = outs
        assert np.array_equal(r0, r1), f"tick {k}: fan-out records differ"
        for a, b in zip(h0, h1):
            assert np.array_equal(a, b)
        for a, b in zip(u0, u1):
            assert np.array_equal(a, b)
        assert np.array_equal(s0, s1)
        total += len(r0)
    assert worlds[1].literal_mismatch() == 0
    return total


def test_literal_equals_window_2x2():
    total = run_pair("spatial_static_2x2.json", 300, 40, 14, 50, 0xC0FFEE00, capq=4)
    assert total > 1000


def test_literal_equals_window_benchmark_grid_sparse_updates():
    total = run_pair("spatial_static_benchmark.json", 400, 30, 10, 33, 0xC0FFEE01, capq=128, drop_every=3)
    assert total > 1000


def test_literal_equals_window_irregular_tick():
    # 7 ms ticks: several windows per tick for none, one tick per several windows for others
    total = run_pair("spatial_static_4x4.json", 200, 24, 20, 7, 0xC0FFEE03, capq=16, aoi_scale=0.5)
    assert total > 500


def test_literal_equals_window_after_long_pauses():
    # ticks that arrive 90 s / 7.3 s late: more than 4096 catch-up windows for the 20 ms interval class
    # (the window formulation once capped its list there and spread the catch-up over two ticks)
    total = run_pair("spatial_static_2x2.json", 300, 32, 18, 50, 0xC0FFEE09, capq=4, pauses={5: 90_000, 11: 7_300, 12: 61})
    assert total > 5000
