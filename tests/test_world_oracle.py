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


def test_digest_mode_equals_the_digest_of_the_stored_records():
    """Full-size GPU parity folds records into digests instead of storing them (tests/test_gpu_fullsize.py):
    the oracle's digest mode must be the digest of exactly the records its storing mode returns."""
    cfg = synth.load_config("spatial_static_2x2.json")
    g = orc.grid_from_config(cfg)
    N, S = 800, 128
    capq = 4
    worlds = [orc.World(g, N, S, capq, 20, 0, literal=False) for _ in range(2)]
    worlds[1].set_digest_only(True)
    worlds[0].set_threads(1)
    worlds[1].set_threads(3)
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xD16E57))
    for w in worlds:
        w.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
        for s in range(S):
            w.add_sub(s, int(sw.sub_conn[s]))

    def mix64(k):
        with np.errstate(over="ignore"):
            k = (k ^ (k >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            k = (k ^ (k >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return k ^ (k >> np.uint64(31))

    total = 0
    for k in range(9):
        sw.step()
        q = sw.queries()
        for w in worlds:
            w.tick(sw.now_ns() + (40_000_000 if k == 5 else 0), None, sw.x, sw.z, None, None, None, None, q)
        conn, chan = worlds[0].records()
        masks = worlds[0].record_masks()
        h = mix64((conn.astype(np.uint64) << np.uint64(32)) | chan.astype(np.uint64))
        (cnt, dsum, dxor, dmask), per_conn = worlds[1].digest()
        assert cnt == len(conn)
        assert dsum == int(np.add.reduce(h, dtype=np.uint64)) and dxor == int(np.bitwise_xor.reduce(h)) if len(h) else dsum == 0
        with np.errstate(over="ignore"):
            assert dmask == int(np.add.reduce(mix64(h + masks.astype(np.uint64)), dtype=np.uint64))
        want = np.zeros(S, dtype=np.uint64)
        with np.errstate(over="ignore"):
            np.add.at(want, (conn & 0x7FFFFFFF).astype(np.int64) - 1000, h)
        assert np.array_equal(per_conn, want)
        total += cnt
    assert total > 10000


def test_sorted_newest_first_walk_equals_the_forward_walk_and_the_literal_list():
    """tests/golden/make_bench_digests.py runs the oracle with the update buffers walked newest-first and an early exit
    (window_has_update_sorted) so that hundreds of full-size ticks with 512-deep buffers finish in minutes.  That walk selects the
    same elements as the reference's forward walk (data.go:225-269) while a channel's arrival stamps do not decrease: here against
    the forward walk AND the literal linked-list tickData, with mid-tick arrival stamps, sparse updates, record masks and digests;
    and a world whose stamps DO go backwards falls back to the forward walk by itself."""
    cfg = synth.load_config("spatial_static_4x4.json")
    g = orc.grid_from_config(cfg)
    N, S = 240, 30
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0x50F7, tick_ms=50, outside_frac=0.01, locked_frac=0.02))
    worlds = [orc.World(g, N, S, 16, 20, 0, literal=lit) for lit in (False, False, True)]
    worlds[0].set_sorted_walk(True)
    for w in worlds:
        w.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
        for s in range(S):
            w.add_sub(s, int(sw.sub_conn[s]))
    rng = np.random.default_rng(5)
    now = total = 0
    for k in range(40):
        sw.step()
        prev, now = now, now + int(rng.choice([20, 50, 50, 120])) * 1_000_000
        idx = np.sort(rng.choice(N, int(N * rng.choice([1.0, 0.6])), replace=False)).astype(np.uint32)
        arr = np.where(rng.random(len(idx)) < 0.3, now, rng.integers(prev + 1, now + 1, len(idx))).astype(np.int64)
        q = sw.queries()
        recs = []
        for w in worlds:
            w.tick(now, idx, sw.x[idx], sw.z[idx], None, None, None, None, q, upd_arrival=arr)
            recs.append(canon(*w.records()))
        assert np.array_equal(recs[0], recs[1]) and np.array_equal(recs[0], recs[2]), k
        c0, ch0 = worlds[0].records()
        c1, ch1 = worlds[1].records()
        o0, o1 = np.argsort((c0.astype(np.uint64) << np.uint64(32)) | ch0, kind="stable"), np.argsort((c1.astype(np.uint64) << np.uint64(32)) | ch1, kind="stable")
        assert np.array_equal(np.sort(worlds[0].record_masks()[o0]), np.sort(worlds[1].record_masks()[o1]))
        total += len(recs[0])
    assert total > 20_000 and not worlds[0].unsorted() and worlds[2].literal_mismatch() == 0
    # a stamp that goes backwards on one channel: the sorted walk stands down (and still equals the forward walk)
    sw.step()
    now += 50_000_000
    arr = np.full(N, now, dtype=np.int64)
    arr[7] = now - 400_000_000
    for w in worlds[:2]:
        w.tick(now, None, sw.x, sw.z, None, None, None, None, None, upd_arrival=arr)
    assert worlds[0].unsorted()
    assert np.array_equal(canon(*worlds[0].records()), canon(*worlds[1].records()))


def test_max_fanout_interval_is_kept_per_channel():
    """subscription.go:83-86 + data.go:165-171 in the tick model (chd_world_oracle.c: cmax / emax): the scene of
    tests/max_iv_scene.py.  A channel whose own subscribers all use 20 ms evicts beyond 512 elements by 20 ms although another
    cell has a 100 ms subscriber — its buffer stays at 512 elements — and a subscriber that regains access finds the older
    catch-up windows empty."""
    import max_iv_scene as sc
    from channeld_amd import synth

    cfg = synth.load_config("spatial_static_2x2.json")
    ow = orc.World(orc.grid_from_config(cfg), sc.N, sc.S, 4, 20, 0)
    x, z = sc.positions(cfg)
    ow.spawn(np.arange(sc.N), 0x80000 + np.arange(sc.N), x, z, np.zeros(sc.N, dtype=np.uint32), np.full(sc.N, 900, dtype=np.uint32))
    for s in range(sc.S):
        ow.add_sub(s, 50 + s)
    for o in sc.subscriptions():
        assert ow.set_sub_options(0, o["slot"], o["channel"], fanout_interval_ms=o["fanout_interval_ms"]) == 1
    assert ow.cell_max_interval(0) == 20 and ow.cell_max_interval(3) == 100
    regained = None
    for k in range(sc.TICKS):
        now, idx, ux, uz, arr, off = sc.tick_inputs(k, x, z)
        ow.tick(now, idx, ux, uz, None, None, None, None, None, upd_arrival=arr)
        if k == sc.BLOCK_AT or k == sc.REGAIN_AT:
            ow.set_sub_options(now, 2, 0x10000, data_access=0 if k == sc.BLOCK_AT else 1)
        if k == sc.REGAIN_AT + 1:
            conn, chan = ow.records()
            regained = int(((conn & 0x7FFFFFFF) == 52).sum())
    assert ow.entity_max_interval(0) == 20 and ow.entity_max_interval(5) == 100
    assert ow.entity_buffer_len(0) == 512          # 20 ms rule: one out per push once beyond 512
    assert ow.entity_buffer_len(5) == sc.TICKS      # one update per tick, never beyond 512
    # connection 2 catches up 14 windows of 20 ms over four channels (+ the cell's own); only the windows the 512 newest elements
    # (~43 ms) reach into carry a message: 3 per channel.  Under a world-wide maximum of 100 ms it would have been 5 or 6.
    assert regained == 4 * 3, regained
