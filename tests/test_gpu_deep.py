"""GPU parity of the EXACT update buffers (chd_world_cfg.history_depth; SURVEY row a12: ChannelData.updateMsgBuffer,
data.go:53-55,149-173,225-269) against the oracle's element-for-element buffers:

* arrival stamps at ENQUEUE time (channel.go:296-310 -> message.go:651 -> data.go:159-164), anywhere inside a tick interval,
* several updates of one channel between two ticks (update rounds), each its own Notify and its own buffer element,
* more than two senders per channel,
* windows far older than the 32-tick ring (a NO_ACCESS subscriber that regains access after 100+ ticks),
* the reference's eviction beyond MaxUpdateMsgBufferSize = 512 elements,

record multisets per connection, handovers, entity maps and subscription state every tick, history_overflow == 0 throughout;
one world also against the LITERAL linked-list tickData."""
import json

import numpy as np
import pytest

from channeld_amd import synth
from oracle import pyoracle as orc
from test_gpu_world import canon, compare_tick

pytestmark = pytest.mark.gpu

MS = 1_000_000


@pytest.fixture(scope="module")
def amd():
    import channeld_amd

    channeld_amd.load()
    return channeld_amd


def make_pair(amd, cfg_name, N, S, depth, flags, literal=False, seed=1, tick_ms=50, aoi_scale=1.0):
    cfg = synth.load_config(cfg_name)
    g = orc.grid_from_config(cfg)
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=tick_ms, aoi_scale=aoi_scale, outside_frac=0.01, locked_frac=0.02))
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    gw = amd.SpatialWorld(ctl, N, S, flags=flags, history_depth=depth)
    ow = orc.World(g, N, S, gw.capq, 20, 0, literal=literal)
    ow.set_threads(4)
    ow.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    gw.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    for s in range(S):
        ow.add_sub(s, int(sw.sub_conn[s]))
    gw.add_subscribers(None, sw.sub_conn)
    return cfg, sw, ctl, gw, ow


@pytest.mark.parametrize("flags", [1, 2, 1 | 64], ids=["conn-major", "cell-major", "conn-major-1w"])
def test_arrival_stamps_rounds_and_senders_match_the_oracle(amd, flags):
    """Every irregularity at once, on every emit form: per-update arrival stamps anywhere in (previous tick, this tick],
    a quarter of the ticks fully on the grid (the fast mask paths and the exact path hand over to each other), some entities
    updated twice per tick (rounds), three alternating senders, the spatial channels' own updates with stamps, irregular
    tick lengths."""
    N, S = 260, 24
    cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_4x4.json", N, S, 64, flags, seed=0xD11)
    rng = np.random.default_rng(11)
    now = 0
    total = deep_ticks = 0
    for k in range(40):
        sw.step()
        prev, now = now, now + int(rng.choice([20, 50, 50, 70])) * MS
        q = sw.queries()
        on_grid = k % 4 == 3
        # round 0: every entity; round 1: a second update of a few of them (a step further along)
        second = np.sort(rng.choice(N, 0 if on_grid else 30, replace=False)).astype(np.uint32)
        x1, z1 = sw.x.copy(), sw.z.copy()
        sw.step()
        idx = np.concatenate([np.arange(N, dtype=np.uint32), second])
        ux = np.concatenate([x1, sw.x[second]])
        uz = np.concatenate([z1, sw.z[second]])
        if len(second) == 0:
            sw.x, sw.z = x1, z1
        else:  # (entities without a second update stay where round 0 put them)
            keep = np.ones(N, dtype=bool)
            keep[second] = False
            sw.x[keep], sw.z[keep] = x1[keep], z1[keep]
        a0 = np.full(N, now, dtype=np.int64) if on_grid else rng.integers(prev + 1, now + 1, N).astype(np.int64)
        a1 = np.minimum(a0[second] + rng.integers(1, 5 * MS, len(second)), now).astype(np.int64)
        arr = np.concatenate([a0, a1])
        snd = np.concatenate([sw.sender, sw.sender[second]]).astype(np.uint32)
        if not on_grid:  # a third and fourth sender on some channels
            flip = rng.random(len(snd)) < 0.2
            snd = np.where(flip, rng.choice([901, 902, int(sw.sub_conn[0])], len(snd)), snd).astype(np.uint32)
        ncu = int(rng.integers(0, 4))
        cu = (0x10000 + rng.choice(16, ncu, replace=False)).astype(np.uint32)
        cus = rng.choice([1, 2, 903], ncu).astype(np.uint32)
        cua = np.sort(rng.integers(prev + 1, now + 1, ncu)).astype(np.int64)
        ow.tick(now, idx, ux, uz, snd, cu - 0x10000, cus, None, q, upd_arrival=arr, cu_arrival=cua)
        res = gw.tick(now, upd_idx=idx, upd_x=ux, upd_z=uz, upd_sender=snd, cell_upd_channel=cu, cell_upd_sender=cus,
                      queries=q, upd_arrival_ns=arr, cell_upd_arrival_ns=cua, upd_round_off=[0, N, N + len(second)])
        compare_tick(k, res, ow, S, check_pairs=range(0, S, 5), gw=gw)
        total += res.n_records
    cell, member = gw.entity_state()
    ocell, omember = ow.entity_state()
    to_id = lambda a: np.where(a == 0xFFFFFFFF, 0, a + 0x10000).astype(np.uint32)
    assert np.array_equal(cell, to_id(ocell)) and np.array_equal(member, to_id(omember))
    assert total > 10_000


def test_exact_buffers_against_the_literal_list_walk(amd):
    """... and the same kind of world against the oracle's LITERAL mode: every channel an orc_channel whose buffer holds the
    given arrival stamps, orc_tick_data = the linked-list walk of data.go:175-291."""
    N, S = 120, 10
    cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_2x2.json", N, S, 64, 1, literal=True, seed=0xD12)
    rng = np.random.default_rng(12)
    now = 0
    for k in range(24):
        sw.step()
        prev, now = now, now + int(rng.choice([30, 50, 80])) * MS
        arr = rng.integers(prev + 1, now + 1, N).astype(np.int64)
        q = sw.queries()
        ow.tick(now, None, sw.x, sw.z, None, None, None, None, q, upd_arrival=arr)
        res = gw.tick(now, upd_x=sw.x, upd_z=sw.z, queries=q, upd_arrival_ns=arr)
        compare_tick(k, res, ow, S, check_pairs=range(S), gw=gw)
    assert ow.literal_mismatch() == 0


def test_an_update_enqueued_before_a_window_edge_belongs_to_that_window(amd):
    """an update enqueued at 249 ms and handled by the tick at 260 ms is delivered in window [200, 250] AT that
    tick (data.go:247) — a batch stamped with its tick's time would deliver it one fan-out interval later."""
    cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_2x2.json", 4, 1, 32, 1, seed=3)
    x = np.array([-100.0, -150.0, 120.0, 100.0])
    z = np.array([-100.0, -120.0, 130.0, 100.0])
    chan = int(orc.channel_ids(orc.grid_from_config(cfg), x[:1], z[:1])[0])
    opts = dict(slot=0, channel=chan, fanout_interval_ms=50, fanout_delay_ms=0, skip_self_update_fanout=0)
    gw.set_sub_options(100 * MS, [opts])
    ow.set_sub_options(100 * MS, 0, chan, fanout_interval_ms=50, fanout_delay_ms=0, skip_self_update_fanout=0)
    seen = {}
    for now_ms, upd in ((100, None), (150, None), (160, None), (210, None), (260, 249), (310, None), (360, None)):
        now = now_ms * MS
        if upd is None:
            ow.tick(now)
            res = gw.tick(now)
        else:
            i, a = np.array([0], dtype=np.uint32), np.array([upd * MS], dtype=np.int64)
            ow.tick(now, i, x[:1], z[:1], None, upd_arrival=a)
            res = gw.tick(now, upd_idx=i, upd_x=x[:1], upd_z=z[:1], upd_arrival_ns=a)
        oc, och = ow.records()
        assert np.array_equal(canon(res.records["conn"], res.records["channel"]), canon(oc, och)), now_ms
        assert res.history_overflow == 0
        seen[now_ms] = [int(c) for c in res.records["channel"] if c >= 0x80000]
    assert seen[260] == [0x80000]   # window [200, 250] holds the update that was enqueued at 249
    assert seen[310] == [] and seen[360] == []


def test_a_subscriber_that_regains_access_after_100_ticks_catches_up_exactly(amd):
    """NO_ACCESS subscriptions are skipped but stay queued (data.go:194-197): when access comes back, tickData's catch-up
    loop walks every fan-out interval since and the buffers decide window by window — far beyond the 32-tick ring."""
    N, S = 150, 12
    cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_2x2.json", N, S, 256, 1 | 64, seed=0xD14)
    rng = np.random.default_rng(14)
    g = orc.grid_from_config(cfg)
    now = 0
    blocked = [1, 4, 7]
    for k in range(140):
        sw.step()
        prev, now = now, now + 50 * MS
        q = sw.queries()
        upd = np.sort(rng.choice(N, N // 2, replace=False)).astype(np.uint32)  # sparse updates: windows differ per entity
        arr = np.where(rng.random(len(upd)) < 0.5, now, rng.integers(prev + 1, now + 1, len(upd))).astype(np.int64)
        ow.tick(now, upd, sw.x[upd], sw.z[upd], None, None, None, None, q, upd_arrival=arr)
        res = gw.tick(now, upd_idx=upd, upd_x=sw.x[upd], upd_z=sw.z[upd], queries=q, upd_arrival_ns=arr, records_cap=1 << 22)
        compare_tick(k, res, ow, S, check_pairs=blocked, gw=gw)
        if k == 8 or k == 118:
            access = 0 if k == 8 else 1
            for s in blocked:
                ch = gw.subscriptions(s)[0]
                gw.set_sub_options(now, [dict(slot=s, channel=int(c), data_access=access) for c in ch])
                for c in ch:
                    ow.set_sub_options(now, s, int(c), data_access=access)
        if k == 119:
            assert res.n_records > 5_000  # the catch-up of three connections over 110 ticks


def test_the_reference_eviction_beyond_512_buffered_updates(amd):
    """MaxUpdateMsgBufferSize (data.go:53-55,165-171): beyond 512 elements the oldest goes once it is older than
    maxFanOutIntervalMs, one per push.  A subscriber that was blocked for 620 ticks sees, in the reference, only what the
    buffer still holds — and so here (history_depth 1024)."""
    N, S = 12, 3
    cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_2x2.json", N, S, 1024, 1, seed=0xD15, tick_ms=10)
    now = 0
    rng = np.random.default_rng(15)
    for k in range(660):
        sw.step()
        prev, now = now, now + 10 * MS
        q = sw.queries()
        arr = rng.integers(prev + 1, now + 1, N).astype(np.int64)
        ow.tick(now, None, sw.x, sw.z, None, None, None, None, q, upd_arrival=arr)
        res = gw.tick(now, upd_x=sw.x, upd_z=sw.z, queries=q, upd_arrival_ns=arr, records_cap=1 << 22)
        if k < 6 or k > 625 or k % 50 == 0:
            compare_tick(k, res, ow, S, check_pairs=range(S), gw=gw)
        else:
            assert res.history_overflow == 0 and res.overflow == 0 and res.n_records == len(ow.records()[0])
        if k == 5 or k == 630:
            access = 0 if k == 5 else 1
            ch = gw.subscriptions(2)[0]
            gw.set_sub_options(now, [dict(slot=2, channel=int(c), data_access=access) for c in ch])
            for c in ch:
                ow.set_sub_options(now, 2, int(c), data_access=access)


def test_max_fanout_interval_is_per_channel_beyond_512_elements(amd):
    """subscription.go:83-86 / data.go:165-171: the eviction beyond 512 elements tests the CHANNEL's own maxFanOutIntervalMs.
    tests/max_iv_scene.py: the channels of cell 0 (subscribers at 20 ms) take 120 updates per 10 ms tick while cell 3 has a
    100 ms subscriber; a connection of cell 0 that regains access must find the catch-up windows beyond the 512 newest elements
    EMPTY, as in the reference — under one world-wide maximum (round 4) it got a message for them.  Records every tick against
    the oracle, history_overflow 0 (history_depth 1024 holds what the reference holds)."""
    import max_iv_scene as sc

    cfg = synth.load_config("spatial_static_2x2.json")
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    gw = amd.SpatialWorld(ctl, sc.N, sc.S, flags=1, history_depth=1024)
    ow = orc.World(orc.grid_from_config(cfg), sc.N, sc.S, gw.capq, 20, 0)
    x, z = sc.positions(cfg)
    chan = (0x80000 + np.arange(sc.N)).astype(np.uint32)
    ow.spawn(np.arange(sc.N), chan, x, z, np.zeros(sc.N, dtype=np.uint32), np.full(sc.N, 900, dtype=np.uint32))
    gw.spawn(None, chan, x, z, np.zeros(sc.N, dtype=np.uint32), np.full(sc.N, 900, dtype=np.uint32))
    conns = np.array([50, 51, 52], dtype=np.uint32)
    gw.add_subscribers(None, conns)
    for s in range(sc.S):
        ow.add_sub(s, int(conns[s]))
    ss, st = gw.set_sub_options(0, sc.subscriptions())
    assert (st == 0).all() and (ss == 1).all()
    for o in sc.subscriptions():
        assert ow.set_sub_options(0, o["slot"], o["channel"], fanout_interval_ms=o["fanout_interval_ms"]) == 1
    regained = None
    for k in range(sc.TICKS):
        now, idx, ux, uz, arr, off = sc.tick_inputs(k, x, z)
        ow.tick(now, idx, ux, uz, None, None, None, None, None, upd_arrival=arr)
        res = gw.tick(now, upd_idx=idx, upd_x=ux, upd_z=uz, upd_arrival_ns=arr, upd_round_off=off, records_cap=1 << 20)
        compare_tick(k, res, ow, sc.S, check_pairs=range(sc.S), gw=gw)
        if k == sc.BLOCK_AT or k == sc.REGAIN_AT:
            access = 0 if k == sc.BLOCK_AT else 1
            gw.set_sub_options(now, [dict(slot=2, channel=0x10000, data_access=access)])
            ow.set_sub_options(now, 2, 0x10000, data_access=access)
        if k == sc.REGAIN_AT + 1:
            regained = int(((res.records["conn"] & 0x7FFFFFFF) == 52).sum())
    assert regained == 12 and ow.entity_buffer_len(0) == 512
    ctl.close()


def test_a_spatial_channels_own_update_is_delivered_once(amd):
    """A spatial channel's own update (its entity map changed) with an enqueue-time stamp, on the descriptor path with sub-tick
    offsets (flags 1 | 64): ONE sender, so the channel stays regular and the plan — not the element walk — decides from the
    channel's offsets, which are stored aligned to the tick of its last update.  Round 4 compared them unaligned: a cell updated at
    tick t - 1 and not at t was delivered once more at t to every window starting exactly at t - 1 (every 50 ms subscription on a
    50 ms world).  Two random cells per tick, some ticks none, 50 ms ticks: records equal the oracle's every tick."""
    N, S = 600, 32
    cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_4x4.json", N, S, 64, 1 | 64, seed=0xD31)
    rng = np.random.default_rng(31)
    now = 0
    own = 0
    for k in range(24):
        sw.step()
        prev, now = now, now + 50 * MS
        q = sw.queries()
        arr = rng.integers(prev + 1, now + 1, N).astype(np.int64)
        ncu = 0 if k % 5 == 4 else 2
        cu = (0x10000 + rng.choice(16, ncu, replace=False)).astype(np.uint32)
        cus = np.full(ncu, 5, dtype=np.uint32)
        cua = np.sort(rng.integers(prev + 1, now + 1, ncu)).astype(np.int64)
        ow.tick(now, None, sw.x, sw.z, None, cu - 0x10000, cus, None, q, upd_arrival=arr, cu_arrival=cua)
        res = gw.tick(now, upd_x=sw.x, upd_z=sw.z, cell_upd_channel=cu, cell_upd_sender=cus, queries=q, upd_arrival_ns=arr,
                      cell_upd_arrival_ns=cua, records_cap=1 << 22)
        compare_tick(k, res, ow, S, check_pairs=range(0, S, 6), gw=gw)
        own += int((res.records["channel"] < 0x80000).sum())
    assert own > 500  # (the spatial channels' own messages were there to be counted)
    ctl.close()


def test_cells_in_arrival_order_give_the_same_records(amd, monkeypatch):
    """CHD_SORT_ARRIVALS=1 (opt-in): every cell's entries in the order of the tick's arrival offsets, and a fan-out
    window that lies inside the tick's own arrivals copied as a RUN of the cell's column instead of tested per entity.  The order
    inside a cell is free (a connection's records are a multiset): records, handovers, subscription state equal the oracle's every
    tick — 20 ms subscriptions on a 50 ms world (the windows the run form is for), stamps on and off the tick grid, ties in the
    stamps, entities that skip ticks, cells beyond the 512-entry tile (2x2 grid: ~900 per cell) beside small ones (4x4)."""
    monkeypatch.setenv("CHD_SORT_ARRIVALS", "1")
    for cfg_name, N, S in (("spatial_static_4x4.json", 3000, 80), ("spatial_static_2x2.json", 3600, 40)):
        cfg, sw, ctl, gw, ow = make_pair(amd, cfg_name, N, S, 64, 1 | 64, seed=0xD21)
        rng = np.random.default_rng(21)
        now = 0
        total = 0
        for k in range(24):
            sw.step()
            prev, now = now, now + (50 * MS if k % 3 else 47 * MS + int(rng.integers(0, 6 * MS)))
            q = sw.queries()
            upd = np.sort(rng.choice(N, N - N // 10, replace=False)).astype(np.uint32) if k % 5 == 4 else np.arange(N, dtype=np.uint32)
            arr = rng.integers(prev + 1, now + 1, len(upd)).astype(np.int64)
            arr[rng.random(len(upd)) < 0.02] = now                                  # on the tick's own stamp
            arr[: len(upd) // 50] = prev + 1 + (now - prev) // 2                    # ties
            ow.tick(now, upd, sw.x[upd], sw.z[upd], None, None, None, None, q, upd_arrival=arr)
            res = gw.tick(now, upd_idx=upd, upd_x=sw.x[upd], upd_z=sw.z[upd], queries=q, upd_arrival_ns=arr, records_cap=1 << 23)
            compare_tick(k, res, ow, S, check_pairs=range(0, S, 7), gw=gw)
            total += res.n_records
        assert total > 100_000
        ctl.close()


def test_populous_cells_with_exact_buffers_stay_on_the_descriptor_path(amd):
    """max_entities / cells >= 1024 selects the cell-major emit form by default — but not on a world with exact update buffers where
    the descriptor path can run: only that path keeps the sub-tick arrival offsets; the cell-major form sends every window that cuts
    through a tick's arrivals to the element walk (config C with enqueue-time stamps: 155 ms per tick against 2.2,
    profiles/r06j_kernel_stats_c1m_aj*.csv).  A 2x2 world of 1100 entities per cell, no form asked for (flags = ONE_WAVE_EMIT only):
    the schedule says connection-major with arrival offsets, and the records equal the oracle's every tick — 20 and 50 ms windows
    through cells beyond the 512-entry tile, stamps anywhere inside the tick, entities that skip ticks.  Asking for the cell-major
    form still gives it (same records)."""
    N, S = 4400, 40
    for flags, want_cm in ((64, False), (64 | 2, True)):
        cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_2x2.json", N, S, 64, flags, seed=0xD41)
        sched = gw.stats()["schedule"]
        assert bool(sched & 8) == want_cm and bool(sched & 16) == (not want_cm), sched
        rng = np.random.default_rng(41)
        now = 0
        total = 0
        for k in range(14 if not want_cm else 6):
            sw.step()
            prev, now = now, now + (50 * MS if k % 3 else 47 * MS + int(rng.integers(0, 6 * MS)))
            q = sw.queries()
            upd = np.sort(rng.choice(N, N - N // 10, replace=False)).astype(np.uint32) if k % 5 == 4 else np.arange(N, dtype=np.uint32)
            arr = rng.integers(prev + 1, now + 1, len(upd)).astype(np.int64)
            ow.tick(now, upd, sw.x[upd], sw.z[upd], None, None, None, None, q, upd_arrival=arr)
            res = gw.tick(now, upd_idx=upd, upd_x=sw.x[upd], upd_z=sw.z[upd], queries=q, upd_arrival_ns=arr, records_cap=1 << 23)
            compare_tick(k, res, ow, S, check_pairs=range(0, S, 7), gw=gw)
            total += res.n_records
        assert total > 100_000
        ctl.close()


def test_a_short_buffer_says_what_it_dropped(amd):
    """history_depth smaller than what a window reaches back to: never silently short — history_overflow counts it."""
    N, S = 12, 2
    cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_2x2.json", N, S, 32, 1, seed=0xD16, tick_ms=10)
    now = 0
    flagged = False
    for k in range(80):
        sw.step()
        now += 10 * MS
        q = sw.queries()
        res = gw.tick(now, upd_x=sw.x, upd_z=sw.z, queries=q, upd_arrival_ns=np.full(N, now - 1, dtype=np.int64))
        if k == 5 or k == 70:
            ch = gw.subscriptions(1)[0]
            gw.set_sub_options(now, [dict(slot=1, channel=int(c), data_access=0 if k == 5 else 1) for c in ch])
        flagged = flagged or res.history_overflow != 0
        if k < 70:
            assert res.history_overflow == 0
    assert flagged


def test_arrival_stamps_need_history_depth(amd):
    cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_2x2.json", 8, 2, 0, 1, seed=4)
    with pytest.raises(amd.ChdError):
        gw.tick(50 * MS, upd_x=sw.x, upd_z=sw.z, upd_arrival_ns=np.full(8, 49 * MS, dtype=np.int64))
    ctl2 = amd.StaticGrid2DSpatialController()
    assert ctl2.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    with pytest.raises(amd.ChdError):  # below the ring's own depth
        amd.SpatialWorld(ctl2, 8, 2, history_depth=8)


def test_messages_delivered_one_by_one_through_the_update_batch(amd):
    """The reference's calling pattern: entity-channel updates arrive as single messages in any interleaving — an entity may send
    none, one or several between two ticks — and each is its own OnUpdate + Notify in its channel's order (channel.go:296-310,
    spatial.go:612); interest updates arrive per connection, a later one replacing an earlier one.  The host records them in
    an UpdateBatch (channeld_amd/engine.py) and the tick takes its layout (rounds, arrival stamps); the oracle is fed the raw
    message stream in arrival order."""
    N, S = 220, 18
    cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_4x4.json", N, S, 64, 1, seed=0xD17)
    rng = np.random.default_rng(17)
    x, z = sw.x.copy(), sw.z.copy()
    lo_x, hi_x = sw.offx, sw.offx + sw.W
    lo_z, hi_z = sw.offz, sw.offz + sw.H
    batch = amd.UpdateBatch(True)
    now = total = multi = 0
    for k in range(30):
        prev, now = now, now + int(rng.choice([20, 50, 50, 80])) * MS
        M = int(rng.integers(N // 2, 2 * N))
        who = rng.integers(0, N, M).astype(np.uint32)        # some entities several times, some not at all
        arr = np.sort(rng.integers(prev + 1, now + 1, M)).astype(np.int64)
        ux, uz = np.empty(M), np.empty(M)
        snd = np.empty(M, dtype=np.uint32)
        for m in range(M):
            i = int(who[m])
            if not sw.outside[i]:  # a step of up to a third of a cell: several handovers of one entity inside a tick happen
                x[i] = float(np.float32(min(max(x[i] + rng.uniform(-0.35, 0.35) * sw.gw, lo_x), np.nextafter(np.float32(hi_x), np.float32(-np.inf)))))
                z[i] = float(np.float32(min(max(z[i] + rng.uniform(-0.35, 0.35) * sw.gh, lo_z), np.nextafter(np.float32(hi_z), np.float32(-np.inf)))))
            ux[m], uz[m] = x[i], z[i]
            snd[m] = int(rng.choice([int(sw.sender[i]), 901, 902, int(sw.sub_conn[0])], p=[0.7, 0.1, 0.1, 0.1]))
            batch.on_update(i, ux[m], uz[m], int(snd[m]), int(arr[m]))
        multi += int(np.sum(np.bincount(who, minlength=N) > 1))
        ncu = int(rng.integers(0, 4))
        cu = (0x10000 + rng.integers(0, 16, ncu)).astype(np.uint32)
        cus = rng.choice([1, 2, 903], ncu).astype(np.uint32)
        cua = np.sort(rng.integers(prev + 1, now + 1, ncu)).astype(np.int64)
        for c, s_, a in zip(cu, cus, cua):
            batch.on_cell_update(int(c), int(s_), int(a))
        # interest: most connections once, a few twice (an older query from positions of the previous tick first), a few not at all
        sw.x, sw.z = x.copy(), z.copy()
        sw.heading = 2.0 * np.pi * rng.random(N)
        q_new = sw.queries()
        subs = [int(v) for v in rng.permutation(S) if rng.random() < 0.85]
        for s_ in subs[:4]:
            stale = q_new[s_].copy()
            for f in ("sph_cx", "cone_cx", "box_cx"):
                stale[f] = stale[f] + 0.5 * sw.gw if stale[f] != 0 else stale[f]
            batch.on_interest(s_, stale)
        for s_ in subs:
            batch.on_interest(s_, q_new[s_])
        kw = batch.tick_args()
        assert len(kw["upd_round_off"]) - 1 == int(np.bincount(who, minlength=N).max())
        ow.tick(now, who, ux, uz, snd, cu - 0x10000, cus, np.asarray(subs, dtype=np.uint32) if subs else None,
                q_new[subs] if subs else None, upd_arrival=arr, cu_arrival=cua)
        res = gw.tick(now, **kw)
        batch.clear()
        compare_tick(k, res, ow, S, check_pairs=range(0, S, 4), gw=gw)
        total += res.n_records
    cell, member = gw.entity_state()
    ocell, omember = ow.entity_state()
    to_id = lambda a: np.where(a == 0xFFFFFFFF, 0, a + 0x10000).astype(np.uint32)
    assert np.array_equal(cell, to_id(ocell)) and np.array_equal(member, to_id(omember))
    assert total > 5_000 and multi > 500


def test_stamps_on_window_edges_count_twice_and_fit_their_segment(amd):
    """an arrival stamp exactly on a window edge lies in TWO windows (both ends inclusive, data.go:236-241), so a
    subscription that catches up over more windows than the buffer holds elements writes up to two records per element — the
    segment the plan reserves must hold them (it used to reserve one per element: an out-of-bounds write into the next
    subscription's records).  20 ms ticks with tick-aligned stamps, a 10 ms subscriber blocked for 90 ticks, history_depth 32:
    on its return every buffered stamp is the upper edge of one of its windows and the lower edge of the next."""
    N, S = 40, 4
    cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_2x2.json", N, S, 32, 1, seed=0xD18, tick_ms=20)
    now = 0
    for k in range(110):
        sw.step()
        now += 20 * MS
        q = sw.queries() if k < 3 else None  # (the interest sets stay: a re-query would overwrite the 10 ms interval with the damped one)
        # (explicit stamps equal to the tick's own: the exact buffers are filled, the masks stay regular where they can)
        arr = np.full(N, now, dtype=np.int64)
        ow.tick(now, None, sw.x, sw.z, None, None, None, None, q, upd_arrival=arr)
        res = gw.tick(now, upd_x=sw.x, upd_z=sw.z, queries=q, upd_arrival_ns=arr, records_cap=1 << 22)
        if k == 4:
            for s in range(S):
                ch = gw.subscriptions(s)[0]
                gw.set_sub_options(now, [dict(slot=s, channel=int(c), fanout_interval_ms=10) for c in ch])
                for c in ch:
                    ow.set_sub_options(now, s, int(c), fanout_interval_ms=10)
        if k == 8 or k == 100:
            access = 0 if k == 8 else 1
            for s in (1, 2):
                ch = gw.subscriptions(s)[0]
                gw.set_sub_options(now, [dict(slot=s, channel=int(c), data_access=access) for c in ch])
                for c in ch:
                    ow.set_sub_options(now, s, int(c), data_access=access)
        if k < 12 or k >= 99:
            # (the short buffer has dropped what the reference still holds: history_overflow says so on the catch-up tick; the
            # records the device DOES write must be the oracle's for every connection that was never blocked)
            oc, och = ow.records()
            assert res.overflow == 0
            for s in (0, 3):
                conn = int(sw.sub_conn[s])
                mine = (res.records["conn"] & 0x7FFFFFFF) == conn
                theirs = (oc & 0x7FFFFFFF) == conn
                assert np.array_equal(canon(res.records["conn"][mine], res.records["channel"][mine]), canon(oc[theirs], och[theirs])), (k, s)
        if k == 101:
            # the blocked connections caught up over 180 windows with 32 buffered elements per channel: two records per element
            per_conn = {s: int(np.sum((res.records["conn"] & 0x7FFFFFFF) == int(sw.sub_conn[s]))) for s in range(S)}
            assert per_conn[1] > 40 * N // 4 and per_conn[2] > 40 * N // 4, per_conn


@pytest.mark.parametrize("flags", [1 | 32, 1 | 32 | 64], ids=["conn-major", "conn-major-1w"])
def test_merged_update_sets_of_exact_buffers_come_as_ranges(amd, flags):
    """VERDICT r3 #6 (f3 over the real buffer): WHICH buffered updates a message merges (data.go:225-269) on a world whose update
    buffers are the reference's (history_depth) and whose updates carry enqueue-time stamps, three senders, a subscriber that
    catches up over 40 ticks.  Records the tick ring answers carry a mask over the ring (bit 31 clear); every other record carries
    the RANGE of the channel's update numbers whose arrival lies in its window (bit 31 set: include/chd_spatial.h,
    chd_tick_out.record_masks) — equal, record for record, to what the oracle's buffer walk selects; history_overflow == 0."""
    N, S = 200, 16
    cfg, sw, ctl, gw, ow = make_pair(amd, "spatial_static_4x4.json", N, S, 128, flags, seed=0xD19)
    rng = np.random.default_rng(19)
    now = 0
    n_range = n_ring = 0
    key = lambda c, ch, m: np.sort((c.astype(np.uint64) << np.uint64(43)) ^ (ch.astype(np.uint64) << np.uint64(32)) ^ m.astype(np.uint64))
    for k in range(70):
        sw.step()
        prev, now = now, now + int(rng.choice([20, 50, 50])) * MS
        q = sw.queries() if k < 4 or k % 7 == 0 else None
        on_grid = k < 36 or k % 3 == 0  # (the first 36 ticks stamp their updates with the tick's own time: the ring answers)
        idx = np.sort(rng.choice(N, int(N * 0.8), replace=False)).astype(np.uint32)
        arr = np.full(len(idx), now, dtype=np.int64) if on_grid else rng.integers(prev + 1, now + 1, len(idx)).astype(np.int64)
        snd = np.where(rng.random(len(idx)) < 0.15, rng.choice([901, int(sw.sub_conn[1])], len(idx)), sw.sender[idx]).astype(np.uint32)
        ow.tick(now, idx, sw.x[idx], sw.z[idx], snd, None, None, None, q, upd_arrival=arr)
        res = gw.tick(now, upd_idx=idx, upd_x=sw.x[idx], upd_z=sw.z[idx], upd_sender=snd, queries=q, upd_arrival_ns=arr, records_cap=1 << 22)
        assert res.overflow == 0 and res.history_overflow == 0, (k, res.overflow, res.history_overflow)
        oc, och = ow.records()
        om, orr = ow.record_masks(), ow.record_ranges()
        assert res.n_records == len(oc)
        gm = res.record_masks
        is_range = (gm >> 31).astype(bool)
        # per record the device chose a form; the oracle offers both: same multiset of (conn, channel, word) either way
        order_g = np.lexsort((gm, res.records["channel"], res.records["conn"]))
        g_conn, g_chan, g_word, g_isr = res.records["conn"][order_g], res.records["channel"][order_g], gm[order_g], is_range[order_g]
        # match the oracle's records to the device's by (conn, channel, word in the device's form): build both candidate keys
        want_ring, want_range = key(oc, och, om), key(oc, och, orr)
        got_ring, got_range = key(g_conn[~g_isr], g_chan[~g_isr], g_word[~g_isr]), key(g_conn[g_isr], g_chan[g_isr], g_word[g_isr])
        assert np.isin(got_ring, want_ring).all(), f"tick {k}: a ring-form word the oracle does not have"
        assert np.isin(got_range, want_range).all(), f"tick {k}: a range-form word the oracle does not have"
        # ... and the records themselves are the oracle's
        assert np.array_equal(canon(res.records["conn"], res.records["channel"]), canon(oc, och)), k
        n_range += int(is_range.sum())
        n_ring += int((~is_range & ((res.records["conn"] >> 31) == 0)).sum())
        if k == 20 or k == 60:
            access = 0 if k == 20 else 1
            for s_ in (3, 9):
                ch = gw.subscriptions(s_)[0]
                gw.set_sub_options(now, [dict(slot=s_, channel=int(c), data_access=access) for c in ch])
                for c in ch:
                    ow.set_sub_options(now, s_, int(c), data_access=access)
    assert n_range > 5_000 and n_ring > 1_000, (n_range, n_ring)  # (a third sender inside the ring sends a cell to the buffers as well)
