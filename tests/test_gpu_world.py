"""GPU parity of the batched tick (chd_tick through the C-ABI) against the
tick-pipeline oracle: handover records, entity maps, interest sets with their
fan-out state, unsub/new-sub lists and the multiset of fan-out records per
connection, tick after tick on the same seeded synthetic input."""
import json

import numpy as np
import pytest

from channeld_amd import synth
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


EMIT_FLAGS = 0


@pytest.fixture(autouse=True, params=["cell-major", "conn-major", "conn-major-1w", "conn-major-1w-pipe"])
def emit_mode(request):
    """Every world test runs against every form of the fan-out emit kernel: cell-major, connection-major with four
    waves per connection, and connection-major with one wave per connection (the pipelined kernel + its deferred
    launch: what config B's 10K connections take), and the latter with CHD_WORLD_PIPELINE_TICKS (stages on the second stream;
    these tests fetch after every tick, so ticks never overlap here — tests/test_gpu_fullsize.py issues them back to back)."""
    global EMIT_FLAGS
    # CHD_WORLD_CONN_MAJOR_EMIT / CHD_WORLD_CELL_MAJOR_EMIT / | CHD_WORLD_ONE_WAVE_EMIT
    EMIT_FLAGS = {"conn-major": 1, "cell-major": 2, "conn-major-1w": 1 | 64, "conn-major-1w-pipe": 1 | 64 | 128}[request.param]
    yield request.param


def make(amd, cfg, N, S, capq=0, max_records=0, extra_flags=0, damping=None):
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False, **({"Damping": damping} if damping else {})) is None
    w = amd.SpatialWorld(ctl, N, S, max_interest_cells=capq, max_records=max_records, flags=EMIT_FLAGS | extra_flags)
    return ctl, w


def canon(conn, chan):
    return np.sort((conn.astype(np.uint64) << np.uint64(32)) | chan.astype(np.uint64))


def compare_tick(k, res, ow, S, id_start=0x10000, check_pairs=None, gw=None):
    # handovers (order is not defined in the reference: compare as sets keyed by entity)
    ent, src, dst, ssrc, sdst = ow.handovers()
    got = np.sort(res.handovers, order=["entity", "src", "dst"])  # (an entity can hand over twice in a tick: update rounds)
    o = np.lexsort((dst, src, ent))
    assert len(got) == len(ent), f"tick {k}: {len(got)} handovers vs oracle {len(ent)}"
    assert np.array_equal(got["entity"], ent[o]) and np.array_equal(got["src"], src[o]) and np.array_equal(got["dst"], dst[o])
    assert np.array_equal(got["src_server"], ssrc[o]) and np.array_equal(got["dst_server"], sdst[o])
    assert res.n_locked_aborts == ow.locked_aborts()
    # interest updates
    assert np.array_equal(res.query_status, ow.query_status()), f"tick {k}: query status"
    us, uc = ow.unsubs()
    assert np.array_equal(canon(res.unsub_sub, res.unsub_channel), canon(us, uc)), f"tick {k}: unsubs"
    # fan-out records
    oc, och = ow.records()
    assert res.n_records == len(oc), f"tick {k}: {res.n_records} records vs oracle {len(oc)}"
    assert np.array_equal(canon(res.records["conn"], res.records["channel"]), canon(oc, och)), f"tick {k}: records"
    assert res.overflow == 0 and res.history_overflow == 0
    if res.record_masks is not None:
        # CHD_WORLD_UPDATE_MASKS: (connection, channel, merged-updates mask) triples as multisets
        om = ow.record_masks()
        key = lambda c, ch, m: np.sort(np.rec.fromarrays([c, ch, m], names="c,ch,m"), order=["c", "ch", "m"])
        got, want = key(res.records["conn"], res.records["channel"], res.record_masks), key(oc, och, om)
        assert np.array_equal(got, want), f"tick {k}: merged-update masks"
    # records are grouped per connection slot
    assert int(res.conn_rec_off[S]) == res.n_records
    if check_pairs is not None:
        new_pairs = []
        for s in check_pairs:
            gch, giv, glast, ghf, gnew = gw.subscriptions(s)
            wch, wiv, wlast, whf, wnew = ow.pairs(s)
            assert np.array_equal(gch, wch) and np.array_equal(giv, wiv), f"tick {k} sub {s}: interest set"
            assert np.array_equal(glast, wlast) and np.array_equal(ghf, whf), f"tick {k} sub {s}: fan-out state"
            assert np.array_equal(gnew, wnew), f"tick {k} sub {s}: is_new"


def run_world(amd, cfg_name, N, S, ticks, seed, tick_ms=50, capq=0, aoi_scale=1.0, sparse=False, check_subs=8, pauses=None, extra_flags=0, literal=False, damping=None):
    cfg = synth.load_config(cfg_name)
    g = orc.grid_from_config(cfg)
    spec = synth.WorldSpec(cfg, N, S, seed, tick_ms=tick_ms, aoi_scale=aoi_scale, outside_frac=0.01, locked_frac=0.02)
    sw = synth.SynthWorld(spec)
    ctl, gw = make(amd, cfg, N, S, capq, extra_flags=extra_flags, damping=damping)
    ow = orc.World(g, N, S, gw.capq, 20, 0, literal=literal)
    if damping:
        ow.set_damping(damping)
    ow.set_threads(4)
    ow.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    gw.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    for s in range(S):
        ow.add_sub(s, int(sw.sub_conn[s]))
    gw.add_subscribers(None, sw.sub_conn)
    rng = np.random.default_rng(seed & 0xFFFF)
    total = 0
    n_ho = 0
    late_ns = 0  # accumulated lateness of the tick clock (pauses: {tick: extra ms})
    for k in range(ticks):
        sw.step()
        late_ns += (pauses or {}).get(k, 0) * 1_000_000
        now = sw.now_ns() + late_ns
        q = sw.queries()
        if sparse and k % 3 == 1:
            idx = np.sort(rng.choice(N, N // 2, replace=False)).astype(np.uint32)
            qsub = np.sort(rng.choice(S, S // 2, replace=False)).astype(np.uint32)
        else:
            idx = np.arange(N, dtype=np.uint32)
            qsub = np.arange(S, dtype=np.uint32)
        ow.tick(now, idx, sw.x[idx], sw.z[idx], None, None, None, qsub, q[qsub])
        res = gw.tick(now, upd_idx=idx, upd_x=sw.x[idx], upd_z=sw.z[idx], query_sub=qsub, queries=q[qsub],
                      records_cap=max(1 << 20, 4 * len(ow.records()[0])))
        subs = rng.choice(S, min(check_subs, S), replace=False)
        compare_tick(k, res, ow, S, check_pairs=subs, gw=gw)
        # new-sub list == the oracle's is_new pairs of the re-queried subscribers
        want = []
        for s in qsub:
            ch, iv, _, _, nw = ow.pairs(int(s))
            want += [(int(s), int(c), int(i)) for c, i, n in zip(ch, iv, nw) if n]
        got = sorted(zip(res.newsub_sub.tolist(), res.newsub_channel.tolist(), res.newsub_interval_ms.tolist()))
        assert got == sorted(want), f"tick {k}: new-sub list"
        total += res.n_records
        n_ho += len(res.handovers)
    # entity maps at the end
    cell, member = gw.entity_state()
    ocell, omember = ow.entity_state()
    to_id = lambda a: np.where(a == 0xFFFFFFFF, 0, a + 0x10000).astype(np.uint32)
    assert np.array_equal(cell, to_id(ocell)) and np.array_equal(member, to_id(omember))
    if literal:
        assert ow.literal_mismatch() == 0  # every channel of a cell evolved its subscribers' state alike
    return total, n_ho


@pytest.fixture(scope="module")
def amd():
    import channeld_amd

    channeld_amd.load()
    return channeld_amd


def test_world_against_the_literal_list_walk(amd):
    """The GPU against the oracle's LITERAL mode: every spatial and entity channel is a real fanOutQueue walked by the
    restated tickData (move-to-back + revisit, data.go:175-291) — not the window formulation the other world tests
    (and the CPU baseline) use.  Irregular and slow ticks so that catch-up windows and 100 ms subscribers occur."""
    total, n_ho = run_world(amd, "spatial_static_2x2.json", 600, 96, 24, 0xC0FFEE2A, literal=True, pauses={5: 35, 9: 130, 10: 7, 15: 260})
    assert total > 50000 and n_ho > 0
    total, n_ho = run_world(amd, "spatial_static_benchmark.json", 1500, 64, 10, 0xC0FFEE2B, literal=True, sparse=True)
    assert total > 5000


def test_world_strict_reference_flat_interval(amd):
    """SURVEY 9.6's strict-reference mode: one flat fan-out interval for every subscription (the ENTITY channels' 50 ms of
    config/channel_settings_ue.json) instead of the distance-damped one — a one-entry damping table — and a custom
    three-step table; both against the oracle with the same table."""
    total, n_ho = run_world(amd, "spatial_static_benchmark.json", 3000, 160, 12, 0xC0FFEE3A, damping=[(0xFFFFFFFF, 50)])
    assert total > 50000 and n_ho > 0
    total, n_ho = run_world(amd, "spatial_static_4x4.json", 1500, 96, 12, 0xC0FFEE3B, tick_ms=33, damping=[(0, 33), (2, 66), (5, 200)])
    assert total > 20000


def test_world_config_a_2x2(amd):
    # BASELINE config A: spatial_static_2x2.json, 1K entities / 256 subscribers
    total, n_ho = run_world(amd, "spatial_static_2x2.json", 1000, 256, 30, 0xC0FFEE00)
    assert total > 100000 and n_ho > 0


def test_world_benchmark_grid(amd):
    total, n_ho = run_world(amd, "spatial_static_benchmark.json", 6000, 400, 24, 0xC0FFEE01)
    assert total > 200000 and n_ho > 20


def test_world_sparse_updates_and_requeries(amd):
    total, n_ho = run_world(amd, "spatial_static_benchmark.json", 3000, 200, 24, 0xC0FFEE05, tick_ms=33, sparse=True)
    assert total > 50000


def test_world_irregular_ticks_4x4(amd):
    # 7 ms ticks against 20/50/100 ms intervals; derived config D grid
    total, n_ho = run_world(amd, "spatial_static_4x4.json", 2000, 150, 40, 0xC0FFEE03, tick_ms=7, aoi_scale=0.5)
    assert total > 50000


def test_world_8x8_small_aoi(amd):
    total, n_ho = run_world(amd, "spatial_static_8x8.json", 4000, 300, 20, 0xC0FFEE04, tick_ms=20, aoi_scale=0.4)
    assert total > 50000


def test_world_40x40_grid_1600_cells(amd):
    # > 1024 cells: the index build's cross-workgroup total scan, 25 bitmap words per connection
    total, n_ho = run_world(amd, "spatial_static_40x40.json", 5000, 300, 12, 0xC0FFEE08, tick_ms=40, aoi_scale=1.0)
    assert total > 50000 and n_ho > 0


def test_world_slow_ticks_catch_up(amd):
    # 170 ms ticks: several catch-up windows per tick for every interval class
    total, n_ho = run_world(amd, "spatial_static_benchmark.json", 2000, 100, 16, 0xC0FFEE06, tick_ms=170)
    assert total > 50000


def test_world_interest_on_second_stream(amd):
    # CHD_WORLD_OVERLAP_INTEREST: interest updates beside ingest + index build, joined before the fan-out plan
    from channeld_amd import _lib

    total, n_ho = run_world(amd, "spatial_static_benchmark.json", 3000, 200, 14, 0xC0FFEE0A, sparse=True,
                            extra_flags=_lib.WORLD_OVERLAP_INTEREST)
    assert total > 50000 and n_ho > 0


def test_world_update_masks(amd, emit_mode):
    # per message, the buffered updates it merges (window x SkipSelfUpdateFanOut selection of
    # data.go:225-269) - slow ticks so that 20 / 50 / 100 ms subscriptions merge different sets; sparse updates so
    # that histories differ per entity; connection-major emit only
    if emit_mode == "cell-major":
        pytest.skip("update masks are written by the connection-major emit")
    from channeld_amd import _lib

    for cfg_name, tick_ms, seed in (("spatial_static_benchmark.json", 70, 0xC0FFEE0B), ("spatial_static_4x4.json", 33, 0xC0FFEE0C)):
        total, n_ho = run_world(amd, cfg_name, 2500, 150, 14, seed, tick_ms=tick_ms, sparse=True,
                                extra_flags=_lib.WORLD_UPDATE_MASKS)
        assert total > 20000


def test_world_long_pauses_fold_empty_windows(amd):
    # ticks 5, 11 and 12 arrive 90 s, 7.3 s and 61 ms late: thousands of empty fan-out windows lie between the
    # buffered stamps; the kernels fold them (empty_windows), the oracle walks every one of them
    total, n_ho = run_world(amd, "spatial_static_2x2.json", 300, 32, 18, 0xC0FFEE09, pauses={5: 90_000, 11: 7_300, 12: 61})
    assert total > 5000


def test_world_despawn_lock_and_remove_subscriber(amd):
    cfg = synth.load_config("spatial_static_2x2.json")
    g = orc.grid_from_config(cfg)
    N, S = 64, 8
    ctl, gw = make(amd, cfg, N, S)
    ow = orc.World(g, N, S, gw.capq, 20, 0)
    x = np.linspace(-1900, 1900, N)
    z = np.linspace(-1900, 1900, N)[::-1].copy()
    chan = (0x80000 + np.arange(N)).astype(np.uint32)
    flags = np.zeros(N, dtype=np.uint32)
    sender = np.full(N, 5, dtype=np.uint32)
    ow.spawn(np.arange(N), chan, x, z, flags, sender)
    gw.spawn(None, chan, x, z, flags, sender)
    conns = (1000 + np.arange(S)).astype(np.uint32)
    for s in range(S):
        ow.add_sub(s, int(conns[s]))
    gw.add_subscribers(None, conns)
    qs = [orc.QueryBuilder(sphere=(float(x[s * 8]), float(z[s * 8]), 900.0)) for s in range(S)]
    import channeld_amd as A

    gq = [A.SpatialInterestQuery(SphereAOI=A.SphereAOI(Center=A.SpatialInfo(X=float(x[s * 8]), Z=float(z[s * 8])), Radius=900.0)) for s in range(S)]
    t = 0
    for k in range(12):
        t += 40_000_000
        x = x + 130.0
        if k == 3:  # lock a few entities right before they cross the x = 0 border
            for i in (30, 31, 33):
                ow.set_flags(i, 1)
            gw.set_entity_flags([30, 31, 33], [1, 1, 1])
        if k == 5:
            ow.despawn(10)
            gw.despawn([10])
            ow.remove_sub(2)
            gw.remove_subscribers([2])
        if k == 7:
            for i in (30, 31, 33):
                ow.set_flags(i, 0)
            gw.set_entity_flags([30, 31, 33], [0, 0, 0])
        queries = qs if k % 4 == 0 else None
        gqueries = gq if k % 4 == 0 else None
        ow.tick(t, None, x, z, None, None, None, None, queries)
        res = gw.tick(t, upd_x=x, upd_z=z, queries=gqueries)
        compare_tick(k, res, ow, S, check_pairs=range(S), gw=gw)
    cell, member = gw.entity_state()
    ocell, omember = ow.entity_state()
    to_id = lambda a: np.where(a == 0xFFFFFFFF, 0, a + 0x10000).astype(np.uint32)
    live = np.ones(N, dtype=bool)
    live[10] = False
    assert np.array_equal(cell[live], to_id(ocell)[live]) and np.array_equal(member[live], to_id(omember)[live])


def mask_variants(emit_mode):
    """world flags to run a scenario with: plain, and with update masks where the emit form writes them"""
    from channeld_amd import _lib

    return (0, _lib.WORLD_UPDATE_MASKS) if emit_mode.startswith("conn-major") else (0,)


def test_cell_channel_updates_and_self_skip(amd, emit_mode):
    for extra in mask_variants(emit_mode):
        cell_channel_updates_and_self_skip(amd, extra)


def cell_channel_updates_and_self_skip(amd, extra_flags):
    # spatial-channel data updates sent by a subscriber itself are not fanned back to it
    cfg = synth.load_config("spatial_static_2x2.json")
    g = orc.grid_from_config(cfg)
    N, S = 16, 3
    ctl, gw = make(amd, cfg, N, S, extra_flags=extra_flags)
    ow = orc.World(g, N, S, gw.capq, 20, 0)
    x = np.full(N, 500.0)
    z = np.full(N, 500.0)
    chan = (0x80000 + np.arange(N)).astype(np.uint32)
    conns = np.array([7, 8, 9], dtype=np.uint32)
    sender = np.full(N, 7, dtype=np.uint32)  # connection 7 sends every entity update
    ow.spawn(np.arange(N), chan, x, z, np.zeros(N, dtype=np.uint32), sender)
    gw.spawn(None, chan, x, z, None, sender)
    for s in range(S):
        ow.add_sub(s, int(conns[s]))
    gw.add_subscribers(None, conns)
    import channeld_amd as A

    qs = [orc.QueryBuilder(sphere=(500.0, 500.0, 100.0)) for _ in range(S)]
    gq = [A.SpatialInterestQuery(SphereAOI=A.SphereAOI(Center=A.SpatialInfo(X=500.0, Z=500.0), Radius=100.0)) for _ in range(S)]
    t = 0
    seen_delta = {7: 0, 8: 0, 9: 0}
    for k in range(10):
        t += 20_000_000
        cu_cell = np.array([3], dtype=np.uint32)   # cell index 3 = channel 0x10003 holds (500,500)
        cu_sender = np.array([8], dtype=np.uint32)
        ow.tick(t, None, x, z, None, cu_cell, cu_sender, None, qs if k == 0 else None)
        res = gw.tick(t, upd_x=x, upd_z=z, cell_upd_channel=cu_cell + 0x10000, cell_upd_sender=cu_sender,
                      queries=gq if k == 0 else None)
        compare_tick(k, res, ow, S)
        for r in res.records:
            if not (r["conn"] & 0x80000000):
                seen_delta[int(r["conn"])] += 1
    # 7 never receives entity deltas (its own), 8 never receives the cell channel's deltas
    assert seen_delta[9] > seen_delta[7] > 0 and seen_delta[9] > seen_delta[8] > 0


def test_changing_senders_keep_per_update_self_skip(amd, emit_mode):
    for extra in mask_variants(emit_mode):
        changing_senders_keep_per_update_self_skip(amd, extra)


def changing_senders_keep_per_update_self_skip(amd, extra_flags):
    """SkipSelfUpdateFanOut compares each BUFFERED update's senderConnId with the subscriber
    (data.go:242-245).  Entities are updated by connection 7 on some ticks and 8 on others
    (an ownership change); subscribers 7 and 8 must each miss exactly their own updates, also
    while the other sender's older updates are still inside their slower fan-out windows."""
    cfg = synth.load_config("spatial_static_2x2.json")
    g = orc.grid_from_config(cfg)
    N, S = 40, 3
    ctl, gw = make(amd, cfg, N, S, extra_flags=extra_flags)
    ow = orc.World(g, N, S, gw.capq, 20, 0)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1900, 1900, N)
    z = rng.uniform(-1900, 1900, N)
    chan = (0x80000 + np.arange(N)).astype(np.uint32)
    conns = np.array([7, 8, 9], dtype=np.uint32)
    sender0 = np.full(N, 7, dtype=np.uint32)
    ow.spawn(np.arange(N), chan, x, z, np.zeros(N, dtype=np.uint32), sender0)
    gw.spawn(None, chan, x, z, None, sender0)
    for s in range(S):
        ow.add_sub(s, int(conns[s]))
    gw.add_subscribers(None, conns)
    import channeld_amd as A

    # a sphere covering the whole 2x2 world: cells at distance 1-2 get 50/100 ms intervals, so windows span several ticks
    qs = [orc.QueryBuilder(sphere=(-1000.0, -1000.0, 3000.0)) for _ in range(S)]
    gq = [A.SpatialInterestQuery(SphereAOI=A.SphereAOI(Center=A.SpatialInfo(X=-1000.0, Z=-1000.0), Radius=3000.0)) for _ in range(S)]
    t = 0
    deltas = {7: 0, 8: 0, 9: 0}
    for k in range(24):
        t += 20_000_000
        snd = np.where((np.arange(N) + k // 3) % 2 == 0, 7, 8).astype(np.uint32)  # ownership flips every 3 ticks
        idx = np.arange(N, dtype=np.uint32)
        ow.tick(t, idx, x, z, snd, None, None, None, qs if k == 0 else None)
        res = gw.tick(t, upd_idx=idx, upd_x=x, upd_z=z, upd_sender=snd, queries=gq if k == 0 else None)
        compare_tick(k, res, ow, S)
        for r in res.records:
            if not (r["conn"] & 0x80000000):
                deltas[int(r["conn"])] += 1
    assert deltas[9] > deltas[7] > 0 and deltas[9] > deltas[8] > 0


def test_third_sender_inside_the_history_is_flagged(amd):
    cfg = synth.load_config("spatial_static_2x2.json")
    N, S = 4, 1
    ctl, gw = make(amd, cfg, N, S)
    x = np.full(N, 500.0)
    z = np.full(N, 500.0)
    gw.spawn(None, (0x80000 + np.arange(N)).astype(np.uint32), x, z, None, np.full(N, 5, dtype=np.uint32))
    gw.add_subscribers(None, np.array([9], dtype=np.uint32))
    flagged = 0
    for k, snd in enumerate((5, 6, 7, 5)):
        res = gw.tick((k + 1) * 20_000_000, upd_x=x, upd_z=z, upd_sender=np.full(N, snd, dtype=np.uint32))
        flagged += res.history_overflow
    assert flagged > 0


def test_subscription_options_drive_every_fan_out_branch(amd, emit_mode):
    """chd_subs_set_options = SubscribeToChannel with explicit ChannelSubscriptionOptions (subscription.go:34-102), against
    the oracle's restatement, on a moving world: connections that are themselves senders of entity updates with
    SkipSelfUpdateFanOut = false (they get their own updates back) and true (they do not); subscriptions put to
    NO_ACCESS for some ticks (skipped but still queued, data.go:194-197) and back to READ (their catch-up windows come
    out at once); server-like connections that subscribe explicitly with WRITE access, SkipFirstFanOut, positive and
    negative FanOutDelayMs and their own interval; interest updates in between that keep, re-damp and drop such
    subscriptions.  Results of SubscribeToChannel (should-send) and the stored options are compared as well."""
    from channeld_amd import _lib

    cfg = synth.load_config("spatial_static_4x4.json")
    g = orc.grid_from_config(cfg)
    N, S = 1200, 48
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0x5AB0, tick_ms=33, outside_frac=0.01, locked_frac=0.01))
    ctl, gw = make(amd, cfg, N, S)
    ow = orc.World(g, N, S, gw.capq, 20, 0, literal=False)
    conn = sw.sub_conn.copy()
    # connections 0..7 send the updates of a third of the entities (entity i -> connection i % 8 for i % 3 == 0)
    sender = sw.sender.copy()
    mine = np.arange(N) % 3 == 0
    sender[mine] = conn[np.arange(N)[mine] % 8]
    ow.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sender)
    gw.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sender)
    for s in range(S):
        ow.add_sub(s, int(conn[s]))
    gw.add_subscribers(None, conn)
    rng = np.random.default_rng(77)
    clients = np.arange(0, 40, dtype=np.uint32)   # follow entities, send interest updates
    servers = np.arange(40, 48)                   # subscribe explicitly, never query
    ncell = g.cols * g.rows

    def apply(now, opts):
        ss, st = gw.set_sub_options(now, opts)
        for o, a, b in zip(opts, ss, st):
            r = ow.set_sub_options(now, o["slot"], o["channel"], o.get("data_access"), o.get("fanout_interval_ms"), o.get("fanout_delay_ms"),
                                   o.get("skip_self_update_fanout"), o.get("skip_first_fanout"))
            assert (int(b), int(a)) == ((0, r) if r >= 0 else ((_lib.E_INVAL, 0) if r == -1 else (_lib.E_CAPACITY, 0))), (o, a, b, r)

    total = n_full_self = 0
    for k in range(30):
        sw.step()
        now = sw.now_ns() + (45_000_000 if k >= 17 else 0)  # one long pause
        q = sw.queries()
        # ---- explicit SUB_TO_CHANNEL messages before this tick's updates
        opts = []
        if k == 0:
            for s in servers:  # a server subscribes to "its" cells: WRITE, own interval, delays of both signs, skip-first for some
                for c in range(int(s) % 4, ncell, 4):
                    opts.append(dict(slot=int(s), channel=0x10000 + c, data_access=2, fanout_interval_ms=int(rng.choice([20, 33, 70])),
                                     fanout_delay_ms=int(rng.choice([-40, 0, 30])), skip_first_fanout=int(s) % 2,
                                     skip_self_update_fanout=int(rng.integers(0, 2))))
        if k == 2:
            for s in range(0, 8, 2):  # sender connections that want their own updates back, on every channel they hold
                ch, *_ = ow.pairs(s)
                opts += [dict(slot=s, channel=int(c), skip_self_update_fanout=0) for c in ch]
        if k in (5, 6):
            for s in (9, 10, 11, 41):  # lose access ...
                ch, *_ = ow.pairs(s)
                opts += [dict(slot=s, channel=int(c), data_access=0) for c in ch[::2]]
        if k == 12:
            for s in (9, 10, 11, 41):  # ... and get it back: the skipped windows are caught up in one tick
                ch, *_ = ow.pairs(s)
                opts += [dict(slot=s, channel=int(c), data_access=1) for c in ch]
        if k == 8:
            opts += [dict(slot=3, channel=0x10000 + c, fanout_interval_ms=50, skip_first_fanout=1, fanout_delay_ms=-100) for c in (0, 5, 10, 15)]
            opts += [dict(slot=20, channel=0x10000 + 7, data_access=2), dict(slot=20, channel=0x10000 + 7, data_access=2), dict(slot=20, channel=0x10000 + 7, data_access=1)]
        if opts:
            apply(now, [o for o in opts if o])
        qsub = clients if k % 4 != 3 else clients[::2]
        idx = np.arange(N, dtype=np.uint32)
        ow.tick(now, idx, sw.x, sw.z, sender, None, None, qsub, q[qsub])
        res = gw.tick(now, upd_idx=idx, upd_x=sw.x, upd_z=sw.z, upd_sender=sender, query_sub=qsub, queries=q[qsub],
                      records_cap=max(1 << 20, 4 * len(ow.records()[0])))
        compare_tick(k, res, ow, S, check_pairs=list(range(0, S, 5)) + [3, 9, 20, 41], gw=gw)
        for s in (0, 2, 3, 9, 20, 41, 44):
            ga, gs = gw.sub_options(s)
            wa, ws = ow.pair_options(s)
            assert np.array_equal(ga, wa) and np.array_equal(gs, ws), f"tick {k} slot {s}: stored options"
        total += res.n_records
        # a connection with SkipSelfUpdateFanOut = false receives updates of entities it sends
        own = np.isin(res.records["channel"], sw.chan_id[mine]) & ((res.records["conn"] & 0x7FFFFFFF) == conn[0]) & ((res.records["conn"] >> 31) == 0)
        n_full_self += int((own & np.isin(res.records["channel"], sw.chan_id[(np.arange(N) % 3 == 0) & (np.arange(N) % 8 == 0)])).sum())
    assert total > 100000
    assert n_full_self > 0  # the !SkipSelfUpdateFanOut branch produced records


def test_handover_groups_move_together_and_locks_abort(amd, emit_mode):
    """chd_world_set_entity_groups (entity.go handover groups): members that ride with a leader — same positions, some
    of them without updates of their own — change cell with it; a locked member aborts the whole group's handover
    until it is unlocked; lone entities are untouched.  Handover records, locked aborts, entity maps (every tick) and
    fan-out records against the oracle's sequential restatement."""
    cfg = synth.load_config("spatial_static_4x4.json")
    g = orc.grid_from_config(cfg)
    N, S = 900, 40
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0x6A0B, tick_ms=50, outside_frac=0.0, locked_frac=0.0))
    ctl, gw = make(amd, cfg, N, S)
    ow = orc.World(g, N, S, gw.capq, 20, 0, literal=False)
    # groups of three: leader 3k, riders 3k+1 (updated with the leader's position) and 3k+2 (never updated) for k < 120
    K = 120
    leader = np.arange(K) * 3
    group = np.zeros(N, dtype=np.uint32)
    for k in range(K):
        group[3 * k: 3 * k + 3] = 1000 + k
    x0, z0 = sw.x.copy(), sw.z.copy()
    for k in range(K):
        x0[3 * k + 1: 3 * k + 3] = x0[3 * k]
        z0[3 * k + 1: 3 * k + 3] = z0[3 * k]
    flags = np.zeros(N, dtype=np.uint32)
    flags[3 * np.arange(0, K, 10) + 2] = 1  # every tenth group has a locked (and never updated) rider
    ow.spawn(np.arange(N), sw.chan_id, x0, z0, flags, sw.sender)
    gw.spawn(None, sw.chan_id, x0, z0, flags, sw.sender)
    gidx = np.nonzero(group)[0].astype(np.uint32)
    gw.set_entity_groups(gidx, group[gidx])
    for i in gidx:
        ow.set_group(int(i), int(group[i]))
    for s in range(S):
        ow.add_sub(s, int(sw.sub_conn[s]))
    gw.add_subscribers(None, sw.sub_conn)
    upd = np.array([i for i in range(N) if not (i < 3 * K and i % 3 == 2)], dtype=np.uint32)  # riders 3k+2 never update
    rng = np.random.default_rng(9)
    total_ho = total_lock = moved_riders = 0
    for k in range(22):
        sw.step()
        # bigger steps for the leaders so that groups cross often; riders 3k+1 copy the leader
        jump = rng.random(K) < 0.25
        sw.x[leader] = np.where(jump, np.float64(np.float32(sw.offx + rng.random(K) * sw.W * 0.999)), sw.x[leader])
        sw.x[leader + 1], sw.z[leader + 1] = sw.x[leader], sw.z[leader]
        if k == 10:  # unlock: those groups move again from now on
            idx = (3 * np.arange(0, K, 10) + 2).astype(np.uint32)
            gw.set_entity_flags(idx, np.zeros(len(idx), dtype=np.uint32))
            for i in idx:
                ow.set_flags(int(i), 0)
        if k == 14:  # lock some leaders instead
            idx = (3 * np.arange(5, K, 10)).astype(np.uint32)
            gw.set_entity_flags(idx, np.ones(len(idx), dtype=np.uint32))
            for i in idx:
                ow.set_flags(int(i), 1)
        q = sw.queries()
        ow.tick(sw.now_ns(), upd, sw.x[upd], sw.z[upd], None, None, None, None, q)
        res = gw.tick(sw.now_ns(), upd_idx=upd, upd_x=sw.x[upd], upd_z=sw.z[upd], queries=q, records_cap=1 << 21)
        compare_tick(k, res, ow, S)
        cell, member = gw.entity_state()
        ocell, omember = ow.entity_state()
        to_id = lambda a: np.where(a == 0xFFFFFFFF, 0, a + 0x10000).astype(np.uint32)
        assert np.array_equal(cell, to_id(ocell)), f"tick {k}: position cells"
        assert np.array_equal(member, to_id(omember)), f"tick {k}: entity maps (group members move with the notifier)"
        total_ho += len(res.handovers)
        total_lock += res.n_locked_aborts
        riders = 3 * np.arange(K) + 2
        moved_riders += int((member[riders] != cell[riders]).sum())  # in another cell's map than their own position's
    assert total_ho > 100 and total_lock > 5 and moved_riders > 50


def test_segments_expand_to_the_dense_records(amd):
    """chd_tick_fetch_segments — per connection the plan's descriptors + the cells' entity-channel columns + the
    explicit records of the subscriptions that needed a per-entity decision — expands on the host to exactly the records
    chd_tick returns densely, connection by connection, on every emit form (descriptor path: mostly column references;
    the other forms: all explicit), through first fan-outs, sparse updates (filtered windows) and steady state."""
    from channeld_amd.engine import expand_segments

    cfg = synth.load_config("spatial_static_4x4.json")
    N, S = 900, 48
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0x5E6, tick_ms=50, outside_frac=0.01, locked_frac=0.02))
    ctl, gw = make(amd, cfg, N, S)
    gw.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    gw.add_subscribers(None, sw.sub_conn)
    rng = np.random.default_rng(6)
    n_col = n_exp = 0
    for k in range(14):
        sw.step()
        idx = np.arange(N, dtype=np.uint32) if k % 5 != 3 else np.sort(rng.choice(N, N // 2, replace=False)).astype(np.uint32)
        res = gw.tick(sw.now_ns(), upd_idx=idx, upd_x=sw.x[idx], upd_z=sw.z[idx], queries=sw.queries())
        seg = gw.fetch_segments(pinned=(k % 2 == 0))
        assert seg["n_records"] == res.n_records
        rec = expand_segments(seg, sw.sub_conn)
        assert len(rec) == res.n_records
        # per connection: the same multiset
        off = np.zeros(S + 1, dtype=np.int64)
        for s in range(S):
            a, b = int(seg["conn_seg_off"][s]), int(seg["conn_seg_off"][s + 1])
            off[s + 1] = off[s] + int(seg["segments"]["n_records"][a:b].sum())
        for s in range(S):
            want = res.records_of(s)
            got = rec[off[s]: off[s + 1]]
            assert np.array_equal(canon(got["conn"], got["channel"]), canon(want["conn"], want["channel"])), f"tick {k} slot {s}"
        expl = (seg["segments"]["n_info"] & (1 << 24)) != 0
        n_exp += int(expl.sum())
        n_col += int((~expl).sum())
    assert n_exp > 0 or (EMIT_FLAGS & 64)  # (window columns: the descriptor path may leave nothing explicit)
    if EMIT_FLAGS & 64:
        assert n_col > n_exp  # the descriptor path: segments are mostly references into the columns


def test_tick_segments_in_one_call_equals_the_two_calls(amd):
    """chd_tick_segments = chd_tick (host buffers, no dense records) + chd_tick_fetch_segments with two host synchronisations instead
    of five: two identical worlds, one ticked by the two calls, one by the one — handover records, unsub / new-sub lists, query
    status, counts, and the expanded segments equal tick for tick; segment buffers that start too small (CHD_E_CAPACITY: the tick is
    done, the lists are fetched, the caller grows and fetches the segments) included."""
    from channeld_amd.engine import REC_DTYPE, expand_segments

    cfg = synth.load_config("spatial_static_4x4.json")
    N, S = 900, 48
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0x5E7, tick_ms=50, outside_frac=0.01, locked_frac=0.02))
    worlds = []
    for _ in range(2):
        ctl, gw = make(amd, cfg, N, S)
        gw.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
        gw.add_subscribers(None, sw.sub_conn)
        worlds.append((ctl, gw))
    (_, ga), (_, gb) = worlds
    gb._seg_bufs = dict(seg=np.zeros(4, dtype=gb.SEG_DTYPE), off=np.zeros(S + 1, dtype=np.uint32), col=np.zeros(8, dtype=np.uint32),
                        rec=np.zeros(2, dtype=REC_DTYPE), roff=np.zeros(S + 1, dtype=np.uint64))  # (far too small: the grow-and-fetch path)
    rng = np.random.default_rng(7)
    total = 0
    for k in range(12):
        sw.step()
        idx = np.arange(N, dtype=np.uint32) if k % 4 != 2 else np.sort(rng.choice(N, N // 2, replace=False)).astype(np.uint32)
        kw = dict(upd_idx=idx, upd_x=sw.x[idx], upd_z=sw.z[idx], queries=sw.queries())
        ra = ga.tick(sw.now_ns(), want_records=False, **kw)
        sa = ga.fetch_segments(pinned=False)
        rb, sb = gb.tick_segments(sw.now_ns(), pinned=False, **kw)
        assert ra.n_records == rb.n_records == sa["n_records"] == sb["n_records"] and ra.overflow == rb.overflow == 0
        assert np.array_equal(np.sort(ra.handovers, order=["entity", "src", "dst"]), np.sort(rb.handovers, order=["entity", "src", "dst"]))
        assert ra.n_locked_aborts == rb.n_locked_aborts and np.array_equal(ra.query_status, rb.query_status)
        assert np.array_equal(canon(ra.unsub_sub, ra.unsub_channel), canon(rb.unsub_sub, rb.unsub_channel))
        assert np.array_equal(canon(ra.newsub_sub, ra.newsub_channel), canon(rb.newsub_sub, rb.newsub_channel))
        ea, eb = expand_segments(sa, sw.sub_conn), expand_segments(sb, sw.sub_conn)
        assert len(ea) == len(eb) == ra.n_records and np.array_equal(canon(ea["conn"], ea["channel"]), canon(eb["conn"], eb["channel"]))
        assert np.array_equal(sa["conn_seg_off"][: S + 1], sb["conn_seg_off"][: S + 1])
        total += ra.n_records
    assert total > 50_000
    for ctl, _ in worlds:
        ctl.close()


def test_tick_segments_begin_end_pair_equals_tick_segments(amd):
    """chd_tick_segments_begin / chd_tick_segments_end — tick t+1 enqueued while tick t's results travel, one wait, one
    copy and one page-locked block per tick.  Two identical worlds, one ticked by chd_tick_segments, one by the pair kept TWO ticks deep
    (begin(t+1) before end(t)): the offsets, the segments, the columns, the explicit records' offsets and every count are
    byte-identical tick for tick; the explicit records and the lists (whose order inside a connection / a list is the order of
    device atomics in both forms) equal as multisets per connection / per list.  The world starts with no subscriptions and then
    subscribes everybody at once (block sizes from a few KB to MBs from one tick to the next); a third begin without an end is
    refused, and so is an end with nothing in flight."""
    from channeld_amd.engine import expand_segments

    cfg = synth.load_config("spatial_static_40x40.json")
    N, S = 30_000, 1500
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0x5E8, tick_ms=50, outside_frac=0.01, locked_frac=0.02))
    worlds = []
    for _ in range(2):
        ctl, gw = make(amd, cfg, N, S)
        gw.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
        gw.add_subscribers(None, sw.sub_conn)
        worlds.append((ctl, gw))
    (_, ga), (_, gb) = worlds
    with pytest.raises(Exception):
        gb.tick_segments_end()
    rng = np.random.default_rng(8)
    T = 14
    ins, want = [], []
    for k in range(T):
        sw.step()
        idx = np.arange(N, dtype=np.uint32) if k % 4 != 2 else np.sort(rng.choice(N, N // 2, replace=False)).astype(np.uint32)
        kw = dict(upd_idx=idx, upd_x=sw.x[idx].copy(), upd_z=sw.z[idx].copy())
        if k >= 2:
            kw["queries"] = sw.queries()
        ins.append((sw.now_ns(), kw))
        ra, sa = ga.tick_segments(sw.now_ns(), pinned=False, **kw)
        want.append((ra, {a: (np.array(v) if isinstance(v, np.ndarray) else v) for a, v in sa.items()}))
    total = 0
    sizes = []

    def check(k):
        nonlocal total
        rb, sb, info = gb.tick_segments_end()
        ra, sa = want[k]
        sizes.append(info["block_bytes"])
        assert ra.n_records == rb.n_records == sa["n_records"] == sb["n_records"] and ra.overflow == rb.overflow == 0, f"tick {k}"
        for name in ("conn_seg_off", "conn_rec_off", "segments", "columns"):
            assert len(sa[name]) == len(sb[name]) and sa[name].tobytes() == sb[name].tobytes(), f"tick {k}: {name}"
        assert len(sa["records"]) == len(sb["records"])
        for s_ in range(S):
            a, b = int(sa["conn_rec_off"][s_]), int(sa["conn_rec_off"][s_ + 1])
            if b > a:
                assert np.array_equal(canon(sa["records"]["conn"][a:b], sa["records"]["channel"][a:b]), canon(sb["records"]["conn"][a:b], sb["records"]["channel"][a:b])), f"tick {k} slot {s_}"
        assert np.array_equal(np.sort(ra.handovers, order=["entity", "src", "dst"]), np.sort(rb.handovers, order=["entity", "src", "dst"]))
        assert ra.n_locked_aborts == rb.n_locked_aborts and np.array_equal(ra.query_status, rb.query_status)
        assert np.array_equal(canon(ra.unsub_sub, ra.unsub_channel), canon(rb.unsub_sub, rb.unsub_channel))
        assert np.array_equal(canon(ra.newsub_sub, ra.newsub_channel), canon(rb.newsub_sub, rb.newsub_channel))
        ia, ib = np.lexsort((ra.newsub_channel, ra.newsub_sub)), np.lexsort((rb.newsub_channel, rb.newsub_sub))
        assert np.array_equal(ra.newsub_interval_ms[ia], rb.newsub_interval_ms[ib])
        eb = expand_segments(sb, sw.sub_conn)
        assert len(eb) == ra.n_records
        total += ra.n_records
        assert info["block_bytes"] > 12 * (S + 1)

    gb.tick_segments_begin(ins[0][0], **ins[0][1])
    for k in range(1, T):
        gb.tick_segments_begin(ins[k][0], **ins[k][1])
        if k == 5:
            with pytest.raises(Exception):  # a third tick in flight
                gb.tick_segments_begin(ins[k][0] + 1, **ins[k][1])
        check(k - 1)
    check(T - 1)
    assert total > 1_000_000 and max(sizes) > 4 * min(sizes)
    # the synchronous calls work again once nothing is in flight, and see the same world
    sw.step()
    idx = np.arange(N, dtype=np.uint32)
    kw = dict(upd_idx=idx, upd_x=sw.x, upd_z=sw.z, queries=sw.queries())
    ra, sa = ga.tick_segments(sw.now_ns(), pinned=False, **kw)
    rb, sb = gb.tick_segments(sw.now_ns(), pinned=False, **kw)
    assert ra.n_records == rb.n_records and sa["segments"].tobytes() == sb["segments"].tobytes()
    for ctl, _ in worlds:
        ctl.close()


def test_tick_segments_begin_end_pair_on_exact_update_buffers(amd):
    """The two-call form on a world with exact update buffers (history_depth) and enqueue-time stamps, 5 000 connections (the
    descriptor path with sub-tick offsets: the filtered kernel's records come out as EXPLICIT segments): against chd_tick_segments
    on an identical world, two ticks deep, arrival stamps passed through the same staging."""
    from channeld_amd.engine import expand_segments

    cfg = synth.load_config("spatial_static_8x8.json")
    N, S = 20_000, 5_000
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0x5E9, tick_ms=50, aoi_scale=0.4))
    worlds = []
    for _ in range(2):
        ctl = amd.StaticGrid2DSpatialController()
        assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
        gw = amd.SpatialWorld(ctl, N, S, max_records=60_000_000, history_depth=256)
        gw.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
        gw.add_subscribers(None, sw.sub_conn)
        worlds.append((ctl, gw))
    (_, ga), (_, gb) = worlds
    assert ga.stats()["schedule"] & 16  # arrival offsets
    rng = np.random.default_rng(9)
    T, prev = 10, 0
    ins, want = [], []
    for k in range(T):
        sw.step()
        now = sw.now_ns()
        arr = now - rng.integers(0, now - prev, N)
        prev = now
        kw = dict(upd_x=sw.x.copy(), upd_z=sw.z.copy(), queries=sw.queries(), upd_arrival_ns=arr)
        ins.append((now, kw))
        ra, sa = ga.tick_segments(now, pinned=False, **kw)
        want.append((ra, {a: (np.array(v) if isinstance(v, np.ndarray) else v) for a, v in sa.items()}))
    n_expl = total = 0

    def check(k):
        nonlocal n_expl, total
        rb, sb, info = gb.tick_segments_end()
        ra, sa = want[k]
        assert ra.n_records == rb.n_records == sa["n_records"] == sb["n_records"] and rb.overflow == 0 and rb.history_overflow == 0, f"tick {k}"
        for name in ("conn_seg_off", "conn_rec_off", "segments", "columns"):
            assert len(sa[name]) == len(sb[name]) and sa[name].tobytes() == sb[name].tobytes(), f"tick {k}: {name}"
        assert len(sa["records"]) == len(sb["records"])
        assert np.array_equal(canon(sa["records"]["conn"], sa["records"]["channel"]), canon(sb["records"]["conn"], sb["records"]["channel"])), f"tick {k}: explicit records"
        ea, eb = expand_segments(sa, sw.sub_conn), expand_segments(sb, sw.sub_conn)
        assert len(eb) == ra.n_records and np.array_equal(canon(ea["conn"], ea["channel"]), canon(eb["conn"], eb["channel"]))
        n_expl += len(sb["records"])
        total += ra.n_records

    gb.tick_segments_begin(ins[0][0], **ins[0][1])
    for k in range(1, T):
        gb.tick_segments_begin(ins[k][0], **ins[k][1])
        check(k - 1)
    check(T - 1)
    assert total > 5_000_000 and n_expl > 100_000, (total, n_expl)
    for ctl, _ in worlds:
        ctl.close()


def test_segments_only_world_hands_out_the_same_segments(amd):
    """CHD_WORLD_SEGMENTS_ONLY: the same segments, columns, offsets, explicit records and counts as a world that also writes the
    dense records — through first fan-outs, partially updating ticks (window columns / deferred subscriptions) and steady state, on
    every emit form (only the descriptor path has anything to skip) —, and the expansion equals the other world's dense records
    connection by connection."""
    from channeld_amd import _lib
    from channeld_amd.engine import expand_segments

    cfg = synth.load_config("spatial_static_4x4.json")
    N, S = 900, 48
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0x5EA, tick_ms=50, outside_frac=0.01, locked_frac=0.02))
    if EMIT_FLAGS & 2:
        with pytest.raises(Exception):  # (the cell-major emit is made of dense records)
            make(amd, cfg, N, S, extra_flags=_lib.WORLD_SEGMENTS_ONLY)
        return
    worlds = []
    for extra in (0, _lib.WORLD_SEGMENTS_ONLY):
        ctl, gw = make(amd, cfg, N, S, extra_flags=extra)
        gw.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
        gw.add_subscribers(None, sw.sub_conn)
        worlds.append((ctl, gw))
    (_, ga), (_, gb) = worlds
    rng = np.random.default_rng(10)
    total = 0
    for k in range(14):
        sw.step()
        idx = np.arange(N, dtype=np.uint32) if k % 5 != 3 else np.sort(rng.choice(N, N // 2, replace=False)).astype(np.uint32)
        kw = dict(upd_idx=idx, upd_x=sw.x[idx], upd_z=sw.z[idx], queries=sw.queries())
        ra = ga.tick(sw.now_ns(), **kw)
        sa = {a: (np.array(v) if isinstance(v, np.ndarray) else v) for a, v in ga.fetch_segments(pinned=False).items()}
        rb, sb = gb.tick_segments(sw.now_ns(), pinned=False, **kw)
        assert ra.n_records == rb.n_records == sb["n_records"]
        for name in ("conn_seg_off", "conn_rec_off", "segments", "columns"):
            assert len(sa[name]) == len(sb[name]) and sa[name].tobytes() == sb[name].tobytes(), f"tick {k}: {name}"
        assert np.array_equal(canon(sa["records"]["conn"], sa["records"]["channel"]), canon(sb["records"]["conn"], sb["records"]["channel"]))
        eb = expand_segments(sb, sw.sub_conn)
        assert np.array_equal(canon(eb["conn"], eb["channel"]), canon(ra.records["conn"], ra.records["channel"])), f"tick {k}"
        total += ra.n_records
    assert total > 50_000
    with pytest.raises(Exception):
        gb.digest()
    for ctl, _ in worlds:
        ctl.close()


def test_tick_device_reports_a_repeated_slot(amd):
    """VERDICT r2 #10 / ADVICE r1: chd_tick_device cannot check its precondition on the host (the inputs are device arrays) —
    the device does while it ingests: an entity slot twice in one round of updates, or a subscriber slot twice, sets overflow
    bit 256 and the fetch fails loudly instead of racing silently."""
    cfg = synth.load_config("spatial_static_2x2.json")
    N, S = 64, 8
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xD0B, tick_ms=50))
    ctl, gw = make(amd, cfg, N, S)
    gw.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    gw.add_subscribers(None, sw.sub_conn)
    sw.step()
    q = sw.queries()
    dq = gw.device_array(q)
    idx = np.arange(N, dtype=np.uint32)
    dx, dz = gw.device_array(sw.x), gw.device_array(sw.z)
    gw.tick_device(sw.now_ns(), n_updates=N, d_upd_x=dx.at(0), d_upd_z=dz.at(0), d_upd_idx=gw.device_array(idx).at(0), n_queries=S, d_queries=dq.at(0))
    assert gw.fetch().overflow == 0
    sw.step()
    bad = idx.copy()
    bad[7] = bad[3]  # slot 3 twice
    gw.tick_device(sw.now_ns(), n_updates=N, d_upd_x=dx.at(0), d_upd_z=dz.at(0), d_upd_idx=gw.device_array(bad).at(0), n_queries=S, d_queries=dq.at(0))
    assert gw.fetch(check=False).overflow & 256
    sw.step()
    qsub = np.arange(S, dtype=np.uint32)
    qsub[5] = 2  # subscriber slot 2 twice
    gw.tick_device(sw.now_ns(), n_updates=N, d_upd_x=dx.at(0), d_upd_z=dz.at(0), n_queries=S, d_queries=dq.at(0), d_query_sub=gw.device_array(qsub).at(0))
    assert gw.fetch(check=False).overflow & 256
    sw.step()
    gw.tick_device(sw.now_ns(), n_updates=N, d_upd_x=dx.at(0), d_upd_z=dz.at(0), n_queries=S, d_queries=dq.at(0))
    assert gw.fetch().overflow == 0


def test_record_kernel_event_pairs_can_be_sampled(amd):
    """chd_set_profiling_scope(CHD_PROF_RECORD_KERNEL_EVERY(n)): the HIP event pair around the record kernel on every n-th tick
    only (each event idles the tick's stream for a few microseconds); the ticks in between report emit_main_us == 0, the counts of
    every tick stay what they are."""
    cfg = synth.load_config("spatial_static_benchmark.json")
    sw = synth.SynthWorld(synth.WorldSpec(cfg, 6000, 4200, 0xC0FFEE21))
    ctl, gw = make(amd, cfg, 6000, 4200, 0)
    gw.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    gw.add_subscribers(None, sw.sub_conn)
    gw.set_profiling(16)
    gw.set_profiling_scope(True, every=4)
    for k in range(12):
        sw.step()
        gw.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=sw.queries(), want_records=False, records_cap=1)
    hist = gw.history(12)  # [0] = the most recent tick = tick 12
    timed = [h["emit_main_us"] > 0 for h in hist]
    on = [j for j, t in enumerate(timed) if t]
    assert len(on) == 3 and all(b - a == 4 for a, b in zip(on, on[1:])), timed  # (every 4th tick by the library's tick counter)
    assert sum(h["n_records"] for h in hist) > 0
    gw.set_profiling_scope(True)  # every launch again
    sw.step()
    gw.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=sw.queries(), want_records=False, records_cap=1)
    assert gw.history(1)[0]["emit_main_us"] > 0
    ctl.close()
