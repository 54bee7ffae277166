"""BASELINE.json's full-size configurations on the GPU, compared with the oracle RECORD BY RECORD.

Nothing of the ~10^8 (config B) / ~10^9 (config C) fan-out records of a tick crosses PCIe: the device folds
them where they lie into an order-independent digest (chd_tick_digest: count, sum / xor of a 64-bit hash of
every {connection | FULL, channel} record, and per connection the sum of its records' hashes) and the oracle's
window formulation folds its own records the same way instead of storing them (digest mode).  Equal digests
per connection <=> equal record multisets per connection (up to a 2^-64 collision), tick after tick.

Config B (spatial_static_benchmark.json, 100K entities / 10K subscribers): 26 ticks of the connection-major
emit — past the point where cells hold entities of two servers (handovers) and 100 ms subscribers have cycled
several times — and the same through the cell-major emit for 8 ticks.  The first four ticks are ALSO compared
through the host-facing chd_tick (records downloaded, digested with numpy): the digest kernel itself is checked
against the records it digests.
Config C (1M entities / 10K subscribers, cell-major emit auto-selected): 12 ticks, ~0.8 G records each.
Handover records, locked aborts and unsubs are compared exactly every tick."""
import json
import os

import numpy as np
import pytest

from channeld_amd import synth
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import channeld_amd

    channeld_amd.load()
    return channeld_amd


def mix64(k):
    with np.errstate(over="ignore"):
        k = (k ^ (k >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        k = (k ^ (k >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return k ^ (k >> np.uint64(31))


def digest(conn, chan):
    h = mix64((conn.astype(np.uint64) << np.uint64(32)) | chan.astype(np.uint64))
    return len(h), int(np.add.reduce(h, dtype=np.uint64)), int(np.bitwise_xor.reduce(h)) if len(h) else 0


def build(amd, N, S, seed, max_records=0, flags=0, cfg_name="spatial_static_benchmark.json"):
    cfg = synth.load_config(cfg_name)
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed))
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    w = amd.SpatialWorld(ctl, N, S, max_records=max_records, flags=flags)
    w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    return cfg, sw, ctl, w


def oracle_world(cfg, sw, N, S, capq, digest_only=True):
    g = orc.grid_from_config(cfg)
    ow = orc.World(g, N, S, capq, 20, 0, literal=False)
    ow.set_threads(min(os.cpu_count() or 8, 128))
    ow.set_digest_only(digest_only)
    ow.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    for s in range(S):
        ow.add_sub(s, int(sw.sub_conn[s]))
    return ow


def compare_tick(k, w, ow, res, sw):
    """device digests vs the oracle's, per connection; exact handover / unsub lists"""
    (cnt, dsum, dxor, _), conn_sum = w.digest()
    (ocnt, osum, oxor, _), oconn = ow.digest()
    assert cnt == ocnt == res.n_records, f"tick {k}: {cnt} records on the device, {ocnt} in the oracle"
    bad = np.nonzero(conn_sum != oconn)[0]
    assert len(bad) == 0, f"tick {k}: record multisets differ for {len(bad)} connections, e.g. slots {bad[:8]}"
    assert (dsum, dxor) == (osum, oxor), f"tick {k}: global digest"
    ent, src, dst, ssrc, sdst = ow.handovers()
    got = np.sort(res.handovers, order="entity")
    o = np.argsort(ent)
    assert np.array_equal(got["entity"], ent[o]) and np.array_equal(got["dst"], dst[o]) and np.array_equal(got["src"], src[o]), f"tick {k}: handovers"
    assert np.array_equal(got["src_server"], ssrc[o]) and np.array_equal(got["dst_server"], sdst[o])
    assert res.n_locked_aborts == ow.locked_aborts()
    us, uc = ow.unsubs()
    key = lambda a, b: np.sort((a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64))
    assert np.array_equal(key(res.unsub_sub, res.unsub_channel), key(us, uc)), f"tick {k}: unsubs"
    assert res.overflow == 0 and res.history_overflow == 0, f"tick {k}: overflow flags"
    return cnt


def run_digest_ticks(amd, N, S, seed, ticks, flags=0, max_records=0, host_ticks=0, host_cap=0):
    cfg, sw, ctl, w = build(amd, N, S, seed, max_records=max_records, flags=flags)
    ow = oracle_world(cfg, sw, N, S, w.capq)
    total = 0
    per_tick = []
    for k in range(ticks):
        sw.step()
        q = sw.queries()
        ow.tick(sw.now_ns(), None, sw.x, sw.z, None, None, None, None, q)
        host = k < host_ticks
        res = w.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=q, want_records=host, records_cap=host_cap if host else 1)
        n = compare_tick(k, w, ow, res, sw)
        if host:
            # the digest kernel against the records it digested, and the dense per-connection grouping
            (cnt, dsum, dxor, _), conn_sum = w.digest()
            assert (cnt, dsum, dxor) == digest(res.records["conn"], res.records["channel"]), f"tick {k}: device digest vs downloaded records"
            want_cnt = np.bincount((res.records["conn"] & 0x7FFFFFFF).astype(np.int64) - 1000, minlength=S)
            assert np.array_equal(res.conn_rec_cnt.astype(np.int64), want_cnt), f"tick {k}: per-connection counts"
            for s in np.random.default_rng(k).choice(S, 32, replace=False):
                r = res.records_of(int(s))
                assert ((r["conn"] & 0x7FFFFFFF) == 1000 + s).all()
                h = mix64((r["conn"].astype(np.uint64) << np.uint64(32)) | r["channel"].astype(np.uint64))
                assert int(np.add.reduce(h, dtype=np.uint64)) == int(conn_sum[s])
        per_tick.append(n)
        total += n
    return total, per_tick, (cfg, sw, ctl, w, ow)


def test_config_b_full_size_26_ticks_record_digests(amd):
    total, per_tick, (cfg, sw, ctl, w, ow) = run_digest_ticks(amd, 100_000, 10_000, 0xC0FFEE01, 26, max_records=200_000_000,
                                                             host_ticks=4, host_cap=150_000_000)
    assert per_tick[0] == 0 and per_tick[1] > 30_000_000  # subscriptions created, then the first (FULL) fan-out: one record per pair
    assert min(per_tick[2:]) > 30_000_000 and total > 1_500_000_000
    # by now cells hold entities that were spawned under different servers (mixed-sender cells: the
    # per-sender emit paths and the sender-range shortcut have been exercised, not only the copy path)
    cell, member = w.entity_state()
    inw = member != 0
    senders_per_cell = [len(np.unique(sw.sender[inw & (member == c)])) for c in np.unique(member[inw])[:225]]
    assert max(senders_per_cell) >= 2


def test_small_cells_at_full_connection_count_record_digests(amd):
    """10 K entities under config B's 10 K subscribers (44 entities per cell): the descriptor path with short descriptors — where
    k_fanout_emit_seg runs 16 persistent waves per CU and a ticket is a few KB of records (chd_world_create: fewer than 256 entity
    slots per cell) — every connection's record digest against the oracle for 10 ticks, the first three also through the
    host-facing records."""
    total, per_tick, (cfg, sw, ctl, w, ow) = run_digest_ticks(amd, 10_000, 10_000, 0xC0FFEE05, 10, max_records=40_000_000, host_ticks=3, host_cap=30_000_000)
    assert per_tick[0] == 0 and per_tick[1] > 3_000_000 and min(per_tick[2:]) > 3_000_000 and total > 50_000_000
    ctl.close()


def test_config_b_cell_major_emit_record_digests(amd):
    total, per_tick, _ = run_digest_ticks(amd, 100_000, 10_000, 0xC0FFEE03, 8, flags=2, max_records=200_000_000)
    assert total > 400_000_000


def run_pipelined_groups(amd, N, S, seed, groups, per_group, flags, max_records, cfg_name="spatial_static_benchmark.json"):
    """CHD_WORLD_PIPELINE_TICKS: `per_group` ticks issued back to back through chd_tick_device (inputs resident, nothing
    fetched in between, so tick t+1's stages overlap tick t's record kernel), then the group's LAST tick compared with
    the oracle in full (records by digest, handovers, unsubs, state) and the ticks before it by their counts from the
    device-side history ring.  The serial schedule must be reproduced tick for tick."""
    cfg, sw, ctl, w = build(amd, N, S, seed, max_records=max_records, flags=flags, cfg_name=cfg_name)
    ow = oracle_world(cfg, sw, N, S, w.capq)
    k = 0
    total = 0
    for g in range(groups):
        xs = np.empty((per_group, N)); zs = np.empty((per_group, N)); qs = np.empty((per_group, S), dtype=synth.AOI_DTYPE)
        now, want = [], []
        for t in range(per_group):
            sw.step()
            xs[t], zs[t], qs[t] = sw.x, sw.z, sw.queries()
            now.append(sw.now_ns())
            ow.tick(now[-1], None, sw.x, sw.z, None, None, None, None, qs[t])
            want.append((ow.digest()[0][0], len(ow.handovers()[0]), len(ow.unsubs()[0])))
        dx, dz, dq = w.device_array(xs), w.device_array(zs), w.device_array(qs)
        w.sync()
        for t in range(per_group):  # back to back: every call after the first is chained to a pipelined tick
            w.tick_device(now[t], n_updates=N, d_upd_x=dx.at(t * N * 8), d_upd_z=dz.at(t * N * 8), n_queries=S, d_queries=dq.at(t * S * 128))
        res = w.fetch(want_records=False)
        compare_tick(k + per_group - 1, w, ow, res, sw)
        hist = w.history(per_group)
        got = [(h["n_records"], h["n_handovers"], h["n_unsubs"]) for h in hist]
        assert got == want, f"group {g}: per-tick counts {got} != oracle {want}"
        total += sum(c for c, _, _ in want)
        k += per_group
        for a in (dx, dz, dq):
            a.free()
    return total, (cfg, sw, ctl, w, ow)


def test_config_b_pipelined_ticks_match_the_serial_oracle(amd):
    """BASELINE config B with CHD_WORLD_PIPELINE_TICKS (what bench.py's `value` runs): 6 groups of 4 back-to-back ticks."""
    total, (cfg, sw, ctl, w, ow) = run_pipelined_groups(amd, 100_000, 10_000, 0xC0FFEE21, 6, 4, flags=128, max_records=120_000_000)
    assert total > 1_200_000_000
    # ... and switched off on the same world: the serial schedule continues from the same state
    w.set_pipelining(False)
    for k in range(3):
        sw.step()
        q = sw.queries()
        ow.tick(sw.now_ns(), None, sw.x, sw.z, None, None, None, None, q)
        res = w.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=q, want_records=False, records_cap=1)
        compare_tick(100 + k, w, ow, res, sw)
    w.set_pipelining(True)
    for k in range(3):  # single pipelined ticks with a fetch after each (never chained)
        sw.step()
        q = sw.queries()
        ow.tick(sw.now_ns(), None, sw.x, sw.z, None, None, None, None, q)
        res = w.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=q, want_records=False, records_cap=1)
        compare_tick(200 + k, w, ow, res, sw)


def test_pipelined_ticks_with_deferred_subscriptions_and_full_buffer(amd):
    """The same on a world whose subscriptions partly take the deferred (filtering) launch — only 30 % of the entities
    update per tick, so cells hold entities with and without buffered updates — and with a record buffer that does not hold
    a whole tick: connections that do not fit keep their state and catch up (overflow flag), identically in both schedules."""
    N, S = 20_000, 4_096
    # (with the window columns a partially updating world hardly defers anything: this test is about the deferred launch
    # beside the record kernel, so it runs the per-entity filtering of round 2 — CHD_WINDOW_COLUMNS=0, read every tick)
    os.environ["CHD_WINDOW_COLUMNS"] = "0"
    try:
        _pipelined_deferred_and_full_buffer(amd, N, S)
    finally:
        del os.environ["CHD_WINDOW_COLUMNS"]


def _pipelined_deferred_and_full_buffer(amd, N, S):
    for max_records, expect_overflow in ((40_000_000, False), (1_500_000, True)):
        seed = 0xC0FFEE22
        worlds = []
        for flags in (1 | 64, 1 | 64 | 128):
            cfg, sw, ctl, w = build(amd, N, S, seed, max_records=max_records, flags=flags)
            worlds.append((sw, w))
        rng = np.random.default_rng(5)
        ovf = 0
        for g in range(4):
            frames = []
            for t in range(3):
                for sw, _ in worlds:
                    sw.step()
                sw0 = worlds[0][0]
                idx = np.sort(rng.choice(N, int(0.3 * N), replace=False)).astype(np.uint32)
                frames.append((sw0.now_ns(), idx, sw0.x[idx].copy(), sw0.z[idx].copy(), sw0.queries().copy()))
            outs = []
            for (sw, w), chained in zip(worlds, (False, True)):
                dev = [(now, len(idx), w.device_array(idx), w.device_array(x), w.device_array(z), w.device_array(q)) for now, idx, x, z, q in frames]
                w.sync()
                per = []
                for now, n, di, dx, dz, dq in dev:
                    w.tick_device(now, n_updates=n, d_upd_x=dx.at(0), d_upd_z=dz.at(0), d_upd_idx=di.at(0), n_queries=S, d_queries=dq.at(0))
                    if not chained:
                        w.sync()
                try:
                    res = w.fetch(want_records=False)
                    overflow, ho = res.overflow, np.sort(res.handovers, order="entity").copy()
                except amd.ChdError as e:  # "tick output truncated (overflow mask 0x4)": what the small buffer is there for
                    assert expect_overflow and "0x4" in str(e), e
                    overflow, ho = 4, None
                (cnt, dsum, dxor, _), conn_sum = w.digest()
                hist = [(h["n_records"], h["n_handovers"], h["n_unsubs"], h["n_deferred_records"]) for h in w.history(3)]
                outs.append((cnt, dsum, dxor, conn_sum.copy(), hist, overflow, ho))
                for _, _, di, dx, dz, dq in dev:
                    for a in (di, dx, dz, dq):
                        a.free()
            a, b = outs
            assert a[:3] == b[:3] and np.array_equal(a[3], b[3]), f"group {g}: record digests differ between the schedules"
            assert a[4] == b[4] and a[5] == b[5] and (a[6] is None or np.array_equal(a[6], b[6])), f"group {g}: {a[4]} vs {b[4]}, overflow {a[5]} vs {b[5]}"
            ovf |= a[5]
            if g == 3 and not expect_overflow:
                assert sum(h[3] for h in a[4]) > 0, "the world never took the deferred launch"
        assert bool(ovf & 4) == expect_overflow, f"overflow mask {ovf:#x}"  # OVF_RECORDS
        for (sw, w) in worlds:
            c0, m0 = worlds[0][1].entity_state()
            c1, m1 = w.entity_state()
            assert np.array_equal(c0, c1) and np.array_equal(m0, m1)


def test_config_c_one_million_entities_12_ticks_record_digests(amd):
    """BASELINE config C: 1M entities / 10K subscribers (cell-major emit auto-selected), every tick compared."""
    N, S = 1_000_000, 10_000
    total, per_tick, (cfg, sw, ctl, w, ow) = run_digest_ticks(amd, N, S, 0xC0FFEE02, 12, max_records=2_000_000_000)
    assert per_tick[0] == 0 and per_tick[1] > 300_000_000 and total > 4_000_000_000
    # size-independent properties on top: per-connection totals, FULL records of a first fan-out = 1 per
    # interest cell + 1 per entity in it, and the due test is idempotent at an unchanged channel time
    g = orc.grid_from_config(cfg)
    cell, member = w.entity_state()
    assert np.array_equal(cell, np.where(ow.entity_state()[0] == 0xFFFFFFFF, 0, ow.entity_state()[0] + 0x10000).astype(np.uint32))
    r3 = w.tick(sw.now_ns(), want_records=False)
    assert r3.n_records == 0


def test_config_c_arrival_stamps_record_digests(amd):
    """BASELINE config C (1 M entities / 10 K subscribers) with the REFERENCE's stamp semantics — every update stamped when it is
    enqueued (channel.go:296-310), exact update buffers (history_depth 1024) — against the ORACLE's committed per-tick digests
    (tests/golden/bench_digests_C_jitter.json = make_bench_digests.py --config C --arrival-jitter: the 18-70 s per tick of the CPU
    oracle do not run inside the suite).  And the SCHEDULE is asserted, not only the records: round 5 found this configuration 70 x
    slow on its last day (every populous cell fell to the serial element walk) because nothing looked at which path
    a world of this shape takes — the descriptor path with arrival offsets, and not one record from the element walk."""
    import os

    N, S, seed, ticks = 1_000_000, 10_000, 0xC0FFEE02, 8
    with open(os.path.join(os.path.dirname(__file__), "golden", "bench_digests_C_jitter.json")) as f:
        golden = json.load(f)["ticks"]
    cfg = synth.load_config("spatial_static_benchmark.json")
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed))
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    w = amd.SpatialWorld(ctl, N, S, max_records=3_000_000_000, history_depth=1024)
    w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    aj = synth.ArrivalJitter(seed, N, 0)
    total = 0
    for k in range(1, ticks + 1):
        sw.step()
        now, arr = aj.next(sw.now_ns())
        res = w.tick(now, upd_x=sw.x, upd_z=sw.z, queries=sw.queries(), upd_arrival_ns=arr, want_records=False, records_cap=1)
        assert res.overflow == 0 and res.history_overflow == 0, (k, res.overflow, res.history_overflow)
        (cnt, dsum, dxor, _), _ = w.digest(per_connection=False)
        assert [cnt, dsum, dxor] == golden[str(k)], f"tick {k}: records digest {[cnt, dsum, dxor]} != the oracle's {golden[str(k)]}"
        h = w.history(1)[0]
        sched = w.stats()["schedule"]
        assert sched & 16 and not (sched & 8), sched                          # CHD_SCHED_ARRIVAL_OFFSETS, not CHD_SCHED_CELL_MAJOR
        assert h["n_deep_records"] == 0, (k, h["n_deep_records"])               # nothing from the element walk
        total += cnt
    assert total > 4_000_000_000


def test_emit_form_follows_the_update_pattern(amd):
    """A world of >= 4096 connections created without emit flags takes the descriptor path every tick; what its windows copy
    changes with the update pattern (chd_api.hip, tick_locked): the cells' full columns while every live entity sends an update
    tick after tick, the WINDOW COLUMNS (per cell the entities updated within the last 1..4 ticks, k_window_columns) as soon as
    some skip a tick, and back.  Full / partial / full update phases, every tick compared with the oracle record by record
    (digests per connection), so the state each phase leaves is what the next one continues from."""
    N, S = 40_000, 4_096
    cfg, sw, ctl, w = build(amd, N, S, 0xC0FFEE31, max_records=60_000_000, flags=0)
    ow = oracle_world(cfg, sw, N, S, w.capq)
    rng = np.random.default_rng(11)
    pattern = [1.0] * 4 + [0.6, 0.97] + [1.0] * 4 + [0.3] + [1.0] * 3
    seen_deferred = 0
    for k, frac in enumerate(pattern):
        sw.step()
        q = sw.queries()
        if frac >= 1.0:
            idx, x, z = None, sw.x, sw.z
        else:
            idx = np.sort(rng.choice(N, int(frac * N), replace=False)).astype(np.uint32)
            x, z = sw.x[idx].copy(), sw.z[idx].copy()
        ow.tick(sw.now_ns(), idx, x, z, None, None, None, None, q)
        res = w.tick(sw.now_ns(), upd_idx=idx, upd_x=x, upd_z=z, queries=q, want_records=False, records_cap=1)
        compare_tick(k, w, ow, res, sw)
        seen_deferred += w.history(1)[0]["n_deferred_records"]
    # (the descriptor path's deferred launch shows in n_deferred_records whenever a cell holds several senders' entities:
    # informational here — which form ran is a performance matter, the records are what is compared)
    print(f"deferred records over the full-update ticks: {seen_deferred}")
    # a world that never updates fully: window columns every tick — most records still come from the descriptor kernel
    cfg, sw, ctl, w = build(amd, N, S, 0xC0FFEE32, max_records=60_000_000, flags=0)
    ow = oracle_world(cfg, sw, N, S, w.capq)
    for k in range(4):
        sw.step()
        q = sw.queries()
        idx = np.sort(rng.choice(N, int(0.9 * N), replace=False)).astype(np.uint32)
        x, z = sw.x[idx].copy(), sw.z[idx].copy()
        ow.tick(sw.now_ns(), idx, x, z, None, None, None, None, q)
        res = w.tick(sw.now_ns(), upd_idx=idx, upd_x=x, upd_z=z, queries=q, want_records=False, records_cap=1)
        compare_tick(100 + k, w, ow, res, sw)
        h = w.history(1)[0]
        if k >= 2:  # (past the first, full-state fan-outs)
            assert h["n_deferred_records"] < h["n_records"] // 4, h


def test_config_b_segments_expand_to_the_device_digest(amd):
    """VERDICT r2 #3 at full size: config B's fan-out leaves the device as segment descriptors + columns (a few MB instead of
    644 MB of records); expanded on the host they digest to what the device computes over the records where they lie
    (chd_tick_digest), in total and for every connection — first fan-out, the crossing to the descriptor path, steady state."""
    from channeld_amd.engine import expand_segments

    N, S = 100_000, 10_000
    cfg, sw, ctl, w = build(amd, N, S, 0xC0FFEE31)
    compact = []
    for k in range(6):
        sw.step()
        w.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=sw.queries(), want_records=False)
        (cnt, sm, xr, _), conn_sum = w.digest(per_connection=True)
        seg = w.fetch_segments()
        assert seg["n_records"] == cnt
        rec = expand_segments(seg, sw.sub_conn)
        assert digest(rec["conn"], rec["channel"]) == (cnt, sm, xr), f"tick {k}"
        # per connection: sums of the record hashes
        h = mix64((rec["conn"].astype(np.uint64) << np.uint64(32)) | rec["channel"].astype(np.uint64))
        per_seg = seg["segments"]["n_records"].astype(np.int64)
        slot_of_seg = np.repeat(np.arange(S), np.diff(seg["conn_seg_off"].astype(np.int64)))
        slot_of_rec = np.repeat(slot_of_seg, per_seg)
        got = np.zeros(S, dtype=np.uint64)
        with np.errstate(over="ignore"):
            np.add.at(got, slot_of_rec, h)
        assert np.array_equal(got, conn_sum), f"tick {k}: per-connection digests"
        nbytes = seg["segments"].nbytes + seg["columns"].nbytes + seg["records"].nbytes + seg["conn_seg_off"].nbytes + seg["conn_rec_off"].nbytes
        compact.append((cnt, nbytes))
    # steady state: three orders of magnitude fewer bytes than the expanded records
    cnt, nbytes = compact[-1]
    assert cnt > 50_000_000 and nbytes < 8 * cnt / 50, compact


def test_config_b_segments_only_world_expands_to_the_oracle_digests(amd):
    """CHD_WORLD_SEGMENTS_ONLY at full size: a world whose fan-out is consumed in the segment form writes no plain-copy records at
    all (the host expands them from the columns) — the bench's world (its seed), ticked through chd_tick_segments, every tick's
    segments expanded on the host and digested: equal to the CPU ORACLE's committed list (tests/golden/bench_digests_B.json), first
    fan-out, the crossing to the descriptor path and steady state; the dense outputs answer CHD_E_STATE."""
    from channeld_amd import _lib
    from channeld_amd.engine import expand_segments

    N, S, seed = 100_000, 10_000, 0xC0FFEE01
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_digests_B.json")) as f:
        golden = json.load(f)["ticks"]
    cfg, sw, ctl, w = build(amd, N, S, seed, max_records=200_000_000, flags=_lib.WORLD_SEGMENTS_ONLY)
    total = 0
    for k in range(8):
        sw.step()
        res, seg = w.tick_segments(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=sw.queries())
        rec = expand_segments(seg, sw.sub_conn)
        cnt, sm, xr = digest(rec["conn"], rec["channel"])
        assert [cnt, sm, xr] == golden[str(k + 1)], f"tick {k + 1}: expanded segments digest {[cnt, sm, xr]} != the oracle's {golden[str(k + 1)]}"
        assert res.n_records == cnt and res.overflow == 0
        total += cnt
    assert total > 500_000_000
    with pytest.raises(amd.ChdError):
        w.digest(per_connection=False)
    with pytest.raises(amd.ChdError):
        w.fetch(want_records=True, records_cap=1 << 20)
    ctl.close()


@pytest.mark.parametrize("variant", ["full", "partial", "merged"])
def test_config_b_wire_streams_from_images_equal_the_record_path(amd, monkeypatch, variant):
    """SURVEY 8f-1 at full size: the packet streams of config B's ticks (3 - 11 GB each) built from the fan-out descriptors
    (per-cell message images + copy ranges, k_wire_layout_img) equal, byte for byte, the streams the record path builds for
    the same ticks (k_wire_layout + copy kernels, CHD_WIRE_IMAGES=0) — and the record path is what tests/test_gpu_wire.py
    pins to the wire oracle.  Payload lengths vary per channel (20 - 120 bytes, full states 100 - 400) so that message
    boundaries, packet cuts and the 16-byte alignment of every copied range differ from connection to connection.
    full: every entity updates every tick.  partial: 90 % do (the image world keeps the window columns and builds their
    images, the record-path world takes the one-launch filtering emit: same records, same order).  merged: a masks world,
    every message the merge of the updates its window selected (one image per window mask and cell)."""
    N, S, WIRE, MASKS = 100_000, 10_000, 8, 32
    cfg = synth.load_config("spatial_static_benchmark.json")
    rng = np.random.default_rng(0xB17E5)
    upd_len = rng.integers(20, 121, N)
    full_len = rng.integers(100, 401, N)
    blob = rng.integers(0, 256, 400, dtype=np.uint8).tobytes() * 2
    ncell = int(cfg["GridCols"]) * int(cfg["GridRows"])
    merged = variant == "merged"
    upd_payloads = [blob[i % 37: i % 37 + (int(upd_len[i]) if not merged else 12 + i % 30)] for i in range(N)]

    def make(images):
        if images:
            monkeypatch.delenv("CHD_WIRE_IMAGES", raising=False)
        else:
            monkeypatch.setenv("CHD_WIRE_IMAGES", "0")
        sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE51))
        ctl = amd.StaticGrid2DSpatialController()
        assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
        w = amd.SpatialWorld(ctl, N, S, flags=WIRE | (MASKS if merged else 0), max_records=400_000_000, wire_max_update_len=128, wire_max_full_len=400)
        w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
        w.add_subscribers(None, sw.sub_conn)
        if merged:
            w.wire_set_type_url(False, b"type.googleapis.com/tpspb.EntityChannelData")
            w.wire_set_type_url(True, b"type.googleapis.com/unrealpb.SpatialChannelData")
        else:
            w.wire_set_payloads(0, np.arange(N), upd_payloads)
        w.wire_set_payloads(1, np.arange(N), [blob[i % 41: i % 41 + int(full_len[i])] for i in range(N)])
        w.wire_set_payloads(2, 0x10000 + np.arange(ncell), [blob[c % 29: c % 29 + 40 + c % 50] for c in range(ncell)])
        w.wire_set_payloads(3, 0x10000 + np.arange(ncell), [blob[c % 31: c % 31 + 200 + c % 90] for c in range(ncell)])
        return sw, ctl, w

    worlds = [make(True), make(False)]
    pick = np.random.default_rng(0x9A87)
    checked = 0
    for k in range(6):
        idx = None
        if variant == "partial" and k >= 2:
            idx = np.flatnonzero(pick.random(N) < 0.9).astype(np.uint32)
        outs = []
        for sw, _ctl, w in worlds:
            sw.step()
            if merged:
                w.wire_set_payloads(0, np.arange(N), upd_payloads)  # (this tick's update messages: the tick's ring slot)
            w.tick(sw.now_ns(), upd_idx=idx, upd_x=sw.x if idx is None else sw.x[idx], upd_z=sw.z if idx is None else sw.z[idx],
                   queries=sw.queries(), want_records=False)
            if k < 3:
                continue  # (the first fan-outs are full states of ~10^8 channels: tens of GB of packets)
            nbytes, npackets, ndropped = w.wire_build()
            outs.append((nbytes, npackets, ndropped, w.wire_build_info(), w.wire_fetch()))
        if k < 3:
            continue
        (nb0, np0, nd0, info0, (off0, npk0, data0)), (nb1, np1, nd1, info1, (off1, npk1, data1)) = outs
        assert info0[0] > 100_000 and info1[0] == 0, "one world takes the image path, the other the record path"
        assert (nb0, np0, nd0) == (nb1, np1, nd1) and nb0 > 2_000_000_000
        assert np.array_equal(off0, off1) and np.array_equal(npk0, npk1)
        step = 1 << 28
        for a in range(0, nb0, step):
            if not np.array_equal(data0[a:a + step], data1[a:a + step]):
                bad = a + int(np.flatnonzero(data0[a:a + step] != data1[a:a + step])[0])
                s = int(np.searchsorted(off0, bad, side="right")) - 1
                raise AssertionError(f"tick {k}: streams differ at byte {bad} (connection slot {s}, offset {bad - int(off0[s])} of {int(off0[s + 1] - off0[s])})")
        checked += 1
        del data0, data1, outs
    assert checked == 3


@pytest.mark.parametrize("tick_jitter_us", [0, 3000], ids=["ticks-on-grid", "ticks-off-grid"])
def test_config_b_arrival_stamps_at_enqueue_time_record_digests(amd, tick_jitter_us):
    """the reference stamps every update when it is ENQUEUED (Channel.PutMessage, channel.go:296-310), tickData compares
    those stamps with the window edges (data.go:236-241).  Config B with every update stamped uniformly inside its tick interval,
    on a world with exact update buffers (history_depth 1024): every connection's record digest equals the oracle's buffer walk
    (orc_world_tick_arrivals; the sorted newest-first walk, tests/test_world_oracle.py), tick after tick, history_overflow == 0 —
    and nearly every record comes from the streaming kernels (copy + filtered descriptors), not from the element-buffer walk.
    ticks-off-grid: the tick times themselves are irregular, so the subscriptions' phases are off the tick grid and EVERY window
    cuts through a tick's arrivals (all records through the per-entity compare)."""
    N, S, seed, ticks = 100_000, 10_000, 0xC0FFEE05, 14 if tick_jitter_us else 26
    cfg = synth.load_config("spatial_static_benchmark.json")
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed))
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    w = amd.SpatialWorld(ctl, N, S, max_records=200_000_000, history_depth=1024)
    w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    ow = oracle_world(cfg, sw, N, S, w.capq)
    ow.set_sorted_walk(True)
    rng = np.random.default_rng(seed)
    prev = total = deep = filt = 0
    for k in range(ticks):
        sw.step()
        now = sw.now_ns() + (int(rng.integers(-tick_jitter_us, tick_jitter_us + 1)) * 1000 + int(rng.integers(0, 1000)) if tick_jitter_us else 0)
        arr = now - rng.integers(0, now - prev, N)  # in (prev, now]
        q = sw.queries()
        ow.tick(now, None, sw.x, sw.z, None, None, None, None, q, upd_arrival=arr)
        res = w.tick(now, upd_x=sw.x, upd_z=sw.z, queries=q, upd_arrival_ns=arr, want_records=False, records_cap=1)
        total += compare_tick(k, w, ow, res, sw)
        h = w.history(1)[0]
        deep += h["n_deep_records"]
        filt += h["n_filtered_records"]
        prev = now
    assert not ow.unsorted() and total > (600_000_000 if tick_jitter_us else 1_400_000_000)
    # the element-buffer walk is the exception, not the path: nothing in this world is irregular
    assert deep == 0, (deep, filt, total)
    # on the tick grid only the 20 ms subscriptions' windows cut through a tick's arrivals; off the grid every window does
    assert filt > (0.4 * total if tick_jitter_us else 0.01 * total), (filt, total)


@pytest.mark.parametrize("tick_jitter_us", [0, 3000], ids=["ticks-on-grid", "ticks-off-grid"])
def test_config_b_pipelined_ticks_on_exact_update_buffers_match_the_oracle(amd, tick_jitter_us):
    """CHD_WORLD_PIPELINE_TICKS on a world with exact update buffers and enqueue-time stamps.  Tick t's record kernels
    (k_fanout_emit_seg + k_fanout_emit_filt_cm) run beside tick t+1's stages; what the filtered kernel reads or writes meanwhile —
    the cells' compact entries and offset columns, the filtered descriptors' lists / items / windows, the per-subscription and
    per-connection record counts — exists once per tick parity, and its record count joins the tick's row behind the epilogue
    (k_filt_fold).  Groups of back-to-back chd_tick_device calls: the group's LAST tick against the oracle in full (every
    connection's digest, handovers, unsubs, state), every tick of the group by its counts from the history ring (records — the
    filtered ones included —, handovers, unsubs), history_overflow 0, no element walk; then the schedule switched off and on
    again on the same world."""
    N, S, seed, groups, per_group = 100_000, 10_000, 0xC0FFEE25, 5, 3
    cfg = synth.load_config("spatial_static_benchmark.json")
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed))
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    w = amd.SpatialWorld(ctl, N, S, max_records=200_000_000, history_depth=1024, flags=128)
    assert w.stats()["schedule"] & 16 and w.stats()["schedule"] & 4, w.stats()["schedule"]  # arrival offsets, pipelined
    w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    ow = oracle_world(cfg, sw, N, S, w.capq)
    ow.set_sorted_walk(True)
    rng = np.random.default_rng(seed)
    prev = total = filt = k = 0

    def frame():
        nonlocal prev
        sw.step()
        now = sw.now_ns() + (int(rng.integers(-tick_jitter_us, tick_jitter_us + 1)) * 1000 + int(rng.integers(0, 1000)) if tick_jitter_us else 0)
        arr = now - rng.integers(0, now - prev, N)  # in (prev, now]
        prev = now
        return now, arr, sw.queries()

    for g in range(groups):
        xs = np.empty((per_group, N)); zs = np.empty((per_group, N)); qs = np.empty((per_group, S), dtype=synth.AOI_DTYPE)
        arrs = np.empty((per_group, N), dtype=np.int64)
        nows, want = [], []
        for t in range(per_group):
            now, arrs[t], qs[t] = frame()
            xs[t], zs[t] = sw.x, sw.z
            nows.append(now)
            ow.tick(now, None, sw.x, sw.z, None, None, None, None, qs[t], upd_arrival=arrs[t])
            want.append((ow.digest()[0][0], len(ow.handovers()[0]), len(ow.unsubs()[0])))
        dx, dz, dq, da = w.device_array(xs), w.device_array(zs), w.device_array(qs), w.device_array(arrs)
        w.sync()
        for t in range(per_group):  # back to back: every call after the first is chained to a pipelined tick
            w.tick_device(nows[t], n_updates=N, d_upd_x=dx.at(t * N * 8), d_upd_z=dz.at(t * N * 8), n_queries=S, d_queries=dq.at(t * S * 128),
                          d_upd_arrival=da.at(t * N * 8))
        res = w.fetch(want_records=False)
        assert res.history_overflow == 0
        compare_tick(k + per_group - 1, w, ow, res, sw)
        hist = w.history(per_group)
        got = [(h["n_records"], h["n_handovers"], h["n_unsubs"]) for h in hist]
        assert got == want, f"group {g}: per-tick counts {got} != oracle {want}"
        assert all(h["n_deep_records"] == 0 for h in hist)
        filt += sum(h["n_filtered_records"] for h in hist)
        total += sum(c for c, _, _ in want)
        k += per_group
        for a in (dx, dz, dq, da):
            a.free()
    assert not ow.unsorted() and total > 500_000_000 and filt > (0.4 * total if tick_jitter_us else 0.01 * total), (total, filt)
    for on in (False, True):  # the serial schedule continues from the same state, and back (single ticks, never chained)
        w.set_pipelining(on)
        for _ in range(2):
            now, arr, q = frame()
            ow.tick(now, None, sw.x, sw.z, None, None, None, None, q, upd_arrival=arr)
            res = w.tick(now, upd_x=sw.x, upd_z=sw.z, queries=q, upd_arrival_ns=arr, want_records=False, records_cap=1)
            compare_tick(k, w, ow, res, sw)
            k += 1
    ctl.close()


def test_config_b_gated_overlap_back_to_back_device_ticks_equal_the_oracle_list(amd):
    """CHD_WORLD_OVERLAP_INTEREST | CHD_WORLD_GATED_OVERLAP — the bench's serial schedule: the interest updates on a second stream,
    forked and joined by device-side flags instead of HIP events.  Config B (the bench's seed) as the bench drives it: groups of
    back-to-back chd_tick_device calls on inputs resident in HBM (nothing else asked of the context in between: the gated fork),
    every group's LAST tick digested — a wrong join or fork of any tick of the group changes the subscription state that tick
    sees — against the CPU oracle's committed list (tests/golden/bench_digests_B.json).  A group that follows the digest call
    starts with the event form, the others with the flag form."""
    from channeld_amd import _lib

    N, S, seed, groups, per = 100_000, 10_000, 0xC0FFEE01, 6, 5
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_digests_B.json")) as f:
        golden = json.load(f)["ticks"]
    cfg, sw, ctl, w = build(amd, N, S, seed, max_records=200_000_000, flags=_lib.WORLD_OVERLAP_INTEREST | _lib.WORLD_GATED_OVERLAP)
    T = groups * per
    xs = np.empty((T, N)); zs = np.empty((T, N)); qs = np.empty((T, S), dtype=synth.AOI_DTYPE); now = np.empty(T, dtype=np.int64)
    for t in range(T):
        sw.step()
        xs[t], zs[t], qs[t], now[t] = sw.x, sw.z, sw.queries(), sw.now_ns()
    dx, dz, dq = w.device_array(xs), w.device_array(zs), w.device_array(qs)
    w.sync()
    total = 0
    for gi in range(groups):
        for t in range(gi * per, (gi + 1) * per):
            w.tick_device(int(now[t]), n_updates=N, d_upd_x=dx.at(t * N * 8), d_upd_z=dz.at(t * N * 8), n_queries=S, d_queries=dq.at(t * S * 128))
        (cnt, dsum, dxor, _), _ = w.digest(per_connection=False)
        assert [cnt, dsum, dxor] == golden[str((gi + 1) * per)], f"tick {(gi + 1) * per}: digest {[cnt, dsum, dxor]} != the oracle's {golden[str((gi + 1) * per)]}"
        res = w.fetch()
        assert res.overflow == 0 and res.history_overflow == 0
        total += cnt
    assert total > 300_000_000
    ctl.close()
