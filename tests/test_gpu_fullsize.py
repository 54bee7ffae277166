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


def test_config_b_cell_major_emit_record_digests(amd):
    total, per_tick, _ = run_digest_ticks(amd, 100_000, 10_000, 0xC0FFEE03, 8, flags=2, max_records=200_000_000)
    assert total > 400_000_000


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
