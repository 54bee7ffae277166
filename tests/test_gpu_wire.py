"""GPU: wire-format fan-out buffers (chd_wire_* through the C-ABI).

The per-connection byte streams built on the device must equal what the reference's send path
produces for the same messages in the same order: MessagePack{channelId, msgType=8,
msgBody=ChannelDataUpdateMessage{data}} (data.go:293-318, connection.go:57-83), greedy Packets of
at most 65535 bytes behind the 5-byte tag (connection.go:626-714).  Expected bytes come from
oracle/wire.py, which tests/test_wire_oracle.py pins against packets serialized with the
reference's own protobuf descriptor.  WHICH messages go to whom is the records' parity
(tests/test_gpu_world.py); here the records of each tick are the message list."""
import json
import os

import numpy as np
import pytest

from channeld_amd import synth
from oracle import wire

pytestmark = pytest.mark.gpu

WIRE, CELL_MAJOR, CONN_MAJOR, ONE_WAVE = 8, 2, 1, 64
DESC = CONN_MAJOR | ONE_WAVE  # the fan-out's descriptor path: the streams are assembled from per-cell message images
ENT_UPD, ENT_FULL, CELL_UPD, CELL_FULL = 0, 1, 2, 3


@pytest.fixture(scope="module")
def amd():
    import channeld_amd

    channeld_amd.load()
    return channeld_amd


def any_bytes(rng, n):
    return bytes(rng.integers(0, 256, int(n), dtype=np.uint8))


def run_wire(amd, cfg_name, N, S, ticks, seed, flags, upd_len, full_len, max_upd=0, max_full=0, tick_ms=50, aoi_scale=1.0, update_frac=1.0):
    cfg = synth.load_config(cfg_name)
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=tick_ms, aoi_scale=aoi_scale))
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    w = amd.SpatialWorld(ctl, N, S, flags=flags | WIRE, max_records=1 << 22, wire_max_update_len=max_upd, wire_max_full_len=max_full)
    w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    rng = np.random.default_rng(seed & 0xFFFF)
    ncell = ctl.GridCols * ctl.GridRows
    ent = {0: {}, 1: {}}     # full? -> slot -> Any bytes
    cell = {0: {}, 1: {}}    # full? -> channel id -> Any bytes
    for i in range(N):
        ent[1][i] = any_bytes(rng, full_len())
    for c in range(ncell):
        cell[0][0x10000 + c] = any_bytes(rng, upd_len())
        cell[1][0x10000 + c] = any_bytes(rng, full_len())
    w.wire_set_payloads(ENT_FULL, list(ent[1]), list(ent[1].values()))
    w.wire_set_payloads(CELL_UPD, list(cell[0]), list(cell[0].values()))
    w.wire_set_payloads(CELL_FULL, list(cell[1]), list(cell[1].values()))
    total_bytes = total_packets = total_dropped = 0
    run_wire.image_ranges = run_wire.record_path_conns = 0
    for k in range(ticks):
        sw.step()
        # this tick's merged update of every entity, as the host would marshal it
        upd = {i: any_bytes(rng, upd_len()) for i in range(N)}
        ent[0].update(upd)
        w.wire_set_payloads(ENT_UPD, list(upd), list(upd.values()))
        cu = 0x10000 + rng.integers(0, ncell, 3).astype(np.uint32)
        if update_frac >= 1.0:
            idx = None
        else:  # a partially updating tick (ticks 0 and 1 complete, so that the world is on the descriptor path before)
            idx = np.arange(N, dtype=np.uint32) if k < 2 else np.flatnonzero(rng.random(N) < update_frac).astype(np.uint32)
        res = w.tick(sw.now_ns(), upd_idx=idx, upd_x=sw.x if idx is None else sw.x[idx], upd_z=sw.z if idx is None else sw.z[idx],
                     queries=sw.queries(), cell_upd_channel=cu, cell_upd_sender=np.full(3, 5, dtype=np.uint32), records_cap=1 << 22)
        nbytes, npackets, ndropped = w.wire_build()
        nr, nc = w.wire_build_info()
        run_wire.image_ranges += nr
        run_wire.record_path_conns += nc
        off, npk, data = w.wire_fetch()
        assert int(off[S]) == nbytes == len(data) and int(npk.sum()) == npackets
        dropped = 0
        for s in range(S):
            recs = res.records_of(s)
            packs = []
            for r in recs:
                full = int(r["conn"]) >> 31
                ch = int(r["channel"])
                a = cell[full][ch] if ch < 0x80000 else ent[full][ch - 0x80000]
                packs.append(wire.fanout_message_pack(ch, a))
            dropped += sum(1 for p in packs if len(p) >= 65530)
            want, counts = wire.flush_stream(packs)
            got = data[int(off[s]):int(off[s + 1])].tobytes()
            assert len(got) == len(want), f"tick {k} slot {s}: {len(got)} bytes vs {len(want)}"
            assert got == want, f"tick {k} slot {s}: stream bytes"
            assert int(npk[s]) == len(counts)
        assert ndropped == dropped
        total_bytes += nbytes
        total_packets += npackets
        total_dropped += ndropped
    return total_bytes, total_packets, total_dropped


@pytest.mark.parametrize("mode", [CONN_MAJOR, CELL_MAJOR, DESC])
def test_wire_streams_small_world(amd, mode):
    rng = np.random.default_rng(1)
    tb, tp, td = run_wire(amd, "spatial_static_2x2.json", 300, 24, 6, 0xC0FFEE41, mode,
                          upd_len=lambda: rng.integers(0, 100), full_len=lambda: rng.integers(40, 300))
    assert tb > 100_000 and tp >= 24 and td == 0
    assert (run_wire.image_ranges > 0) == (mode == DESC)


def test_wire_streams_from_cell_images_match_the_record_path(amd, monkeypatch):
    """The descriptor-driven builder (per-cell message images + copy ranges, k_wire_layout_img) against the oracle on a
    world with every kind of subscription: first fan-outs (full states), 20 / 50 / 100 ms windows (one to three windows per
    fan-out, the spatial channels' own updates in front), cells of a few hundred entities so that a piece is cut by several
    packets, and the subscriptions the fan-out defers (record path) in the same streams."""
    rng = np.random.default_rng(7)
    tb, tp, td = run_wire(amd, "spatial_static_4x4.json", 2400, 32, 6, 0xC0FFEE47, DESC, tick_ms=33,
                          upd_len=lambda: rng.integers(20, 120), full_len=lambda: rng.integers(100, 600), max_full=600)
    assert run_wire.image_ranges > 1000 and td == 0
    ranges = run_wire.image_ranges
    # the same world through the record path: same bytes (both equal the oracle), no image ranges
    monkeypatch.setenv("CHD_WIRE_IMAGES", "0")
    rng = np.random.default_rng(7)
    tb2, tp2, td2 = run_wire(amd, "spatial_static_4x4.json", 2400, 32, 6, 0xC0FFEE47, DESC, tick_ms=33,
                             upd_len=lambda: rng.integers(20, 120), full_len=lambda: rng.integers(100, 600), max_full=600)
    assert (tb2, tp2, td2) == (tb, tp, td) and run_wire.image_ranges == 0 and ranges > 0


@pytest.mark.parametrize("frac", [0.9, 0.5])
def test_wire_streams_of_partially_updating_ticks_from_window_column_images(amd, frac):
    """Wire worlds keep the window columns on partially updating ticks (the record kernel writes no position words there;
    WorldDev::seg_no_pos): a subscription's window copies the cell's WINDOW COLUMN, and its bytes come from that column's
    image — one more image per (cell, window shape).  33 ms ticks: one- to three-tick windows, so several shapes are live."""
    rng = np.random.default_rng(8)
    tb, tp, td = run_wire(amd, "spatial_static_4x4.json", 2400, 32, 8, 0xC0FFEE48, DESC, tick_ms=33, update_frac=frac,
                          upd_len=lambda: rng.integers(20, 120), full_len=lambda: rng.integers(100, 600), max_full=600)
    assert run_wire.image_ranges > 1000 and td == 0


@pytest.mark.parametrize("mode", [CONN_MAJOR, DESC])
def test_wire_many_packets_per_connection(amd, mode):
    # ~30 KB full states: a first fan-out of a few dozen channels spans many packets
    rng = np.random.default_rng(2)
    tb, tp, td = run_wire(amd, "spatial_static_2x2.json", 120, 6, 4, 0xC0FFEE42, mode, max_full=30000,
                          upd_len=lambda: rng.integers(10, 90), full_len=lambda: rng.integers(15000, 30000))
    assert tp > 60 and td == 0


@pytest.mark.parametrize("mode", [CONN_MAJOR, DESC])
def test_wire_oversized_messages_are_dropped_like_send_does(amd, mode):
    rng = np.random.default_rng(3)
    tb, tp, td = run_wire(amd, "spatial_static_2x2.json", 40, 4, 3, 0xC0FFEE43, mode, max_full=66000,
                          upd_len=lambda: rng.integers(10, 60), full_len=lambda: rng.choice([200, 65600]))
    assert td > 0


@pytest.mark.parametrize("mode", [CONN_MAJOR, DESC])
def test_wire_benchmark_grid_position_updates(amd, mode):
    # the minimal position update of SURVEY a14: a 21-byte value inside a 66-byte Any
    rng = np.random.default_rng(4)
    tb, tp, td = run_wire(amd, "spatial_static_benchmark.json", 3000, 60, 5, 0xC0FFEE44, mode,
                          upd_len=lambda: 66, full_len=lambda: rng.integers(100, 400))
    assert tb > 1_000_000 and td == 0


@pytest.mark.parametrize("mode,dense", [(CONN_MAJOR, False), (DESC, False), (DESC, True)])
def test_wire_merged_updates_for_slow_subscribers(amd, mode, dense):
    """SURVEY 8f-3 on the device: a WIRE | UPDATE_MASKS world builds every message from the buffered updates the
    subscriber's window selected (data.go:225-269) — Any{type_url, value = those updates' bytes, oldest first}.  Sparse
    updates and 33 ms ticks so that 20 / 50 / 100 ms subscriptions merge different sets (one, two, three or four ticks'
    updates); the expected streams are composed from the per-record masks (whose parity with the reference's buffer
    walk tests/test_gpu_world.py establishes) and the host's own history of payloads, through the pinned wire oracle."""
    MASKS = 32
    cfg = synth.load_config("spatial_static_4x4.json")
    N, S = 500, 24
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE45, tick_ms=33))
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    # DESC: the fan-out's descriptor path; dense: every entity updates every tick, so that a window's mask is the same for every
    # entity of a cell and its bytes come from the (mask, cell) image (k_wire_layout_img, merge mode)
    w = amd.SpatialWorld(ctl, N, S, flags=mode | WIRE | MASKS, max_records=1 << 21, wire_max_update_len=64, wire_max_full_len=256)
    w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    rng = np.random.default_rng(45)
    image_ranges = 0
    ncell = ctl.GridCols * ctl.GridRows
    url_e, url_c = b"type.googleapis.com/tpspb.EntityChannelData", b"type.googleapis.com/unrealpb.SpatialChannelData"
    w.wire_set_type_url(False, url_e)
    w.wire_set_type_url(True, url_c)
    full_e = {i: any_bytes(rng, rng.integers(20, 200)) for i in range(N)}
    full_c = {0x10000 + c: any_bytes(rng, rng.integers(20, 200)) for c in range(ncell)}
    w.wire_set_payloads(ENT_FULL, list(full_e), list(full_e.values()))
    w.wire_set_payloads(CELL_FULL, list(full_c), list(full_c.values()))
    upd_e, upd_c = {}, {}   # (entity slot | channel id, tick index) -> the update message's bytes
    merged2 = merged3 = 0
    for k in range(14):
        sw.step()
        idx = np.sort(rng.choice(N, N // 2, replace=False)).astype(np.uint32) if (k % 3 and not dense) else np.arange(N, dtype=np.uint32)
        pay = [any_bytes(rng, rng.integers(0, 60)) for _ in idx]
        for i, b in zip(idx, pay):
            upd_e[(int(i), k)] = b
        w.wire_set_payloads(ENT_UPD, idx, pay)
        cu = np.unique(0x10000 + rng.integers(0, ncell, 4)).astype(np.uint32)
        cpay = [any_bytes(rng, rng.integers(1, 40)) for _ in cu]
        for c, b in zip(cu, cpay):
            upd_c[(int(c), k)] = b
        w.wire_set_payloads(CELL_UPD, cu, cpay)
        res = w.tick(sw.now_ns(), upd_idx=idx, upd_x=sw.x[idx], upd_z=sw.z[idx], queries=sw.queries(), cell_upd_channel=cu,
                     cell_upd_sender=np.full(len(cu), 5, dtype=np.uint32), records_cap=1 << 21)
        assert res.overflow == 0 and res.history_overflow == 0
        nbytes, npackets, ndropped = w.wire_build()
        image_ranges += w.wire_build_info()[0]
        off, npk, data = w.wire_fetch()
        assert int(off[S]) == nbytes == len(data) and ndropped == 0
        for s in range(S):
            a, n = int(res.conn_rec_off[s]), int(res.conn_rec_cnt[s])
            packs = []
            for r, mask in zip(res.records[a:a + n], res.record_masks[a:a + n]):
                ch, full = int(r["channel"]), int(r["conn"]) >> 31
                if full:
                    body = full_c[ch] if ch < 0x80000 else full_e[ch - 0x80000]
                else:
                    bits = [j for j in range(31, -1, -1) if (int(mask) >> j) & 1]  # oldest update first
                    assert bits, "a delta record merges at least one buffered update"
                    src, key, url = (upd_c, ch, url_c) if ch < 0x80000 else (upd_e, ch - 0x80000, url_e)
                    value = b"".join(src[(key, k - j)] for j in bits)
                    body = wire.field_bytes(1, url) + (wire.field_bytes(2, value) if value else b"")
                    merged2 += len(bits) == 2
                    merged3 += len(bits) >= 3
                packs.append(wire.fanout_message_pack(ch, body))
            want, counts = wire.flush_stream(packs)
            got = data[int(off[s]):int(off[s + 1])].tobytes()
            if got != want and os.environ.get("CHD_TEST_DEBUG"):
                c0 = counts[0]
                recs_s, masks_s = res.records[a:a + n], res.record_masks[a:a + n]
                print("DEBUG tick", k, "slot", s, "len", len(got), len(want), "npk", int(npk[s]), len(counts), "first packet msgs", c0)
                for q in range(max(0, c0 - 3), min(n, c0 + 3)):
                    print("  rec", q, hex(int(recs_s[q]["conn"])), hex(int(recs_s[q]["channel"])), bin(int(masks_s[q])), "pack len", len(packs[q]))
                d0 = next(i for i in range(min(len(got), len(want))) if got[i] != want[i])
                print("  first diff", d0, "want", want[d0 - 24:d0 + 16].hex(), "got", got[d0 - 24:d0 + 16].hex())
                tags = [i for i in range(len(got) - 4) if got[i] == 67 and got[i + 1] == 72 and got[i + 4] == 0]
                print("  got CH.. candidates", tags[:12], "want", [i for i in range(len(want) - 4) if want[i] == 67 and want[i + 1] == 72 and want[i + 4] == 0][:12])
                acc = 5
                for q in range(n):
                    e = 1 + (2 if len(packs[q]) >= 128 else 1) + len(packs[q])
                    if c0 - 4 <= q <= c0 + 1:
                        print("  msg", q, "starts at", acc, "entry", e)
                    acc += e
                segs = w.fetch_segments()
                so = segs["conn_seg_off"]
                tot = 0
                for g in segs["segments"][int(so[s]):int(so[s + 1])]:
                    print("  seg", hex(int(g["channel"])), int(g["off"]), hex(int(g["n_info"])), int(g["n_records"]), "records before", tot)
                    tot += int(g["n_records"])
            assert got == want, f"tick {k} slot {s}: merged stream bytes ({len(got)} vs {len(want)})"
            assert int(npk[s]) == len(counts)
    assert merged2 > 1000 and merged3 > 10  # windows that merged two / three and more ticks' updates were exercised
    assert (image_ranges > (800 if dense else 200)) == (mode == DESC), image_ranges  # (sparse: first fan-outs and the complete ticks)


def test_handover_messages_on_the_device(amd):
    """chd_handover_messages builds, per handover of the tick, the two MessagePacks Notify sends
    (spatial.go:738-857) — ChannelDataHandoverMessage{src, dst, contextConnId, Any{SpatialChannelData{entities}}} with and
    without the entities' full data — from the host's per-entity UnrealObjectRef / full-state payloads.  Compared byte
    for byte with oracle/wire.py's composition (pinned to the reference's descriptors by tests/test_wire_oracle.py), for
    lone entities and for handover groups; chd_handover_recipients says who gets which of the two."""
    RECIPIENTS = 4
    cfg = synth.load_config("spatial_static_4x4.json")
    N, S = 600, 32
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE46, outside_frac=0.0, locked_frac=0.0))
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    w = amd.SpatialWorld(ctl, N, S, flags=CONN_MAJOR | WIRE | RECIPIENTS, max_records=1 << 21, wire_max_update_len=96, wire_max_full_len=512)
    # groups of two among the first 100 entities: the rider copies the leader's position
    x0, z0 = sw.x.copy(), sw.z.copy()
    x0[1:100:2], z0[1:100:2] = x0[0:100:2], z0[0:100:2]
    w.spawn(None, sw.chan_id, x0, z0, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    gidx = np.arange(100, dtype=np.uint32)
    w.set_entity_groups(gidx, 500 + gidx // 2)
    rng = np.random.default_rng(46)
    url = b"type.googleapis.com/unrealpb.SpatialChannelData"
    w.wire_set_type_url(2, url)
    objref = {i: any_bytes(rng, rng.integers(0, 90)) for i in range(N)}
    full = {i: any_bytes(rng, rng.integers(10, 500)) for i in range(N)}
    w.wire_set_payloads(4, list(objref), list(objref.values()))
    w.wire_set_payloads(ENT_FULL, list(full), list(full.values()))
    ncell = ctl.GridCols * ctl.GridRows
    last_sender = {}
    n_total = n_group = n_ctx = 0
    for k in range(10):
        sw.step()
        jump = rng.random(N) < 0.1
        sw.x = np.where(jump, np.float64(np.float32(sw.offx + rng.random(N) * sw.W * 0.999)), sw.x)
        sw.x[1:100:2], sw.z[1:100:2] = sw.x[0:100:2], sw.z[0:100:2]
        cu = np.unique(0x10000 + rng.integers(0, ncell, 5)).astype(np.uint32)
        cs = rng.integers(1, 400, len(cu)).astype(np.uint32)
        res = w.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=sw.queries(), cell_upd_channel=cu, cell_upd_sender=cs, records_cap=1 << 21)
        for c, s_ in zip(cu, cs):
            last_sender[int(c)] = int(s_)  # srcChannel.latestDataUpdateConnId
        nh = len(res.handovers)
        msgs = w.handover_messages(nh)
        assert len(msgs) == nh
        for h, (ref_only, with_data) in zip(res.handovers, msgs):
            e = int(h["entity"])
            members = [e] if e >= 100 else [e & ~1, (e & ~1) + 1]  # group members in slot order (both alive)
            ctx = last_sender.get(int(h["src"]), 0)
            for blob, fulls in ((ref_only, False), (with_data, True)):
                entries = [(int(sw.chan_id[m]), wire.spatial_entity_state(objref[m], full[m] if fulls else None)) for m in members]
                want = wire.handover_message_pack(int(h["src"]), int(h["dst"]), ctx, url, entries)
                assert blob == want, f"tick {k} entity {e} full={fulls}: {len(blob)} vs {len(want)} bytes"
            n_total += 1
            n_group += len(members) == 2
            n_ctx += ctx != 0
        # and the recipients still line up with the handover list
        off, conn, kind = w.handover_recipients(nh)
        assert len(off) == nh + 1 and int(off[-1]) == len(conn) == len(kind)
    assert n_total > 100 and n_group > 5 and n_ctx > 5


@pytest.mark.parametrize("mode,dense", [(CONN_MAJOR, False), (DESC, True)])
def test_typed_merge_is_byte_identical_to_the_reference_marshal(amd, mode, dense):
    """VERDICT r2 #9 / SURVEY 8f-3: with CHD_MERGE_SCHEMA_TPS_ENTITY_MOVEMENT the accumulated update of a slow subscriber is
    merged FIELD BY FIELD on the device and marshalled as Go marshals it — proto.Merge of the first selected update, tpspb's
    EntityChannelData.Merge for the rest (data.go:249-253, tpspb/data.go:227-252), fmutils.Filter with the subscription's
    DataFieldMasks (data.go:294), fields in field-number order — not the concatenation of the selected updates.  Expected
    bytes: oracle/merge.py, pinned by python-protobuf on the reference's embedded descriptors (tests/test_merge_oracle.py).
    Updates outside the subset (here: one that carries an objRef) keep the generic, concatenated form."""
    from channeld_amd.engine import movement_field_mask
    from oracle import merge

    MASKS = 32
    cfg = synth.load_config("spatial_static_4x4.json")
    N, S = 400, 20
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE46, tick_ms=33))
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    # (DESC + dense: the typed merge inside the (mask, cell) images; subscriptions with DataFieldMasks keep the record path)
    w = amd.SpatialWorld(ctl, N, S, flags=mode | WIRE | MASKS, max_records=1 << 21, wire_max_update_len=96, wire_max_full_len=256)
    w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    w.wire_set_merge_schema(1)
    image_ranges = 0
    rng = np.random.default_rng(46)
    ncell = ctl.GridCols * ctl.GridRows
    url_e, url_c = b"type.googleapis.com/tpspb.EntityChannelData", b"type.googleapis.com/unrealpb.SpatialChannelData"
    w.wire_set_type_url(False, url_e)
    w.wire_set_type_url(True, url_c)
    full_e = {i: any_bytes(rng, rng.integers(20, 200)) for i in range(N)}
    full_c = {0x10000 + c: any_bytes(rng, rng.integers(20, 200)) for c in range(ncell)}
    w.wire_set_payloads(ENT_FULL, list(full_e), list(full_e.values()))
    w.wire_set_payloads(CELL_FULL, list(full_c), list(full_c.values()))
    foreign = bytes([0x0A, 0x02, 0x08, 0x05])  # objRef { netGUID: 5 }: outside the movement subset
    upd_e = {}
    fmask = {}  # (slot, spatial channel) -> data_field_mask
    typed2 = typed_masked = generic = 0
    for k in range(14):
        sw.step()
        idx = np.sort(rng.choice(N, N // 2, replace=False)).astype(np.uint32) if (k % 3 and not dense) else np.arange(N, dtype=np.uint32)
        pay = [foreign + merge.make_update(rng) if rng.random() < 0.03 else merge.make_update(rng) for _ in idx]
        for i, b in zip(idx, pay):
            upd_e[(int(i), k)] = b
        w.wire_set_payloads(ENT_UPD, idx, pay)
        res = w.tick(sw.now_ns(), upd_idx=idx, upd_x=sw.x[idx], upd_z=sw.z[idx], queries=sw.queries(), records_cap=1 << 21)
        assert res.overflow == 0 and res.history_overflow == 0
        nbytes, npackets, ndropped = w.wire_build()
        image_ranges += w.wire_build_info()[0]
        off, npk, data = w.wire_fetch()
        member = w.entity_state()[1]  # the spatial channel whose entity map holds each entity
        for s in range(S):
            a, n = int(res.conn_rec_off[s]), int(res.conn_rec_cnt[s])
            packs = []
            for r, mask in zip(res.records[a:a + n], res.record_masks[a:a + n]):
                ch, full = int(r["channel"]), int(r["conn"]) >> 31
                if full:
                    body = full_c[ch] if ch < 0x80000 else full_e[ch - 0x80000]
                else:
                    bits = [j for j in range(31, -1, -1) if (int(mask) >> j) & 1]  # oldest update first
                    if ch < 0x80000:
                        value = b""  # (no spatial channel updates in this world)
                        body = wire.field_bytes(1, url_c)
                    else:
                        e = ch - 0x80000
                        ups = [upd_e[(e, k - j)] for j in bits]
                        fm = fmask.get((s, int(member[e])), 0)
                        try:
                            value = merge.merged_update(ups, fm)
                            typed2 += len(ups) >= 2
                            typed_masked += fm != 0
                        except merge.NotInSubset:
                            value = b"".join(ups)
                            generic += 1
                        body = wire.field_bytes(1, url_e) + (wire.field_bytes(2, value) if value else b"")
                packs.append(wire.fanout_message_pack(ch, body))
            want, counts = wire.flush_stream(packs)
            got = data[int(off[s]):int(off[s + 1])].tobytes()
            assert got == want, f"tick {k} slot {s}: stream bytes ({len(got)} vs {len(want)})"
        if k == 2:
            # DataFieldMasks on every subscription of some connections (SUB_TO_CHANNEL with options; subscription.go:44-57 merges them)
            opts = []
            for s, paths in ((1, ["actorState.replicatedMovement.location"]), (4, ["actorState.replicatedMovement.location", "actorState.replicatedMovement.rotation",
                                                                                     "actorState.replicatedMovement.bRepPhysics"]),
                             (7, ["characterState"]), (9, ["actorState"])):
                m = movement_field_mask(paths)
                for c in w.subscriptions(s)[0]:
                    opts.append(dict(slot=s, channel=int(c), data_field_mask=m))
                    fmask[(s, int(c))] = m
            w.set_sub_options(sw.now_ns(), opts)
    assert typed2 > 1000 and typed_masked > 300 and generic > 20, (typed2, typed_masked, generic)
    assert (image_ranges > 200) == (mode == DESC), image_ranges
