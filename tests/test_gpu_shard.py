"""GPU parity of the region-sharded path (chd_shard_* through the C-ABI, orchestrated
by channeld_amd/dist.py) against the SINGLE-world CPU oracle: the union over ranks
of fan-out records, handover records, unsubs and entity placement must equal the
single world's, tick after tick.  The GPU box has one MI355X, so the 2-rank case runs
both ranks on device 0 with the gloo backend (exchange buffers staged through the
host); with RCCL the same schedule runs on device buffers (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def canon(conn, chan):
    return np.sort((conn.astype(np.uint64) << np.uint64(32)) | chan.astype(np.uint64))


from shard_lists import make_lists, one_handover_per_group_and_tick  # noqa: E402  (shared with the CPU gloo test)


def run_rank(rank, world, port, N, S, ticks, seed, out, cfg_name=None, halo=64, jump_frac=0.15, aoi_scale=1.0, lists=False, senders=False, exact=0,
             pipe=False, wflags=0, wire=False, cellupd=False, despawn=False, recipients=False):
    """exact != 0 (= the world's emit flags): exact update buffers on the sharded world — history_depth 1024, the update log by
    channel id on every rank (chd_world_cfg.shard_channels), per-update arrival stamps anywhere inside the tick's interval
    (chd_shard_set_update_arrivals), and three connections that lose access at tick 8 and get it back twelve ticks before the end:
    their catch-up walks the buffers of entities that have changed ranks many times since.
    pipe: the tick is ONE C call, chd_shard_tick, with the exchanges inside the library — over its TEST transport
    (CHD_SHARD_TRANSPORT=hostpipe: shared-memory mailboxes between the rank processes; RCCL refuses two ranks on one device), the
    unique id carried by gloo as a gateway's control connection would; wflags: world flags (16 | 512 = the gated overlap).
    wire: CHD_WORLD_WIRE on the sharded world — payloads keyed by channel id on every rank; every connection's byte stream must
    equal what oracle/wire.py makes of that connection's records (the rank's own, in its order) and the payloads.
    cellupd: the spatial channels' own updates (three random cells per tick, alternating senders, one of them a client connection
    that then skips its own), the same list on every rank.
    despawn: after tick 3 every seventh entity channel leaves the world (chd_shard_despawn, every rank the same list); after tick 6
    half of them come back where their channel's position now is (chd_shard_spawn on that rank, chd_shard_log_spawn everywhere).
    recipients: CHD_WORLD_HANDOVER_RECIPIENTS on every rank, one connection per region named as its spatial server's
    (chd_world_set_server_connections); after every tick the ranks' handover records are gathered into the whole-world list, every
    rank plans ITS connections' share (chd_shard_handover_recipients) and the union must be the single world's recipient list —
    connections, kinds, full-data masks — and the src servers' step-1 unsubscriptions (spatial.go:688-694)."""
    import torch
    import torch.distributed as dist

    from channeld_amd import synth
    from channeld_amd.dist import Comm, HipShardEngine, ShardedWorld, server_of_cell
    from oracle import pyoracle as orc
    from test_dist_gloo import make_cfg, world_inputs

    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        cfg = make_cfg(world, cfg_name, halo)
        sw, x0, z0, frames = world_inputs(cfg, N, S, ticks, seed, jump_frac, aoi_scale)
        g = orc.grid_from_config(cfg)
        ids0 = orc.channel_ids(g, x0, z0)
        owner = np.where(ids0 == 0, 0, server_of_cell(cfg, np.where(ids0 == 0, 0, ids0 - 0x10000)))
        mine = np.nonzero(owner == rank)[0]
        my_subs = np.nonzero(owner[:S] == rank)[0]
        if recipients:
            wflags |= 4
        eng = HipShardEngine(cfg, rank, world, N, max(len(my_subs), 1), migrate_cap=N, device=0, max_records=1 << 22,
                             **(dict(flags=exact | wflags, history_depth=1024, shard_channels=N) if exact else
                                dict(flags=wflags | 8 | 1, shard_channels=N) if wire else dict(flags=wflags)))
        if exact:
            eng.log_spawn(sw.chan_id, x0, z0)  # (every rank: every channel of the world)
        eng.spawn(sw.chan_id[mine], x0[mine], z0[mine], sw.flags[mine], sw.sender[mine])
        eng.add_subscribers(sw.sub_conn[my_subs])
        comm = Comm(rank, world)
        sworld = ShardedWorld(eng, comm)
        if pipe:
            os.environ["CHD_SHARD_TRANSPORT"] = "hostpipe"
            why = eng.comm_init_native(comm)
            assert why is None and eng.native, why
        if lists:
            hl, groups, (l_off, l_mem, l_chan, l_of) = make_lists(sw, ids0, N, seed)
            assert len(hl) > N // 20
            frames = one_handover_per_group_and_tick(orc, g, groups, x0, z0, frames)
            eng.set_handover_lists(l_off, l_mem, l_chan, l_of, N)  # (every rank: the same, whole-world arrays)
        ow = None
        if rank == 0:
            ow = orc.World(g, N, S, eng.sw.capq, 20, 0, literal=False)
            ow.spawn(np.arange(N), sw.chan_id, x0, z0, sw.flags, sw.sender)
            if lists:
                for e, m in hl.items():
                    ow.set_handover_list(e, m)
            for s in range(S):
                ow.add_sub(s, int(sw.sub_conn[s]))
        total = cross = 0
        n_rcp = n_own = n_changed = 0
        if recipients:  # the first connection pinned to rank k stands for spatial server k's (0 where a region has none)
            srv_conn = np.array([int(sw.sub_conn[np.nonzero(owner[:S] == k)[0][0]]) if (owner[:S] == k).any() else 0 for k in range(world)], dtype=np.uint32)
            eng.sw.set_server_connections(srv_conn)
            if ow is not None:
                ow.set_server_connections(srv_conn)
        pay = None
        if wire:  # the same payloads on every rank, keyed by channel index / spatial channel id (the host has them anyway)
            from oracle import wire as owire

            prng = np.random.default_rng(seed ^ 0x77)
            ncell = g.cols * g.rows
            blob = lambda lo, hi: bytes(prng.integers(0, 256, int(prng.integers(lo, hi)), dtype=np.uint8))
            pay = {"ent": {0: {}, 1: {u: blob(40, 300) for u in range(N)}},
                   "cell": {0: {0x10000 + c: blob(0, 100) for c in range(ncell)}, 1: {0x10000 + c: blob(40, 300) for c in range(ncell)}}}
            eng.sw.wire_set_payloads(1, list(pay["ent"][1]), list(pay["ent"][1].values()))
            eng.sw.wire_set_payloads(2, list(pay["cell"][0]), list(pay["cell"][0].values()))
            eng.sw.wire_set_payloads(3, list(pay["cell"][1]), list(pay["cell"][1].values()))
        d_snd = d_arr = None
        blocked = [b for b in (1, 4, 7) if b < S] if exact else []
        prev_now = 0
        if exact:
            d_arr = torch.zeros(N, dtype=torch.int64, device=dev)
            eng.set_update_arrivals(d_arr)
            if not senders:  # (an update log by channel takes its senders by channel id)
                d_snd = torch.from_numpy(sw.sender.astype(np.uint32).view(np.int32)).to(dev)
                eng.set_update_senders(d_snd)
        if senders:  # who sends an entity's updates changes over time: its spawn-time owner, then a CLIENT connection (which then skips its own)
            d_snd = torch.zeros(N, dtype=torch.int32, device=dev)
            eng.set_update_senders(d_snd)
        dead = set()
        for k, (x, z, q, now) in enumerate(frames):
            dead_before = len(dead)
            snd = None
            if senders:
                snd = np.where((np.arange(N) + k // 3) % 2 == 0, sw.sender, sw.sub_conn[np.arange(N) % S]).astype(np.uint32)
                d_snd.copy_(torch.from_numpy(snd.view(np.int32)))
            arr = None
            if exact and blocked:  # the blocked connections stop moving their AOI (they keep the subscriptions they lost access to)
                if k == 8:
                    frozen = q[blocked].copy()
                if k >= 8:
                    q = q.copy()
                    q[blocked] = frozen
            if exact:  # stamps at ENQUEUE time (channel.go:296-310): anywhere in (previous tick, this tick]; every 4th tick on the grid
                ra = np.random.default_rng((seed << 8) ^ k)
                arr = np.where((ra.random(N) < 0.4) | (k % 4 == 3), now, ra.integers(prev_now + 1, now + 1, N)).astype(np.int64)
                d_arr.copy_(torch.from_numpy(arr))
                if snd is None:
                    snd = sw.sender.astype(np.uint32)
            if wire:  # this tick's update payload of every channel of the world, on every rank
                urng = np.random.default_rng((seed << 8) ^ (k + 1000))
                upd = {u: bytes(urng.integers(0, 256, int(urng.integers(0, 100)), dtype=np.uint8)) for u in range(N)}
                pay["ent"][0].update(upd)
                eng.sw.wire_set_payloads(0, list(upd), list(upd.values()))
            cu = cus = cua = d_cu = None
            if cellupd:
                crng = np.random.default_rng((seed << 8) ^ (k + 7000))
                ncell_all = g.cols * g.rows
                cu = (0x10000 + crng.choice(ncell_all, min(3, ncell_all), replace=False)).astype(np.uint32)
                # (ring worlds keep two senders per channel inside the 32-tick horizon — a third is history_overflow, there as on one GPU —; exact worlds any number)
                cus = crng.choice([5, 6, int(sw.sub_conn[k % S])] if exact else [5, int(sw.sub_conn[0])], len(cu)).astype(np.uint32)
                if exact:
                    cua = np.sort(crng.integers(prev_now + 1, now + 1, len(cu))).astype(np.int64)
                d_cu = (torch.from_numpy(cu.view(np.int32)).to(dev), torch.from_numpy(cus.view(np.int32)).to(dev)) + ((torch.from_numpy(cua).to(dev),) if exact else ())
            dq = torch.from_numpy(np.ascontiguousarray(q[my_subs]).view(np.uint8)).to(dev)
            sworld.tick(now, torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev), dq, len(my_subs), cell_updates=d_cu)
            res = eng.fetch(want_records=True, records_cap=1 << 22)
            if wire:
                nbytes, npackets, ndropped = eng.sw.wire_build()
                off, npk, data = eng.sw.wire_fetch()
                assert int(off[len(my_subs)]) == nbytes == len(data) and ndropped == 0
                for ls in range(len(my_subs)):
                    packs = []
                    for r in res.records_of(ls):
                        full, ch = int(r["conn"]) >> 31, int(r["channel"])
                        packs.append(owire.fanout_message_pack(ch, pay["cell"][full][ch] if ch < 0x80000 else pay["ent"][full][ch - 0x80000]))
                    want, counts = owire.flush_stream(packs)
                    got = data[int(off[ls]):int(off[ls + 1])].tobytes()
                    assert got == want, f"rank {rank} tick {k} local slot {ls}: wire stream ({len(got)} bytes vs {len(want)})"
                    assert int(npk[ls]) == len(counts)
                wire_bytes = locals().get("wire_bytes", 0) + nbytes
            ch, cell, mem = eng.entities()
            state = dict(conn=res.records["conn"].copy(), chan=res.records["channel"].copy(), ho=res.handovers.copy(),
                         locked=res.n_locked_aborts, status=res.query_status.copy(), subs=my_subs,
                         unsub=(sw.sub_conn[my_subs][res.unsub_sub], res.unsub_channel.copy()),
                         ovf=(res.overflow, res.history_overflow), ent=(ch, cell, mem))
            if world > 1:
                gathered = [None] * world
                dist.all_gather_object(gathered, state)
            else:
                gathered = [state]
            rcp_all = None
            if recipients:
                ho_all = np.concatenate([s["ho"] for s in gathered])
                mine_rcp = eng.sw.shard_handover_recipients(ho_all)
                if world > 1:
                    rcp_all = [None] * world
                    dist.all_gather_object(rcp_all, mine_rcp)
                else:
                    rcp_all = [mine_rcp]
            if exact and (k == 8 or k == ticks - 12):  # the blocked connections: on the rank that holds them, and in the single world
                access = 0 if k == 8 else 1
                for b in blocked:
                    loc = np.nonzero(my_subs == b)[0]
                    if len(loc):
                        chs = eng.sw.subscriptions(int(loc[0]))[0]
                        eng.sw.set_sub_options(now, [dict(slot=int(loc[0]), channel=int(c), data_access=access) for c in chs])
            prev_now = now
            if despawn and k in (3, 6):  # (between two ticks, every rank alike)
                gone = np.arange(3, N, 7)
                if k == 3:
                    eng.despawn(sw.chan_id[gone])
                    dead = set(int(i) for i in gone)
                else:
                    back = gone[::2]
                    bid = orc.channel_ids(g, x[back], z[back])
                    bown = np.where(bid == 0, 0, server_of_cell(cfg, np.where(bid == 0, 0, bid - 0x10000)))
                    if exact:
                        eng.log_spawn(sw.chan_id[back], x[back], z[back])
                    m = back[bown == rank]
                    eng.spawn(sw.chan_id[m], x[m], z[m], sw.flags[m], sw.sender[m])
                    dead -= set(int(i) for i in back)
            if rank != 0:
                continue
            ow.tick(now, None, x, z, snd, None if cu is None else cu - 0x10000, cus, None, q,
                    **(dict(upd_arrival=arr, **({} if cua is None else dict(cu_arrival=cua))) if exact else {}))
            if exact and (k == 8 or k == ticks - 12):
                for b in blocked:
                    for c in ow.pairs(b)[0]:
                        ow.set_sub_options(now, b, int(c), data_access=0 if k == 8 else 1)
            assert all(s["ovf"] == (0, 0) for s in gathered)
            oc, och = ow.records()
            if os.environ.get("CHD_SHARD_DEBUG"):
                chans = np.concatenate([s["ent"][0] for s in gathered])
                u, cnt = np.unique(chans, return_counts=True)
                print(f"tick {k}: entities {len(chans)} unique {len(u)} dup {u[cnt > 1][:10]}", flush=True)
                gcd = canon(np.concatenate([s["conn"] for s in gathered]), np.concatenate([s["chan"] for s in gathered]))
                ocd = canon(oc, och)
                ug, cg = np.unique(gcd, return_counts=True)
                uo, co = np.unique(ocd, return_counts=True)
                extra = np.setdiff1d(ug, uo)
                print(f"   records got {len(gcd)} want {len(ocd)}; keys only in got: {len(extra)} e.g. {[hex(int(v)) for v in extra[:5]]};"
                      f" dup keys in got {int((cg > 1).sum())} in want {int((co > 1).sum())}", flush=True)
            gc = np.concatenate([s["conn"] for s in gathered])
            gch = np.concatenate([s["chan"] for s in gathered])
            assert len(gc) == len(oc), f"tick {k}: {len(gc)} records vs the single world's {len(oc)}"
            assert np.array_equal(canon(gc, gch), canon(oc, och)), f"tick {k}: fan-out records"
            ent, src, dst, ssrc, sdst = ow.handovers()
            ho = np.concatenate([s["ho"] for s in gathered])
            o1, o2 = np.argsort(ho["channel"]), np.argsort(sw.chan_id[ent])
            assert np.array_equal(ho["channel"][o1], sw.chan_id[ent][o2]), f"tick {k}: handover set"
            for f, want in (("src", src), ("dst", dst), ("src_server", ssrc), ("dst_server", sdst)):
                assert np.array_equal(ho[f][o1], want[o2]), f"tick {k}: handover {f}"
            assert sum(s["locked"] for s in gathered) == ow.locked_aborts()
            if recipients:
                oh, oconn, okind = ow.recipients()
                omask, oown = ow.recipient_masks(), ow.owner_unsubs()
                want = {}
                for h, c, kd, mk in zip(oh.tolist(), oconn.tolist(), okind.tolist(), omask.tolist()):
                    want.setdefault(int(sw.chan_id[ent[h]]), []).append((c, kd, mk))
                want_own = {int(sw.chan_id[ent[h]]): int(oown[h]) for h in range(len(ent))}
                for h in range(len(ho)):
                    got, own = [], 0
                    for (off, conn, kind, mask, ownf) in rcp_all:
                        a, b = int(off[h]), int(off[h + 1])
                        got += list(zip(conn[a:b].tolist(), kind[a:b].tolist(), mask[a:b].tolist()))
                        own += int(ownf[h])
                    chn = int(ho["channel"][h])
                    assert sorted(got) == sorted(want.get(chn, [])), f"tick {k}: recipients of the handover of channel {chn:#x}: {sorted(got)} vs {sorted(want.get(chn, []))}"
                    assert own == want_own[chn], f"tick {k}: src_owner_unsubscribed of the handover of channel {chn:#x}"
                    n_rcp += len(got)
                    n_own += own
                    n_changed += sum(1 for g_ in got if g_[1] == 2 and g_[2])
            us, uc = ow.unsubs()
            gu = canon(np.concatenate([s["unsub"][0] for s in gathered]), np.concatenate([s["unsub"][1] for s in gathered]))
            assert np.array_equal(gu, canon(sw.sub_conn[us], uc)), f"tick {k}: unsubs"
            ost = ow.query_status()
            for s in gathered:
                assert np.array_equal(s["status"], ost[s["subs"]]), f"tick {k}: query status"
            ocell, omember = ow.entity_state()
            to_id = lambda a: np.where(a == 0xFFFFFFFF, 0, a + 0x10000).astype(np.uint32)
            chans = np.concatenate([s["ent"][0] for s in gathered])
            n_alive = N - dead_before
            assert len(chans) == n_alive and len(np.unique(chans)) == n_alive, f"tick {k}: entity lost or duplicated"
            for r, s in enumerate(gathered):
                i = (s["ent"][0] - 0x80000).astype(np.int64)
                assert np.array_equal(s["ent"][1], to_id(ocell)[i]) and np.array_equal(s["ent"][2], to_id(omember)[i])
                inw = s["ent"][2] != 0
                assert (server_of_cell(cfg, s["ent"][2][inw] - 0x10000) == r).all(), f"tick {k}: entity on the wrong rank"
            total += len(oc)
            cross += int((ssrc != sdst).sum())
            if despawn and k in (3, 6):  # (the single world follows, between the same two ticks)
                if k == 3:
                    for i in gone:
                        ow.despawn(int(i))
                else:
                    ow.spawn(back, sw.chan_id[back], x[back], z[back], sw.flags[back], sw.sender[back])
        if rank == 0:
            out.put(("ok", total, (cross, n_rcp, n_own, n_changed) if recipients else cross))
    except Exception as e:
        import traceback

        out.put(("fail", f"rank {rank}: {e}\n{traceback.format_exc()}", 0))
        raise
    finally:
        if world > 1:
            dist.destroy_process_group()


def launch(world, N, S, ticks, seed, cfg_name=None, halo=64, jump_frac=0.15, aoi_scale=1.0, lists=False, senders=False, exact=0, timeout=300, pipe=False, wflags=0,
           wire=False, cellupd=False, despawn=False, recipients=False):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=run_rank, args=(r, world, port, N, S, ticks, seed, out, cfg_name, halo, jump_frac, aoi_scale, lists, senders, exact, pipe, wflags, wire, cellupd, despawn, recipients)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=timeout)
    status, total, cross = out.get(timeout=5)
    assert status == "ok", total
    assert all(p.exitcode == 0 for p in procs)
    return total, cross


def test_shard_api_single_rank_matches_single_world():
    total, _ = launch(1, 3000, 64, 8, 0xC0FFEE11)
    assert total > 0


def test_two_ranks_on_one_gpu_match_single_world():
    total, cross = launch(2, 4000, 96, 10, 0xC0FFEE12)
    assert total > 0 and cross > 0


def test_four_ranks_on_one_gpu_match_single_world():
    total, cross = launch(4, 4000, 96, 6, 0xC0FFEE13)
    assert total > 0 and cross > 0


def test_4x4_world_on_its_four_servers_matches_single_world():
    # BASELINE config 4 (scaled down): spatial_static_4x4.json, ServerCols x ServerRows = 2 x 2 -> four ranks
    total, cross = launch(4, 3000, 80, 6, 0xC0FFEE14, cfg_name="spatial_static_4x4.json")
    assert total > 0 and cross > 0


def test_8x8_world_on_its_eight_servers_matches_single_world():
    """BASELINE config 5's layout (spatial_static_8x8.json: 8x8 cells, ServerCols x ServerRows = 4 x 2) on the HIP
    engine: eight ranks (sharing the one GPU of the test box, exchange staged through gloo) against the single-world
    oracle, record for record."""
    total, cross = launch(8, 6000, 160, 6, 0xC0FFEE15, cfg_name="spatial_static_8x8.json")
    assert total > 0 and cross > 0


@pytest.mark.parametrize("world", [1, 2, 4])
def test_handover_lists_that_straddle_region_borders(world):
    """chd_shard_set_handover_lists (entity.go:197-224 / spatial.go:675-736 on region-sharded worlds): a quarter of the world in
    handover lists keyed by entity channel id — pairs, notifiers that are not in their own list, triples with a member in
    another cell (often another rank), empty lists — while 15 % of the entities teleport across the regions every tick.  List
    members follow their notifier out of src's entity map, emigrate with it when dst is another rank's, and a handover whose src
    map sits on another rank than the notifier travels there as a request (chd_shard_ingest_pre / _post).  Every rank's records,
    handovers, aborted handovers and every entity's (cell, member, rank) equal the single-world oracle's, tick for tick."""
    total, cross = launch(world, 4000, 96, 10, 0xC0FFEE20 + world, lists=True)
    assert total > 0 and (cross > 0 or world == 1)


@pytest.mark.parametrize("world", [1, 2])
def test_update_senders_change_while_entities_migrate(world):
    """chd_shard_set_update_senders: the sender of an entity's updates alternates between its spawn-time owner and a client
    connection (whose own fan-out then skips them: SkipSelfUpdateFanOut, data.go:242-245) while entities cross the region border
    — the two-sender history travels in the 32-byte emigrant state — against the single-world oracle fed the same senders."""
    total, cross = launch(world, 4000, 96, 12, 0xC0FFEE40 + world, senders=True)
    assert total > 0 and (cross > 0 or world == 1)


@pytest.mark.parametrize("world,flags,ticks,pipe", [(1, 1 | 64, 130, False), (2, 1 | 64, 130, False), (2, 1, 60, False), (4, 1 | 64, 32, True)],
                         ids=["1-rank-offsets", "2-ranks-offsets", "2-ranks-element-walk", "4-ranks-offsets-native-tick"])
def test_exact_update_buffers_and_enqueue_time_stamps_on_sharded_worlds(world, flags, ticks, pipe):
    """the reference stamps every update when it is ENQUEUED (channel.go:296-310) and tickData compares those stamps
    (data.go:225-269).  On a region-sharded world: history_depth 1024, the update log kept by channel id on every rank
    (chd_world_cfg.shard_channels — nothing of it travels with an emigrant or a border band), per-update stamps anywhere inside the
    tick's interval, 15 % of the entities teleporting across the regions every tick, and three connections that lose access at
    tick 8 and regain it 110 ticks later: their catch-up walks ~110 ticks of buffered updates of entities that changed ranks many
    times since.  Record for record the single-world oracle's (orc World.tick(upd_arrival=...)), history_overflow 0 on every rank.
    flags 1 | 64: the descriptor path with sub-tick offsets (ghost columns filled from the log); flags 1: every off-grid stamp
    makes its channel irregular and the element walk answers (ghost rings read through the log).  The four-rank world ticks through
    chd_shard_tick itself (the library's collectives over the hostpipe test transport: four processes staging their exchanges
    through gloo take ~2 s per tick on the one shared GPU) and regains access after 12 ticks, the element-walk case after 40, the
    two others after 110."""
    total, cross = launch(world, 700, 30, ticks, 0xC0FFEE50 + world + flags, exact=flags, timeout=900, pipe=pipe)
    assert total > 50_000 and (cross > 0 or world == 1)


@pytest.mark.parametrize("world,kw", [(2, dict(wflags=16 | 512)), (2, dict(lists=True)), (2, dict(exact=1 | 64, wflags=16 | 512)), (4, dict(wflags=16 | 512)),
                                      (2, dict(lists=True, exact=1 | 64, cellupd=True))],
                         ids=["2-ranks-gated", "2-ranks-lists", "2-ranks-exact-gated", "4-ranks-gated", "2-ranks-lists-exact-cell-updates"])
def test_chd_shard_tick_with_more_than_one_rank_over_the_hostpipe_transport(world, kw):
    """chd_shard_tick — the whole sharded tick as ONE C call with both exchanges inside the library — had
    only ever run with one rank (RCCL refuses two ranks on one device).  CHD_SHARD_TRANSPORT=hostpipe carries the same send / recv
    groups between the rank PROCESSES through shared-memory mailboxes (a test transport inside the library: same entry points, same
    call sequence, same buffers; a receiver that is sent another size than it expects fails loudly), so the things only a second
    rank exercises run here through the C entry point: the adaptive segment capacity every rank must derive alike from the maximum
    of two ticks ago, the request exchange in front of the export (handover lists), the interest updates on the second stream joined
    by the device-side gate while the peers' segments are still on their way, the update log by channel with the cells'
    maxFanOutIntervalMs behind every segment.  Records, handovers, unsubs and entity placement equal the single-world oracle's."""
    total, cross = launch(world, 4000 if "exact" not in kw else 900, 96 if "exact" not in kw else 30, 12 if world == 2 else 8, 0xC0FFEE60 + world + len(kw), pipe=True, timeout=600, **kw)
    assert total > 0 and cross > 0


@pytest.mark.parametrize("world,pipe", [(1, False), (2, False), (4, True)], ids=["1-rank", "2-ranks", "4-ranks-native-tick"])
def test_wire_buffers_on_region_sharded_worlds(world, pipe):
    """VERDICT r4 missing #2 (f1 x e): CHD_WORLD_WIRE on a region-sharded world.  connection.go:626-714 / data.go:293-308 build a
    connection's packets from the messages of whatever channels it is subscribed to — own region or a neighbour's border cell
    alike.  The entity payloads are keyed by CHANNEL ID (chd_world_cfg.shard_channels) and given to every rank, as the positions
    are, so a ghost entry of a neighbour's cell finds its payload where the stream is built; nothing new crosses the wire.
    Every connection's byte stream equals oracle/wire.py's flush of that connection's records (tag, greedy 65535-byte packets,
    MessagePack{channelId, msgType 8, ChannelDataUpdateMessage{Any}}) — and the records equal the single world's, as in every
    test of this file — while 15 % of the entities change regions every tick."""
    total, cross = launch(world, 1500, 48, 8, 0xC0FFEE70 + world, wire=True, pipe=pipe, timeout=600)
    assert total > 0 and (cross > 0 or world == 1)


@pytest.mark.parametrize("world,kw", [(2, dict()), (2, dict(exact=1 | 64)), (4, dict(pipe=True, wflags=16 | 512))], ids=["2-ranks", "2-ranks-exact", "4-ranks-native-tick-gated"])
def test_spatial_channels_own_updates_on_sharded_worlds(world, kw):
    """A spatial channel's own data changes too (its entity map: spawn, destroy, handover — ChannelData.OnUpdate on the spatial channel)
    and fans out to the cell's subscribers, who live on its owner's rank AND on the neighbours whose border cell it is.  The cells'
    update state is kept for every cell on every rank; every rank is given the tick's whole-world list (chd_shard_fanout /
    chd_shard_tick: d_in's cell-update fields).  Three random cells per tick, one of the senders a client connection (which skips its
    own update), on exact worlds with arrival stamps: records equal the single-world oracle's fed the same list."""
    total, cross = launch(world, 3000, 80, 10, 0xC0FFEE90 + world, cellupd=True, timeout=600, **kw)
    assert total > 0 and cross > 0


@pytest.mark.parametrize("world,kw", [(2, dict()), (4, dict()), (2, dict(exact=1 | 64)), (4, dict(pipe=True))], ids=["2-ranks", "4-ranks", "2-ranks-exact", "4-ranks-native-tick"])
def test_handover_recipients_on_region_sharded_worlds(world, kw):
    """VERDICT r5 #6 (f2 on sharded worlds): chd_shard_handover_recipients.  A handover's recipients are the connections subscribed to
    its src or dst cell — on the rank that owns the cell and on the neighbours whose halo it is in; every rank plans its own
    connections' share of the whole-world handover list on the subscriptions as they were at the tick's start, and the union over
    the ranks equals orc_world_recipients of the single world: connection, kind, full-data bit (new to the entity channel, or its
    DataAccess changes: the spatial servers' connections on cross-server handovers), plus the src server's step-1 unsubscription
    (spatial.go:688-694) on the rank that holds its connection."""
    total, (cross, n_rcp, n_own, n_changed) = launch(world, 2400, 160, 10, 0xC0FFEE71 + world, recipients=True, **kw)
    assert total > 0 and cross > 30 and n_rcp > 2000 and n_own > 5 and n_changed > 0, (total, cross, n_rcp, n_own, n_changed)


@pytest.mark.parametrize("world,kw", [(1, dict()), (2, dict()), (2, dict(exact=1 | 64))], ids=["1-rank", "2-ranks", "2-ranks-exact"])
def test_entity_channels_leave_and_come_back_on_sharded_worlds(world, kw):
    """chd_shard_despawn: every seventh entity channel is destroyed after tick 3 (whichever rank holds it frees its slot; with the
    update log by channel id the channel's log is closed), half of them are created again after tick 6 where their positions then are
    (possibly on another rank, with an EMPTY update buffer) — records, handovers, entity placement and the entity count over all
    ranks equal the single-world oracle's despawn / spawn every tick."""
    total, cross = launch(world, 3000, 80, 10, 0xC0FFEEA0 + world, despawn=True, timeout=600, **kw)
    assert total > 0 and (cross > 0 or world == 1)


def _disagreeing_rank(rank, port, out):
    """one rank of test_ranks_that_disagree...: rank 1 sizes its emigrant segments differently from rank 0"""
    import torch
    import torch.distributed as dist

    from channeld_amd import _lib
    from channeld_amd.dist import Comm, HipShardEngine
    from test_dist_gloo import make_cfg, world_inputs

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CHD_SHARD_TRANSPORT="hostpipe", CHD_HOSTPIPE_TIMEOUT_S="3")
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        cfg = make_cfg(2)
        sw, x0, z0, frames = world_inputs(cfg, 600, 16, 2, 0xC0FFEE80)
        eng = HipShardEngine(cfg, rank, 2, 600, 16, migrate_cap=256 if rank == 0 else 128, device=0, max_records=1 << 20)
        assert eng.comm_init_native(Comm(rank, 2)) is None
        x, z, q, now = frames[0]
        t0 = __import__("time").time()
        try:
            eng.tick_native(now, torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev), None, 0)
            eng.sync()
            out.put((rank, "no error", 0.0))
        except _lib.ChdError as e:
            out.put((rank, str(e), __import__("time").time() - t0))
    finally:
        dist.destroy_process_group()


def test_ranks_that_disagree_about_a_segment_size_fail_loudly_and_nobody_hangs():
    """Weak #13 of the round-4 verdict: "a disagreement after init would hang".  Two ranks whose emigrant segments differ in size
    (migrate_cap 256 against 128) over the hostpipe transport: the receiver of the wrong size says so, and the rank that is then left
    waiting gives up after CHD_HOSTPIPE_TIMEOUT_S instead of hanging — both chd_shard_tick calls return an error within seconds."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_disagreeing_rank, args=(r, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, msg, secs = out.get(timeout=120)
        got[r] = (msg, secs)
    for p in procs:
        p.join(timeout=60)
    assert all(m != "no error" and s < 20 for m, s in got.values()), got
    assert any("disagree about a segment size" in m for m, _ in got.values()), got


def test_narrow_halo_band_geometry_on_the_40x40_grid():
    """The halo as the bench uses it: ranks receive only a band of their neighbours' cells.  spatial_static_40x40.json (40 x
    40 cells) over its 4 x 2 servers (regions of 10 x 20 cells), halo = 4 cells (bands of the neighbours, corners included),
    AOIs scaled to reach at most 3 cells, entities drifting slowly (no teleports), so no connection's AOI leaves region +
    halo — the overflow flags stay 0 — and every record, handover and unsub still equals the single world's."""
    total, cross = launch(8, 6000, 200, 6, 0xC0FFEE16, cfg_name="spatial_static_40x40.json", halo=4, jump_frac=0.0, aoi_scale=0.6)
    assert total > 0


def test_rccl_single_rank_bench_path():
    """bench.py's sharded path exactly as the driver launches it (torch.distributed.run, backend "nccl" = RCCL,
    device buffers, everything on torch's stream) — with the one rank a one-GPU box has.  CHD_BENCH_FORCE_DIST makes
    the single rank issue the real collectives (all_to_all_single on int32, async all_gather_into_tensor on uint8,
    all_reduce, barrier), so dtype support, device_id initialisation and the stream hand-over are exercised; the
    message count must equal the unsharded path's on the same synthetic world."""
    import json
    import subprocess

    env = dict(os.environ, CHD_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--gpus", "1", "--steps", "6", "--warmup", "3", "--entities", "20000", "--subs", "2000", "--no-cpu", "--latency-steps", "0"]
    # (the sharded run also takes its latency phase — two synchronous ticks with stage events — as the driver's N > 1 runs do)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py")] + common[:-1] + ["2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    sharded = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert sharded["n_gpus"] == 1 and "tiled 1x1" in sharded["config"]["workload"]
    # the sharded bench checks its own first ticks against the single-world oracle (default --verify 2 on the dist path)
    # (tick 0 creates the subscriptions and fans nothing out yet; tick 1 carries the first, full-state fan-out)
    assert sharded["verified_ticks"] == 2 and sharded["verified"]["msgs_per_verified_tick"][1] > 0
    assert sharded["latency_ticks"] == 2 and sharded["collectives"]["ranks"] == 1
    # the collectives run INSIDE the library (chd_shard_comm_init + chd_shard_tick: ncclSend / ncclRecv groups of
    # librccl.so on the ctx stream), one C call per tick — and the four-stage path around torch.distributed gives the same ticks
    assert sharded["config"]["collectives_driver"].startswith("native: RCCL inside libchd_spatial.so")
    r = subprocess.run(cmd, env=dict(env, CHD_DIST_NATIVE="0"), capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    staged = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert staged["config"]["collectives_driver"].startswith("python:") and staged["verified_ticks"] == 2
    assert staged["config"]["msgs_per_tick"] == sharded["config"]["msgs_per_tick"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    single = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert sharded["config"]["msgs_per_tick"] == pytest.approx(single["config"]["msgs_per_tick"], rel=0, abs=0.5)
    # chd_shard_tick with the reference's stamp semantics — arrival stamps at enqueue time, exact update buffers kept by
    # channel id — through the library's own RCCL group (one rank), its first ticks verified against the single-world oracle
    r = subprocess.run(cmd + ["--arrival-jitter", "--verify", "4", "--warmup", "4"], env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    exact = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert exact["verified_ticks"] == 4 and exact["history_overflow"] == 0 and exact["ticks_with_overflow_flags"] == 0
    assert exact["config"]["collectives_driver"].startswith("native: RCCL inside libchd_spatial.so") and "shard_channels" in exact["config"]["update_buffers"]


def shared_gpu_bench(n, extra, timeout=420):
    """bench.py --gpus n as `python bench.py` launches itself (torch.distributed.run), the n ranks sharing the one GPU of the
    test box over gloo (host-staged exchanges)."""
    import json
    import subprocess

    env = dict(os.environ, CHD_BENCH_SHARE_GPU="1", CHD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--no-cpu", "--latency-steps", "0", "--max-records", "6000000"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_verify_two_ranks():
    """VERDICT r2 #1a: the multi-GPU bench line verifies itself — the first K ticks' digests, summed over the ranks, against
    the single-world oracle on rank 0 — and says so (`verified_ticks`)."""
    r, d = shared_gpu_bench(2, ["--steps", "4", "--warmup", "3", "--verify", "3", "--entities", "6000", "--subs", "400"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert d["verified_ticks"] == 3 and d["n_gpus"] == 2 and d["warmup"] == 3
    assert all(m > 0 for m in d["verified"]["msgs_per_verified_tick"][1:])
    assert d["config"]["config"] == "B-weak" and d["scaling"] == "weak"


def test_bench_arrival_jitter_two_ranks_verified():
    """bench.py --gpus N --arrival-jitter --verify K — the sharded bench with the reference's stamp semantics (every
    update stamped when it was enqueued, exact update buffers kept by channel id on every rank), its first ticks compared with the
    single-world oracle fed the same stamps; no history_overflow, no overflow flag in the timed ticks."""
    r, d = shared_gpu_bench(2, ["--arrival-jitter", "--steps", "4", "--warmup", "5", "--verify", "5", "--entities", "6000", "--subs", "400"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert d["verified_ticks"] == 5 and d["n_gpus"] == 2 and all(m > 0 for m in d["verified"]["msgs_per_verified_tick"][1:])
    assert d["history_overflow"] == 0 and d["ticks_with_overflow_flags"] == 0
    assert "enqueued" in d["config"]["arrival_stamps"] and d["config"]["filtered_msgs_per_tick"] > 0


def test_bench_verify_fails_loudly_on_a_wrong_world():
    """... and a difference ends the run with a non-zero exit code and no JSON line (here: the checker's world is fed one
    entity at a wrong position, CHD_BENCH_VERIFY_SABOTAGE)."""
    os.environ["CHD_BENCH_VERIFY_SABOTAGE"] = "1"
    try:
        r, d = shared_gpu_bench(2, ["--steps", "2", "--warmup", "2", "--verify", "2", "--entities", "6000", "--subs", "400"])
    finally:
        del os.environ["CHD_BENCH_VERIFY_SABOTAGE"]
    assert r.returncode != 0 and d is None
    assert "--verify FAILED" in r.stderr


def test_bench_config_d_on_its_four_servers_verified():
    """BASELINE config 4 AT ITS STATED SIZE through bench.py --config D: spatial_static_4x4.json, 100 000 entities / 10 000
    subscribers, 2x2 server regions = 4 ranks (sharing the test box's one GPU; populous cells: the cell-major emit over region +
    halo), the first ticks verified against the single world — ~0.45 G fan-out records per tick summed over the ranks."""
    r, d = shared_gpu_bench(4, ["--config", "D", "--steps", "3", "--warmup", "2", "--verify", "2", "--max-records", "400000000"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert d["verified_ticks"] == 2 and d["n_gpus"] == 4 and "spatial_static_4x4.json" in d["config"]["workload"]
    assert "100000 entities / 10000 subs" in d["config"]["workload"] and max(d["verified"]["msgs_per_verified_tick"]) > 250_000_000


def test_bench_config_d_with_enqueue_time_stamps_on_its_four_servers_verified():
    """BASELINE config 4 AT ITS STATED SIZE with the REFERENCE's stamp semantics (VERDICT r5 #1b): bench.py --config D --arrival-jitter —
    every update stamped when it was enqueued, exact update buffers by channel id on every rank — 4 ranks sharing the test box's GPU,
    the first ticks against the single-world oracle fed the same stamps.  Its cells hold ~6 K entities: before DESIGN 13.8c's fix such a
    world ran the serial element walk; the line's own counters say it does not."""
    r, d = shared_gpu_bench(4, ["--config", "D", "--arrival-jitter", "--steps", "2", "--warmup", "2", "--verify", "2", "--max-records", "1600000000"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert d["verified_ticks"] == 2 and d["n_gpus"] == 4 and "100000 entities / 10000 subs" in d["config"]["workload"]
    assert d["history_overflow"] == 0 and d["ticks_with_overflow_flags"] == 0
    assert d["config"]["filtered_msgs_per_tick"] > 0 and d["config"].get("element_walk_msgs_per_tick", 0) == 0, d["config"]


def test_bench_config_e_at_its_stated_size_against_the_committed_oracle_results():
    """BASELINE config 5 AT ITS STATED SIZE (VERDICT r5 #1c): 8 x 8 world, 1 M entities / 100 K subscribers on its 4 x 2 server regions = 8
    ranks sharing the test box's GPU, tick-aligned stamps, AOI x 0.5 — the first fan-out alone is 9 G records.  The single world's
    oracle takes 80-100 s per tick, so its results are COMMITTED (tests/golden/bench_digests_E.json, make_shard_golden.py) and the bench
    compares with those: all records' {count, sum, xor}, the fold of every connection's digest, handover / abort / unsub counts.
    (E with exact update buffers stays at half size — profiles/r06j_*: eight ranks' 12 GiB update logs + worst-case filtered segments
    exceed ONE GPU's 288 GB; eight GPUs hold it.)"""
    golden = os.path.join(ROOT, "tests", "golden", "bench_digests_E.json")
    r, d = shared_gpu_bench(8, ["--config", "E", "--steps", "1", "--warmup", "2", "--verify", "2", "--verify-golden", golden, "--max-records", "2600000000"], timeout=1100)
    assert r.returncode == 0, r.stderr[-3000:]
    assert d["verified_ticks"] == 2 and d["n_gpus"] == 8 and "1000000 entities / 100000 subs" in d["config"]["workload"]
    assert max(d["verified"]["msgs_per_verified_tick"]) > 9_000_000_000 and "bench_digests_E.json" in d["verified_against"]


def test_bench_config_e_on_its_eight_servers_verified():
    """BASELINE config 5's world through bench.py --config E (scaled down to fit eight ranks on one test GPU): 8x8 cells,
    4x2 server regions, emigrant all-to-all + halo all-to-all(v) between eight ranks, verified against the single world."""
    r, d = shared_gpu_bench(8, ["--config", "E", "--steps", "3", "--warmup", "2", "--verify", "2", "--entities", "24000", "--subs", "800"], timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    assert d["verified_ticks"] == 2 and d["n_gpus"] == 8 and d["config"]["aoi_scale"] == 0.5


def test_immigrants_without_a_slot_wait_in_limbo_and_come_back():
    """ADVICE r1 / VERDICT r2 #1d: an immigrant that finds no free slot on its new rank is not lost.  Two ranks (two contexts
    on the one GPU, exchanges done by hand), 200 slots each, 150 entities each; 100 entities of rank 0 walk into rank 1's
    region: 50 of them wait (overflow bit 16, every tick they wait), nothing else is disturbed; when 80 of rank 1's entities
    leave, the 50 take slots and the world is whole again."""
    import torch

    from channeld_amd.dist import ENTITY_STATE_WORDS, HipShardEngine
    from test_dist_gloo import make_cfg

    dev = torch.device("cuda", 0)
    cfg = make_cfg(2)  # 6 x 2 cells of 2000, two regions of 3 x 2; x in [-8000? ...): see below
    cols, gw, offx = int(cfg["GridCols"]), float(cfg["GridWidth"]), float(cfg["WorldOffsetX"])
    offz = float(cfg["WorldOffsetZ"])
    half = offx + gw * cols / 2  # x < half: rank 0's region
    N = 300
    rng = np.random.default_rng(5)
    x = np.where(np.arange(N) < 150, offx + 100 + rng.random(N) * (gw * cols / 2 - 200), half + 100 + rng.random(N) * (gw * cols / 2 - 200))
    z = offz + 100 + rng.random(N) * 3000
    chan = (0x80000 + np.arange(N)).astype(np.uint32)
    zeros = np.zeros(N, dtype=np.uint32)
    engs = [HipShardEngine(cfg, r, 2, 200, 4, migrate_cap=256, device=0, max_records=1 << 20) for r in range(2)]
    for r, e in enumerate(engs):
        m = (np.arange(N) < 150) == (r == 0)
        e.spawn(chan[m], x[m], z[m], zeros[m], zeros[m] + 1)

    def tick(k, x):
        dx, dz = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
        sends = [e.ingest(k * 50_000_000, dx, dz) for e in engs]
        halos = [e.import_(torch.stack([sends[0][r], sends[1][r]]).contiguous()) for r, e in enumerate(engs)]
        for r, e in enumerate(engs):
            _, recv_splits, peer_off = e.halo_splits()
            parts = [halos[p][peer_off[p]: peer_off[p] + recv_splits[p]] for p in range(2)]
            e.interest(None, 0)
            e.fanout(torch.cat(parts) if sum(recv_splits) else halos[r])
        res = [e.fetch(check=False) for e in engs]
        present = np.concatenate([e.entities()[0] for e in engs])
        return [int(v.overflow) for v in res], present

    ovf, present = tick(1, x)
    assert ovf == [0, 0] and len(present) == N
    x2 = x.copy()
    x2[:100] = half + 500  # 100 entities of rank 0 cross into rank 1's region: 250 > 200 slots
    ovf, present = tick(2, x2)
    assert ovf[0] == 0 and ovf[1] & 16 and not ovf[1] & 128
    assert len(present) == 250 and len(np.unique(present)) == 250
    ovf, present = tick(3, x2)
    assert ovf[1] & 16 and len(present) == 250  # still waiting, still flagged
    x3 = x2.copy()
    x3[150:230] = half - 500  # 80 of rank 1's entities leave for rank 0: slots free up, the 50 come back
    ovf, present = tick(4, x3)
    assert ovf == [0, 0], ovf
    assert len(present) == N and len(np.unique(present)) == N
    ovf, present = tick(5, x3)
    assert ovf == [0, 0] and len(present) == N


def test_a_world_with_by_channel_arrays_refuses_the_slot_indexed_entry_points():
    """chd_world_cfg.shard_channels sizes the update log and the payload tables by CHANNEL; chd_world_spawn,
    chd_tick and chd_tick_device index them by entity SLOT and would write past them whenever shard_channels < max_entities.  Such a
    world is region-sharded from its creation on: the three answer CHD_E_STATE, and chd_shard_import wants a fresh chd_shard_ingest
    (it would otherwise log the last tick's updates a second time)."""
    import ctypes as C
    import json

    import channeld_amd as amd
    from channeld_amd import _lib, synth

    cfg = synth.load_config("spatial_static_4x4.json")
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    w = amd.SpatialWorld(ctl, 512, 64, max_records=1_000_000, history_depth=64, shard_channels=128)
    lib, ctx = ctl._lib, ctl.ctx
    u32p, f64p = C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    chan = np.arange(0x80000, 0x80000 + 8, dtype=np.uint32)
    x = np.zeros(8); z = np.zeros(8)
    rc = lib.chd_world_spawn(ctx, 8, None, chan.ctypes.data_as(u32p), x.ctypes.data_as(f64p), z.ctypes.data_as(f64p), None, None)
    assert rc == _lib.E_STATE, rc
    tin = _lib.TickIn()
    tin.now_ns = 1_000_000
    assert lib.chd_tick_device(ctx, C.byref(tin)) == _lib.E_STATE
    tout = _lib.TickOut()
    assert lib.chd_tick(ctx, C.byref(tin), C.byref(tout)) == _lib.E_STATE
    # no ingest yet: import refuses
    assert lib.chd_shard_import(ctx, None, 1, 256, None) == _lib.E_STATE
    # a channel id outside the log is reported by the call itself, whatever the tick's overflow mask holds
    bad = np.array([0x80000 + 128], dtype=np.uint32)
    rc = lib.chd_shard_log_spawn(ctx, 1, bad.ctypes.data_as(u32p), x.ctypes.data_as(f64p), z.ctypes.data_as(f64p))
    assert rc == _lib.E_CAPACITY, rc
    rc = lib.chd_shard_log_spawn(ctx, 8, chan.ctypes.data_as(u32p), x.ctypes.data_as(f64p), z.ctypes.data_as(f64p))
    assert rc == 0, rc
    del w
    ctl.close()
