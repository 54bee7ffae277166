"""world_size-2 (and 4, 8) gloo runs of the region-sharded tick schedule on CPU.

The orchestration under test is channeld_amd/dist.py (ShardedWorld + Comm: all-to-all
of emigrants, all-gather of cell tables).  The engine is the numpy stand-in of
tests/shard_sim.py; the reference point is the single-world CPU oracle: after every
tick the union of the ranks' entity tables must equal the oracle's (cell, member) per
entity, every entity must live on exactly the rank that owns its member cell, and
every connection's visible set must equal {e : member(e) in interest(conn)} of the
single world."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from channeld_amd import synth  # noqa: E402
from channeld_amd.dist import Comm, ShardedWorld, server_layout, server_of_cell, weak_scaled_config  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_cfg(world, name=None, halo=64):
    """halo = ServerInterestBorderSize: how many cells of every neighbour's border a rank receives.  Connections are pinned
    to ranks while the entities they follow roam (world_inputs teleports 15 % of them per tick), so the default here is the
    whole world; tests of the band geometry pass a small halo with slowly drifting entities."""
    if name:  # one of the repo's StaticGrid2D configs whose server layout already has `world` regions
        cfg = dict(synth.load_config(name))
        assert int(cfg["ServerCols"]) * int(cfg["ServerRows"]) == world
        cfg["ServerInterestBorderSize"] = halo
        return cfg
    base = {"WorldOffsetX": -4000, "WorldOffsetZ": -4000, "GridWidth": 2000, "GridHeight": 2000, "GridCols": 3,
            "GridRows": 2, "ServerCols": 1, "ServerRows": 1, "ServerInterestBorderSize": halo}
    return weak_scaled_config(base, world)


def world_inputs(cfg, N, S, ticks, seed, jump_frac=0.15, aoi_scale=1.0):
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=50, outside_frac=0.01, locked_frac=0.03, aoi_scale=aoi_scale))
    frames = []
    x0, z0 = sw.x.copy(), sw.z.copy()
    rng = np.random.default_rng(seed & 0xFFFF)
    for _ in range(ticks):
        sw.step()
        # amplify the motion so that region borders are crossed often
        jump = rng.random(N) < jump_frac
        sw.x = np.where(jump & ~sw.outside, np.float64(np.float32(sw.offx + rng.random(N) * sw.W * 0.999)), sw.x)
        frames.append((sw.x.copy(), sw.z.copy(), sw.queries().copy(), sw.now_ns()))
    return sw, x0, z0, frames


def worker(rank, world, port, N, S, ticks, seed, out, cfg_name=None, halo=64, jump_frac=0.15, aoi_scale=1.0, lists=False):
    from oracle import pyoracle as orc
    from shard_sim import SimShardEngine

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = make_cfg(world, cfg_name, halo)
        sw, x0, z0, frames = world_inputs(cfg, N, S, ticks, seed, jump_frac, aoi_scale)
        g = orc.grid_from_config(cfg)
        ids0 = orc.channel_ids(g, x0, z0)
        owner = np.where(ids0 == 0, 0, server_of_cell(cfg, np.where(ids0 == 0, 0, ids0 - 0x10000)))
        mine = np.nonzero(owner == rank)[0]
        my_subs = np.nonzero(owner[:S] == rank)[0]
        eng = SimShardEngine(cfg, rank, world, N)
        eng.spawn(sw.chan_id[mine], x0[mine], z0[mine], sw.flags[mine])
        eng.add_subscribers(sw.sub_conn[my_subs])
        sworld = ShardedWorld(eng, Comm(rank, world))
        hl = None
        if lists:
            from shard_lists import make_lists, one_handover_per_group_and_tick

            hl, groups, _ = make_lists(sw, ids0, N, seed)
            assert len(hl) > N // 20
            frames = one_handover_per_group_and_tick(orc, g, groups, x0, z0, frames)
            eng.set_handover_lists({e + 0x80000: [m + 0x80000 for m in v] for e, v in hl.items()})  # (every rank: the whole world's)
        # the single world, on rank 0 only
        ow = None
        if rank == 0:
            ow = orc.World(g, N, S, min(g.cols * g.rows, 256), 20, 0, literal=False)
            ow.spawn(np.arange(N), sw.chan_id, x0, z0, sw.flags, sw.sender)
            for e, v in (hl or {}).items():
                ow.set_handover_list(e, v)
            for s in range(S):
                ow.add_sub(s, int(sw.sub_conn[s]))
        n_cross = 0
        for k, (x, z, q, now) in enumerate(frames):
            oq = orc.queries_from_aoi(q[my_subs])
            qb = []
            for i in range(len(my_subs)):
                b = orc.QueryBuilder()
                b.q = orc.Query.from_buffer_copy(oq[i].tobytes())
                qb.append(b)
            sworld.tick(now, torch.from_numpy(x), torch.from_numpy(z), qb, len(qb))
            state = dict(chan=eng.chan.copy(), cell=eng.cell.copy(), member=eng.member.copy(), visible=eng.visible,
                         handovers=eng.handovers, locked=eng.locked_aborts, requests=getattr(eng, "n_requests", 0))
            gathered = [None] * world
            dist.all_gather_object(gathered, state)
            if rank != 0:
                continue
            ow.tick(now, None, x, z, None, None, None, None, q)
            ocell, omember = ow.entity_state()
            chans = np.concatenate([s["chan"] for s in gathered])
            assert len(chans) == N and len(np.unique(chans)) == N, f"tick {k}: an entity was lost or duplicated"
            for r, s in enumerate(gathered):
                i = (s["chan"] - 0x80000).astype(np.int64)
                assert np.array_equal(s["cell"], ocell[i]) and np.array_equal(s["member"], omember[i]), f"tick {k} rank {r}"
                valid = s["member"] != 0xFFFFFFFF
                assert (server_of_cell(cfg, s["member"][valid]) == r).all(), f"tick {k}: entity on the wrong rank"
            ent, src, dst, ssrc, sdst = ow.handovers()
            got = sorted(h for s in gathered for h in s["handovers"])
            want = sorted((int(sw.chan_id[e]), int(a) - 0x10000, int(b) - 0x10000) for e, a, b in zip(ent, src, dst))
            assert got == want, f"tick {k}: handovers"
            if lists:
                assert sum(s["locked"] for s in gathered) == ow.locked_aborts(), f"tick {k}: aborted handovers"
                if k == len(frames) - 1:  # (the case the protocol exists for did occur: a handover whose src map is another rank's)
                    assert sum(s["requests"] for s in gathered) > 0, "no handover ever concerned another rank's entity map"
            n_cross += int((ssrc != sdst).sum())
            # visible sets
            member_of = omember
            for s_idx in range(S):
                cells, _, _, _, _ = ow.pairs(s_idx)
                want_vis = set(int(c) for c in sw.chan_id[np.isin(member_of, cells.astype(np.int64) - 0x10000)])
                conn = int(sw.sub_conn[s_idx])
                got_vis = next(st["visible"][conn] for st in gathered if conn in st["visible"])
                assert got_vis == want_vis, f"tick {k}: visible set of connection {conn}"
        if rank == 0:
            out.put(("ok", n_cross))
    except Exception as e:  # surface the failure in the parent
        import traceback

        out.put(("fail", f"rank {rank}: {e}\n{traceback.format_exc()}"))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_schedule_matches_single_world(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, 600, 48, 6, 0xC0FFEE07, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    status, info = out.get(timeout=5)
    assert status == "ok", info
    assert all(p.exitcode == 0 for p in procs)
    assert info > 0, "the test world never crossed a region border"


@pytest.mark.parametrize("world", [2, 4])
def test_handover_lists_across_ranks_follow_the_request_protocol(world):
    """chd_shard_set_handover_lists' protocol (k_shard.hip; include/chd_spatial.h: chd_shard_ingest_pre / _post) restated in
    numpy on the stand-in engine: lists keyed by entity channel id on every rank, members in src's entity map follow, a handover
    whose src map is another rank's travels there as a request before the export — against the single-world oracle with the same
    lists (entity.go:197-224, spatial.go:675-736): every entity's (cell, member, rank), handovers, aborted handovers, visible sets."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, 900, 32, 8, 0xC0FFEE31, out, None, 64, 0.15, 1.0, True)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    status, info = out.get(timeout=5)
    assert status == "ok", info
    assert all(p.exitcode == 0 for p in procs)
    assert info > 0, "the test world never crossed a region border"


@pytest.mark.parametrize("cfg_name,world", [("spatial_static_4x4.json", 4), ("spatial_static_8x8.json", 8)])
def test_named_configs_sharded_by_their_own_server_layout(cfg_name, world):
    """BASELINE configs 4 and 5 (scaled down): the 4x4 world over its 2x2 servers = 4 ranks, the seamless 8x8 world over
    its 4x2 servers = 8 ranks, handover all-to-all and table all-gather every tick, against the single world."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, 500, 40, 5, 0xC0FFEE0D, out, cfg_name)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    status, info = out.get(timeout=5)
    assert status == "ok", info
    assert all(p.exitcode == 0 for p in procs)
    assert info > 0, "the test world never crossed a region border"


def test_narrow_halo_on_the_40x40_grid():
    """spatial_static_40x40.json over its 4 x 2 servers (regions of 10 x 20 cells) with a halo of 4 cells: each rank receives
    only bands of its neighbours (the bench's geometry).  AOIs reach 3 cells and entities only drift, so every connection's
    interest stays inside region + halo and the sharded world still equals the single one record for record."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 8, port, 3000, 120, 5, 0xC0FFEE1A, out, "spatial_static_40x40.json", 4, 0.0, 0.6))
             for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    status, info = out.get(timeout=5)
    assert status == "ok", info
    assert all(p.exitcode == 0 for p in procs)


def test_layout_helpers():
    assert [server_layout(w) for w in (1, 2, 4, 8)] == [(1, 1), (2, 1), (2, 2), (4, 2)]
    base = synth.load_config("spatial_static_benchmark.json")
    cfg = weak_scaled_config(base, 8)
    assert (cfg["GridCols"], cfg["GridRows"], cfg["ServerCols"], cfg["ServerRows"]) == (60, 30, 4, 2)
    # the host-side routing arithmetic agrees with the oracle's GetRegions restatement
    from oracle import pyoracle as orc

    g = orc.grid_from_config(cfg)
    srv = orc.regions(g)[5]
    assert np.array_equal(server_of_cell(cfg, np.arange(60 * 30)), srv.astype(np.int64))
    for w in (1, 2, 4, 8):
        c = weak_scaled_config(base, w)
        counts = np.bincount(server_of_cell(c, np.arange(c["GridCols"] * c["GridRows"])), minlength=w)
        assert (counts == 225).all()


def test_bench_halo_covers_the_drift_of_pinned_connections():
    """bench.py --gpus N pins connection j to the rank that owns entity j's cell at the start, while the entity random-walks:
    its AOI (cones reach 5 cells) must stay inside region + halo for the whole run, or chd_tick_fetch fails with overflow
    bit 64.  Conservative bound (every shape reaches its full radius in every direction) over the default run length."""
    base = synth.load_config("spatial_static_benchmark.json")
    world = 2
    cfg = weak_scaled_config(base, world)
    halo = cfg["ServerInterestBorderSize"]
    N, S = 100_000 * world, 10_000 * world
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE01, tick_ms=50))
    cols, rows = cfg["GridCols"], cfg["GridRows"]
    sc, sr = server_layout(world)
    sgc, sgr = cols // sc, rows // sr
    gx, gy = np.floor((sw.x - sw.offx) / sw.gw), np.floor((sw.z - sw.offz) / sw.gh)
    inside = (gx >= 0) & (gx < cols) & (gy >= 0) & (gy < rows)
    owner = np.where(inside, server_of_cell(cfg, np.where(inside, gx + gy * cols, 0).astype(np.int64)), 0)[:S]
    rx0, ry0 = (owner % sc) * sgc, (owner // sc) * sgr
    R = np.where(sw.shape == synth.SHAPE_CONE, 5.0, np.where(sw.shape == synth.SHAPE_SPHERE, 3.0, 2.0)) * sw.gw
    worst = 0
    for t in range(20 + 200 + 100):  # bench defaults: warm-up + timed + latency ticks of the sharded run
        sw.step()
        x, z = sw.x[:S], sw.z[:S]
        ok = (x >= sw.offx) & (x < sw.offx + cols * sw.gw) & (z >= sw.offz) & (z < sw.offz + rows * sw.gh)
        lo_x = np.floor((np.maximum(x - R, sw.offx) - sw.offx) / sw.gw)
        hi_x = np.floor((np.minimum(x + R, sw.offx + cols * sw.gw - 1e-9) - sw.offx) / sw.gw)
        lo_z = np.floor((np.maximum(z - R, sw.offz) - sw.offz) / sw.gh)
        hi_z = np.floor((np.minimum(z + R, sw.offz + rows * sw.gh - 1e-9) - sw.offz) / sw.gh)
        need = np.maximum.reduce([rx0 - lo_x, hi_x - (rx0 + sgc - 1), ry0 - lo_z, hi_z - (ry0 + sgr - 1)])
        worst = max(worst, int(np.where(ok, need, 0).max()))
    assert 5 < worst <= halo, f"AOIs reach {worst} cells beyond the owner's region, halo = {halo}"


def _comm_helpers_rank(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        comm = Comm(rank, world)
        assert comm.backend == "gloo" and comm.staged
        assert comm.sum_int(10 + rank) == sum(10 + r for r in range(world))
        assert comm.max_float(0.5 * (rank + 1)) == 0.5 * world
        got = comm.gather_floats([float(rank), 2.0 * rank, -1.0])
        assert got == [[float(r), 2.0 * r, -1.0] for r in range(world)]
        comm.barrier()
        # equal-sized all-to-all: segment d of rank s = [100 s + d] * 3
        send = torch.tensor([[100 * rank + d] * 3 for d in range(world)], dtype=torch.int32)
        recv = comm.all_to_all(send)
        assert recv.tolist() == [[100 * s + rank] * 3 for s in range(world)]
        # all-to-all(v) of byte segments with static, uneven splits (zero to self): rank s sends rank d (s + d + 1) bytes of value 16 s + d
        send_splits = [0 if d == rank else rank + d + 1 for d in range(world)]
        recv_splits = [0 if s == rank else s + rank + 1 for s in range(world)]
        buf = torch.cat([torch.full((n,), 16 * rank + d, dtype=torch.uint8) for d, n in enumerate(send_splits)] + [torch.zeros(5, dtype=torch.uint8)])
        offs_of_peers = [sum((0 if d == s else s + d + 1) for d in range(rank)) for s in range(world)]  # where rank s keeps the segment for THIS rank
        ran = []
        got = comm.halo_exchange(buf, send_splits, recv_splits, offs_of_peers, overlap=lambda: ran.append(1))
        want = [16 * s + rank for s in range(world) for _ in range(recv_splits[s])]
        assert got.tolist() == want and ran == [1]
        if rank == 0:
            out.put("ok")
    except Exception as e:
        import traceback

        out.put(f"rank {rank}: {e}\n{traceback.format_exc()}")
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_comm_helpers_across_ranks(world):
    """Comm's reductions, gathers and the two exchange forms with more than one rank (gloo: host-staged; the same calls with
    device tensors are RCCL's on the GPU boxes, where only one rank is available to the tests)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_comm_helpers_rank, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    status = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
    assert status == "ok", status
    assert all(p.exitcode == 0 for p in procs)
