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


def run_rank(rank, world, port, N, S, ticks, seed, out, cfg_name=None, halo=64, jump_frac=0.15, aoi_scale=1.0):
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
        eng = HipShardEngine(cfg, rank, world, N, max(len(my_subs), 1), migrate_cap=N, device=0, max_records=1 << 22)
        eng.spawn(sw.chan_id[mine], x0[mine], z0[mine], sw.flags[mine], sw.sender[mine])
        eng.add_subscribers(sw.sub_conn[my_subs])
        sworld = ShardedWorld(eng, Comm(rank, world))
        ow = None
        if rank == 0:
            ow = orc.World(g, N, S, eng.sw.capq, 20, 0, literal=False)
            ow.spawn(np.arange(N), sw.chan_id, x0, z0, sw.flags, sw.sender)
            for s in range(S):
                ow.add_sub(s, int(sw.sub_conn[s]))
        total = cross = 0
        for k, (x, z, q, now) in enumerate(frames):
            dq = torch.from_numpy(np.ascontiguousarray(q[my_subs]).view(np.uint8)).to(dev)
            sworld.tick(now, torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev), dq, len(my_subs))
            res = eng.fetch(want_records=True, records_cap=1 << 22)
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
            if rank != 0:
                continue
            ow.tick(now, None, x, z, None, None, None, None, q)
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
            us, uc = ow.unsubs()
            gu = canon(np.concatenate([s["unsub"][0] for s in gathered]), np.concatenate([s["unsub"][1] for s in gathered]))
            assert np.array_equal(gu, canon(sw.sub_conn[us], uc)), f"tick {k}: unsubs"
            ost = ow.query_status()
            for s in gathered:
                assert np.array_equal(s["status"], ost[s["subs"]]), f"tick {k}: query status"
            ocell, omember = ow.entity_state()
            to_id = lambda a: np.where(a == 0xFFFFFFFF, 0, a + 0x10000).astype(np.uint32)
            chans = np.concatenate([s["ent"][0] for s in gathered])
            assert len(chans) == N and len(np.unique(chans)) == N, f"tick {k}: entity lost or duplicated"
            for r, s in enumerate(gathered):
                i = (s["ent"][0] - 0x80000).astype(np.int64)
                assert np.array_equal(s["ent"][1], to_id(ocell)[i]) and np.array_equal(s["ent"][2], to_id(omember)[i])
                inw = s["ent"][2] != 0
                assert (server_of_cell(cfg, s["ent"][2][inw] - 0x10000) == r).all(), f"tick {k}: entity on the wrong rank"
            total += len(oc)
            cross += int((ssrc != sdst).sum())
        if rank == 0:
            out.put(("ok", total, cross))
    except Exception as e:
        import traceback

        out.put(("fail", f"rank {rank}: {e}\n{traceback.format_exc()}", 0))
        raise
    finally:
        if world > 1:
            dist.destroy_process_group()


def launch(world, N, S, ticks, seed, cfg_name=None, halo=64, jump_frac=0.15, aoi_scale=1.0):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=run_rank, args=(r, world, port, N, S, ticks, seed, out, cfg_name, halo, jump_frac, aoi_scale)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
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
    assert sharded["latency_ticks"] == 2 and sharded["collectives"]["ranks"] == 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    single = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert sharded["config"]["msgs_per_tick"] == pytest.approx(single["config"]["msgs_per_tick"], rel=0, abs=0.5)
