#!/usr/bin/env python
"""Generates the committed golden vectors under tests/golden/.

Provenance: the reference (Go) cannot be built or imported in this image, so these
vectors are produced by the CPU oracle (oracle/chd_oracle.c) AFTER it has been pinned
against the reference's own test vectors (tests/test_oracle_golden.py transcribes
spatial_test.go / data_test.go).  They freeze the oracle's answers on seeded inputs so
that (a) an accidental change of the oracle is caught on CPU and (b) the GPU parity
tests compare against files, not only against a live oracle run.

    python tests/golden/make_golden.py      # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from channeld_amd import synth  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

GRIDS = {
    "2x2": "spatial_static_2x2.json", "4x1": "spatial_static_4x1.json", "benchmark": "spatial_static_benchmark.json",
    "4x4": "spatial_static_4x4.json", "8x8": "spatial_static_8x8.json",
}


def points(cfg, n, rng):
    gw, gh = float(cfg["GridWidth"]), float(cfg["GridHeight"])
    ox, oz = float(cfg["WorldOffsetX"]), float(cfg["WorldOffsetZ"])
    W, H = gw * cfg["GridCols"], gh * cfg["GridRows"]
    x = ox + (rng.random(n) * 1.2 - 0.1) * W
    z = oz + (rng.random(n) * 1.2 - 0.1) * H
    # exact cell edges, world edges, float32-representable values, specials
    k = n // 8
    x[:k] = ox + gw * rng.integers(0, cfg["GridCols"] + 1, k)
    z[k:2 * k] = oz + gh * rng.integers(0, cfg["GridRows"] + 1, k)
    x[2 * k:3 * k] = np.float64(np.float32(x[2 * k:3 * k]))
    z[2 * k:3 * k] = np.float64(np.float32(z[2 * k:3 * k]))
    sp = [np.nan, np.inf, -np.inf, np.finfo(np.float64).max, -np.finfo(np.float64).max, 0.0, -0.0,
          np.nextafter(ox, -np.inf), np.nextafter(ox + W, -np.inf), ox + W]
    x[3 * k:3 * k + len(sp)] = sp
    z[3 * k + len(sp):3 * k + 2 * len(sp)] = sp
    return x, z


def queries(cfg, n, rng):
    gw = float(cfg["GridWidth"])
    ox, oz = float(cfg["WorldOffsetX"]), float(cfg["WorldOffsetZ"])
    W, H = gw * cfg["GridCols"], float(cfg["GridHeight"]) * cfg["GridRows"]
    q = np.zeros(n, dtype=synth.AOI_DTYPE)
    kind = rng.integers(0, 4, n)
    cx = ox + (rng.random(n) * 1.1 - 0.05) * W
    cz = oz + (rng.random(n) * 1.1 - 0.05) * H
    r = gw * rng.choice([0.3, 0.5, 1.0, 1.5, 2.0, 3.0, 4.7], n)
    ang = rng.random(n) * 2 * np.pi
    half = rng.choice([0.1, 0.5236, np.pi / 4, 1.2], n)
    for i in range(n):
        if kind[i] in (0, 3):
            q["shapes"][i] |= synth.SHAPE_SPHERE
            q["sph_cx"][i], q["sph_cz"][i], q["sph_r"][i] = cx[i], cz[i], r[i]
        if kind[i] == 1:
            q["shapes"][i] |= synth.SHAPE_BOX
            q["box_cx"][i], q["box_cz"][i], q["box_ex"][i], q["box_ez"][i] = cx[i], cz[i], r[i], r[i] * 0.6
        if kind[i] in (2, 3):
            q["shapes"][i] |= synth.SHAPE_CONE
            q["cone_cx"][i], q["cone_cz"][i] = cx[i], cz[i]
            q["cone_dx"][i], q["cone_dz"][i] = np.cos(ang[i]), np.sin(ang[i])
            q["cone_r"][i], q["cone_cos"][i] = r[i] * 1.5, synth.go_cos(float(half[i]))
    return q


def run_queries(g, q):
    oq = orc.queries_from_aoi(q)
    off, ids, dists, status = [0], [], [], []
    for i in range(len(q)):
        b = orc.QueryBuilder()
        b.q = orc.Query.from_buffer_copy(oq[i].tobytes())
        rc, m = orc.query_channel_ids(g, b)
        status.append(rc)
        for c in sorted(m):
            ids.append(c)
            dists.append(m[c])
        off.append(len(ids))
    return (np.array(off, dtype=np.uint32), np.array(ids, dtype=np.uint32), np.array(dists, dtype=np.uint32),
            np.array(status, dtype=np.int32))


def world_trace(cfg_name, N, S, ticks, seed, tick_ms):
    cfg = synth.load_config(cfg_name)
    g = orc.grid_from_config(cfg)
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=tick_ms, outside_frac=0.01, locked_frac=0.02))
    ow = orc.World(g, N, S, min(g.cols * g.rows, 256), 20, 0, literal=True)
    ow.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    for s in range(S):
        ow.add_sub(s, int(sw.sub_conn[s]))
    out = {}
    for k in range(ticks):
        sw.step()
        ow.tick(sw.now_ns(), None, sw.x, sw.z, None, None, None, None, sw.queries())
        conn, chan = ow.records()
        out[f"rec_{k}"] = np.sort((conn.astype(np.uint64) << np.uint64(32)) | chan.astype(np.uint64))
        ent, src, dst, ssrc, sdst = ow.handovers()
        o = np.argsort(ent)
        out[f"ho_{k}"] = np.stack([ent[o], src[o], dst[o], ssrc[o], sdst[o]]).astype(np.uint32)
        us, uc = ow.unsubs()
        out[f"unsub_{k}"] = np.sort((us.astype(np.uint64) << np.uint64(32)) | uc.astype(np.uint64))
    return out


def main():
    rng = np.random.default_rng(20260922)
    pts, aoi = {}, {}
    for name, fn in GRIDS.items():
        cfg = synth.load_config(fn)
        g = orc.grid_from_config(cfg)
        x, z = points(cfg, 512, rng)
        pts[f"{name}_x"], pts[f"{name}_z"] = x, z
        pts[f"{name}_id"] = orc.channel_ids(g, x, z)
        q = queries(cfg, 160, rng)
        off, ids, dists, status = run_queries(g, q)
        aoi[f"{name}_q"] = q.view(np.uint8).reshape(len(q), -1)
        aoi[f"{name}_off"], aoi[f"{name}_ids"], aoi[f"{name}_dists"], aoi[f"{name}_status"] = off, ids, dists, status
    np.savez_compressed(os.path.join(HERE, "channel_ids.npz"), **pts)
    np.savez_compressed(os.path.join(HERE, "aoi_queries.npz"), **aoi)
    np.savez_compressed(os.path.join(HERE, "world_2x2_trace.npz"),
                        **world_trace("spatial_static_2x2.json", 400, 48, 8, 0xC0FFEE21, 50))
    np.savez_compressed(os.path.join(HERE, "world_4x4_trace.npz"),
                        **world_trace("spatial_static_4x4.json", 500, 40, 8, 0xC0FFEE22, 33))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
