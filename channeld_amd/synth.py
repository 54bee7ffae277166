"""Deterministic synthetic worlds for parity tests and bench.py.

Everything is generated once on the host and fed unchanged to the oracle, the CPU
baseline and the GPU, so the generator's own libm calls cannot affect parity.
PRNG: SplitMix64 (counter form, vectorised), u = (next() >> 11) * 2**-53.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .gomath import go_cos

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
AOI_DTYPE = np.dtype([
    ("shapes", "<u4"), ("spot_off", "<u4"), ("n_spots", "<u4"), ("n_spot_dists", "<u4"),
    ("box_cx", "<f8"), ("box_cz", "<f8"), ("box_ex", "<f8"), ("box_ez", "<f8"),
    ("sph_cx", "<f8"), ("sph_cz", "<f8"), ("sph_r", "<f8"),
    ("cone_cx", "<f8"), ("cone_cz", "<f8"), ("cone_dx", "<f8"), ("cone_dz", "<f8"), ("cone_r", "<f8"), ("cone_cos", "<f8"),
    ("_reserved", "<f8"),
])
assert AOI_DTYPE.itemsize == 128
SHAPE_BOX, SHAPE_SPHERE, SHAPE_CONE = 2, 4, 8
CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")


class SplitMix64:
    def __init__(self, seed: int):
        self.state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def next_u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            k = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + GOLDEN * k
            self.state = self.state + GOLDEN * np.uint64(n)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def uniform(self, n: int) -> np.ndarray:
        return (self.next_u64(n) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def load_config(name: str) -> dict:
    path = name if os.path.exists(name) else os.path.join(CONFIG_DIR, name)
    with open(path) as f:
        d = json.load(f)
    return d.get("Config", d)


@dataclass
class WorldSpec:
    cfg: dict
    n_entities: int
    n_subs: int
    seed: int
    tick_ms: int = 50
    aoi_scale: float = 1.0          # 1.0: sphere R = 3 cells; secondary run 0.5
    outside_frac: float = 0.001
    locked_frac: float = 0.005


class SynthWorld:
    """Entity trajectories + per-tick AOI queries of the subscribers that follow them."""

    def __init__(self, spec: WorldSpec):
        self.spec = spec
        c = spec.cfg
        self.gw, self.gh = float(c["GridWidth"]), float(c["GridHeight"])
        self.offx, self.offz = float(c["WorldOffsetX"]), float(c["WorldOffsetZ"])
        self.cols, self.rows = int(c["GridCols"]), int(c["GridRows"])
        self.W, self.H = self.gw * self.cols, self.gh * self.rows
        self.rng = SplitMix64(spec.seed)
        N, S = spec.n_entities, spec.n_subs
        assert S <= N
        self.x = self._place(self.offx, self.W, N)
        self.z = self._place(self.offz, self.H, N)
        # a fixed fraction sits just outside the world (error path: never hands over)
        n_out = int(round(N * spec.outside_frac))
        self.outside = np.zeros(N, dtype=bool)
        if n_out:
            idx = (self.rng.next_u64(n_out) % np.uint64(N)).astype(np.int64)
            self.outside[idx] = True
            self.x[self.outside] = np.float64(np.float32(self.offx + self.W + 0.25 * self.gw))
        n_lock = int(round(N * spec.locked_frac))
        self.flags = np.zeros(N, dtype=np.uint32)
        if n_lock:
            idx = (self.rng.next_u64(n_lock) % np.uint64(N)).astype(np.int64)
            self.flags[idx] = 1
        self.chan_id = (np.uint32(0x80000) + np.arange(N, dtype=np.uint32)).astype(np.uint32)
        # sender = connection of the spatial server owning the spawn cell (never a subscriber)
        sgc = -(-self.cols // int(c["ServerCols"]))
        sgr = -(-self.rows // int(c["ServerRows"]))
        gx = np.clip(np.floor((self.x - self.offx) / self.gw), 0, self.cols - 1).astype(np.int64)
        gy = np.clip(np.floor((self.z - self.offz) / self.gh), 0, self.rows - 1).astype(np.int64)
        self.sender = (1 + gx // sgc + (gy // sgr) * int(c["ServerCols"])).astype(np.uint32)
        self.sub_conn = (np.uint32(1000) + np.arange(S, dtype=np.uint32)).astype(np.uint32)
        self.heading = 2.0 * math.pi * self.rng.uniform(N)
        # shape mix per subscriber: 70 % sphere, 20 % cone, 10 % box
        u = self.rng.uniform(S)
        self.shape = np.where(u < 0.7, SHAPE_SPHERE, np.where(u < 0.9, SHAPE_CONE, SHAPE_BOX)).astype(np.uint32)
        self.cone_cos = go_cos(0.5236)  # spatial_test.go:233
        self.tick_index = 0

    def _place(self, off, extent, n):
        v = (off + self.rng.uniform(n) * extent).astype(np.float32).astype(np.float64)
        hi = np.float64(off + extent)
        bad = v >= hi  # float32 rounding landed on the exclusive upper edge: nudge down one ulp
        v[bad] = np.nextafter(v[bad].astype(np.float32), np.float32(-np.inf)).astype(np.float64)
        lo_bad = v < off
        v[lo_bad] = np.nextafter(v[lo_bad].astype(np.float32), np.float32(np.inf)).astype(np.float64)
        return v

    def step(self):
        """Advance every in-world entity by one tick (SURVEY §8d motion model)."""
        N = self.spec.n_entities
        self.heading = 2.0 * math.pi * self.rng.uniform(N)
        dist = self.rng.uniform(N) * 0.02 * self.gw
        nx = self.x + dist * np.cos(self.heading)
        nz = self.z + dist * np.sin(self.heading)
        # reflect at the world borders
        lo, hi = self.offx, self.offx + self.W
        nx = np.where(nx < lo, 2 * lo - nx, nx)
        nx = np.where(nx >= hi, 2 * hi - nx, nx)
        lo, hi = self.offz, self.offz + self.H
        nz = np.where(nz < lo, 2 * lo - nz, nz)
        nz = np.where(nz >= hi, 2 * hi - nz, nz)
        nx = nx.astype(np.float32).astype(np.float64)
        nz = nz.astype(np.float32).astype(np.float64)
        edge = nx >= self.offx + self.W
        nx[edge] = np.nextafter(nx[edge].astype(np.float32), np.float32(-np.inf)).astype(np.float64)
        edge = nz >= self.offz + self.H
        nz[edge] = np.nextafter(nz[edge].astype(np.float32), np.float32(-np.inf)).astype(np.float64)
        keep = self.outside
        self.x = np.where(keep, self.x, nx)
        self.z = np.where(keep, self.z, nz)
        self.tick_index += 1

    def queries(self) -> np.ndarray:
        """AOI query of every subscriber at the current positions (AOI_DTYPE[S])."""
        S = self.spec.n_subs
        q = np.zeros(S, dtype=AOI_DTYPE)
        cx, cz = self.x[:S], self.z[:S]
        k = self.spec.aoi_scale
        q["shapes"] = self.shape
        sph, cone, box = self.shape == SHAPE_SPHERE, self.shape == SHAPE_CONE, self.shape == SHAPE_BOX
        q["sph_cx"][sph], q["sph_cz"][sph], q["sph_r"][sph] = cx[sph], cz[sph], 3.0 * self.gw * k
        q["cone_cx"][cone], q["cone_cz"][cone] = cx[cone], cz[cone]
        q["cone_dx"][cone], q["cone_dz"][cone] = np.cos(self.heading[:S][cone]), np.sin(self.heading[:S][cone])
        q["cone_r"][cone], q["cone_cos"][cone] = 5.0 * self.gw * k, self.cone_cos
        q["box_cx"][box], q["box_cz"][box] = cx[box], cz[box]
        q["box_ex"][box], q["box_ez"][box] = 2.0 * self.gw * k, 2.0 * self.gh * k
        return q

    def now_ns(self) -> int:
        return int(self.tick_index) * self.spec.tick_ms * 1_000_000


class ArrivalJitter:
    """The reference's arrival stamps for a synthetic world: Channel.PutMessage stamps an update when it is ENQUEUED
    (arrivalTime = ch.GetTime(), channel.go:296-310), the tick that handles the queue comes later — so a tick's updates carry
    stamps anywhere inside (previous tick, this tick].  next(t) takes the tick's nominal channel time and returns (now,
    arrivals[n]): now = t, or t off the tick grid by up to +-tick_jitter_us (a real gateway's ticks are never exactly periodic;
    subscriptions made at such a tick keep that phase, so every later fan-out window cuts through a tick's arrivals), arrivals
    uniform in (previous now, now].  Deterministic in (seed, call sequence): bench.py, tests/golden/make_bench_digests.py and the
    tests draw the same stamps."""

    def __init__(self, seed: int, n: int, tick_jitter_us: int = 0):
        self.rng = np.random.default_rng((int(seed) ^ 0x71773) & 0xFFFFFFFF)
        self.n, self.tick_jitter_us, self.prev = int(n), int(tick_jitter_us), 0

    def next(self, nominal_now_ns: int):
        now = int(nominal_now_ns)
        if self.tick_jitter_us:
            now += int(self.rng.integers(-self.tick_jitter_us, self.tick_jitter_us + 1)) * 1000 + int(self.rng.integers(0, 1000))
        assert now > self.prev
        arr = now - self.rng.integers(0, now - self.prev, self.n)
        self.prev = now
        return now, arr.astype(np.int64)
