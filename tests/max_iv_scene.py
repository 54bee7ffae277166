"""A world that tells a per-CHANNEL maxFanOutIntervalMs from a world-global one (subscription.go:83-86, data.go:165-171):

spatial_static_2x2: entities 0..3 stand in cell 0, entities 4..7 in cell 3.  Connection 0 subscribes cell 0 at 20 ms, connection 1
cell 3 at 100 ms, connection 2 cell 0 at 20 ms — and loses access for 28 ticks.  Ticks are 10 ms apart; the entities of cell 0
take 120 updates per tick (update rounds: one buffer element each), those of cell 3 one.  Once a cell-0 channel holds more than
512 elements every push evicts its oldest if it is older than the CHANNEL's maximum, 20 ms: the buffer stays at 512 elements =
~43 ms.  Under one world-wide maximum (100 ms, from cell 3's subscriber) it would hold ~100 ms.  When connection 2 gets its
access back, its catch-up windows older than ~43 ms are therefore EMPTY in the reference — no message — and full under the
global rule: the record counts differ by the windows between."""
import numpy as np

MS = 1_000_000
N, S, ROUNDS, TICKS, BLOCK_AT, REGAIN_AT = 8, 3, 120, 34, 2, 30


def positions(cfg):
    ox, oz, gw_, gh = float(cfg["WorldOffsetX"]), float(cfg["WorldOffsetZ"]), float(cfg["GridWidth"]), float(cfg["GridHeight"])
    x = np.array([ox + 0.2 * gw_ + 10.0 * i for i in range(4)] + [ox + 1.3 * gw_ + 10.0 * i for i in range(4)])
    z = np.array([oz + 0.3 * gh] * 4 + [oz + 1.4 * gh] * 4)
    return x, z


def subscriptions():
    c0, c3 = 0x10000, 0x10000 + 3
    return [dict(slot=0, channel=c0, fanout_interval_ms=20), dict(slot=1, channel=c3, fanout_interval_ms=100),
            dict(slot=2, channel=c0, fanout_interval_ms=20)]


def tick_inputs(k, x, z):
    """(now, idx, x, z, arrival, round_off) of tick k: round 0 = all eight entities, rounds 1..119 = entities 0..3 again"""
    prev, now = k * 10 * MS, (k + 1) * 10 * MS
    idx = [np.arange(N, dtype=np.uint32)] + [np.arange(4, dtype=np.uint32)] * (ROUNDS - 1)
    arr = [np.full(N, prev + (10 * MS) // ROUNDS, dtype=np.int64)]
    for r in range(1, ROUNDS):
        arr.append(np.full(4, prev + (r + 1) * (10 * MS) // ROUNDS, dtype=np.int64))
    off = np.concatenate([[0], np.cumsum([len(i) for i in idx])]).astype(np.uint32)
    idx = np.concatenate(idx)
    return now, idx, x[idx], z[idx], np.concatenate(arr), off
