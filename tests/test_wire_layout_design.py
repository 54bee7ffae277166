"""The arithmetic the lane-parallel wire layout of DESIGN.md §10.3 rests on, checked against the wire oracle's packer
(oracle/wire.py: flush_stream = connection.go:626-714) before any kernel is written on it:

  * lay the connection's entries out as if there were no packets (U = prefix sums of the entry lengths),
  * the packets' cuts are found one after the other — packet k starts at entry boundary c_k and ends at the largest
    boundary B with B - c_k <= 65535 (at least one entry) —, which is the only serial part,
  * every entry then knows its place without looking at its neighbours: stream offset = U + 5 * (cuts <= U), and a range of
    entries [a, b) is one contiguous copy unless a cut falls strictly inside (a, b), where it splits.

Host arithmetic only; no device."""
import bisect

import numpy as np
import pytest

from oracle import wire

MAXP = wire.MAX_PACKET_SIZE


def cuts_of(bounds):
    """bounds[i] = end of entry i in the uncut stream (ascending).  -> start boundaries of the packets [c_0 = 0, c_1, ...]"""
    cuts, c = [0], 0
    total = bounds[-1] if len(bounds) else 0
    while total - c > MAXP:
        k = bisect.bisect_right(bounds, c + MAXP) - 1
        nxt = bounds[k] if k >= 0 and bounds[k] > c else bounds[bisect.bisect_right(bounds, c)]  # (an entry always fits an empty packet)
        cuts.append(nxt)
        c = nxt
    return cuts


def lay_out(entries, pieces):
    """entries: list of byte strings (Packet.messages entries); pieces: list of (first, last) entry index ranges that are one
    source range each (a cell image's run of messages).  -> (stream bytes, copy ranges [(dst, src_entry_first, src_entry_last)])"""
    lens = np.array([len(e) for e in entries], dtype=np.int64)
    bounds = np.cumsum(lens).tolist()
    starts = [0] + bounds[:-1]
    cuts = cuts_of(bounds)
    total = bounds[-1] if bounds else 0
    out = bytearray(total + 5 * len(cuts) if total else 0)
    for k, c in enumerate(cuts):  # the tags
        end = cuts[k + 1] if k + 1 < len(cuts) else total
        n = end - c
        out[c + 5 * k: c + 5 * k + 5] = bytes([67, 72, (n >> 8) & 0xFF, n & 0xFF, 0])
    ranges = []
    for a, b in pieces:  # every piece on its own: no state carried from the piece before
        ua, ub = starts[a], bounds[b - 1]
        inside = [c for c in cuts if ua < c < ub]
        edges = [ua] + inside + [ub]
        for lo, hi in zip(edges[:-1], edges[1:]):
            dst = lo + 5 * bisect.bisect_right(cuts, lo)
            ea, eb = bisect.bisect_left(starts, lo), bisect.bisect_left(bounds, hi) + 1
            blob = b"".join(entries[ea:eb])
            assert len(blob) == hi - lo
            out[dst: dst + len(blob)] = blob
            ranges.append((dst, ea, eb))
    return bytes(out), ranges


@pytest.mark.parametrize("seed", range(6))
def test_cut_then_place_equals_the_sequential_packer(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 4000))
    big = seed % 3 == 0
    packs = [bytes(rng.integers(0, 256, int(rng.integers(20, 30000 if big and rng.random() < 0.02 else 400)), dtype=np.uint8)) for _ in range(n)]
    want, counts = wire.flush_stream(packs)
    entries = [wire.field_bytes(1, p) for p in packs]
    # pieces: random runs of entries (the per-window ranges of the cell images)
    edges = sorted(set([0, n] + [int(v) for v in rng.integers(0, n + 1, max(n // 50, 1))]))
    pieces = [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]
    got, ranges = lay_out(entries, pieces)
    assert got == want
    assert len(cuts_of(np.cumsum([len(e) for e in entries]).tolist())) == len(counts)
    # a piece is split only where a packet boundary falls strictly inside it
    assert len(ranges) <= len(pieces) + len(counts) - 1


def test_an_entry_that_ends_exactly_on_the_limit_stays_in_the_packet():
    entries = [bytes(60000), bytes(5535), bytes(1), bytes(65535), bytes(2)]
    bounds = np.cumsum([len(e) for e in entries]).tolist()
    assert cuts_of(bounds) == [0, 65535, 65536, 131071]
