"""CPU: the wire-format oracle (oracle/wire.py) against golden packets serialized with the
reference's own protobuf schema (tests/golden/wire_packets.npz, made by
tests/golden/make_wire_golden.py from the descriptor embedded in pkg/channeldpb/channeld.pb.go)."""
import os

import numpy as np
import pytest

from oracle import wire

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wire_packets.npz"))
CASES = sorted({k.rsplit("_", 1)[0] for k in G.files if k.endswith("_stream")})


def case(name):
    chan = G[f"{name}_chan"]
    lens = G[f"{name}_anylen"]
    blob = G[f"{name}_any"].tobytes()
    off = np.concatenate([[0], np.cumsum(lens.astype(np.int64))]).astype(np.int64).tolist()
    anys = [blob[off[i]:off[i + 1]] for i in range(len(chan))]
    return chan, anys, G[f"{name}_stream"].tobytes(), G[f"{name}_counts"].tolist()


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_packets(name):
    chan, anys, stream, counts = case(name)
    packs = [wire.fanout_message_pack(int(c), a) for c, a in zip(chan, anys)]
    got, got_counts = wire.flush_stream(packs)
    assert got_counts == counts
    assert got == stream


def test_fixture_covers_the_interesting_cases():
    assert len(case("split_many_packets")[3]) > 3                      # several packets
    chan, anys, stream, counts = case("oversized_dropped")
    assert sum(counts) < len(chan)                                     # some packs were dropped by Send's size check
    assert any(len(a) == 0 for a in case("empty_any")[1])              # a set-but-empty Any
    assert stream[:2] == b"CH" and stream[4] == 0                       # tag: 'C','H',hi,lo,compression
    n = (stream[2] << 8) | stream[3]
    assert n + 5 == len(stream)
