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


def _fields(buf):
    """(field number, wire type, value) of one serialized message (varint and length-delimited fields only)"""
    i, out = 0, []

    def rd():
        nonlocal i
        v, sh = 0, 0
        while True:
            b = buf[i]
            i += 1
            v |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                return v

    while i < len(buf):
        key = rd()
        if key & 7 == 0:
            out.append((key >> 3, 0, rd()))
        else:
            assert key & 7 == 2
            n = rd()
            out.append((key >> 3, 2, bytes(buf[i:i + n])))
            i += n
    return out


def test_encoders_reproduce_the_references_recorded_packets():
    """tests/golden/cpr_packs.npz: every MessagePack of the two client-packet recordings shipped with the reference
    (examples/replay/{webchat,tps}/*.cpr) - bytes the Go server's protobuf marshaller wrote - next to its decoded
    fields.  Re-encoding the fields with oracle/wire.py must give the recorded bytes: MessagePack for all 2637
    packs (user-space and system messages alike), ChannelDataUpdateMessage for the 39 CHANNEL_DATA_UPDATE bodies, and
    the Packet envelope for all 2634 packets."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cpr_packs.npz"))
    rows, blob, anyb = d["rows"], d["pack_bytes"].tobytes(), d["any_bytes"].tobytes()
    n_cdu = 0
    packs = []
    for ch, bc, stub, mt, off, ln, blen, aoff, alen, ctx, has_any in rows.tolist():
        rec = blob[off:off + ln]
        body = b"".join(v for n, w, v in _fields(rec) if (n, w) == (5, 2))
        assert len(body) == blen
        assert wire.message_pack(ch, mt, body, broadcast=bc, stub_id=stub) == rec
        if mt == wire.MSG_CHANNEL_DATA_UPDATE:
            n_cdu += 1
            assert has_any
            assert wire.channel_data_update(anyb[aoff:aoff + alen], ctx) == body
        packs.append(rec)
    assert n_cdu == 39 and len(packs) == 2637
    # Packet{messages = 1}: the envelope flush() marshals (connection.go:671)
    pk, po = d["packet_bytes"].tobytes(), d["packet_off"]
    k = 0
    for i in range(len(po) - 1):
        raw = pk[po[i]:po[i + 1]]
        n_msgs = len(_fields(raw))
        assert b"".join(wire.field_bytes(1, p) for p in packs[k:k + n_msgs]) == raw
        k += n_msgs
    assert k == len(packs)


def test_read_size_golden():
    """connection_test.go:71-93 (TestReadSize), transcribed."""
    tag = bytearray([0, 0, 0, 0])
    assert wire.read_size(tag) == 0
    tag[3] = 1
    assert wire.read_size(tag) == 0
    tag = bytearray([67, 0, 0, 0])
    assert wire.read_size(tag) == 0
    tag[3] = 1
    assert wire.read_size(tag) == 0
    tag = bytearray([67, 72, 78, 0])
    assert wire.read_size(tag) == 78 << 8
    tag[3] = 1
    assert wire.read_size(tag) == (78 << 8) + 1
    tag = bytearray([67, 72, 0, 0])
    assert wire.read_size(tag) == 0
    tag[3] = 1
    assert wire.read_size(tag) == 1


@pytest.mark.parametrize("name", CASES)
def test_streams_walk_with_the_receivers_rule(name):
    """what flush() writes, read back the way readPacket does (connection.go:455-541): tag -> size -> next tag"""
    stream = G[f"{name}_stream"].tobytes()
    counts = G[f"{name}_counts"].tolist() if f"{name}_counts" in G.files else None
    i, n_packets = 0, 0
    while i < len(stream):
        size = wire.read_size(stream[i:i + 4])
        assert size > 0 and stream[i + 4] == 0      # compression type NONE
        i += wire.PACKET_HEADER_SIZE + size
        n_packets += 1
    assert i == len(stream)
    if counts is not None:
        assert n_packets == len(counts)


def test_concatenated_updates_decode_as_their_merge():
    """What the device's merged-update messages rely on (include/chd_spatial.h): parsing the concatenation of
    serialized messages gives proto.Merge of them in order — last scalar wins, repeated fields append, sub-messages and
    map entries merge — which is how tickData accumulates the buffered updates (data.go:249-253, proto.Merge / the default
    ReflectMerge).  Checked with python-protobuf on a message type built here (scalars, a nested message, a repeated
    field, a map), over random update sequences."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    fd = descriptor_pb2.FileDescriptorProto(name="chd_merge_test.proto", package="chdtest", syntax="proto3")
    vec = fd.message_type.add(name="Vec")
    for i, n in enumerate(("x", "y", "z")):
        vec.field.add(name=n, number=i + 1, type=descriptor_pb2.FieldDescriptorProto.TYPE_FLOAT, label=1)
    st = fd.message_type.add(name="State")
    st.field.add(name="pos", number=1, type=11, type_name=".chdtest.Vec", label=1)
    st.field.add(name="health", number=2, type=descriptor_pb2.FieldDescriptorProto.TYPE_UINT32, label=1)
    st.field.add(name="name", number=3, type=descriptor_pb2.FieldDescriptorProto.TYPE_STRING, label=1)
    st.field.add(name="tags", number=4, type=descriptor_pb2.FieldDescriptorProto.TYPE_UINT32, label=3)
    ent = st.nested_type.add(name="AttrEntry")
    ent.options.map_entry = True
    ent.field.add(name="key", number=1, type=descriptor_pb2.FieldDescriptorProto.TYPE_UINT32, label=1)
    ent.field.add(name="value", number=2, type=11, type_name=".chdtest.Vec", label=1)
    st.field.add(name="attr", number=5, type=11, type_name=".chdtest.State.AttrEntry", label=3)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    State = message_factory.GetMessageClass(pool.FindMessageTypeByName("chdtest.State"))
    rng = np.random.default_rng(3)
    for trial in range(200):
        updates = []
        for _ in range(int(rng.integers(1, 5))):
            u = State()
            if rng.random() < 0.7:
                u.pos.x = float(rng.random())
                if rng.random() < 0.5:
                    u.pos.z = float(rng.random())
            if rng.random() < 0.4:
                u.health = int(rng.integers(0, 100))
            if rng.random() < 0.3:
                u.name = "n%d" % rng.integers(0, 9)
            u.tags.extend(int(v) for v in rng.integers(0, 9, int(rng.integers(0, 3))))
            for key in rng.integers(0, 4, int(rng.integers(0, 3))):
                u.attr[int(key)].y = float(rng.random())
            updates.append(u)
        acc = State()
        for u in updates:
            acc.MergeFrom(u)  # proto.Merge
        got = State()
        got.ParseFromString(b"".join(u.SerializeToString() for u in updates))
        assert got == acc, trial


def test_handover_message_oracle_matches_the_reference_schema():
    """oracle/wire.py's handover_message_pack (the restatement the GPU handover-message tests compare with) against
    MessagePack{ChannelDataHandoverMessage{Any{unrealpb.SpatialChannelData}}} serialized with the reference's own
    descriptors (tests/golden/make_handover_golden.py): 24 single-entity handovers with / without entityData, zero and
    multi-byte contextConnId and net ids, an empty objRef; and a two-entity (handover group) message."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "handover_msgs.npz"))
    url = g["type_url"].tobytes()

    def split(prefix):
        lens, blob = g[prefix + "_len"], g[prefix + "_bytes"].tobytes()
        off = np.concatenate([[0], np.cumsum(lens)]).astype(int)
        return [blob[off[i]:off[i + 1]] for i in range(len(lens))]

    objrefs, anys, packs = split("objref"), split("any"), split("pack")
    for i in range(len(packs)):
        state = wire.spatial_entity_state(objrefs[i], anys[i] if g["full"][i] else None)
        got = wire.handover_message_pack(int(g["src"][i]), int(g["dst"][i]), int(g["ctx"][i]), url, [(int(g["net"][i]), state)])
        assert got == packs[i], i
    grefs = split("group_objref")
    # (the order of map entries on the wire is unspecified — Go's is random, python's is its hash order: either is the message)
    ents = [(0x80010, wire.spatial_entity_state(grefs[0])), (0x80020, wire.spatial_entity_state(grefs[1]))]
    assert g["group_pack"].tobytes() in (wire.handover_message_pack(0x10001, 0x10002, 0, url, ents), wire.handover_message_pack(0x10001, 0x10002, 0, url, ents[::-1]))
    # a list of five of whom the destination connection already knows two: entityData for the other three only (the
    # per-(connection, entity) decision of spatial.go:797-857; chd_handover_variants builds these)
    mrefs, manys, mask = split("mixed_objref"), split("mixed_any"), int(g["mixed_mask"][0])
    ents = [(0x80100 + j, wire.spatial_entity_state(mrefs[j], manys[j] if (mask >> j) & 1 else None)) for j in range(5)]
    import itertools

    # (map entries: any order is the message — python-protobuf wrote this one back to front)
    assert any(g["mixed_pack"].tobytes() == wire.handover_message_pack(0x10005, 0x10006, 9, url, [ents[p] for p in perm])
               for perm in itertools.permutations(range(5)))
