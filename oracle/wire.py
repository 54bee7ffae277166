"""CPU restatement of the server -> client wire format of the fan-out path
(TEST INFRASTRUCTURE ONLY, see chd_oracle.h).

What the reference does for one fan-out message (paths relative to the channeld tree):
  data.go:293-318        fanOutDataUpdate: ChannelDataUpdateMessage{Data: anypb.New(update)},
                         MessageContext{MsgType: CHANNEL_DATA_UPDATE (8), ChannelId: ch.id,
                         Broadcast: 0, StubId: 0}
  connection.go:57-83    queuedMessagePackSender.Send: MessagePack{ChannelId, Broadcast, StubId,
                         MsgType, MsgBody: proto.Marshal(msg)}; dropped if proto.Size(mp) >=
                         MaxPacketSize - PacketHeaderSize (65535 - 5)
  connection.go:626-714  flush: MessagePacks are appended to one Packet until proto.Size(packet)
                         would exceed MaxPacketSize; the offending pack opens the next packet;
                         tag = {'C','H', size_hi, size_lo, compressionType} + proto.Marshal(packet)

Third-party dependency restated: google.golang.org/protobuf v1.28.1 (go.mod:14) — the standard
proto3 wire format (varint field keys, length-delimited submessages, zero-valued scalar fields
omitted, fields in field-number order).  Pinned against python-protobuf driven by the reference's
OWN embedded descriptor (tests/golden/make_wire_golden.py -> tests/golden/wire_packets.npz) and against the
MessagePacks the Go server itself marshalled into the recordings under examples/replay/ (make_cpr_golden.py ->
cpr_packs.npz).
"""
from typing import Iterable, List, Tuple

MAX_PACKET_SIZE = 0xFFFF      # connection.go:27
PACKET_HEADER_SIZE = 5        # connection.go:28
MSG_CHANNEL_DATA_UPDATE = 8   # channeld.proto:121


def varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def field_bytes(num: int, payload: bytes) -> bytes:
    return varint((num << 3) | 2) + varint(len(payload)) + payload


def field_varint(num: int, v: int) -> bytes:
    return b"" if v == 0 else varint(num << 3) + varint(v)  # proto3: zero scalars are not emitted


def channel_data_update(any_bytes: bytes, context_conn_id: int = 0) -> bytes:
    """ChannelDataUpdateMessage (channeld.proto:333-340): data = 1 (Any), contextConnId = 2.
    `any_bytes` is the serialized google.protobuf.Any; a set-but-empty Any still emits the field."""
    return field_bytes(1, any_bytes) + field_varint(2, context_conn_id)


def message_pack(channel_id: int, msg_type: int, body: bytes, broadcast: int = 0, stub_id: int = 0) -> bytes:
    """MessagePack (channeld.proto:15-34)."""
    return (field_varint(1, channel_id) + field_varint(2, broadcast) + field_varint(3, stub_id)
            + field_varint(4, msg_type) + (field_bytes(5, body) if body else b""))


def fanout_message_pack(channel_id: int, any_bytes: bytes) -> bytes:
    return message_pack(channel_id, MSG_CHANNEL_DATA_UPDATE, channel_data_update(any_bytes))


MSG_CHANNEL_DATA_HANDOVER = 12  # channeld.proto:133


def spatial_entity_state(obj_ref: bytes, entity_data_any: bytes = None) -> bytes:
    """unrealpb.SpatialEntityState (unreal_common.proto:138-143): objRef = 1 (always set by MergeTo, tpspb/data.go:329-331),
    removed = 2 (false: not emitted), entityData = 3 (the Any of the full entity channel data, only when fullData)."""
    return field_bytes(1, obj_ref) + (field_bytes(3, entity_data_any) if entity_data_any is not None else b"")


def spatial_channel_data(entries) -> bytes:
    """unrealpb.SpatialChannelData{entities: map<uint32, SpatialEntityState>} (:145-147): one map-entry sub-message
    {key = 1, value = 2} per (net id, SpatialEntityState bytes), in the order given (Go's map order is random; any order
    decodes to the same message)."""
    return b"".join(field_bytes(1, field_varint(1, net_id) + field_bytes(2, state)) for net_id, state in entries)


def handover_message_pack(src: int, dst: int, context_conn_id: int, type_url: bytes, entries) -> bytes:
    """The MessagePack Notify sends for one handover (spatial.go:738-773,797-857): ChannelDataHandoverMessage{srcChannelId = 1,
    dstChannelId = 2, contextConnId = 3, data = 4: Any(SpatialChannelData)} (channeld.proto:416-425) inside
    MessageContext{MsgType: CHANNEL_DATA_HANDOVER, ChannelId: dstChannelId} (:763-773)."""
    sd = spatial_channel_data(entries)
    any_bytes = (field_bytes(1, type_url) if type_url else b"") + (field_bytes(2, sd) if sd else b"")
    hom = field_varint(1, src) + field_varint(2, dst) + field_varint(3, context_conn_id) + field_bytes(4, any_bytes)
    return message_pack(dst, MSG_CHANNEL_DATA_HANDOVER, hom)


def frame(packet_bytes: bytes, compression: int = 0) -> bytes:
    n = len(packet_bytes)
    assert n <= MAX_PACKET_SIZE
    return bytes([67, 72, (n >> 8) & 0xFF, n & 0xFF, compression]) + packet_bytes  # connection.go:683-687


def read_size(tag: bytes) -> int:
    """readSize (connection.go:445-453), the receiver's rule for the tag flush() writes: 0 unless 'C','H'."""
    if tag[0] != 67 or tag[1] != 72:
        return 0
    return tag[3] | (tag[2] << 8)


def flush_stream(packs: Iterable[bytes]) -> Tuple[bytes, List[int]]:
    """The byte stream successive flush() calls write for one connection's queue of MessagePacks,
    and the number of packs per packet.  Packs of size >= 65530 are dropped by Send (:72-77)."""
    out = bytearray()
    counts: List[int] = []
    cur = bytearray()
    n_in = 0
    for mp in packs:
        if len(mp) >= MAX_PACKET_SIZE - PACKET_HEADER_SIZE:
            continue
        entry = field_bytes(1, mp)  # Packet.messages = 1
        if n_in and len(cur) + len(entry) > MAX_PACKET_SIZE:
            out += frame(bytes(cur))
            counts.append(n_in)
            cur = bytearray()
            n_in = 0
        cur += entry
        n_in += 1
    if n_in:
        out += frame(bytes(cur))
        counts.append(n_in)
    return bytes(out), counts
