#!/usr/bin/env python
"""Golden wire bytes for the fan-out path, produced with the REFERENCE'S OWN protobuf schema.

channeld's generated Go code embeds the serialized FileDescriptorProto of channeld.proto
(/root/reference/pkg/channeldpb/channeld.pb.go: file_channeld_proto_rawDesc).  This script parses
that descriptor with python-protobuf, builds Packet / MessagePack / ChannelDataUpdateMessage
classes from it, serializes seeded fan-out streams exactly as the reference assembles them
(data.go:293-318, connection.go:57-83,626-714) and stores inputs + expected bytes in
tests/golden/wire_packets.npz.  It needs /root/reference and therefore runs only in the build
container; the tests read the .npz.

    python tests/golden/make_wire_golden.py
"""
import os
import re
import sys

import numpy as np
from google.protobuf import any_pb2, descriptor_pb2, descriptor_pool, message_factory

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/pkg/channeldpb/channeld.pb.go"


def reference_classes():
    src = open(REF).read()
    m = re.search(r"var file_channeld_proto_rawDesc = \[\]byte\{(.*?)\n\}", src, re.S)
    raw = bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", m.group(1)))
    fd = descriptor_pb2.FileDescriptorProto()
    fd.ParseFromString(raw)
    pool = descriptor_pool.DescriptorPool()
    anyfd = descriptor_pb2.FileDescriptorProto()
    any_pb2.DESCRIPTOR.CopyToProto(anyfd)
    pool.Add(anyfd)
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("channeldpb." + n))
    return get("Packet"), get("MessagePack"), get("ChannelDataUpdateMessage"), pool


def main():
    Packet, MessagePack, CDU, pool = reference_classes()
    AnyCls = message_factory.GetMessageClass(pool.FindMessageTypeByName("google.protobuf.Any"))
    rng = np.random.default_rng(20260923)
    type_urls = [b"type.googleapis.com/tpspb.EntityChannelData", b"type.googleapis.com/unrealpb.SpatialChannelData", b""]
    out = {}
    # streams: (number of messages, payload length chooser)
    cases = [
        ("small_mixed", 40, lambda: int(rng.integers(0, 60))),
        ("position_updates", 900, lambda: 21),            # 21-byte value: the minimal position update of SURVEY a14
        ("split_many_packets", 2600, lambda: int(rng.choice([21, 21, 21, 64, 300]))),
        ("big_payloads", 9, lambda: int(rng.integers(20000, 30000))),
        ("oversized_dropped", 5, lambda: int(rng.choice([100, 65600]))),
        ("empty_any", 5, lambda: -1),
    ]
    for name, n, plen in cases:
        chans, anys, lens = [], [], []
        packs = []
        for i in range(n):
            ch = int(rng.choice([0x10000 + rng.integers(0, 225), 0x80000 + rng.integers(0, 1 << 20), 1, 127, 128, 0x7FFFFFFF,
                                 0xFFFFFFFF]))
            L = plen()
            a = AnyCls()
            if L >= 0:
                a.type_url = type_urls[int(rng.integers(0, 3))].decode()
                a.value = rng.integers(0, 256, L, dtype=np.uint8).tobytes()
            any_bytes = a.SerializeToString()
            body = CDU(data=a)                                  # data.go:302
            mp = MessagePack(channelId=ch, broadcast=0, stubId=0, msgType=8, msgBody=body.SerializeToString())
            chans.append(ch)
            anys.append(any_bytes)
            lens.append(len(any_bytes))
            packs.append(mp)
        # connection.go:626-714 with the reference's own size function (ByteSize of the growing Packet)
        stream = bytearray()
        counts = []
        p = Packet()
        for mp in packs:
            if mp.ByteSize() >= 0xFFFF - 5:                     # connection.go:72-77
                continue
            p.messages.append(mp)
            if p.ByteSize() > 0xFFFF:
                del p.messages[-1]
                b = p.SerializeToString()
                stream += bytes([67, 72, (len(b) >> 8) & 0xFF, len(b) & 0xFF, 0]) + b
                counts.append(len(p.messages))
                p = Packet()
                p.messages.append(mp)
        if len(p.messages):
            b = p.SerializeToString()
            stream += bytes([67, 72, (len(b) >> 8) & 0xFF, len(b) & 0xFF, 0]) + b
            counts.append(len(p.messages))
        out[f"{name}_chan"] = np.array(chans, dtype=np.uint32)
        out[f"{name}_anylen"] = np.array(lens, dtype=np.uint32)
        out[f"{name}_any"] = np.frombuffer(b"".join(anys), dtype=np.uint8)
        out[f"{name}_stream"] = np.frombuffer(bytes(stream), dtype=np.uint8)
        out[f"{name}_counts"] = np.array(counts, dtype=np.uint32)
    np.savez_compressed(os.path.join(HERE, "wire_packets.npz"), **out)
    print("wire_packets.npz", os.path.getsize(os.path.join(HERE, "wire_packets.npz")),
          {k: len(v) for k, v in out.items() if k.endswith("_counts")})


if __name__ == "__main__":
    main()
