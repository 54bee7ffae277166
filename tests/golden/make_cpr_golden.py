#!/usr/bin/env python
"""Golden wire bytes RECORDED FROM THE REFERENCE: the client-packet recordings shipped with channeld
(examples/replay/*/*.cpr = serialized replaypb.ReplaySession{packets: [{offsetTime, channeldpb.Packet}]},
written by the Go server with google.golang.org/protobuf, connection.go:768-821).

Every MessagePack inside them is a byte string the reference's marshaller produced.  This script cuts the
sessions into their Packets and MessagePacks (plain wire-format walk, no schema needed for that), decodes the
fields of every pack with python-protobuf driven by the reference's embedded descriptor, and stores
(fields, recorded bytes) in tests/golden/cpr_packs.npz; tests/test_wire_oracle.py re-encodes the fields with
oracle/wire.py and must reproduce the recorded bytes.  Needs /root/reference: runs only in the build container.

    python tests/golden/make_cpr_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_wire_golden import reference_classes  # noqa: E402

SESSIONS = ["/root/reference/examples/replay/webchat/session_1_22-09-07_14-41-02.cpr",
            "/root/reference/examples/replay/tps/session_2_22-09-16_16-44-04.cpr"]


def fields(buf):
    """(field number, wire type, value) of one serialized message; value = int or bytes"""
    i, out = 0, []
    while i < len(buf):
        key = 0
        sh = 0
        while True:
            b = buf[i]
            i += 1
            key |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                break
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, sh = 0, 0
            while True:
                b = buf[i]
                i += 1
                v |= (b & 0x7F) << sh
                sh += 7
                if not b & 0x80:
                    break
            out.append((num, wt, v))
        elif wt == 2:
            n, sh = 0, 0
            while True:
                b = buf[i]
                i += 1
                n |= (b & 0x7F) << sh
                sh += 7
                if not b & 0x80:
                    break
            out.append((num, wt, bytes(buf[i:i + n])))
            i += n
        elif wt == 1:
            out.append((num, wt, bytes(buf[i:i + 8])))
            i += 8
        elif wt == 5:
            out.append((num, wt, bytes(buf[i:i + 4])))
            i += 4
        else:
            raise ValueError(f"wire type {wt}")
    return out


def main():
    Packet, MessagePack, CDU, pool = reference_classes()
    packs, packets = [], []
    for path in SESSIONS:
        raw = open(path, "rb").read()
        for num, wt, rp in fields(raw):            # ReplaySession.packets = 1
            assert (num, wt) == (1, 2)
            pk = [v for n, w, v in fields(rp) if (n, w) == (2, 2)]   # ReplayPacket.packet = 2
            if not pk:
                continue
            packets.append(pk[0])
            for n, w, mp in fields(pk[0]):          # Packet.messages = 1
                assert (n, w) == (1, 2)
                packs.append(mp)
    rows = []
    blob = bytearray()
    any_blob = bytearray()
    for mp in packs:
        m = MessagePack()
        m.ParseFromString(mp)
        any_off, any_len, ctx, has_any = 0, 0, 0, 0
        if m.msgType == 8:                          # CHANNEL_DATA_UPDATE: msgBody = ChannelDataUpdateMessage
            body = CDU()
            body.ParseFromString(m.msgBody)
            ab = [v for n, w, v in fields(m.msgBody) if (n, w) == (1, 2)]
            has_any = 1 if ab else 0
            if ab:
                any_off, any_len = len(any_blob), len(ab[0])
                any_blob += ab[0]
            ctx = body.contextConnId
        rows.append((m.channelId, m.broadcast, m.stubId, m.msgType, len(blob), len(mp), len(m.msgBody), any_off, any_len, ctx, has_any))
        blob += mp
    # body bytes are recoverable from the pack bytes (field 5): the test cuts them with the same walk
    pk_off = np.cumsum([0] + [len(p) for p in packets])
    out = os.path.join(HERE, "cpr_packs.npz")
    np.savez_compressed(out, rows=np.array(rows, dtype=np.int64), pack_bytes=np.frombuffer(bytes(blob), dtype=np.uint8),
                        any_bytes=np.frombuffer(bytes(any_blob), dtype=np.uint8),
                        packet_bytes=np.frombuffer(b"".join(packets), dtype=np.uint8), packet_off=pk_off.astype(np.int64))
    types = np.bincount(np.array(rows)[:, 3])
    print(f"{len(packets)} packets, {len(packs)} message packs, by msgType: {dict((i, int(c)) for i, c in enumerate(types) if c)}; {out} "
          f"({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main()
