#!/usr/bin/env python
"""Golden bytes for the handover messages Notify assembles (spatial.go:738-857), produced with the REFERENCE'S OWN
protobuf schemas: the FileDescriptorProtos embedded in pkg/channeldpb/channeld.pb.go and pkg/unrealpb/unreal_common.pb.go
are parsed with python-protobuf, MessagePack{ChannelDataHandoverMessage{Any{unrealpb.SpatialChannelData}}} is built the
way the reference builds it (HandoverDataMerger.MergeTo, examples/channeld-ue-tps/tpspb/data.go:323-347: SpatialEntityState
{objRef[, entityData]} keyed by NetGUID) and serialized; inputs + expected bytes go to tests/golden/handover_msgs.npz.
Needs /root/reference, so it runs only in the build container; tests read the .npz.

    python tests/golden/make_handover_golden.py
"""
import os
import re
import sys

import numpy as np
from google.protobuf import any_pb2, descriptor_pb2, descriptor_pool, message_factory

HERE = os.path.dirname(os.path.abspath(__file__))


def raw_desc(path, var):
    src = open(path).read()
    m = re.search(r"var %s = \[\]byte\{(.*?)\n\}" % var, src, re.S)
    return bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", m.group(1)))


def main():
    pool = descriptor_pool.DescriptorPool()
    anyfd = descriptor_pb2.FileDescriptorProto()
    any_pb2.DESCRIPTOR.CopyToProto(anyfd)
    pool.Add(anyfd)
    from google.protobuf import descriptor_pb2 as dpb2  # descriptor.proto is imported by unreal_common.proto (MessageOptions extension)
    dfd = descriptor_pb2.FileDescriptorProto()
    dpb2.DESCRIPTOR.CopyToProto(dfd)
    pool.Add(dfd)
    for path, var in (("/root/reference/pkg/channeldpb/channeld.pb.go", "file_channeld_proto_rawDesc"),
                      ("/root/reference/pkg/unrealpb/unreal_common.pb.go", "file_unreal_common_proto_rawDesc")):
        fd = descriptor_pb2.FileDescriptorProto()
        fd.ParseFromString(raw_desc(path, var))
        pool.Add(fd)
    cls = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName(n))
    MessagePack, Handover = cls("channeldpb.MessagePack"), cls("channeldpb.ChannelDataHandoverMessage")
    SpatialChannelData, ObjRef, AnyCls = cls("unrealpb.SpatialChannelData"), cls("unrealpb.UnrealObjectRef"), cls("google.protobuf.Any")
    rng = np.random.default_rng(20260924)
    url = "type.googleapis.com/unrealpb.SpatialChannelData"
    out = {"type_url": np.frombuffer(url.encode(), dtype=np.uint8)}
    n = 24
    src, dst, ctx, net, full = [], [], [], [], []
    objrefs, anys, packs = [], [], []
    for i in range(n):
        s, d = 0x10000 + int(rng.integers(0, 225)), 0x10000 + int(rng.integers(0, 225))
        c = int(rng.choice([0, 3, 300, 70000]))
        nid = int(rng.choice([0x80000 + int(rng.integers(0, 100000)), 5, 200, 20000]))
        ref = ObjRef(netGUID=nid)
        if rng.random() < 0.5:
            ref.classPath = "/Game/BP_%d" % rng.integers(0, 99)
        if rng.random() < 0.3:
            ref.owningConnId = int(rng.integers(1, 500))
        if i == 0:
            ref = ObjRef()  # an empty (but set) objRef: still emitted
        ent_any = AnyCls(type_url="type.googleapis.com/tpspb.EntityChannelData", value=bytes(rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8)))
        with_full = bool(i % 2)
        sd = SpatialChannelData()
        st = sd.entities[nid]
        st.objRef.CopyFrom(ref)
        if with_full:
            st.entityData.CopyFrom(ent_any)
        data = AnyCls()
        data.Pack(sd, type_url_prefix="type.googleapis.com/")
        assert data.type_url == url
        hom = Handover(srcChannelId=s, dstChannelId=d, contextConnId=c, data=data)
        mp = MessagePack(channelId=d, msgType=12, msgBody=hom.SerializeToString())
        src.append(s); dst.append(d); ctx.append(c); net.append(nid); full.append(with_full)
        objrefs.append(ref.SerializeToString()); anys.append(ent_any.SerializeToString()); packs.append(mp.SerializeToString())
    # one two-entity handover (a handover group): entries in ascending key order, as python serializes maps deterministically
    sd = SpatialChannelData()
    for nid in (0x80010, 0x80020):
        sd.entities[nid].objRef.netGUID = nid
    data = AnyCls()
    data.Pack(sd, type_url_prefix="type.googleapis.com/", deterministic=True)  # (map entries in key order; Go's order is random)
    hom = Handover(srcChannelId=0x10001, dstChannelId=0x10002, data=data)
    out["group_pack"] = np.frombuffer(MessagePack(channelId=0x10002, msgType=12, msgBody=hom.SerializeToString(deterministic=True)).SerializeToString(), dtype=np.uint8)
    out["group_objrefs"] = np.array([ObjRef(netGUID=nid).SerializeToString() for nid in (0x80010, 0x80020)], dtype=object)
    # ... and one handover list of five of whom the destination connection already knows two (members in a cell it is subscribed
    # to): entityData for the other three only — what the per-(connection, entity) `shouldSend` of spatial.go:797-857 produces
    sd = SpatialChannelData()
    mixed_ids = [0x80100, 0x80101, 0x80102, 0x80103, 0x80104]
    mixed_mask = 0b10110
    mixed_any = []
    for j, nid in enumerate(mixed_ids):
        st = sd.entities[nid]
        st.objRef.netGUID = nid
        ea = AnyCls(type_url="type.googleapis.com/tpspb.EntityChannelData", value=bytes(rng.integers(0, 256, 20 + 7 * j, dtype=np.uint8)))
        mixed_any.append(ea.SerializeToString())
        if (mixed_mask >> j) & 1:
            st.entityData.CopyFrom(ea)
    data = AnyCls()
    data.Pack(sd, type_url_prefix="type.googleapis.com/", deterministic=True)
    hom = Handover(srcChannelId=0x10005, dstChannelId=0x10006, contextConnId=9, data=data)
    out["mixed_pack"] = np.frombuffer(MessagePack(channelId=0x10006, msgType=12, msgBody=hom.SerializeToString(deterministic=True)).SerializeToString(), dtype=np.uint8)
    out["mixed_mask"] = np.array([mixed_mask], dtype=np.uint32)
    out["mixed_any_len"] = np.array([len(b) for b in mixed_any], dtype=np.uint32)
    out["mixed_any_bytes"] = np.frombuffer(b"".join(mixed_any), dtype=np.uint8)
    out["mixed_objref_len"] = np.array([len(ObjRef(netGUID=nid).SerializeToString()) for nid in mixed_ids], dtype=np.uint32)
    out["mixed_objref_bytes"] = np.frombuffer(b"".join(ObjRef(netGUID=nid).SerializeToString() for nid in mixed_ids), dtype=np.uint8)
    for k, v in (("src", src), ("dst", dst), ("ctx", ctx), ("net", net), ("full", full)):
        out[k] = np.array(v, dtype=np.uint32)
    for k, v in (("objref", objrefs), ("any", anys), ("pack", packs)):
        out[k + "_len"] = np.array([len(b) for b in v], dtype=np.uint32)
        out[k + "_bytes"] = np.frombuffer(b"".join(v), dtype=np.uint8)
    out["group_objref_len"] = np.array([len(b) for b in out["group_objrefs"]], dtype=np.uint32)
    out["group_objref_bytes"] = np.frombuffer(b"".join(out.pop("group_objrefs")), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "handover_msgs.npz"), **out)
    print("wrote handover_msgs.npz:", n, "single-entity handovers + one group of two")


if __name__ == "__main__":
    main()
