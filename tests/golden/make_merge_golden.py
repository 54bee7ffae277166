#!/usr/bin/env python
"""Golden vectors for the TYPED merge of buffered channel-data updates (data.go:225-269): what tickData hands
to fanOutDataUpdate when a subscriber's window holds several updates — proto.Merge of the first into an empty
accumulatedUpdateMsg, EntityChannelData.Merge (tpspb/data.go:227-252: srcData.ObjRef = nil; proto.Merge) for the rest —
then fmutils.Filter with the subscription's DataFieldMasks (data.go:294) and anypb.New / proto.Marshal.

Produced with the REFERENCE'S OWN schemas: the FileDescriptorProtos embedded in pkg/channeldpb/channeld.pb.go,
pkg/unrealpb/unreal_common.pb.go and examples/channeld-ue-tps/tpspb/tps.pb.go, driven by python-protobuf (MergeFrom has
proto.Merge's semantics; SerializeToString(deterministic) emits known fields in field-number order, as Go's Marshal does).
The updates are the MOVEMENT subset of tpspb.EntityChannelData: actorState.replicatedMovement{linearVelocity,
angularVelocity, location, rotation: FVector{x, y, z}; bSimulatedPhysicSleep, bRepPhysics} with any subset of the leaves
present, present-but-empty sub-messages included.
fmutils.Filter (github.com/indiest/fmutils v0.1.2, go.mod; not vendored) is restated here from the published algorithm of
its upstream github.com/mennanov/fmutils (NestedMask.Filter): keep the fields a path names, recurse into named
sub-messages, clear everything else; an empty mask keeps the message as it is.  PARITY UNPINNED for the filter (no Go
toolchain, no vendored source): said so in DESIGN.md.
Needs /root/reference, so it runs only in the build container; tests read tests/golden/merge_vectors.npz.

    python tests/golden/make_merge_golden.py
"""
import os
import re

import numpy as np
from google.protobuf import any_pb2, descriptor_pb2, descriptor_pool, message_factory

HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS = ["linearVelocity", "angularVelocity", "location", "rotation", "bSimulatedPhysicSleep", "bRepPhysics"]  # FRepMovement 1..6


def raw_desc(path, var):
    src = open(path).read()
    m = re.search(r"var %s = \[\]byte\{(.*?)\n\}" % var, src, re.S)
    return bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", m.group(1)))


def pool():
    p = descriptor_pool.DescriptorPool()
    for mod in (any_pb2, descriptor_pb2):
        fd = descriptor_pb2.FileDescriptorProto()
        mod.DESCRIPTOR.CopyToProto(fd)
        p.Add(fd)
    for path, var in (("/root/reference/pkg/channeldpb/channeld.pb.go", "file_channeld_proto_rawDesc"),
                      ("/root/reference/pkg/unrealpb/unreal_common.pb.go", "file_unreal_common_proto_rawDesc"),
                      ("/root/reference/examples/channeld-ue-tps/tpspb/tps.pb.go", "file_tps_proto_rawDesc")):
        fd = descriptor_pb2.FileDescriptorProto()
        fd.ParseFromString(raw_desc(path, var))
        p.Add(fd)
    return p


def nested_mask(paths):
    root = {}
    for path in paths:
        m = root
        for part in path.split("."):
            m = m.setdefault(part, {})
    return root


def fm_filter(msg, mask):
    """mennanov/fmutils NestedMask.Filter"""
    if not mask:
        return
    for fd, value in list(msg.ListFields()):
        m = mask.get(fd.name)
        if m is None:
            msg.ClearField(fd.name)
        elif m and fd.message_type is not None and not fd.is_repeated:
            fm_filter(value, m)


def main():
    p = pool()
    cls = lambda n: message_factory.GetMessageClass(p.FindMessageTypeByName(n))
    Entity = cls("tpspb.EntityChannelData")
    rng = np.random.default_rng(20260925)

    def random_update():
        u = Entity()
        r = rng.random()
        if r < 0.04:
            return u                       # an empty update
        a = u.actorState
        a.SetInParent()
        if r < 0.08:
            return u                       # actorState present, empty
        mv = a.replicatedMovement
        mv.SetInParent()
        for name in FIELDS[:4]:
            q = rng.random()
            if q < 0.45:
                continue
            v = getattr(mv, name)
            v.SetInParent()
            for ax in "xyz":
                if rng.random() < 0.8:
                    setattr(v, ax, float(np.float32(rng.normal() * 1000)))
        for name in FIELDS[4:]:
            if rng.random() < 0.3:
                setattr(mv, name, bool(rng.random() < 0.5))
        return u

    n = 400
    ins, in_off, in_cnt, masks, want, want_off = [], [0], [], [], [], [0]
    for k in range(n):
        cnt = int(rng.choice([1, 1, 2, 2, 3, 4]))
        ups = [random_update() for _ in range(cnt)]
        # the bit form of the DataFieldMasks the engine takes: bit f = "actorState.replicatedMovement.<FIELDS[f]>" is listed,
        # bit 6 = some other top-level field is listed and actorState is not (everything of the subset is cleared); 0 = no masks
        mk = 0
        q = rng.random()
        if q < 0.35:
            mk = int(rng.integers(1, 64))
        elif q < 0.40:
            mk = 64
        paths = ["actorState.replicatedMovement." + FIELDS[f] for f in range(6) if (mk >> f) & 1] + (["characterState"] if mk & 64 else [])
        acc = Entity()
        for j, u in enumerate(ups):
            b = u.SerializeToString(deterministic=True)
            ins.append(np.frombuffer(b, dtype=np.uint8))
            in_off.append(in_off[-1] + len(b))
            if j == 0:
                acc.MergeFrom(u)               # data.go:250: proto.Merge(accumulatedUpdateMsg, be.updateMsg)
            else:
                u2 = Entity()
                u2.CopyFrom(u)
                u2.ClearField("objRef")        # tpspb/data.go:251
                acc.MergeFrom(u2)              # tpspb/data.go:252
        fm_filter(acc, nested_mask(paths))     # data.go:294
        w = acc.SerializeToString(deterministic=True)
        in_cnt.append(cnt)
        masks.append(mk)
        want.append(np.frombuffer(w, dtype=np.uint8))
        want_off.append(want_off[-1] + len(w))
    np.savez_compressed(os.path.join(HERE, "merge_vectors.npz"),
                        inputs=np.concatenate(ins) if ins else np.zeros(0, np.uint8), in_off=np.array(in_off, dtype=np.int64),
                        in_cnt=np.array(in_cnt, dtype=np.int32), masks=np.array(masks, dtype=np.uint32),
                        want=np.concatenate(want), want_off=np.array(want_off, dtype=np.int64))
    print("cases", n, "updates", len(ins), "bytes", in_off[-1], "max update", max(len(a) for a in ins), "max merged", max(len(a) for a in want))


if __name__ == "__main__":
    main()
