"""TEST INFRASTRUCTURE (oracle): the typed merge of buffered channel-data updates for the MOVEMENT subset of
tpspb.EntityChannelData, restated without protobuf so that it can run on the GPU box.

Reference: tickData merges the updates a subscriber's window selected (data.go:225-269): proto.Merge of the first into an
empty accumulatedUpdateMsg (:250), EntityChannelData.Merge for the rest (examples/channeld-ue-tps/tpspb/data.go:227-252:
objRef dropped, then proto.Merge); fanOutDataUpdate filters the result with the subscription's DataFieldMasks
(fmutils.Filter, data.go:294) and marshals it (anypb.New, :295).  proto.Merge on this schema: sub-messages merge
recursively, a scalar with explicit presence (`optional`) is overwritten when the source has it.  Go's Marshal and
python-protobuf's deterministic SerializeToString both emit known fields in field-number order.

Subset (unreal_common.proto:161-184, tps.proto:22-35), all lengths below 128 (one-byte varints):
    EntityChannelData { 2: ActorState { 11: FRepMovement { 1 linearVelocity, 2 angularVelocity, 3 location, 4 rotation:
    FVector { 1 x, 2 y, 3 z: float }; 5 bSimulatedPhysicSleep, 6 bRepPhysics: bool } } }
State of a message: presence of actorState / replicatedMovement / each vector / each leaf, and the leaf values.
Pinned by tests/golden/merge_vectors.npz (python-protobuf on the reference's embedded descriptors,
tests/golden/make_merge_golden.py) in tests/test_merge_oracle.py."""
import struct


class NotInSubset(Exception):
    """the message carries a field outside the movement subset (the engine then falls back to concatenation)"""


def _varint(b, i):
    v = s = 0
    while True:
        if i >= len(b):
            raise NotInSubset("truncated varint")
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        s += 7
        if not c & 0x80:
            return v, i
        if s > 63:
            raise NotInSubset("varint too long")


def _fields(b):
    i = 0
    while i < len(b):
        key, i = _varint(b, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 5:
            if i + 4 > len(b):
                raise NotInSubset("truncated fixed32")
            v, i = b[i:i + 4], i + 4
        elif wt == 2:
            n, i = _varint(b, i)
            if i + n > len(b):
                raise NotInSubset("truncated bytes")
            v, i = b[i:i + n], i + n
        else:
            raise NotInSubset(f"wire type {wt}")
        yield fn, wt, v


def new_state():
    return {"actor": False, "mv": False, "vec": [False] * 4, "leaf": [[None] * 3 for _ in range(4)], "bool": [None, None]}


def merge_into(st, update: bytes):
    """proto.Merge(st, parse(update)) for the subset; raises NotInSubset otherwise (objRef, field 1, is outside it)"""
    for fn, wt, v in _fields(update):
        if fn != 2 or wt != 2:
            raise NotInSubset(f"EntityChannelData field {fn}")
        st["actor"] = True
        for fn2, wt2, v2 in _fields(v):
            if fn2 != 11 or wt2 != 2:
                raise NotInSubset(f"ActorState field {fn2}")
            st["mv"] = True
            for fn3, wt3, v3 in _fields(v2):
                if 1 <= fn3 <= 4 and wt3 == 2:
                    st["vec"][fn3 - 1] = True
                    for fn4, wt4, v4 in _fields(v3):
                        if not (1 <= fn4 <= 3 and wt4 == 5):
                            raise NotInSubset(f"FVector field {fn4}")
                        st["leaf"][fn3 - 1][fn4 - 1] = bytes(v4)
                elif fn3 in (5, 6) and wt3 == 0:
                    st["bool"][fn3 - 5] = 1 if v3 else 0
                else:
                    raise NotInSubset(f"FRepMovement field {fn3}")


def apply_field_mask(st, mask: int):
    """fmutils.Filter with DataFieldMasks in the engine's bit form: bit f (0..5) = "actorState.replicatedMovement.<field f+1>"
    is listed; bit 6 = only fields outside actorState are listed (the whole subset is cleared); 0 = no masks."""
    if not mask:
        return
    if mask & 64 and not mask & 63:
        st.update(new_state())
        return
    for f in range(4):
        if not (mask >> f) & 1:
            st["vec"][f] = False
            st["leaf"][f] = [None] * 3
    for f in (4, 5):
        if not (mask >> f) & 1:
            st["bool"][f - 4] = None


def serialize(st) -> bytes:
    if not st["actor"]:
        return b""
    mv = b""
    if st["mv"]:
        for f in range(4):
            if st["vec"][f]:
                vec = b"".join(bytes([0x0D + 8 * a]) + st["leaf"][f][a] for a in range(3) if st["leaf"][f][a] is not None)
                mv += bytes([0x0A + 8 * f, len(vec)]) + vec
        for k in range(2):
            if st["bool"][k] is not None:
                mv += bytes([0x28 + 8 * k, st["bool"][k]])
    actor = (bytes([0x5A, len(mv)]) + mv) if st["mv"] else b""
    return bytes([0x12, len(actor)]) + actor


def merged_update(updates, field_mask: int = 0) -> bytes:
    """bytes of the accumulated update message for `updates` (oldest first) as the reference marshals it"""
    st = new_state()
    for u in updates:
        merge_into(st, u)
    apply_field_mask(st, field_mask)
    return serialize(st)


def make_update(rng, p_vec=0.55, p_leaf=0.8, p_bool=0.3) -> bytes:
    """a random movement update in canonical encoding (numpy Generator)"""
    st = new_state()
    st["actor"] = st["mv"] = True
    for f in range(4):
        if rng.random() < p_vec:
            st["vec"][f] = True
            for a in range(3):
                if rng.random() < p_leaf:
                    st["leaf"][f][a] = struct.pack("<f", float(rng.normal() * 1000))
    for k in range(2):
        if rng.random() < p_bool:
            st["bool"][k] = int(rng.random() < 0.5)
    return serialize(st)
