"""The typed-merge oracle (oracle/merge.py: the movement subset of tpspb.EntityChannelData merged as tickData merges buffered
updates, filtered by DataFieldMasks, marshalled) against the golden vectors python-protobuf produced from the reference's
own embedded descriptors (tests/golden/make_merge_golden.py)."""
import os

import numpy as np

from oracle import merge

HERE = os.path.dirname(os.path.abspath(__file__))


def test_merge_oracle_reproduces_protobuf_on_the_reference_descriptors():
    g = np.load(os.path.join(HERE, "golden", "merge_vectors.npz"))
    inputs, in_off, want, want_off = g["inputs"].tobytes(), g["in_off"], g["want"].tobytes(), g["want_off"]
    u = 0
    multi = masked = 0
    for k, (cnt, mask) in enumerate(zip(g["in_cnt"], g["masks"])):
        ups = [inputs[in_off[u + j]:in_off[u + j + 1]] for j in range(cnt)]
        u += cnt
        got = merge.merged_update(ups, int(mask))
        assert got == want[want_off[k]:want_off[k + 1]], f"case {k}: {cnt} updates, mask {mask:#x}"
        multi += cnt > 1
        masked += mask != 0
    assert multi > 100 and masked > 100


def test_generated_updates_are_canonical_and_a_foreign_field_is_refused():
    rng = np.random.default_rng(3)
    for _ in range(200):
        b = merge.make_update(rng)
        assert merge.merged_update([b]) == b  # parse + serialize is the identity on canonical encodings
    import pytest

    with pytest.raises(merge.NotInSubset):
        merge.merged_update([bytes([0x0A, 0x02, 0x08, 0x05])])  # objRef { netGUID: 5 }


def test_field_mask_paths_outside_the_bit_form_are_refused():
    """fmutils.Filter keeps a sub-message that a path walks INTO present but emptied, and a specific path under a
    listed message narrows it — neither is one of the engine's bit forms, so the path mapper must refuse them (the host then keeps
    that subscription's merge to itself) instead of returning a mask with other semantics."""
    import pytest

    from channeld_amd.engine import movement_field_mask

    assert movement_field_mask([]) == 0
    assert movement_field_mask(["actorState.replicatedMovement.location"]) == 1 << 2
    assert movement_field_mask(["actorState"]) == 63 and movement_field_mask(["actorState.replicatedMovement"]) == 63
    assert movement_field_mask(["characterState"]) == 64
    for bad in (["actorState.owner"], ["actorState.replicatedMovement.nosuchfield"], ["actorState.replicatedMovement.location.x"],
                ["actorState", "actorState.replicatedMovement.location"], ["actorState.replicatedMovement", "actorState.replicatedMovement.rotation"]):
        with pytest.raises(ValueError):
            movement_field_mask(bad)
