"""Host mirror of channeld's entity group controllers for the engine.

The reference keeps, per ENTITY channel, a `FlatEntityGroupController` (pkg/channeld/entity.go:58-224) with a
pointer to a handover group and a pointer to a lock group; group instances are shared between the channels a
cascade has reached (entity.go:80-102).  `Notify` asks the notifying entity's controller for
`GetHandoverEntities()` (entity.go:197-224; spatial.go:675-679): the members of its handover group, or nothing
when one of them is in ITS lock group.  These per-channel views are not equivalence classes — a locked entity
does not take over the group it is added to, an entity removed from a group keeps an EMPTY group and cannot hand
over until it is added again (entity_test.go:82-88) — so the engine is not given group ids but the evaluated
lists: `EntityGroupTable.engine_lists()` -> `SpatialWorld.set_handover_lists(...)`
(C-ABI `chd_world_set_handover_lists`).

In a Go gateway this table IS the reference's controllers: the cgo shim walks them after every
AddEntityGroupMessage / RemoveEntityGroupMessage (message handlers entity.go:226-290) and uploads the lists.
Same method names and argument meaning as the Go interface `EntityGroupController` (entity.go:49-56).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

# channeldpb.EntityGroupType (channeld.proto)
EntityGroupType_HANDOVER = 0
EntityGroupType_LOCK = 1

NO_LIST = 0xFFFFFFFF


class EntityGroupTable:
    """All entity channels' group controllers of one gateway.

    Group instances live in an arena (`_sets[k]` = set of entity ids) so that "two channels share one group
    instance" is "they hold the same index"; a controller is the pair (handover index, lock index), -1 = nil.
    """

    def __init__(self):
        self._sets: List[set] = []
        self._handover: Dict[int, int] = {}  # entity id -> arena index (-1 = nil pointer)
        self._lock: Dict[int, int] = {}
        self._slot: Dict[int, int] = {}      # entity id -> engine entity slot

    # -- channel lifecycle -------------------------------------------------------------------------------------------
    def CreateChannel(self, entity_id: int, slot: int):
        """An ENTITY channel was created (its controller is Initialize()d, entity.go:66-68); `slot` = the engine's
        entity slot given to chd_world_spawn for it."""
        self._handover[entity_id] = -1
        self._lock[entity_id] = -1
        self._slot[entity_id] = int(slot)

    def RemoveChannel(self, entity_id: int):
        """Uninitialize (entity.go:70-78): the entity leaves its current groups, which other channels may share."""
        if entity_id not in self._slot:
            return
        for t in (EntityGroupType_HANDOVER, EntityGroupType_LOCK):
            if self._pointer(t)[entity_id] >= 0:
                self.RemoveFromGroup(entity_id, t, [entity_id])
        for d in (self._handover, self._lock, self._slot):
            d.pop(entity_id, None)

    # -- the EntityGroupController interface, per channel ------------------------------------------------------------
    def AddToGroup(self, entity_id: int, t: int, entities: Iterable[int]) -> Optional[str]:
        """entity.go:104-158, on the controller of channel `entity_id`."""
        ptr = self._pointer(t)
        if ptr[entity_id] < 0:
            ptr[entity_id] = self._new_set()
        for e in entities:
            k = ptr[entity_id]
            self._sets[k].add(e)
            if e in self._slot:  # GetChannel(e) != nil: the member's own controller joins the shared instance
                self._cascade(e, t, k)
        return None

    def RemoveFromGroup(self, entity_id: int, t: int, entities: Iterable[int]) -> Optional[str]:
        """entity.go:160-195; returns the reference's error text when the group pointer is nil."""
        ptr = self._pointer(t)
        if ptr[entity_id] < 0:
            return f"{'handover' if t == EntityGroupType_HANDOVER else 'lock'} group is nil, entityId: {entity_id}"
        for e in entities:
            # (the controller's pointer is read anew for every entity, as the Go loop does: once the channel removes ITSELF
            # its pointer is a fresh empty group and the rest of the list no longer touches the shared instance)
            self._sets[ptr[entity_id]].discard(e)
            if e in self._slot:
                ptr[e] = self._new_set()  # the removed entity's channel starts over with an EMPTY group
        return None

    def GetHandoverEntities(self, entity_id: int) -> List[int]:
        """entity.go:197-224 (entity ids, ascending)."""
        h = self._handover[entity_id]
        if h < 0:
            return [entity_id]
        members = self._sets[h]
        lk = self._lock[entity_id]
        if lk >= 0 and not members.isdisjoint(self._sets[lk]):
            return []
        return sorted(members)

    # -- what the engine takes ---------------------------------------------------------------------------------------
    def engine_lists(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """(list_off, list_members, idx, list_of) for SpatialWorld.set_handover_lists: every channel's evaluated
        GetHandoverEntities, members translated to entity slots (ids without an entity channel — a character's
        PlayerController in the reference test — have no engine state and are left out), identical lists shared."""
        lists: Dict[Tuple[int, ...], int] = {}
        idx, list_of = [], []
        for e in sorted(self._slot):
            idx.append(self._slot[e])
            if self._handover[e] < 0:
                list_of.append(NO_LIST)
                continue
            got = self.GetHandoverEntities(e)
            # an entity whose result is non-empty but holds no live entity cannot move anything either: keep it distinct
            # from "locked" only through the list contents (both are empty lists for the engine)
            key = tuple(sorted(self._slot[m] for m in got if m in self._slot))
            list_of.append(lists.setdefault(key, len(lists)))
        order = sorted(lists, key=lists.get)
        off = np.zeros(len(order) + 1, dtype=np.uint32)
        for k, key in enumerate(order):
            off[k + 1] = off[k] + len(key)
        mem = np.array([m for key in order for m in key], dtype=np.uint32)
        return off, mem, np.array(idx, dtype=np.uint32), np.array(list_of, dtype=np.uint32)

    def shard_lists(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """(list_off, list_member_chan, chan_id, list_of) for HipShardEngine.set_handover_lists / chd_shard_set_handover_lists: the
        same evaluated lists keyed by ENTITY CHANNEL ID (an entity's id is its channel id, entity.go:58-73) — on a region-sharded
        world slots are the library's and change when an entity changes ranks; every rank is given the same, whole-world arrays.
        Members without an entity channel are left out, identical lists shared."""
        lists: Dict[Tuple[int, ...], int] = {}
        chan, list_of = [], []
        for e in sorted(self._slot):
            chan.append(e)
            if self._handover[e] < 0:
                list_of.append(NO_LIST)
                continue
            key = tuple(sorted(m for m in self.GetHandoverEntities(e) if m in self._slot))
            list_of.append(lists.setdefault(key, len(lists)))
        order = sorted(lists, key=lists.get)
        off = np.zeros(len(order) + 1, dtype=np.uint32)
        for k, key in enumerate(order):
            off[k + 1] = off[k] + len(key)
        mem = np.array([m for key in order for m in key], dtype=np.uint32)
        return off, mem, np.array(chan, dtype=np.uint32), np.array(list_of, dtype=np.uint32)

    # -- internals ---------------------------------------------------------------------------------------------------
    def _pointer(self, t: int) -> Dict[int, int]:
        if t == EntityGroupType_HANDOVER:
            return self._handover
        if t == EntityGroupType_LOCK:
            return self._lock
        raise ValueError(f"unknown EntityGroupType {t}")

    def _new_set(self) -> int:
        self._sets.append(set())
        return len(self._sets) - 1

    def _cascade(self, e: int, t: int, k: int):
        """cascadeGroup (entity.go:80-102) on channel e's controller with the shared instance k."""
        lk = self._lock[e]
        if lk >= 0 and self._sets[lk]:
            return  # "Current entity is already locked, won't cascade."
        h = self._handover[e]
        if t == EntityGroupType_HANDOVER:
            if h >= 0:
                self._sets[k] |= self._sets[h]
            self._handover[e] = k
        else:
            # LOCK outranks HANDOVER: the cascade brings the handover group's entities into the lock group
            if h >= 0:
                self._sets[k] |= self._sets[h]
            if lk >= 0:
                self._sets[k] |= self._sets[lk]
            self._lock[e] = k
