"""oracle/groups.py — TEST INFRASTRUCTURE ONLY (never imported by channeld_amd/).

Literal restatement of the reference's FlatEntityGroupController (pkg/channeld/entity.go:58-224):
one controller per ENTITY channel, each holding a pointer to a handover group and to a lock
group; group instances are SHARED between the controllers a cascade has reached.  Pinned by the
reference's own TestEntityChannelGroupController (pkg/channeld/entity_test.go:11-105), transcribed
in tests/test_oracle_golden.py.

Only entities for which a channel exists have a controller (GetChannel(id) != nil,
entity.go:115-118); ids without a channel (a character's PlayerController / PlayerState in the
reference test) are plain members of group instances.
"""
from __future__ import annotations

HANDOVER, LOCK = 0, 1  # channeldpb.EntityGroupType_HANDOVER / _LOCK (channeld.proto)


class EntityGroup:
    """entity.go:19-37: a set of entity ids (xsync.MapOf used as a set)."""

    def __init__(self):
        self.ids = set()

    def add(self, other):  # EntityGroup.Add, entity.go:28-37
        if other is None:
            return
        self.ids |= other.ids


class FlatEntityGroupController:
    """entity.go:58-224.  `channels` is the registry GetChannel consults: {entity id: controller}."""

    def __init__(self, entity_id: int, channels: dict):
        self.entity_id = entity_id  # Initialize, entity.go:66-68
        self.handover_group = None
        self.lock_group = None
        self.channels = channels
        channels[entity_id] = self

    def uninitialize(self):  # entity.go:70-78 (entity channel removed)
        if self.handover_group is not None:
            self.remove_from_group(HANDOVER, [self.entity_id])
        if self.lock_group is not None:
            self.remove_from_group(LOCK, [self.entity_id])
        self.channels.pop(self.entity_id, None)

    def cascade_group(self, t: int, group: EntityGroup):  # entity.go:80-102
        # "Current entity is already locked, won't cascade."
        if self.lock_group is not None and len(self.lock_group.ids) > 0:
            return
        if t == HANDOVER:
            group.add(self.handover_group)
            self.handover_group = group
        elif t == LOCK:
            # LOCK has higher priority than HANDOVER: the cascade brings the handover group into the lock group
            group.add(self.handover_group)
            group.add(self.lock_group)
            self.lock_group = group

    def add_to_group(self, t: int, entities):  # entity.go:104-158
        if t == HANDOVER:
            if self.handover_group is None:
                self.handover_group = EntityGroup()
            for e in entities:
                self.handover_group.ids.add(e)
                ch = self.channels.get(e)
                if ch is None:
                    continue
                ch.cascade_group(t, self.handover_group)  # all channels of a group share the instance
        elif t == LOCK:
            if self.lock_group is None:
                self.lock_group = EntityGroup()
            for e in entities:
                self.lock_group.ids.add(e)
                ch = self.channels.get(e)
                if ch is None:
                    continue
                ch.cascade_group(t, self.lock_group)

    def remove_from_group(self, t: int, entities):  # entity.go:160-195
        if t == HANDOVER:
            if self.handover_group is None:
                raise ValueError(f"handover group is nil, entityId: {self.entity_id}")
            for e in entities:
                self.handover_group.ids.discard(e)
                ch = self.channels.get(e)
                if ch is not None:
                    ch.handover_group = EntityGroup()  # "Reset the removed entity's entity channel's handover group"
        elif t == LOCK:
            if self.lock_group is None:
                raise ValueError(f"lock group is nil, entityId: {self.entity_id}")
            for e in entities:
                self.lock_group.ids.discard(e)
                ch = self.channels.get(e)
                if ch is not None:
                    ch.lock_group = EntityGroup()

    def get_handover_entities(self):  # entity.go:197-224
        if self.handover_group is None:  # "If AddToGroup is never called, return the entity itself"
            return [self.entity_id]
        if self.lock_group is not None:
            for e in self.handover_group.ids:
                if e in self.lock_group.ids:  # any entity of the handover group is locked: no handover
                    return []
        return sorted(self.handover_group.ids)
