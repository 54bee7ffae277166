"""A numpy stand-in for the shard engine (TEST INFRASTRUCTURE, CPU only).

channeld_amd/dist.py's tick schedule is engine-agnostic: ingest -> all-to-all ->
import_ -> all-gather -> fanout.  The product engine is HipShardEngine (HIP kernels
through the C-ABI).  This stand-in implements the same three phases with numpy and
the CPU oracle's cell / AOI arithmetic, with the same buffer layouts (send segments
with a header record, published cell tables), so that the world_size-2 gloo tests
can exercise the exchange logic without a GPU.  It does not model fan-out timing:
`fanout` yields each connection's visible set {entity channel : cell(e) in interest}.
"""
import numpy as np
import torch

from channeld_amd.dist import ENTITY_STATE_WORDS, server_of_cell
from oracle import pyoracle as orc

INVALID = 0xFFFFFFFF
ID0 = 0x10000
EID0 = 0x80000


class SimShardEngine:
    def __init__(self, cfg, rank, world, max_entities, cap=256):
        self.cfg, self.rank, self.world, self.N, self.cap = cfg, rank, world, max_entities, cap
        self.g = orc.grid_from_config(cfg)
        self.ncell = int(cfg["GridCols"]) * int(cfg["GridRows"])
        self.chan = np.zeros(0, dtype=np.uint32)
        self.cell = np.zeros(0, dtype=np.uint32)
        self.member = np.zeros(0, dtype=np.uint32)
        self.flags = np.zeros(0, dtype=np.uint32)
        self.subs = []          # (conn_id,)
        self.interest_sets = {} # slot -> sorted cell indices
        self.visible = {}       # conn -> set of entity channels (last tick)
        self.handovers = []     # (chan, src_cell, dst_cell) of the last tick
        self.table_words = 4 * max_entities + self.ncell + 1
        self.table_words += (-self.table_words) % 4
        self.table_bytes = 4 * self.table_words
        self.lists, self.lists_on, self.locked_aborts, self.n_requests = {}, False, 0, 0

    def halo_splits(self):
        """This stand-in ships its whole cell table to every rank (the schedule and the split-size plumbing of dist.py are
        under test, not the band geometry, which tests/test_gpu_shard.py covers on the HIP engine): equal segments, the one
        for rank p at offset p * table_bytes of the send buffer."""
        T = self.table_bytes
        return [T] * self.world, [T] * self.world, [self.rank * T] * self.world

    def _cells(self, x, z):
        ids = orc.channel_ids(self.g, x, z)
        return np.where(ids == 0, INVALID, ids - ID0).astype(np.uint32)

    def spawn(self, chan_id, x, z, flags, sender=None):
        c = self._cells(x, z)
        self.chan = np.concatenate([self.chan, np.asarray(chan_id, dtype=np.uint32)])
        self.cell = np.concatenate([self.cell, c])
        self.member = np.concatenate([self.member, c])
        self.flags = np.concatenate([self.flags, np.asarray(flags, dtype=np.uint32)])

    def add_subscribers(self, conn_ids):
        self.subs = [int(c) for c in conn_ids]

    # ---- handover lists keyed by entity channel id (chd_shard_set_handover_lists): the protocol of k_shard.hip in numpy ----
    REQ_CAP = 64

    def set_handover_lists(self, lists):
        """lists: {entity channel id: [member channel ids]} — the whole world's, the same on every rank"""
        self.lists = {int(k): [int(m) for m in v] for k, v in lists.items()}
        self.lists_on = True
        self.locked_aborts = 0

    def ingest_pre(self, now_ns, x_by_chan, z_by_chan, has_update=None):
        x, z = np.asarray(x_by_chan), np.asarray(z_by_chan)
        k = (self.chan - EID0).astype(np.int64)
        dst = self._cells(x[k], z[k])
        src = self.cell.copy()
        self.cell = dst
        cross = (src != INVALID) & (dst != INVALID) & (src != dst)
        self.handovers, self.locked_aborts = [], 0
        slot_of = {int(c): i for i, c in enumerate(self.chan)}
        req = np.zeros((self.world, (self.REQ_CAP + 1) * 4), dtype=np.uint32)
        for i in np.nonzero(cross)[0]:
            ch = int(self.chan[i])
            lst = self.lists.get(ch)
            if (self.flags[i] & 1) or (lst is not None and len(lst) == 0):
                self.locked_aborts += 1
                continue
            self.handovers.append((ch, int(src[i]), int(dst[i])))
            if lst is None:
                self.member[i] = dst[i]
                continue
            own = int(server_of_cell(self.cfg, np.array([src[i]]))[0])
            if ch in lst:
                self.member[i] = dst[i]
            if own == self.rank or self.world == 1:
                for m in lst:
                    j = slot_of.get(m)
                    if m != ch and j is not None and self.member[j] == src[i]:
                        self.member[j] = dst[i]
            else:  # src's entity map is another rank's: the handover travels there
                n = int(req[own][0])
                assert n < self.REQ_CAP
                req[own][4 * (1 + n): 4 * (2 + n)] = (ch, src[i], dst[i], ch)
                req[own][0] = n + 1
                self.n_requests += 1
        return torch.from_numpy(req.view(np.int32))

    def ingest_post(self, req_recv):
        r = req_recv.numpy().view(np.uint32)
        slot_of = {int(c): i for i, c in enumerate(self.chan)}
        for p in range(self.world):
            for q in range(int(r[p][0])):
                ch, s, d, _ = (int(v) for v in r[p][4 * (1 + q): 4 * (2 + q)])
                for m in self.lists[ch]:
                    j = slot_of.get(m)
                    if m != ch and j is not None and self.member[j] == s:
                        self.member[j] = d
        return self._export()

    def ingest(self, now_ns, x_by_chan, z_by_chan, has_update=None):
        x = np.asarray(x_by_chan)
        z = np.asarray(z_by_chan)
        k = (self.chan - EID0).astype(np.int64)
        dst = self._cells(x[k], z[k])
        src = self.cell.copy()
        self.cell = dst
        cross = (src != INVALID) & (dst != INVALID) & (src != dst)
        move = cross & ((self.flags & 1) == 0)
        self.handovers = [(int(c), int(s), int(d)) for c, s, d in zip(self.chan[move], src[move], dst[move])]
        self.member = np.where(move, dst, self.member)
        return self._export()

    def _export(self):
        send = np.zeros((self.world, (self.cap + 1) * ENTITY_STATE_WORDS), dtype=np.int32)
        if self.world > 1:
            valid = self.member != INVALID
            owner = np.where(valid, server_of_cell(self.cfg, np.where(valid, self.member, 0)), self.rank)
            leave = owner != self.rank
            for i in np.nonzero(leave)[0]:
                seg = send[owner[i]].view(np.uint32)
                n = int(seg[0])
                assert n < self.cap
                rec = seg[(1 + n) * ENTITY_STATE_WORDS:(2 + n) * ENTITY_STATE_WORDS]
                rec[0], rec[1], rec[2], rec[3] = self.chan[i], self.cell[i], self.member[i], self.flags[i]
                seg[0] = n + 1
            keep = ~leave
            self.chan, self.cell, self.member, self.flags = self.chan[keep], self.cell[keep], self.member[keep], self.flags[keep]
        return torch.from_numpy(send)

    def import_(self, recv):
        if recv is not None:
            r = recv.numpy().view(np.uint32)
            for src in range(self.world):
                seg = r[src]
                n = int(seg[0])
                for j in range(n):
                    rec = seg[(1 + j) * ENTITY_STATE_WORDS:(2 + j) * ENTITY_STATE_WORDS]
                    self.chan = np.append(self.chan, np.uint32(rec[0]))
                    self.cell = np.append(self.cell, np.uint32(rec[1]))
                    self.member = np.append(self.member, np.uint32(rec[2]))
                    self.flags = np.append(self.flags, np.uint32(rec[3]))
        assert len(self.chan) <= self.N
        # publish the cell table: entries sorted by member cell, then CSR offsets
        table = np.zeros(self.table_words, dtype=np.uint32)
        inw = np.nonzero(self.member != INVALID)[0]
        order = inw[np.argsort(self.member[inw], kind="stable")]
        ent = table[: 4 * self.N].reshape(self.N, 4)
        ent[: len(order), 0] = self.chan[order]
        ent[: len(order), 3] = order
        counts = np.bincount(self.member[order].astype(np.int64), minlength=self.ncell)
        table[4 * self.N: 4 * self.N + self.ncell + 1] = np.concatenate([[0], np.cumsum(counts)])
        return torch.from_numpy(np.tile(table.view(np.uint8), self.world))

    def interest(self, queries=None, n_queries=0):
        if queries is not None:
            for slot in range(n_queries):
                rc, m = orc.query_channel_ids(self.g, queries[slot])
                if rc == 0:
                    self.interest_sets[slot] = sorted(c - ID0 for c in m)

    def fanout(self, halo_recv):
        t = halo_recv.numpy().view(np.uint32).reshape(self.world, self.table_words)
        owner = server_of_cell(self.cfg, np.arange(self.ncell))
        self.visible = {}
        for slot, conn in enumerate(self.subs):
            vis = set()
            for c in self.interest_sets.get(slot, []):
                tab = t[owner[c]]
                off = tab[4 * self.N: 4 * self.N + self.ncell + 1]
                ent = tab[: 4 * self.N].reshape(self.N, 4)
                vis.update(int(v) for v in ent[off[c]: off[c + 1], 0])
            self.visible[conn] = vis
