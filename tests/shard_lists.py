"""Handover lists for the sharded-world tests (tests/test_gpu_shard.py on the HIP engine, tests/test_dist_gloo.py on the numpy
stand-in): the same lists and the same held-back frames on every rank and in the single-world oracle.  TEST INFRASTRUCTURE."""
import numpy as np


def make_lists(sw, ids0, N, seed):
    """Handover lists for a quarter of the world, by entity (index = channel id - 0x80000), every entity in at most one: pairs that
    start in one cell and name each other; notifiers whose list names only a mate (they stay in src's map themselves); triples
    with a member in another cell; empty lists (a locked member: no handover).  Returns ({entity: [members]}, the groups =
    sets of entities that share lists, and the flat arrays of the C call)."""
    rng = np.random.default_rng(seed ^ 0x11575)
    free = np.nonzero((sw.flags == 0) & (ids0 != 0))[0]
    order = free[np.argsort(ids0[free], kind="stable")]
    lists, groups, used = {}, [], set()
    k = 0
    while k + 3 < len(order):
        a, b = int(order[k]), int(order[k + 1])
        if a in used or b in used:
            k += 1
            continue
        r = rng.random()
        if r < 0.10 and ids0[a] == ids0[b]:
            lists[a] = lists[b] = [a, b]
            groups.append([a, b])
            used.update((a, b))
            k += 2
        elif r < 0.14 and ids0[a] == ids0[b]:
            lists[a] = [b]
            groups.append([a, b])
            used.update((a, b))
            k += 2
        elif r < 0.18:
            far = int(order[(k + len(order) // 2) % len(order)])
            if far not in used and far not in (a, b):
                lists[a] = lists[b] = lists[far] = [a, b, far]
                groups.append([a, b, far])
                used.update((a, b, far))
            k += 2
        elif r < 0.20:
            lists[a] = []
            used.add(a)
            k += 1
        else:
            k += 1
    ents = np.array(sorted(lists), dtype=np.uint32)
    off, mem = [0], []
    for e in ents:
        mem += [m + 0x80000 for m in lists[int(e)]]
        off.append(len(mem))
    return lists, groups, (np.array(off, np.uint32), np.array(mem, np.uint32), ents + np.uint32(0x80000), np.arange(len(ents), dtype=np.uint32))


def one_handover_per_group_and_tick(orc, g, groups, x0, z0, frames):
    """Two handovers that touch one entity in the same tick have no defined order (the reference runs them from one goroutine
    per channel; the device runs them concurrently, the oracle in update order): hold back every member but the first of a group
    that would change cells in a tick — it stays where it was and catches up later."""
    px, pz = x0, z0
    out = []
    for (x, z, q, now) in frames:
        x, z = x.copy(), z.copy()
        was, to = orc.channel_ids(g, px, pz), orc.channel_ids(g, x, z)
        for grp in groups:
            moving = [m for m in grp if was[m] != to[m]]
            for m in moving[1:]:
                x[m], z[m] = px[m], pz[m]
        out.append((x, z, q, now))
        px, pz = x, z
    return out
