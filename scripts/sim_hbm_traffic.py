"""Arena traffic of walk_hbm_cat_kernel on a tree, counted on the CPU (vector transfers per tree: one transfer = one
partial-likelihood vector of a (tile, category) written to or read from the HBM arena).  Mirrors the kernel's hand-over
rules (bito_amd/csrc/walk_hbm_cat.hip: `last`, the `pend` column, the `fwd` column) and prices alternatives: another
visiting order (heavier subtree first), a deeper stack of pending vectors, pitchforks rebuilt like cherries (`fold`).

    python scripts/sim_hbm_traffic.py [taxa] [trees]
"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bito_amd import workloads  # noqa: E402


def children_of(parent_ids, n):
    """child lists as SetupTopologyCore builds them (ascending ids, detrifurcated)"""
    M = len(parent_ids) + 1
    NI = n - 1
    ch = -np.ones((NI, 2), dtype=np.int64)
    third = -1
    for child in range(M - 1):
        k = parent_ids[child] - n
        if ch[k, 0] < 0:
            ch[k, 0] = child
        elif ch[k, 1] < 0:
            ch[k, 1] = child
        else:
            third = child
    if third >= 0:
        r = M - 1
        a, bb = ch[r - n]
        ch[r - n] = (bb, third)
        ch[r + 1 - n] = (a, r)
    return ch


def shapes(ch, n):
    """(cherries, pitchforks, caterpillars, twins) as walk_hbm_cat.hip tells them: a pitchfork is a tip and a cherry
    under one node, a caterpillar a tip and a pitchfork, twins two cherries; the root is none of them"""
    N = n + len(ch)
    root = N - 1
    cherry = {v for v in range(n, N) if ch[v - n, 0] < n and ch[v - n, 1] < n and v != root}
    fork, cat, twin = set(), set(), set()
    for v in range(n, N - 1):
        c0, c1 = ch[v - n]
        if (c0 < n and c1 in cherry) or (c1 < n and c0 in cherry):
            fork.add(v)
    for v in range(n, N - 1):
        c0, c1 = ch[v - n]
        if (c0 < n and c1 in fork) or (c1 < n and c0 in fork):
            cat.add(v)
        if c0 in cherry and c1 in cherry:
            twin.add(v)
    return cherry, fork, cat, twin


def unstored_nodes(ch, n, fold):
    """nodes without a step and a cell, rebuilt where they are used: cherries; fold >= 1 (True): pitchforks as well
    (round 4); fold 2: and the two shapes of a four-tip subtree, caterpillars and twin cherries (round 6)"""
    cherry, fork, cat, twin = shapes(ch, n)
    out = set(cherry)
    if fold:
        out |= fork
    if int(fold) >= 2:
        # ... unless its sibling is a four-tip subtree with a lower id: a step carries ONE of them, in its first slot
        four = cat | twin
        for v in range(n, n + len(ch)):
            a, b = ch[v - n]
            if a in four and b in four:
                four = four - {max(a, b)}
        out |= four
    return out


def kernel_traffic(ch, n, order=None, depth=1, pre_depth=None, fold=False):
    """(post-order reads, post-order writes, pre-order reads, pre-order writes) of the gradient pass.
    order: the sequence of internal nodes (default: ascending ids); depth: entries of the pending stack;
    fold: pitchforks are not stored either (round 4)."""
    N = n + len(ch)
    root = N - 1
    if order is None:
        order = list(range(n, N))
    cherry = unstored_nodes(ch, n, fold)
    stored = lambda v: v >= n and v not in cherry
    steps = [v for v in order if v not in cherry]
    # ---- post-order
    reads = writes = 0
    pend = []  # youngest last
    last = -1
    for v in steps:
        c0, c1 = ch[v - n]
        if last >= 0 and c0 != last and c1 != last:
            pend.append(last)
            if len(pend) > depth:
                pend.pop(0)  # (already in the arena)
        for c in (c0, c1):
            if not stored(c) or c == last:
                continue
            if c in pend:
                pend.remove(c)
            else:
                reads += 1
        last = v
        if v != root:
            writes += 1
    post = (reads, writes)
    # ---- pre-order (reverse order)
    if pre_depth is not None:
        depth = pre_depth
    reads = writes = 0
    pend = []
    forwarded = -1
    rsteps = steps[::-1]
    for i, v in enumerate(rsteps):
        nxt = rsteps[i + 1] if i + 1 < len(rsteps) else -1
        if v != root:
            if forwarded == v:
                pass
            elif v in pend:
                pend.remove(v)
            else:
                reads += 1
        forwarded = -1
        c0, c1 = ch[v - n]
        for c in (c0, c1):
            if stored(c):
                reads += 1  # its post-order partial, for the message
        for c in (c0, c1):
            if not stored(c):
                continue
            if c == nxt:
                forwarded = c
            else:
                pend.append(c)
                if len(pend) > depth:
                    pend.pop(0)
                    writes += 1
    return post + (reads, writes)


def heavy_first_order(ch, n, light_first=False, fold=False):
    """depth-first post-order that visits the child whose subtree needs more pending vectors first (Sethi-Ullman)"""
    N = n + len(ch)
    root = N - 1
    need = np.zeros(N, dtype=np.int64)
    unstored = unstored_nodes(ch, n, fold)
    cherry = lambda v: v in unstored
    for v in range(n, N):
        c0, c1 = ch[v - n]
        a = 0 if (c0 < n or cherry(c0)) else need[c0]
        b = 0 if (c1 < n or cherry(c1)) else need[c1]
        if c0 < n or cherry(c0) or c1 < n or cherry(c1):
            need[v] = max(a, b, 1)
        else:
            need[v] = max(a, b) if a != b else a + 1
    order = []
    stack = [(root, False)]
    while stack:
        v, done = stack.pop()
        if v < n:
            continue
        if done:
            order.append(v)
            continue
        stack.append((v, True))
        c0, c1 = ch[v - n]
        k0 = need[c0] if c0 >= n else -1
        k1 = need[c1] if c1 >= n else -1
        first, second = (c0, c1) if (k0 >= k1) != light_first else (c1, c0)
        stack.append((second, False))  # popped after first
        stack.append((first, False))
    return order, need


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rows = []
    for t in range(T):
        tree = workloads.random_unrooted_tree(n, np.random.default_rng(2 + t), 0.1)
        ch = children_of(np.asarray(tree.parent_ids), n)
        N = n + len(ch)
        ncherry = sum(1 for v in range(n, N - 1) if ch[v - n, 0] < n and ch[v - n, 1] < n)
        row = {"stored": N - n - ncherry - 1}
        row["ids d1"] = kernel_traffic(ch, n)
        order, need = heavy_first_order(ch, n)
        row["max need"] = int(need.max())
        for depth in (1, 2, 3, 4, 8):
            row[f"heavy d{depth}"] = kernel_traffic(ch, n, order, depth)
        row["heavy d2/1"] = kernel_traffic(ch, n, order, 2, 1)
        forder, _ = heavy_first_order(ch, n, fold=True)
        row["fold d2/1"] = kernel_traffic(ch, n, forder, 2, 1, fold=True)  # the kernel of rounds 4 and 5
        f2order, _ = heavy_first_order(ch, n, fold=2)
        row["fold2 d2/1"] = kernel_traffic(ch, n, f2order, 2, 1, fold=2)  # the kernel since round 6: four-tip subtrees too
        row["stored, fold 2"] = N - n - len(unstored_nodes(ch, n, 2)) - 1
        row["fold d2"] = kernel_traffic(ch, n, forder, 2, fold=True)
        row["fold d4"] = kernel_traffic(ch, n, forder, 4, fold=True)
        row["stored, folded"] = N - n - len(unstored_nodes(ch, n, True)) - 1
        row["ids d2/1"] = kernel_traffic(ch, n, None, 2, 1)
        row["ids d2"] = kernel_traffic(ch, n, None, 2)
        row["ids d4"] = kernel_traffic(ch, n, None, 4)
        rows.append(row)
    keys = [k for k in rows[0] if k not in ("stored", "max need", "stored, folded", "stored, fold 2")]
    print(f"{n} taxa, {T} trees: stored vectors {np.mean([r['stored'] for r in rows]):.0f} "
          f"(pitchforks folded: {np.mean([r['stored, folded'] for r in rows]):.0f}, four-tip subtrees too: "
          f"{np.mean([r['stored, fold 2'] for r in rows]):.0f}), "
          f"Sethi-Ullman need max {max(r['max need'] for r in rows)}")
    for k in keys:
        a = np.array([r[k] for r in rows], dtype=float).mean(axis=0)
        print(f"  {k:10s} post r/w {a[0]:6.0f} {a[1]:6.0f}   pre r/w {a[2]:6.0f} {a[3]:6.0f}   total {a.sum():6.0f}")


if __name__ == "__main__":
    main()
