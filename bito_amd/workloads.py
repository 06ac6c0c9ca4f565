"""Workload builders for the BASELINE.json configurations (harness code).

Everything here produces *inputs* in the engine's wire format: site patterns,
parent-id vectors, branch lengths, parameter rows.  Used by bench.py, by
__graft_entry__.smoke() and by the parity tests so that CPU oracle and GPU engine
see byte-identical inputs.  Data files come from tests/golden/data (copies of the
reference's own test data); nothing here reads /root/reference.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from . import treeio
from .site_pattern import CodonSitePattern, SitePattern

DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")

_MASK = (1 << 64) - 1


class Xoshiro256ss:
    """xoshiro256** seeded through splitmix64 (the generator BASELINE.md names for
    the config-3 branch lengths)."""

    def __init__(self, seed: int):
        x = seed & _MASK
        self.s = []
        for _ in range(4):
            x = (x + 0x9E3779B97F4A7C15) & _MASK
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
            self.s.append(z ^ (z >> 31))

    @staticmethod
    def _rotl(x, k):
        return ((x << k) & _MASK) | (x >> (64 - k))

    def next_u64(self) -> int:
        s = self.s
        result = (self._rotl((s[1] * 5) & _MASK, 7) * 9) & _MASK
        t = (s[1] << 17) & _MASK
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return result

    def uniform(self) -> float:
        return (self.next_u64() >> 11) * (1.0 / (1 << 53))

    def exponential(self, mean: float) -> float:
        return -mean * np.log1p(-self.uniform())


@dataclass
class Workload:
    name: str
    substitution: str
    site: str
    clock: str
    patterns: np.ndarray  # int32 [n][P]
    weights: np.ndarray  # float64 [P]
    parent_ids: np.ndarray  # int32 [T][M-1]
    branch_lengths: np.ndarray  # float64 [T][M]
    params: np.ndarray  # float64 [T][param_count]
    rescaling: bool
    want_gradient: bool
    rates: Optional[np.ndarray] = None

    @property
    def tree_count(self) -> int:
        return int(self.parent_ids.shape[0])

    @property
    def taxon_count(self) -> int:
        return int(self.patterns.shape[0])

    def subset(self, count: int) -> "Workload":
        return Workload(self.name, self.substitution, self.site, self.clock, self.patterns, self.weights,
                        self.parent_ids[:count].copy(), self.branch_lengths[:count].copy(),
                        self.params[:count].copy(), self.rescaling, self.want_gradient,
                        None if self.rates is None else self.rates[:count].copy())

    def shard(self, rank: int, world: int) -> "Workload":
        """Contiguous block of trees for one rank."""
        T = self.tree_count
        lo, hi = (T * rank) // world, (T * (rank + 1)) // world
        return Workload(self.name, self.substitution, self.site, self.clock, self.patterns, self.weights,
                        self.parent_ids[lo:hi].copy(), self.branch_lengths[lo:hi].copy(), self.params[lo:hi].copy(),
                        self.rescaling, self.want_gradient, None if self.rates is None else self.rates[lo:hi].copy())


GTR_FREQS = [0.1, 0.2, 0.3, 0.4]  # the reference's GTR test values, src/rooted_sbn_instance.hpp:357-358
GTR_RATES = [0.05, 0.1, 0.15, 0.20, 0.25, 0.25]


def gtr_weibull_params(tree_count: int, shape: float = 0.5, clock: bool = False) -> np.ndarray:
    row = GTR_FREQS + GTR_RATES + [shape] + ([1.0] if clock else [])
    return np.tile(np.array(row), (tree_count, 1))


def load_ds1(trees: str = "DS1.100_topologies.nwk") -> Tuple[treeio.TreeCollection, SitePattern]:
    path = os.path.join(DATA_DIR, trees)
    tc = treeio.read_nexus_file(path) if trees.endswith(".t") else treeio.read_newick_file(path)
    sp = SitePattern(treeio.read_fasta(os.path.join(DATA_DIR, "DS1.fasta")), tc.taxon_names)
    return tc, sp


def ds1_jc69(replicas: int = 1) -> Workload:
    """BASELINE config 2: DS1, 100 topologies, every branch 0.1, JC69, log-likelihood only."""
    tc, sp = load_ds1()
    pid = np.tile(tc.parent_id_matrix(), (replicas, 1))
    bl = np.full((pid.shape[0], pid.shape[1] + 1), 0.1)
    return Workload("DS1 x100 topologies JC69 LL", "JC69", "constant", "none", sp.patterns, sp.weights, pid, bl,
                    np.zeros((pid.shape[0], 0)), False, False)


def _xoshiro_exponentials(seeds: np.ndarray, count: int, mean: float) -> np.ndarray:
    """Xoshiro256ss(seed).exponential(mean), `count` draws for every seed at once: [len(seeds)][count].  Same
    integer arithmetic on uint64 arrays (wrapping), so the values are those of the scalar class bit for bit."""
    u64 = np.uint64
    x = seeds.astype(np.uint64)
    state = []
    with np.errstate(over="ignore"):
        for _ in range(4):
            x = x + u64(0x9E3779B97F4A7C15)
            z = x.copy()
            z = (z ^ (z >> u64(30))) * u64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> u64(27))) * u64(0x94D049BB133111EB)
            state.append(z ^ (z >> u64(31)))
        rotl = lambda v, k: (v << u64(k)) | (v >> u64(64 - k))  # noqa: E731
        out = np.empty((len(seeds), count))
        s0, s1, s2, s3 = state
        for k in range(count):
            result = rotl(s1 * u64(5), 7) * u64(9)
            t = s1 << u64(17)
            s2 = s2 ^ s0
            s3 = s3 ^ s1
            s1 = s1 ^ s2
            s0 = s0 ^ s3
            s2 = s2 ^ t
            s3 = rotl(s3, 45)
            uniform = (result >> u64(11)).astype(np.float64) * (1.0 / (1 << 53))
            out[:, k] = -mean * np.log1p(-uniform)
    return out


def ds1_gtr_weibull4(replicas: int = 1, first_tree: int = 0, tree_count: Optional[int] = None) -> Workload:
    """BASELINE config 3 (headline): DS1, 100 topologies, GTR + weibull+4, shape 0.5,
    branch lengths Exp(mean 0.1) clamped to [1e-6, 1] from xoshiro256** seeded with
    20240601 + tree index; log-likelihood + branch-length gradient.  Replicas re-seed
    with their own tree index so every tree of the batch is distinct work.  first_tree / tree_count:
    only that block of the 100 x replicas trees (what one rank of a sharded run needs)."""
    tc, sp = load_ds1()
    base = tc.parent_id_matrix()
    total = base.shape[0] * replicas
    count = total - first_tree if tree_count is None else tree_count
    index = np.arange(first_tree, first_tree + count)
    pid = base[index % base.shape[0]].copy()
    M1 = pid.shape[1]
    bl = np.zeros((count, M1 + 1))
    bl[:, :M1] = np.clip(_xoshiro_exponentials(20240601 + index, M1, 0.1), 1e-6, 1.0)
    return Workload("DS1 x100 topologies GTR+weibull4 LL+grad", "GTR", "weibull+4", "none", sp.patterns, sp.weights,
                    pid, bl, gtr_weibull_params(count), False, True)


class _N:
    __slots__ = ("children", "name", "length", "id")

    def __init__(self, name="", children=None, length=None):
        self.children = children or []
        self.name = name
        self.length = length
        self.id = -1


def random_unrooted_tree(n: int, rng: np.random.Generator, mean_bl: float) -> treeio.ParsedTree:
    """Random topology by random pair joining, trifurcating at the root, ids as Node::Polish."""
    nodes = [_N(str(i), length=rng.exponential(mean_bl)) for i in range(n)]
    while len(nodes) > 3:
        i, j = sorted(rng.choice(len(nodes), 2, replace=False))
        b = nodes.pop(j)
        a = nodes.pop(i)
        nodes.append(_N(children=[a, b], length=rng.exponential(mean_bl)))
    root = _N(children=nodes)
    return treeio._polish(root, {str(i): i for i in range(n)})


def simulate_patterns(n: int, P: int, seed: int, mean_bl: float = 0.05) -> np.ndarray:
    """JC69 evolution down one seeded random tree -> int32 [n][P] states."""
    rng = np.random.default_rng(seed)
    tree = random_unrooted_tree(n, rng, mean_bl)
    M = tree.node_count
    children: List[List[int]] = [[] for _ in range(M)]
    for child, parent in enumerate(tree.parent_ids):
        children[parent].append(child)
    states = np.zeros((M, P), dtype=np.int8)
    states[M - 1] = rng.integers(0, 4, P)
    for node in range(M - 1, -1, -1):
        for ch in children[node]:
            p_same = 0.25 + 0.75 * np.exp(-4.0 / 3.0 * tree.branch_lengths[ch])
            change = rng.random(P) >= p_same
            shift = rng.integers(1, 4, P)
            states[ch] = np.where(change, (states[node] + shift) % 4, states[node])
    return states[:n].astype(np.int32)


def synthetic_gtr_weibull4(n: int = 1000, P: int = 10000, tree_count: int = 125, first_tree: int = 0) -> Workload:
    """BASELINE config 4: synthetic n-taxon x P-pattern alignment (seed 1), seeded random
    topologies (seed 2 + tree index) with Exp(0.1) branch lengths, GTR + weibull+4,
    rescaling on, log-likelihood + gradient."""
    patterns = simulate_patterns(n, P, seed=1)
    trees = [random_unrooted_tree(n, np.random.default_rng(2 + first_tree + i), 0.1) for i in range(tree_count)]
    pid = np.stack([t.parent_ids for t in trees]).astype(np.int32)
    bl = np.stack([t.branch_lengths for t in trees])
    bl[:, -1] = 0.0
    return Workload(f"synthetic {n}x{P} GTR+weibull4 LL+grad", "GTR", "weibull+4", "none", patterns, np.ones(P), pid,
                    bl, gtr_weibull_params(tree_count), True, True)


CODON_PARAMS = [0.3, 0.2, 0.25, 0.25, 2.5, 0.3]  # nucleotide frequencies A,C,G,T | kappa, omega


def flua_codon(tree_count: int = 64, site: str = "constant", seed: int = 20240605) -> Workload:
    """BASELINE config 5: fluA.fa read as codons (69 taxa, 329 codon columns -> 242 patterns, 61 states),
    the fluA.tree topology (rooted), GY94 with F1x4 frequencies; every tree of the batch is the same
    topology with its own seeded branch lengths (fluA.tree's, in substitutions per codon site, times
    U(0.5, 1.5)), so each tree is distinct work: log-likelihood + branch-length gradient.  The
    reference has no codon model: model and workload are defined by this build."""
    tc = treeio.read_newick_file(os.path.join(DATA_DIR, "fluA.tree"))
    sp = CodonSitePattern(treeio.read_fasta(os.path.join(DATA_DIR, "fluA.fa")), tc.taxon_names)
    pid = np.tile(tc.parent_id_matrix(), (tree_count, 1))
    rng = np.random.default_rng(seed)
    bl = np.tile(tc.branch_length_matrix(), (tree_count, 1)) * 0.002 * rng.uniform(0.5, 1.5, (tree_count, pid.shape[1] + 1))
    bl[:, -1] = 0.0
    row = CODON_PARAMS + ([0.7] if site != "constant" else [])
    return Workload(f"fluA codon GY94+{site} LL+grad", "GY94", site, "none", sp.patterns, sp.weights, pid, bl,
                    np.tile(np.array(row), (tree_count, 1)), False, True)


def codon_rows(tree_count: int, distinct: int, site: str = "constant", seed: int = 20240607) -> np.ndarray:
    """Parameter rows for the codon workload with `distinct` different models among the trees (the reference hands every
    tree its own parameter row, src/fat_beagle.hpp:173-181): row k of the distinct ones has kappa ~ U(1.5, 4) and
    omega ~ U(0.1, 0.9) from a seeded generator (row 0 = CODON_PARAMS), tree t carries row t % distinct."""
    rng = np.random.default_rng(seed)
    rows = np.tile(np.array(CODON_PARAMS + ([0.7] if site != "constant" else [])), (max(int(distinct), 1), 1))
    rows[1:, 4] = rng.uniform(1.5, 4.0, len(rows) - 1)
    rows[1:, 5] = rng.uniform(0.1, 0.9, len(rows) - 1)
    return np.ascontiguousarray(rows[np.arange(tree_count) % len(rows)])


def other_bits(params: np.ndarray, column: int) -> np.ndarray:
    """The same parameter rows with the last bit of one column flipped upwards (a rate, kappa): another model as far as
    any cache that compares rows is concerned, the same model to 1e-16 for every result."""
    out = np.array(params, dtype=np.float64, copy=True)
    if out.shape[1] > column:
        out[:, column] = np.nextafter(out[:, column], np.inf)
    return np.ascontiguousarray(out)


def postorder_parent_ids(parents) -> np.ndarray:
    """Parent-id vector with arbitrary internal ids (root = the largest id) -> bito's ids: leaves keep theirs, internal
    nodes are numbered in post-order with the children visited in id order (Node::Polish, src/node.cpp:383-402)."""
    parents = [int(x) for x in parents]
    count = len(parents) + 1
    n = (count + 1) // 2
    kids = {}
    for c, p in enumerate(parents):
        kids.setdefault(p, []).append(c)
    new_id, next_id = {}, [n]
    stack = [(count - 1, False)]
    while stack:  # (iterative: the synthetic trees are deep)
        v, done = stack.pop()
        if v < n:
            new_id[v] = v
        elif done:
            new_id[v] = next_id[0]
            next_id[0] += 1
        else:
            stack.append((v, True))
            stack.extend((c, False) for c in sorted(kids[v], reverse=True))
    out = [0] * (count - 1)
    for c, p in enumerate(parents):
        out[new_id[c]] = new_id[p]
    return np.array(out, dtype=np.int32)


def ds1_subsplit_dag(tree_count: int = 10):
    """Path B's workload: the subsplit DAG of DS1 topologies -- the ten trees of
    DS1.subsampled_10.t (the reference's GP test set) or, beyond ten, the first `tree_count` of DS1.100_topologies.nwk --
    each unrooted tree rooted on its first root child, as GPInstance reads rooted trees.  Returns (dag, SitePattern)."""
    from .gp_dag import SubsplitDAG

    if tree_count <= 10:
        tc = treeio.read_nexus_file(os.path.join(DATA_DIR, "DS1.subsampled_10.t"))
    else:
        tc = treeio.read_newick_file(os.path.join(DATA_DIR, "DS1.100_topologies.nwk"))
    sp = SitePattern(treeio.read_fasta(os.path.join(DATA_DIR, "DS1.fasta")), tc.taxon_names)
    pids = []
    for t in tc.trees[:tree_count]:
        p = np.asarray(t.parent_ids).copy()
        M = len(p) + 1
        kids = [c for c in range(M - 1) if p[c] == M - 1]
        q = np.append(p, M)  # a new root above the old one and its first child
        q[kids[0]] = M
        pids.append(postorder_parent_ids(q))
    return SubsplitDAG(len(tc.taxon_names), pids), sp


def seeded_subsplit_dag(tree_count: int = 20, seed: int = 3):
    """A larger Path B workload: the subsplit DAG of `tree_count` seeded random rooted topologies over DS1's 27 taxa (random
    pair joining, seed + tree index) on the DS1 alignment -- random trees share few subsplits, so the DAG grows with
    every tree.  Returns (dag, SitePattern)."""
    from .gp_dag import SubsplitDAG

    tc = treeio.read_nexus_file(os.path.join(DATA_DIR, "DS1.subsampled_10.t"))
    sp = SitePattern(treeio.read_fasta(os.path.join(DATA_DIR, "DS1.fasta")), tc.taxon_names)
    n = len(tc.taxon_names)
    pids = []
    for i in range(tree_count):
        rng = np.random.default_rng(seed + i)
        nodes = list(range(n))
        parents = {}
        next_id = n
        while len(nodes) > 1:
            a, b = sorted(rng.choice(len(nodes), 2, replace=False))
            y, x = nodes.pop(b), nodes.pop(a)
            parents[x] = parents[y] = next_id
            nodes.append(next_id)
            next_id += 1
        pids.append(postorder_parent_ids([parents[c] for c in range(2 * n - 2)]))
    return SubsplitDAG(n, pids), sp
