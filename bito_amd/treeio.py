"""Input conventions of the hot path: FASTA, Newick/Nexus -> bito node ids.

Host-side harness code.  It reproduces the id
conventions of the reference so that trees, site patterns and gradient vectors
line up index-for-index with bito:

* leaf ids: first-appearance order in the first tree, alphabetical when
  ``sort_taxa`` is set, translate-block order for Nexus
  (reference src/driver.cpp:38-63,120-186,217-226; src/parser.yy:90-108);
* internal ids: post-order (``Node::Polish``, reference src/node.cpp:383-402),
  so the root of a bifurcating tree has id 2n-2;
* branch lengths are indexed by child node id, missing ones are 0
  (reference src/tree.cpp:16-30);
* the wire format of a topology is the parent-id vector
  (``Node::ParentIdVector`` / ``OfParentIdVector``, reference
  src/node.cpp:511-551).
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def read_fasta(path: str) -> Dict[str, str]:
    """``Alignment::ReadFasta`` (reference src/alignment.cpp:49-81)."""
    seqs: Dict[str, List[str]] = {}
    name = None
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                name = line[1:].strip()
                if name in seqs:
                    raise RuntimeError(f"Duplicate taxon '{name}' in {path}")
                seqs[name] = []
            else:
                if name is None:
                    raise RuntimeError(f"{path}: sequence data before first header")
                seqs[name].append(line)
    out = {k: "".join(v) for k, v in seqs.items()}
    lengths = {len(v) for v in out.values()}
    if len(lengths) > 1:
        raise RuntimeError("Sequences of the alignment are not all the same length.")
    return out


@dataclass
class ParsedTree:
    """One tree in wire format: ids follow ``Node::Polish``."""

    parent_ids: np.ndarray  # int32 [node_count - 1]
    branch_lengths: np.ndarray  # float64 [node_count], by child id
    leaf_count: int

    @property
    def node_count(self) -> int:
        return int(self.branch_lengths.shape[0])

    @property
    def rooted(self) -> bool:
        return self.node_count == 2 * self.leaf_count - 1


@dataclass
class TreeCollection:
    trees: List[ParsedTree]
    taxon_names: List[str]  # index = leaf id
    dates: Optional[np.ndarray] = field(default=None)

    def parent_id_matrix(self) -> np.ndarray:
        return np.ascontiguousarray(np.stack([t.parent_ids for t in self.trees]), dtype=np.int32)

    def branch_length_matrix(self) -> np.ndarray:
        return np.ascontiguousarray(np.stack([t.branch_lengths for t in self.trees]), dtype=np.float64)


class _Tok:
    __slots__ = ("s", "i")

    def __init__(self, s: str):
        self.s = s
        self.i = 0

    def peek(self) -> str:
        self._skip()
        return self.s[self.i] if self.i < len(self.s) else ""

    def _skip(self):
        s = self.s
        while self.i < len(s):
            ch = s[self.i]
            if ch.isspace():
                self.i += 1
            elif ch == "[":  # metadata comment
                j = s.find("]", self.i)
                if j < 0:
                    raise RuntimeError("Unterminated comment in Newick string")
                self.i = j + 1
            else:
                break

    def take(self, ch: str):
        if self.peek() != ch:
            raise RuntimeError(f"Newick parse error: expected '{ch}' at {self.i}")
        self.i += 1

    def label(self) -> str:
        self._skip()
        s = self.s
        if self.i < len(s) and s[self.i] in "'\"":
            q = s[self.i]
            j = s.find(q, self.i + 1)
            if j < 0:
                raise RuntimeError("Unterminated quoted label")
            out = s[self.i + 1 : j]
            self.i = j + 1
            return out
        j = self.i
        while j < len(s) and s[j] not in "(),:;[" and not s[j].isspace():
            j += 1
        out = s[self.i : j]
        self.i = j
        return out


class _Node:
    __slots__ = ("children", "name", "length", "id")

    def __init__(self):
        self.children: List["_Node"] = []
        self.name = ""
        self.length: Optional[float] = None
        self.id = -1


def _parse_node(tok: _Tok) -> _Node:
    node = _Node()
    if tok.peek() == "(":
        tok.take("(")
        while True:
            node.children.append(_parse_node(tok))
            if tok.peek() == ",":
                tok.take(",")
                continue
            break
        tok.take(")")
        # optional internal label (ignored, like the reference grammar's labels
        # on inner nodes are not used for ids)
        if tok.peek() not in (":", ",", ")", ";", ""):
            tok.label()
    else:
        node.name = tok.label()
        if node.name == "":
            raise RuntimeError("Newick parse error: empty leaf label")
    if tok.peek() == ":":
        tok.take(":")
        text = tok.label()
        try:
            node.length = float(text)
        except ValueError as exc:
            raise RuntimeError(f"Float conversion failed on branch length '{text}'") from exc
    return node


def _leaf_names(root: _Node) -> List[str]:
    out, stack = [], [root]
    while stack:
        nd = stack.pop()
        if nd.children:
            stack.extend(reversed(nd.children))
        else:
            out.append(nd.name)
    return out


def _polish(root: _Node, taxa: Dict[str, int]) -> ParsedTree:
    """Leaf ids from the taxon map, children ordered by the largest leaf id beneath them (the
    ``Node`` constructor, reference src/node.cpp:33-46: ids must not depend on how the Newick string
    happens to order siblings), then internal ids in post-order (``Node::Polish``, :383-402)."""
    n = len(taxa)

    def postorder() -> List[_Node]:
        order: List[_Node] = []
        stack: List[Tuple[_Node, bool]] = [(root, False)]
        while stack:
            nd, visited = stack.pop()
            if visited or not nd.children:
                order.append(nd)
            else:
                stack.append((nd, True))
                for ch in reversed(nd.children):
                    stack.append((ch, False))
        return order

    seen = set()
    max_leaf: Dict[int, int] = {}  # id(node) -> largest leaf id in its subtree
    for nd in postorder():  # (children before parents in any sibling order)
        if nd.children:
            keys = [max_leaf[id(ch)] for ch in nd.children]
            if len(set(keys)) != len(keys):
                raise RuntimeError("Tie observed between sibling subtrees.\nDo you have a taxon name repeated?")
            nd.children = [ch for _, ch in sorted(zip(keys, nd.children), key=lambda kc: kc[0])]
            max_leaf[id(nd)] = max(keys)
        else:
            if nd.name not in taxa:
                raise RuntimeError(
                    f"Taxon '{nd.name}' is not known in our taxon set.\n"
                    "Either it is missing in the translate block or it didn't appear in the first tree."
                )
            nd.id = taxa[nd.name]
            if nd.id in seen:
                raise RuntimeError(f"Taxon '{nd.name}' appears twice in a tree")
            seen.add(nd.id)
            max_leaf[id(nd)] = nd.id
    if len(seen) != n:
        raise RuntimeError("Tree does not contain every taxon of the collection")
    order = postorder()  # now in the canonical sibling order
    next_id = n
    for nd in order:
        if nd.children:
            nd.id = next_id
            next_id += 1
    node_count = next_id
    parents = np.zeros(node_count - 1, dtype=np.int32)
    lengths = np.zeros(node_count, dtype=np.float64)
    for nd in order:
        if nd.length is not None:
            lengths[nd.id] = nd.length
        for ch in nd.children:
            parents[ch.id] = nd.id
    return ParsedTree(parents, lengths, n)


def parse_newick_strings(lines: Sequence[str], sort_taxa: bool = False,
                         taxa: Optional[Dict[str, int]] = None) -> TreeCollection:
    """``Driver::ParseNewick`` (reference src/driver.cpp:38-63)."""
    roots = []
    for line in lines:
        start = line.find("(")
        if not line.strip() or start < 0:
            continue
        tok = _Tok(line[start:])
        root = _parse_node(tok)
        if tok.peek() == ";":
            tok.take(";")
        roots.append(root)
    if not roots:
        raise RuntimeError("No trees found.")
    if taxa is None:
        names = _leaf_names(roots[0])
        if len(set(names)) != len(names):
            raise RuntimeError("Duplicate taxon name in first tree")
        if sort_taxa:
            names = sorted(names)
        taxa = {name: i for i, name in enumerate(names)}
    trees = [_polish(r, taxa) for r in roots]
    names_by_id = [""] * len(taxa)
    for name, i in taxa.items():
        names_by_id[i] = name
    return TreeCollection(trees, names_by_id)


def read_newick_file(path: str, sort_taxa: bool = False) -> TreeCollection:
    with open(path) as fh:
        return parse_newick_strings(fh.read().splitlines(), sort_taxa)


def read_nexus_file(path: str) -> TreeCollection:
    """``Driver::ParseNexus`` (reference src/driver.cpp:120-186): leaf ids follow
    the order of the translate block; long names are kept for the alignment."""
    with open(path) as fh:
        lines = fh.read().splitlines()
    if not lines or lines[0].strip() != "#NEXUS":
        raise RuntimeError("Putative Nexus file doesn't begin with #NEXUS.")
    i = 1
    while i < len(lines) and lines[i].strip().lower() != "begin trees;":
        i += 1
    if i >= len(lines):
        raise RuntimeError("Finished reading and couldn't find 'begin trees;'")
    i += 1
    if i >= len(lines) or not re.match(r"^\s*translate", lines[i].lower()):
        raise RuntimeError("Missing translate block.")
    i += 1
    item = re.compile(r"^\s*(\d+)\s([^,;]*)([,;]?)$")
    short: Dict[str, int] = {}
    long_names: List[str] = []
    while i < len(lines):
        m = item.match(lines[i])
        if not m:
            if re.match(r"^\s*;\s*$", lines[i]):
                i += 1
            break
        short[m.group(1)] = len(long_names)
        long_names.append(m.group(2).strip().strip("'\""))
        i += 1
        if m.group(3) == ";":
            break
    if not long_names:
        raise RuntimeError("No taxa found in translate block!")
    coll = parse_newick_strings(lines[i:], taxa=short)
    return TreeCollection(coll.trees, long_names)


def tree_from_parent_ids(parent_ids: Sequence[int], branch_lengths: Optional[Sequence[float]] = None) -> ParsedTree:
    """``Tree::OfParentIdVector`` (reference src/tree.cpp:69-72): unit branch
    lengths unless given."""
    parents = np.asarray(parent_ids, dtype=np.int32)
    node_count = parents.shape[0] + 1
    n = int(parents.min())
    bl = np.ones(node_count) if branch_lengths is None else np.asarray(branch_lengths, dtype=np.float64)
    return ParsedTree(parents, bl, n)


def parse_dates_from_taxon_names(names: Sequence[str]) -> np.ndarray:
    """``TaxonNameMunging::ParseDatesFromTagTaxonMap`` + ``MakeDatesRelativeToMaximum``
    (reference src/taxon_name_munging.cpp): the date is the text after the last
    underscore; heights are max(date) - date."""
    dates = []
    for nm in names:
        m = _DATE_RE.match(nm)
        if not m:
            raise RuntimeError("Couldn't parse a date from:" + nm)
        dates.append(float(m.group(1)))
    dates = np.array(dates)
    return dates.max() - dates


_DATE_RE = re.compile(r"^.+_(\d*\.?\d+(?:[eE][-+]?\d+)?)$")


def parse_dates_from_csv(path: str, names: Sequence[str]) -> np.ndarray:
    """``RootedTreeCollection::ParseDatesFromCSVButDontInitializeTimeTrees`` (reference
    src/rooted_tree_collection.cpp:48-63): headerless two-column CSV of quoted taxon names and
    dates; heights are max(date) - date."""
    table = {}
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            name, value = line.rsplit(",", 1)
            table[name.strip().strip('"')] = float(value)
    dates = []
    for nm in names:
        if nm not in table:
            raise RuntimeError("Taxon " + nm + " found in current tree collection but not in " + path)
        dates.append(table[nm])
    dates = np.array(dates)
    return dates.max() - dates
