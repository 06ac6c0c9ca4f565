"""Site-pattern compression.

Mirrors ``SitePattern`` of the reference (src/site_pattern.cpp:16-131): DNA
symbol table with everything that is not ACGT mapped to the gap state 4, column
deduplication with integer multiplicities as double weights, rows indexed by
leaf id.  The reference's pattern order is the iteration order of an
``unordered_map`` (implementation defined); results are sums over patterns, so
first-appearance order is used here.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np

_SYMBOLS = {c: i for i, c in enumerate("ACGT")}
_SYMBOLS.update({c: i for i, c in enumerate("acgt")})
for _c in "-NX?BDHKMRSUVWY":  # reference src/site_pattern.cpp:25-43
    _SYMBOLS[_c] = 4

GAP_STATE = 4
STATE_COUNT = 4


def symbol_vector(sequence: str) -> np.ndarray:
    out = np.empty(len(sequence), dtype=np.int32)
    for i, ch in enumerate(sequence):
        try:
            out[i] = _SYMBOLS[ch]
        except KeyError:
            raise RuntimeError(f"Symbol '{ch}' not known.") from None
    return out


class SitePattern:
    """patterns: int32 [taxon_count][pattern_count]; weights: float64 [pattern_count]."""

    def __init__(self, alignment: Dict[str, str], taxon_names: Sequence[str]):
        rows = []
        for name in taxon_names:
            if name not in alignment:
                raise RuntimeError(f"Taxon '{name}' not found in alignment.")
            rows.append(symbol_vector(alignment[name]))
        if len(alignment) != len(taxon_names):
            raise RuntimeError("Alignment and tree collection have different taxon sets.")
        columns = np.stack(rows, axis=0)  # [n][L]
        uniq: Dict[bytes, int] = {}
        counts = []
        keep = []
        colsT = np.ascontiguousarray(columns.T)
        for pos in range(colsT.shape[0]):
            key = colsT[pos].tobytes()
            idx = uniq.get(key)
            if idx is None:
                uniq[key] = len(keep)
                keep.append(pos)
                counts.append(1.0)
            else:
                counts[idx] += 1.0
        self.patterns = np.ascontiguousarray(columns[:, keep], dtype=np.int32)
        self.weights = np.asarray(counts, dtype=np.float64)

    @classmethod
    def from_arrays(cls, patterns: np.ndarray, weights: np.ndarray) -> "SitePattern":
        obj = cls.__new__(cls)
        obj.patterns = np.ascontiguousarray(patterns, dtype=np.int32)
        obj.weights = np.ascontiguousarray(weights, dtype=np.float64)
        if obj.patterns.ndim != 2 or obj.patterns.shape[1] != obj.weights.shape[0]:
            raise RuntimeError("patterns/weights shape mismatch")
        return obj

    @property
    def taxon_count(self) -> int:
        return int(self.patterns.shape[0])

    @property
    def pattern_count(self) -> int:
        return int(self.patterns.shape[1])

    def partials(self, sequence_idx: int) -> np.ndarray:
        """``SitePattern::GetPartials`` (reference src/site_pattern.cpp:117-131)."""
        st = self.patterns[sequence_idx]
        out = np.zeros((self.pattern_count, STATE_COUNT))
        gap = st >= STATE_COUNT
        out[gap, :] = 1.0
        idx = np.nonzero(~gap)[0]
        out[idx, st[idx]] = 1.0
        return out.reshape(-1)


# ---- codon alphabet (BASELINE config 5) ---------------------------------------------------------
# The reference has no codon model; this is the build's definition.
# 61 sense codons of the standard genetic code, numbered in lexicographic A,C,G,T order with the
# stop codons TAA, TAG, TGA removed; a codon with any non-ACGT symbol, and a stop codon, is the gap
# state 61 (the same rule the reference applies to nucleotides: everything unknown is a gap,
# src/site_pattern.cpp:25-43).
_CODE_TCAG = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
_TO_TCAG = (2, 1, 3, 0)  # A, C, G, T -> index in T, C, A, G order
CODON_STATE_COUNT = 61
CODON_GAP_STATE = 61


def amino_acid(a: int, b: int, c: int) -> str:
    return _CODE_TCAG[16 * _TO_TCAG[a] + 4 * _TO_TCAG[b] + _TO_TCAG[c]]


def _codon_index() -> np.ndarray:
    table = np.full(64, CODON_GAP_STATE, dtype=np.int32)
    state = 0
    for a in range(4):
        for b in range(4):
            for c in range(4):
                if amino_acid(a, b, c) != "*":
                    table[16 * a + 4 * b + c] = state
                    state += 1
    assert state == CODON_STATE_COUNT
    return table


_CODON_INDEX = _codon_index()


def codon_state_vector(sequence: str) -> np.ndarray:
    """Nucleotide string (length a multiple of 3; a trailing partial codon is dropped) -> codon states."""
    nuc = symbol_vector(sequence)
    L = len(nuc) // 3
    tri = nuc[: 3 * L].reshape(L, 3)
    ok = (tri < 4).all(axis=1)
    idx = np.where(ok, 16 * tri[:, 0] + 4 * tri[:, 1] + tri[:, 2], 0)
    return np.where(ok, _CODON_INDEX[idx], CODON_GAP_STATE).astype(np.int32)


class CodonSitePattern(SitePattern):
    """SitePattern over codon columns: patterns int32 [taxon_count][pattern_count] with states 0..60
    and 61 = gap; weights = codon-column multiplicities."""

    def __init__(self, alignment: Dict[str, str], taxon_names: Sequence[str]):
        if len(alignment) != len(taxon_names):
            raise RuntimeError("Alignment and tree collection have different taxon sets.")
        rows = []
        for name in taxon_names:
            if name not in alignment:
                raise RuntimeError(f"Taxon '{name}' not found in alignment.")
            rows.append(codon_state_vector(alignment[name]))
        columns = np.stack(rows, axis=0)
        uniq: Dict[bytes, int] = {}
        counts, keep = [], []
        colsT = np.ascontiguousarray(columns.T)
        for pos in range(colsT.shape[0]):
            key = colsT[pos].tobytes()
            idx = uniq.get(key)
            if idx is None:
                uniq[key] = len(keep)
                keep.append(pos)
                counts.append(1.0)
            else:
                counts[idx] += 1.0
        self.patterns = np.ascontiguousarray(columns[:, keep], dtype=np.int32)
        self.weights = np.asarray(counts, dtype=np.float64)
