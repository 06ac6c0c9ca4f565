#!/usr/bin/env python3
"""Independent known answers for the 61-state codon model (BASELINE config 5) -> tests/golden/codon_fixtures.json.

The reference has no codon model, so oracle/gs_oracle.c and the GPU kernels cannot be pinned to a reference value
at S = 61, and they share one algorithm (symmetrised eigendecomposition, P = V exp(L t) V^-1).  This script is a
second, unrelated route to the same numbers:

  * the GY94 rate matrix is built here from its definition (include/bito_amd.h), with its own genetic-code table;
  * P(t) = exp(Q t) by a scaled Taylor series in 80-bit extended precision (numpy longdouble) -- no
    eigendecomposition anywhere; checked below against scipy.linalg.expm (Pade) on every branch;
  * Felsenstein pruning over the UNCOMPRESSED codon columns (no site patterns, no weights), in extended precision;
  * branch gradients analytically (pre-order partials, dP/dt = Q P), spot-checked by central differences.

Extended precision matters: for codons two nucleotide changes apart P_ij(t) = O(t^2) ~ 1e-10 at fluA's branch
lengths, and a double-precision P(t) carries an absolute error of 1e-16 on such entries whichever way it is formed
(DESIGN.md section 3), so a double-precision second route would differ from the first by its own rounding.

Cases: two fluA trees of workloads.flua_codon (constant rate), one with weibull+3 (shape 0.7).  Inputs (branch
lengths, parameters) are stored with the results so the tests do not depend on the generator's random stream.
Run from the repo root:  python scripts/gen_codon_fixtures.py   (about a minute)
"""
import json
import os
import sys

import numpy as np
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bito_amd import treeio, workloads  # noqa: E402  (file readers and the workload's seeded inputs only)

LD = np.longdouble
NUC = "ACGT"
# the standard genetic code, first / second / third position each running over T, C, A, G
AMINO = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
TCAG = "TCAG"


def sense_codons():
    """[(a, b, c)] with A, C, G, T = 0..3, lexicographic, stop codons left out; and the amino acid of each."""
    codons, amino = [], []
    for a in range(4):
        for b in range(4):
            for c in range(4):
                aa = AMINO[16 * TCAG.index(NUC[a]) + 4 * TCAG.index(NUC[b]) + TCAG.index(NUC[c])]
                if aa != "*":
                    codons.append((a, b, c))
                    amino.append(aa)
    assert len(codons) == 61
    return codons, amino


def gy94(freqs, kappa, omega):
    """Q (61x61, one expected substitution per unit time) and pi (F1x4), in extended precision."""
    codons, amino = sense_codons()
    f = np.array(freqs, dtype=LD)
    pi = np.array([f[a] * f[b] * f[c] for a, b, c in codons], dtype=LD)
    pi /= pi.sum()
    Q = np.zeros((61, 61), dtype=LD)
    for i, ci in enumerate(codons):
        for j, cj in enumerate(codons):
            diff = [k for k in range(3) if ci[k] != cj[k]]
            if len(diff) != 1:
                continue
            x, y = ci[diff[0]], cj[diff[0]]
            q = pi[j]
            if {x, y} in ({0, 2}, {1, 3}):  # A<->G, C<->T
                q = q * LD(kappa)
            if amino[i] != amino[j]:
                q = q * LD(omega)
            Q[i, j] = q
    for i in range(61):
        Q[i, i] = -Q[i].sum()
    Q /= -(pi * np.diag(Q)).sum()
    return Q, pi


def expm_ld(A):
    """exp(A) by scaling and squaring around a Taylor series, in extended precision."""
    norm = float(np.abs(A).sum(axis=1).max())
    s = max(0, int(np.ceil(np.log2(max(norm, 1e-300) / 0.25)))) if norm > 0.25 else 0
    B = A / LD(2 ** s)
    E = np.eye(A.shape[0], dtype=LD)
    term = np.eye(A.shape[0], dtype=LD)
    for k in range(1, 40):
        term = term @ B / LD(k)
        E = E + term
        if float(np.abs(term).max()) < 1e-24:
            break
    for _ in range(s):
        E = E @ E
    return E


def weibull_rates(shape, C):
    """WeibullSiteModel::UpdateRates (reference src/site_model.cpp:37-62): quantile midpoints, mean one."""
    r = np.array([(-np.log(LD(1) - LD(2 * i + 1) / LD(2 * C))) ** (LD(1) / LD(shape)) for i in range(C)], dtype=LD)
    return r / r.mean()


def codon_columns(alignment, names):
    """[taxon][column] codon states 0..60, 61 = gap (any non-ACGT symbol, or a stop codon)."""
    codons, _ = sense_codons()
    index = {c: k for k, c in enumerate(codons)}
    rows = []
    for name in names:
        seq = alignment[name].upper()
        row = []
        for p in range(0, len(seq) - len(seq) % 3, 3):
            tri = seq[p:p + 3]
            key = tuple(NUC.find(ch) for ch in tri)
            row.append(index.get(key, 61) if min(key) >= 0 else 61)
        rows.append(row)
    return np.array(rows, dtype=np.int64)


def evaluate(columns, parent_ids, branch_lengths, freqs, kappa, omega, shape=None, C=1):
    """log-likelihood and d/d(branch length) of one rooted tree; ids as bito (leaves 0..n-1, root = 2n-2)."""
    n, L = columns.shape
    N = 2 * n - 1
    Q, pi = gy94(freqs, kappa, omega)
    rates = weibull_rates(shape, C) if shape is not None else np.ones(1, dtype=LD)
    children = [[] for _ in range(N)]
    for child, parent in enumerate(parent_ids):
        children[parent].append(child)
    tips = np.ones((n, 61, L), dtype=LD)
    for t in range(n):
        known = columns[t] < 61
        tips[t][:, known] = 0
        tips[t][columns[t][known], np.nonzero(known)[0]] = 1
    site = np.zeros(L, dtype=LD)
    numer = np.zeros((N, L), dtype=LD)
    worst_pade = 0.0
    for c, rate in enumerate(rates):
        P = [None] * N
        for b in range(N - 1):
            P[b] = expm_ld(Q * (LD(branch_lengths[b]) * rate))
            pade = scipy.linalg.expm(np.asarray(Q, dtype=np.float64) * float(LD(branch_lengths[b]) * rate))
            worst_pade = max(worst_pade, float(np.abs(np.asarray(P[b], dtype=np.float64) - pade).max()))
        post = [None] * N
        msg = [None] * N  # P_b post[b]
        for v in range(N):  # ids are in post-order
            post[v] = tips[v] if v < n else msg[children[v][0]] * msg[children[v][1]]
            if v < N - 1:
                msg[v] = P[v] @ post[v]
        root = N - 1
        like = (pi[:, None] * post[root]).sum(axis=0)
        site += like / LD(len(rates))
        pre = [None] * N
        pre[root] = np.repeat(pi[:, None], L, axis=1)
        for v in range(N - 1, n - 1, -1):
            a, b = children[v]
            for child, sister in ((a, b), (b, a)):
                top = pre[v] * msg[sister]
                numer[child] += (top * (Q @ msg[child])).sum(axis=0) * rate / LD(len(rates))
                pre[child] = P[child].T @ top
    ll = np.log(site).sum()
    grad = np.zeros(N, dtype=LD)
    grad[:N - 1] = (numer[:N - 1] / site[None, :]).sum(axis=1)
    return ll, grad, worst_pade


def main():
    data = workloads.DATA_DIR
    tc = treeio.read_newick_file(os.path.join(data, "fluA.tree"))
    columns = codon_columns(treeio.read_fasta(os.path.join(data, "fluA.fa")), tc.taxon_names)
    cases = []
    for name, site, tree_count in (("flua_gy94_constant", "constant", 2), ("flua_gy94_weibull3", "weibull+3", 1)):
        w = workloads.flua_codon(tree_count, site=site)
        f, kappa, omega = list(w.params[0, :4]), float(w.params[0, 4]), float(w.params[0, 5])
        shape = float(w.params[0, 6]) if site != "constant" else None
        C = 3 if shape is not None else 1
        lls, grads = [], []
        for t in range(tree_count):
            ll, grad, worst = evaluate(columns, w.parent_ids[t], w.branch_lengths[t], f, kappa, omega, shape, C)
            # spot check of the analytic gradient: central differences on three branches
            for b in (0, 70, 135):
                h = 1e-4 * float(w.branch_lengths[t, b])
                bl = w.branch_lengths[t].astype(LD)
                bl[b] += LD(h)
                up = evaluate(columns, w.parent_ids[t], bl, f, kappa, omega, shape, C)[0]
                bl[b] -= 2 * LD(h)
                down = evaluate(columns, w.parent_ids[t], bl, f, kappa, omega, shape, C)[0]
                fd = float((up - down) / (2 * LD(h)))
                assert abs(fd - float(grad[b])) < 1e-6 * max(1.0, abs(fd)), (name, t, b, fd, float(grad[b]))
            print(f"{name} tree {t}: LL {float(ll):.12f}, |grad|max {float(np.abs(grad).max()):.6f}, "
                  f"extended-precision P(t) vs scipy expm: {worst:.2e}", flush=True)
            assert worst < 1e-14
            lls.append(float(ll))
            grads.append([float(g) for g in grad])
        cases.append({"name": name, "substitution": "GY94", "site": site,
                      "parent_ids": w.parent_ids.tolist(), "branch_lengths": w.branch_lengths.tolist(),
                      "params": w.params.tolist(), "log_likelihoods": lls, "branch_gradients": grads})
    out = {"generator": "scripts/gen_codon_fixtures.py",
           "method": "GY94 from its definition; P(t) by an extended-precision Taylor series (no eigendecomposition), "
                     "checked against scipy.linalg.expm; pruning over the uncompressed codon columns of fluA.fa; "
                     "analytic branch gradients spot-checked by central differences",
           "cases": cases}
    path = os.path.join(ROOT, "tests", "golden", "codon_fixtures.json")
    with open(path, "w") as fh:
        json.dump(out, fh)
    print("wrote", path)


if __name__ == "__main__":
    main()
