"""Comparison of two records of the Brent optimiser's function evaluations (rows of (edge, x = log branch length,
f = negative log-likelihood, kind)): the CPU checker's (oracle/gp_oracle.c, gp_oracle_set_trace) and the device's
(bito_amd_gp_set_optimizer_trace).  Brent is deterministic: two faithful restatements of
Optimization::BrentMinimize(WithGradients) (reference src/optimization.hpp:71-331) whose function values agree to
rounding make the same accept / reject decisions and so visit the same points, except where a decision of the
reference run is itself within rounding of a tie.  Test infrastructure (tests/ and scripts/ only)."""
import numpy as np

# what rounding noise does to the iterates, measured on the CPU checker with its evaluations perturbed by 1e-15
# relative (tests/test_gp.py::test_brent_trace_comparison): the trial points move by up to 2.4e-6 in the log length (a
# parabolic step divides differences of function values, which are 1e-6 themselves at the lower bound of the lengths,
# x = -13.7), by 1e-12 in the length itself.  A flipped decision moves the next point by Brent's own tolerance at the
# least (2^-9 |x| + 2^-11: 5e-4 and more), so the two are far apart.
# The bar on a trial point: |dx| <= X_TOL + T_TOL / t  (1e-6 in the log length where the length is 1e-4 and more,
# growing to 1e-4 at the lower bound of the lengths, 9e-7).
X_TOL, T_TOL = 1e-6, 1e-10
# Near-ties.  Two kinds of decision make Brent's path: comparisons of function values (fu <= fx, <= fw, <= fv) -- a near-tie
# when the values are within VALUE_TIE + 1e-13 |f| of each other (f itself is known to about 1e-15 relative: two hundred
# times less on the workloads of the tests) -- and the comparisons that choose the next point: the convergence test, the
# three tests that accept or reject the parabolic step, the clamps.  Those are formed from DIFFERENCES of function values,
# so rounding noise reaches them amplified; they are near-ties below CHOICE_TIE relative.  One of them is a tie BY
# CONSTRUCTION: a parabolic step that was rejected as a point (worse than the three best) becomes the new bound, the next
# iteration fits the same parabola through the same three points, and `p >= q * (max - x)` compares p with q * (p / q)
# (src/optimization.hpp:121-123) -- hello's edge 3 does it; which way it falls is decided by the last bit of the
# function values, in the reference binary as in any restatement, and the two continuations end Brent's own tolerance
# apart (2^-9 relative in the log length).  The CPU checker records both margins with every evaluation (rows of 6).
VALUE_TIE, CHOICE_TIE = 1e-9, 1e-7


def by_edge(trace):
    """rows of each edge in order (concurrently optimised edges interleave in the device's record); an edge optimised
    several times (several sweeps) keeps its runs apart: a run starts at a row of kind 0"""
    runs = {}
    for row in np.asarray(trace):
        e = int(row[0])
        if row[3] == 0 or e not in runs:
            runs.setdefault(e, []).append([])
        runs[e][-1].append(row)
    return {e: [np.array(r) for r in rs] for e, rs in runs.items()}


def tie_rows(run):
    """indices of a reference run's evaluations around which a decision was a near-tie (rows of 6 with the checker's
    margins; for rows of 4 the value comparisons with the best point so far only)"""
    ties, best = [], None
    for j, row in enumerate(run):
        f, kind = row[2], row[3]
        if len(row) >= 6:
            if row[4] <= CHOICE_TIE or row[5] <= VALUE_TIE + 1e-13 * abs(f):
                ties.append(j)
            continue
        if kind == 0:
            continue
        if kind == 1 or best is None:
            best = f
            continue
        if abs(f - best) <= VALUE_TIE + 1e-13 * abs(best):
            ties.append(j)
        if f <= best:
            best = f
    return ties


def compare(reference, other, x_tol=X_TOL, t_tol=T_TOL):
    """Returns (problems, stats).  problems: strings, empty when `other` follows `reference` -- per edge and run the same
    number of evaluations of the same kinds at points within x_tol + t_tol / length in the log length.  A difference that
    shows up at or after a near-tie decision of the reference run (in the order the reference optimised the edges: a
    flipped decision changes every later edge of a sweep) is explained by it: the comparison stops there without a problem
    and stats["explained_at"] says where.  stats: evaluations compared / in all, largest |dx|, |dt|, |df|, near-ties met."""
    ref, oth = by_edge(reference), by_edge(other)
    problems = []
    stats = {"rows": int(len(reference)), "compared": 0, "max_dx": 0.0, "max_dt": 0.0, "max_df": 0.0, "ties": 0,
             "smallest_choice_margin": float("inf"), "smallest_value_margin": float("inf"), "explained_at": None}
    order, seen = [], {}
    for row in np.asarray(reference):
        if row[3] == 0:
            e = int(row[0])
            seen[e] = seen.get(e, -1) + 1
            order.append((e, seen[e]))
    tie_seen = False
    for e, k in order:
        a = ref[e][k]
        b = oth[e][k] if e in oth and k < len(oth[e]) else np.zeros((0, 4))
        ties = tie_rows(a)
        if a.shape[1] >= 6:
            stats["smallest_choice_margin"] = min(stats["smallest_choice_margin"], float(a[:, 4].min()))
            stats["smallest_value_margin"] = min(stats["smallest_value_margin"], float(a[:, 5].min()))
        stats["ties"] += len(ties)
        mismatch = None  # (index, message)
        for j in range(max(len(a), len(b))):
            if j >= len(a) or j >= len(b):
                mismatch = (j, f"edge {e} run {k}: {len(a)} evaluations in the reference run, {len(b)} in the other")
                break
            dx = abs(a[j, 1] - b[j, 1])
            dt = abs(np.exp(a[j, 1]) - np.exp(b[j, 1]))
            if a[j, 3] != b[j, 3]:
                mismatch = (j, f"edge {e} run {k}: evaluation {j} is of kind {int(b[j, 3])}, the reference's of kind {int(a[j, 3])}")
                break
            if dx > x_tol + t_tol / np.exp(a[j, 1]):
                mismatch = (j, f"edge {e} run {k}: evaluation {j} at x = {b[j, 1]!r} against the reference's {a[j, 1]!r} "
                               f"(f {b[j, 2]!r} against {a[j, 2]!r})")
                break
            stats["compared"] += 1
            stats["max_dx"] = max(stats["max_dx"], float(dx))
            stats["max_dt"] = max(stats["max_dt"], float(dt))
            stats["max_df"] = max(stats["max_df"], float(abs(a[j, 2] - b[j, 2])))
        if mismatch is not None:
            # (a near-tie around evaluation t decides evaluation t itself -- the comparisons that chose it -- or t + 1)
            if tie_seen or any(t <= mismatch[0] for t in ties):
                stats["explained_at"] = mismatch[1]
                break
            problems.append(mismatch[1])
        tie_seen = tie_seen or bool(ties)
    return problems, stats
