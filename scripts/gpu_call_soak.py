"""Soak of the engine level: blocking calls of random sizes (1 to 9000 DS1 trees, so one chunk to seven), gradients /
log-likelihoods / site-model flag in turn, on a one-slot and a two-slot engine, every result compared with a reference
engine that never chunks (and a sample with the CPU checker).  usage: python scripts/gpu_call_soak.py [iterations] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bito_amd
from bito_amd import _capi, workloads
from oracle import oracle

iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
full = workloads.ds1_gtr_weibull4(90)
spec = bito_amd.PhyloModelSpecification(full.substitution, full.site, full.clock)
os.environ["BITO_AMD_CHUNK_FIRST"] = "1000000"
plain = bito_amd.Engine(spec, full.patterns, full.weights)  # one chunk whatever the size
del os.environ["BITO_AMD_CHUNK_FIRST"]
# (the multi-slot engines run every slot on its own issuing thread: round 4)
engines = {"one slot": bito_amd.Engine(spec, full.patterns, full.weights),
           "two slots": bito_amd.Engine(spec, full.patterns, full.weights, devices=[0, 0]),
           "five slots": bito_amd.Engine(spec, full.patterns, full.weights, devices=[0] * 5)}
cpu = oracle.OracleEngine(full.substitution, full.site, full.clock, full.patterns, full.weights, 8)
bad = 0
last_row = None
t0 = time.time()
for it in range(iterations):
    T = int(rng.choice([1, 3, 50, 100, 511, 512, 1023, 1024, 1025, 2000, 3333, 6400, 9000]))
    start = int(rng.integers(0, full.tree_count - T + 1))
    pid = full.parent_ids[start:start + T]
    bl = full.branch_lengths[start:start + T] * rng.uniform(0.5, 2.0)
    par = full.params[start:start + T].copy()
    par[:, -1] = rng.uniform(0.3, 2.0, T)
    # (round 4: small calls keep the last call's model when the parameter rows repeat -- rows that are all equal, equal
    # to the call before, or equal but for one tree, in turn with a row per tree)
    style = it % 5
    if style == 1 or (style in (2, 3) and last_row is None):
        par[:] = par[0]
    elif style in (2, 3):
        par[:] = last_row
        if style == 3:
            par[T // 2, -1] = rng.uniform(0.3, 2.0)
    last_row = par[0].copy()
    mode = it % 3
    want = plain.gradients(pid, bl, par, flags=_capi.GRAD_SITE_MODEL if mode == 2 else 0) if mode else \
        {"log_likelihood": plain.log_likelihoods(pid, bl, par)}
    for name, eng in engines.items():
        got = eng.gradients(pid, bl, par, flags=_capi.GRAD_SITE_MODEL if mode == 2 else 0) if mode else \
            {"log_likelihood": eng.log_likelihoods(pid, bl, par)}
        ok = np.allclose(got["log_likelihood"], want["log_likelihood"], rtol=2e-15, atol=1e-11)
        if mode:
            ok = ok and np.allclose(got["branch_lengths"], want["branch_lengths"], rtol=1e-10, atol=1e-7)
        if mode == 2:
            ok = ok and np.allclose(got["site_model"], want["site_model"], rtol=1e-10, atol=1e-7)
        if not ok:
            bad += 1
            print(f"MISMATCH iteration {it} {name}: T={T} start={start} mode={mode} "
                  f"dLL={np.abs(got['log_likelihood'] - want['log_likelihood']).max():.3e}")
    if it % 4 == 0:  # a sample against the CPU checker
        sel = rng.choice(T, size=min(T, 8), replace=False)
        ref = cpu.log_likelihoods(pid[sel], bl[sel], par[sel])
        if not np.allclose(want["log_likelihood"][sel], ref, rtol=2e-14, atol=1e-10):
            bad += 1
            print(f"MISMATCH with the CPU checker at iteration {it}")
print(f"{iterations} iterations, {bad} bad, {time.time() - t0:.0f} s")
