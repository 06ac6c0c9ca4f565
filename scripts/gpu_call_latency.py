"""Latency of the blocking engine calls (host buffers in, host buffers out: what bito's Engine::Gradients sees)
on DS1 GTR+weibull4 for several batch sizes, next to the resident-batch pass time.  Environment:
BITO_AMD_CHUNK_FIRST / _GROWTH / _CAP / _LANES set the chunking of a blocking call (engine.cpp)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bito_amd
from bito_amd import workloads

sizes = [int(a) for a in sys.argv[1:]] or [1, 10, 100, 400, 1600, 6400]
full = workloads.ds1_gtr_weibull4(-(-max(sizes) // 100))
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(full.substitution, full.site, full.clock), full.patterns, full.weights)
N = 2 * full.taxon_count - 1
for T in sizes:
    w = full.subset(T)
    pid = np.ascontiguousarray(w.parent_ids, dtype=np.int32)
    bls = [np.ascontiguousarray(w.branch_lengths), np.ascontiguousarray(w.branch_lengths * 1.03125)]
    par = np.ascontiguousarray(w.params)
    ll, grad = np.zeros(T), np.zeros((T, N))
    for k in range(3):
        eng.gradients_into(pid, bls[k & 1], par, ll, grad)
    reps = 30
    t0 = time.perf_counter()
    for k in range(reps):
        eng.gradients_into(pid, bls[k & 1], par, ll, grad)
    call_ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for k in range(reps):
        eng.log_likelihoods_into(pid, bls[k & 1], par, ll)
    ll_ms = (time.perf_counter() - t0) / reps * 1e3
    eng.upload(pid, bls[0], par)
    eng.time_runs(True, False, 3)
    total, k, launches = eng.time_runs(True, False, 20)
    print(f"T={T:5d}: gradients_into {call_ms:.3f} ms per call ({T / call_ms:.0f} k trees/s), log_likelihoods_into {ll_ms:.3f} ms; "
          f"resident pass {total / 20:.3f} ms, walk kernel {k / launches:.3f} ms", flush=True)
