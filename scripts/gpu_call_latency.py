"""Latency of the blocking engine calls (host buffers in, host buffers out: what bito's Engine::Gradients sees)
on DS1 GTR+weibull4 for several batch sizes, next to the resident-batch pass time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bito_amd
from bito_amd import workloads

full = workloads.ds1_gtr_weibull4(16)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(full.substitution, full.site, full.clock), full.patterns, full.weights)
for T in (1, 10, 100, 400, 1600):
    w = full.subset(T)
    for _ in range(3):
        eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    call_ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        ll = eng.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
    ll_ms = (time.perf_counter() - t0) / reps * 1e3
    eng.upload(w.parent_ids, w.branch_lengths, w.params)
    eng.time_runs(True, False, 3)
    total, k, launches = eng.time_runs(True, False, 20)
    print(f"T={T:5d}: gradients() {call_ms:.3f} ms per call ({T / call_ms:.0f} k trees/s), log_likelihoods() {ll_ms:.3f} ms; "
          f"resident pass {total / 20:.3f} ms, walk kernel {k / launches:.3f} ms")
