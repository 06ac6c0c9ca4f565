"""Host-side time line of a small blocking gradients call (BITO_AMD_TRACE_CALL=1 prints it; the last calls are shown)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bito_amd
from bito_amd import workloads

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w = workloads.ds1_gtr_weibull4(1).subset(T)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
for _ in range(30):
    eng.gradients(w.parent_ids, w.branch_lengths, w.params)
t0 = time.perf_counter()
for _ in range(5):
    eng.gradients(w.parent_ids, w.branch_lengths, w.params)
print(f"{(time.perf_counter() - t0) / 5 * 1e3:.4f} ms per call (tracing on)")
