"""Where a blocking call spends its time on config 3 (100 and 1600 trees): upload, pass + wait, download."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bito_amd
from bito_amd import workloads
full = workloads.ds1_gtr_weibull4(16)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(full.substitution, full.site, full.clock), full.patterns, full.weights)
for T in (100, 1600):
    w = full.subset(T)
    for _ in range(3): eng.gradients(w.parent_ids, w.branch_lengths, w.params)
    tu = tr = td = 0.0
    reps = 20
    for _ in range(reps):
        t0 = time.perf_counter(); eng.upload(w.parent_ids, w.branch_lengths, w.params); t1 = time.perf_counter()
        eng.run(True, False); eng.sync(); t2 = time.perf_counter()
        eng.download(True); t3 = time.perf_counter()
        tu += t1 - t0; tr += t2 - t1; td += t3 - t2
    print(f"T={T}: upload {tu/reps*1e3:.3f} ms, run+sync {tr/reps*1e3:.3f} ms, download {td/reps*1e3:.3f} ms")
