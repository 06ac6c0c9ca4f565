"""Experiment: LDS kernel and HBM-arena kernel on two streams at once (two engines)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bito_amd
from bito_amd import workloads
w = workloads.ds1_gtr_weibull4(16)
spec = bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock)
for frac in (1.0, 0.7, 0.6, 0.5):
    ta = int(1600 * frac)
    A = bito_amd.Engine(spec, w.patterns, w.weights); A.set_kernel(2)
    A.upload(w.parent_ids[:ta], w.branch_lengths[:ta], w.params[:ta])
    B = None
    if ta < 1600:
        B = bito_amd.Engine(spec, w.patterns, w.weights); B.set_kernel(1)
        B.upload(w.parent_ids[ta:], w.branch_lengths[ta:], w.params[ta:])
    def step():
        A.run(True)
        if B: B.run(True)
    for _ in range(3): step()
    A.sync(); B and B.sync()
    t0 = time.perf_counter()
    for _ in range(20): step()
    A.sync(); B and B.sync()
    dt = (time.perf_counter() - t0) / 20
    print(f"LDS share {frac:.1f}: {dt*1e3:.3f} ms/step = {1600/dt:.0f} trees/s")
