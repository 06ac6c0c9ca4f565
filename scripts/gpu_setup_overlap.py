"""What the overlapped set-up costs the traversal: config 3 (1600 trees) with the set-up kernels on their own
stream (default) and in front of the traversal on one stream (BITO_AMD_SERIAL_SETUP=1).
usage: python scripts/gpu_setup_overlap.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bito_amd
from bito_amd import workloads

big = workloads.ds1_gtr_weibull4(16)
for serial in ("0", "1", "2", "0", "1", "2"):
    os.environ["BITO_AMD_SERIAL_SETUP"] = serial
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(big.substitution, big.site, big.clock), big.patterns, big.weights)
    eng.upload(big.parent_ids, big.branch_lengths, big.params)
    eng.time_runs(True, False, 3)
    total, k, launches = eng.time_runs(True, False, 30)
    print(f"serial set-up {serial}: step {total / 30:.4f} ms, walk kernel {k / launches:.4f} ms")
