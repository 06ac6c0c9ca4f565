"""walk_pipe_kernel timing on config 3 (1600 trees) for forced splits into whole-tree units and runs of tiles.
usage: python scripts/gpu_pipe_units.py   (sets BITO_AMD_PIPE_WHOLE_TREES / BITO_AMD_LDS_TILE_RUN per engine)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bito_amd
from bito_amd import _capi, workloads

big = workloads.ds1_gtr_weibull4(16)
for whole, run in ((0, 5), (1024, 5), (1280, 5), (1536, 5), (1536, 3), (1536, 1), (1600, 5), (1280, 3), (1408, 5)):
    os.environ["BITO_AMD_PIPE_WHOLE_TREES"] = str(whole)
    os.environ["BITO_AMD_LDS_TILE_RUN"] = str(run)
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(big.substitution, big.site, big.clock), big.patterns, big.weights)
    eng.set_kernel(_capi.KERNEL_LDS_PIPE)
    eng.upload(big.parent_ids, big.branch_lengths, big.params)
    eng.time_runs(True, False, 3)
    total, k, launches = eng.time_runs(True, False, 20)
    print(f"whole trees {whole:5d}, runs of {run}: walk kernel {k / launches:.4f} ms, step {total / 20:.4f} ms")
