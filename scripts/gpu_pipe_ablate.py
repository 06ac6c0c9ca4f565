"""Timing-only variants of walk_pipe_kernel (scripts/build_pipe_variants.sh; results are wrong by design) on config 3:
resident gradient passes over 6400 trees, one wave per SIMD (kernel 5) and two (kernel 6).  One line per form:
ms per pass of the walk kernel, and cycles per pattern tile and wave at 2.4 GHz (6400 trees x 15 tiles over 256 CUs).
usage: BITO_AMD_LIB=bito_amd/variants/<name>.so python scripts/gpu_pipe_ablate.py <name>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bito_amd
from bito_amd import _capi, workloads

name = sys.argv[1] if len(sys.argv) > 1 else "shipped"
big = workloads.ds1_gtr_weibull4(64)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(big.substitution, big.site, big.clock), big.patterns, big.weights)
for kern, form in ((_capi.KERNEL_LDS_PIPE, "one wave/SIMD"), (_capi.KERNEL_LDS_PIPE2, "two waves/SIMD")):
    eng.set_kernel(kern)
    eng.upload(big.parent_ids, big.branch_lengths, big.params)
    eng.time_runs(True, False, 3)
    total, k, launches = eng.time_runs(True, False, 10)
    ms = k / 10
    # a workgroup walks 6400 * 15 / 256 tiles one after the other
    cycles = ms * 1e-3 * 2.4e9 / (6400 * 15 / 256)
    print(f"{name:<12} {form:<15} walk {ms:7.3f} ms/pass  {cycles:8.0f} cycles per tile  [{eng.kernel_form()}]", flush=True)
