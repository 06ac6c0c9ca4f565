import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bito_amd
from bito_amd import _capi, workloads
big = workloads.ds1_gtr_weibull4(16)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(big.substitution, big.site, big.clock), big.patterns, big.weights)
eng.set_kernel(_capi.KERNEL_LDS_PIPE)
eng.upload(big.parent_ids, big.branch_lengths, big.params)
eng.time_runs(True, False, 1)
