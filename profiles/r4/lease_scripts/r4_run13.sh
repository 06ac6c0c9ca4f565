cd $GRAFT_REPO_ROOT
BITO_AMD_LIB=bito_amd/variants/stamps.so timeout 120 python3 - <<'PY' 2>&1 | tail -60
import os, sys
sys.path.insert(0, os.getcwd())
import bito_amd
from bito_amd import _capi, workloads
big = workloads.ds1_gtr_weibull4(64)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(big.substitution, big.site, big.clock), big.patterns, big.weights)
eng.set_kernel(_capi.KERNEL_LDS_PIPE)
eng.upload(big.parent_ids, big.branch_lengths, big.params)
eng.run(True); eng.sync()
eng.run(True); eng.sync()
PY
