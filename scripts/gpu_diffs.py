import numpy as np, sys
sys.path.insert(0,'.')
import bito_amd
from bito_amd import workloads
from oracle import oracle
w = workloads.ds1_gtr_weibull4(1)
gpu = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 16)
out = gpu.gradients(w.parent_ids, w.branch_lengths, w.params)
ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params)
d = out['log_likelihood']-ref['log_likelihood']
print('LL diff max', np.abs(d).max(), 'mean', d.mean(), 'ulp(8000)=', np.spacing(8000.0))
print('grad diff max', np.abs(out['branch_lengths']-ref['branch_lengths']).max(), 'rel', (np.abs(out['branch_lengths']-ref['branch_lengths'])/(np.abs(ref['branch_lengths'])+1e-300)).max())
ll2 = gpu.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
print('LL-only vs oracle', np.abs(ll2-ref['log_likelihood']).max(), 'LL-only vs grad LL', np.abs(ll2-out['log_likelihood']).max())
