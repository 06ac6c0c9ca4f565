"""Step time of config 3 (1600 trees) against the number of resident workgroups of walk_pipe_kernel
(BITO_AMD_PIPE_WORKGROUPS; the CUs it leaves free serve the set-up kernels of the next pass)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys; sys.path.insert(0, %r)
import bito_amd
from bito_amd import workloads
big = workloads.ds1_gtr_weibull4(16)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(big.substitution, big.site, big.clock), big.patterns, big.weights)
eng.upload(big.parent_ids, big.branch_lengths, big.params)
eng.time_runs(True, False, 3)
total, k, launches = eng.time_runs(True, False, 30)
print("step %%.4f ms, walk kernel %%.4f ms" %% (total / 30, k / launches))
""" % ROOT
for wgs in [int(a) for a in sys.argv[1:]] or [256, 252, 248, 244, 240, 232, 224]:
    env = dict(os.environ, BITO_AMD_PIPE_WORKGROUPS=str(wgs))
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print("workgroups", wgs, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
