#!/bin/bash
# rocprofv3 kernel + memory-copy + HIP-API trace of a few blocking calls (where a call's time goes).
# usage: scripts/gpu_call_trace.sh <tag> <trees> [reps]
set -e
TAG=$1; T=$2; REPS=${3:-5}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$TAG
cat > /tmp/trace_calls.py <<PY
import os, sys, time
sys.path.insert(0, "$R")
import numpy as np
import bito_amd
from bito_amd import workloads
T, reps = $T, $REPS
w = workloads.ds1_gtr_weibull4(-(-T // 100)).subset(T)
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
pid = np.ascontiguousarray(w.parent_ids, dtype=np.int32)
bls = [np.ascontiguousarray(w.branch_lengths), np.ascontiguousarray(w.branch_lengths * 1.03125)]
par = np.ascontiguousarray(w.params)
ll, grad = np.zeros(T), np.zeros((T, 2 * w.taxon_count - 1))
for k in range(3):
    eng.gradients_into(pid, bls[k & 1], par, ll, grad)
time.sleep(0.01)
for k in range(reps):
    t0 = time.perf_counter()
    eng.gradients_into(pid, bls[k & 1], par, ll, grad)
    print(f"call {k}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
    time.sleep(0.002)
PY
rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d $R/gpurun_out/$TAG -- python3 /tmp/trace_calls.py > $R/gpurun_out/$TAG/run.log 2>&1
ls -R $R/gpurun_out/$TAG | head -20
