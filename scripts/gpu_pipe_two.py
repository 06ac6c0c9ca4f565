"""walk_pipe_kernel with two waves per SIMD (round 4) against the CPU checker on config-3 trees, then its timing next
to the one-wave form: resident passes over 1600 and 6400 trees, and the blocking call.
usage: python scripts/gpu_pipe_two.py [replicas of the 100 DS1 topologies for the timing, default 64]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bito_amd
from bito_amd import _capi, workloads
from oracle import oracle

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = workloads.ds1_gtr_weibull4(1)
spec = bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock)
eng = bito_amd.Engine(spec, w.patterns, w.weights)
cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, 8)
ref = cpu.gradients(w.parent_ids, w.branch_lengths, w.params, flags=oracle.GRAD_SITE_MODEL)
for kern in (_capi.KERNEL_LDS_PIPE2, _capi.KERNEL_LDS_PIPE):
    eng.set_kernel(kern)
    out = eng.gradients(w.parent_ids, w.branch_lengths, w.params, flags=_capi.GRAD_SITE_MODEL)
    dg = np.abs(out["branch_lengths"] - ref["branch_lengths"])
    print(f"kernel {kern} {eng.kernel_name()} [{eng.kernel_form()}]: max|dLL| {np.abs(out['log_likelihood'] - ref['log_likelihood']).max():.3e} "
          f"max|dgrad| {dg.max():.3e} max|dsite| {np.abs(out['site_model'] - ref['site_model']).max():.3e}", flush=True)
    ll = eng.log_likelihoods(w.parent_ids, w.branch_lengths, w.params)
    print(f"   log-likelihood only: max|dLL| {np.abs(ll - ref['log_likelihood']).max():.3e}", flush=True)
for count in (16 * 100, reps * 100):
    big = workloads.ds1_gtr_weibull4(count // 100)
    for kern in (_capi.KERNEL_LDS_PIPE, _capi.KERNEL_LDS_PIPE2, _capi.KERNEL_AUTO):
        eng.set_kernel(kern)
        eng.upload(big.parent_ids, big.branch_lengths, big.params)
        for g in (True, False):
            eng.time_runs(g, False, 3)
            total, k, launches = eng.time_runs(g, False, 10)
            print(f"{count} trees kernel={kern} [{eng.kernel_form()}] grad={g}: total {total / 10:.3f} ms/pass, walk kernels {k / 10:.3f} ms/pass "
                  f"({launches // 10} launches)", flush=True)
    pid = np.ascontiguousarray(big.parent_ids, dtype=np.int32)
    bl = [np.ascontiguousarray(big.branch_lengths), np.ascontiguousarray(big.branch_lengths * 1.03125)]
    par = np.ascontiguousarray(big.params)
    ll, grad = np.zeros(count), np.zeros((count, 2 * big.taxon_count - 1))
    for kern in (_capi.KERNEL_LDS_PIPE, _capi.KERNEL_LDS_PIPE2):
        eng.set_kernel(kern)
        for k in range(5):
            eng.gradients_into(pid, bl[k & 1], par, ll, grad)
        t0 = time.perf_counter()
        for k in range(20):
            eng.gradients_into(pid, bl[k & 1], par, ll, grad)
        print(f"{count} trees kernel={kern} blocking call: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms [{eng.kernel_form()}]", flush=True)
