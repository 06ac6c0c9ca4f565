#!/usr/bin/env python3
"""bench.py -- tree log-likelihoods+gradients/sec on BASELINE.json's headline config.

Workload (config 3): DS1.fasta (27 taxa, 934 site patterns), the 100 topologies of
DS1.100_topologies.nwk replicated R times per GPU with per-tree seeded branch lengths,
GTR + weibull+4 (4 rate categories), FP64, log-likelihood + branch-length gradient.

A "step" is ONE blocking call of the engine's gradients entry point -- the span of the
reference's Engine::Gradients (src/fat_beagle.hpp:173-181): host arrays in
(parent-id vectors, FRESH branch lengths -- two sets alternate, so no step sees the values the
device already holds -- and parameter rows), then on the device per-tree model set-up +
eigendecomposition, transition matrices, post-order partials, pre-order partials + edge
derivatives, per-tree reductions, and host arrays out (log-likelihoods and gradients).  That
rate is `value`.  The compressed alignment is resident in HBM (engine creation is outside the
span, as in the reference).  `resident` reports, beside it, the rate of the same passes over a
batch that stays in HBM (bito_amd_engine_run back to back: no host arrays cross PCIe), which
is what round 1 and 2 reported as `value`.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); every rank owns
R x 100 trees (weak scaling) and calls its own engine; each step ends with an all-reduce of
the summed log-likelihood, the only exchange the path has.

Other workloads: --workload config4 (synthetic 1000 taxa x 10 000 patterns, rescaling on,
1000 trees / ranks), --workload codon (config 5).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 68.0  # measured, scripts: bito_amd/csrc/microbench.hip
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_PATTERN_CEILING_GBS = 5000.0  # measured: random 2 KB pieces, reads and writes in equal parts (hbm_pattern_bench.hip)
FP64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X dense FP64 matrix peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_tree(n: int, P: int, C: int, gradient: bool, S: int = 4) -> float:
    """B_plv = C*P*S*8; LL: (3(n-1)+1) B_plv; LL+grad: (13n-12) B_plv."""
    b_plv = C * P * S * 8
    return ((13 * n - 12) if gradient else (3 * (n - 1) + 1)) * b_plv


def algorithmic_flops_per_tree(n: int, P: int, C: int, S: int) -> float:
    """SURVEY.md section 8d, LL + gradient: C P [(3n-3)(4S^2-S) + (2n-2)(2S^2+3S-1)]."""
    return C * P * ((3 * n - 3) * (4 * S * S - S) + (2 * n - 2) * (2 * S * S + 3 * S - 1))


def usable_cpus():
    """(logical CPUs this process may run on, CPU quota of its cgroup or None): os.cpu_count() ignores both."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, period = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, period = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    return affinity, quota


def physical_cores():
    try:
        seen = set()
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        return len(seen) or None
    except OSError:
        return None


def cpu_baseline(w, seconds: float):
    """The CPU oracle driven like the reference Engine (one instance per thread, dynamic queue over trees,
    src/fat_beagle.hpp:160-181) on a bounded sample of the same workload.  Thread counts are swept -- 1, powers
    of four, the physical cores, every usable logical CPU (affinity mask and cgroup quota, not os.cpu_count()) --
    with calls long enough that the per-call thread start-up does not show; the best sustained rate is `value`,
    the one-thread rate is reported beside it."""
    affinity, quota = usable_cpus()
    limit = max(1, min(affinity, int(quota + 0.5)) if quota else affinity)
    phys = physical_cores()
    if w.substitution == "GY94":
        from oracle import gs

        make = lambda threads: gs.GsOracleEngine(w.substitution, w.site, w.patterns, w.weights, threads)  # noqa: E731
        source = "oracle/gs_oracle.c"
    else:
        from oracle import oracle

        make = lambda threads: oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, threads)  # noqa: E731
        source = "oracle/bito_oracle.c"

    def sample(count):
        reps = -(-count // w.tree_count)
        return (np.tile(w.parent_ids, (reps, 1))[:count], np.tile(w.branch_lengths, (reps, 1))[:count],
                np.tile(w.params, (reps, 1))[:count])

    def rate(threads, budget):
        """sustained trees/s of `threads` threads over about `budget` seconds (after a warm-up call that makes
        every thread touch its own buffers); calls of at least a quarter of the budget each"""
        eng = make(threads)
        run = eng.gradients if w.want_gradient else eng.log_likelihoods
        chunk = max(8 * threads, 32)  # (every thread gets trees: first touch of its own buffers)
        pid, bl, par = sample(chunk)
        t0 = time.perf_counter()
        run(pid, bl, par, rescaling=w.rescaling)
        per_tree = (time.perf_counter() - t0) / chunk  # (an overestimate: first touch included)
        chunk = int(min(max(chunk, budget / 4 / max(per_tree, 1e-9)), 1 << 16))
        count, dt = 0, 0.0
        while dt < budget:
            pid, bl, par = sample(chunk)
            t0 = time.perf_counter()
            run(pid, bl, par, rescaling=w.rescaling)
            took = time.perf_counter() - t0
            count += chunk
            dt += took
            chunk = int(min(max(2 * threads, chunk * budget / 3 / max(took, 1e-3)), 1 << 17))
        return count / dt, count, dt

    candidates = sorted({c for c in (1, 4, 16, 64, phys or 0, limit) if 1 <= c <= limit})
    sweep_budget = seconds * 0.4 / max(len(candidates), 1)
    sweep = {}
    for c in candidates:
        sweep[c] = rate(c, sweep_budget)[0]
    best = max(sweep, key=sweep.get)
    value, count, dt = rate(best, seconds * 0.6)
    value = max(value, sweep[best])
    return {"value": value, "unit": "trees/s", "cores": best, "kind": "port",
            "one_thread": sweep.get(1), "per_thread_at_cores": value / best,
            "sweep_trees_per_s": {str(k): v for k, v in sweep.items()},
            "host": {"logical_cpus": os.cpu_count(), "affinity": affinity, "cgroup_quota": quota,
                     "physical_cores": phys},
            "sample": f"{count} trees of the same workload, {dt:.1f} s, {source} with {best} threads, the best of "
                      f"the thread counts {candidates} (one engine instance per thread, dynamic queue over trees, "
                      "as the reference's Engine; FP64 restatement of the BEAGLE CPU path -- the reference binary "
                      "cannot be built here)"}


def _committed_row(filename: str, kernel: str, workload: str, trees_per_launch: float):
    """the row of a hand-maintained summary under profiles/ (each names the rocprofv3 pass it was read from) for this
    kernel and workload, the one taken nearest to this launch size"""
    try:
        with open(os.path.join(ROOT, "profiles", filename)) as fh:
            rows = [r for r in json.load(fh) if r["kernel"] == kernel and r.get("workload", "ds1") == workload]
    except (OSError, ValueError, KeyError):
        return None
    if not rows:
        return None
    return min(rows, key=lambda r: abs(r.get("trees_per_launch", trees_per_launch) - trees_per_launch))


def roofline_object(kernel, n, P, C, S, want_gradient, trees_per_launch, avg_kernel_s, workload_key):
    """`roofline` of the dominant kernel.  Four views, every one labelled with where its numerator comes from:
    the algorithmic bytes and flops of SURVEY.md 8d (an op-by-op model: every partial crosses HBM, every child message
    is a full matrix-vector product) and what the kernel really did -- HBM bytes and matrix instructions from committed
    rocprofv3 counter passes (profiles/traffic.json, profiles/executed.json; replayed per tree, the files name the
    passes).  `achieved` / `frac` follow the bound: the LDS-resident walks are priced on the FP64 matrix peak by
    algorithmic flops; the 61-state walk on the matrix peak by EXECUTED flops (the algorithmic model counts tip
    children as full 61 x 61 products, which the kernel rightly does not compute: that model exceeds the peak and is
    kept as a ratio, not a fraction); the HBM-arena walks on HBM by measured traffic (fused: fewer bytes than op by op)."""
    alg_bytes_tree = algorithmic_bytes_per_tree(n, P, C, want_gradient, S)
    alg_flops_tree = algorithmic_flops_per_tree(n, P, C, S)
    alg_gbps = alg_bytes_tree * trees_per_launch / avg_kernel_s / 1e9
    alg_tflops = alg_flops_tree * trees_per_launch / avg_kernel_s / 1e12
    trow = _committed_row("traffic.json", kernel, workload_key, trees_per_launch)
    traffic = trow["hbm_bytes_per_launch"] / trow["trees_per_launch"] * trees_per_launch if trow else None
    erow = _committed_row("executed.json", kernel, workload_key, trees_per_launch)
    executed = None
    if erow:
        flops = erow["matrix_instructions_per_tree"] * erow["flops_per_instruction"] * trees_per_launch
        executed = {"achieved": flops / avg_kernel_s / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flops / avg_kernel_s / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                    "matrix_instructions_per_tree": erow["matrix_instructions_per_tree"],
                    "flops_per_instruction": erow["flops_per_instruction"], "source": erow["source"]}
    hbm_measured = None
    if traffic:
        hbm_measured = {"achieved": traffic / avg_kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBS}
    model = {"note": "SURVEY.md 8d, op-by-op: ratios to the peaks, not fractions (a fused kernel moves fewer bytes; tip "
                     "children need no matrix product)",
             "algorithmic_bytes_per_tree": alg_bytes_tree, "algorithmic_flops_per_tree": alg_flops_tree,
             "bytes_rate_GBps": alg_gbps, "bytes_ratio_to_hbm_peak": alg_gbps / HBM_PEAK_GBS,
             "flops_rate_TFLOPs": alg_tflops, "flops_ratio_to_matrix_peak": alg_tflops / FP64_MATRIX_PEAK_TFLOPS}
    common = {"traffic": traffic, "traffic_source": trow["source"] if trow else None, "kernel": kernel,
              "avg_kernel_ms": avg_kernel_s * 1e3, "trees_per_launch": trees_per_launch}
    if kernel in ("walk_pipe_kernel", "walk_lds_kernel", "walk_tree_kernel"):
        # every partial stays in LDS (HBM sees 0.2 % of the byte model): bounded by the FP64 matrix / vector pipe
        out = {"bound": "mfma", "achieved": alg_tflops, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": alg_tflops / FP64_MATRIX_PEAK_TFLOPS, "numerator": "algorithmic flops (SURVEY.md 8d)",
               "sustained_peak": FP64_MFMA_PEAK_TFLOPS, "frac_of_sustained": alg_tflops / FP64_MFMA_PEAK_TFLOPS,
               "algorithmic_flops_per_tree": alg_flops_tree, "executed": executed, "hbm_measured": hbm_measured}
    elif S != 4 and executed is not None:
        out = {"bound": "mfma", "achieved": executed["achieved"], "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": executed["frac"], "numerator": "executed matrix instructions x flops per instruction (counters)",
               "executed": executed, "hbm_measured": hbm_measured,
               "note": "v_mfma_f64_16x16x4 sustains 47.6 TFLOP/s on this part (profiles/r1_microbench.json)"}
    elif hbm_measured is not None:
        out = {"bound": "hbm", "achieved": hbm_measured["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": hbm_measured["frac"], "numerator": "HBM bytes measured by rocprofv3 (FETCH_SIZE x 2 + WRITE_SIZE)",
               "executed": executed,
               # what this memory takes from independent waves that read and write 2 KB pieces at random places in equal
               # parts (the arena walk's pattern): bito_amd/csrc/hbm_pattern_bench.hip, profiles/r4_hbm/hbm_pattern_bench.txt
               "pattern_ceiling": {"GBps": HBM_PATTERN_CEILING_GBS, "frac": hbm_measured["achieved"] / HBM_PATTERN_CEILING_GBS,
                                   "source": "profiles/r4_hbm/hbm_pattern_bench.txt (1 read + 1 write, 2 KB pieces: 4.96-5.07 TB/s "
                                             "whatever the piece size; reads alone 6.9, a sequential copy 6.3)"}}
    else:  # no committed counter pass for this kernel: the op-by-op model alone, held to the contract's frac <= 1
        out = {"bound": "hbm", "achieved": alg_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": alg_gbps / HBM_PEAK_GBS if alg_gbps <= HBM_PEAK_GBS else None,
               "numerator": "algorithmic bytes (SURVEY.md 8d); no measured traffic committed for this kernel"}
    return {**out, **common, "model": model}


def check_against_oracle(w, out_ll, out_grad, count=8):
    """(outside every timed region) `count` trees spread over the timed batch, the engine's last results against the CPU
    checker on the inputs of the last step: max |dLL| and max |dgrad| go into the line"""
    T = w.tree_count
    sel = np.unique(np.linspace(0, T - 1, min(count, T)).astype(int))
    if w.substitution == "GY94":
        from oracle import gs

        cpu = gs.GsOracleEngine(w.substitution, w.site, w.patterns, w.weights, min(len(sel), 8))
    else:
        from oracle import oracle

        cpu = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, min(len(sel), 8))
    bl = w.last_branch_lengths[sel]
    par = getattr(w, "last_params", w.params)[sel]
    if w.want_gradient:
        ref = cpu.gradients(w.parent_ids[sel], bl, par, rescaling=w.rescaling)
        ref_ll, ref_grad = ref["log_likelihood"], ref["branch_lengths"]
    else:
        ref_ll, ref_grad = cpu.log_likelihoods(w.parent_ids[sel], bl, par, rescaling=w.rescaling), None
    dll = float(np.max(np.abs(out_ll[sel] - ref_ll) / (1.0 + 2e-4 * np.abs(ref_ll))))  # 1e-10 + 2e-14 |LL|, as the tests
    res = {"trees_checked": [int(t) for t in sel], "max_dll": float(np.max(np.abs(out_ll[sel] - ref_ll))),
           "max_dll_scaled": dll, "checker": "the CPU oracle (oracle/), on the inputs of the last timed step"}
    if ref_grad is not None:
        res["max_dgrad"] = float(np.max(np.abs(out_grad[sel] - ref_grad)))
        res["max_dgrad_relative"] = float(np.max(np.abs(out_grad[sel] - ref_grad) / (1.0 + 1e-3 * np.abs(ref_grad))))
    return res


def gp_workload(args):
    """Path B: one step = the branch lengths set, then the three schedules
    GPInstance::EstimateBranchLengths alternates (reference src/gp_instance.cpp:241-308, src/gp_dag.cpp:78-121,177-304) --
    PopulatePLVs, ComputeLikelihoods and ONE BranchLengthOptimization sweep over every edge of the DAG -- through
    bito_amd_gp_process_operations, the per-GPCSP log-likelihoods read back at the end.  One DAG is one shared structure:
    it does not shard by trees, so this workload is one GPU, replicas only.  Reported: edges x site
    patterns per second, ms per step and per schedule, the dependency levels the executor cuts the schedules into, the
    algorithmic bytes they move, and the same step on the CPU oracle (oracle/gp_oracle.c, one thread)."""
    import torch

    from bito_amd import gp, workloads

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("--workload gp is one DAG on one GPU (it does not shard by trees): run it without torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible")
    dag, sp = workloads.ds1_subsplit_dag(10) if args.gp_dag == "ds1" else workloads.seeded_subsplit_dag(20)
    P = sp.patterns.shape[1]
    bl0 = np.random.default_rng(1).uniform(0.01, 0.2, dag.gpcsp_count)
    schedules = {"populate_plvs": dag.populate_plvs(), "compute_likelihoods": dag.compute_likelihoods(),
                 "branch_length_optimization": dag.branch_length_optimization()}

    def step(eng):
        eng.set_branch_lengths(bl0)  # (every step the same work: a sweep moves the lengths)
        eng.reset_optimization_count()
        for s in schedules.values():
            eng.process_operations(s)

    eng = gp.GPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)
    eng.set_sbn_parameters(dag.uniform_on_topological_support_prior())
    for _ in range(args.warmup):
        step(eng)
    eng.get_per_gpcsp_log_likelihoods()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(eng)
    per_edge = eng.get_per_gpcsp_log_likelihoods()  # (a device-to-host copy on the executor's stream: the fence)
    elapsed = time.perf_counter() - t0
    after = eng.get_branch_lengths()
    # the schedules one by one (outside the timed region), each fenced by a read-back
    parts = {}
    for name, s in schedules.items():
        eng.set_branch_lengths(bl0)
        eng.reset_optimization_count()
        if name != "populate_plvs":
            eng.process_operations(schedules["populate_plvs"])
        eng.get_branch_lengths()
        p0 = time.perf_counter()
        for _ in range(5):
            eng.process_operations(s)
        eng.get_branch_lengths()
        parts[name] = (time.perf_counter() - p0) / 5 * 1e3
    # what the schedules are made of: operations by kind, bytes by the op-by-op model (a PLV is 4 x P doubles + P counts)
    plv_bytes = 4 * P * 8 + P * 4
    names = {0: "ZeroPLV", 1: "SetToStationaryDistribution", 2: "IncrementWithWeightedEvolvedPLV", 3: "Multiply",
             4: "Likelihood", 5: "OptimizeBranchLength", 6: "UpdateSBNProbabilities", 7: "ResetMarginalLikelihood",
             8: "IncrementMarginalLikelihood", 9: "PrepForMarginalization"}  # (gp_operation.hpp:162-167)
    reads_writes = {0: 1, 1: 1, 2: 3, 3: 3, 4: 2, 5: 2, 6: 0, 7: 0, 8: 1, 9: 2}  # PLVs an operation reads + writes
    kinds, alg_bytes, op_total = {}, 0, 0
    for s in schedules.values():
        for op in s.ops:
            kinds[names.get(op[0], str(op[0]))] = kinds.get(names.get(op[0], str(op[0])), 0) + 1
            alg_bytes += reads_writes.get(op[0], 2) * plv_bytes
            op_total += 1
    ms = elapsed / args.steps * 1e3
    out = {
        "metric": "GP subsplit-DAG sweeps: edges x site patterns per second (PopulatePLVs + ComputeLikelihoods + one "
                  "BranchLengthOptimization sweep)",
        "value": dag.gpcsp_count * P * args.steps / elapsed, "unit": "edge-patterns/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"Path B: subsplit DAG of {'the ten DS1 golden trees' if args.gp_dag == 'ds1' else '20 seeded random topologies over the DS1 taxa'}"
                               f" ({dag.node_count} nodes, {dag.gpcsp_count} edges, 27 taxa, {P} patterns, JC69), branch lengths U(0.01, 0.2) set every step",
                   "operations_per_step": op_total, "operations_by_kind": kinds, "ms_by_schedule": parts,
                   "multi_gpu": "one DAG is one shared structure: replicas only (SURVEY.md 8e)",
                   **({"switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("BITO_AMD_") and k != "BITO_AMD_LIB"}}
                      if any(k.startswith("BITO_AMD_") and k != "BITO_AMD_LIB" for k in os.environ) else {})},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "numerator": "op-by-op bytes of the step's operations (PLVs read + written, 4 x P doubles + P counts each)",
                     "algorithmic_bytes_per_step": alg_bytes,
                     "note": "the whole arena of this DAG is %.1f MB and stays in L2 / Infinity Cache: the step is bound by its "
                             "dependent chains, not by bytes -- an edge's optimisation is one workgroup running some thirty "
                             "function evaluations one after the other (39 us per edge, 81 %% of the step's GPU time), and the "
                             "edges of a sweep depend on one another; the PLV schedules cost 1.35 us per dependency level "
                             "(profiles/r4_gp_ds1_kernel_stats.csv, DESIGN.md section 9)" % (6 * dag.node_count * plv_bytes / 1e6)},
        "results": {"sum_per_gpcsp_log_likelihood": float(np.sum(per_edge)),
                    "mean_abs_branch_length_change": float(np.mean(np.abs(after - bl0)))},
    }
    if not args.no_cpu_baseline:
        from oracle import gp as ogp  # (the checker, timed on the host: the cpu_baseline leg)

        cpu = ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)
        cpu.set_sbn_parameters(dag.uniform_on_topological_support_prior())
        step(cpu)
        count, c0 = 0, time.perf_counter()
        while time.perf_counter() - c0 < min(args.cpu_seconds, 20.0) or count < 2:
            step(cpu)
            count += 1
        dt = time.perf_counter() - c0
        ref_edge, ref_after = cpu.get_per_gpcsp_log_likelihoods(), cpu.get_branch_lengths()
        # the two schedules without an optimiser in them, from the same branch lengths: held to the likelihood bar
        fixed = []
        for engine in (eng, cpu):
            engine.set_branch_lengths(bl0)
            engine.process_operations(schedules["populate_plvs"])
            engine.process_operations(schedules["compute_likelihoods"])
            fixed.append(engine.get_per_gpcsp_log_likelihoods())
        d_fixed = float(np.max(np.abs(fixed[0] - fixed[1]) / (1.0 + 2e-4 * np.abs(fixed[1]))))
        if d_fixed > 1e-10:
            raise SystemExit(f"per-GPCSP log-likelihoods differ from the CPU checker's: {d_fixed}")
        out["cpu_baseline"] = {"value": dag.gpcsp_count * P * count / dt, "unit": "edge-patterns/s", "cores": 1, "kind": "port",
                               "ms_per_step": dt / count * 1e3,
                               "sample": f"{count} steps of the same workload, {dt:.1f} s, oracle/gp_oracle.c on one thread "
                                         "(the reference's GPEngine is single-threaded Eigen code, src/gp_engine.cpp)"}
        out["parity"] = {"max_d_per_gpcsp_log_likelihood_fixed_lengths": float(np.max(np.abs(fixed[0] - fixed[1]))),
                         "max_d_scaled_fixed_lengths": d_fixed,
                         "after_one_sweep": {"max_d_branch_length": float(np.max(np.abs(after - ref_after))),
                                             "max_d_per_gpcsp_log_likelihood": float(np.max(np.abs(per_edge - ref_edge))),
                                             "bars": {"branch_length": 1e-6, "per_gpcsp_log_likelihood": 1e-6},
                                             "note": "Brent (the reference's default optimiser) is deterministic: rounding noise of 1e-15 in "
                                                     "the function values moves the optimised lengths by 1e-10 unless a decision of the sweep is "
                                                     "itself within rounding of a tie (tests/test_gp.py::test_brent_trace_comparison, tests/gp_trace.py)"},
                         "checker": "oracle/gp_oracle.c on the same schedules and branch lengths"}
        # Brent has one tie by construction (a rejected parabolic step becomes the bracket's bound, the same parabola is
        # fitted again and p is compared with q * (p / q): DESIGN.md section 9) -- about one edge in a thousand meets it,
        # and which way it falls is decided by the last bit of the function values.  The checker records how far from a
        # tie its decisions were: when one of them was within rounding, the sweep is held to Brent's own tolerance
        # (2^-9 relative in the log length and 2^-11) instead of to the iterates.
        cpu.start_optimizer_trace(1 << 18)
        step(cpu)
        rows = cpu.optimizer_trace()
        near_ties = int(np.sum((rows[:, 4] <= 1e-7) | (rows[:, 5] <= 1e-9 + 1e-13 * np.abs(rows[:, 2])))) if len(rows) else 0
        sweep = out["parity"]["after_one_sweep"]
        sweep["near_tie_decisions_of_the_checker"] = near_ties
        tight = sweep["max_d_branch_length"] < 1e-6 and sweep["max_d_per_gpcsp_log_likelihood"] < 1e-6
        with np.errstate(divide="ignore", invalid="ignore"):
            d_log = np.abs(np.log(after) - np.log(ref_after))
            loose = bool(np.all((d_log <= 4 * (2.0 ** -9 * np.abs(np.log(ref_after)) + 2.0 ** -11)) | (after == ref_after)))
        sweep["held_to"] = "the checker's lengths (1e-6)" if tight else "Brent's tolerance (the checker met a near-tie)"
        if not (tight or (near_ties > 0 and loose)) and not os.environ.get("BENCH_ABLATION"):
            print(json.dumps(out), flush=True)
            raise SystemExit(f"after one Brent sweep the device is {sweep['max_d_branch_length']:.3e} from the CPU checker in the branch "
                             f"lengths, {sweep['max_d_per_gpcsp_log_likelihood']:.3e} in the per-GPCSP log-likelihoods "
                             f"({near_ties} near-tie decisions in the checker's sweep)")
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


def large_batch_child(args):
    """`large_batch` of the ds1 line, in a process of its own (bench.py --large-batch-child): one engine, --large-batch x the
    trees of `value` in ONE blocking gradients call, parameter rows changing every call; the first trees (those of `value`'s
    batch) again in a call of their own.  Prints one JSON object."""
    import bito_amd
    from bito_amd import workloads

    big = workloads.ds1_gtr_weibull4(args.replicas * args.large_batch)
    T = 100 * args.replicas
    Tb = big.tree_count
    N = 2 * big.patterns.shape[0] - 1
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(big.substitution, big.site, big.clock), big.patterns, big.weights,
                          device_id=int(os.environ.get("BENCH_CHILD_DEVICE", "0")),
                          host_threads=int(os.environ.get("BENCH_CHILD_HOST_THREADS", "0")))
    eng.set_kernel(args.kernel)
    pid_b = np.ascontiguousarray(big.parent_ids, dtype=np.int32)
    par_b = np.ascontiguousarray(big.params, dtype=np.float64)
    par_sets_b = [par_b, workloads.other_bits(par_b, 4)]
    bl_sets_b = [np.ascontiguousarray(big.branch_lengths, dtype=np.float64),
                 np.ascontiguousarray(big.branch_lengths * 1.03125, dtype=np.float64)]
    ll_b, grad_b = np.zeros(Tb), np.zeros((Tb, N))
    for k in range(2):
        eng.gradients_into(pid_b, bl_sets_b[k & 1], par_sets_b[k & 1], ll_b, grad_b)
    eng.sync()
    reps = max(2, min(args.steps, 5))
    eng.kernel_timing(True)
    b0 = time.perf_counter()
    for k in range(reps):
        eng.gradients_into(pid_b, bl_sets_b[k & 1], par_sets_b[k & 1], ll_b, grad_b)
    eng.sync()
    b_elapsed = time.perf_counter() - b0
    b_kernel_ms, b_launches = eng.kernel_elapsed()
    eng.kernel_timing(False)
    ablation = bool(os.environ.get("BENCH_ABLATION"))
    if not ablation and not (np.all(np.isfinite(ll_b)) and np.all(np.isfinite(grad_b))):
        raise SystemExit("non-finite results in the large batch")
    # (a tree's results depend on its batch only through the order of the pattern-tile sums: DESIGN.md section 3)
    same = min(T, Tb)
    ref_ll, ref_grad = np.zeros(same), np.zeros((same, N))
    eng.gradients_into(np.ascontiguousarray(pid_b[:same]), np.ascontiguousarray(bl_sets_b[(reps - 1) & 1][:same]),
                       np.ascontiguousarray(par_sets_b[(reps - 1) & 1][:same]), ref_ll, ref_grad)
    large = {"trees": Tb, "calls": reps, "ms_per_call": b_elapsed / reps * 1e3, "trees_per_s": Tb * reps / b_elapsed,
             "launches_per_step": b_launches / reps, "kernel_ms_per_call": b_kernel_ms / reps, "kernel": eng.kernel_name(),
             "max_dll_against_the_same_trees_in_a_call_of_their_own": float(np.max(np.abs(ll_b[:same] - ref_ll) / (1.0 + 2e-4 * np.abs(ref_ll)))),
             "max_dgrad_against_the_same_trees_in_a_call_of_their_own": float(np.max(np.abs(grad_b[:same] - ref_grad))),
             "note": f"{args.large_batch} x the trees of `value` in one blocking call (parameter rows change every call), mean of "
                     "the calls, in a process of its own; `value` stays the call BASELINE's metric is quoted on"}
    if not ablation and (large["max_dll_against_the_same_trees_in_a_call_of_their_own"] > 1e-11 or
                         large["max_dgrad_against_the_same_trees_in_a_call_of_their_own"] > 1e-7):
        raise SystemExit(f"the large batch's trees differ from the same trees in a call of their own: {large}")
    sys.stdout.flush()
    print(json.dumps(large), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--replicas", type=int, default=64,
                    help="x100 DS1 topologies per GPU and call (SURVEY 8d: replicated to fill the device; 6400 trees = 4 ms)")
    ap.add_argument("--workload", choices=["ds1", "config4", "codon", "gp"], default="ds1",
                    help="ds1 = BASELINE config 3 (the headline metric); config4 = synthetic 1000 taxa x 10 000 patterns, "
                         "1000 trees over the ranks, rescaling on; codon = config 5 (fluA as codons, GY94); gp = Path B, the "
                         "GPOperation schedules of a subsplit DAG (--gp-dag ds1 | seeded)")
    ap.add_argument("--gp-dag", choices=["ds1", "seeded"], default="ds1",
                    help="gp workload: the DAG of the ten DS1 golden trees (84 nodes, 119 edges) or of 20 seeded random "
                         "topologies over the same taxa (502 nodes, 970 edges)")
    ap.add_argument("--trees", type=int, default=0,
                    help="codon workload: trees per GPU (default 4096); config4: trees in all (default 1000, "
                         "125 per GPU at 8 GPUs; one GPU alone takes 125)")
    ap.add_argument("--distinct-models", type=int, default=1,
                    help="codon workload: how many different (kappa, omega) rows the trees of a batch carry (tree t has row "
                         "t %% K; the reference hands every tree its own row, src/fat_beagle.hpp:173-181).  `value` is measured "
                         "with this K; the line also carries K = 1, 64 and one row per tree (`distinct_models`)")
    ap.add_argument("--large-batch", type=int, default=16,
                    help="ds1 workload: beside `value`, one blocking call of this many times its trees (16 x 6400 = 102 400 "
                         "trees, 60 ms and more per call) reported as `large_batch`; 0 = skip")
    ap.add_argument("--large-batch-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-resident", action="store_true", help="the timed region only: skip the small-collection calls and the second timed region (resident batch)")
    ap.add_argument("--resident-only", action="store_true",
                    help="profiling: one blocking call, then only the second timed region (one traversal launch per pass); "
                         "`value` is then null")
    ap.add_argument("--no-parity-check", action="store_true",
                    help="skip the check of eight of the timed batch's trees against the CPU oracle behind the timed region")
    ap.add_argument("--engine-devices", type=str, default="",
                    help="ONE process, one engine over several device slots (the in-process counterpart of --gpus N under "
                         "torch.distributed.run): a number N = devices 0..N-1, or a list such as 0,0,0,0 (one GPU named four "
                         "times: the N-slot code path on a one-GPU box); every slot gets --replicas x 100 trees")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 HBM arena, 2 LDS, 5 / 6 walk_pipe_kernel with one / two waves per SIMD")
    ap.add_argument("--sum-ll-reduce", choices=("auto", "on", "off"), default="auto",
                    help="per step, all-reduce the summed log-likelihood over the ranks (RCCL); auto = when there "
                         "is more than one rank")
    args = ap.parse_args()

    if args.workload == "gp":
        return gp_workload(args)
    if args.large_batch_child:
        return large_batch_child(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world

    # The headline call at a device-filling size (calls of 50 ms and more): 16 x the trees of `value` in ONE
    # blocking call, so that a driver's clock around the loop measures more than a 0.08 s region.  In a process of its
    # own -- a batch sixteen times the largest the engine has run on a device must not be able to take this line with it
    # (the sample is reported with whatever went wrong instead) -- and FIRST, before this process has touched the GPU:
    # the child is over by the time anything here is timed, and no program is started from a process that holds a device.
    large = None
    if (args.workload == "ds1" and world == 1 and not args.no_resident and not args.resident_only and not args.engine_devices
            and args.large_batch > 0):
        import subprocess

        from bito_amd.dist import host_threads_for_rank as _threads

        # (BENCH_CHILD_CMD: scripts/bench_dry_run.py's interpreter line, whose child must see the same tiny workloads)
        head = json.loads(os.environ["BENCH_CHILD_CMD"]) if os.environ.get("BENCH_CHILD_CMD") else [sys.executable, os.path.abspath(__file__)]
        cmd = head + ["--large-batch-child", "--large-batch", str(args.large_batch),
                      "--replicas", str(args.replicas), "--steps", str(args.steps), "--kernel", str(args.kernel)]
        env = dict(os.environ, BENCH_CHILD_DEVICE=str(local_rank), BENCH_CHILD_HOST_THREADS=str(_threads()))
        try:
            done = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            lines = [ln for ln in done.stdout.strip().splitlines() if ln.startswith("{")]
            large = json.loads(lines[-1]) if done.returncode == 0 and lines else {
                "error": f"exit code {done.returncode}: " + (done.stderr.strip().splitlines() or ["no output"])[-1][:300]}
        except (subprocess.TimeoutExpired, OSError, ValueError) as err:
            large = {"error": repr(err)[:300]}

    import torch

    import bito_amd
    from bito_amd import workloads

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible")
    torch.cuda.set_device(local_rank)
    dist = None
    # BENCH_FORCE_DIST=1 builds a one-rank RCCL group on a single GPU, to exercise the reduce path there
    force_group = world == 1 and os.environ.get("BENCH_FORCE_DIST") == "1"
    if world > 1 or force_group:
        import torch.distributed as dist

        if force_group:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    reduce_ll = dist is not None and args.sum_ll_reduce != "off"
    if args.sum_ll_reduce == "on" and dist is None:
        raise SystemExit("--sum-ll-reduce on needs a process group (launch with torch.distributed.run)")

    slots = []
    if args.engine_devices:
        if world > 1:
            raise SystemExit("--engine-devices is the one-process mode: do not launch it under torch.distributed.run")
        slots = ([int(x) for x in args.engine_devices.split(",")] if "," in args.engine_devices
                 else list(range(int(args.engine_devices))))
    shards = max(len(slots), 1)  # device slots of this process's engine

    # every rank builds its own block of the same replicated workload
    codon = args.workload == "codon"
    config4 = args.workload == "config4"
    scaling = "weak"
    if codon:
        per_rank = (args.trees or 4096) * shards
        w = workloads.flua_codon(per_rank * world).shard(rank, world)
    elif config4:
        # BASELINE config 4: 1000 sampled trees sharded across the ranks (strong scaling: the collection is fixed);
        # one GPU alone takes one GPU's share of the 8-GPU run, 125 trees
        total = args.trees or (1000 if world > 1 else 125)
        lo, hi = total * rank // world, total * (rank + 1) // world
        w = workloads.synthetic_gtr_weibull4(1000, 10000, tree_count=hi - lo, first_tree=lo)
        scaling = "strong" if world > 1 else "weak"
    else:  # (a rank generates its own block of the 100 x replicas x world trees: same trees as the whole, sharded)
        per_rank = 100 * args.replicas * shards
        w = workloads.ds1_gtr_weibull4(args.replicas * world * shards, first_tree=rank * per_rank, tree_count=per_rank)
    T = w.tree_count
    n, P = w.patterns.shape
    C = 1 if codon else 4
    S = 61 if codon else 4
    N = 2 * n - 1

    # host threads of a blocking call (checking and packing inputs, copying results out): the engine's default is
    # min(8, usable CPUs); with several ranks on one node the CPUs are shared among them
    from bito_amd.dist import host_threads_for_rank

    host_threads = host_threads_for_rank()
    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights,
                          device_id=local_rank, host_threads=host_threads, devices=slots or None)
    if not codon:
        eng.set_kernel(args.kernel)

    # fresh host inputs per step: two sets of branch lengths take turns, and two sets of parameter rows -- the second
    # differs from the first in the last bit of one rate (GTR's AC rate, kappa), so that no step finds the models of the
    # step before still standing (round 4's model caches compare rows: worker.cpp UploadModelIndex, model_reuse) and
    # `value` includes every tree's parameter row -> eigendecomposition, as SURVEY.md 8d's span says.  The rate of a loop
    # of calls over FIXED parameter rows (the caches hit) is measured beside it: `model_cache_hit`.
    pid = np.ascontiguousarray(w.parent_ids, dtype=np.int32)
    if codon and args.distinct_models != 1:
        w.params = workloads.codon_rows(T, args.distinct_models if args.distinct_models > 0 else T)
    params = np.ascontiguousarray(w.params, dtype=np.float64)
    rate_column = 4  # GTR: frequencies [0, 4), rates [4, 10); GY94: frequencies [0, 4), kappa, omega
    param_sets = [params, workloads.other_bits(params, rate_column)]
    bl_sets = [np.ascontiguousarray(w.branch_lengths, dtype=np.float64),
               np.ascontiguousarray(w.branch_lengths * 1.03125, dtype=np.float64)]
    out_ll = np.zeros(T)
    out_grad = np.zeros((T, N))
    pending = []  # (work handle, tensor) of the reductions in flight
    counter = [0]
    fixed_rows = [False]

    def step():
        # the call the reference's Engine::Gradients is: host trees + parameter rows in, host results out
        bl = bl_sets[counter[0] & 1]
        par = param_sets[0 if fixed_rows[0] else counter[0] & 1]
        counter[0] += 1
        w.last_branch_lengths = bl
        w.last_params = par
        if w.want_gradient:
            eng.gradients_into(pid, bl, par, out_ll, out_grad, rescaling=w.rescaling)
        else:
            eng.log_likelihoods_into(pid, bl, par, out_ll, rescaling=w.rescaling)
        if reduce_ll:
            # trees are independent: the only exchange the path has is the summed log-likelihood of the whole
            # collection (the caller's objective), 8 bytes per step
            total = torch.tensor([float(out_ll.sum())], dtype=torch.float64).cuda(non_blocking=True)
            pending.append((dist.all_reduce(total, async_op=True), total))

    def fence():
        eng.sync()
        for work, _ in pending:
            work.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    timed_steps = 1 if args.resident_only else args.steps
    for _ in range(0 if args.resident_only else args.warmup):
        step()
    fence()
    pending.clear()
    eng.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(timed_steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = eng.kernel_elapsed()
    span_sum_ms = eng.kernel_span_sum()
    eng.kernel_timing(False)
    kernel = eng.kernel_name()

    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # sanity: results of the timed calls are finite
    # (BENCH_ABLATION=1: timing-only kernel variants of scripts/build_*_variants.sh, whose results mean nothing)
    if not os.environ.get("BENCH_ABLATION") and not (np.all(np.isfinite(out_ll)) and (not w.want_gradient or np.all(np.isfinite(out_grad)))):
        bad = np.flatnonzero(~np.isfinite(out_ll) | (~np.isfinite(out_grad).all(axis=1) if w.want_gradient else False))
        raise SystemExit(f"non-finite results in the timed batch: {len(bad)} of {len(out_ll)} trees, the first {bad[:8].tolist()}")
    # ... and they are the right numbers: eight trees of the last step against the CPU checker
    parity = None
    if rank == 0 and not args.no_parity_check and not os.environ.get("BENCH_ABLATION"):
        parity = check_against_oracle(w, out_ll, out_grad if w.want_gradient else None)
        bad = parity["max_dll_scaled"] > 1e-10 or parity.get("max_dgrad_relative", 0.0) > 1e-6
        if bad:
            raise SystemExit(f"the timed batch's results differ from the CPU checker's: {parity}")
    # the same loop over FIXED parameter rows: every call finds the models of the call before (round 4's caches hit)
    cache_hit = None
    if not args.resident_only and not args.no_resident:
        fixed_rows[0] = True
        for _ in range(2):
            step()
        fence()
        pending.clear()
        h0 = time.perf_counter()
        for _ in range(timed_steps):
            step()
        fence()
        h_elapsed = time.perf_counter() - h0
        # (the reductions of these steps stay in `pending`: the last one is checked against the gathered sum below)
        if dist is not None:
            tmax = torch.tensor([h_elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            h_elapsed = float(tmax.item())
        cache_hit = {"trees_per_s": world * T * timed_steps / h_elapsed, "ms_per_step": h_elapsed / timed_steps * 1e3,
                     "note": "the timed loop again with the SAME parameter rows on every step: the models of the step before are "
                             "found standing (worker.cpp: UploadModelIndex, DeviceBatch::model_reuse); `value` is the loop whose "
                             "rows change every step"}
        fixed_rows[0] = False
    # config 5 with K different (kappa, omega) rows among the trees: 1, 64, one per tree (rows change every step)
    distinct = None
    if codon and world == 1 and not args.no_resident and not args.resident_only:
        distinct = {}
        for K in sorted({1, 64, T}):
            rows = workloads.codon_rows(T, K)
            sets = [np.ascontiguousarray(rows), workloads.other_bits(rows, rate_column)]
            for k in range(2):
                eng.gradients_into(pid, bl_sets[k & 1], sets[k & 1], out_ll, out_grad, rescaling=w.rescaling)
            reps = max(3, min(args.steps, 8))
            d0 = time.perf_counter()
            for k in range(reps):
                eng.gradients_into(pid, bl_sets[k & 1], sets[k & 1], out_ll, out_grad, rescaling=w.rescaling)
            eng.sync()
            dt = time.perf_counter() - d0
            distinct[str(K)] = {"trees_per_s": T * reps / dt, "ms_per_step": dt / reps * 1e3}
        distinct["note"] = ("K different parameter rows (kappa ~ U(1.5, 4), omega ~ U(0.1, 0.9), seeded) among the batch's trees, "
                            "tree t carries row t % K; every row needs its own 64 x 64 eigensystem (gs_eigen_kernel)")
        # (leave the engine as the timed loop left it)
        eng.gradients_into(pid, w.last_branch_lengths, w.last_params, out_ll, out_grad, rescaling=w.rescaling)
    summed_ll = None
    if reduce_ll:
        # the last reduction must be the sum over every rank's block: check it against a gather of the
        # per-rank sums (outside the timed region)
        summed_ll = float(pending[-1][1].item())
        mine = torch.tensor([float(out_ll.sum())], dtype=torch.float64, device="cuda")
        parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
        expect = float(sum(p.item() for p in parts))
        if not abs(summed_ll - expect) <= 1e-9 * abs(expect):
            raise SystemExit(f"summed log-likelihood {summed_ll} differs from the gathered sum {expect}")

    # the same blocking call on small collections (BASELINE's literal "100 topologies", vip's particle loop): ms per call
    small_calls = small_calls_hit = None
    if args.workload == "ds1" and world == 1 and not args.no_resident and not args.resident_only and not slots:
        small_calls, small_calls_hit = {}, {}
        for count in (1, 100, 400, 1600):
            if count > T:
                continue
            ll_s, grad_s = np.zeros(count), np.zeros((count, N))
            pid_s = np.ascontiguousarray(pid[:count])
            par_s = [np.ascontiguousarray(ps[:count]) for ps in param_sets]
            bl_s = [np.ascontiguousarray(b[:count]) for b in bl_sets]
            for fixed, into in ((False, small_calls), (True, small_calls_hit)):  # parameter rows change every call / never
                for k in range(5):
                    eng.gradients_into(pid_s, bl_s[k & 1], par_s[0 if fixed else k & 1], ll_s, grad_s)
                reps = 40
                s0 = time.perf_counter()
                for k in range(reps):
                    eng.gradients_into(pid_s, bl_s[k & 1], par_s[0 if fixed else k & 1], ll_s, grad_s)
                into[str(count)] = (time.perf_counter() - s0) / reps * 1e3

    # second timed region: the same passes over a batch that stays in HBM (no host arrays cross PCIe)
    resident = None
    if not args.no_resident:
        from bito_amd import dist as bdist

        eng.upload(pid, bl_sets[0], params)
        # (more than one rank: every pass ends with the asynchronous all-reduce of the summed log-likelihood, taken
        # from where the engine left the values -- bito_amd/dist.py)
        reducer = bdist.ResidentSumReducer(eng) if reduce_ll else None

        def resident_pass():
            if reducer is not None:
                reducer.run(w.want_gradient, w.rescaling)
            else:
                eng.run(w.want_gradient, w.rescaling)

        def resident_fence():
            if reducer is not None:
                reducer.finish()
            eng.sync()
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize()

        for _ in range(args.warmup):
            resident_pass()
        resident_fence()
        eng.kernel_timing(True)
        r0 = time.perf_counter()
        for _ in range(args.steps):
            resident_pass()
        resident_fence()
        r_elapsed = time.perf_counter() - r0
        r_kernel_ms, r_launches = eng.kernel_elapsed()
        eng.kernel_timing(False)
        r_kernel = eng.kernel_name()
        if dist is not None:
            tmax = torch.tensor([r_elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            r_elapsed = float(tmax.item())
        resident = {"trees_per_s": world * T * args.steps / r_elapsed, "ms_per_step": r_elapsed / args.steps * 1e3,
                    "note": "bito_amd_engine_run back to back over a batch resident in HBM" +
                            ("; per pass one asynchronous RCCL all-reduce of the summed log-likelihood" if reduce_ll else ""),
                    "roofline": roofline_object(r_kernel, n, P, C, S, w.want_gradient, T * args.steps / max(r_launches, 1),
                                                r_kernel_ms * 1e-3 / max(r_launches, 1), args.workload)}

    out = None
    if rank == 0:
        total_trees = world * T
        value = None if args.resident_only else total_trees * args.steps / elapsed
        trees_per_launch = T * timed_steps / max(launches, 1)
        avg_kernel_s = kernel_ms * 1e-3 / max(launches, 1)
        if codon:
            metric = "tree log-likelihoods+gradients/sec (fluA codon GY94, 61 states)"
            workload = (f"BASELINE config 5: fluA.fa as codons (69 taxa, 329 codon columns = {P} patterns, 61 states), "
                        f"fluA.tree topology x {T} trees per GPU with seeded branch lengths, GY94 (kappa, omega, F1x4), "
                        "log-likelihood + branch-length gradient")
        elif config4:
            metric = "tree log-likelihoods+gradients/sec (synthetic 1000 taxa x 10000 patterns, GTR+Gamma4, rescaling)"
            workload = (f"BASELINE config 4: synthetic alignment, 1000 taxa x 10 000 distinct patterns (JC69 down a seeded "
                        f"tree), {total_trees} seeded random topologies with Exp(0.1) branch lengths sharded over "
                        f"{world} GPU(s) ({T} per GPU), GTR+weibull4, rescaling on, log-likelihood + branch-length gradient")
        else:
            metric = "tree log-likelihoods+gradients/sec (DS1 GTR+Gamma4)"
            workload = ("BASELINE config 3: DS1.fasta (27 taxa, 934 patterns) x 100 topologies x "
                        f"{args.replicas} replicas per GPU, GTR+weibull4 (4 categories), seeded Exp(0.1) branch "
                        "lengths, log-likelihood + branch-length gradient")
        out = {
            "metric": metric,
            "value": value,
            "unit": "trees/s",
            "n_gpus": len(set(slots)) if slots else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / timed_steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "span": ("one blocking Engine::Gradients-shaped call per step: host parent ids, fresh host branch "
                         "lengths and parameter rows in, host log-likelihoods and gradients out (PCIe inclusive)"),
                "trees_per_gpu": T,
                "trees_total": total_trees,
                "kernel": kernel,
                **({"kernel_form": eng.kernel_form()} if eng.kernel_form() else {}),
                **({"engine_device_slots": slots} if slots else {}),
                "host_threads_per_rank": host_threads,
                "multi_gpu": ("trees sharded by rank; per step one asynchronous RCCL all-reduce of the summed "
                              "log-likelihood (8 bytes)" if reduce_ll else
                              "trees sharded by rank, no data-path collective (barrier + max-over-ranks timing only)")
                if dist is not None else "single GPU, no collective",
                **({"summed_log_likelihood": summed_ll} if reduce_ll else {}),
            },
            "roofline": roofline_object(kernel, n, P, C, S, w.want_gradient, trees_per_launch, avg_kernel_s,
                                        args.workload),
        }
        out["roofline"]["launches_per_step"] = launches / timed_steps
        if parity is not None:
            out["parity"] = parity
        # `avg_kernel_ms` is the time the kernel was running per launch: the union of the launches' spans (the chunks
        # of a call overlap: the second chunk's workgroups move in while the first chunk's leave).  A profiler's
        # per-launch average is the spans' sum / launches:
        out["roofline"]["avg_launch_span_ms"] = span_sum_ms / max(launches, 1)
        if small_calls:
            out["blocking_call_ms"] = {"trees_per_call": small_calls, "trees_per_call_model_cache_hit": small_calls_hit,
                                       "note": "the same gradients call on 1 / 100 / 400 / 1600 of the trees, mean of 40 calls; "
                                               "parameter rows that change with every call (trees_per_call) and rows that never "
                                               "change (the last call's model is found standing)"}
        if large is not None:
            out["large_batch"] = large
        if cache_hit is not None:
            out["model_cache_hit"] = cache_hit
        if distinct is not None:
            out["distinct_models"] = distinct
        if codon:
            out["config"]["distinct_models_in_value"] = int(len(np.unique(params, axis=0)))
        switches = {k: v for k, v in sorted(os.environ.items()) if k.startswith("BITO_AMD_") and k != "BITO_AMD_LIB"}
        if switches:  # (an A/B line says which of the library's switches it ran with: scripts/README.md lists them)
            out["config"]["switches"] = switches
        out["config"]["parameter_rows"] = "two sets take turns step by step (they differ in the last bit of one rate): no step finds the models of the step before"
        if resident is not None:
            out["resident"] = resident
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # (the one JSON line goes out last and flushed: RCCL writes its version banner straight to file descriptor 1)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
