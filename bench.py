#!/usr/bin/env python3
"""bench.py -- tree log-likelihoods+gradients/sec on BASELINE.json's headline config.

Workload (config 3): DS1.fasta (27 taxa, 934 site patterns), the 100 topologies of
DS1.100_topologies.nwk replicated R times per GPU with per-tree seeded branch lengths,
GTR + weibull+4 (4 rate categories), FP64, log-likelihood + branch-length gradient.
A "step" is one pass of the hot path over the resident batch: per-tree model set-up +
eigendecomposition, transition matrices, post-order partials, pre-order partials + edge
derivatives, per-tree reductions.  Inputs (parent-id vectors, branch lengths, parameter
rows, compressed alignment) are resident in HBM before the timed region.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); every rank owns
R x 100 trees (weak scaling); each step ends with an all-gather of the per-tree results
and an all-reduce of the summed log-likelihood, the only exchange the path has.

Prints ONE JSON line on rank 0.
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


FP64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X dense FP64 matrix peak (MI355X_MICROARCH.md); codon workload


def algorithmic_bytes_per_tree(n: int, P: int, C: int, gradient: bool, S: int = 4) -> float:
    """B_plv = C*P*S*8; LL: (3(n-1)+1) B_plv; LL+grad: (13n-12) B_plv."""
    b_plv = C * P * S * 8
    return ((13 * n - 12) if gradient else (3 * (n - 1) + 1)) * b_plv


def algorithmic_flops_per_tree(n: int, P: int, C: int, S: int) -> float:
    """SURVEY.md section 8d, LL + gradient: C P [(3n-3)(4S^2-S) + (2n-2)(2S^2+3S-1)]."""
    return C * P * ((3 * n - 3) * (4 * S * S - S) + (2 * n - 2) * (2 * S * S + 3 * S - 1))


def cpu_baseline(w, seconds: float):
    """The CPU oracle driven like the reference Engine (one instance per thread, dynamic
    queue over trees) on a bounded sample of the same workload."""
    threads = os.cpu_count() or 1
    if w.substitution == "GY94":
        from oracle import gs

        eng = gs.GsOracleEngine(w.substitution, w.site, w.patterns, w.weights, threads)
        source = "oracle/gs_oracle.c"
    else:
        from oracle import oracle

        eng = oracle.OracleEngine(w.substitution, w.site, w.clock, w.patterns, w.weights, threads)
        source = "oracle/bito_oracle.c"
    run = eng.gradients if w.want_gradient else eng.log_likelihoods

    def sample(count):
        reps = -(-count // w.tree_count)
        return (np.tile(w.parent_ids, (reps, 1))[:count], np.tile(w.branch_lengths, (reps, 1))[:count],
                np.tile(w.params, (reps, 1))[:count])

    probe = max(4 * threads, 64)
    pid, bl, par = sample(probe)
    run(pid, bl, par, rescaling=w.rescaling)  # warm-up: first touch of every thread's buffers
    # Timed in chunks until the budget is used: a short probe overestimates the sustained rate of a
    # many-core host severalfold, so the sample size is not extrapolated from it.
    count, dt, chunk = 0, 0.0, probe
    while dt < seconds:
        pid, bl, par = sample(chunk)
        t0 = time.perf_counter()
        run(pid, bl, par, rescaling=w.rescaling)
        took = time.perf_counter() - t0
        count += chunk
        dt += took
        chunk = int(min(max(probe, chunk / max(took, 1e-3) * seconds / 4), 64 * probe))  # about a quarter of the budget
    return {"value": count / dt, "unit": "trees/s", "cores": threads, "kind": "port",
            "sample": f"{count} trees of the same workload, {dt:.1f} s, {source} with {threads} threads "
                      "(FP64 restatement of the BEAGLE CPU path; the reference binary cannot be built here)"}


def arithmetic_view(n, P, C, S, trees_per_launch, avg_kernel_s):
    """Algorithmic FP64 flops per launch (every child message a full matrix-vector
    product, tips included) against the guide's dense FP64 matrix peak; the rate v_mfma_f64_4x4x4_4b sustains
    from one wave per SIMD on the box (68 TFLOP/s, profiles/r1_microbench.json) is given beside it."""
    flops = algorithmic_flops_per_tree(n, P, C, S) * trees_per_launch
    achieved = flops / avg_kernel_s / 1e12
    return {"bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP64_MATRIX_PEAK_TFLOPS, "sustained_peak": FP64_MFMA_PEAK_TFLOPS,
            "frac_of_sustained": achieved / FP64_MFMA_PEAK_TFLOPS,
            "algorithmic_flops_per_tree": algorithmic_flops_per_tree(n, P, C, S)}


def measured_traffic(kernel: str, trees_per_launch: int):
    """HBM bytes per launch of the traversal kernel from the committed rocprofv3 PMC pass
    (profiles/*_traffic.json), if one matches this launch shape."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            for row in json.load(fh):
                if row["kernel"] == kernel and row["trees_per_launch"] == trees_per_launch:
                    return row["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--replicas", type=int, default=64,
                    help="x100 DS1 topologies per GPU and pass (SURVEY 8d: replicated to fill the device; 6400 trees = 4 ms)")
    ap.add_argument("--workload", choices=["ds1", "codon"], default="ds1",
                    help="ds1 = BASELINE config 3 (the headline metric); codon = config 5 (fluA as codons, GY94)")
    ap.add_argument("--trees", type=int, default=4096, help="codon workload: trees per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 HBM arena, 2 LDS")
    ap.add_argument("--sum-ll-reduce", choices=("auto", "on", "off"), default="auto",
                    help="per step, all-reduce the summed log-likelihood over the ranks (RCCL); auto = when there "
                         "is more than one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world

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

    # every rank builds the same replicated workload and takes its own block of trees
    codon = args.workload == "codon"
    if codon:
        w = workloads.flua_codon(args.trees * world).shard(rank, world)
    else:  # (a rank generates its own block of the 100 x replicas x world trees: same trees as the whole, sharded)
        per_rank = 100 * args.replicas
        w = workloads.ds1_gtr_weibull4(args.replicas * world, first_tree=rank * per_rank, tree_count=per_rank)
    T = w.tree_count
    n, P = w.patterns.shape
    C = 1 if codon else 4
    S = 61 if codon else 4

    eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights,
                          device_id=local_rank)
    if not codon:
        eng.set_kernel(args.kernel)
    eng.upload(w.parent_ids, w.branch_lengths, w.params)

    # Summed log-likelihood over the ranks, one asynchronous RCCL all-reduce per step.  Nothing waits on the
    # host and nothing is added to the engine's stream: torch's stream waits for the pass through the event the
    # engine records behind it anyway, sums the per-tree values where the engine left them (a ring of four
    # buffers, so the next passes do not touch them), and hands the scalar to RCCL; the next pass (and its
    # set-up) is submitted meanwhile.  The engine's stream waits for the sum that last read a ring slot before
    # the pass that rewrites it, four passes later.
    kRing = 4
    engine_stream = None
    if reduce_ll:
        try:
            engine_stream = torch.cuda.ExternalStream(eng.stream_handle())
        except Exception as exc:  # noqa: BLE001 -- then hand results over with a host wait per step instead
            print(f"bench: no external-stream wrapper ({exc!r}); the reduction waits on the host each step", file=sys.stderr)
    ll_host_ring = torch.zeros(T, dtype=torch.float64, device="cuda") if reduce_ll and engine_stream is None else None
    sum_done = [None] * kRing
    pending = []  # (work handle, tensor) of the reductions in flight
    step_index = [0]

    class _DeviceVector:
        """zero-copy view of `count` doubles at a device address, for torch.as_tensor"""

        def __init__(self, address, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (address, False), "version": 2}

    def step():
        # trees are independent: each rank evaluates its own block; the only exchange the path has is the
        # summed log-likelihood of the whole collection (the caller's objective), 8 bytes per step
        if reduce_ll:
            slot = step_index[0] % kRing
            step_index[0] += 1
            if engine_stream is not None and sum_done[slot] is not None:
                engine_stream.wait_event(sum_done[slot])
        eng.run(w.want_gradient, w.rescaling)
        if reduce_ll:
            here = torch.cuda.current_stream()
            if engine_stream is not None:
                ll_address, _ = eng.results_async(here.cuda_stream)
                values = torch.as_tensor(_DeviceVector(ll_address, T), device="cuda")
            else:
                torch.cuda.synchronize()  # earlier sums have read the buffer
                eng.download_to(ll_host_ring.data_ptr(), None)  # waits for the pass
                values = ll_host_ring
            total = values.sum().reshape(1)
            sum_done[slot] = torch.cuda.Event()
            sum_done[slot].record(here)
            pending.append((dist.all_reduce(total, async_op=True), total))

    def fence():
        eng.sync()
        for work, _ in pending:
            work.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    pending.clear()
    eng.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = eng.kernel_elapsed()
    eng.kernel_timing(False)

    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # sanity: results of the timed batch are finite
    ll_host, grad_host = eng.download(True)
    if not (np.all(np.isfinite(ll_host)) and np.all(np.isfinite(grad_host))):
        raise SystemExit("non-finite results in the timed batch")
    summed_ll = None
    if reduce_ll:
        # the last reduction must be the sum over every rank's block: check it against a gather of the
        # per-rank sums (outside the timed region)
        summed_ll = float(pending[-1][1].item())
        mine = torch.tensor([float(ll_host.sum())], dtype=torch.float64, device="cuda")
        parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
        expect = float(sum(p.item() for p in parts))
        if not abs(summed_ll - expect) <= 1e-9 * abs(expect):
            raise SystemExit(f"summed log-likelihood {summed_ll} differs from the gathered sum {expect}")

    if rank == 0:
        total_trees = world * T
        value = total_trees * args.steps / elapsed
        trees_per_launch = T * args.steps / max(launches, 1)
        avg_kernel_s = kernel_ms * 1e-3 / max(launches, 1)
        alg_bytes = algorithmic_bytes_per_tree(n, P, C, w.want_gradient, S) * trees_per_launch
        achieved = alg_bytes / avg_kernel_s / 1e9
        kernel = eng.kernel_name()
        if codon:
            flops = algorithmic_flops_per_tree(n, P, C, S) * trees_per_launch
            arithmetic = {"bound": "mfma", "achieved": flops / avg_kernel_s / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": flops / avg_kernel_s / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                          "note": "algorithmic flops of SURVEY 8d (tip children counted as full products); "
                                  "v_mfma_f64_16x16x4 sustains 47.6 TFLOP/s on this part (profiles/r1_microbench.json)"}
            workload = (f"BASELINE config 5: fluA.fa as codons (69 taxa, 329 codon columns = {P} patterns, 61 states), "
                        f"fluA.tree topology x {T} trees per GPU with seeded branch lengths, GY94 (kappa, omega, F1x4), "
                        "log-likelihood + branch-length gradient")
        else:
            arithmetic = arithmetic_view(n, P, C, S, trees_per_launch, avg_kernel_s)
            workload = ("BASELINE config 3: DS1.fasta (27 taxa, 934 patterns) x 100 topologies x "
                        f"{args.replicas} replicas per GPU, GTR+weibull4 (4 categories), seeded Exp(0.1) branch "
                        "lengths, log-likelihood + branch-length gradient")
        traffic = measured_traffic(kernel, int(trees_per_launch))
        out = {
            "metric": ("tree log-likelihoods+gradients/sec (fluA codon GY94, 61 states)" if codon
                       else "tree log-likelihoods+gradients/sec (DS1 GTR+Gamma4)"),
            "value": value,
            "unit": "trees/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "trees_per_gpu": T,
                "trees_total": total_trees,
                "kernel": kernel,
                "multi_gpu": ("trees sharded by rank; per step one asynchronous RCCL all-reduce of the summed "
                              "log-likelihood (8 bytes)" if reduce_ll else
                              "trees sharded by rank, no data-path collective (barrier + max-over-ranks timing only)")
                if dist is not None else "single GPU, no collective",
                **({"summed_log_likelihood": summed_ll} if reduce_ll else {}),
            },
            "roofline": None,
        }
        hbm_view = {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # the PMC-measured bytes over the same launch time: what the kernel really asks of HBM
            "actual_hbm_GBps": (traffic / avg_kernel_s / 1e9) if traffic else None,
            "algorithmic_bytes_per_tree": algorithmic_bytes_per_tree(n, P, C, w.want_gradient, S),
        }
        common = {"traffic": traffic, "kernel": kernel, "avg_kernel_ms": avg_kernel_s * 1e3,
                  "trees_per_launch": trees_per_launch}
        if kernel in ("walk_pipe_kernel", "walk_lds_kernel", "walk_tree_kernel"):
            # These kernels keep every partial in LDS: HBM sees 0.3 % of the op-by-op byte model (`traffic`), so
            # the resource that bounds them is the FP64 matrix / vector pipe.  The byte view is kept beside it.
            out["roofline"] = {**arithmetic, **common, "hbm_view": hbm_view}
        else:
            out["roofline"] = {**hbm_view, **common, "arithmetic": arithmetic}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
