"""Path B on the GPU box: the GPDAG schedules (PopulatePLVs + ComputeLikelihoods + MarginalLikelihood) of
the subsplit DAG of the ten DS1 golden trees (27 taxa, 934 patterns), GPU executor against the CPU oracle.
usage: python scripts/gpu_gp.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bito_amd import gp, treeio
from bito_amd.gp_dag import SubsplitDAG
from bito_amd.site_pattern import SitePattern
from oracle import gp as ogp
from test_tp import _renumber

D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
tc = treeio.read_nexus_file(os.path.join(D, "DS1.subsampled_10.t"))
sp = SitePattern(treeio.read_fasta(os.path.join(D, "DS1.fasta")), tc.taxon_names)
pids = []
for t in tc.trees:  # root the unrooted golden trees on their first root child
    p = np.asarray(t.parent_ids).copy()
    M = len(p) + 1
    kids = [c for c in range(M - 1) if p[c] == M - 1]
    q = np.append(p, M)
    q[kids[0]] = M
    pids.append(_renumber(q))
dag = SubsplitDAG(len(tc.taxon_names), pids)
bl = np.random.default_rng(1).uniform(0.01, 0.2, dag.gpcsp_count)
streams = [dag.populate_plvs(), dag.compute_likelihoods(), dag.marginal_likelihood()]
nops = sum(len(s.arrays()[0]) for s in streams)
print(f"DAG: {dag.node_count} nodes, {dag.gpcsp_count} edges, {int(dag.topology_count)} trees spanned; {nops} operations per pass")


def run(eng, reps):
    eng.set_branch_lengths(bl)
    eng.set_sbn_parameters(dag.uniform_on_topological_support_prior())
    t0 = time.perf_counter()
    for _ in range(reps):
        for s in streams:
            eng.process_operations(s)
    value = eng.get_log_marginal_likelihood()
    return (time.perf_counter() - t0) / reps, value


gpu = gp.GPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)
run(gpu, 2)
tg, vg = run(gpu, 20)
cpu = ogp.OracleGPEngine(sp.patterns, sp.weights, dag.node_count, dag.gpcsp_count)
tcpu, vc = run(cpu, 2)
print(f"GPU executor: {tg*1e3:.3f} ms per pass ({nops/tg/1e6:.2f} M ops/s); CPU oracle (1 thread): {tcpu*1e3:.1f} ms per pass; "
      f"log marginal {vg:.6f} vs {vc:.6f} (diff {abs(vg-vc):.2e})")

# -- NNI proposals on spare slots (bito_amd/nni.py): all proposals in one launch vs one call per proposal --
from bito_amd.nni import NNIEvalEngineViaGP

gpu.set_sbn_parameters(np.ones(dag.gpcsp_count))
ev = NNIEvalEngineViaGP(dag, gpu)
ev.prep()
t0 = time.perf_counter()
scores = ev.score_adjacent_nnis()
t_first = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(10):
    gpu.process_operation_batches([p.stream for p in ev.proposals])
t_batch = (time.perf_counter() - t0) / 10
t0 = time.perf_counter()
for p in ev.proposals:
    gpu.process_operations(p.stream)
t_seq = time.perf_counter() - t0
nops = sum(len(p.stream.ops) for p in ev.proposals)
cpu.set_sbn_parameters(np.ones(dag.gpcsp_count))
evc = NNIEvalEngineViaGP(dag, cpu)
evc.prep()
t0 = time.perf_counter()
cscores = evc.score_adjacent_nnis()
t_cpu = time.perf_counter() - t0
worst = max(abs(scores[k] - cscores[k]) for k in scores)
print(f"NNI proposals: {len(scores)} adjacent NNIs, {nops} operations; batched launch {t_batch*1e3:.3f} ms "
      f"({len(scores)/t_batch:.0f} NNIs/s; first call incl. schedule building {t_first*1e3:.1f} ms), "
      f"one call per NNI {t_seq*1e3:.2f} ms, CPU (1 thread) {t_cpu*1e3:.1f} ms; max |GPU - CPU| = {worst:.2e}")

# optimize_new_edges: one workgroup per proposal interprets PLV ops and optimiser ops alike
for method, name in ((4, "Newton"), (0, "Brent")):
    gpu.set_optimization_method(method)
    cpu.set_optimization_method(method)
    evo = NNIEvalEngineViaGP(dag, gpu, optimize_new_edges=True, optimization_max_iteration=3)
    t0 = time.perf_counter()
    so = evo.score_adjacent_nnis()
    t_g = time.perf_counter() - t0
    evoc = NNIEvalEngineViaGP(dag, cpu, optimize_new_edges=True, optimization_max_iteration=3)
    t0 = time.perf_counter()
    sc = evoc.score_adjacent_nnis()
    t_c = time.perf_counter() - t0
    nopt = sum(1 for p in evo.proposals for op in p.stream.ops if op[0] == 5)
    src, dst = [x for p in evo.proposals for x in p.copy_src], [x for p in evo.proposals for x in p.copy_dst]
    gpu.copy_gpcsp_data(src, dst)
    t0 = time.perf_counter()
    gpu.process_operation_batches([p.stream for p in evo.proposals])
    t_call = time.perf_counter() - t0
    print(f"optimised proposals ({name}, 3 rounds): {len(so)} NNIs, {nopt} edge optimisations; GPU {t_g*1e3:.1f} ms "
          f"(the batched call alone, stream merge in Python included: {t_call*1e3:.1f} ms), "
          f"CPU (1 thread) {t_c*1e3:.0f} ms; max |GPU - CPU| = {max(abs(so[k]-sc[k]) for k in so):.2e}; "
          f"mean gain over unoptimised {np.mean([so[k]-scores[k] for k in so]):.3f}")
