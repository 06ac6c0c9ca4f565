"""Data-race check of the kernels on the CPU: the emulated build of tests/hip_emu under ThreadSanitizer, every GPU thread a
TSan fiber, __syncthreads() and the wave operations the only happens-before edges inside a workgroup (hip/hip_runtime.h).
An LDS or global access that two threads of a workgroup make without a barrier between them -- which neither the serial
fibers of the normal emulated build nor a passing device run would show -- is reported with both source lines.  Drivers:
the two compiled C++ clients of the C ABI (tests/cabi_client.cpp, tests/gp_binding_client.cpp), linked with the sanitized
objects of tests/hip_emu/_build/race (ThreadSanitizer) and _build/memcheck (AddressSanitizer: out-of-bounds and
use-after-free accesses of kernels and host code -- the sanitizers this project can run, on the CPU build only), on case
files written here:
  gp       ds1-reduced-5's multi-tree DAG and the DS1 ten-tree DAG (bench.py's Path B workload): populate, likelihoods, one
           scheduled optimisation sweep (concurrent workgroups of gp_optimize_kernel, gp_levels_kernel), marginal
  codon    two 9-taxon trees under GY94 (61 states): gs_eigen_kernel (round 5: two barriers per Jacobi round instead of
           four), gs_matrices_kernel, gs_schedule / gs_walk kernels
  hbm      weibull+6 (walk_hbm_kernel) and 70 taxa with weibull+4 (walk_hbm_cat_kernel) on small alignments
  pipe     DS1-shaped trees through walk_pipe_kernel (the C++ around the interpreted assembly; the interpreter's own LDS
           accesses are made on a wave's first lane, its s_barrier is a happens-before edge like __syncthreads())
  slots    150 trees in one blocking call over three device slots: the engine's own threads (an issuing thread per slot, the
           shared helper threads that pack the chunks' ranges: round 5) -- real OS threads, real races if there were any
Reports whose two stacks lie inside the stand-in runtime itself are listed separately; a report that names kernel code
fails the run.  Within a wave the hardware's lockstep orders accesses TSan cannot know about: a report naming two lanes of
one wave reads "relies on lockstep" (none in the kernels; pipe_prepare's hand-over is sealed by prepare.py).
usage: python scripts/emu_race_check.py [gp codon hbm pipe slots]   (CPU only; needs /root/reference for the gp client)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hip_emu")
RACE = os.path.join(EMU, "_build", "race")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build():
    subprocess.check_call(["make", "-s", "-C", EMU, "race", "memcheck"])
    for kind, flag in (("race", "-fsanitize=thread"), ("memcheck", "-fsanitize=address")):
        objects = sorted(glob.glob(os.path.join(EMU, "_build", kind, "*.o")))
        common = [CLANG, "-std=c++17", "-O1", "-g", flag, "-pthread"]
        subprocess.check_call(common + [os.path.join(ROOT, "tests", "cabi_client.cpp")] + objects + ["-o", os.path.join(EMU, "_build", kind, "cabi_client")])
        if os.path.isdir("/root/reference/src"):
            subprocess.check_call(common + ["-I/root/reference/src", os.path.join(ROOT, "tests", "gp_binding_client.cpp")] + objects +
                                  ["-o", os.path.join(EMU, "_build", kind, "gp_binding_client")])


def engine_case(path, w):
    with open(path, "w") as fh:
        fh.write(f"{w.substitution} {w.site} {w.clock}\n{w.patterns.shape[0]} {w.patterns.shape[1]}\n")
        fh.write(" ".join(str(int(x)) for x in w.patterns.reshape(-1)) + "\n")
        fh.write(" ".join(repr(float(x)) for x in w.weights) + "\n")
        fh.write(f"0 {w.tree_count} {w.parent_ids.shape[1] + 1}\n")
        fh.write(" ".join(str(int(x)) for x in w.parent_ids.reshape(-1)) + "\n")
        fh.write(" ".join(repr(float(x)) for x in w.branch_lengths.reshape(-1)) + "\n")
        fh.write(f"{w.params.shape[1]}\n" + " ".join(repr(float(x)) for x in w.params.reshape(-1)) + "\n")


def cases(which, tmp):
    import numpy as np

    from bito_amd import workloads

    out = []
    if which == "gp":
        import test_gp_binding_client as t

        sp, dag, bl, q, streams = t._instance(os.path.join(ROOT, "tests", "golden", "data"))
        t._case(os.path.join(tmp, "gp.txt"), sp, dag, bl, q, streams)
        out.append(("gp_binding_client", os.path.join(tmp, "gp.txt")))
        # ... and Path B's bench workload: the DS1 ten-tree DAG (84 nodes, 119 edges, 934 patterns), 118 optimisations in 56
        # launches of concurrent workgroups, the 1082-operation passes through gp_levels_kernel
        dag, sp = workloads.ds1_subsplit_dag(10)
        bl = np.random.default_rng(1).uniform(0.01, 0.2, dag.gpcsp_count)
        streams = [dag.populate_plvs(), dag.compute_likelihoods(), dag.branch_length_optimization(), dag.populate_plvs(),
                   dag.compute_likelihoods(), dag.marginal_likelihood()]
        t._case(os.path.join(tmp, "gp_ds1.txt"), sp, dag, bl, dag.uniform_on_topological_support_prior(), streams)
        out.append(("gp_binding_client", os.path.join(tmp, "gp_ds1.txt")))
    if which == "codon":
        w = workloads.flua_codon(2)
        keep = 9  # (the first nine taxa of fluA as a tree of their own: small enough for TSan, the same kernels)
        # (64 patterns = four tiles: every wave of gs_walk_kernel's workgroup has a tile of its own.  A wave past the last
        # tile repeats the last tile's work and stores nothing -- its loads of the vectors the tile's own wave stores are a
        # race by the book and harmless by construction, and would drown a real one)
        s = workloads.synthetic_gtr_weibull4(keep, 64, tree_count=2)
        rng = np.random.default_rng(5)
        s.substitution, s.site = "GY94", "constant"
        s.patterns = rng.integers(0, 61, (keep, 64)).astype(np.int32)
        s.params = np.ascontiguousarray(w.params[:2])
        s.branch_lengths = s.branch_lengths * 0.05
        engine_case(os.path.join(tmp, "codon.txt"), s)
        out.append(("cabi_client", os.path.join(tmp, "codon.txt")))
    if which == "hbm":
        a = workloads.synthetic_gtr_weibull4(9, 70, tree_count=2)
        a.site = "weibull+6"
        engine_case(os.path.join(tmp, "hbm6.txt"), a)
        b = workloads.synthetic_gtr_weibull4(70, 24, tree_count=1)
        engine_case(os.path.join(tmp, "hbm70.txt"), b)
        out += [("cabi_client", os.path.join(tmp, "hbm6.txt")), ("cabi_client", os.path.join(tmp, "hbm70.txt"))]
    if which == "pipe":
        a = workloads.synthetic_gtr_weibull4(12, 70, tree_count=3)
        engine_case(os.path.join(tmp, "pipe.txt"), a)
        out.append(("cabi_client", os.path.join(tmp, "pipe.txt")))
    if which == "slots":
        a = workloads.synthetic_gtr_weibull4(6, 24, tree_count=150)
        engine_case(os.path.join(tmp, "slots.txt"), a)
        out.append(("cabi_client", os.path.join(tmp, "slots.txt")))
    return out


# `slots`: the engine's own threads -- one blocking call of 150 trees over THREE device slots (an issuing thread each), cut
# into chunks, the chunks' ranges packed by the shared helper threads (round 5): real OS threads, real races if any
EXTRA = {"slots.txt": (["3"], {"BITO_AMD_CHUNK_FIRST": "8", "BITO_AMD_CHUNK_GROWTH": "2", "BITO_AMD_CHUNK_CAP": "40", "BITO_AMD_HOST_MIN_TREES": "8"})}


RUNTIME = ("hip_emu::RunAsm", "hip_emu::Execute", "hip_emu::WaveMachine", "hip_emu::ProgramOf", "hip_emu::Block", "hip_emu::ParseProgram",
           "hip_emu::Trampoline", "hip_emu::RunBlock", "hip_emu::Launch")


def run(program, case, tmp, tag):
    """one case under ThreadSanitizer (this image's TSan runtime dies at start-up now and then -- a SEGV inside the runtime
    before the first kernel, dependent on the address-space layout: such a run is repeated) and under AddressSanitizer"""
    log = os.path.join(tmp, "tsan_" + tag)
    args, more = EXTRA.get(os.path.basename(case), ([], {}))
    env = dict(os.environ, TSAN_OPTIONS=f"report_signal_unsafe=0 exitcode=0 history_size=4 log_path={log}", **more)
    for attempt in range(12):
        for f in glob.glob(log + ".*"):
            os.remove(f)
        done = subprocess.run([os.path.join(EMU, "_build", "race", program), case] + args, capture_output=True, text=True, env=env, timeout=3000)
        if done.stdout.strip():
            break
    else:
        attempt = -1  # (not checked for races: said so in the summary line; the memory check below still runs)
    text = "".join(open(f).read() for f in glob.glob(log + ".*")) if attempt >= 0 else ""
    reports = [r for r in text.split("==================") if "WARNING: ThreadSanitizer" in r]
    kernel, runtime = [], []
    for r in reports:
        # the innermost frame of each of the report's two stacks that is not the C++ library's
        tops = []
        for stack in re.split(r"\n\s*\n", r):
            frames = re.findall(r"#\d+ (.+?) /", stack)
            frames = [f for f in frames if not f.startswith(("std::", "__tsan", "operator", "malloc", "void std::", "__gnu_cxx::", "void __gnu_cxx::",
                                                             "memcmp", "memcpy", "memset", "free"))]
            frames = [f for f in frames if "std::" not in f.split("(")[0] and "__gnu_cxx::" not in f.split("(")[0]]
            if frames and re.match(r"\s*(Read|Write|Previous read|Previous write|Atomic|Previous atomic)\b.* of size", stack):
                tops.append(frames[0])
        # (everything in namespace hip_emu is the stand-in runtime and the interpreter; the device builtins' stand-ins are
        # hip_emu_* free functions and count as kernel code)
        if not tops:  # (both stacks inside the C++ library: the summary line names the function they were called from)
            tops = re.findall(r"SUMMARY: ThreadSanitizer: data race \S+ in (.+)", r)
        inside_runtime = tops and all("hip_emu::" in t.split("(")[0] or t.startswith(RUNTIME) for t in tops)
        (runtime if inside_runtime else kernel).append(r)
    mem_env = dict(os.environ, ASAN_OPTIONS="detect_stack_use_after_return=0 detect_leaks=0 exitcode=23", **more)
    mem = subprocess.run([os.path.join(EMU, "_build", "memcheck", program), case] + args, capture_output=True, text=True, env=mem_env, timeout=3000)
    memory_errors = mem.stderr.count("ERROR: AddressSanitizer")
    if mem.returncode not in (0, 23) and not memory_errors:
        raise SystemExit(f"{program} {case} (AddressSanitizer build): exit code {mem.returncode}\n{mem.stderr[-2000:]}")
    if memory_errors:
        kernel.append(mem.stderr[-3000:])
    return (done.stdout if attempt >= 0 else mem.stdout), kernel, runtime, attempt + 1


def main():
    wanted = [a for a in sys.argv[1:] if not a.startswith("-")] or ["gp", "codon", "hbm", "pipe", "slots"]
    build()
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for which in wanted:
            for program, case in cases(which, tmp):
                if not os.path.exists(os.path.join(EMU, "_build", "race", program)):
                    print(f"{which}: {program} not built here (the reference's sources are absent): skipped")
                    continue
                tag = os.path.splitext(os.path.basename(case))[0]
                out, kernel, runtime, attempts = run(program, case, tmp, tag)
                lines = len(out.strip().splitlines())
                how = f"attempt {attempts}" if attempts else "ThreadSanitizer's runtime crashed in all 12 attempts: memory check only"
                print(f"{which:6s} {os.path.basename(case):12s} ran ({lines} result lines, {how}): {len(kernel)} data races / memory errors in "
                      f"kernel or host code, {len(runtime)} reports inside the stand-in runtime's own bookkeeping (the interpreter is not annotated)")
                for r in kernel[:3]:
                    print("\n".join(r.strip().splitlines()[:14]))
                bad += len(kernel)
    print("data races and memory errors reported in kernel or host code:", bad)
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
