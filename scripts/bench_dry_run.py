"""bench.py's control flow, end to end, WITHOUT a GPU: the emulated library (tests/hip_emu) in place of libbito_amd.so, the
three torch.cuda calls bench.py makes turned into no-ops, and the workloads shrunk to a handful of tiny trees -- so that a
slip in the bench script (a misspelt key, a wrong shape) is found on the CPU and not by the one GPU run a round may get.
The numbers it prints mean nothing.  usage: python scripts/bench_dry_run.py [ds1 | ds1-dist | ds1-2ranks | codon | config4 | gp | gp-seeded]   (ds1-dist: the
summed-log-likelihood all-reduce of a multi-rank run, on a one-rank gloo group; ds1-2ranks: two processes under
torch.distributed.run as the driver launches --gpus 2, gloo in RCCL's place)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (BENCH_DRY_RUN_LIB: another build of the emulated library, e.g. tests/hip_emu/_build/memcheck/libbito_amd_asan.so with
# the AddressSanitizer runtime preloaded)
EMU = os.environ.get("BENCH_DRY_RUN_LIB", os.path.join(ROOT, "tests", "hip_emu", "_build", "libbito_amd_emu.so"))

BODY = r'''
import os, runpy, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a: None
if os.environ.get("BENCH_DRY_RUN_RANKS"):  # (two ranks under torch.distributed.run: the emulated runtime has one device)
    os.environ["LOCAL_RANK"] = "0"
if os.environ.get("BENCH_FORCE_DIST") == "1":
    # the multi-rank code path on one rank: a gloo group in place of RCCL, "cuda" tensors that stay on the host
    import torch.distributed as dist
    real_init, real_tensor = dist.init_process_group, torch.tensor
    def init(backend, **kw):
        kw.pop("device_id", None)
        return real_init("gloo", **kw)
    dist.init_process_group = init
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.tensor = lambda *a, **k: real_tensor(*a, **{{kk: vv for kk, vv in k.items() if kk != "device"}})
    real_device = torch.device
    class FakeDevice:  # (bench.py builds torch.device("cuda", rank) for device_id, which init() above drops)
        def __new__(cls, *a, **k):
            return real_device("cpu") if a and a[0] == "cuda" else real_device(*a, **k)
    torch.device = FakeDevice
    import bito_amd.dist as bdist
    class HostReducer:  # (ResidentSumReducer sums on the device and reduces with RCCL: its own tests are tests/test_dist_*.py)
        def __init__(self, eng):
            self.eng = eng
        def run(self, want_gradient, rescaling):
            self.eng.run(want_gradient, rescaling)
        def finish(self):
            pass
    bdist.ResidentSumReducer = HostReducer
from bito_amd import workloads
small = lambda n, P, T: workloads.synthetic_gtr_weibull4(n, P, tree_count=T)
def tiny_ds1(replicas=1, first_tree=0, tree_count=None):
    w = small(6, 24, tree_count or 100 * replicas)
    w.rescaling = False
    return w
workloads.ds1_gtr_weibull4 = tiny_ds1
real_codon = workloads.flua_codon
workloads.flua_codon = lambda T=64, site="constant", seed=20240605: real_codon(min(T, 2), site, seed)
real_synth = workloads.synthetic_gtr_weibull4
workloads.synthetic_gtr_weibull4 = lambda n=1000, P=10000, tree_count=125, first_tree=0: real_synth(40, 130, min(tree_count, 3), first_tree)
sys.argv = ["bench.py"] + (sys.argv[1:] if len(sys.argv) > 1 else {argv!r})  # (arguments given: bench.py's own child)
runpy.run_path(os.path.join({root!r}, "bench.py"), run_name="__main__")
'''


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "ds1"
    # ds1-2ranks: bench.py as the driver launches it for --gpus 2 (torch.distributed.run, two processes), gloo in RCCL's place
    two_ranks = which.endswith("-2ranks")
    force_dist = which.endswith("-dist") or two_ranks  # ds1-dist: BENCH_FORCE_DIST=1, the reduce path of a multi-rank run on one rank
    which = which.replace("-dist", "").replace("-2ranks", "")
    argv = {"ds1": ["--steps", "2", "--warmup", "1", "--replicas", "1", "--cpu-seconds", "2"],
            "codon": ["--workload", "codon", "--trees", "2", "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"],
            "config4": ["--workload", "config4", "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"],
            "gp": ["--workload", "gp", "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"],
            "gp-seeded": ["--workload", "gp", "--gp-dag", "seeded", "--steps", "1", "--warmup", "1", "--cpu-seconds", "1"]}[which]
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hip_emu")])
    env = dict(os.environ, BITO_AMD_LIB=EMU)
    if force_dist:
        env["BENCH_FORCE_DIST"] = "1"
    if two_ranks:
        argv = ["--gpus", "2"] + argv
    body = BODY.format(root=ROOT, argv=argv)
    env["BENCH_CHILD_CMD"] = json.dumps([sys.executable, "-c", body])  # (bench.py's large-batch child: the same tiny workloads)
    if two_ranks:
        import tempfile

        env["BENCH_DRY_RUN_RANKS"] = "2"
        with tempfile.NamedTemporaryFile("w", suffix="_bench_dry_run.py", delete=False) as fh:
            fh.write(body)
        try:
            done = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                   "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29553"), fh.name],
                                  capture_output=True, text=True, env=env)
        finally:
            os.unlink(fh.name)
    else:
        done = subprocess.run([sys.executable, "-c", body], capture_output=True, text=True, env=env)
    if done.returncode != 0:
        sys.stderr.write(done.stdout[-2000:] + done.stderr[-4000:])
        raise SystemExit(f"bench.py {which}: exit code {done.returncode}")
    line = json.loads([ln for ln in done.stdout.strip().splitlines() if ln.startswith("{")][-1])
    if two_ranks:
        assert line["n_gpus"] == 2 and "summed_log_likelihood" in line["config"], line["config"]
    keys = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline", "parity", "model_cache_hit", "resident"]
    if which.startswith("gp"):
        keys = [k for k in keys if k not in ("model_cache_hit", "resident")]
    if which == "ds1" and not force_dist:
        keys.append("large_batch")
    if two_ranks:  # (the CPU baseline is rank 0's at N = 1 only)
        keys.remove("cpu_baseline")
    missing = [k for k in keys if k not in line]
    print(f"bench.py {which}: one JSON line, {len(line)} keys; missing {missing}; parity {line.get('parity')}")
    print("   large_batch", line.get("large_batch"))
    print("   model_cache_hit", line.get("model_cache_hit"), "\n   blocking_call_ms", line.get("blocking_call_ms"),
          "\n   distinct_models", line.get("distinct_models"), "\n   config", {k: v for k, v in line["config"].items() if k != "workload"})
    if missing:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
