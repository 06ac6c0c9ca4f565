"""bench.py is the driver's contract and may get ONE run on the GPU per round: its control flow is exercised on the CPU --
the emulated library (tests/hip_emu) in place of libbito_amd.so, torch.cuda's calls as no-ops, workloads of a few tiny
trees (scripts/bench_dry_run.py) -- so that a misspelt key or a list emptied too early is found here.  The headline
workload runs with BENCH_FORCE_DIST=1 on a one-rank gloo group: the summed-log-likelihood all-reduce of a multi-rank run
and its check against the gathered sum are part of the flow (round 5: the cache-hit loop had cleared the pending
reductions that check reads); ds1-2ranks runs bench.py as the driver launches it for --gpus 2 -- two processes under
torch.distributed.run, gloo in RCCL's place: sharding by rank, the max-over-ranks timing, the reduced sum against the
gathered one.  The numbers mean nothing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["ds1-dist", "ds1-2ranks", "gp"])
def test_bench_py_runs_end_to_end_on_the_emulated_library(workload):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    done = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_dry_run.py"), workload], capture_output=True,
                          text=True, timeout=900, env=env, cwd=ROOT)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-3000:]
    assert "one JSON line" in done.stdout and "missing []" in done.stdout, done.stdout[-1500:]
