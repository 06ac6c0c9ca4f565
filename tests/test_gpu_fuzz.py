"""A seeded slice of the randomised parity sweeps under the driver's eyes (the full sweeps are run by hand, their
logs kept under profiles/): scripts/gpu_fuzz.py -- random tree shapes (3 to 90 taxa, 40 / 44 / 48 among them),
pattern counts, gap rates, models, rate categories, rooted and unrooted, branch lengths at three scales with a tenth of
the branches exactly zero in a fifth of the cases, rescaling on and off, GPU against the CPU checker at 1e-10 / 1e-6 --
with the kernel drawn at random and with each traversal kernel forced in turn, and scripts/gpu_gp_fuzz.py for Path B
(random subsplit DAGs, schedules, NNI proposals, optimisation sweeps)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sweep(script, *args, env=None):
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), *[str(a) for a in args]],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500, cwd=ROOT,
                          env={**os.environ, **(env or {})})
    out = proc.stdout
    try:  # (kept next to the other measurements when the directory is there)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"fuzz_{script[:-3]}_{'_'.join(str(a) for a in args)}.log"), "w") as fh:
            fh.write(out)
    except OSError:
        pass
    assert proc.returncode == 0, out[-2000:]
    summary = re.search(r"(\d+) cases, (\d+) bad(?:, (\d+) declined by a forced kernel)?", out)
    assert summary, out[-2000:]
    assert int(summary.group(2)) == 0, out[-4000:]
    return int(summary.group(1)), int(summary.group(3) or 0)


@pytest.mark.parametrize("cases,seed,kernel", [(160, 3101, None), (80, 3102, 5), (80, 3103, 1), (80, 3104, 2), (120, 3105, 6)],
                         ids=["any-kernel", "walk_pipe_kernel", "hbm-arena", "walk_lds_kernel", "walk_pipe_kernel-two-waves"])
def test_seeded_sweep_per_tree_path(cases, seed, kernel):
    args = (cases, seed) if kernel is None else (cases, seed, kernel)
    done, declined = _sweep("gpu_fuzz.py", *args)
    assert done == cases
    assert declined < cases  # (a forced kernel declines the shapes it does not take; it must take some)


def test_seeded_sweep_hbm_arena_walk_large_trees_folded_and_not():
    """The HBM-arena walk pinned, trees of up to 333 taxa among the sizes: as built (pitchforks -- a tip and a cherry under
    one node -- rebuilt from their tips' matrix rows like cherries: no step, no cell) and with a step and a cell for every
    pitchfork (BITO_AMD_HBM_FOLD=0, read once per process: hence the sweep's own process)."""
    for env in ({"FUZZ_LARGE_TREES": "1"}, {"FUZZ_LARGE_TREES": "1", "BITO_AMD_HBM_FOLD": "0"}):
        done, declined = _sweep("gpu_fuzz.py", 100, 3106, 1, env=env)
        assert done == 100 and declined == 0


def test_seeded_sweep_gp_executor():
    done, _ = _sweep("gpu_gp_fuzz.py", 100, 3202)
    assert done == 100
