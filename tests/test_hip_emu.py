"""The stand-in HIP runtime of tests/hip_emu against itself and against a record from the device.  selftest.hip: a block
reduction through shuffles, dynamic LDS and two barriers with three of four waves leaving in between; the DPP /
permlane sums of bito_amd/csrc/wave_sums.hpp, shuffles, readfirstlane, ballots and votes (in uniform and in divergent
code: a vote counts the lanes that take part, as the hardware's EXEC mask does); atomics across workgroups.
probe_mfma16.hip: the program that measured v_mfma_f64_16x16x4's lane layout and rounding on an MI355X in round 1 must
print under emulation what it printed on the device (profiles/r1_mfma16_probe.json)."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU = os.path.join(HERE, "hip_emu")


def _build():
    built = subprocess.run(["make", "-s", "-C", EMU, "tools"], capture_output=True, text=True)
    assert built.returncode == 0, built.stdout + built.stderr


def test_emulator_semantics():
    _build()
    done = subprocess.run([os.path.join(EMU, "_build", "selftest")], capture_output=True, text=True, timeout=120,
                          env=dict(os.environ, HIP_EMU_ASM_HAZARDS="1"))  # (hazards reported and counted, not fatal)
    assert done.returncode == 0, done.stdout + done.stderr
    # (asm_interpreter: the gfx950 interpreter's matrix instruction bit for bit the builtin's, and its hazard check on ten
    # small programs -- an LDS read used behind / before its s_waitcnt, s_movrels with / without the wait state behind a
    # write of M0, lgkmcnt(1) with a scalar load in flight, vector reads and matrix operands with and without their wait states -- raising hazards exactly where the ISA's rules are broken)
    assert done.stdout.split() == ["asm_interpreter", "ok", "block_sum", "ok", "wave_ops", "ok", "counter", "ok"]


def test_emulated_mfma_is_what_the_device_measured():
    _build()
    done = subprocess.run([os.path.join(EMU, "_build", "probe_mfma16_emu")], capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stdout + done.stderr
    got = json.loads(done.stdout)
    with open(os.path.join(ROOT, "profiles", "r1_mfma16_probe.json")) as fh:
        device = json.load(fh)
    for key in ("matched", "of", "column_is_lane_mod_16", "bit_equal_to_sequential_fma_chain", "row_of[lane/16][register]"):
        assert got[key] == device[key], key


def test_thread_sanitizer_sees_a_missing_barrier():
    """The ThreadSanitizer build of the stand-in runtime (every GPU thread a TSan fiber; __syncthreads() and the wave
    operations the only happens-before edges inside a workgroup): a kernel that reads LDS another wave wrote with no
    barrier between is reported as a data race with both source lines -- whatever order the serial fibers happened to
    run it in --; the same kernel with its __syncthreads() is clean.  scripts/emu_race_check.py runs the product's kernels this
    way (GP executor, codon kernels, walk_pipe_kernel's C++) and under AddressSanitizer."""
    built = subprocess.run(["make", "-s", "-C", EMU, "_build/racetest"], capture_output=True, text=True)
    assert built.returncode == 0, built.stdout + built.stderr
    env = dict(os.environ, TSAN_OPTIONS="report_signal_unsafe=0 exitcode=0")

    def attempt(arg):
        # (this image's TSan runtime dies at start-up now and then -- a SEGV inside the runtime before the kernel runs,
        # dependent on the process's address-space layout: such a run says nothing and is repeated)
        for _ in range(12):
            done = subprocess.run([os.path.join(EMU, "_build", "racetest"), arg], capture_output=True, text=True, timeout=120, env=env)
            if "DEADLYSIGNAL" not in done.stderr and done.stdout.startswith("out["):  # (the program ran to its end)
                return done
        pytest.skip("ThreadSanitizer's runtime does not start in this environment")

    racy, clean = attempt("0"), attempt("1")
    assert "out[0] 128" in clean.stdout
    assert "WARNING: ThreadSanitizer: data race" in racy.stderr and "exchange_kernel<false>" in racy.stderr
    assert "racetest.hip:14" in racy.stderr and "racetest.hip:16" in racy.stderr  # (the LDS write and the read)
    assert "ThreadSanitizer" not in clean.stderr, clean.stderr[-2000:]
