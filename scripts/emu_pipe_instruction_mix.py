"""The instruction stream of walk_pipe_kernel, counted twice: by the gfx950 interpreter of tests/hip_emu (wave-level
instructions of the kernel's asm statements by class, plus the matrix instructions the C++ around them issues through
builtins) and by the hardware's SQ_INSTS_* counters on an MI355X (round 4's last build, kept per dispatch in
profiles/r4_v2_pipe_one_wave_pmc_per_dispatch.json).  Workload: BASELINE config 3, DS1's 100 topologies x 64 replicas =
6400 trees per launch, log-likelihood + gradient, one wave per SIMD x 4 pattern groups.  The interpreter walks the 100
distinct trees once (counts scale by 64: replicas are the same trees) with the unit plan of the large launch (whole-tree
units; BITO_AMD_PIPE_WHOLE_TREES / BITO_AMD_LDS_TILE_RUN).  Matrix instructions must agree EXACTLY -- they are issued by
the asm statements and the root's builtin alone; the other classes agree up to what the compiled C++ around the
statements adds on the device (fibers here: not counted).
usage: python scripts/emu_pipe_instruction_mix.py [--write]   (--write: profiles/r5_emulated/pipe_instruction_mix.json)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hip_emu", "_build", "libbito_amd_emu.so")

BODY = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import bito_amd
from bito_amd import workloads
w = workloads.ds1_gtr_weibull4(1).subset({trees})
eng = bito_amd.Engine(bito_amd.PhyloModelSpecification(w.substitution, w.site, w.clock), w.patterns, w.weights)
out = eng.gradients(w.parent_ids, w.branch_lengths, w.params)
assert eng.kernel_name() == "walk_pipe_kernel" and "4 pattern groups" in eng.kernel_form(), eng.kernel_form()
assert np.all(np.isfinite(out["log_likelihood"]))
'''


def emulated_counts(trees, whole_tree_units=True):
    env = dict(os.environ, BITO_AMD_LIB=EMU, HIP_EMU_ASM_COUNT="1", HIP_EMU_ASM_HAZARDS="abort")
    if whole_tree_units:
        env.update(BITO_AMD_PIPE_WHOLE_TREES=str(trees), BITO_AMD_LDS_TILE_RUN="5")
    done = subprocess.run([sys.executable, "-c", BODY.format(root=ROOT, trees=trees)], capture_output=True, text=True, env=env)
    if done.returncode != 0:
        raise SystemExit(done.stdout[-2000:] + done.stderr[-4000:])
    counts = {}
    for line in done.stderr.splitlines():
        if line.startswith("{"):
            counts.update(json.loads(line))
    return counts


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hip_emu")])
    device = json.load(open(os.path.join(ROOT, "profiles", "r4_v2_pipe_one_wave_pmc_per_dispatch.json")))
    full = device["dispatches"]["19"]  # a resident pass: 6400 trees
    c100 = emulated_counts(100)
    c24 = emulated_counts(24)
    mfma100 = c100["mfma"] + c100["builtin_mfma_f64_4x4x4"]
    mfma24 = c24["mfma"] + c24["builtin_mfma_f64_4x4x4"]
    rows = [
        ("matrix (SQ_INSTS_MFMA)", 64 * mfma100, full["SQ_INSTS_MFMA"]),
        ("vector incl. matrix (SQ_INSTS_VALU)", 64 * (mfma100 + c100["valu_other"]), full["SQ_INSTS_VALU"]),
        ("LDS (SQ_INSTS_LDS)", 64 * c100["lds"], full["SQ_INSTS_LDS"]),
        ("scalar ALU + branches (SQ_INSTS_SALU)", 64 * (c100["salu"] + c100["branch"]), full["SQ_INSTS_SALU"]),
        ("scalar memory (SQ_INSTS_SMEM)", 64 * c100["smem"], full["SQ_INSTS_SMEM"]),
    ]
    print(f"{'class':40s} {'interpreter x 64':>18s} {'MI355X, 6400 trees':>20s} {'device - interpreter':>22s}")
    for name, emu, dev in rows:
        print(f"{name:40s} {emu:18,d} {int(dev):20,d} {int(dev) - emu:22,d}")
    chunk_1024 = 10 * mfma100 + mfma24  # the blocking call's first chunk: trees 0..1023 of the replicated collection
    chunk_5376 = 64 * mfma100 - chunk_1024
    print(f"blocking call's chunks, matrix instructions: 1024 trees {chunk_1024:,d} (device {int(device['dispatches']['8']['SQ_INSTS_MFMA']):,d}), "
          f"5376 trees {chunk_5376:,d} (device {int(device['dispatches']['15']['SQ_INSTS_MFMA']):,d})")
    exact = (64 * mfma100 == int(full["SQ_INSTS_MFMA"]) and chunk_1024 == int(device["dispatches"]["8"]["SQ_INSTS_MFMA"])
             and chunk_5376 == int(device["dispatches"]["15"]["SQ_INSTS_MFMA"]))
    print("matrix instructions agree exactly:", exact)
    print(f"per tree: {mfma100 / 100:.1f} matrix instructions ({mfma100 / 100 * 512 / 1e6:.2f} MFLOP executed on the matrix pipe)")
    if "--write" in sys.argv:
        out = {"workload": "DS1 100 topologies x 64 = 6400 trees, GTR + weibull+4, log-likelihood + gradient, walk_pipe_kernel<4,4,true,0>",
               "interpreter_100_trees": c100, "interpreter_24_trees": c24,
               "device_dispatch_6400_trees": {k: v for k, v in full.items() if k.startswith("SQ_INSTS")},
               "matrix_instructions_per_tree": mfma100 / 100, "matrix_instructions_agree_exactly": exact,
               "table": [{"class": n, "interpreter_x64": e, "device": int(d)} for n, e, d in rows]}
        path = os.path.join(ROOT, "profiles", "r5_emulated", "pipe_instruction_mix.json")
        json.dump(out, open(path, "w"), indent=1)
        print("wrote", path)
    if not exact:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
