#!/bin/bash
# usage: scripts/emu_gpu_suite.sh [log]   (CPU only)
# The `-m gpu` tests AS THEY ARE against the emulated library (tests/hip_emu: the .hip sources as fibers, walk_pipe_kernel's
# assembly interpreted with its hazard checks fatal) -- what a round without GPU access can say about them.  Deselected:
# the parameters that pin KERNEL_LDS / KERNEL_LDS_TREE (walk_lds.hip / walk_tree.hip are outside the emulated build).
# Tests that need more than 300 s as fibers (full-size batches) show up as timeouts in the log; it is evidence of the
# kernels' logic, not of the hardware.
cd "$(dirname "$0")/.."
make -s -C tests/hip_emu || exit 1
mkdir -p tests/hip_emu/_build/as_product
ln -sf ../libbito_amd_emu.so tests/hip_emu/_build/as_product/libbito_amd.so
LOG=${1:-profiles/r5_emulated/gpu_suite_emulated.log}
BITO_AMD_LIB=$PWD/tests/hip_emu/_build/libbito_amd_emu.so HIP_EMU_ASM_HAZARDS=abort \
LD_LIBRARY_PATH=$PWD/tests/hip_emu/_build/as_product:$LD_LIBRARY_PATH \
python -m pytest tests -m gpu -q -p no:cacheprovider -n 4 --timeout 300 -rfE --durations=15 > "$LOG" 2>&1
tail -5 "$LOG"
