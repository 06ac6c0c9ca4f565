#!/bin/bash
# usage: scripts/emu_gpu_suite.sh [log]   (CPU only)
# The `-m gpu` tests AS THEY ARE against the emulated library (tests/hip_emu: the .hip sources as fibers, walk_pipe_kernel's
# assembly interpreted with its hazard checks fatal) -- what a round without GPU access can say about them.  Deselected:
# the parameters that pin KERNEL_LDS / KERNEL_LDS_TREE (walk_lds.hip / walk_tree.hip are outside the emulated build).
# A test that needs more than 240 s as fibers (full-size batches: thousands of trees, 1000 taxa x 10 000 patterns) is cut off
# by pytest-timeout's thread method -- the worker process is ended and replaced, the log shows "worker ... crashed" or
# "Timeout" for it; the summary at the end of the log lists them.  Evidence of the kernels' logic, not of the hardware.
cd "$(dirname "$0")/.."
make -s -C tests/hip_emu || exit 1
mkdir -p tests/hip_emu/_build/as_product
ln -sf ../libbito_amd_emu.so tests/hip_emu/_build/as_product/libbito_amd.so
LOG=${1:-profiles/r5_emulated/gpu_suite_emulated.log}
BITO_AMD_LIB=$PWD/tests/hip_emu/_build/libbito_amd_emu.so HIP_EMU_ASM_HAZARDS=abort \
LD_LIBRARY_PATH=$PWD/tests/hip_emu/_build/as_product:$LD_LIBRARY_PATH \
python -m pytest tests -m gpu -q -p no:cacheprovider -n 6 --timeout 240 --timeout-method=thread -rfE --durations=15 > "$LOG" 2>&1
tail -5 "$LOG"
