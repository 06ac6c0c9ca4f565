// The kernels the emulated build leaves out: walk_lds.hip / walk_tree.hip (inline assembly; pinned-only kernels that AUTO
// never picks).  Their planners report "does not apply" and their launchers are never reached (they abort if they are).
// walk_pipe.hip IS part of the build: its three asm statements run through the gfx950 interpreter (gfx950_asm.hpp), so
// AUTO routes as in the product -- walk_pipe_kernel up to 64 taxa and four rate categories without rescaling, the
// HBM-arena walks for the rest.
// Test infrastructure only (tests/hip_emu).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "kernels.hpp"

namespace bito_amd {

[[noreturn]] static void NotEmulated(const char* what) {
  std::fprintf(stderr, "hip_emu: %s is not part of the emulated build\n", what);
  std::abort();
}

LdsPlan PlanLds(const BatchDims&) { return LdsPlan{}; }
size_t LdsScheduleInts(const BatchDims&) { return 0; }
void LaunchLdsSchedule(const BatchDims&, const DeviceBatch&, const LdsPlan&, hipStream_t) { NotEmulated("walk_lds_kernel"); }
void LaunchWalkLds(const BatchDims&, const DeviceBatch&, const LdsPlan&, int, int, hipStream_t) { NotEmulated("walk_lds_kernel"); }
void LaunchMatrixImages(const BatchDims&, const DeviceBatch&, int, int, hipStream_t) { NotEmulated("walk_lds_kernel's matrix images"); }
TreePlan PlanTree(const BatchDims&) { return TreePlan{}; }
void LaunchWalkTree(const BatchDims&, const DeviceBatch&, const TreePlan&, int, hipStream_t) { NotEmulated("walk_tree_kernel"); }
}  // namespace bito_amd
