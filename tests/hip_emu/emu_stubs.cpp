// The kernels the emulated build leaves out: walk_pipe.hip (gfx950 assembly), walk_lds.hip / walk_tree.hip (DPP and
// inline assembly; pinned-only kernels) and gs_kernels.hip (MFMA, DMA to LDS).  Their planners report "does not apply",
// so AUTO routes every four-state batch to the HBM-arena walks -- the code path the product takes for rescaling,
// five to eight rate categories and more than 64 taxa -- and their launchers are never reached (they abort if they are).
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
LdsPlan PlanPipe(const BatchDims&) { return LdsPlan{}; }
LdsPlan PlanPipeClass(const BatchDims&, int, int, int, int) { return LdsPlan{}; }
int PipeMaxSlots(const BatchDims&, int, int) { return 0; }
int PipeSlotsOfTree(const BatchDims&, int) { return 1 << 20; }
bool PipeTwoApplies(const BatchDims&) { return false; }
size_t PipeScheduleInts(const BatchDims&) { return 0; }
size_t PipeMaskInts(const BatchDims&, const LdsPlan&) { return 0; }
void LaunchPipeMasks(const BatchDims&, const DeviceBatch&, const LdsPlan&, uint32_t*, hipStream_t) { NotEmulated("walk_pipe_kernel"); }
void LaunchPipePrepare(const BatchDims&, const DeviceBatch&, const LdsPlan&, hipStream_t, bool, int, int, int, const uint8_t*) { NotEmulated("walk_pipe_kernel"); }
void LaunchWalkPipe(const BatchDims&, const DeviceBatch&, const LdsPlan&, int, int, int, hipStream_t, const PipeClass&) { NotEmulated("walk_pipe_kernel"); }
void LaunchMatrixImages(const BatchDims&, const DeviceBatch&, int, int, hipStream_t) { NotEmulated("walk_lds_kernel's matrix images"); }
TreePlan PlanTree(const BatchDims&) { return TreePlan{}; }
void LaunchWalkTree(const BatchDims&, const DeviceBatch&, const TreePlan&, int, hipStream_t) { NotEmulated("walk_tree_kernel"); }
size_t GsArenaDoublesPerTree(const BatchDims&, int, int) { NotEmulated("the general-state kernels"); }
size_t GsImageDoublesPerTree(const BatchDims&) { NotEmulated("the general-state kernels"); }
void LaunchGsSetup(const BatchDims&, const ModelSpec&, const DeviceBatch&, const int32_t*, double*, hipStream_t, bool) { NotEmulated("the general-state kernels"); }
void LaunchGsMatrices(const BatchDims&, int, int, int, const double*, const int32_t*, const double*, double*, int, int, hipStream_t) { NotEmulated("the general-state kernels"); }
int GsScheduleStride(const BatchDims&) { NotEmulated("the general-state kernels"); }
void LaunchGsSchedule(const BatchDims&, const DeviceBatch&, hipStream_t) { NotEmulated("the general-state kernels"); }
void LaunchGsWalk(const BatchDims&, int, const DeviceBatch&, const int32_t*, const double*, int, int, int, int, int, int, hipStream_t) { NotEmulated("the general-state kernels"); }

}  // namespace bito_amd
