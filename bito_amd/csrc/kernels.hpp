// kernels.hpp -- launch interface between the host engine and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>

#include "model.hpp"

namespace bito_amd {

// Per (tree, branch, category) matrix record, 72 doubles:
//   [ 0,16) P      row-major          P[i][j]   = V exp(lambda t r_c) V^-1
//   [16,32) dP     row-major          dP = r_c Q P  (first derivative wrt t)
//   [32,52) PT5    [state 0..4][i]    column lookup for tip children; row 4 (gap) = 1
//   [52,72) dPT5   [state 0..4][i]    same for dP; row 4 (gap) = 0 (rows of Q sum to 0)
constexpr int kMatStride = 72;
// (what walk_hbm_cat_kernel reads -- P and the rows of P^T -- first: that walk's records are these kMatHot doubles only,
// half the bytes that must stay in an XCD's L2 while a tree is walked)
constexpr int kMatP = 0, kMatPT = 16, kMatHot = 36, kMatDP = 36, kMatDPT = 52;

struct BatchDims {
  int32_t taxon_count;    // n
  int32_t node_count;     // N = 2n-1 (after detrifurcation)
  int32_t in_node_count;  // M = 2n-2 (unrooted input) or 2n-1 (rooted)
  int32_t rooted;
  int32_t pattern_count;  // P
  int32_t pattern_stride; // Ppad (multiple of 64)
  int32_t category_count; // C
  int32_t tree_count;     // T
  int32_t min_cherries;   // fewest cherries (internal nodes over two tips, root excepted) of any tree of the batch
  int32_t min_unstored;   // fewest nodes walk_pipe_kernel keeps no vector for: cherries and, with pipe_fold, the pitchforks
                          // (a tip and a cherry under one node) it folds into their parents' steps
  int32_t pipe_fold;      // walk_pipe_kernel folds pitchforks like cherries (worker.cpp counts them the same way)
};

// MFMA operand images, one set per (tree, branch), 3 x 64 doubles, lane order of
// v_mfma_f64_4x4x4_4b's A operand (lane = 16 k + 4 block + row holds M_block[row][k];
// block = category (+ C * pattern sub-group when C < 4)):
//   [0,128)   (P, dP) pairs, one pair per lane: P -> (P x) child message, dP -> (dP x) its derivative
//             wrt the branch length; interleaved so that one 16-byte load fetches both
//   [128,192) P^T   ->  P^T (u.a)  pre-order partial of the child
constexpr int kImgStride = 192;
constexpr int kImgP = 0, kImgDP = 64, kImgPT = 128;

struct DeviceBatch {
  // inputs, resident in HBM (wire format of the reference)
  const int32_t* parent_ids;  // [T][M-1]
  const double* branch_in;    // [T][M]
  const double* rates;        // [T][M-1] or nullptr
  const double* params;       // [T][param_count]
  // A blocking call's chunk hands the set-up kernel its inputs in pinned HOST memory (the four pointers above;
  // device-readable) instead of copying them first; the kernel then also leaves the device copies later passes
  // over the batch read (null otherwise).  setup_trees_lds_kernel only: SetupReadsHostInputs(d).
  int32_t* copy_parent_ids;
  double *copy_branch_in, *copy_rates, *copy_params;
  // alignment
  const uint8_t* tip_states;  // [n][Ppad], 4 = gap (padding = gap)
  const double* weights;      // [Ppad], padding = 0
  // produced by the set-up kernels
  int32_t* children;          // [T][n-1][2]  children of internal node n+k
  double* branch;             // [T][N]       effective branch lengths
  TreeModel* model;           // [T]
  // A small blocking call whose parameter rows all equal the row the LAST such call's tree 0 had (the host compares them
  // while it stages the call): setup_trees_lds_kernel copies that tree's model -- kept in model_cache -- instead of
  // forming the rate matrix, the eigensystem and the category rates again (the serial 4 x 4 Jacobi is 10 of a 100-tree
  // call's 26 us of set-up).  Same bits: the copy is what the same arithmetic on the same row gave.  Null otherwise.
  const TreeModel* model_reuse;
  TreeModel* model_cache;     // where tree 0's model is left when it was computed (null: nowhere)
  double* mats;               // [T][N-1][C][kMatStride]   (HBM-arena kernel; [T][N-1][C][kMatHot] for walk_hbm_cat_kernel)
  double* images;             // [T][N-1][kImgStride]      (LDS kernel)
  int32_t* sched;             // [T][2][n+1][16]           step descriptors (LDS kernel); step records in visiting order (HBM-arena walk)
  const uint32_t* pipe_masks; // [tiles][n][waves][16/C]   packed tip masks per pattern tile (walk_pipe_kernel)
  int32_t* pipe_queue;        // [2] next unit of work, workgroups that have left (walk_pipe_kernel; zero between launches)
  uint8_t* pipe_done;         // [T] or null: walk_pipe_kernel's whole-tree units write their tree's final results themselves
                              //     (out_ll, out_grad) and set the tree's flag; the final-sums kernel leaves those trees alone
  int hbm_fold;               // what walk_hbm_cat_kernel rebuilds where it is used (HbmFoldLevel() when the pass's step records
                              // were written: the walk that reads them must be the one they were written for)
  // traversal scratch + outputs
  double* arena;              // [chunk][n-1][C][4][Ppad]
  double* scale_arena;        // [chunk][n-1][Ppad]  post-order 1/scale factors (rescaled gradients)
  double* part_ll;            // [T][tiles]
  double* part_grad;          // [T][tiles][N]  (HBM-arena kernel: [T][tiles][4 waves][N])
  double* out_ll;             // [T]
  double* out_grad;           // [T][N]
  double* out_site;           // [T] site-model gradient, when the traversal kernel produces it (else unused)
};

// Topology set-up of one tree, shared by the 4-state and the general-state set-up kernels:
// parent-id vector -> child lists (+ detrifurcation), effective branch lengths.  The arrays may live
// in global memory or in LDS; bl must already hold the M input branch lengths.
__device__ inline void SetupTopologyCore(const BatchDims& d, const int32_t* parent, int32_t* ch, double* bl,
                                         const double* rates) {
  const int n = d.taxon_count, N = d.node_count, M = d.in_node_count, NI = n - 1;
  for (int k = 0; k < NI * 2; k++) ch[k] = -1;
  int third = -1;
  // Children in ascending id order, as Node::OfParentIdVector builds them
  // (reference src/node.cpp:511-551).
  for (int child = 0; child < M - 1; child++) {
    const int k = parent[child] - n;
    if (ch[k * 2] < 0) {
      ch[k * 2] = child;
    } else if (ch[k * 2 + 1] < 0) {
      ch[k * 2 + 1] = child;
    } else {
      third = child;
    }
  }
  if (!d.rooted) {
    // UnrootedTree::Detrifurcate (reference src/unrooted_tree.cpp:27-37): children 1 and
    // 2 of the trifurcation are joined under a node that re-uses the old root id
    // with branch length 0; the new root (id+1) joins child 0 with it.
    // Tree::SlideRootPosition (src/tree.cpp:82-88) is then the identity apart from
    // pinning that branch to 0.
    const int r = M - 1;
    const int a = ch[(r - n) * 2], bb = ch[(r - n) * 2 + 1];
    ch[(r - n) * 2] = bb;
    ch[(r - n) * 2 + 1] = third;
    bl[r] = 0.0;
    ch[(r + 1 - n) * 2] = a;
    ch[(r + 1 - n) * 2 + 1] = r;
    bl[r + 1] = 0.0;
  } else if (rates != nullptr) {
    // FatBeagle::LogLikelihood(RootedTree) (reference src/fat_beagle.cpp:86-90).
    for (int i = 0; i < N - 1; i++) bl[i] *= rates[i];
  }
}

__device__ inline void SetupTopology(const BatchDims& d, const DeviceBatch& b, int t) {
  const int n = d.taxon_count, N = d.node_count, M = d.in_node_count;
  double* bl = b.branch + (size_t)t * N;
  const double* bl_in = b.branch_in + (size_t)t * M;
  for (int i = 0; i < M; i++) bl[i] = bl_in[i];
  SetupTopologyCore(d, b.parent_ids + (size_t)t * (M - 1), b.children + (size_t)t * (n - 1) * 2, bl,
                    b.rates != nullptr ? b.rates + (size_t)t * (M - 1) : nullptr);
}

// ---- the staging set-up kernels (kernels.hip: setup_trees_lds_kernel; walk_pipe.hip: pipe_setup_kernel) ----
inline size_t SetupLdsBytesPerTree(const BatchDims& d, const ModelSpec& spec) {
  // effective branch lengths [N], parameter row, rates [M-1] (rooted trees only: 0 doubles when there are none is
  // decided at launch; sized for them here), parent ids [M-1], child lists [2 NI]
  return ((size_t)d.node_count + std::max(spec.param_count, 1) + (d.rooted ? d.in_node_count - 1 : 0)) * sizeof(double) +
         (d.in_node_count - 1 + 2 * (d.taxon_count - 1)) * sizeof(int32_t);
}

// The wire-format rows of the workgroup's trees go through LDS: coalesced loads (from HBM, or straight from the
// caller's pinned staging buffer over PCIe -- a blocking call's chunk, which then also leaves the device copies for
// later passes, DeviceBatch::copy_*), one thread per tree out of LDS, coalesced stores.
// (element i of a workgroup's slice of a wire-format array: loaded by thread i mod kStep (the workgroup size), four elements per thread in
// flight at once -- over PCIe a load takes two microseconds, and a loop that waits for each one before it issues the
// next costs a workgroup of sixteen trees thirty)
template <int kStep, typename T, typename Store>
__device__ __forceinline__ void StageSlice(const T* __restrict__ src, T* __restrict__ copy, int total, int tid,
                                           const Store& store) {
  constexpr int kInFlight = 4;
  for (int i0 = tid; i0 < total; i0 += kStep * kInFlight) {
    T v[kInFlight];
#pragma unroll
    for (int u = 0; u < kInFlight; u++) {
      const int i = i0 + u * kStep;
      v[u] = i < total ? src[i] : T();
    }
#pragma unroll
    for (int u = 0; u < kInFlight; u++) {
      const int i = i0 + u * kStep;
      if (i < total) {
        store(i, v[u]);
        if (copy != nullptr) copy[i] = v[u];
      }
    }
  }
}


// beside_traversal: the launch will run next to a traversal that needs whole CUs, so it should sit on as few
// CUs as possible (up to 128 trees per workgroup) rather than finish as early as possible (16 per workgroup)
void LaunchSetup(const BatchDims& d, const ModelSpec& spec, const DeviceBatch& b, int want_gradient,
                 hipStream_t stream, bool beside_traversal = false);
// does LaunchSetup pick the kernel that stages a workgroup's wire-format rows through LDS -- the one that can read
// them from pinned host memory and leave the device copies (DeviceBatch::copy_*)?
bool SetupReadsHostInputs(const BatchDims& d, const ModelSpec& spec);
// deriv_mode 0: dP = P (r_c Q); 1: dP = P ((d r_c / d shape) Q) for the site-model gradient pass.
void LaunchMatrices(const BatchDims& d, const DeviceBatch& b, int want_gradient, int deriv_mode,
                    hipStream_t stream, bool hot_only = false);
// Same arithmetic, written as MFMA operand images (one wave per tree-branch).
void LaunchMatrixImages(const BatchDims& d, const DeviceBatch& b, int want_gradient, int deriv_mode,
                        hipStream_t stream);

// LDS-resident traversal (v_mfma_f64_4x4x4_4b): 4 waves per workgroup, each wave owns
// G groups of 16/C site patterns and keeps all n-2 stored PLVs of its patterns in LDS.
struct LdsPlan {
  int groups;          // G (0 = the tree does not fit: use the HBM-arena kernel)
  int patterns_per_block;
  int tiles;           // workgroups per tree
  size_t lds_bytes;
  int tile_run = 1;    // (walk_pipe_kernel) consecutive tiles of a tree walked by one workgroup
  int grad_rows = 0;   // (walk_pipe_kernel) partial gradient rows per tree: one per run; 0 = one per tile
  int whole_trees = 0; // (walk_pipe_kernel) the first so many trees are walked by one workgroup each, all tiles
  int slots = 0;       // (walk_pipe_kernel) vectors a wave keeps in LDS
  int layout = 0;      // (walk_pipe_kernel) 0 / 1: one wave per SIMD (1: the wide loops, 49 taxa and more), 2: two waves per SIMD
};
LdsPlan PlanLds(const BatchDims& d);
size_t LdsScheduleInts(const BatchDims& d);
void LaunchLdsSchedule(const BatchDims& d, const DeviceBatch& b, const LdsPlan& plan, hipStream_t stream);
// want_site: also produce the site-model gradient (per-tile value in the root's slot of part_grad)
void LaunchWalkLds(const BatchDims& d, const DeviceBatch& b, const LdsPlan& plan, int want_gradient, int want_site,
                   hipStream_t stream);

// completion flag of a blocking call's chunk (pinned host memory), stored by the last workgroup that counts itself
struct ReduceDone {
  unsigned long long* flag = nullptr;
  unsigned long long ticket = 0;
  int* counter = nullptr;
};
// LDS-resident traversal with hand-scheduled loops (walk_pipe.hip): the mapping of walk_lds_kernel,
// stored child MESSAGES instead of partials, both tree loops software-pipelined gfx950 assembly.
// Images: [T][N-1][128] doubles ((P, P^T) per lane); step tables: [T][2][n+1][8] dwords in b.sched; the
// packed tip masks of every pattern tile (a function of the alignment and the plan) in b.pipe_masks.
LdsPlan PlanPipe(const BatchDims& d);
// A batch may be walked as two classes of trees, each with its own plan (trees with few cherries keep more
// vectors per wave and leave room for fewer pattern groups): the plan for `tree_count` trees that keep at most
// `slots` vectors, the most vectors that fit beside G groups, and a tree's count from its cherries.
constexpr int kPipePlanAuto = 0, kPipePlanTwoWaves = 2;  // (LdsPlan::layout as PlanPipeClass takes it)
LdsPlan PlanPipeClass(const BatchDims& d, int tree_count, int slots, int force_groups, int layout = kPipePlanAuto);
int PipeMaxSlots(const BatchDims& d, int G, int layout = kPipePlanAuto);
// two waves per SIMD (round 4): trees of up to 28 taxa, one image per branch (reversible form), 256 registers per wave
bool PipeTwoApplies(const BatchDims& d);
int PipeSlotsOfTree(const BatchDims& d, int unstored);  // (cherries + folded pitchforks: Worker::tree_cherries)
struct PipeClass {
  int tree_count;          // trees of this launch
  const int32_t* order;    // their ids (device), or nullptr: trees 0 .. tree_count-1
  const uint32_t* masks;   // packed tip masks built for this launch's plan
  int row_stride;          // partial rows per tree the reduction adds up
  int reserve_cus = 0;     // CUs this launch leaves free: a blocking call's next chunk has its set-up kernels to run
                           // while this traversal holds every other CU (engine.cpp)
  // round 6 (BITO_AMD_PIPE_LAST_UNIT=1): a tree's LAST run-of-tiles unit forms its final sums, so that no final-sums
  // launch follows the traversal -- a counter per tree (zero before the launch; the unit that finds it at runs - 1
  // resets it), and the chunk's completion flag stored by the workgroup that finishes the launch's last tree
  int* tree_units = nullptr;
  ReduceDone done{};
};
size_t PipeScheduleInts(const BatchDims& d);
size_t PipeMaskInts(const BatchDims& d, const LdsPlan& plan);
void LaunchPipeMasks(const BatchDims& d, const DeviceBatch& b, const LdsPlan& plan, uint32_t* masks, hipStream_t stream);
// split_slots > 0: the batch is walked as two classes -- class A's trees (at most split_slots stored vectors and, when
// class_a is given, flagged there: a device array of one byte per tree) get their step tables for split_groups pattern
// groups per wave in layout split_layout, the others for `plan`
void LaunchPipePrepare(const BatchDims& d, const DeviceBatch& b, const LdsPlan& plan, hipStream_t stream,
                       bool beside_traversal, int split_slots, int split_groups = 4, int split_layout = kPipePlanAuto,
                       const uint8_t* class_a = nullptr);
// tree set-up + step tables + matrix images of a small batch in one launch (walk_pipe.hip, pipe_small_prepare_kernel):
// LaunchSetup and LaunchPipePrepare in one, the same bits
bool PipeSmallPrepareApplies(const BatchDims& d, const ModelSpec& spec);
void LaunchPipeSmallPrepare(const BatchDims& d, const ModelSpec& spec, const DeviceBatch& b, const LdsPlan& plan, hipStream_t stream,
                            int split_slots, int split_groups = 4, int split_layout = kPipePlanAuto,
                            const uint8_t* class_a = nullptr);
// deriv_mode 1: the edge derivatives use d r_c / d shape in place of r_c (site-model pass)
void LaunchWalkPipe(const BatchDims& d, const DeviceBatch& b, const LdsPlan& plan, int want_gradient, int want_site,
                    int deriv_mode, hipStream_t stream, const PipeClass& cls);

// LDS-resident traversal, second generation (walk_tree.hip): 8 waves per workgroup (two per
// SIMD), one group image per wave, the tree's P/dP images staged in LDS and shared.
struct TreePlan {
  int waves;            // 0 = not available for this batch
  int patterns_per_tile;
  int tiles;            // pattern tiles per tree
  int tiles_per_block;  // a workgroup serves this many consecutive tiles of one tree
  int blocks_per_tree;
  size_t lds_bytes;
};
TreePlan PlanTree(const BatchDims& d);
void LaunchWalkTree(const BatchDims& d, const DeviceBatch& b, const TreePlan& plan, int want_gradient,
                    hipStream_t stream);

// HBM-arena traversal: one thread per site pattern walks the whole tree.
constexpr int kHbmBlock = 256;
inline int HbmTiles(int pattern_count) { return (pattern_count + kHbmBlock - 1) / kHbmBlock; }
size_t HbmArenaBytesPerTree(const BatchDims& d);
// deriv_mode as in LaunchMatrices (1: the site-model pass, rates replaced by d r_c / d shape)
void LaunchWalkHbm(const BatchDims& d, const DeviceBatch& b, int tree0, int chunk_trees,
                   int want_gradient, int rescaling, hipStream_t stream, int deriv_mode = 0);
const char* WalkHbmKernelName(int category_count, int want_gradient, int rescaling);
// pattern tiles and partial gradient rows per tree of the HBM-arena kernel that LaunchWalkHbm picks for d
// (walk_hbm_cat_kernel, one wave per rate category, for up to 4 categories; walk_hbm_kernel otherwise)
int HbmWalkTiles(const BatchDims& d);
int HbmWalkGradRows(const BatchDims& d);
// walk_pipe_kernel: up to this many taxa it keeps (P, P^T) pairs; beyond, one image per branch and the reversible
// form of the pre-order recursion, which the engine uses only when every branch length is at least
// kPipeReversibleMinBranch (about 1e-6) (walk_pipe.hip, scripts/gen_walk_pipe.py)
constexpr int kPipeExactTaxa = 38;
// AUTO takes walk_pipe_kernel up to the size it is built for, 64 taxa (a VGPR per tip for its packed masks, 4n - 4 image
// registers).  Round 3 stopped at 58: beyond, most trees kept too many vectors for two pattern groups per wave.  With
// pitchforks folded into their parents' steps (round 4) nine 64-taxon trees in ten run with two: 1600 trees x 1000
// patterns, ms per pass, walk_pipe_kernel before / now / walk_hbm_cat_kernel: 56 taxa 3.56 / 2.62 / 3.91, 60: 4.40 / 2.82 /
// 4.20, 64: 4.68 / 3.16 / 4.50 (profiles/r4_midsize_fold.log)
constexpr int kPipeAutoTaxa = 64;
constexpr double kPipeReversibleRateScale = 0.2;  // smallest off-diagonal rate of the matrices that bound was measured on
constexpr double kPipeReversibleMinBranch = 9e-7;  // (just below exp(-13.9), the reference optimiser's own floor: src/dag_branch_handler.hpp:272)
bool HbmCatKernelApplies(const BatchDims& d);
// out_site from walk_hbm_cat_kernel's per-category gradient rows (after the walk of every chunk), no second traversal
void LaunchSiteFromCategoryRows(const BatchDims& d, const DeviceBatch& b, int rows, hipStream_t stream);
// the order in which walk_hbm_cat_kernel takes a tree's internal nodes, as step records in b.sched (HbmOrderInts int32)
size_t HbmOrderInts(const BatchDims& d);
void LaunchHbmOrder(const BatchDims& d, const DeviceBatch& b, hipStream_t stream);
// BITO_AMD_HBM_FOLD, read now: 0 cherries only, 1 (default) pitchforks as well, 2 four-tip subtrees too (walk_hbm_cat.hip)
int HbmFoldLevel();
void LaunchWalkHbmCat(const BatchDims& d, const DeviceBatch& b, int tree0, int chunk_trees, int want_gradient,
                      int rescaling, int deriv_mode, hipStream_t stream);

// grad_rows: partial gradient rows per tree (default: one per tile; the HBM-arena kernel writes one per wave)
// done: when the sums are the last kernel of a blocking call's chunk and go straight to pinned host memory, the
// workgroup that finishes last stores `ticket` to `flag` (pinned, polled by the host) behind everybody's results;
// `counter` is a zeroed int in device memory that the kernel leaves zeroed.
void LaunchReduce(const BatchDims& d, const DeviceBatch& b, int tiles, int want_gradient,
                  hipStream_t stream, int grad_rows = 0, ReduceDone done = ReduceDone{}, const uint8_t* skip = nullptr,
                  int first_tree = 0);
// one thread stores `value` to `flag` (pinned host memory) behind everything enqueued on the stream so far
void LaunchSignal(unsigned long long* flag, unsigned long long value, hipStream_t stream);

// ---- general-state-count path (gs_kernels.hip): the 61-state codon model, states padded to 64 ----
// Per-model record, doubles: V [64][64], V^-1 [64][64], Q [64][64], lambda [64], pi [64] (padding 0),
// sqrt(pi) [64], category rates / weights / d rate / d shape.  Trees with identical parameter rows
// share one record: model_index[t] = first tree with t's row.
constexpr int kGsV = 0, kGsVinv = 4096, kGsQ = 8192, kGsLambda = 12288, kGsPi = kGsLambda + 64, kGsSq = kGsPi + 64,
              kGsCatRate = kGsSq + 64, kGsCatWeight = kGsCatRate + kMaxCategories,
              kGsCatRateDeriv = kGsCatWeight + kMaxCategories, kGsQtImage = kGsCatRateDeriv + kMaxCategories,
              // kGsQtImage: Q^T as an MFMA A-operand image; behind it the nonzero entries of every COLUMN of Q in ascending
              // row order (a codon model's column has at most ten), for dP = P (r_c Q) on the vector ALU: count per column
              // (kGsQnzFlag: 1 when every column fits kGsQnzMax entries), row indices and values [64][kGsQnzMax]
              kGsQnzMax = 10, kGsQnzFlag = kGsQtImage + 4096, kGsQnzCount = kGsQnzFlag + 8, kGsQnzIdx = kGsQnzCount + 64,
              kGsQnzVal = kGsQnzIdx + 64 * kGsQnzMax, kGsModelStride = kGsQnzVal + 64 * kGsQnzMax;
inline int GsTiles(int pattern_count) { return (pattern_count + 15) / 16; }  // 16 site patterns per wave
size_t GsArenaDoublesPerTree(const BatchDims& d, int tiles, int want_gradient);
size_t GsImageDoublesPerTree(const BatchDims& d);
void LaunchGsSetup(const BatchDims& d, const ModelSpec& spec, const DeviceBatch& b, const int32_t* model_index,
                   double* gs_model, hipStream_t stream, bool models_stand = false);
// b.images is the chunk's record array [chunk][N-1][C][3][4096]; b.arena the chunk's PLV arena
void LaunchGsMatrices(const BatchDims& d, int S, int tree0, int chunk, const double* branch,
                      const int32_t* model_index, const double* gs_model, double* imgs, int want_gradient,
                      int deriv_mode, hipStream_t stream);
int GsScheduleStride(const BatchDims& d);  // int32 entries per tree of the image-order list (b.sched)
void LaunchGsSchedule(const BatchDims& d, const DeviceBatch& b, hipStream_t stream);
void LaunchGsWalk(const BatchDims& d, int S, const DeviceBatch& b, const int32_t* model_index,
                  const double* gs_model, int tree0, int chunk, int tiles, int want_gradient, int rescaling,
                  int deriv_mode, hipStream_t stream);


// time_tree.hip: RootedTree's height-ratio parameterisation and the rooted gradient transforms,
// one thread per tree (reference src/rooted_tree.cpp:36-121, src/rooted_gradient_transforms.cpp)
void LaunchTimeTreeFromBranchLengths(int T, int n, const int32_t* parent_ids, const double* branch_lengths,
                                     const double* tip_dates, double* bounds, double* heights, double* ratios,
                                     double* max_diff, hipStream_t stream);
void LaunchTimeTreeFromRatios(int T, int n, const int32_t* parent_ids, const double* bounds, const double* ratios,
                              double* heights, double* branch_lengths, hipStream_t stream);
void LaunchLogDetJacobian(int T, int n, const int32_t* parent_ids, const double* heights, const double* bounds,
                          double* out, double* add_to, hipStream_t stream);
// mode: 0 height gradient -> ratio gradient, 1 gradient of the log-det-Jacobian,
// 2 branch gradient -> ratio gradient (+4: add the log-det-Jacobian gradient); work [T][3(n-1)]
void LaunchRatioGradient(int T, int n, int mode, const int32_t* parent_ids, const double* heights,
                         const double* bounds, const double* ratios, const double* in, int in_stride,
                         const double* rates, double* work, double* out, hipStream_t stream);
void LaunchClockGradient(int T, int N, int rate_count, const double* branch_grad, const double* branch_lengths,
                         int bl_stride, double* out, hipStream_t stream);

}  // namespace bito_amd
