// walk_hbm_cat.hip -- the HBM-arena traversal with one wave per (64 patterns, rate category).
// Reference path: FatBeagle::LogLikelihoodInternals / Gradient (src/fat_beagle.cpp:253-373), the arithmetic of
// kernels.hip (walk_hbm_kernel) re-dealt over threads.  FP64, no atomics, fixed summation order.
#include "kernels.hpp"
#include "wave_sums.hpp"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

// model.hpp switches fused multiply-add contraction off for the set-up arithmetic (errors in P(t) are coherent across
// the site patterns; its operation order is part of the numerical contract).  The walk's own sums are per pattern,
// their rounding errors independent from pattern to pattern: contracted, a 4 x 4 product is 4 instructions per row
// instead of 7, and the walk spends more than half its time issuing them.
#pragma clang fp contract(fast)

namespace bito_amd {

__device__ __forceinline__ double WaveSum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ void MatVec(const double* __restrict__ M, const double x[4], double out[4]) {
#pragma unroll
  for (int i = 0; i < 4; i++)
    out[i] = M[i * 4 + 0] * x[0] + M[i * 4 + 1] * x[1] + M[i * 4 + 2] * x[2] + M[i * 4 + 3] * x[3];
}

__device__ __forceinline__ void MatVecT(const double* __restrict__ M, const double x[4], double out[4]) {
#pragma unroll
  for (int j = 0; j < 4; j++)
    out[j] = M[0 * 4 + j] * x[0] + M[1 * 4 + j] * x[1] + M[2 * 4 + j] * x[2] + M[3 * 4 + j] * x[3];
}

// --------------------------------------------------------------------------
// The same traversal with one WAVE per (64 patterns, rate category): walk_hbm_cat_kernel, up to 4 categories
// (with one category: walk_hbm_kernel's thread per pattern, in 64-thread workgroups, with this kernel's step).
//
// walk_hbm_kernel above is bound by memory latency at the occupancy its registers allow (a thread carries all
// categories of its pattern: 128 registers, four waves per SIMD, a load-wait round per category and step).
// Here a thread owns ONE category of one pattern: a workgroup is C waves over the same 64 patterns, wave c
// walks the tree in category c.  The matrices stay wave-uniform (scalar loads), a thread needs a third of the
// registers, and twice as many waves per SIMD have their loads in flight (eight until round 3, seven since the
// pitchforks are rebuilt in the step: HBM_CAT_WAVES).  Categories meet only twice:
//   * after the post-order pass, through LDS: site likelihood L_p = sum_c w_c s_c, and each thread's share
//     sigma_c = w_c s_c / L_p of it;
//   * never in the pre-order pass: d log L_p / dt_e = sum_c sigma_c num_c / den_c, where num_c and den_c are the
//     step's two sums IN category c (den_c is category c's site likelihood seen from that node; the ratio does
//     not depend on how the vectors of category c are scaled) -- so each wave sums w_p sigma_c num_c / den_c over
//     its lanes into its OWN gradient row and the final reduction adds the categories' rows.
// Rescaling is per (pattern, category) and by powers of two (exponent of the largest entry; exact, no
// logarithm, nothing stored): the log-likelihood takes the exponent sums of the C categories through
// max / ldexp, the pre-order partials are scaled the same way on the fly, and the ratio num_c / den_c never sees
// a factor.  BEAGLE's manual scaling divides by the maximum over ALL categories of a pattern
// (SURVEY A9); the log-likelihood and derivatives are the same numbers up to rounding.
// Arena: [tree][tile][node][category][state][64 patterns] -- a wave moves 512 contiguous bytes per access.
constexpr int kCatTile = 64;
inline int HbmCatTiles(int pattern_count) { return (pattern_count + kCatTile - 1) / kCatTile; }

// Tried on top of this kernel and dropped: a node's child pair requested one node ahead (the scalar load off the
// step's dependent chain: config 2 0.585 against 0.51-0.53 ms per 6400 trees, config 4 72.5 against 70.1 ms -- the
// request in flight turns the step's first wait for LDS into a wait for everything).  BOTH columns as pending vectors in
// the pre-order pass too, the next step's partial handed over in registers (6 % fewer vector transfers): round 3, at 64
// registers, it spilled (69.1 against 63.1 ms); round 4, on the folded walk at seven and six waves per SIMD (72 / 78
// registers), 49.4 / 47.5 ms against 46.6 -- the walk is no longer bound by the arena alone (below).
// Round 4, what bounds this walk (config 4, 125 trees per launch; profiles/r4_hbm/):
//   * hbm_pattern_bench.hip: independent waves that read and write 2 KB pieces at random places, in equal parts, move
//     5.0 TB/s on this part -- whatever the piece size (2 to 32 KB), the width of the accesses, the window a workgroup
//     stays in or the cache policy; reads alone 6.9, writes alone 5.5, a sequential copy 6.3.  Round 3's kernel moved its
//     256.9 GB at 4.7: at that ceiling.  Its arithmetic alone (HBM_EXP_NOLOAD / NOSTORE builds) takes 33 of the 55 ms, the
//     time does not change between four and eight waves per SIMD, and a build with a third fewer vector instructions
//     (seven waves: no spilled scalars) gains 3 %.
//   * So: fewer vectors.  PITCHFORKS (a tip and a cherry under one node -- a sixth of a random tree's internal nodes)
//     are rebuilt from their three tips' matrix rows where they are used, like cherries: no step, no cell.  Round 3 had
//     measured this at 2.5-4 % (the kernel was not at the memory's ceiling then, and the nested step spilled at 64
//     registers); at seven waves per SIMD (72 registers, 24 lane moves of scalars instead of 384): 1478 -> 1104 vector
//     transfers per tree (scripts/sim_hbm_traffic.py), 256.9 -> 193.5 GB per launch (PMC), 54.9 -> 46.6 ms, 64 taxa
//     4.57 -> 3.74 ms per 1600 trees, 100 taxa 7.23 -> 5.92.  The arithmetic alone is 30 ms now (fewer steps).
//   * Not the way: the same walk with a pattern's four states in four LANES and every 4 x 4 product on
//     v_mfma_f64_4x4x4_4b (parity-green at the first run; 72.5 ms against 54.9, 64 taxa 5.76 against 4.57).  On this part
//     the FP64 matrix pipe has the vector ALU's rate and does not run beside it (issue_bench p_mfma_mul64: a sibling
//     wave's v_mul_f64 gets one issue slot per matrix instruction), sums over states cost a whole product with a matrix
//     of ones, per-pattern scalars are four registers instead of one.  Four generic columns with owners and ages in
//     scalar registers instead of two hand-written ones: 74.1 ms (788 lane moves of spilled scalars).  Q fetched where
//     it is used instead of kept in 32 scalar registers for the loop (no spills left): 98.1 ms -- the step then waits
//     for two more scalar loads.  Half of the first round's workgroups started late (were the post-order phases of all
//     resident waves colliding?): no change.
// ---- visiting order -------------------------------------------------------------------------------------------
// The walk keeps the vector it has just computed in registers and ONE or TWO older ones in LDS columns (below); every
// other operand is read back from the arena.  How often that happens depends on the ORDER in which a node's two subtrees
// are visited: the one that needs more pending vectors first (Sethi-Ullman labelling of expression trees), so that the
// single vector that waits while the second subtree is walked is rarely displaced.  The node ids of the wire format are
// a post-order already, but with the children in id order: on BASELINE config 4's trees (1000 taxa) the walk in id order
// moves 1672 vectors per tree and pass, in this order 1478 (scripts/sim_hbm_traffic.py; the least possible, one store
// and one load of every stored vector, is 1329).  Cherries are no steps of their own (rebuilt where they are used) and
// do not appear in the order.
// order[tree] = NI + 1 records of sixteen int32: record 0 = {steps}, record 1 + k = step k = {node, child 0, child 1, node
// of step k - 1 (the step that follows in the pre-order pass) | children of child 0, children of child 1 (-1, -1 under a
// tip; a pitchfork: its tip, then its cherry) | the tips of a pitchfork child's cherry (else -1, -1) | the last two ids of a
// four-tip child (ChildInfo below; round 6)} -- everything a step must know about the topology in scalar loads whose
// address does not depend on an earlier load (the child lists are not read by the walk at all; the last quarter only by
// the steps that have a four-tip child).  One workgroup per tree, the tree's tables in LDS (34 bytes per internal
// node), one lane labels them, all write the records; trees too large for that are walked in id order.
constexpr int kStepInts = 16;
// Unstored nodes: cherries, and with `fold` PITCHFORKS -- a tip and a cherry under one node (a sixth of a random tree's
// internal nodes) --, rebuilt from their tips' matrix rows where they are used, like cherries: no step, no cell traffic.
__device__ __forceinline__ bool IsCherry(const int32_t* __restrict__ c, int n, int root, int v) {
  return v >= n && v != root && c[2 * (v - n)] < n && c[2 * (v - n) + 1] < n;
}
__device__ __forceinline__ bool IsFork(const int32_t* __restrict__ c, int n, int root, int v) {
  if (v < n || v == root) return false;
  const int a = c[2 * (v - n)], b = c[2 * (v - n) + 1];
  return (a < n && IsCherry(c, n, root, b)) || (b < n && IsCherry(c, n, root, a));
}
// Round 6, fold level 2: the two shapes of a FOUR-tip subtree are rebuilt where they are used as well -- a CATERPILLAR
// (a tip and a pitchfork under one node: 69 of a random 1000-taxon tree's 999 internal nodes) and TWIN cherries (two
// cherries under one node: 32), one per step -- stored vectors per tree 498 -> 402, vector transfers 1104 -> 888
// (scripts/sim_hbm_traffic.py).  A caterpillar is a pitchfork with one more tip on top: its pre-order part is one more
// level (two edge sums, one transposed product) in front of the pitchfork's, with no more vectors live at a time.
__device__ __forceinline__ bool IsCaterpillar(const int32_t* __restrict__ c, int n, int root, int v) {
  if (v < n || v == root) return false;
  const int a = c[2 * (v - n)], b = c[2 * (v - n) + 1];
  return (a < n && IsFork(c, n, root, b)) || (b < n && IsFork(c, n, root, a));
}
__device__ __forceinline__ bool IsTwin(const int32_t* __restrict__ c, int n, int root, int v) {
  if (v < n || v == root) return false;
  return IsCherry(c, n, root, c[2 * (v - n)]) && IsCherry(c, n, root, c[2 * (v - n) + 1]);
}
__device__ __forceinline__ bool IsFourTips(const int32_t* __restrict__ c, int n, int root, int v) {
  return IsCaterpillar(c, n, root, v) || IsTwin(c, n, root, v);
}
// ... folded into its parent's step unless its sibling is a four-tip subtree with a lower id: a step carries ONE of them,
// in its first slot (five of a random 1000-taxon tree's 101 have such a sibling); par: the parents by node id
__device__ __forceinline__ bool IsFoldedFour(const int32_t* __restrict__ c, const int32_t* __restrict__ par, int n, int root, int v) {
  if (par == nullptr || !IsFourTips(c, n, root, v)) return false;
  const int p = par[v];
  const int sib = c[2 * (p - n)] == v ? c[2 * (p - n) + 1] : c[2 * (p - n)];
  return !(sib < v && IsFourTips(c, n, root, sib));
}
// a node without a step and without a cell (fold 0: cherries; 1: and pitchforks; 2: and four-tip subtrees)
__device__ __forceinline__ bool IsUnstored(const int32_t* __restrict__ c, const int32_t* __restrict__ par, int n, int root, int fold, int v) {
  return IsCherry(c, n, root, v) || (fold >= 1 && IsFork(c, n, root, v)) || (fold >= 2 && IsFoldedFour(c, par, n, root, v));
}
// what a step knows about child cc -- six ids:
//   stored node / cherry: {its children}            pitchfork: {its tip, its cherry | the cherry's tips}
//   caterpillar: {its tip, its pitchfork F | F's tip, F's cherry H | H's tips}
//   twin cherries: {cherry H1, cherry H2 | H1's tips | H2's tips}
__device__ __forceinline__ void ChildInfo(const int32_t* __restrict__ c, const int32_t* __restrict__ par, int n, int root, int fold, int cc,
                                          int& a, int& b, int& hb, int& hc, int& x, int& y) {
  a = b = hb = hc = x = y = -1;
  if (cc < n) return;
  a = c[2 * (cc - n)];
  b = c[2 * (cc - n) + 1];
  const bool four = fold >= 2 && IsFoldedFour(c, par, n, root, cc);
  const bool fork = fold >= 1 && IsFork(c, n, root, cc), cat = four && IsCaterpillar(c, n, root, cc);
  if (fork || cat) {
    const int tip = a < n ? a : b, rest = a < n ? b : a;  // (the tip first)
    a = tip;
    b = rest;
    const int u = c[2 * (rest - n)], v = c[2 * (rest - n) + 1];
    hb = (fork || u < n) ? u : v;
    hc = (fork || u < n) ? v : u;
    if (cat) {
      x = c[2 * (hc - n)];
      y = c[2 * (hc - n) + 1];
    }
  } else if (four) {  // twin cherries
    hb = c[2 * (a - n)];
    hc = c[2 * (a - n) + 1];
    x = c[2 * (b - n)];
    y = c[2 * (b - n) + 1];
  }
}
__device__ __forceinline__ void WriteStep(int32_t* __restrict__ out, int k, int n, int root, int fold, int node, int next, int c0, int c1,
                                          const int32_t* __restrict__ c, const int32_t* __restrict__ par) {
  // (a folded four-tip child goes into the first slot -- there is at most one: the walk has its code there alone, and its
  // pre-order part parks a vector in the hand-over column, which a stored child that is processed next writes after it)
  if (fold >= 2 && IsFoldedFour(c, par, n, root, c1)) {
    const int t = c0;
    c0 = c1;
    c1 = t;
  }
  int4 lo, mid, hi, ex;
  lo.x = node; lo.y = c0; lo.z = c1; lo.w = next;
  ChildInfo(c, par, n, root, fold, c0, mid.x, mid.y, hi.x, hi.y, ex.x, ex.y);
  ChildInfo(c, par, n, root, fold, c1, mid.z, mid.w, hi.z, hi.w, ex.z, ex.w);
  int4* rec = reinterpret_cast<int4*>(out + (size_t)(1 + k) * kStepInts);
  // (a four-tip child is told by its third id, written as -2 - id: the walk reads the record's last quarter only in
  // the steps that have one)
  if (ex.x >= 0) hi.x = -2 - hi.x;
  if (ex.z >= 0) hi.z = -2 - hi.z;
  rec[0] = lo;
  rec[1] = mid;
  rec[2] = hi;
  rec[3] = ex;
}

__global__ void __launch_bounds__(64)
hbm_order_kernel(BatchDims d, const int32_t* __restrict__ children, int32_t* __restrict__ order, int in_lds, int fold) {
  extern __shared__ int32_t order_lds[];
  const int n = d.taxon_count, NI = n - 1, lane = threadIdx.x, root = n + NI - 1;
  const int32_t* __restrict__ ch = children + (size_t)blockIdx.x * NI * 2;
  int32_t* __restrict__ out = order + (size_t)blockIdx.x * (NI + 1) * kStepInts;
  if (!in_lds) {
    if (lane == 0) {
      // (a tree too large for the tables: id order, and no four-tip folding -- that rule asks for a node's parent)
      const int f = min(fold, 1);
      int k = 0, prev = -1;
      for (int v = 0; v < NI; v++) {
        if (IsUnstored(ch, nullptr, n, root, f, n + v)) continue;
        WriteStep(out, k++, n, root, f, n + v, prev, ch[2 * v], ch[2 * v + 1], ch, nullptr);
        prev = n + v;
      }
      out[0] = k;
    }
    return;
  }
  int32_t* c = order_lds;        // [NI][2] children
  int32_t* size = c + 2 * NI;    // steps in the subtree (0: an unstored node)
  int32_t* need = size + NI;     // pending vectors the subtree's walk needs at once
  int32_t* start = need + NI;    // position of the subtree's first step
  int32_t* at = start + NI;      // node of step k
  int32_t* par = at + NI;        // [n + NI] parent by node id
  int8_t* flip = reinterpret_cast<int8_t*>(par + n + NI);  // child 1's subtree is visited first
  int8_t* unstored = flip + NI;  // the node has no step (told by all lanes at once: the labelling below is one lane's)
  for (int i = lane; i < 2 * NI; i += 64) c[i] = ch[i];
  __syncthreads();
  for (int i = lane; i < 2 * NI; i += 64) par[c[i]] = n + (i >> 1);
  if (lane == 0) par[root] = -1;
  __syncthreads();
  for (int v = lane; v < NI; v += 64) unstored[v] = IsUnstored(c, par, n, root, fold, n + v);
  __syncthreads();
  if (lane == 0) {
    for (int v = 0; v < NI; v++) {  // ids ascend from the tips to the root
      const int c0 = c[2 * v], c1 = c[2 * v + 1];
      const int s0 = c0 >= n ? size[c0 - n] : 0, s1 = c1 >= n ? size[c1 - n] : 0;
      const int a = s0 ? need[c0 - n] : 0, b = s1 ? need[c1 - n] : 0;
      flip[v] = b > a;  // the heavier subtree first (ties: id order)
      if (unstored[v]) {
        size[v] = 0;
        need[v] = 0;
      } else {
        size[v] = s0 + s1 + 1;
        need[v] = (s0 == 0 || s1 == 0) ? max(max(a, b), 1) : (a == b ? a + 1 : max(a, b));
      }
    }
    start[NI - 1] = 0;
    for (int v = NI - 1; v >= 0; v--) {  // and back down: where each subtree's steps begin
      if (size[v] == 0) continue;
      const int st = start[v], f = c[2 * v + flip[v]], s = c[2 * v + 1 - flip[v]];
      int sf = 0;
      if (f >= n) {
        start[f - n] = st;
        sf = size[f - n];
      }
      if (s >= n) start[s - n] = st + sf;
      at[st + size[v] - 1] = n + v;
    }
    out[0] = size[NI - 1];
  }
  __syncthreads();
  const int steps = size[NI - 1];
  for (int k = lane; k < steps; k += 64) {
    const int node = at[k];
    WriteStep(out, k, n, root, fold, node, k > 0 ? at[k - 1] : -1, c[2 * (node - n)], c[2 * (node - n) + 1], c, par);
  }
}

inline size_t HbmOrderLdsBytes(const BatchDims& d) {
  return (size_t)(d.taxon_count - 1) * (6 * sizeof(int32_t) + 2) + (size_t)(2 * d.taxon_count - 1) * sizeof(int32_t) + 16;
}
size_t HbmOrderInts(const BatchDims& d) { return (size_t)d.tree_count * d.taxon_count * kStepInts; }

// BITO_AMD_HBM_FOLD: 0 a step and a cell for every pitchfork; 1 (the default) pitchforks rebuilt where they are used (round
// 4); 2 four-tip subtrees as well (round 6: built and held to the checker without GPU access -- it becomes the default
// when a device has run and timed it, scripts/gpu_round6.sh)
// (read once per pass by the worker, which hands the level to the order kernel's launch and to the walk's through
// DeviceBatch::hbm_fold: a process may compare the levels, and the walk is always the one its records were written for)
int HbmFoldLevel() {
  const char* e = getenv("BITO_AMD_HBM_FOLD");
  return e ? std::max(0, std::min(2, atoi(e))) : 1;
}

void LaunchHbmOrder(const BatchDims& d, const DeviceBatch& b, hipStream_t stream) {
  const size_t lds = HbmOrderLdsBytes(d);
  const bool in_lds = lds <= 150 * 1024;
  if (in_lds && lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hbm_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(hbm_order_kernel, dim3(d.tree_count), dim3(64), in_lds ? lds : 0, stream, d, b.children, b.sched, in_lds ? 1 : 0, b.hbm_fold);
}

#ifndef HBM_CAT_WAVES
#define HBM_CAT_WAVES 7
#endif

// Buffer accesses: a wave-uniform 128-bit descriptor (base in scalar registers), a wave-uniform byte offset, and
// ONE 32-bit lane offset in a vector register -- no 64-bit per-lane pointers (the compiler otherwise keeps one
// per stream, registers this kernel does not have at seven or eight waves per SIMD).
using BufferRsrc = __amdgpu_buffer_rsrc_t;
typedef unsigned UInt2 __attribute__((ext_vector_type(2)));
typedef unsigned UInt4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ BufferRsrc MakeRsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
// Cache policy of the arena accesses (auxiliary bits of the buffer instructions: 1 sc0, 2 nt, 16 sc1).  A vector that is
// written is read back a whole subtree later, or in the other pass: kept in L2 it only displaces what IS read again and
// again -- the tree's transition matrices, which every wave fetches with scalar loads the step then waits for.
// Non-temporal loads and stores: config 4 64.0 -> 59.7 ms per 125 trees, 100 taxa 8.03 -> 7.47 ms per 1600 (stores
// alone 62.6, loads alone 63.8; sc1 variants the same as nt within a percent).
#ifndef HBM_CAT_LOAD_AUX
#define HBM_CAT_LOAD_AUX 2
#endif
#ifndef HBM_CAT_STORE_AUX
#define HBM_CAT_STORE_AUX 2
#endif
__device__ __forceinline__ double BufLoad(BufferRsrc r, unsigned lane_bytes, unsigned uniform_bytes) {
  const UInt2 v = __builtin_amdgcn_raw_buffer_load_b64(r, lane_bytes, uniform_bytes, HBM_CAT_LOAD_AUX);
  return __hiloint2double(v.y, v.x);
}
__device__ __forceinline__ void BufStore(BufferRsrc r, unsigned lane_bytes, unsigned uniform_bytes, double x) {
  UInt2 v;
  v.x = __double2loint(x);
  v.y = __double2hiint(x);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, lane_bytes, uniform_bytes, HBM_CAT_STORE_AUX);
}
// four consecutive doubles (a row of a transposed matrix, picked per lane)
__device__ __forceinline__ void BufLoadRow(BufferRsrc r, unsigned lane_bytes, unsigned uniform_bytes, double out[4]) {
  const UInt4 a = __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, uniform_bytes, 0);
  const UInt4 b = __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes + 16, uniform_bytes, 0);
  out[0] = __hiloint2double(a.y, a.x);
  out[1] = __hiloint2double(a.w, a.z);
  out[2] = __hiloint2double(b.y, b.x);
  out[3] = __hiloint2double(b.w, b.z);
}

// A stored vector of a (node, category) is four rows of 64 patterns, 8 bytes per lane and access.  HBM_CAT_ROWS=0 builds
// the other layout -- per lane its four states side by side, two 16-byte accesses per lane and vector (1 KB per
// wave-instruction; 16 bytes per lane is what the memory system moves best, MI355X_MICROARCH.md).  Measured SLOWER
// with the gradient (round 4, same box): config 4 63.5 against 55.1 ms per 125 trees, 64 taxa 5.09 against 4.52 ms per
// 1600, log-likelihood only the same -- the pre-order step has no four consecutive registers to land a 16-byte load in
// at 64 VGPRs, and the four 8-byte requests of a vector were never the limit (the walk changes by under 2 % between
// five and eight waves per SIMD: scripts/build_hbm_cat_variants.sh w7 / w6 / w5).
#ifndef HBM_CAT_ROWS
#define HBM_CAT_ROWS 1
#endif
__device__ __forceinline__ void ArenaLoad(BufferRsrc r, unsigned lane, unsigned node_offset, double x[4]) {
#if HBM_EXP_NOLOAD  // (timing only: what the walk costs without its arena reads)
  for (int i = 0; i < 4; i++) x[i] = 0.25 + 0.125 * i;
  return;
#endif
#if HBM_CAT_ROWS
#pragma unroll
  for (int i = 0; i < 4; i++) x[i] = BufLoad(r, lane * 8, node_offset + i * 512);
#else
  const UInt4 a = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 32, node_offset, HBM_CAT_LOAD_AUX);
  const UInt4 b = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 32, node_offset + 16, HBM_CAT_LOAD_AUX);
  x[0] = __hiloint2double(a.y, a.x);
  x[1] = __hiloint2double(a.w, a.z);
  x[2] = __hiloint2double(b.y, b.x);
  x[3] = __hiloint2double(b.w, b.z);
#endif
}
__device__ __forceinline__ void ArenaStore(BufferRsrc r, unsigned lane, unsigned node_offset, const double x[4]) {
#if HBM_EXP_NOSTORE  // (timing only)
  return;
#endif
#if HBM_CAT_ROWS
#pragma unroll
  for (int i = 0; i < 4; i++) BufStore(r, lane * 8, node_offset + i * 512, x[i]);
#else
  UInt4 a, b;
  a.x = __double2loint(x[0]); a.y = __double2hiint(x[0]); a.z = __double2loint(x[1]); a.w = __double2hiint(x[1]);
  b.x = __double2loint(x[2]); b.y = __double2hiint(x[2]); b.z = __double2loint(x[3]); b.w = __double2hiint(x[3]);
  __builtin_amdgcn_raw_buffer_store_b128(a, r, lane * 32, node_offset, HBM_CAT_STORE_AUX);
  __builtin_amdgcn_raw_buffer_store_b128(b, r, lane * 32, node_offset + 16, HBM_CAT_STORE_AUX);
#endif
}

__device__ __forceinline__ void ScalePow2(double v[4], int& exponent_sum) {
  const double mx = fmax(fmax(v[0], v[1]), fmax(v[2], v[3]));
  const int ex = __builtin_amdgcn_frexp_exp(mx);  // 0 for mx == 0
#pragma unroll
  for (int i = 0; i < 4; i++) v[i] = __builtin_amdgcn_ldexp(v[i], -ex);
  exponent_sum += ex;
}

// FOUR: the walk for step records of fold level 2 (four-tip children: the code of the two shapes in a step's first slot,
// and with it the column addresses formed at their uses and gw parked in LDS -- what it takes to stay at 72 registers).
// FOUR = false is the walk of levels 0 and 1 as round 4's device runs had it: none of that in its step.
template <bool GRAD, bool RESCALE, bool FOUR>
__global__ void __launch_bounds__(256, HBM_CAT_WAVES)
walk_hbm_cat_kernel(BatchDims d, int tree0, const int32_t* __restrict__ children, const int32_t* __restrict__ order,
                    const double* __restrict__ all_mats, const TreeModel* __restrict__ models,
                    const uint8_t* __restrict__ tip_states, const double* __restrict__ weights,
                    double* __restrict__ arena_base, double* __restrict__ part_ll, double* __restrict__ part_grad,
                    int deriv_mode, int tile_count, int chunk, int by_xcd) {
  extern __shared__ double lds[];  // [threads][4] hand-over column | [threads][4] pending column | [threads] terms | [threads] exponents
  const int n = d.taxon_count, N = d.node_count, NI = n - 1, Ppad = d.pattern_stride, C = d.category_count;
  // Workgroups are dealt to the 8 XCDs round-robin by linear id (observed, not promised: a wrong guess costs speed
  // only).  A tree's tiles all read the same 4.6 KB-per-node table of transition matrices, wave by wave through the
  // scalar cache and L2: with id % 8 picking the tree among eight, each XCD's L2 holds the matrices of the one or two
  // trees its CUs are walking instead of those of every tree in flight (a dozen at config 4's size: 60 MB through
  // 4 MB).  The trees past the last full eight are dealt tile by tile.
  int tree_local, tile_id;
  {
    const int id = blockIdx.x, full = by_xcd ? (chunk & ~7) : 0;
    if (id < full * tile_count) {
      const int j = id >> 3;
      tree_local = (j / tile_count) * 8 + (id & 7);
      tile_id = j % tile_count;
    } else {
      const int rest = id - full * tile_count;
      tree_local = full + rest / tile_count;
      tile_id = rest % tile_count;
    }
  }
  const int tree = tree0 + tree_local;
  const int tid = threadIdx.x, lane = tid & 63, threads = blockDim.x;
  const int c = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = tile_id * kCatTile + lane;
  // the steps' records (hbm_order_kernel): {node, child 0, child 1, next | children of child 0, of child 1}
  const int4* __restrict__ ord = reinterpret_cast<const int4*>(order + (size_t)tree * (NI + 1) * kStepInts);
  const int steps = __builtin_amdgcn_readfirstlane(ord[0].x);
  const double* __restrict__ mats = all_mats + ((size_t)tree * (N - 1) * C + c) * kMatHot;  // + node * C * kMatHot
  const size_t node_mat = (size_t)C * kMatHot;
  const TreeModel* __restrict__ tm = models + tree;
  // every per-lane access: descriptor (wave-uniform base) + wave-uniform byte offset + this lane's 32-bit offset
  const BufferRsrc tips = MakeRsrc(tip_states + (size_t)tile_id * kCatTile);  // [taxon * Ppad][lane]
  // [(node - n) * C * 2 KB][state * 512 + lane * 8]
  const BufferRsrc arena = MakeRsrc(arena_base + (((size_t)tree_local * tile_count + tile_id) * NI * C + c) * 4 * kCatTile);
  const BufferRsrc matrows = MakeRsrc(mats);  // [node * C * kMatHot * 8][(kMatPT + state * 4) * 8]
  const unsigned node_bytes = (unsigned)C * 4 * kCatTile * 8, mat_bytes = (unsigned)C * kMatHot * 8;
  const unsigned ulane = lane;
  struct Child {
    int kind;  // 0 tip, 1 stored internal node, 2 cherry, 3 pitchfork, 4 caterpillar, 5 twin cherries
    int a, b;  // its children (a pitchfork, a caterpillar: its tip, then the rest)
    int hb, hc;  // a pitchfork's cherry's tips | a caterpillar's pitchfork's tip and cherry | the first twin's tips
    int st;  // states, a byte each: the tip's own | a cherry's two tips' | a pitchfork's tip's and its cherry's two
  };
  auto tip_state = [&](int tip) { return (int)__builtin_amdgcn_raw_buffer_load_b8(tips, ulane, (unsigned)tip * Ppad, 0); };
  // (a, b, hb, hc: out of the step's record; a four-tip subtree has hb < -1, and its last two ids and its states are
  // fetched where its rows are -- one child in ten is one, and whatever every step carried for them would cost the walk
  // registers it does not have at seven waves per SIMD)
  auto classify = [&](int cc, int a, int b, int hb, int hc) {
    Child ci{0, a, b, hb, hc, 0};
    if (cc < n) {
      ci.st = tip_state(cc);
    } else if (FOUR && hb < -1) {
      ci.kind = a < n ? 4 : 5;
      ci.hb = -2 - hb;
    } else if (hb >= 0) {
      ci.kind = 3;
      ci.st = tip_state(a) | (tip_state(hb) << 8) | (tip_state(hc) << 16);
    } else if (a < n && b < n) {
      ci.kind = 2;
      ci.st = tip_state(a) | (tip_state(b) << 8);
    } else {
      ci.kind = 1;
    }
    return ci;
  };
  int step_k = 0;  // the step at hand (its record's last quarter: a four-tip child's last two ids)
  auto last_ids = [&](int& x, int& y) {  // (of the first slot's child: the only one that can be a four-tip subtree)
    const int4 four = ord[7 + 4 * step_k];
    x = __builtin_amdgcn_readfirstlane(four.x);
    y = __builtin_amdgcn_readfirstlane(four.y);
  };
  const std::true_type kFirst{};
  const std::false_type kSecond{};
  // Two thread-private LDS columns of four doubles.  Pre-order: `fwd` hands a vector to the step that follows
  // immediately (the partial of the child that is processed next) and `pend` keeps ONE vector that is needed later -- the
  // partial of the child that is NOT processed next -- until its reader comes, unless a younger such vector takes the
  // column first (the walk is depth-first, so the youngest is needed soonest; the older one moves to the arena then).
  // Post-order: nothing is handed over through LDS (the step's result stays in registers), so BOTH columns keep
  // vectors that are needed later: the partial of a node whose parent is not the next step (also stored in the arena:
  // the pre-order pass reads it there), the oldest giving way.  Hits save the arena round trip of most of those vectors
  // (hbm_order_kernel's order is chosen for it).
  // (the columns' addresses are formed where a column is touched, from the thread's id through an empty asm the compiler
  // cannot look through: held in two registers from the kernel's start to its end they were the first values it
  // spilled to scratch once the four-tip steps were in the loop -- round 6)
  // (the thread's id is put together again from the wave's number and the lane: no register holds it through the loops)
  auto Tid = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return (c << 6) + l;
  };
  double* const fwd_held = lds + 4 * tid;  // (FOUR = false: the two addresses held in registers, as rounds 3 to 5 had them)
  double* const pend_held = lds + 4 * threads + 4 * tid;
  auto Fwd = [&]() -> double* {
    if constexpr (FOUR) return lds + 4 * Tid();
    else return fwd_held;
  };
  auto Pend = [&]() -> double* {
    if constexpr (FOUR) return lds + 4 * threads + 4 * Tid();
    else return pend_held;
  };
  int pend_owner = -1;  // node whose vector `pend` holds (wave-uniform)
  int fwd_owner = -1;   // post-order only: node whose vector `fwd` holds
  bool fwd_younger = false;  // post-order, both columns taken: `fwd` holds the younger vector
  // row `state` of the transposed transition matrix of the branch above `node` (a tip's message)
  auto tip_row = [&](int node, int state, double out[4]) {
    // (opaque: the row's offset is formed HERE, where the row is needed -- formed where the state is loaded, as the
    // compiler would have it, the step waits for a tip's byte from HBM before it has requested anything else)
    asm volatile("" : "+v"(state));
    BufLoadRow(matrows, (unsigned)(kMatPT * 8) + (unsigned)state * 32, (unsigned)node * mat_bytes, out);
  };
  auto fetch = [&](const Child& ci, int cc, auto first, double x[4]) {
    if (ci.kind == 2) {
      double ra[4], rb[4];
      tip_row(ci.a, ci.st & 255, ra);
      tip_row(ci.b, (ci.st >> 8) & 255, rb);
#pragma unroll
      for (int i = 0; i < 4; i++) x[i] = ra[i] * rb[i];
    } else if (ci.kind == 3) {  // a pitchfork's partial: a_tip . P_H (a_b . a_c)
      double ra[4], rb[4];
      tip_row(ci.hb, (ci.st >> 8) & 255, ra);
      tip_row(ci.hc, ci.st >> 16, rb);
#pragma unroll
      for (int i = 0; i < 4; i++) ra[i] *= rb[i];
      MatVec(mats + ci.b * node_mat + kMatP, ra, rb);
      tip_row(ci.a, ci.st & 255, ra);
#pragma unroll
      for (int i = 0; i < 4; i++) x[i] = ra[i] * rb[i];
    } else if (FOUR && decltype(first)::value && ci.kind == 4) {  // a caterpillar's partial: a_tip . P_F (a_b . P_H (a_c . a_d))
      double ra[4], rb[4];
      int tx, ty;
      last_ids(tx, ty);
      tip_row(tx, tip_state(tx), ra);
      tip_row(ty, tip_state(ty), rb);
#pragma unroll
      for (int i = 0; i < 4; i++) ra[i] *= rb[i];
      MatVec(mats + ci.hc * node_mat + kMatP, ra, rb);
      tip_row(ci.hb, tip_state(ci.hb), ra);
#pragma unroll
      for (int i = 0; i < 4; i++) ra[i] *= rb[i];
      MatVec(mats + ci.b * node_mat + kMatP, ra, rb);
      tip_row(ci.a, tip_state(ci.a), ra);
#pragma unroll
      for (int i = 0; i < 4; i++) x[i] = ra[i] * rb[i];
    } else if (FOUR && decltype(first)::value && ci.kind == 5) {  // twin cherries: P_H1 (a_a . a_b) . P_H2 (a_c . a_d)
      double ra[4], rb[4], m[4];
      tip_row(ci.hb, tip_state(ci.hb), ra);
      tip_row(ci.hc, tip_state(ci.hc), rb);
#pragma unroll
      for (int i = 0; i < 4; i++) ra[i] *= rb[i];
      MatVec(mats + ci.a * node_mat + kMatP, ra, m);
      int tx, ty;
      last_ids(tx, ty);
      tip_row(tx, tip_state(tx), ra);
      tip_row(ty, tip_state(ty), rb);
#pragma unroll
      for (int i = 0; i < 4; i++) ra[i] *= rb[i];
      MatVec(mats + ci.b * node_mat + kMatP, ra, rb);
#pragma unroll
      for (int i = 0; i < 4; i++) x[i] = m[i] * rb[i];
    } else if (cc == pend_owner) {
      {
        double* const pend = Pend();
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = pend[i];
      }
      pend_owner = -1;
    } else if (cc == fwd_owner) {
      {
        double* const fwd = Fwd();
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = fwd[i];
      }
      fwd_owner = -1;
    } else {
      ArenaLoad(arena, ulane, (unsigned)(cc - n) * node_bytes, x);
    }
  };
  // post-order: the vector of node `owner` (in v) into a column -- a free one, else the one with the older vector,
  // which log-likelihood-only walks (no copy in the arena yet) move to its cell first
  auto keep = [&](int owner, const double v[4]) {
    // (the owners are updated by selects, not inside the branches: stores to two variables that the optimiser can
    // merge into one store through a selected ADDRESS keep both variables in scratch memory)
    const bool into_fwd = __builtin_amdgcn_readfirstlane((int)(pend_owner >= 0 && (fwd_owner < 0 || !fwd_younger))) != 0;
    const int old = into_fwd ? fwd_owner : pend_owner;
    if (into_fwd) {
      if (!GRAD && old >= 0) {
        double o[4];
        {
          double* const fwd = Fwd();
#pragma unroll
          for (int i = 0; i < 4; i++) o[i] = fwd[i];
        }
        ArenaStore(arena, ulane, (unsigned)(old - n) * node_bytes, o);
      }
      {
        double* const fwd = Fwd();
#pragma unroll
        for (int i = 0; i < 4; i++) fwd[i] = v[i];
      }
    } else {
      if (!GRAD && old >= 0) {
        double o[4];
        {
          double* const pend = Pend();
#pragma unroll
          for (int i = 0; i < 4; i++) o[i] = pend[i];
        }
        ArenaStore(arena, ulane, (unsigned)(old - n) * node_bytes, o);
      }
      {
        double* const pend = Pend();
#pragma unroll
        for (int i = 0; i < 4; i++) pend[i] = v[i];
      }
    }
    fwd_owner = into_fwd ? owner : fwd_owner;
    pend_owner = into_fwd ? pend_owner : owner;
    fwd_younger = into_fwd;
  };

  // ---- post-order ------------------------------------------------------------
  int exponent_sum = 0;
  double site = 0.0;
  double dd[4];
  int last = -1;
  for (int k = 0; k < steps; ++k) {
    const int4 rec = ord[4 + 4 * k], sub = ord[5 + 4 * k], fork = ord[6 + 4 * k];
    step_k = k;
    const int node = __builtin_amdgcn_readfirstlane(rec.x);
    const int c0 = __builtin_amdgcn_readfirstlane(rec.y), c1 = __builtin_amdgcn_readfirstlane(rec.z);
    const Child k0 = classify(c0, __builtin_amdgcn_readfirstlane(sub.x), __builtin_amdgcn_readfirstlane(sub.y),
                              __builtin_amdgcn_readfirstlane(fork.x), __builtin_amdgcn_readfirstlane(fork.y));
    const Child k1 = classify(c1, __builtin_amdgcn_readfirstlane(sub.z), __builtin_amdgcn_readfirstlane(sub.w),
                              __builtin_amdgcn_readfirstlane(fork.z), __builtin_amdgcn_readfirstlane(fork.w));
    // the previous step's vector, when its parent comes later: keep a copy at hand (log-likelihood only: the column
    // is the partial's only home until a younger one needs it)
    if (last >= 0 && c0 != last && c1 != last) keep(last, dd);
    double A[4], B[4];
    const double* m0 = mats + c0 * node_mat;
    const double* m1 = mats + c1 * node_mat;
    if (k0.kind == 0) {
      tip_row(c0, k0.st, A);
    } else {
      double x[4];
      if (c0 == last) {
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = dd[i];
      } else {
        fetch(k0, c0, kFirst, x);
      }
      MatVec(m0 + kMatP, x, A);
    }
    if (k1.kind == 0) {
      tip_row(c1, k1.st, B);
    } else {
      double x[4];
      if (c1 == last) {
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = dd[i];
      } else {
        fetch(k1, c1, kSecond, x);
      }
      MatVec(m1 + kMatP, x, B);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) dd[i] = A[i] * B[i];
    last = node;
    if (RESCALE) ScalePow2(dd, exponent_sum);
    if (node == N - 1) {
      site = tm->cat_weight[c] * (tm->pi[0] * dd[0] + tm->pi[1] * dd[1] + tm->pi[2] * dd[2] + tm->pi[3] * dd[3]);
    } else if (GRAD) {
      ArenaStore(arena, ulane, (unsigned)(node - n) * node_bytes, dd);
    }
  }
  // ---- the categories of a pattern meet: L_p and this category's share of it ---
  double* __restrict__ terms = lds + 8 * threads;                        // [C][64]
  int* __restrict__ exps = reinterpret_cast<int*>(lds + 9 * threads);    // [C][64]
  terms[tid] = site;
  exps[tid] = exponent_sum;
  __syncthreads();
  int emax = exps[lane];
  for (int k = 1; k < C; k++) emax = max(emax, exps[k * 64 + lane]);
  double total = 0.0;
  for (int k = 0; k < C; k++) total += RESCALE ? __builtin_amdgcn_ldexp(terms[k * 64 + lane], exps[k * 64 + lane] - emax) : terms[k * 64 + lane];
  const double mine = RESCALE ? __builtin_amdgcn_ldexp(site, exponent_sum - emax) : site;
  const double weight = weights[p];  // (loaded here: two registers the post-order loop does not have to carry)
  const double ll = weight * (log(total) + (RESCALE ? emax * 0.693147180559945309417232121458 : 0.0));

  if (c == 0) {
    const double wll = WaveSum(ll);
    if (lane == 0) part_ll[(size_t)tree * tile_count + tile_id] = wll;
  }
  // ---- pre-order + edge derivatives -----------------------------------------
  // Messages: a_k = P_k x_k.  With u the pre-order partial of the node,
  //     den = sum_i u_i a0_i a1_i,   num_0 = sum_i u_i a1_i (Q a0)_i   (dP x = r_c Q P x: Q and P commute; the
  //     factor r_c, or d r_c / d shape for the site-model pass, multiplies the wave's weight once),
  //     pre(child 0) = P_0^T (u . a1).
  // The step is written child by child so that few vectors are live at a time (seven waves per SIMD: 72 registers).
  if (GRAD) {
    const double rate = deriv_mode ? tm->cat_rate_deriv[c] : tm->cat_rate[c];
    // rescaled: w_p sigma_c r_c (then times num_c / den_c per edge); plain: w_p w_c r_c / L_p (then times num_c)
    // Without rescaling the whole pre-order pass of a pattern is scaled by the power of two that brings L_p to [0.5, 1)
    // (through the root's partial, below; exact, and w_p w_c r_c / L_p is formed with the scaled L_p): on trees of
    // hundreds of taxa L_p sits near the bottom of the double range -- where the reference still returns finite
    // derivatives -- and 1 / L_p alone is infinite, the edge sums denormal or zero.  L_p = 0: no shift, and the
    // derivatives are non-finite as the reference's.
    const int shift = RESCALE ? 0 : min(-__builtin_amdgcn_frexp_exp(total), 1000);
    const double gw = RESCALE ? weight * (mine / total) * rate
                              : weight * (tm->cat_weight[c] / __builtin_amdgcn_ldexp(total, shift)) * rate;
    // (gw waits in the thread's slot of `terms`, free since the categories met: read once per step, it is two registers
    // the walk does not carry through its longest stretches)
    if constexpr (FOUR) {
      __syncthreads();  // (every wave has read the categories' terms)
      terms[Tid()] = gw;
    }
    double* __restrict__ my_row = part_grad + (((size_t)tree * tile_count + tile_id) * C + c) * N;
    const double* __restrict__ Q = tm->Q;
    bool u_forwarded = false;
    // (what the post-order pass left in the columns is in the arena as well.)  The root's partial -- the stationary
    // frequencies -- waits for the first step in the column like any pending vector
    fwd_owner = -1;
    pend_owner = N - 1;
    {
      double* const pend = Pend();
#pragma unroll
      for (int i = 0; i < 4; i++) pend[i] = __builtin_amdgcn_ldexp(tm->pi[i], shift);
    }
    // message of a child: tip -> its row of P^T; stored -> P x; cherry -> P (a_a . a_b)
    auto message = [&](const Child& ci, int cc, auto first, double A[4]) {
      const double* m = mats + cc * node_mat;
      if (ci.kind == 0) {
        tip_row(cc, ci.st, A);
      } else {
        double x[4];
        fetch(ci, cc, first, x);
        MatVec(m + kMatP, x, A);
      }
    };
    // the two tip edges of a cherry child whose pre-order partial is q
    // (states travel packed, a byte each, and are taken apart where a row is asked for: one register, not one per tip)
    auto cherry_edges = [&](int ta, int tb, int states, const double q[4], double rden) {
      double aa[4], ab[4], qa[4];
      tip_row(ta, states & 255, aa);
      tip_row(tb, (states >> 8) & 255, ab);
      MatVec(Q, aa, qa);
      double sa = 0.0, sb = 0.0;
#pragma unroll
      for (int i = 0; i < 4; i++) sa += q[i] * (ab[i] * qa[i]);
      MatVec(Q, ab, qa);
#pragma unroll
      for (int i = 0; i < 4; i++) sb += q[i] * (aa[i] * qa[i]);
      const double g = PairSum(sa * rden, sb * rden);
      if ((lane & 31) == 31) my_row[lane < 32 ? ta : tb] = g;
    };
    // a child's edge term: sum_i (u . a_sibling)_i (Q a)_i, times w_p sigma_c r_c / den
    auto edge_term = [&](const double A[4], const double UAs[4], double rden) {
      double qa[4];
      MatVec(Q, A, qa);
      return (UAs[0] * qa[0] + UAs[1] * qa[1] + UAs[2] * qa[2] + UAs[3] * qa[3]) * rden;
    };
    // a pitchfork whose pre-order partial is q: tip ta and cherry H (tips tb, tc) under it.  With a_a the tip's row and
    // m = P_H (a_b . a_c):
    //   edge of a:  sum_i (q . m)_i (Q a_a)_i      edge of H:  sum_i (q . a_a)_i (Q m)_i      partial of H:  P_H^T (q . a_a)
    auto fork_edges = [&](int ta, int H, int tb, int tc, int states, const double q[4], double rden) {
      double ra[4], m[4], t[4];
      {
        double rb[4];
        tip_row(tb, (states >> 8) & 255, t);
        tip_row(tc, states >> 16, rb);
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] *= rb[i];
      }
      MatVec(mats + H * node_mat + kMatP, t, m);
      tip_row(ta, states & 255, ra);
      MatVec(Q, ra, t);
      double ea = 0.0, eh = 0.0;
#pragma unroll
      for (int i = 0; i < 4; i++) ea += (q[i] * m[i]) * t[i];
      MatVec(Q, m, t);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        ra[i] *= q[i];  // q . a_a
        eh += ra[i] * t[i];
      }
      const double g = PairSum(ea * rden, eh * rden);
      if ((lane & 31) == 31) my_row[lane < 32 ? ta : H] = g;
      MatVecT(mats + H * node_mat + kMatP, ra, t);
      cherry_edges(tb, tc, states >> 8, t, rden);
    };
    // an internal child's pre-order partial P^T (u . a_sibling): stored in place (the child's cell held its
    // post-order partial, read for this step's message), handed to the next step through the thread's LDS
    // column when the child is processed next, or consumed here by a cherry's two tip edges
    auto pre_part = [&](const Child& ci, int cc, auto first, const double UAs[4], double rden, bool forward) {
      if (ci.kind == 0) return;
      double q[4];
      MatVecT(mats + cc * node_mat + kMatP, UAs, q);
      if (ci.kind == 2) {
        cherry_edges(ci.a, ci.b, ci.st, q, rden);
        return;
      }
      if (ci.kind == 3) {
        fork_edges(ci.a, ci.b, ci.hb, ci.hc, ci.st, q, rden);
        return;
      }
      if (FOUR && decltype(first)::value && ci.kind == 4) {
        // caterpillar: tip a and pitchfork F = ci.b (tip ci.hb, cherry H = ci.hc with tips x, y) under cc.  With a_a the
        // tip's row and m = P_F (a_b . P_H (a_c . a_d)):
        //   edge of a:  sum_i (q . m)_i (Q a_a)_i      edge of F:  sum_i (q . a_a)_i (Q m)_i      partial of F:  P_F^T (q . a_a)
        // and below F the pitchfork's own part (which rebuilds H's message: nothing is kept across the two levels)
        double ra[4], m[4], t[4];
        int tx, ty;
        last_ids(tx, ty);
        {
          double rb[4];
          tip_row(tx, tip_state(tx), t);
          tip_row(ty, tip_state(ty), rb);
#pragma unroll
          for (int i = 0; i < 4; i++) t[i] *= rb[i];
          MatVec(mats + ci.hc * node_mat + kMatP, t, rb);
          tip_row(ci.hb, tip_state(ci.hb), t);
#pragma unroll
          for (int i = 0; i < 4; i++) t[i] *= rb[i];
        }
        MatVec(mats + ci.b * node_mat + kMatP, t, m);
        tip_row(ci.a, tip_state(ci.a), ra);
        MatVec(Q, ra, t);
        double ea = 0.0, ef = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) ea += (q[i] * m[i]) * t[i];
        MatVec(Q, m, t);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          ra[i] *= q[i];  // q . a_a
          ef += ra[i] * t[i];
        }
        const double g = PairSum(ea * rden, ef * rden);
        if ((lane & 31) == 31) my_row[lane < 32 ? ci.a : ci.b] = g;
        MatVecT(mats + ci.b * node_mat + kMatP, ra, t);
        __builtin_amdgcn_sched_barrier(0);
        fork_edges(ci.hb, ci.hc, tx, ty, tip_state(ci.hb) | (tip_state(tx) << 8) | (tip_state(ty) << 16), t, rden);
        return;
      }
      if (FOUR && decltype(first)::value && ci.kind == 5) {
        // twin cherries H1 = ci.a (tips hb, hc) and H2 = ci.b (tips x, y) under cc.  With m_k = P_Hk (the cherry's rows):
        //   edge of H1:  sum_i (q . m2)_i (Q m1)_i      edge of H2:  sum_i (q . m1)_i (Q m2)_i
        //   partial of H1:  P_H1^T (q . m2)             partial of H2:  P_H2^T (q . m1)
        double m1[4], m2[4], t[4];
        int tx, ty;
        last_ids(tx, ty);
        {
          double rb[4];
          tip_row(ci.hb, tip_state(ci.hb), t);
          tip_row(ci.hc, tip_state(ci.hc), rb);
#pragma unroll
          for (int i = 0; i < 4; i++) t[i] *= rb[i];
          MatVec(mats + ci.a * node_mat + kMatP, t, m1);
          tip_row(tx, tip_state(tx), t);
          tip_row(ty, tip_state(ty), rb);
#pragma unroll
          for (int i = 0; i < 4; i++) t[i] *= rb[i];
          MatVec(mats + ci.b * node_mat + kMatP, t, m2);
        }
        MatVec(Q, m1, t);
        double e1 = 0.0, e2 = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) e1 += (q[i] * m2[i]) * t[i];
        MatVec(Q, m2, t);
#pragma unroll
        for (int i = 0; i < 4; i++) e2 += (q[i] * m1[i]) * t[i];
        const double g = PairSum(e1 * rden, e2 * rden);
        if ((lane & 31) == 31) my_row[lane < 32 ? ci.a : ci.b] = g;
        // (q . m1, what H2's partial is made from, waits in the hand-over column while H1's tips take their edges: the
        // column is free here -- hbm_order_kernel puts a four-tip child in front of a stored one -- and with it the
        // step has no more vectors live at a time than a cherry child's)
        {
          double* const fwd = Fwd();
#pragma unroll
          for (int i = 0; i < 4; i++) {
            fwd[i] = q[i] * m1[i];
            m2[i] *= q[i];  // q . m2: what H1's partial is made from
          }
        }
        MatVecT(mats + ci.a * node_mat + kMatP, m2, t);
        __builtin_amdgcn_sched_barrier(0);
        cherry_edges(ci.hb, ci.hc, tip_state(ci.hb) | (tip_state(ci.hc) << 8), t, rden);
        __builtin_amdgcn_sched_barrier(0);
        {
          double* const fwd = Fwd();
#pragma unroll
          for (int i = 0; i < 4; i++) m2[i] = fwd[i];
        }
        MatVecT(mats + ci.b * node_mat + kMatP, m2, t);
        cherry_edges(tx, ty, tip_state(tx) | (tip_state(ty) << 8), t, rden);
        return;
      }
      if (RESCALE) { int unused = 0; ScalePow2(q, unused); }
      if (forward) {
        {
          double* const fwd = Fwd();
#pragma unroll
          for (int i = 0; i < 4; i++) fwd[i] = q[i];
        }
        return;
      }
      if (pend_owner >= 0) {  // an older vector waits in the column: it moves to its cell in the arena
        double o[4];
        {
          double* const pend = Pend();
#pragma unroll
          for (int i = 0; i < 4; i++) o[i] = pend[i];
        }
        ArenaStore(arena, ulane, (unsigned)(pend_owner - n) * node_bytes, o);
      }
      {
        double* const pend = Pend();
#pragma unroll
        for (int i = 0; i < 4; i++) pend[i] = q[i];
      }
      pend_owner = cc;
    };
    for (int k = steps - 1; k >= 0; --k) {
      const int4 rec = ord[4 + 4 * k], sub = ord[5 + 4 * k], fork = ord[6 + 4 * k];
      step_k = k;
      const int node = __builtin_amdgcn_readfirstlane(rec.x);
      const int c0 = __builtin_amdgcn_readfirstlane(rec.y), c1 = __builtin_amdgcn_readfirstlane(rec.z);
      const int next = __builtin_amdgcn_readfirstlane(rec.w);  // the step that follows
      const Child k0 = classify(c0, __builtin_amdgcn_readfirstlane(sub.x), __builtin_amdgcn_readfirstlane(sub.y),
                                __builtin_amdgcn_readfirstlane(fork.x), __builtin_amdgcn_readfirstlane(fork.y));
      const Child k1 = classify(c1, __builtin_amdgcn_readfirstlane(sub.z), __builtin_amdgcn_readfirstlane(sub.w),
                                __builtin_amdgcn_readfirstlane(fork.z), __builtin_amdgcn_readfirstlane(fork.w));
      double U[4], A0[4], A1[4];
      if (u_forwarded) {
        {
          double* const fwd = Fwd();
#pragma unroll
          for (int i = 0; i < 4; i++) U[i] = fwd[i];
        }
      } else if (node == pend_owner) {
        {
          double* const pend = Pend();
#pragma unroll
          for (int i = 0; i < 4; i++) U[i] = pend[i];
        }
        pend_owner = -1;
      } else {
        ArenaLoad(arena, ulane, (unsigned)(node - n) * node_bytes, U);
      }
      message(k0, c0, kFirst, A0);
      message(k1, c1, kSecond, A1);
      const bool fwd0 = k0.kind == 1 && c0 == next;
      const bool fwd1 = k1.kind == 1 && c1 == next;
      double UA1[4], UA0[4];
#pragma unroll
      for (int i = 0; i < 4; i++) UA1[i] = U[i] * A1[i];
      // Rescaled vectors carry unknown powers of two, the same in num_c and den_c: the ratio form, gw / den by
      // reciprocal and two Newton steps (relative error ~1e-16; the full division's special-case handling costs
      // a dozen registers the step does not have at seven waves per SIMD).  Without rescaling the numerator is
      // the reference's own term and w_p w_c r_c / L_p multiplies it directly: no division, and a category that
      // has underflowed (large tree, short branches: the slow categories go first) contributes its zero instead
      // of 0/0, while a pattern whose likelihood is zero still turns the tree's derivatives non-finite as the
      // reference's do.
      double rden = gw;
      if constexpr (FOUR) rden = terms[Tid()];  // (= gw, parked above the loop)
      if (RESCALE) {
        const double den = UA1[0] * A0[0] + UA1[1] * A0[1] + UA1[2] * A0[2] + UA1[3] * A0[3];
        double r = __builtin_amdgcn_rcp(den);
        r = fma(fma(-den, r, 1.0), r, r);
        r = fma(fma(-den, r, 1.0), r, r);
        rden = rden * r;
      }
      // both edge sums first: after them only u . a1 and u . a0 are live
      const double e0 = edge_term(A0, UA1, rden);
#pragma unroll
      for (int i = 0; i < 4; i++) UA0[i] = U[i] * A0[i];
      const double e1 = edge_term(A1, UA0, rden);
      const double g = PairSum(e0, e1);
      if ((lane & 31) == 31) my_row[lane < 32 ? c0 : c1] = g;
      __builtin_amdgcn_sched_barrier(0);
      pre_part(k0, c0, kFirst, UA1, rden, fwd0);
      __builtin_amdgcn_sched_barrier(0);
      pre_part(k1, c1, kSecond, UA0, rden, fwd1);
      __builtin_amdgcn_sched_barrier(0);
      u_forwarded = fwd0 || fwd1;
    }
  }
}

// The site-model gradient without a second traversal (DiscreteSiteModelGradient, fat_beagle.cpp:401-410,538-550:
// the edge sums with r_c replaced by d r_c / d shape, times the branch lengths).  walk_hbm_cat_kernel's partial
// gradient rows are per rate category -- row (tile, c) holds sum_p w_p sigma_c r_c num_c / den_c per edge -- so it is
// sum over rows and edges of (d r_c / d shape) / r_c * row[e] * t_e.  One workgroup per tree, fixed order.
__global__ void __launch_bounds__(256)
site_from_category_rows_kernel(BatchDims d, DeviceBatch b, int rows) {
  __shared__ double partial[256];
  const int N = d.node_count, C = d.category_count, tree = blockIdx.x, tid = threadIdx.x;
  const TreeModel* __restrict__ tm = b.model + tree;
  const double* __restrict__ mine = b.part_grad + (size_t)tree * rows * N;
  const double* __restrict__ t_e = b.branch + (size_t)tree * N;
  const int edges = d.rooted ? N - 1 : N - 2;  // (the root has no branch; unrooted: nor has the node that took its id)
  double acc = 0.0;
  for (int row = 0; row < rows; row++) {
    const int c = row % C;
    const double ratio = tm->cat_rate_deriv[c] / tm->cat_rate[c];
    double s = 0.0;
    for (int e = tid; e < edges; e += 256) s += mine[(size_t)row * N + e] * t_e[e];
    acc += ratio * s;
  }
  partial[tid] = acc;
  __syncthreads();
  for (int half = 128; half > 0; half >>= 1) {
    if (tid < half) partial[tid] += partial[tid + half];
    __syncthreads();
  }
  if (tid == 0) b.out_site[tree] = partial[0];
}

void LaunchSiteFromCategoryRows(const BatchDims& d, const DeviceBatch& b, int rows, hipStream_t stream) {
  hipLaunchKernelGGL(site_from_category_rows_kernel, dim3(d.tree_count), dim3(256), 0, stream, d, b, rows);
}

bool HbmCatKernelApplies(const BatchDims& d) {
  static const bool classic = [] { const char* v = getenv("BITO_AMD_HBM_CLASSIC"); return v && v[0] == '1'; }();
  return !classic && d.category_count <= 4;
}
int HbmWalkTiles(const BatchDims& d) { return HbmCatKernelApplies(d) ? HbmCatTiles(d.pattern_count) : HbmTiles(d.pattern_count); }
int HbmWalkGradRows(const BatchDims& d) {
  return HbmCatKernelApplies(d) ? HbmCatTiles(d.pattern_count) * d.category_count : HbmTiles(d.pattern_count) * (kHbmBlock / 64);
}

void LaunchWalkHbmCat(const BatchDims& d, const DeviceBatch& b, int tree0, int chunk, int want_gradient,
                      int rescaling, int deriv_mode, hipStream_t stream) {
  const int threads = 64 * d.category_count;
  const int tiles = HbmCatTiles(d.pattern_count);
  const dim3 grid((unsigned)tiles * (unsigned)chunk), block(threads);
  const size_t lds = (size_t)threads * (8 * sizeof(double) + sizeof(double) + sizeof(int));
  static const int by_xcd = [] { const char* v = getenv("BITO_AMD_HBM_BY_XCD"); return v ? atoi(v) : 1; }();
#define BITO_CAT(G, R, F) hipLaunchKernelGGL((walk_hbm_cat_kernel<G, R, F>), grid, block, lds, stream, d, tree0, b.children, b.sched, b.mats, b.model, \
                                             b.tip_states, b.weights, b.arena, b.part_ll, b.part_grad, deriv_mode, tiles, chunk, by_xcd)
  if (b.hbm_fold >= 2) {  // (the step records hold four-tip children)
    if (want_gradient) { if (rescaling) BITO_CAT(true, true, true); else BITO_CAT(true, false, true); }
    else { if (rescaling) BITO_CAT(false, true, true); else BITO_CAT(false, false, true); }
  } else {
    if (want_gradient) { if (rescaling) BITO_CAT(true, true, false); else BITO_CAT(true, false, false); }
    else { if (rescaling) BITO_CAT(false, true, false); else BITO_CAT(false, false, false); }
  }
#undef BITO_CAT
}

}  // namespace bito_amd
