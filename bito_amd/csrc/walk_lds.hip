// walk_lds.hip -- LDS-resident traversal on the FP64 matrix cores (gfx950).
//
// Why this shape.  Measured on MI355X (bito_amd/csrc/microbench.hip, profiles/):
// v_mfma_f64_4x4x4_4b sustains 87 % of the FP64 peak from ONE wave per SIMD, while
// v_fma_f64 needs >= 2 waves per SIMD for 77 %; and a tree's PLVs only fit in LDS for
// ~3 waves' worth of patterns per CU.  So: one wave per SIMD, all FP64 contractions
// on the 4x4x4 (4 blocks) MFMA whose four blocks are the four rate categories:
//
//   D_b[i][j] = sum_k A_b[i][k] B_b[k][j]        b = category, i,k = states, j = pattern
//
// A "group image" is 64 doubles, one per lane: lane = 16*state + 4*block + pattern.
// The D layout of the instruction equals its B layout, so a child message P x, the
// Hadamard product with the sibling's, and the result being the next step's B operand
// never leave that lane layout.  (For C = 1 or 2 categories the spare block bits hold
// more patterns.)  Transition matrices arrive as precomputed A-operand images
// (LaunchMatrixImages), one 8-byte coalesced load per lane per matrix.
//
// Each wave owns G groups (G*16/C patterns) and a private LDS region holding the n-2
// stored PLVs of those patterns; waves never exchange data until the final sums, so
// the walk has no barriers.  Tips are fed to the MFMA as one-hot B operands built from
// the state byte.  Loads for step k+1 (child list, matrix images, LDS operands, tip
// states) are issued before the arithmetic of step k; a result needed by the very next
// step is forwarded in registers.
//
// Arithmetic per step is the one documented in kernels.hip (walk_hbm_kernel): the
// pre-order pass yields both child-edge derivatives and both child pre-order partials
// per internal node, the child's pre-order partial overwriting its post-order partial.
// The site likelihood L_p = sum_c w_c pi^T root_c is computed once per pattern after the
// post-order pass; every edge derivative is then sum_p (w_p / L_p) sum_c w_c (...).
// Not available here: rescaling (small trees do not need it; the engine routes
// rescaling requests to the HBM-arena kernel).
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"

#ifndef LDS_SYNC_FETCH
#define LDS_SYNC_FETCH 1  // 0: descriptors in flight across compiler-visible code (the fault of round 2, see StepFetch)
#endif

namespace bito_amd {

// --------------------------------------------------------------------------
// A-operand images: one wave per (tree, branch); lane = 16 k + 4 b + i.

__global__ void __launch_bounds__(256)
matrix_images_kernel(BatchDims d, DeviceBatch b, int want_gradient, int deriv_mode) {
#pragma clang fp contract(off)  // same operation order as transition_matrices_kernel
  const int C = d.category_count, NB = d.node_count - 1;
  const int lane = threadIdx.x & 63;
  const size_t unit = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= (size_t)d.tree_count * NB) return;
  const int br = (int)(unit % NB);
  const int t = (int)(unit / NB);
  const int k = lane >> 4, blk = (lane >> 2) & 3, i = lane & 3;
  const int c = blk % C;
  const TreeModel* __restrict__ m = b.model + t;
  const double rate = m->cat_rate[c];
  const double time = b.branch[(size_t)t * d.node_count + br] * rate;
  double p = 0;  // P_c[i][k]
#pragma unroll
  for (int q = 0; q < 4; q++) p += m->V[i * 4 + q] * exp(m->lambda[q] * time) * m->Vinv[q * 4 + k];
  double* out = b.images + unit * kImgStride;
  out[2 * lane] = p;  // P and dP interleaved per lane: the pre-order pass fetches both with one 16-byte load
  out[kImgPT + 16 * i + 4 * blk + k] = p;  // image of P^T: lane 16 k' + 4 b + i' holds P[k'][i']
  if (want_gradient) {
    // dP_c[i][k] = sum_q P_c[i][q] (Q[q][k] r_c); P_c[i][q] lives in lane 16 q + 4 b + i.
    double dp = 0;
    const double drate = deriv_mode ? m->cat_rate_deriv[c] : rate;  // site-model pass: d r_c / d shape
#pragma unroll
    for (int q = 0; q < 4; q++) dp += __shfl(p, 16 * q + (lane & 15)) * (m->Q[q * 4 + k] * drate);
    out[2 * lane + 1] = dp;
  }
}

void LaunchMatrixImages(const BatchDims& d, const DeviceBatch& b, int want_gradient, int deriv_mode,
                        hipStream_t stream) {
  const size_t units = (size_t)d.tree_count * (d.node_count - 1);
  hipLaunchKernelGGL(matrix_images_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, stream, d, b,
                     want_gradient, deriv_mode);
}

// --------------------------------------------------------------------------

// build-time knobs for experiments (scripts/build_lds_variants.sh); the defaults are the shipped kernel
#ifndef LDS_WAVES
#define LDS_WAVES 4
#endif
#ifndef LDS_WAVES_PER_EU
#define LDS_WAVES_PER_EU 1
#endif
#ifndef LDS_FORCE_G
#define LDS_FORCE_G 0
#endif
#ifndef LDS_COND_LOADS
#define LDS_COND_LOADS 0
#endif
#ifndef LDS_TILE_RUN
#define LDS_TILE_RUN 0  // 0: the launcher picks the run length; n: force runs of n tiles (experiments)
#endif
#ifndef LDS_NO_CHERRIES
#define LDS_NO_CHERRIES 0  // 1: store every internal node (ablation of the cherry folding)
#endif
constexpr int kLdsWaves = LDS_WAVES;
constexpr size_t kLdsBudget = 160 * 1024;

__device__ __forceinline__ double Mfma(double a, double x, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, x, c, 0, 0, 0);
}

// Tips are held in LDS as 4-bit masks (bit k set when the observed symbol is compatible
// with state k; a gap sets all four).  The MFMA B operand of a tip is then the double
// 1.0 or 0.0 selected by this lane's state bit: only the high dword differs.
// A set bit becomes 2.0, whose high dword is the single bit 30 (one shift instead of a
// compare + select); every site likelihood then carries the exact factor 2^n, removed from the
// log-likelihood at the end, and the gradients are ratios in which it cancels.
__device__ __forceinline__ double TipOperand(int mask, int st) {
  const int hi = (mask << (30 - st)) & 0x40000000;  // 30 - st is loop-invariant per lane
  return __hiloint2double(hi, 0);
}

// ---- cross-lane sums without LDS traffic (DPP / permlane swaps) -------------
template <int kCtrl>
__device__ __forceinline__ double DppMove(double v) {
  const long long bits = __builtin_bit_cast(long long, v);
  // row rotations read every lane, so the "old" operand is never used: mov_dpp leaves it undefined
  // and saves the v_mov that update_dpp(0, ...) needs to materialise it
  const int lo = __builtin_amdgcn_mov_dpp((int)bits, kCtrl, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(bits >> 32), kCtrl, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kRowRor1 = 0x121, kRowRor2 = 0x122, kRowRor4 = 0x124, kRowRor8 = 0x128;

// Sum over the 16 lanes of each DPP row; every lane of the row ends with the row sum.
__device__ __forceinline__ double RowSum16(double v) {
  v += DppMove<kRowRor8>(v);
  v += DppMove<kRowRor4>(v);
  v += DppMove<kRowRor2>(v);
  v += DppMove<kRowRor1>(v);
  return v;
}

// rows (16-lane groups) r0 r1 r2 r3 -> r0+r1, r0+r1, r2+r3, r2+r3
__device__ __forceinline__ double PairRows(double v) {
  const long long bits = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32);
  const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double a = __builtin_bit_cast(double, ((long long)h[0] << 32) | l[0]);
  const double b = __builtin_bit_cast(double, ((long long)h[1] << 32) | l[1]);
  return a + b;
}

// lanes 0-31 <- a[l] + a[l+32], lanes 32-63 <- b[l-32] + b[l]
__device__ __forceinline__ double MergeHalves(double a, double b) {
  const long long ab = __builtin_bit_cast(long long, a), bb = __builtin_bit_cast(long long, b);
  const auto l = __builtin_amdgcn_permlane32_swap((unsigned)ab, (unsigned)bb, false, false);
  const auto h = __builtin_amdgcn_permlane32_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
  const double x = __builtin_bit_cast(double, ((long long)h[0] << 32) | l[0]);
  const double y = __builtin_bit_cast(double, ((long long)h[1] << 32) | l[1]);
  return x + y;
}

// ---- step schedule ---------------------------------------------------------------------------
// CHERRIES (internal nodes over two tips, the root excepted) are never stored and have no step of their
// own: where a step's child is a cherry, its partial is rebuilt from the two tip look-ups
// (x = (P_A e_A) . (P_B e_B), two MFMAs), and in the pre-order pass the cherry's own step -- the
// derivatives of its two tip edges -- is folded into its parent's step.  A third of a tree's internal
// nodes are cherries, so a wave's LDS region holds a third fewer cells and a workgroup a third more
// pattern groups; that matters because a step costs about five groups' worth of fixed latency
// (measured: scripts/gpu_lds_occupancy.py), so groups per SIMD are what fills the pipes.
//
// Everything a step needs that depends only on the topology is tabulated once per tree and read
// with ONE scalar load per step (s_load_dwordx16, issued two steps ahead): LDS byte offsets of
// the operands (tip row or arena cell), of the node's own cell, what kind each child is, whether an
// operand is forwarded in registers, the child ids (gradient rows) and the image offsets to
// prefetch for the step after next.  Two tables per tree: post-order (ascending node) and pre-order
// (descending node), one entry per stored (non-cherry) internal node.
struct alignas(32) StepDesc {
  unsigned off0, off1;  // LDS byte offset of operand 0 / 1: tip row c*PB, arena cell, or (cherry) tip row of its first tip
  unsigned cell_flags;  // bits 0..17: LDS byte offset of this node's arena cell; bits 18..22: flags; bits 24..31: steps of the pass
  unsigned c01;         // child ids, c0 | c1 << 16
  unsigned pf01;        // ids of the children of the step after next (their images are prefetched)
  unsigned pfab;        // cherry tips of the step after next, one byte each: a0, b0, a1, b1 (0: no cherry)
  unsigned ab0, ab1;    // cherry child 0 / 1: its tips' ids, a | b << 16 (0xffffffff: no cherry)
};
// A descriptor is fetched with an explicit scalar load: the compiler only selects s_load for
// memory it can prove unclobbered, which it does not here.  StepFetch two steps ahead, StepWait one step ahead.
// The load and its wait are ONE asm statement since round 2: a tuple whose load is still in flight must not be
// visible to the compiler, which treats the asm's output as written -- it copied such a tuple (s_mov) ahead of
// the wait and used registers of it as temporaries (the image offsets of cherry tips) while the data was still
// on its way.  With the step tables in the scalar cache the data always beat those instructions; in later tiles
// of a run of tiles, lines that other CUs had evicted meanwhile did not, and a few trees per pass came back with
// wrong gradients on cherry tip edges (55 taxa and more, one pattern group per wave, where bodies are shortest;
// scripts/gpu_lds_runs_check.py).  The wait costs 4 % (2.47 -> 2.57 ms per 1600 trees of 27 taxa).
typedef unsigned StepWords __attribute__((ext_vector_type(8)));
// Pull 64 bytes (two descriptors) into the scalar cache; the data itself is discarded.
typedef unsigned WarmWords __attribute__((ext_vector_type(16)));
// (The destination tuple is an in/out operand that stays live up to a final wait: the load
// lands asynchronously, so its registers must not be handed to anything else meanwhile.)
__device__ __forceinline__ void StepWarm(const StepDesc* p, WarmWords& w) {
  asm volatile("s_load_dwordx16 %0, %1, 0x0" : "+s"(w) : "s"(p));
}
__device__ __forceinline__ void StepWarmDone(WarmWords& w) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w)::"memory"); }
__device__ __forceinline__ StepWords StepFetch(const StepDesc* p) {
  StepWords w;
#if LDS_SYNC_FETCH
  asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(p) : "memory");
#else
  asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(w) : "s"(p));
#endif
  return w;
}
// the wait "produces" the descriptor, so no use of it can be scheduled above the wait
__device__ __forceinline__ void StepWait(StepWords& w) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w)::"memory"); }
// word indices of the fields, and their unpacking
enum { kOff0 = 0, kOff1 = 1, kCellFlags = 2, kC01 = 3, kPf01 = 4, kPfAb = 5, kAb0 = 6, kAb1 = 7 };
#define DS_CELL(ds) ((ds)[kCellFlags] & 0x3ffffu)
#define DS_FLAGS(ds) ((ds)[kCellFlags] >> 18)
#define DS_C0(ds) ((ds)[kC01] & 0xffffu)
#define DS_C1(ds) ((ds)[kC01] >> 16)
// tip row of a cherry child's second tip (PB = patterns per workgroup, the tip buffer's row length)
#define DS_OFFB0(ds) (((ds)[kAb0] >> 16) * (unsigned)PB)
#define DS_OFFB1(ds) (((ds)[kAb1] >> 16) * (unsigned)PB)
// entries per pass: at most one per internal node plus two trailing copies of the last one, so the
// two-steps-ahead fetch needs no clamp
__host__ __device__ constexpr int SchedEntries(int NI) { return NI + 2; }
constexpr int kFlagTip0 = 1, kFlagTip1 = 2, kFlagForward = 4, kFlagCherry0 = 8, kFlagCherry1 = 16;
constexpr unsigned kNoCherry = 0xffffffffu;

// One wave per tree: which nodes are cherries, compact cell numbers for the others, both tables
// (one lane per step).
constexpr int kMaxLdsTaxa = 256;  // (step count is an 8-bit field; larger trees do not fit LDS anyway)

__global__ void __launch_bounds__(64)
lds_schedule_kernel(BatchDims d, int G, int PB, const int32_t* __restrict__ children, StepDesc* __restrict__ sched) {
  __shared__ int rank_of[kMaxLdsTaxa];  // internal node j: its cell number, or -1 for a cherry
  __shared__ int nodes[kMaxLdsTaxa];    // stored nodes in ascending order
  __shared__ int step_count;
  const int n = d.taxon_count, N = d.node_count, NI = n - 1;
  const int tree = blockIdx.x, lane = threadIdx.x;
  const int32_t* ch = children + (size_t)tree * NI * 2;
  const int E = SchedEntries(NI);
  StepDesc* post = sched + (size_t)tree * 2 * E;
  StepDesc* pre = post + E;
  auto is_cherry = [&](int c) { return !LDS_NO_CHERRIES && c >= n && c != N - 1 && ch[2 * (c - n)] < n && ch[2 * (c - n) + 1] < n; };
  for (int j = lane; j < NI; j += 64) rank_of[j] = is_cherry(n + j) ? -1 : 0;
  __syncthreads();
  if (lane == 0) {
    int r = 0;
    for (int j = 0; j < NI; j++)
      if (rank_of[j] == 0) {
        rank_of[j] = r;
        nodes[r++] = n + j;
      }
    step_count = r;
  }
  __syncthreads();
  const int steps = step_count;
  auto cherry = [&](int c) { return c >= n && rank_of[c - n] < 0; };
  auto operand = [&](int c) -> unsigned {
    if (c < n) return (unsigned)(c * PB);
    if (cherry(c)) return (unsigned)(ch[2 * (c - n)] * PB);
    return (unsigned)(rank_of[c - n] * G * 512);
  };
  auto tip_bytes = [&](int c) -> unsigned {  // a | b << 8; tip 0 twice when c is no cherry (loaded, unused)
    return cherry(c) ? (unsigned)ch[2 * (c - n)] | ((unsigned)ch[2 * (c - n) + 1] << 8) : 0u;
  };
  auto tips_of = [&](int c) -> unsigned {
    return cherry(c) ? (unsigned)ch[2 * (c - n)] | ((unsigned)ch[2 * (c - n) + 1] << 16) : kNoCherry;
  };
  auto kinds = [&](int c0, int c1) -> unsigned {
    return (c0 < n ? kFlagTip0 : 0) | (c1 < n ? kFlagTip1 : 0) | (cherry(c0) ? kFlagCherry0 : 0) |
           (cherry(c1) ? kFlagCherry1 : 0);
  };
  auto entry = [&](int node, int ahead, bool fwd) {
    const int c0 = ch[2 * (node - n)], c1 = ch[2 * (node - n) + 1];
    const int a0 = ch[2 * (ahead - n)], a1 = ch[2 * (ahead - n) + 1];
    return StepDesc{operand(c0), operand(c1),
                    (unsigned)(rank_of[node - n] * G * 512) | ((kinds(c0, c1) | (fwd ? (unsigned)kFlagForward : 0u)) << 18) |
                        ((unsigned)steps << 24),
                    (unsigned)c0 | ((unsigned)c1 << 16), (unsigned)a0 | ((unsigned)a1 << 16),
                    tip_bytes(a0) | (tip_bytes(a1) << 16), tips_of(c0), tips_of(c1)};
  };
  for (int s = lane; s < E; s += 64) {
    const int q = s < steps ? s : steps - 1;  // the trailing entries repeat the last step
    // post-order step q: node nodes[q]; the previous step's result is still in registers when its node
    // is this step's second child
    const int node = nodes[q];
    post[s] = entry(node, nodes[q + 2 < steps ? q + 2 : steps - 1], q > 0 && ch[2 * (node - n) + 1] == nodes[q - 1]);
    // pre-order step q: node nodes[steps-1-q]; U is still in registers when this node is the (stored)
    // second child of the node processed just before
    const int pnode = nodes[steps - 1 - q];
    const bool fwd = q == 0 ? pnode == N - 1 : ch[2 * (nodes[steps - q] - n) + 1] == pnode;
    pre[s] = entry(pnode, nodes[steps - 1 - (q + 2 < steps ? q + 2 : steps - 1)], fwd);
  }
}

struct alignas(16) PdPair { double p, d; };  // one lane's P and dP entries of a branch image

template <int C, int G, bool GRAD>
__global__ void __launch_bounds__(kLdsWaves * 64, LDS_WAVES_PER_EU)
walk_lds_kernel(BatchDims d, int tiles, int units, int slots, int want_site, int tile_run, const StepDesc* __restrict__ sched,
                const double* __restrict__ images, const TreeModel* __restrict__ models,
                const uint8_t* __restrict__ tip_states, const double* __restrict__ weights,
                const double* __restrict__ branch, double* __restrict__ part_ll, double* __restrict__ part_grad) {
  extern __shared__ double lds[];
  constexpr int PG = 16 / C;                  // patterns per group
  constexpr int PB = kLdsWaves * G * PG;      // patterns per workgroup
  const int n = d.taxon_count, N = d.node_count, NI = n - 1, Ppad = d.pattern_stride;

  // XCD-aware unit order: workgroup b runs on XCD b % 8; give each XCD a contiguous range
  // of units so the tiles of one tree (same matrix images) share an L2.
  int unit = blockIdx.x;
  {
    const int per = units / 8, rem = units % 8, x = unit % 8, q = unit / 8;
    unit = x * per + (x < rem ? x : rem) + q;
  }
  // A workgroup walks a RUN of consecutive pattern tiles of one tree: everything that depends only on the
  // tree (step tables in the scalar cache, the first steps' descriptors and images) is fetched once, and
  // the next tile's tip states and weights are requested while this tile is walked, so only the first
  // tile of a run pays the start-up latency that nothing else on the CU can hide.
  // (Every instantiation but <1,8,.> takes runs: with the loop in that one the compiler's AGPR-copy rewrite
  // for -amdgpu-mfma-vgpr-form crashes.)
  constexpr bool kRuns = !(C == 1 && G == 8);
  const int run = kRuns ? tile_run : 1;
  const int runs = (tiles + run - 1) / run;
  const int tree = unit / runs, tile0 = (unit % runs) * run;
  const int tile_count = min(run, tiles - tile0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int st = lane >> 4, blk = (lane >> 2) & 3, pj = lane & 3;
  const int cat = blk % C, sub = blk / C;

  // LDS carve-up: per-wave arena | tip states of the workgroup's patterns | gradient rows | ll
  char* const arena_b = reinterpret_cast<char*>(lds + (size_t)wave * slots * G * 64 + lane);
  uint8_t* tipbuf = reinterpret_cast<uint8_t*>(lds + (size_t)kLdsWaves * slots * G * 64);
  double* grad_rows = lds + (size_t)kLdsWaves * slots * G * 64 + (n * PB + 7) / 8;
  double* ll_slots = grad_rows + kLdsWaves * 4 * N;

  // wave-uniform base + 32-bit per-lane byte offset: global loads in the SGPR-base form
  const char* __restrict__ img_b = reinterpret_cast<const char*>(images + (size_t)tree * (N - 1) * kImgStride);
  const unsigned lane8 = (unsigned)lane * 8u;
  const TreeModel* __restrict__ tm = models + tree;
  const StepDesc* __restrict__ post_tab = sched + (size_t)tree * 2 * SchedEntries(NI);
  const StepDesc* __restrict__ pre_tab = post_tab + SchedEntries(NI);

  // Workgroup preamble.  Only one workgroup fits a CU (LDS), so nothing hides its start-up
  // latency: every global load of the prologue is issued before the first wait -- the first two
  // step descriptors (scalar), the tile's tip states, the pattern weights and model constants --
  // and the first matrix images as soon as the descriptors are in.
  // Both step tables (2 x (NI+2) x 32 B) are pulled into the scalar cache now: a step's descriptor
  // fetch then never waits on L2, which matters because scalar loads share their counter with LDS.
  StepWords PD0 = StepFetch(post_tab), PD1 = StepFetch(post_tab + 1);
  constexpr int kTipBatch = 8;
  const int tip_total = n * PB;
  const int loc0 = wave * G * PG + sub * 4 + pj;  // pattern of group 0 inside the tile; group g: + g*PG
  const uint8_t* tip_b = tipbuf + loc0;
  auto load_tile_inputs = [&](int tile, int (&sym)[kTipBatch], double (&w)[G]) {
#pragma unroll
    for (int u = 0; u < kTipBatch; u++) {
      const int q = tid + u * kLdsWaves * 64;
      sym[u] = q < tip_total ? tip_states[(size_t)(q / PB) * Ppad + tile * PB + (q % PB)] : 4;
    }
#pragma unroll
    for (int g = 0; g < G; g++) w[g] = weights[tile * PB + loc0 + g * PG];
  };
  int tip_sym[kTipBatch];
  double wgt[G];     // pattern weight
  load_tile_inputs(tile0, tip_sym, wgt);
  const double pi_st = tm->pi[st];
  const double w_cat = tm->cat_weight[cat];
  StepWait(PD0);
  StepWait(PD1);
  const int steps = (int)(PD0[kCellFlags] >> 24);
// P of a branch image (lane pair slot 0), its (P, dP) pair, and its P^T
#define IMAGE_P(off) (*reinterpret_cast<const double*>(img_b + (size_t)((off) + 2u * lane8)))
#define IMAGE_PD(off) (*reinterpret_cast<const PdPair*>(img_b + (size_t)((off) + 2u * lane8)))
#define IMAGE_PT(off) (*reinterpret_cast<const double*>(img_b + (size_t)((off) + 1024u + lane8)))
#define TIP_P(id) IMAGE_P((id) * (unsigned)(kImgStride * 8))
#define TIP_PD(id) IMAGE_PD((id) * (unsigned)(kImgStride * 8))
  // P images of the two child branches, and of the tip branches under a cherry child
  struct Img2 { double m0, m1, a0, b0, a1, b1; };
  constexpr unsigned kImgBytes = kImgStride * 8;
  auto load2 = [&](Img2& r, unsigned id0, unsigned id1) {
    r.m0 = IMAGE_P(id0 * kImgBytes);
    r.m1 = IMAGE_P(id1 * kImgBytes);
  };
  // images of the tip branches under cherry children (prefetched two steps ahead like the others)
  auto load2_cherries = [&](Img2& r, unsigned pfab) {  // bytes a0, b0, a1, b1 (tip 0 where there is no cherry)
    if (LDS_NO_CHERRIES) return;
    // (unconditional: a conditional load leaves a register set that must be merged with the old one at
    // the join, and copying a register with a load in flight waits for the load)
#if LDS_COND_LOADS
    if (pfab & 0xffffu) {
      r.a0 = TIP_P(pfab & 0xffu);
      r.b0 = TIP_P((pfab >> 8) & 0xffu);
    }
    if (pfab >> 16) {
      r.a1 = TIP_P((pfab >> 16) & 0xffu);
      r.b1 = TIP_P(pfab >> 24);
    }
#else
    r.a0 = TIP_P(pfab & 0xffu);
    r.b0 = TIP_P((pfab >> 8) & 0xffu);
    r.a1 = TIP_P((pfab >> 16) & 0xffu);
    r.b1 = TIP_P(pfab >> 24);
#endif
  };
  auto pack_ab = [](unsigned ab0, unsigned ab1) {  // a | b << 16 words -> the byte form
    const unsigned lo = ab0 == kNoCherry ? 0u : (ab0 & 0xffu) | ((ab0 >> 8) & 0xff00u);
    const unsigned hi = ab1 == kNoCherry ? 0u : (ab1 & 0xffu) | ((ab1 >> 8) & 0xff00u);
    return lo | (hi << 16);
  };
  Img2 PS0 = {0, 0, 0, 0, 0, 0}, PS1 = PS0;
  load2(PS0, DS_C0(PD0), DS_C1(PD0));
  load2_cherries(PS0, pack_ab(PD0[kAb0], PD0[kAb1]));
  load2(PS1, DS_C0(PD1), DS_C1(PD1));
  load2_cherries(PS1, pack_ab(PD1[kAb0], PD1[kAb1]));
  // (issued behind the image loads: its latency hides under theirs and the tip states')
  WarmWords warm = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * SchedEntries(NI); i += 2) StepWarm(post_tab + i, warm);
  StepWarmDone(warm);
  for (int tile_index = 0; tile_index < tile_count; tile_index++) {
  const int tile = tile0 + tile_index;
#pragma unroll
  for (int u = 0; u < kTipBatch; u++) {
    const int q = tid + u * kLdsWaves * 64;
    if (q < tip_total) tipbuf[q] = (uint8_t)(tip_sym[u] < 4 ? 1 << tip_sym[u] : 15);
  }
  for (int q = tid + kTipBatch * kLdsWaves * 64; q < tip_total; q += kLdsWaves * 64) {  // trees beyond 8*256/PB taxa
    const int sym = tip_states[(size_t)(q / PB) * Ppad + tile * PB + (q % PB)];
    tipbuf[q] = (uint8_t)(sym < 4 ? 1 << sym : 15);
  }
  if (GRAD)
    for (int q = tid; q < kLdsWaves * 4 * N; q += kLdsWaves * 64) grad_rows[q] = 0.0;
  __syncthreads();
  // the next tile's inputs travel while this one is walked
  int next_sym[kTipBatch];
  double next_wgt[G];
  if (kRuns && tile_index + 1 < tile_count) load_tile_inputs(tile + 1, next_sym, next_wgt);

#define TIP_AT(off, g) tip_b[(off) + (g) * PG]
#define CELL_AT(off, g) (*reinterpret_cast<double*>(arena_b + (off) + (g) * 512))
  // Children are in ascending id order, so a tip never follows an internal node: the second child of a
  // step is a stored cell or a cherry (a node over two tips is itself a cherry and has no step, the root
  // of a tree with at least three taxa always has an internal child).
#define KIND_DISPATCH(flags_, ...)                                                        \
  {                                                                                       \
    const unsigned f_ = (flags_);                                                         \
    using I0 = std::integral_constant<int, 0>;                                            \
    using I1 = std::integral_constant<int, 1>;                                            \
    using I2 = std::integral_constant<int, 2>;                                            \
    if (f_ & kFlagCherry1) {                                                              \
      if (f_ & kFlagTip0) step(I0{}, I2{}, __VA_ARGS__);                                  \
      else if (f_ & kFlagCherry0) step(I2{}, I2{}, __VA_ARGS__);                          \
      else step(I1{}, I2{}, __VA_ARGS__);                                                 \
    } else {                                                                              \
      if (f_ & kFlagTip0) step(I0{}, I1{}, __VA_ARGS__);                                  \
      else if (f_ & kFlagCherry0) step(I2{}, I1{}, __VA_ARGS__);                          \
      else step(I1{}, I1{}, __VA_ARGS__);                                                 \
    }                                                                                     \
  }

  // ---------------- post-order ----------------------------------------------
  // A step is self-contained straight-line code, specialised on which children are tips
  // (children are in ascending id order, so a tip never follows an internal node); a cherry child
  // is a uniform branch inside the non-tip form.  Matrix images and step descriptors are fetched two
  // steps ahead into three rotating register sets (3-way unrolled loop: a set is never copied).  The
  // result of a step is handed to the next one in registers when that step's second child is this
  // node, the common case.
  double res[G];
  {
    StepWords D0 = PD0, D1 = PD1, D2 = PD1;
    Img2 S0 = PS0, S1 = PS1, S2 = PS1;
    int k = 0;
    auto step = [&](auto k0_c, auto k1_c, const StepWords& ds, const Img2& cur, StepWords& nd, Img2& nxt,
                    StepWords& dfill, Img2& fill) {
      // child kinds are compile-time: 0 tip, 1 stored cell, 2 cherry (rebuilt from its two tips)
      constexpr int K0 = decltype(k0_c)::value, K1 = decltype(k1_c)::value;
      const unsigned flags = DS_FLAGS(ds);
      double x0[G], x1[G];
#pragma unroll
      for (int g = 0; g < G; g++) {
        if (K0 == 0) x0[g] = TipOperand(TIP_AT(ds[kOff0], g), st);
        if (K0 == 1) x0[g] = CELL_AT(ds[kOff0], g);
        if (K0 == 2)
          x0[g] = Mfma(cur.a0, TipOperand(TIP_AT(ds[kOff0], g), st), 0.0) *
                  Mfma(cur.b0, TipOperand(TIP_AT(DS_OFFB0(ds), g), st), 0.0);
        if (K1 == 0) x1[g] = TipOperand(TIP_AT(ds[kOff1], g), st);
        if (K1 == 2)
          x1[g] = Mfma(cur.a1, TipOperand(TIP_AT(ds[kOff1], g), st), 0.0) *
                  Mfma(cur.b1, TipOperand(TIP_AT(DS_OFFB1(ds), g), st), 0.0);
      }
      if (K1 == 1) {
        if (flags & kFlagForward) {
#pragma unroll
          for (int g = 0; g < G; g++) x1[g] = res[g];
        } else {
#pragma unroll
          for (int g = 0; g < G; g++) x1[g] = CELL_AT(ds[kOff1], g);
        }
      }
      load2(fill, ds[kPf01] & 0xffffu, ds[kPf01] >> 16);
      load2_cherries(fill, ds[kPfAb]);
      double a0[G], a1[G];
#pragma unroll
      for (int g = 0; g < G; g++) {
        a0[g] = Mfma(cur.m0, x0[g], 0.0);
        a1[g] = Mfma(cur.m1, x1[g], 0.0);
      }
      // scalar fetch of the descriptor two steps ahead, issued where no LDS wait follows soon
      // (scalar and LDS loads share a counter that can only be waited to zero)
      __builtin_amdgcn_sched_barrier(0);
      // Every LDS access issued so far has been consumed, so waiting the shared counter to zero
      // here is free: it confirms the NEXT step's descriptor (fetched a step ago), and the top of
      // a step then needs no wait -- this step's LDS stores drain under the next step's loads.
      StepWait(nd);
      dfill = StepFetch(post_tab + k + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < G; g++) res[g] = a0[g] * a1[g];
      if (k < steps - 1) {  // the root partial is consumed below, never stored
#pragma unroll
        for (int g = 0; g < G; g++) CELL_AT(DS_CELL(ds), g) = res[g];
      }
    };
    auto dispatch = [&](const StepWords& ds, const Img2& cur, StepWords& nd, Img2& nxt, StepWords& dfill, Img2& fill) {
      KIND_DISPATCH(DS_FLAGS(ds), ds, cur, nd, nxt, dfill, fill)
    };
    while (true) {
      dispatch(D0, S0, D1, S1, D2, S2);
      if (++k >= steps) break;
      dispatch(D1, S1, D2, S2, D0, S0);
      if (++k >= steps) break;
      dispatch(D2, S2, D0, S0, D1, S1);
      if (++k >= steps) break;
    }
    // The last steps fetched descriptors that nobody waits for.  A scalar load lands whenever it lands, and
    // its destination registers are dead to the compiler from here on: nothing after the loop may be
    // scheduled, or given those registers, before the load has landed.  (The wait names no register on
    // purpose: an operand would make the compiler copy the in-flight tuple at every loop exit.)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---------------- root: site likelihoods ----------------------------------
  // res = root partial.  L_p = sum_c w_c sum_i pi_i root_c[i]: sum over the state bits
  // (lane bits 4,5) and the category bits (lane bits 2..3 as far as C uses them).
  double ll_acc = 0.0;
  double coef[G];  // w_c * w_p / L_p for this lane's (category, pattern)
  double Ls[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    double L = res[g] * (pi_st * w_cat);
    L += __shfl_xor(L, 16);
    L += __shfl_xor(L, 32);
    if (C >= 2) L += __shfl_xor(L, 4);
    if (C == 4) L += __shfl_xor(L, 8);
    Ls[g] = L;  // the same value on every (state, category) lane of a pattern
    coef[g] = w_cat * (wgt[g] / L);
  }
  // log-likelihood: each (group, pattern) is counted on the lane whose (state, category) index
  // equals the group number, so one log() call serves 4C groups instead of one call per group
  {
    const int sel = st * C + cat;
#pragma unroll
    for (int g0 = 0; g0 < G; g0 += 4 * C) {
      double Lsel = 1.0, wsel = 0.0;
#pragma unroll
      for (int g = g0; g < G && g < g0 + 4 * C; g++)
        if (sel == g - g0) {
          Lsel = Ls[g];
          wsel = wgt[g];
        }
      ll_acc += wsel * (log(Lsel) - n * 0.6931471805599453);
    }
  }

  // ---------------- pre-order + edge derivatives ----------------------------
  if (GRAD) {
    double* my_row = grad_rows + wave * 4 * N;  // [block][edge]
    // P, dP, P^T images of the two child branches; P and dP of the tip branches under a cherry child
    struct Img { double p0, q0, t0, p1, q1, t1, pa0, da0, pb0, db0, pa1, da1, pb1, db1; };
    auto load_img = [&](Img& r, unsigned id0, unsigned id1) {
      const unsigned o0 = id0 * kImgBytes, o1 = id1 * kImgBytes;
      const PdPair c0 = IMAGE_PD(o0), c1 = IMAGE_PD(o1);
      r.p0 = c0.p; r.q0 = c0.d; r.t0 = IMAGE_PT(o0);
      r.p1 = c1.p; r.q1 = c1.d; r.t1 = IMAGE_PT(o1);
    };
    auto load_img_cherries = [&](Img& r, unsigned pfab) {  // unconditional, see load2_cherries
      if (LDS_NO_CHERRIES) return;
      const unsigned a0 = pfab & 0xffu, b0 = (pfab >> 8) & 0xffu, a1 = (pfab >> 16) & 0xffu, b1 = pfab >> 24;
#if LDS_COND_LOADS
      if (pfab & 0xffffu) {
        const PdPair ta = TIP_PD(a0), tb = TIP_PD(b0);
        r.pa0 = ta.p; r.da0 = ta.d; r.pb0 = tb.p; r.db0 = tb.d;
      }
      if (pfab >> 16) {
        const PdPair ta = TIP_PD(a1), tb = TIP_PD(b1);
        r.pa1 = ta.p; r.da1 = ta.d; r.pb1 = tb.p; r.db1 = tb.d;
      }
#else
      const PdPair ta0 = TIP_PD(a0), tb0 = TIP_PD(b0), ta1 = TIP_PD(a1), tb1 = TIP_PD(b1);
      r.pa0 = ta0.p; r.da0 = ta0.d; r.pb0 = tb0.p; r.db0 = tb0.d;
      r.pa1 = ta1.p; r.da1 = ta1.d; r.pb1 = tb1.p; r.db1 = tb1.d;
#endif
    };
    StepWords D0 = StepFetch(pre_tab), D1 = StepFetch(pre_tab + 1), D2;
    StepWait(D0);
    StepWait(D1);
    D2 = D1;
    Img S0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, S1 = S0, S2;
    load_img(S0, DS_C0(D0), DS_C1(D0));
    load_img_cherries(S0, pack_ab(D0[kAb0], D0[kAb1]));
    load_img(S1, DS_C0(D1), DS_C1(D1));
    load_img_cherries(S1, pack_ab(D1[kAb0], D1[kAb1]));
    S2 = S1;
    // U = pre-order partial of the step's node; for the root the stationary frequencies
    // (SetRootPreorderPartialsToStateFrequencies, fat_beagle.cpp:327-336).  It stays in
    // registers when the next node is this node's second child (the usual case).
    double U[G];
    // The weight of a (category, pattern) term, w_c w_p / L_p, is the same at every edge and every operation
    // of this pass is linear in U lane by lane, so it rides along from the root instead of being
    // multiplied into each edge's terms.
#pragma unroll
    for (int g = 0; g < G; g++) U[g] = pi_st * coef[g];
    // edge sums of the previous step, reduced one step late so that the reduction fills the
    // wait for this step's LDS operands: the two child edges, and the tip edges of cherry children
    double ps0 = 0.0, ps1 = 0.0, pa0 = 0.0, pb0 = 0.0, pa1 = 0.0, pb1 = 0.0;
    int pc0 = N - 1, pc1 = N - 1;  // root entry: the reduce kernel writes 0 there
    unsigned pab0 = kNoCherry, pab1 = kNoCherry;
    // The 64-lane sum of an edge on the matrix pipe: with the per-lane terms as the A operand and
    // ones as B, D_b[i][j] = sum_k v[16k+4b+i] (the four state rows); fed back as the B operand
    // under an all-ones A, D_b[i][j] = sum over the block's 16 lanes.  What is left, the sum over
    // the four blocks, is folded into the workgroup sum at the end: a row per block.
    const bool row_writer = (lane & 0x13) == 0;  // lanes 4b (first edge) and 32+4b (second edge)
    auto flush_pair = [&](double v0, double v1, int e0, int e1) {
      const double r0 = Mfma(v0, 1.0, 0.0), r1 = Mfma(v1, 1.0, 0.0);
      const double t0 = Mfma(1.0, r0, 0.0), t1 = Mfma(1.0, r1, 0.0);
      if (row_writer) my_row[blk * N + (lane < 32 ? e0 : e1)] = lane < 32 ? t0 : t1;
    };
    auto flush_edges = [&]() {
      flush_pair(ps0, ps1, pc0, pc1);
      if (pab0 != kNoCherry) flush_pair(pa0, pb0, (int)(pab0 & 0xffffu), (int)(pab0 >> 16));
      if (pab1 != kNoCherry) flush_pair(pa1, pb1, (int)(pab1 & 0xffffu), (int)(pab1 >> 16));
    };
    int j = 0;
    auto step = [&](auto k0_c, auto k1_c, const StepWords& ds, const Img& cur, StepWords& nd, Img& nxt,
                    StepWords& dfill, Img& fill) {
      constexpr int K0 = decltype(k0_c)::value, K1 = decltype(k1_c)::value;  // 0 tip, 1 stored cell, 2 cherry
      constexpr bool kTip0 = K0 == 0, kTip1 = K1 == 0, cherry0 = K0 == 2, cherry1 = K1 == 2;
      const unsigned flags = DS_FLAGS(ds);
      double x0[G], x1[G];
      // cherry children: tip operands and tip messages, kept for the folded step below
      double ta0[G], tb0[G], ma0[G], mb0[G], ta1[G], tb1[G], ma1[G], mb1[G];
#pragma unroll
      for (int g = 0; g < G; g++) {
        if (K0 == 0) x0[g] = TipOperand(TIP_AT(ds[kOff0], g), st);
        if (K0 == 1) x0[g] = CELL_AT(ds[kOff0], g);
        if (K0 == 2) {
          ta0[g] = TipOperand(TIP_AT(ds[kOff0], g), st);
          tb0[g] = TipOperand(TIP_AT(DS_OFFB0(ds), g), st);
          ma0[g] = Mfma(cur.pa0, ta0[g], 0.0);
          mb0[g] = Mfma(cur.pb0, tb0[g], 0.0);
          x0[g] = ma0[g] * mb0[g];
        }
        if (K1 == 0) x1[g] = TipOperand(TIP_AT(ds[kOff1], g), st);
        if (K1 == 1) x1[g] = CELL_AT(ds[kOff1], g);
        if (K1 == 2) {
          ta1[g] = TipOperand(TIP_AT(ds[kOff1], g), st);
          tb1[g] = TipOperand(TIP_AT(DS_OFFB1(ds), g), st);
          ma1[g] = Mfma(cur.pa1, ta1[g], 0.0);
          mb1[g] = Mfma(cur.pb1, tb1[g], 0.0);
          x1[g] = ma1[g] * mb1[g];
        }
      }
      if (!(flags & kFlagForward)) {
#pragma unroll
        for (int g = 0; g < G; g++) U[g] = CELL_AT(DS_CELL(ds), g);
      }
      load_img(fill, ds[kPf01] & 0xffffu, ds[kPf01] >> 16);
      load_img_cherries(fill, ds[kPfAb]);
      flush_edges();
      // this step: all matrix products first, then the element-wise work
      double a0[G], dd0[G], a1[G], dd1[G];
#pragma unroll
      for (int g = 0; g < G; g++) {
        a0[g] = Mfma(cur.p0, x0[g], 0.0);
        a1[g] = Mfma(cur.p1, x1[g], 0.0);
      }
#pragma unroll
      for (int g = 0; g < G; g++) {
        dd0[g] = Mfma(cur.q0, x0[g], 0.0);
        dd1[g] = Mfma(cur.q1, x1[g], 0.0);
      }
      __builtin_amdgcn_sched_barrier(0);
      StepWait(nd);  // free here, see the post-order pass
      dfill = StepFetch(pre_tab + j + 2);
      __builtin_amdgcn_sched_barrier(0);
      double ua0[G], ua1[G];
#pragma unroll
      for (int g = 0; g < G; g++) {
        ua1[g] = U[g] * a1[g];
        ua0[g] = U[g] * a0[g];
      }
      pab0 = pab1 = kNoCherry;
      if (!kTip0) {
        if (cherry0) {
          // the cherry's own step, folded in: q = its pre-order partial; tip edges A and B
          double sa = 0.0, sb = 0.0;
#pragma unroll
          for (int g = 0; g < G; g++) {
            const double q = Mfma(cur.t0, ua1[g], 0.0);
            sa = fma(q * mb0[g], Mfma(cur.da0, ta0[g], 0.0), sa);
            sb = fma(q * ma0[g], Mfma(cur.db0, tb0[g], 0.0), sb);
          }
          pa0 = sa; pb0 = sb; pab0 = ds[kAb0];
        } else {
#pragma unroll
          for (int g = 0; g < G; g++) CELL_AT(ds[kOff0], g) = Mfma(cur.t0, ua1[g], 0.0);
        }
      }
      if (!kTip1) {
        if (cherry1) {
          double sa = 0.0, sb = 0.0;
#pragma unroll
          for (int g = 0; g < G; g++) {
            const double q = Mfma(cur.t1, ua0[g], 0.0);
            sa = fma(q * mb1[g], Mfma(cur.da1, ta1[g], 0.0), sa);
            sb = fma(q * ma1[g], Mfma(cur.db1, tb1[g], 0.0), sb);
          }
          pa1 = sa; pb1 = sb; pab1 = ds[kAb1];
        } else {
          // the second child's pre-order partial goes straight into U: it is the next
          // step's U whenever the next node is that child
#pragma unroll
          for (int g = 0; g < G; g++) {
            U[g] = Mfma(cur.t1, ua0[g], 0.0);
            CELL_AT(ds[kOff1], g) = U[g];
          }
        }
      }
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int g = 0; g < G; g++) {
        s0 = fma(ua1[g], dd0[g], s0);
        s1 = fma(ua0[g], dd1[g], s1);
      }
      ps0 = s0; ps1 = s1; pc0 = DS_C0(ds); pc1 = DS_C1(ds);
    };
    auto dispatch = [&](const StepWords& ds, const Img& cur, StepWords& nd, Img& nxt, StepWords& dfill, Img& fill) {
      KIND_DISPATCH(DS_FLAGS(ds), ds, cur, nd, nxt, dfill, fill)
    };
    while (true) {
      dispatch(D0, S0, D1, S1, D2, S2);
      if (++j >= steps) break;
      dispatch(D1, S1, D2, S2, D0, S0);
      if (++j >= steps) break;
      dispatch(D2, S2, D0, S0, D1, S1);
      if (++j >= steps) break;
    }
    __builtin_amdgcn_sched_barrier(0);  // (as after the post-order pass: no descriptor load is left in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    flush_edges();
  }
#undef TIP_AT
#undef CELL_AT
#undef IMAGE_P
#undef IMAGE_PD
#undef IMAGE_PT
#undef TIP_P
#undef TIP_PD
#undef KIND_DISPATCH

  // ---------------- workgroup sums, fixed order -----------------------------
  double wll = ll_acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wll += __shfl_xor(wll, o);
  if (lane == 0) ll_slots[wave] = wll;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < kLdsWaves; w++) s += ll_slots[w];
    part_ll[(size_t)tree * tiles + tile] = s;
  }
  if (GRAD) {
    // The rows are still separate per block, i.e. per rate category: the site-model gradient
    // (DiscreteSiteModelGradient, fat_beagle.cpp:401-410,538-550 -- the same edge sums with r_c replaced by
    // d r_c / d shape, times the branch lengths) is the same data weighted by (d r_c / d shape) / r_c, so
    // it needs no second traversal.  Its per-tile value travels in the root's slot of the gradient row
    // (the root has no branch; the reduce kernel moves it to out_site and writes the 0).
    double* out = part_grad + ((size_t)tree * tiles + tile) * N;
    for (int e = tid; e < N; e += kLdsWaves * 64) {
      double s = 0.0;
      for (int w = 0; w < kLdsWaves * 4; w++) s += grad_rows[w * N + e];
      out[e] = s;
    }
    if (want_site) {  // (only when the caller asked for the site-model gradient)
      double ratio[C];
#pragma unroll
      for (int c = 0; c < C; c++) ratio[c] = tm->cat_rate_deriv[c] / tm->cat_rate[c];
      double site_acc = 0.0;
      for (int e = tid; e < N; e += kLdsWaves * 64) {
        double sr = 0.0;
        for (int w = 0; w < kLdsWaves * 4; w++) sr += grad_rows[w * N + e] * ratio[(w & 3) % C];
        site_acc += sr * branch[(size_t)tree * N + e];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) site_acc += __shfl_xor(site_acc, o);
      if (lane == 0) ll_slots[kLdsWaves + wave] = site_acc;
      __syncthreads();
      if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < kLdsWaves; w++) s += ll_slots[kLdsWaves + w];
        out[N - 1] = s;
      }
    }
  }
  if (kRuns && tile_index + 1 < tile_count) {
    __syncthreads();  // every wave is done with the tip buffer, the gradient rows and the sums
#pragma unroll
    for (int u = 0; u < kTipBatch; u++) tip_sym[u] = next_sym[u];
#pragma unroll
    for (int g = 0; g < G; g++) wgt[g] = next_wgt[g];
  }
  }  // tiles of the run
}

// cells per pattern group: the stored (non-cherry, non-root) internal nodes of the tree of the batch
// that has the fewest cherries (BatchDims::min_cherries, counted by the host while it validates)
static int LdsSlots(const BatchDims& d) {
  const int cherries = LDS_NO_CHERRIES ? 0 : d.min_cherries;
  return d.taxon_count - 2 - cherries > 1 ? d.taxon_count - 2 - cherries : 1;
}

static size_t LdsBytes(const BatchDims& d, int G) {
  const int PG = 16 / d.category_count, PB = kLdsWaves * G * PG, n = d.taxon_count;
  const size_t arena = (size_t)kLdsWaves * LdsSlots(d) * G * 64;
  const size_t tips = ((size_t)n * PB + 7) / 8;
  return (arena + tips + (size_t)kLdsWaves * 4 * d.node_count + 2 * kLdsWaves) * sizeof(double);
}

LdsPlan PlanLds(const BatchDims& d) {
  LdsPlan plan{0, 0, 0, 0};
  const int C = d.category_count;
  if (C != 1 && C != 2 && C != 4) return plan;
  if (d.taxon_count < 3 || d.taxon_count >= kMaxLdsTaxa) return plan;  // (ids and step counts are 8-bit fields)
  // largest G in {1,2,3,4,6,8} that fits, but no more groups than the alignment can fill
  const int candidates[] = {8, 6, 4, 3, 2, 1};
  const int PG = 16 / C;
  for (int G : candidates) {
    if (LDS_FORCE_G && G != LDS_FORCE_G) continue;
    if (LdsBytes(d, G) > kLdsBudget) continue;
    const int PB = kLdsWaves * G * PG;
    if (G > 1 && PB > d.pattern_count + PB / 2 && PB > 2 * kLdsWaves * PG) continue;  // mostly padding
    plan.groups = G;
    plan.patterns_per_block = PB;
    plan.tiles = (d.pattern_count + PB - 1) / PB;
    plan.lds_bytes = LdsBytes(d, G);
    break;
  }
  return plan;
}

template <int C, int G>
static void LaunchWalkLdsCG(const BatchDims& d, const DeviceBatch& b, const LdsPlan& plan, int want_gradient,
                            int want_site, hipStream_t stream) {
  // Run length: a divisor of the tile count (equal runs), as long as the launch still has about sixteen
  // workgroups per CU to even out -- a run saves start-up latency, a short grid loses to quantisation.
  int tile_run = 1;
  if (!(C == 1 && G == 8)) {
    long long budget = LDS_TILE_RUN ? LDS_TILE_RUN : (long long)d.tree_count * plan.tiles / (16 * 256);
    if (const char* forced = std::getenv("BITO_AMD_LDS_TILE_RUN")) budget = std::atoi(forced);  // tests: runs on small batches
    for (int k = 1; k <= plan.tiles && k <= budget; k++)
      if (plan.tiles % k == 0) tile_run = k;
  }
  const int units = d.tree_count * ((plan.tiles + tile_run - 1) / tile_run);  // a workgroup per run of tiles
  const dim3 grid(units), block(kLdsWaves * 64);
  const StepDesc* sched = reinterpret_cast<const StepDesc*>(b.sched);
  auto kern = want_gradient ? walk_lds_kernel<C, G, true> : walk_lds_kernel<C, G, false>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsBudget);
  hipLaunchKernelGGL(kern, grid, block, plan.lds_bytes, stream, d, plan.tiles, units, LdsSlots(d), want_site, tile_run, sched, b.images, b.model,
                     b.tip_states, b.weights, b.branch, b.part_ll, b.part_grad);
}

template <int C>
static void LaunchWalkLdsC(const BatchDims& d, const DeviceBatch& b, const LdsPlan& plan, int want_gradient,
                           int want_site, hipStream_t stream) {
  switch (plan.groups) {
    case 1: LaunchWalkLdsCG<C, 1>(d, b, plan, want_gradient, want_site, stream); break;
    case 2: LaunchWalkLdsCG<C, 2>(d, b, plan, want_gradient, want_site, stream); break;
    case 3: LaunchWalkLdsCG<C, 3>(d, b, plan, want_gradient, want_site, stream); break;
    case 4: LaunchWalkLdsCG<C, 4>(d, b, plan, want_gradient, want_site, stream); break;
    case 6: LaunchWalkLdsCG<C, 6>(d, b, plan, want_gradient, want_site, stream); break;
    case 8: LaunchWalkLdsCG<C, 8>(d, b, plan, want_gradient, want_site, stream); break;
    default: break;
  }
}

size_t LdsScheduleInts(const BatchDims& d) { return (size_t)d.tree_count * 2 * SchedEntries(d.taxon_count - 1) * 16; }

void LaunchLdsSchedule(const BatchDims& d, const DeviceBatch& b, const LdsPlan& plan, hipStream_t stream) {
  hipLaunchKernelGGL(lds_schedule_kernel, dim3(d.tree_count), dim3(64), 0, stream, d, plan.groups,
                     plan.patterns_per_block, b.children, reinterpret_cast<StepDesc*>(b.sched));
}

void LaunchWalkLds(const BatchDims& d, const DeviceBatch& b, const LdsPlan& plan, int want_gradient, int want_site,
                   hipStream_t stream) {
  switch (d.category_count) {
    case 1: LaunchWalkLdsC<1>(d, b, plan, want_gradient, want_site, stream); break;
    case 2: LaunchWalkLdsC<2>(d, b, plan, want_gradient, want_site, stream); break;
    case 4: LaunchWalkLdsC<4>(d, b, plan, want_gradient, want_site, stream); break;
    default: break;
  }
}

}  // namespace bito_amd
