// kernels.hip -- gfx950 kernels of the per-tree likelihood path.
//
// Pipeline for one resident batch of T trees (all on one HIP stream):
//   setup_trees_kernel     parent-id vector -> children lists (+ detrifurcation),
//                          effective branch lengths, per-tree Q / eigensystem /
//                          category rates            (reference fat_beagle.cpp:42-45,
//                          unrooted_tree.cpp:27-37, tree.cpp:82-88, substitution_model.cpp,
//                          site_model.cpp)
//   transition_matrices_kernel  P(t r_c) and dP/dt for every branch and category
//                          (beagleUpdateTransitionMatrices, fat_beagle.cpp:315-325;
//                          differential matrices :101-111)
//   walk_hbm_kernel        post-order partials + root log-likelihood
//                          (beagleUpdatePartials / CalculateRootLogLikelihoods,
//                          fat_beagle.cpp:54-68) and, for gradients, the pre-order
//                          partials and edge derivatives (:138-160) fused in one walk
//   reduce_tiles_kernel    fixed-order sum of the per-tile partial results
//
// FP64 throughout.  No atomics anywhere: every sum has a fixed order, so results
// are bit-reproducible run to run.
#include "kernels.hpp"
#include <algorithm>
#include <cstdlib>

#ifndef HBM_PRE_UNROLL
#define HBM_PRE_UNROLL 1  // rate categories of a pre-order step whose loads are in flight together
#endif
#ifndef HBM_WAVES_EXPR
#define HBM_WAVES_EXPR ((C <= 4) ? 4 : 1)  // waves per SIMD the register allocation is held to (measured: 3, 4, 5, 6)
#endif

namespace bito_amd {

// --------------------------------------------------------------------------
// Set-up: one thread per tree.  The per-tree recursions are serial, so what matters is that
// their memory accesses do not queue up behind each other: a workgroup stages the wire-format
// rows of its trees in LDS with coalesced loads, each tree's thread works out of LDS, and the
// results leave with coalesced stores (trees too large for that use the direct form).  As many trees per
// workgroup as LDS holds, up to 128: the kernel runs beside the traversal of the previous pass, whose
// workgroups need a whole CU each, so what it costs is the number of CUs it sits on (a thread takes the same
// 50 us whether 15 or 127 others share its workgroup).

constexpr int kSetupTrees = 16;  // trees per workgroup of the staging-free forms

__global__ void __launch_bounds__(64)
setup_trees_kernel(BatchDims d, ModelSpec spec, DeviceBatch b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.tree_count) return;
  SetupTopology(d, b, t);
  SetupTreeModel(spec, b.params + (size_t)t * spec.param_count, &b.model[t]);
}

// Large trees: one workgroup per tree.  A bifurcating node's children in ascending id order are the
// minimum and the maximum of its children's ids (integer atomics in LDS: the result does not depend on
// the order of arrival); the trifurcating root of an unrooted tree also needs the middle one, sum - min - max.
__global__ void __launch_bounds__(256)
setup_large_trees_kernel(BatchDims d, ModelSpec spec, DeviceBatch b) {
  extern __shared__ int setup_ch[];  // [NI][2] min / max, then the root's id sum
  const int n = d.taxon_count, N = d.node_count, M = d.in_node_count, NI = n - 1;
  const int t = blockIdx.x, tid = threadIdx.x;
  int* root_sum = setup_ch + 2 * NI;
  for (int k = tid; k < NI; k += 256) {
    setup_ch[2 * k] = 0x7fffffff;
    setup_ch[2 * k + 1] = -1;
  }
  if (tid == 0) *root_sum = 0;
  __syncthreads();
  const int32_t* parent = b.parent_ids + (size_t)t * (M - 1);
  for (int child = tid; child < M - 1; child += 256) {
    const int p = parent[child], k = p - n;
    atomicMin(&setup_ch[2 * k], child);
    atomicMax(&setup_ch[2 * k + 1], child);
    if (p == M - 1) atomicAdd(root_sum, child);
  }
  double* bl = b.branch + (size_t)t * N;
  const double* bl_in = b.branch_in + (size_t)t * M;
  const double* rates = (d.rooted && b.rates != nullptr) ? b.rates + (size_t)t * (M - 1) : nullptr;
  for (int i = tid; i < M; i += 256) bl[i] = (rates != nullptr && i < N - 1) ? bl_in[i] * rates[i] : bl_in[i];
  __syncthreads();
  if (!d.rooted && tid == 0) {
    // UnrootedTree::Detrifurcate (reference src/unrooted_tree.cpp:27-37), as SetupTopologyCore
    const int r = M - 1;
    const int a = setup_ch[2 * (r - n)], c = setup_ch[2 * (r - n) + 1], mid = *root_sum - a - c;
    setup_ch[2 * (r - n)] = mid;
    setup_ch[2 * (r - n) + 1] = c;
    setup_ch[2 * (r + 1 - n)] = a;
    setup_ch[2 * (r + 1 - n) + 1] = r;
    bl[r] = 0.0;
    bl[r + 1] = 0.0;
  }
  __syncthreads();
  int32_t* ch = b.children + (size_t)t * NI * 2;
  for (int k = tid; k < 2 * NI; k += 256) ch[k] = setup_ch[k];
  if (tid == 0) SetupTreeModel(spec, b.params + (size_t)t * spec.param_count, &b.model[t]);
}

__global__ void __launch_bounds__(256)
setup_trees_lds_kernel(BatchDims d, ModelSpec spec, DeviceBatch b, int trees) {
  extern __shared__ double setup_lds[];
  const int n = d.taxon_count, N = d.node_count, M = d.in_node_count, NI = n - 1;
  const int pc = spec.param_count > 0 ? spec.param_count : 1, RW = (d.rooted && b.rates != nullptr) ? M - 1 : 0;
  const int t0 = blockIdx.x * trees, tid = threadIdx.x, step = blockDim.x;
  const int count = min(trees, d.tree_count - t0);
  double* bl = setup_lds;                                        // [trees][N]
  double* prm = bl + trees * N;                                  // [trees][pc]
  double* rts = prm + trees * pc;                                // [trees][M-1] (rooted trees with rates)
  int32_t* par = reinterpret_cast<int32_t*>(rts + trees * (d.rooted ? M - 1 : 0));  // [trees][M-1]
  int32_t* ch = par + trees * (M - 1);                           // [trees][2 NI]
  StageSlice<256>(b.parent_ids + (size_t)t0 * (M - 1), b.copy_parent_ids ? b.copy_parent_ids + (size_t)t0 * (M - 1) : nullptr,
             count * (M - 1), tid, [&](int i, int32_t v) { par[i] = v; });
  StageSlice<256>(b.branch_in + (size_t)t0 * M, b.copy_branch_in ? b.copy_branch_in + (size_t)t0 * M : nullptr, count * M, tid,
             [&](int i, double v) { bl[(i / M) * N + i % M] = v; });
  if (spec.param_count > 0)
    StageSlice<256>(b.params + (size_t)t0 * pc, b.copy_params ? b.copy_params + (size_t)t0 * pc : nullptr, count * pc, tid,
               [&](int i, double v) { prm[i] = v; });
  if (RW)
    StageSlice<256>(b.rates + (size_t)t0 * RW, b.copy_rates ? b.copy_rates + (size_t)t0 * RW : nullptr, count * RW, tid,
               [&](int i, double v) { rts[i] = v; });
  __syncthreads();
  // Three serial jobs per tree that share nothing: the topology (child lists, effective branch lengths), the rate
  // matrix + eigensystem, the category rates.  Up to 64 trees per workgroup each job has a wave of its own (wave 0 the
  // eigensystems, the longest; wave 1 the topologies, wave 2 the category rates), so that a tree's set-up takes as long
  // as its eigensystem instead of the sum of the three; with 128 trees waves 0-1 take the eigensystems and waves 2-3
  // the other two jobs one after the other.
  const int lanes = trees <= 64 ? 64 : 128;       // threads per job
  const int job = tid / lanes, who = tid % lanes;  // job 0: eigensystem; 1: topology (+ rates when there is no job 2); 2: rates
  if (who < count) {
    TreeModel* mine = &b.model[t0 + who];
    if (job == 0) {
      if (b.model_reuse != nullptr) {  // (the host found every row of the call equal to the cached model's: kernels.hpp)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          mine->V[i] = b.model_reuse->V[i];
          mine->Vinv[i] = b.model_reuse->Vinv[i];
          mine->Q[i] = b.model_reuse->Q[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          mine->lambda[i] = b.model_reuse->lambda[i];
          mine->pi[i] = b.model_reuse->pi[i];
        }
      } else {
        SetupSubstitution(spec, prm + who * pc, mine);
      }
    }
    if (job == 1) SetupTopologyCore(d, par + who * (M - 1), ch + who * 2 * NI, bl + who * N, RW ? rts + who * RW : nullptr);
    if (job == (lanes == 64 ? 2 : 1)) {
      if (b.model_reuse != nullptr) {
        for (int i = 0; i < kMaxCategories; i++) {
          mine->cat_rate[i] = b.model_reuse->cat_rate[i];
          mine->cat_weight[i] = b.model_reuse->cat_weight[i];
          mine->cat_rate_deriv[i] = b.model_reuse->cat_rate_deriv[i];
        }
      } else {
        SetupSiteRates(spec, prm + who * pc, mine);
      }
    }
  }
  __syncthreads();
  // (tree 0's model for the next call; on a reuse the cache holds these very values already)
  if (b.model_cache != nullptr && b.model_reuse == nullptr && blockIdx.x == 0 && tid == 0) *b.model_cache = b.model[0];
  for (int i = tid; i < count * 2 * NI; i += step) b.children[(size_t)t0 * 2 * NI + i] = ch[i];
  for (int i = tid; i < count * N; i += step) b.branch[(size_t)t0 * N + i] = bl[i];
}

// trees per workgroup of setup_trees_lds_kernel for this batch, 0: another kernel
static int SetupLdsTrees(const BatchDims& d, const ModelSpec& spec, bool beside_traversal) {
  const size_t per_tree = SetupLdsBytesPerTree(d, spec);
  for (int trees : {128, 64, 32, 16}) {
    if (per_tree * 16 > 48 * 1024) break;  // (larger trees: a workgroup per tree)
    if (per_tree * trees > 144 * 1024) continue;
    if (trees > 16 && (!beside_traversal || d.tree_count < 4 * trees)) continue;  // (alone, or a small batch: spread it)
    return trees;
  }
  return 0;
}

bool SetupReadsHostInputs(const BatchDims& d, const ModelSpec& spec) { return SetupLdsTrees(d, spec, false) > 0; }

void LaunchSetup(const BatchDims& d, const ModelSpec& spec, const DeviceBatch& b, int,
                 hipStream_t stream, bool beside_traversal) {
  const size_t per_tree = SetupLdsBytesPerTree(d, spec);
  if (const int trees = SetupLdsTrees(d, spec, beside_traversal)) {
    if (per_tree * trees > 48 * 1024)  // (per device: a process may drive several)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(setup_trees_lds_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    const int blocks = (d.tree_count + trees - 1) / trees;
    hipLaunchKernelGGL(setup_trees_lds_kernel, dim3(blocks), dim3(256), per_tree * trees, stream, d, spec, b, trees);
    return;
  }
  const size_t large = ((size_t)2 * (d.taxon_count - 1) + 1) * sizeof(int);
  if (large <= 60 * 1024) {
    hipLaunchKernelGGL(setup_large_trees_kernel, dim3(d.tree_count), dim3(256), large, stream, d, spec, b);
    return;
  }
  const int blocks = (d.tree_count + 63) / 64;
  hipLaunchKernelGGL(setup_trees_kernel, dim3(blocks), dim3(64), 0, stream, d, spec, b);
}

// --------------------------------------------------------------------------
// Transition matrices: one thread per (tree, branch, category).

__global__ void __launch_bounds__(256)
transition_matrices_kernel(BatchDims d, DeviceBatch b, int want_gradient, int deriv_mode, int stride) {
// No FMA contraction and a fixed summation order here: see model.hpp.
#pragma clang fp contract(off)
  const int C = d.category_count, NB = d.node_count - 1;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)d.tree_count * NB * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const size_t tb = idx / C;
  const int br = (int)(tb % NB);
  const int t = (int)(tb / NB);
  const TreeModel* __restrict__ m = b.model + t;
  const double rate = m->cat_rate[c];
  const double time = b.branch[(size_t)t * d.node_count + br] * rate;
  double e[4];
#pragma unroll
  for (int k = 0; k < 4; k++) e[k] = exp(m->lambda[k] * time);
  double P[16];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) s += m->V[i * 4 + k] * e[k] * m->Vinv[k * 4 + j];
      P[i * 4 + j] = s;
    }
  double* out = b.mats + idx * stride;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      out[kMatP + i * 4 + j] = P[i * 4 + j];
      out[kMatPT + j * 4 + i] = P[i * 4 + j];
    }
    out[kMatPT + 16 + i] = 1.0;   // gap: the all-ones column BEAGLE appends
  }
  if (stride == kMatHot) return;  // (walk_hbm_cat_kernel forms dP x as r_c Q (P x))
#pragma unroll
  for (int i = 0; i < 4; i++) out[kMatDPT + 16 + i] = 0.0;  // Q 1 = 0
  // deriv_mode 1: the site-model pass, Q scaled by d r_c / d shape (fat_beagle.cpp:542-546)
  const double drate = deriv_mode ? m->cat_rate_deriv[c] : rate;
  if (want_gradient) {
    // dP/dt = P (r_c Q): the differential matrix of the reference
    // (BuildDifferentialMatrices, fat_beagle.cpp:101-111) folded into the branch's
    // transition matrix, so that  pre^T (r_c Q) post  ==  (u . a_sibling)^T dP x.
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) s += P[i * 4 + k] * (m->Q[k * 4 + j] * drate);
        out[kMatDP + i * 4 + j] = s;
        out[kMatDPT + j * 4 + i] = s;
      }
  }
}

void LaunchMatrices(const BatchDims& d, const DeviceBatch& b, int want_gradient, int deriv_mode,
                    hipStream_t stream, bool hot_only) {
  const size_t total = (size_t)d.tree_count * (d.node_count - 1) * d.category_count;
  const int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(transition_matrices_kernel, dim3(blocks), dim3(256), 0, stream, d, b,
                     want_gradient, deriv_mode, hot_only ? kMatHot : kMatStride);
}

// --------------------------------------------------------------------------
// (the walk's own arithmetic is per pattern and may use fused multiply-add; the set-up kernels above may not)
#pragma clang fp contract(fast)
// Traversal with the PLV arena in HBM.
//
// One thread owns one site pattern and walks the whole tree for it; patterns are
// independent end to end, so there is no inter-thread dependency until the final
// sums.  Arena cell (node, category, state) of a workgroup's 256 patterns is a 2 KB row
// (the arena is tiled by workgroup): a wave reads/writes 64 consecutive doubles (512 B) per access.  Transition
// matrices and the child lists are wave-uniform and come through scalar loads.
//
// Gradient pass.  With u = pre-order partial at the top of a node's two child
// branches, a_k = P_k x_k the child messages and d_k = dP_k x_k:
//     site likelihood  L = sum_c w_c sum_i u_i a0_i a1_i           (any node)
//     dL/dt_0            = sum_c w_c sum_i u_i a1_i d0_i   (Q and P commute)
//     pre(child 0)       = P_0^T (u . a1)
// so one step per internal node, in descending id order (parents first), yields
// both child-edge derivatives and both child pre-order partials.  The child's
// pre-order partial overwrites its post-order partial in place -- it is dead
// once both siblings are done -- so the arena holds n-1 PLVs per tree instead
// of the 3n-2 buffers of the reference's BEAGLE instance (fat_beagle.cpp:218-246).

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

// Read-only inputs are separate __restrict__ kernel arguments so that the
// wave-uniform ones (child lists, matrices, model) are fetched with scalar loads.
template <int C, bool GRAD, bool RESCALE>
// (four waves per SIMD: the walk hides HBM latency with occupancy.  Unrescaled gradients with up to four
// categories fit 128 registers without spilling; the rescaled gradient variant spills a little at 128 and
// is still 12 % faster on config 4 than with its natural 143 registers and three waves)
__global__ void __launch_bounds__(kHbmBlock, HBM_WAVES_EXPR)
walk_hbm_kernel(BatchDims d, int tree0, const int32_t* __restrict__ children,
                const double* __restrict__ all_mats, const TreeModel* __restrict__ models,
                const uint8_t* __restrict__ tip_states, const double* __restrict__ weights,
                double* __restrict__ arena_base, double* __restrict__ scale_base,
                double* __restrict__ part_ll, double* __restrict__ part_grad) {
  extern __shared__ double lds[];  // [waves] log-likelihoods
  constexpr int kWaves = kHbmBlock / 64;
  const int n = d.taxon_count, N = d.node_count, NI = n - 1, Ppad = d.pattern_stride;
  const int tree = tree0 + blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = blockIdx.x * kHbmBlock + tid;
  const int32_t* __restrict__ ch = children + (size_t)tree * NI * 2;
  const double* __restrict__ mats = all_mats + (size_t)tree * (N - 1) * C * kMatStride;
  const TreeModel* __restrict__ tm = models + tree;
  const uint8_t* __restrict__ tips = tip_states + p;
  // The arena is tiled by workgroup: [tree][tile][node][category][state][kHbmBlock patterns], so that the
  // rows a workgroup walks lie in one contiguous region (TLB reach) instead of Ppad doubles apart.
  const int tile_count = gridDim.x;
  double* __restrict__ arena =
      arena_base + ((size_t)blockIdx.y * tile_count + blockIdx.x) * NI * C * 4 * kHbmBlock + tid;
  // per-node, per-pattern reciprocal scale factors of the post-order pass (RESCALE && GRAD)
  double* __restrict__ inv_scale = scale_base + ((size_t)blockIdx.y * tile_count + blockIdx.x) * NI * kHbmBlock + tid;
  const double weight = weights[p];


  // A child of a step is a tip, a stored internal node, or a CHERRY (internal node over two tips).
  // Cherries are never stored: their post-order partial is the product of two tip look-ups,
  // cheaper to rebuild than to move through HBM, and their own pre-order step (two tip edges) is
  // folded into the parent's step below.  That removes every PLV transfer of about a third of
  // the internal nodes.  (Cherries are not rescaled: a product of two probabilities cannot
  // underflow, and the log-likelihood does not depend on which nodes carry a scale factor.)
  struct Child {
    int kind;  // 0 tip, 1 stored internal node, 2 cherry
    int a, b;  // cherry: its two tips (wave-uniform)
    int s, sb; // this pattern's state at the tip / at the cherry's tips
  };
  auto classify = [&](int cc) {
    Child ci{0, 0, 0, 0, 0};
    if (cc < n) {
      ci.s = tips[(size_t)cc * Ppad];
    } else {
      ci.a = __builtin_amdgcn_readfirstlane(ch[(cc - n) * 2]);
      ci.b = __builtin_amdgcn_readfirstlane(ch[(cc - n) * 2 + 1]);
      if (ci.a < n && ci.b < n) {
        ci.kind = 2;
        ci.s = tips[(size_t)ci.a * Ppad];
        ci.sb = tips[(size_t)ci.b * Ppad];
      } else {
        ci.kind = 1;
      }
    }
    return ci;
  };
  // post-order partial of internal child cc in category c
  auto fetch = [&](const Child& ci, int cc, int c, double x[4]) {
    if (ci.kind == 2) {
      const double* ma = mats + (size_t)(ci.a * C + c) * kMatStride + kMatPT + ci.s * 4;
      const double* mb = mats + (size_t)(ci.b * C + c) * kMatStride + kMatPT + ci.sb * 4;
#pragma unroll
      for (int i = 0; i < 4; i++) x[i] = ma[i] * mb[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) x[i] = arena[((size_t)((cc - n) * C + c) * 4 + i) * kHbmBlock];
    }
  };

  // ---- post-order: dest = (P0 x0) . (P1 x1) per category -------------------
  // The partial of the node processed last stays in registers (dd): when the next node has it as a
  // child -- ids are in post-order, so that is the rule -- it is taken from there instead of HBM, and a
  // log-likelihood-only walk then never writes it at all (the store is deferred until a node turns up
  // that does not consume it).
  double log_scale = 0.0, site = 0.0;
  double dd[C][4];
  int last = -1;         // node whose partial dd holds (wave-uniform)
  bool unsaved = false;  // ... and which has not been written to the arena (log-likelihood-only walks)
  for (int node = n; node < N; ++node) {
    const int c0 = __builtin_amdgcn_readfirstlane(ch[(node - n) * 2]);
    const int c1 = __builtin_amdgcn_readfirstlane(ch[(node - n) * 2 + 1]);
    if (c0 < n && c1 < n && node != N - 1) continue;  // a cherry: rebuilt where it is used
    const Child k0 = classify(c0), k1 = classify(c1);
    if (!GRAD && unsaved && c0 != last && c1 != last) {
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int i = 0; i < 4; i++) arena[((size_t)((last - n) * C + c) * 4 + i) * kHbmBlock] = dd[c][i];
    }
#pragma unroll
    for (int c = 0; c < C; c++) {
      double A[4], B[4];
      const double* m0 = mats + (size_t)(c0 * C + c) * kMatStride;
      const double* m1 = mats + (size_t)(c1 * C + c) * kMatStride;
      if (k0.kind == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) A[i] = m0[kMatPT + k0.s * 4 + i];
      } else {
        double x[4];
        if (c0 == last) {
#pragma unroll
          for (int i = 0; i < 4; i++) x[i] = dd[c][i];
        } else {
          fetch(k0, c0, c, x);
        }
        MatVec(m0 + kMatP, x, A);
      }
      if (k1.kind == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) B[i] = m1[kMatPT + k1.s * 4 + i];
      } else {
        double x[4];
        if (c1 == last) {
#pragma unroll
          for (int i = 0; i < 4; i++) x[i] = dd[c][i];
        } else {
          fetch(k1, c1, c, x);
        }
        MatVec(m1 + kMatP, x, B);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) dd[c][i] = A[i] * B[i];
    }
    last = node;
    if (RESCALE) {
      // BEAGLE manual scaling: per pattern, max over categories and states.
      double mx = 0.0;
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int i = 0; i < 4; i++) mx = fmax(mx, dd[c][i]);
      if (mx == 0.0) mx = 1.0;
      const double inv = 1.0 / mx;
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int i = 0; i < 4; i++) dd[c][i] *= inv;
      log_scale += log(mx);
      if (GRAD) inv_scale[(size_t)(node - n) * kHbmBlock] = inv;
    }
    if (node == N - 1) {
#pragma unroll
      for (int c = 0; c < C; c++)
        site += tm->cat_weight[c] * (tm->pi[0] * dd[c][0] + tm->pi[1] * dd[c][1] +
                                     tm->pi[2] * dd[c][2] + tm->pi[3] * dd[c][3]);
    } else if (GRAD) {
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int i = 0; i < 4; i++) arena[((size_t)((node - n) * C + c) * 4 + i) * kHbmBlock] = dd[c][i];
    } else {
      unsaved = true;
    }
  }
  const double ll = weight * (log(site) + log_scale);

  // ---- pre-order + edge derivatives ---------------------------------------
  if (GRAD) {
    // one gradient row per wave, straight in global memory (summed by the reduce kernel in a fixed order): rows
    // in LDS would cost 4 N doubles per workgroup -- 64 KB at a thousand taxa, i.e. two workgroups per CU
    double* __restrict__ my_row = part_grad + (((size_t)tree * gridDim.x + blockIdx.x) * kWaves + wave) * N;
    // The pre-order partial of the child that is processed next (node - 1: ids are in post-order) is
    // handed over through a thread-private LDS column instead of the HBM arena: one store and one load
    // of a PLV less per such node.
    constexpr bool kForward = C <= 4;                       // 32 KB of LDS per workgroup at C = 4
    double* __restrict__ fwd = lds + kWaves + tid;          // [C][4][kHbmBlock]
    bool u_forwarded = false;
    for (int node = N - 1; node >= n; --node) {
      const int c0 = __builtin_amdgcn_readfirstlane(ch[(node - n) * 2]);
      const int c1 = __builtin_amdgcn_readfirstlane(ch[(node - n) * 2 + 1]);
      if (c0 < n && c1 < n && node != N - 1) continue;  // a cherry: handled inside its parent's step
      const Child k0 = classify(c0), k1 = classify(c1);
      const bool tip0 = k0.kind == 0, tip1 = k1.kind == 0;
      double num0 = 0.0, num1 = 0.0, den = 0.0;
      double numa0 = 0.0, numb0 = 0.0, numa1 = 0.0, numb1 = 0.0;  // tip edges of cherry children
      // The cherry's own step with q = its pre-order partial: dL/dt of its two tip edges.  Its site
      // likelihood sum_i q_i x_i equals this step's `den`, so only the numerators are new.
      auto cherry_edges = [&](const Child& ci, int c, double wc, const double q[4], double& na, double& nb) {
        const double* ma = mats + (size_t)(ci.a * C + c) * kMatStride;
        const double* mb = mats + (size_t)(ci.b * C + c) * kMatStride;
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const double aa = ma[kMatPT + ci.s * 4 + i], da = ma[kMatDPT + ci.s * 4 + i];
          const double ab = mb[kMatPT + ci.sb * 4 + i], db = mb[kMatDPT + ci.sb * 4 + i];
          sa += q[i] * (ab * da);
          sb += q[i] * (aa * db);
        }
        na += wc * sa;
        nb += wc * sb;
      };
      // One category of the step: accumulates the three site sums and returns the
      // two child pre-order partials (only meaningful for internal children).
      auto category_step = [&](int c, double q0[4], double q1[4]) {
        double U[4], A0[4], D0[4], A1[4], D1[4];
        if (node == N - 1) {
#pragma unroll
          for (int i = 0; i < 4; i++) U[i] = tm->pi[i];
        } else if (kForward && u_forwarded) {
#pragma unroll
          for (int i = 0; i < 4; i++) U[i] = fwd[(c * 4 + i) * kHbmBlock];
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) U[i] = arena[((size_t)((node - n) * C + c) * 4 + i) * kHbmBlock];
        }
        const double* m0 = mats + (size_t)(c0 * C + c) * kMatStride;
        const double* m1 = mats + (size_t)(c1 * C + c) * kMatStride;
        if (tip0) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            A0[i] = m0[kMatPT + k0.s * 4 + i];
            D0[i] = m0[kMatDPT + k0.s * 4 + i];
          }
        } else {
          double x[4];
          fetch(k0, c0, c, x);
          MatVec(m0 + kMatP, x, A0);
          MatVec(m0 + kMatDP, x, D0);
        }
        if (tip1) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            A1[i] = m1[kMatPT + k1.s * 4 + i];
            D1[i] = m1[kMatDPT + k1.s * 4 + i];
          }
        } else {
          double x[4];
          fetch(k1, c1, c, x);
          MatVec(m1 + kMatP, x, A1);
          MatVec(m1 + kMatDP, x, D1);
        }
        double UA0[4], UA1[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          UA0[i] = U[i] * A0[i];
          UA1[i] = U[i] * A1[i];
        }
        const double wc = tm->cat_weight[c];
        den += wc * (UA1[0] * A0[0] + UA1[1] * A0[1] + UA1[2] * A0[2] + UA1[3] * A0[3]);
        num0 += wc * (UA1[0] * D0[0] + UA1[1] * D0[1] + UA1[2] * D0[2] + UA1[3] * D0[3]);
        num1 += wc * (UA0[0] * D1[0] + UA0[1] * D1[1] + UA0[2] * D1[2] + UA0[3] * D1[3]);
        if (!tip0) MatVecT(m0 + kMatP, UA1, q0);
        if (!tip1) MatVecT(m1 + kMatP, UA0, q1);
        if (k0.kind == 2) cherry_edges(k0, c, wc, q0, numa0, numb0);
        if (k1.kind == 2) cherry_edges(k1, c, wc, q1, numa1, numb1);
      };
      // Rescaling of the pre-order partials.  BEAGLE rescales each pre-order partial by its
      // own maximum (scaleWrite on the pre-order ops, fat_beagle.cpp:362-363); the factor
      // cancels in num/den, so any positive per-pattern factor gives the same derivatives.
      // Dividing by the parent's POST-order factor m_node keeps pre(X) * exp(S(X) - S(root))
      // (S = summed log factors of a subtree), whose product with the scaled post-order
      // partial of X sums to the scaled site likelihood, i.e. stays O(1) -- and the factor
      // is known before the category loop, so nothing has to be held across categories.
      const double step_inv = RESCALE ? inv_scale[(size_t)(node - n) * kHbmBlock] : 1.0;
      // In place, one category at a time: cell (child, c) is read (as the child's
      // post-order partial) before it is overwritten with its pre-order partial.
      // (A forwarded partial takes LDS slot c after this node's own U, read from the same slot, is used.)
      const bool fwd0 = kForward && k0.kind == 1 && c0 == node - 1;
      const bool fwd1 = kForward && k1.kind == 1 && c1 == node - 1;
#pragma unroll HBM_PRE_UNROLL
      for (int c = 0; c < C; c++) {
        double q0[4], q1[4];
        category_step(c, q0, q1);
        if (k0.kind == 1) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const double v = RESCALE ? q0[i] * step_inv : q0[i];
            if (fwd0) fwd[(c * 4 + i) * kHbmBlock] = v;
            else arena[((size_t)((c0 - n) * C + c) * 4 + i) * kHbmBlock] = v;
          }
        }
        if (k1.kind == 1) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const double v = RESCALE ? q1[i] * step_inv : q1[i];
            if (fwd1) fwd[(c * 4 + i) * kHbmBlock] = v;
            else arena[((size_t)((c1 - n) * C + c) * 4 + i) * kHbmBlock] = v;
          }
        }
      }
      u_forwarded = fwd0 || fwd1;
      const double scale = weight / den;
      const double g0 = WaveSum(num0 * scale);
      const double g1 = WaveSum(num1 * scale);
      if (lane == 0) {
        my_row[c0] = g0;
        my_row[c1] = g1;
      }
      if (k0.kind == 2) {
        const double ga = WaveSum(numa0 * scale), gb = WaveSum(numb0 * scale);
        if (lane == 0) {
          my_row[k0.a] = ga;
          my_row[k0.b] = gb;
        }
      }
      if (k1.kind == 2) {
        const double ga = WaveSum(numa1 * scale), gb = WaveSum(numb1 * scale);
        if (lane == 0) {
          my_row[k1.a] = ga;
          my_row[k1.b] = gb;
        }
      }
    }
  }

  // ---- block-level sums, fixed order --------------------------------------
  const double wll = WaveSum(ll);
  double* ll_slots = lds;
  if (lane == 0) ll_slots[wave] = wll;
  __syncthreads();
  const int tiles = gridDim.x;
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < kWaves; w++) s += ll_slots[w];
    part_ll[(size_t)tree * tiles + blockIdx.x] = s;
  }
}

size_t HbmArenaBytesPerTree(const BatchDims& d) {
  return (size_t)(d.taxon_count - 1) * d.category_count * 4 * d.pattern_stride * sizeof(double);
}

template <int C>
static void LaunchWalkHbmC(const BatchDims& d, const DeviceBatch& b, int tree0, int chunk,
                           int want_gradient, int rescaling, hipStream_t stream) {
  const dim3 grid(HbmTiles(d.pattern_count), chunk), block(kHbmBlock);
  // log-likelihood slots, and (gradients, up to four categories) the pre-order forwarding columns
  const size_t lds = ((size_t)(kHbmBlock / 64) + (want_gradient && C <= 4 ? (size_t)C * 4 * kHbmBlock : 0)) * sizeof(double);
  if (want_gradient) {
    if (rescaling)
      hipLaunchKernelGGL((walk_hbm_kernel<C, true, true>), grid, block, lds, stream, d, tree0, b.children, b.mats, b.model,
                         b.tip_states, b.weights, b.arena, b.scale_arena, b.part_ll, b.part_grad);
    else
      hipLaunchKernelGGL((walk_hbm_kernel<C, true, false>), grid, block, lds, stream, d, tree0, b.children, b.mats, b.model,
                         b.tip_states, b.weights, b.arena, b.scale_arena, b.part_ll, b.part_grad);
  } else {
    if (rescaling)
      hipLaunchKernelGGL((walk_hbm_kernel<C, false, true>), grid, block, lds, stream, d, tree0, b.children, b.mats, b.model,
                         b.tip_states, b.weights, b.arena, b.scale_arena, b.part_ll, b.part_grad);
    else
      hipLaunchKernelGGL((walk_hbm_kernel<C, false, false>), grid, block, lds, stream, d, tree0, b.children, b.mats, b.model,
                         b.tip_states, b.weights, b.arena, b.scale_arena, b.part_ll, b.part_grad);
  }
}

void LaunchWalkHbm(const BatchDims& d, const DeviceBatch& b, int tree0, int chunk, int want_gradient,
                   int rescaling, hipStream_t stream, int deriv_mode) {
  if (HbmCatKernelApplies(d)) {
    LaunchWalkHbmCat(d, b, tree0, chunk, want_gradient, rescaling, deriv_mode, stream);  // walk_hbm_cat.hip
    return;
  }
  switch (d.category_count) {
    case 1: LaunchWalkHbmC<1>(d, b, tree0, chunk, want_gradient, rescaling, stream); break;
    case 2: LaunchWalkHbmC<2>(d, b, tree0, chunk, want_gradient, rescaling, stream); break;
    case 3: LaunchWalkHbmC<3>(d, b, tree0, chunk, want_gradient, rescaling, stream); break;
    case 4: LaunchWalkHbmC<4>(d, b, tree0, chunk, want_gradient, rescaling, stream); break;
    case 5: LaunchWalkHbmC<5>(d, b, tree0, chunk, want_gradient, rescaling, stream); break;
    case 6: LaunchWalkHbmC<6>(d, b, tree0, chunk, want_gradient, rescaling, stream); break;
    case 7: LaunchWalkHbmC<7>(d, b, tree0, chunk, want_gradient, rescaling, stream); break;
    case 8: LaunchWalkHbmC<8>(d, b, tree0, chunk, want_gradient, rescaling, stream); break;
    default: break;  // rejected at engine creation
  }
}

const char* WalkHbmKernelName(int category_count, int, int) {
  BatchDims d{};
  d.category_count = category_count;
  return HbmCatKernelApplies(d) ? "walk_hbm_cat_kernel" : "walk_hbm_kernel";
}

// --------------------------------------------------------------------------
// Final per-tree sums over pattern tiles, fixed order.

__global__ void __launch_bounds__(256)
reduce_tiles_kernel(BatchDims d, DeviceBatch b, int tiles, int grad_rows, int want_gradient, ReduceDone done,
                    const uint8_t* __restrict__ skip, int first_tree) {
  const int N = d.node_count;
  const int per_tree = want_gradient ? N + 1 : 1;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (size_t)(d.tree_count - first_tree) * per_tree) {
    const int t = first_tree + (int)(idx / per_tree), e = (int)(idx % per_tree);
    if (skip != nullptr && skip[t]) {
      // (the traversal wrote this tree's results itself: walk_pipe_kernel's whole-tree units)
    } else if (e == per_tree - 1) {
      double s = 0.0;
      for (int k = 0; k < tiles; k++) s += b.part_ll[(size_t)t * tiles + k];
      b.out_ll[t] = s;
    } else {
      double s = 0.0;
      for (int k = 0; k < grad_rows; k++) s += b.part_grad[((size_t)t * grad_rows + k) * N + e];
      // The root's slot carries the site-model gradient of kernels that produce it in the same pass
      // (walk_lds_kernel); other kernels leave 0 there.
      if (e == N - 1 && b.out_site != nullptr) b.out_site[t] = s;
      // Root has no branch; for unrooted trees the node that re-uses the old root id
      // is the fixed node whose gradient is pinned to 0 (reference fat_beagle.cpp:148,553).
      if (e == N - 1 || (!d.rooted && e == N - 2)) s = 0.0;
      b.out_grad[(size_t)t * N + e] = s;
    }
  }
  if (done.flag != nullptr) {
    // every thread's results are on their way to host memory before its workgroup is counted; the workgroup that
    // counts last stores the flag behind them (the pattern of a reduction's last block, at system scope)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      if (atomicAdd(done.counter, 1) == (int)gridDim.x - 1) {
        *done.counter = 0;
        __threadfence_system();
        __hip_atomic_store(done.flag, done.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

void LaunchReduce(const BatchDims& d, const DeviceBatch& b, int tiles, int want_gradient,
                  hipStream_t stream, int grad_rows, ReduceDone done, const uint8_t* skip, int first_tree) {
  if (grad_rows <= 0) grad_rows = tiles;
  // (trees below first_tree have their results already: walk_pipe_kernel's whole-tree units of a one-class launch;
  // at least one workgroup, which stores a blocking call's completion flag)
  const size_t total = (size_t)(d.tree_count - first_tree) * (want_gradient ? d.node_count + 1 : 1);
  const int blocks = std::max(1, (int)((total + 255) / 256));
  hipLaunchKernelGGL(reduce_tiles_kernel, dim3(blocks), dim3(256), 0, stream, d, b, tiles, grad_rows,
                     want_gradient, done, skip, first_tree);
}

// Completion flag of a blocking call's chunk: stored behind the kernels that wrote the chunk's results into pinned
// host memory (a kernel boundary on one stream orders it behind their stores), polled by the host.
__global__ void signal_kernel(unsigned long long* flag, unsigned long long value) {
  __threadfence_system();
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void LaunchSignal(unsigned long long* flag, unsigned long long value, hipStream_t stream) {
  hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, stream, flag, value);
}

}  // namespace bito_amd
