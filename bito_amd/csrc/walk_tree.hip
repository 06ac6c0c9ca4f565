// walk_tree.hip -- LDS-resident traversal, two waves per SIMD (gfx950).
//
// Second generation of walk_lds.hip.  PMC and ablation runs on that kernel
// (profiles/r1_v2_lds_pmc.json) showed it to be bound by the
// instruction issue rate of its single wave per SIMD (one instruction per ~5 cycles,
// MFMA pipe 15 % busy) and, below that, by every wave streaming its own copy of the
// matrix images out of L2.  This kernel changes the work split:
//
//   * a workgroup is 8 waves (two per SIMD, so one wave's waits and dependency stalls
//     are filled by its sibling) and serves ONE tree for several pattern tiles;
//   * the tree's P and dP operand images (all 2n-2 branches, 1 KB each) are staged in
//     LDS once per workgroup and shared by all waves -- the walk issues no global
//     loads at all;  the P^T operand of the pre-order step is the P image with its
//     row/column lane bits exchanged (one ds_bpermute per dword);
//   * each wave owns ONE group image per node (16/C patterns), so the per-wave arena is
//     (n-2) x 512 B and nothing is an array: everything is in scalar registers.
//
// Arithmetic, lane layout (lane = 16 state + 4 block + pattern, block = rate category)
// and the in-place pre-order scheme are those of walk_lds.hip / kernels.hip.
#include <type_traits>

#include "kernels.hpp"

namespace bito_amd {

namespace {

constexpr size_t kTreeLdsBudget = 160 * 1024;

__device__ __forceinline__ double Mfma4(double a, double x) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, x, 0.0, 0, 0, 0);
}

__device__ __forceinline__ double TipOp(int mask, int st) {
  const int hi = (0 - ((mask >> st) & 1)) & 0x3FF00000;
  return __hiloint2double(hi, 0);
}

template <int kCtrl>
__device__ __forceinline__ double Dpp(double v) {
  const long long bits = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, kCtrl, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), kCtrl, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double RowSum(double v) {  // sum over each 16-lane row
  v += Dpp<0x128>(v);  // row_ror:8
  v += Dpp<0x124>(v);  // row_ror:4
  v += Dpp<0x122>(v);  // row_ror:2
  v += Dpp<0x121>(v);  // row_ror:1
  return v;
}

__device__ __forceinline__ double PairRows2(double v) {  // rows r0 r1 r2 r3 -> r0+r1 (x2), r2+r3 (x2)
  const long long bits = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32);
  const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __builtin_bit_cast(double, ((long long)h[0] << 32) | l[0]) +
         __builtin_bit_cast(double, ((long long)h[1] << 32) | l[1]);
}

__device__ __forceinline__ double MergeHalves2(double a, double b) {  // lanes <32: a[l]+a[l+32]; >=32: b
  const long long ab = __builtin_bit_cast(long long, a), bb = __builtin_bit_cast(long long, b);
  const auto l = __builtin_amdgcn_permlane32_swap((unsigned)ab, (unsigned)bb, false, false);
  const auto h = __builtin_amdgcn_permlane32_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
  return __builtin_bit_cast(double, ((long long)h[0] << 32) | l[0]) +
         __builtin_bit_cast(double, ((long long)h[1] << 32) | l[1]);
}

// value of lane `src` (per-lane index), 64-bit
__device__ __forceinline__ double Permute(double v, int src_byte_index) {
  const long long bits = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_ds_bpermute(src_byte_index, (int)bits);
  const int hi = __builtin_amdgcn_ds_bpermute(src_byte_index, (int)(bits >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

}  // namespace

template <int C, int W, bool GRAD, int TABS>
__global__ void __launch_bounds__(W * 64, 2)
walk_tree_kernel(BatchDims d, int tiles, int tiles_per_block, int blocks_per_tree, int units,
                 const int32_t* __restrict__ children, const double* __restrict__ images,
                 const TreeModel* __restrict__ models, const uint8_t* __restrict__ tip_states,
                 const double* __restrict__ weights, double* __restrict__ part_ll,
                 double* __restrict__ part_grad) {
  extern __shared__ double lds[];
  constexpr int PG = 16 / C;   // patterns per wave (one group image)
  constexpr int PB = W * PG;   // patterns per tile
  const int n = d.taxon_count, N = d.node_count, NI = n - 1, NB = N - 1, slots = n - 2, Ppad = d.pattern_stride;

  // XCD-aware unit order (workgroup b runs on XCD b % 8): the workgroups of one tree share an L2
  int unit = blockIdx.x;
  {
    const int per = units / 8, rem = units % 8, x = unit % 8, q = unit / 8;
    unit = x * per + (x < rem ? x : rem) + q;
  }
  const int tree = unit / blocks_per_tree;
  const int tile0 = (unit % blocks_per_tree) * tiles_per_block;
  const int tile1 = tile0 + tiles_per_block < tiles ? tile0 + tiles_per_block : tiles;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int st = lane >> 4, blk = (lane >> 2) & 3, pj = lane & 3;
  const int cat = blk % C, sub = blk / C;

  // LDS: tree images [NB][2][64] | per-wave arenas [W][slots][64] | tip masks [n][PB] | grad rows | ll
  double* limg = lds + lane;                                   // image (branch, which) at (branch*2+which)*64
  double* arena = lds + (size_t)NB * 128 + (size_t)wave * slots * 64 + lane;  // cell(slot) at slot*64
  uint8_t* tipbuf = reinterpret_cast<uint8_t*>(lds + (size_t)NB * 128 + (size_t)W * slots * 64);
  double* grad_rows = lds + (size_t)NB * 128 + (size_t)W * slots * 64 + (n * PB + 7) / 8;
  double* ll_slots = grad_rows + W * N;

  const int32_t* __restrict__ ch = children + (size_t)tree * NI * 2;
  const double* __restrict__ gimg = images + (size_t)tree * NB * kImgStride;
  const TreeModel* __restrict__ tm = models + tree;

  // child list in lane tables (see walk_lds.hip)
  const int tab0 = lane < 2 * NI ? ch[lane] : 0;
  const int tab1 = TABS > 1 && 64 + lane < 2 * NI ? ch[64 + lane] : 0;
  const int tab2 = TABS > 1 && 128 + lane < 2 * NI ? ch[128 + lane] : 0;
  auto child = [&](int idx) -> int {
    if (TABS == 1) return __builtin_amdgcn_readlane(tab0, idx);
    const int t = idx < 64 ? tab0 : (idx < 128 ? tab1 : tab2);
    return __builtin_amdgcn_readlane(t, idx & 63);
  };

  // stage the tree's P and dP images (the global layout also carries P^T: skipped)
  for (int q = tid; q < NB * 128; q += W * 64) {
    const int br = q >> 7, r = q & 127;
    lds[q] = gimg[(size_t)br * kImgStride + (r < 64 ? 2 * r : 2 * (r - 64) + 1)];  // global layout: (P, dP) pairs per lane
  }
  const double pi_st = tm->pi[st];
  const double w_cat = tm->cat_weight[cat];
  // lane that holds P[k][i] when this lane holds P[i][k]: exchange lane bits (0,1) with (4,5)
  const int tr_index = (((lane & 3) << 4) | (lane & 12) | (lane >> 4)) << 2;

#define CELL(node_id) arena[(size_t)((node_id) - n) * 64]
#define IMG(br, which) limg[(size_t)((br) * 2 + (which)) * 64]

  for (int tile = tile0; tile < tile1; tile++) {
    __syncthreads();  // images staged / previous tile's results consumed
    for (int q = tid; q < n * PB; q += W * 64) {
      const int sym = tip_states[(size_t)(q / PB) * Ppad + tile * PB + (q % PB)];
      tipbuf[q] = (uint8_t)(sym < 4 ? 1 << sym : 15);
    }
    if (GRAD)
      for (int q = tid; q < W * N; q += W * 64) grad_rows[q] = 0.0;
    __syncthreads();

    const int loc = wave * PG + sub * 4 + pj;  // this lane's pattern inside the tile
    const double wgt = weights[tile * PB + loc];
    auto operand = [&](int c) -> double { return c < n ? TipOp(tipbuf[c * PB + loc], st) : CELL(c); };

    // ---------------- post-order ----------------------------------------------
    double res = 0.0;
    {
      bool forward = false;
      for (int k = 0; k < NI; k++) {
        const int node = n + k;
        const int c0 = child(2 * k), c1 = child(2 * k + 1);
        const double x0 = operand(c0);
        const double x1 = forward ? res : operand(c1);
        const double a0 = Mfma4(IMG(c0, 0), x0);
        const double a1 = Mfma4(IMG(c1, 0), x1);
        res = a0 * a1;
        if (k < NI - 1) {
          CELL(node) = res;
          forward = child(2 * k + 3) == node;  // next step's second child is this node
        }
      }
    }

    // ---------------- root: site likelihood -----------------------------------
    double L = res * (pi_st * w_cat);
    L += __shfl_xor(L, 16);
    L += __shfl_xor(L, 32);
    if (C >= 2) L += __shfl_xor(L, 4);
    if (C == 4) L += __shfl_xor(L, 8);
    const double ll_lane = (st == 0 && cat == 0) ? wgt * log(L) : 0.0;
    const double coef = w_cat * (wgt / L);

    // ---------------- pre-order + edge derivatives ----------------------------
    if (GRAD) {
      double* my_row = grad_rows + wave * N;
      double U = pi_st;       // pre-order partial of the root (fat_beagle.cpp:327-336)
      bool u_in_regs = true;
      double ps0 = 0.0, ps1 = 0.0;
      int pc0 = N - 1, pc1 = N - 1;
      for (int node = N - 1; node >= n; --node) {
        const int c0 = child(2 * (node - n)), c1 = child(2 * (node - n) + 1);
        const double x0 = operand(c0), x1 = operand(c1);
        if (!u_in_regs) U = CELL(node);
        const double p0 = IMG(c0, 0), q0 = IMG(c0, 1), p1 = IMG(c1, 0), q1 = IMG(c1, 1);
        const double a0 = Mfma4(p0, x0);
        const double a1 = Mfma4(p1, x1);
        const double d0 = Mfma4(q0, x0);
        const double d1 = Mfma4(q1, x1);
        // previous step's edge sums: the cross-lane chain runs under the products above
        {
          const double sm = RowSum(PairRows2(MergeHalves2(ps0, ps1)));
          if ((lane & 31) == 0) my_row[lane == 0 ? pc0 : pc1] = sm;
        }
        const double ua1 = U * a1, ua0 = U * a0;
        ps0 = coef * (ua1 * d0);
        ps1 = coef * (ua0 * d1);
        pc0 = c0;
        pc1 = c1;
        if (c0 >= n) CELL(c0) = Mfma4(Permute(p0, tr_index), ua1);
        u_in_regs = false;
        if (c1 >= n) {
          U = Mfma4(Permute(p1, tr_index), ua0);
          CELL(c1) = U;
          u_in_regs = c1 == node - 1;
        }
      }
      const double sm = RowSum(PairRows2(MergeHalves2(ps0, ps1)));
      if ((lane & 31) == 0) my_row[lane == 0 ? pc0 : pc1] = sm;
    }

    // ---------------- tile sums, fixed order ----------------------------------
    double wll = ll_lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wll += __shfl_xor(wll, o);
    if (lane == 0) ll_slots[wave] = wll;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int w = 0; w < W; w++) s += ll_slots[w];
      part_ll[(size_t)tree * tiles + tile] = s;
    }
    if (GRAD) {
      double* out = part_grad + ((size_t)tree * tiles + tile) * N;
      for (int e = tid; e < N; e += W * 64) {
        double s = 0.0;
        for (int w = 0; w < W; w++) s += grad_rows[w * N + e];
        out[e] = s;
      }
    }
  }
#undef CELL
#undef IMG
}

static size_t TreeLdsBytes(const BatchDims& d, int W) {
  const int PG = 16 / d.category_count, PB = W * PG, n = d.taxon_count;
  const size_t doubles = (size_t)(d.node_count - 1) * 128 + (size_t)W * (n - 2) * 64 + ((size_t)n * PB + 7) / 8 +
                         (size_t)W * d.node_count + W;
  return doubles * sizeof(double);
}

TreePlan PlanTree(const BatchDims& d) {
  TreePlan plan{0, 0, 0, 0, 0, 0};
  const int C = d.category_count;
  if (C != 1 && C != 2 && C != 4) return plan;
  if (d.taxon_count < 3 || d.taxon_count > 80) return plan;
  const int W = 8;
  if (TreeLdsBytes(d, W) > kTreeLdsBudget) return plan;
  plan.waves = W;
  plan.patterns_per_tile = W * (16 / C);
  plan.tiles = (d.pattern_count + plan.patterns_per_tile - 1) / plan.patterns_per_tile;
  // a workgroup serves several tiles of its tree (images are staged once per workgroup)
  // while keeping a few thousand workgroups in flight
  long total = (long)d.tree_count * plan.tiles;
  int tpb = (int)(total / 4096);
  if (tpb < 1) tpb = 1;
  if (tpb > plan.tiles) tpb = plan.tiles;
  plan.tiles_per_block = tpb;
  plan.blocks_per_tree = (plan.tiles + tpb - 1) / tpb;
  plan.lds_bytes = TreeLdsBytes(d, W);
  return plan;
}

template <int C>
static void LaunchWalkTreeC(const BatchDims& d, const DeviceBatch& b, const TreePlan& plan, int want_gradient,
                            hipStream_t stream) {
  constexpr int W = 8;
  const int units = d.tree_count * plan.blocks_per_tree;
  const dim3 grid(units), block(W * 64);
  auto launch = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kTreeLdsBudget);
    hipLaunchKernelGGL(kern, grid, block, plan.lds_bytes, stream, d, plan.tiles, plan.tiles_per_block,
                       plan.blocks_per_tree, units, b.children, b.images, b.model, b.tip_states, b.weights,
                       b.part_ll, b.part_grad);
  };
  const bool one_tab = d.taxon_count <= 33;
  if (want_gradient) {
    if (one_tab) launch(walk_tree_kernel<C, W, true, 1>);
    else launch(walk_tree_kernel<C, W, true, 3>);
  } else {
    if (one_tab) launch(walk_tree_kernel<C, W, false, 1>);
    else launch(walk_tree_kernel<C, W, false, 3>);
  }
}

void LaunchWalkTree(const BatchDims& d, const DeviceBatch& b, const TreePlan& plan, int want_gradient,
                    hipStream_t stream) {
  switch (d.category_count) {
    case 1: LaunchWalkTreeC<1>(d, b, plan, want_gradient, stream); break;
    case 2: LaunchWalkTreeC<2>(d, b, plan, want_gradient, stream); break;
    case 4: LaunchWalkTreeC<4>(d, b, plan, want_gradient, stream); break;
    default: break;
  }
}

}  // namespace bito_amd
