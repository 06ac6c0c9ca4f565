// gs_kernels.hip -- general-state-count likelihood path for gfx950 (BASELINE config 5: the 61-state
// codon model; any state count up to 64 runs through the same kernels).
//
// Same pipeline as the 4-state path (kernels.hip), same reference rows:
//   gs_setup_kernel     topology + rate matrix + eigendecomposition of the symmetrised matrix
//                       (src/substitution_model.cpp:120-187, src/site_model.cpp:37-62); one wave
//                       per tree, the 64x64 Jacobi iteration runs out of LDS
//   gs_matrices_kernel  P(t r_c) = V exp(Lambda t r_c) V^-1 and dP/dt for every branch and
//                       category (beagleUpdateTransitionMatrices, src/fat_beagle.cpp:315-325),
//                       written as MFMA A-operand images (internal branches) or transposed tip
//                       look-up tables (leaf branches)
//   gs_walk_kernel      post-order partials + root log-likelihood (src/fat_beagle.cpp:54-68) and the
//                       fused pre-order pass + edge derivatives (:138-160); a wave owns 16 site
//                       patterns of one tree and walks the whole tree; the states x states
//                       contraction is v_mfma_f64_16x16x4 with the partial-likelihood vector kept in
//                       the instruction's own D/B register layout from step to step
// The reference has no model with more than four states; the codon model ("GY94") is defined by
// this build (include/bito_amd.h) and checked against the CPU restatement the tests hold.
//
// FP64 throughout, no atomics, every sum in a fixed order.
#include "kernels.hpp"
#include "wave_sums.hpp"

// build-time knobs of the traversal kernel (scripts/build_gs_variants.sh measures the alternatives)
#ifndef GS_SCHED_BARRIER
#define GS_SCHED_BARRIER 1
#endif
#ifndef GS_WAVES
#define GS_WAVES 2  // waves per SIMD the register allocation is held to
#endif
// (the traversal's variants; GS_EXP_*: timing-only ablations, their results mean nothing -- BENCH_ABLATION=1 for bench.py)
#ifndef GS_SPARSE_DP
#define GS_SPARSE_DP 1
#endif
#ifndef GS_DP_COLUMN
#define GS_DP_COLUMN 1  // 1: the sparse dP terms with a column's list in registers (round 6); 0: round 3's loop, lists in LDS
#endif
#ifndef GS_SIBLING_EARLY
#define GS_SIBLING_EARLY 0
#endif
#ifndef GS_X_EARLY
#define GS_X_EARLY 0
#endif
#ifndef GS_ASM_FETCH
#define GS_ASM_FETCH 1
#endif
#ifndef GS_EXP_NOLOAD
#define GS_EXP_NOLOAD 0
#endif
#ifndef GS_EXP_NOSTORE
#define GS_EXP_NOSTORE 0
#endif
#ifndef GS_EXP_NOMFMA
#define GS_EXP_NOMFMA 0
#endif
#ifndef GS_OWN_EDGE
#define GS_OWN_EDGE 1  // 0: round 2's pre-order pass (a child's edge derivative in its parent's step, from the child's re-read post-order partial)
#endif
#ifndef GS_WG_WAVES
#define GS_WG_WAVES 4  // waves (16-pattern tiles of one tree) per workgroup of the traversal kernel (8: measured slower)
#endif

namespace bito_amd {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

namespace {

constexpr int kLd = 65;  // LDS row stride of the 64 x 64 work matrices (odd: column walks hit all banks)

// Standard genetic code, TCAG order; bito's nucleotides are A,C,G,T = 0..3.
__device__ const char kCodeTCAG[65] = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";

__device__ __forceinline__ double WaveMax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double WaveSum64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// MFMA A-operand image position of matrix element (row, col); see the matrices kernel below.
__device__ __forceinline__ int GsImageIndex(int row, int col) {
  const int mb = row >> 4, ii = row & 15, ks = col >> 2, kq = col & 3;
  return ((mb * 8 + (ks >> 1)) * 64 + 16 * kq + ii) * 2 + (ks & 1);
}

}  // namespace

// --------------------------------------------------------------------------------------------
// Set-up, part 1: one wave per tree -- topology, rate matrix, its symmetrised form.  The arithmetic
// order is fixed and documented in DESIGN.md section 3 (no FMA contraction): errors in P(t) are
// coherent across site patterns.  Trees whose parameter row equals an earlier tree's share that
// tree's model record (model_index, built by the host while it validates the rows).

__global__ void __launch_bounds__(64)
gs_model_kernel(BatchDims d, ModelSpec spec, DeviceBatch b, const int32_t* __restrict__ model_index,
                double* __restrict__ gs_model, int topology_only) {
#pragma clang fp contract(off)
  __shared__ double A[64 * kLd];
  __shared__ double pi[64], sq[64], rowsum[64];
  __shared__ int codon[64];  // n1 | n2<<2 | n3<<4 | aa<<8
  const int t = blockIdx.x, lane = threadIdx.x;
  const int S = spec.state_count;
  if (lane == 0) SetupTopology(d, b, t);
  if (topology_only || model_index[t] != t) return;  // (topology_only: the models of the batch before still stand)
  const double* __restrict__ row = b.params + (size_t)t * spec.param_count;
  double* __restrict__ out = gs_model + (size_t)t * kGsModelStride;

  for (int j = 0; j < 64; j++) A[lane * kLd + j] = 0.0;
  rowsum[lane] = 0.0;
  double mypi = 0.0;
  if (spec.substitution == kGY94) {
    // sense codons in lexicographic ACGT order: lane = raw codon, state = rank among sense codons
    {
      const int a = lane >> 4, bq = (lane >> 2) & 3, c = lane & 3;
      const int to_tcag[4] = {2, 1, 3, 0};
      const char aa = kCodeTCAG[16 * to_tcag[a] + 4 * to_tcag[bq] + to_tcag[c]];
      const bool sense = aa != '*';
      const unsigned long long mask = __ballot(sense);
      const int state = __popcll(mask & ((1ull << lane) - 1ull));
      if (sense) codon[state] = a | (bq << 2) | (c << 4) | ((int)aa << 8);
    }
    __syncthreads();
    const double f[4] = {row[spec.freq_start], row[spec.freq_start + 1], row[spec.freq_start + 2],
                         row[spec.freq_start + 3]};
    const double kappa = row[spec.rates_start], omega = row[spec.rates_start + 1];
    double praw = 0.0;
    if (lane < S) {
      const int cd = codon[lane];
      praw = (f[cd & 3] * f[(cd >> 2) & 3]) * f[(cd >> 4) & 3];  // F1x4
    }
    pi[lane] = praw;
    __syncthreads();
    double tot = 0.0;
    for (int j = 0; j < S; j++) tot += pi[j];
    __syncthreads();
    mypi = lane < S ? praw / tot : 0.0;
    pi[lane] = mypi;
    __syncthreads();
    if (lane < S) {
      const int ci = codon[lane];
      double rs = 0.0;
      for (int j = 0; j < S; j++) {
        if (j == lane) continue;
        const int cj = codon[j];
        const int x = (ci ^ cj) & 63;
        const int d0 = (x & 3) != 0, d1 = (x & 12) != 0, d2 = (x & 48) != 0;
        double q = 0.0;
        if (d0 + d1 + d2 == 1) {
          const int sh = d0 ? 0 : (d1 ? 2 : 4);
          q = pi[j];
          if ((((ci >> sh) ^ (cj >> sh)) & 3) == 2) q *= kappa;  // transition: A<->G, C<->T
          if ((ci >> 8) != (cj >> 8)) q *= omega;                // amino acid changes
        }
        A[lane * kLd + j] = q;
        rs += q;
      }
      A[lane * kLd + lane] = -rs;
      rowsum[lane] = rs;
    }
  } else {
    // 4-state models through the general path: GTR, and JC69 / HKY written as GTR
    // (GTRModel::UpdateQMatrix, src/substitution_model.cpp:141-166)
    double r[6] = {1. / 6, 1. / 6, 1. / 6, 1. / 6, 1. / 6, 1. / 6}, f[4] = {0.25, 0.25, 0.25, 0.25};
    if (spec.substitution != kJC69)
      for (int i = 0; i < 4; i++) f[i] = row[spec.freq_start + i];
    if (spec.substitution == kGTR)
      for (int i = 0; i < 6; i++) r[i] = row[spec.rates_start + i];
    if (spec.substitution == kHKY) {
      const double kappa = row[spec.rates_start];
      r[0] = r[2] = r[3] = r[5] = 1.0;
      r[1] = r[4] = kappa;
    }
    mypi = lane < 4 ? f[lane] : 0.0;
    pi[lane] = mypi;
    if (lane < 4) {
      double rs = 0.0;
      for (int j = 0; j < 4; j++) {
        if (j == lane) continue;
        const int lo = lane < j ? lane : j, hi = lane < j ? j : lane;
        const int k = lo == 0 ? hi - 1 : (lo == 1 ? hi + 1 : 5);  // AC,AG,AT,CG,CT,GT
        const double q = r[k] * f[j];
        A[lane * kLd + j] = q;
        rs += q;
      }
      A[lane * kLd + lane] = -rs;
      rowsum[lane] = rs;
    }
  }
  __syncthreads();
  double total = 0.0;
  for (int i = 0; i < S; i++) total += rowsum[i] * pi[i];
  if (lane < S)
    for (int j = 0; j < S; j++) A[lane * kLd + j] /= total;
  __syncthreads();
  for (int i = 0; i < 64; i++) {
    out[kGsQ + i * 64 + lane] = A[i * kLd + lane];
    out[kGsQtImage + GsImageIndex(lane, i)] = A[i * kLd + lane];  // MFMA image of Q^T: M[lane][i] = Q[i][lane]
  }
  {
    // the nonzero entries of column `lane` of Q, rows ascending (gs_matrices_kernel: dP = P (r_c Q) term by term --
    // the terms a dense fused multiply-add chain over all rows would add are these and exact zeros)
    int cnt = 0;
    for (int k = 0; k < 64; k++) {
      const double q = A[k * kLd + lane];
      if (q != 0.0) {
        if (cnt < kGsQnzMax) {
          out[kGsQnzIdx + lane * kGsQnzMax + cnt] = (double)k;
          out[kGsQnzVal + lane * kGsQnzMax + cnt] = q;
        }
        cnt++;
      }
    }
    out[kGsQnzCount + lane] = (double)cnt;
    const bool fits = __all(cnt <= kGsQnzMax);
    if (lane == 0) out[kGsQnzFlag] = fits ? 1.0 : 0.0;
  }
  // symmetrise: D^{1/2} Q D^{-1/2}, lower triangle computed, upper mirrored; parked in the V slot
  // of the record until the eigensolver kernel replaces it
  const double mysq = lane < S ? sqrt(mypi) : 1.0;
  sq[lane] = mysq;
  __syncthreads();
  for (int j = 0; j <= lane; j++) A[lane * kLd + j] = mysq * A[lane * kLd + j] / sq[j];
  __syncthreads();
  for (int j = lane + 1; j < 64; j++) A[lane * kLd + j] = A[j * kLd + lane];
  __syncthreads();
  for (int i = 0; i < 64; i++) out[kGsV + i * 64 + lane] = A[i * kLd + lane];
  out[kGsPi + lane] = mypi;
  out[kGsSq + lane] = mysq;
  if (lane == 0) {
    // WeibullSiteModel::UpdateRates (src/site_model.cpp:37-62), as SetupTreeModel
    const int C = spec.category_count;
    double* rate = out + kGsCatRate;
    double* weight = out + kGsCatWeight;
    double* deriv = out + kGsCatRateDeriv;
    if (spec.weibull) {
      const double shape = row[spec.shape_start];
      double mean = 0, dmean = 0;
      double du[kMaxCategories];
      for (int i = 0; i < C; i++) {
        const double log_l = spec.weibull_log_l[i];
        const double rr = DetExp(log_l / shape);  // = pow(-log(1 - quantile), 1 / shape)
        rate[i] = rr;
        mean += rr;
        du[i] = -rr * log_l / (shape * shape);
        dmean += du[i];
      }
      mean /= C;
      dmean /= C;
      for (int i = 0; i < C; i++) {
        deriv[i] = (du[i] * mean - rate[i] * dmean) / (mean * mean);
        rate[i] /= mean;
        weight[i] = 1.0 / C;
      }
    } else {
      rate[0] = 1.0;
      weight[0] = 1.0;
      deriv[0] = 0.0;
    }
  }
}

// Set-up, part 2: symmetric eigensolve, one workgroup (4 waves) per distinct model.  Jacobi
// iteration with the round-robin ordering: 63 rounds of 32 independent rotations per sweep; all
// angles of a round are taken first (each wave its own pairs'), then the column updates (A and U), then
// the row updates, which set the annihilated pairs to exactly zero: two barriers per round.  Wave g owns pairs 8 g .. 8 g + 7 of a round;
// their LDS traffic is issued as eight independent streams.
__device__ __forceinline__ void GsPair(int r, int k, int& p, int& q) {
  // ((r + k) % 63 and (r - k + 63) % 63 for r < 63, k < 32: one conditional subtraction each)
  const int x = r + k, y = r - k + 63;
  const int a = k == 0 ? 63 : (x >= 63 ? x - 63 : x);
  const int bb = k == 0 ? r : (y >= 63 ? y - 63 : y);
  p = a < bb ? a : bb;
  q = a < bb ? bb : a;
}

__device__ __forceinline__ double GsReadLane(double v, int lane) {
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__global__ void __launch_bounds__(256)
gs_eigen_kernel(const int32_t* __restrict__ model_index, double* __restrict__ gs_model) {
#pragma clang fp contract(off)
  __shared__ double A[64 * kLd];
  __shared__ double U[64 * kLd];
  __shared__ double wmax[4];
  // (g is the same in every lane of a wave: said so, the pairs of a round are scalar arithmetic and every LDS address
  // below is one vector add -- round 6; the compiler had kept the pair arithmetic, a modulo by 63 per pair, on the
  // vector ALU: 200 of the 550 instructions of a round)
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, g = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (model_index[t] != t) return;
  double* __restrict__ out = gs_model + (size_t)t * kGsModelStride;
  for (int idx = tid; idx < 4096; idx += 256) {
    const int i = idx >> 6, j = idx & 63;
    A[i * kLd + j] = out[kGsV + idx];
    U[i * kLd + j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int sweep = 0; sweep < 60; sweep++) {
    double mo = 0.0;
    for (int j = g * 16; j < g * 16 + 16; j++)
      if (j != lane) mo = fmax(mo, fabs(A[lane * kLd + j]));
    mo = WaveMax(mo);
    if (lane == 0) wmax[g] = mo;
    __syncthreads();
    mo = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    __syncthreads();
    if (mo < 1e-20) break;
    for (int r = 0; r < 63; r++) {
      // The angles of a wave's eight pairs are taken by the wave itself (lanes 0-7, handed round by shuffles): they read
      // A[p][q], A[p][p], A[q][q] of the wave's OWN pairs, entries that only this wave's column updates touch before the
      // next barrier -- so no barrier stands between the angles and the column updates (round 5: wave 0 took all 32
      // angles and passed them through LDS behind a barrier of the workgroup; the same formulas, the same bits).
      double c_mine = 1.0, s_mine = 0.0;
      int pp = 0, qq = 0;
      if (lane < 8) {
        GsPair(r, g * 8 + lane, pp, qq);
        const double apq = A[pp * kLd + qq];
        if (apq != 0.0) {
          const double theta = (A[qq * kLd + qq] - A[pp * kLd + pp]) / (2.0 * apq);
          const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c_mine = 1.0 / sqrt(tt * tt + 1.0);
          s_mine = tt * c_mine;
        }
      }
      int p[8], q[8];
      double c[8], s[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        GsPair(r, g * 8 + k, p[k], q[k]);
        c[k] = GsReadLane(c_mine, k);  // (a fixed lane: v_readlane into scalar registers, no LDS crossbar -- round 6)
        s[k] = GsReadLane(s_mine, k);
      }
      {  // column updates: lane = row
        double ap[8], aq[8], up[8], uq[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          ap[k] = A[lane * kLd + p[k]];
          aq[k] = A[lane * kLd + q[k]];
          up[k] = U[lane * kLd + p[k]];
          uq[k] = U[lane * kLd + q[k]];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
          A[lane * kLd + p[k]] = c[k] * ap[k] - s[k] * aq[k];
          A[lane * kLd + q[k]] = s[k] * ap[k] + c[k] * aq[k];
          U[lane * kLd + p[k]] = c[k] * up[k] - s[k] * uq[k];
          U[lane * kLd + q[k]] = s[k] * up[k] + c[k] * uq[k];
        }
      }
      __syncthreads();
      {  // row updates: lane = column
        double ap[8], aq[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          ap[k] = A[p[k] * kLd + lane];
          aq[k] = A[q[k] * kLd + lane];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
          A[p[k] * kLd + lane] = c[k] * ap[k] - s[k] * aq[k];
          A[q[k] * kLd + lane] = s[k] * ap[k] + c[k] * aq[k];
        }
        // The annihilated pairs are set to exactly zero by the lanes that took their angles, BEHIND the row update's
        // writes of the same entries: one wave's LDS writes land in program order, and both come from this wave (round
        // 5 selected the zero inside the row update: two compares and four selects per pair in all 64 lanes; round 4
        // had a step of its own behind a barrier).  The same values in LDS before anything reads them.
        __builtin_amdgcn_wave_barrier();  // (no instruction: the order of the two statements, for the compiler and the CPU emulation's fibers)
        if (lane < 8) {
          A[pp * kLd + qq] = 0.0;
          A[qq * kLd + pp] = 0.0;
        }
      }
      __syncthreads();
    }
  }
  // V = D^{-1/2} U, V^-1 = U^T D^{1/2}
  for (int idx = tid; idx < 4096; idx += 256) {
    const int i = idx >> 6, j = idx & 63;
    out[kGsV + idx] = U[i * kLd + j] / out[kGsSq + i];
    out[kGsVinv + idx] = U[j * kLd + i] * out[kGsSq + j];
  }
  if (tid < 64) out[kGsLambda + tid] = A[tid * kLd + tid];
}

// models_stand: the parameter rows are those of the batch before, whose models are still in gs_model (worker.cpp,
// UploadModelIndex): the trees' topologies and effective branch lengths only
void LaunchGsSetup(const BatchDims& d, const ModelSpec& spec, const DeviceBatch& b, const int32_t* model_index,
                   double* gs_model, hipStream_t stream, bool models_stand) {
  hipLaunchKernelGGL(gs_model_kernel, dim3(d.tree_count), dim3(64), 0, stream, d, spec, b, model_index, gs_model,
                     models_stand ? 1 : 0);
  if (!models_stand) hipLaunchKernelGGL(gs_eigen_kernel, dim3(d.tree_count), dim3(256), 0, stream, model_index, gs_model);
}

// --------------------------------------------------------------------------------------------
// Transition matrices.  One workgroup (4 waves) per (branch, category, tree); wave w forms rows
// 16 w .. 16 w + 15 of  P = (V diag(e)) V^-1  on the matrix pipe: 16 chained v_mfma_f64_16x16x4 per
// 16 x 16 block.  Measured on the device (probe_mfma16.hip): such a chain rounds exactly like a
// sequential fma() chain over k = 0..63, which is the fixed operation order of DESIGN.md section 3, so
// P is bit-identical to the CPU restatement's.  P then sits row-major in LDS and every output record
// is written with coalesced 16-byte stores.
//
// Output record per (tree, branch, category): 3 x 4096 doubles.
//   internal branch:  [0] image of P, [2] image of P^T                         (MFMA A operands)
//   leaf branch:      [0] PT[s][.] = P[.][s], [1] dPT[s][.]; row S (gap) is 1 (P) / 0 (dP) on the
//                     real states; dP = P (r_c Q) is formed from the ROUNDED P exactly as the
//                     reference's edge derivative pre^T (r_c Q) post sees it (src/fat_beagle.cpp:101-111)
// Image of a matrix M (out = M x): element M[16 mb + ii][4 ks + kq] sits at
//   ((mb * 8 + ks / 2) * 64 + (16 kq + ii)) * 2 + (ks & 1)
// i.e. the A operand (lane = 16 k + row) of the MFMA for row block mb and k-step ks, two k-steps per
// 16-byte load.  Register r of lane 16 q + j of an MFMA result holds D[4 r + q][j], so register r of
// block m of a vector holds state 16 m + 4 r + q -- which is the B operand (lane = 16 k + column) of
// k-step 4 m + r: a vector flows from one MFMA's result into the next one's operand without leaving
// its registers.  Leaf tables use the same permutation inside each block of 16 states (position
// 16 m + 4 q + r) so that a lane's four values are one 32-byte load.

constexpr int kPld = 66;  // LDS row stride of P (even: 16-byte aligned pairs)

// (Round 3: a workgroup takes kGsMatJobs consecutive (branch, category) jobs of its tree and keeps what they share --
// its rows of V in registers, V^-1 in LDS in the B-operand order, the lists of Q's nonzero entries -- instead of one job
// per workgroup with 80 loads per lane through L1 in front of its 64 matrix instructions: 557 k workgroups per 4096
// config-5 trees spent half the kernel's time outside their products.)
// (Round 4, tried and dropped: ONE image per internal branch read both ways by the walk -- units of a row swizzled by
// unit ^ (4 kq) ^ (row & 1), so that the 16-byte reads of P x keep whole rows and the 8-byte reads of P^T w spread a
// 16-lane group over 16 double-word banks -- a third fewer bytes written here.  Parity-green at the first run; this
// kernel 11.6 -> 10.9 ms per 4096 config-5 trees, the walk 43.3 -> 43.6-44.1 (237 registers instead of 216, twice the
// LDS instructions in a third of its contractions): 73.6 k trees/s against 74.0 k.  The kernel is not bound by its
// writes alone.)  What it IS bound by (timing-only builds of commit 41077d6, GS_MAT_EXP = 1..4; 11.4 ms per 4096
// config-5 trees): without its stores to HBM 10.9 ms, with one k-step of its products 7.1, without the sparse dP terms
// of the leaf branches 8.1 (41 k LDS reads per leaf job, the lists' indices and values among them), leaf branches
// treated as internal ones 8.0 -- arithmetic and LDS instructions, a third each for the products, the sparse dP terms
// and the rest (exponentials, the passes through LDS).  The knobs are not in this file: the build that ships is, line
// for line of device code, the one the round's last full GPU run checked.
constexpr int kGsMatJobs = 8;

__global__ void __launch_bounds__(256, 2)  // (two waves per SIMD, as the 72 KB of LDS allow: 256 registers)
gs_matrices_kernel(BatchDims d, int S, int tree0, const double* __restrict__ branch,
                   const int32_t* __restrict__ model_index, const double* __restrict__ gs_model,
                   double* __restrict__ imgs, int want_gradient, int deriv_mode) {
#pragma clang fp contract(off)
  extern __shared__ double mat_lds[];
  double* const Pl = mat_lds;                                   // [64][kPld]  P, row-major
  v2d* const Bl = reinterpret_cast<v2d*>(Pl + 64 * kPld);       // [16 k-steps][2][64 lanes] pairs of V^-1 entries
  double* const e_all = reinterpret_cast<double*>(Bl + 16 * 2 * 64);  // [kGsMatJobs][64] exp(lambda t r_c) of the workgroup's jobs
#if !GS_DP_COLUMN
  double* const nz_val = e_all + kGsMatJobs * 64;               // [64][kGsQnzMax] nonzero entries of Q's columns
  uint8_t* const nz_idx = reinterpret_cast<uint8_t*>(nz_val + 64 * kGsQnzMax);
  uint8_t* const nz_cnt = nz_idx + 64 * kGsQnzMax;
#endif
  const int C = d.category_count, NB = d.node_count - 1, n = d.taxon_count;
  const int jobs = NB * C;
  const int job0 = blockIdx.x * kGsMatJobs, job1 = min(job0 + kGsMatJobs, jobs);
  const int tree = tree0 + blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, kq = lane >> 4, ii = lane & 15;
  const double* __restrict__ m = gs_model + (size_t)model_index[tree] * kGsModelStride;
  // what the jobs share
  double v_row[16];  // V[16 w + ii][4 ks + kq]: this lane's A operands before the scaling by e
#pragma unroll
  for (int ks = 0; ks < 16; ks++) v_row[ks] = m[kGsV + (16 * w + ii) * 64 + 4 * ks + kq];
  for (int i = tid; i < 16 * 2 * 64; i += 256) {
    const int l = i & 63, h = (i >> 6) & 1, ks = i >> 7;
    const double* src = m + kGsVinv + (4 * ks + (l >> 4)) * 64 + 32 * h + (l & 15);
    Bl[i] = v2d{src[0], src[16]};
  }
  const bool sparse_dp = GS_SPARSE_DP && want_gradient && m[kGsQnzFlag] != 0.0;
#if GS_DP_COLUMN
  // Round 6: four threads per column of dP^T keep that column's list of Q -- row offsets into P and values -- in
  // REGISTERS for all the jobs of the workgroup (round 3 staged the lists in LDS and read index and value again for
  // every term of every pair of outputs: four LDS reads per term and pair, now two).  Lists shorter than kGsQnzMax are
  // padded with (offset 0, value 0): a term P x (+-0) leaves the sum as it is, bit for bit, and the loop below has no
  // trip count -- every register index is static.
  const int dp_col = tid >> 2, dp_q = tid & 3;
  int nz_off[kGsQnzMax];
  double nz_v[kGsQnzMax];
  if (sparse_dp) {
    const int cnt = dp_col < S ? (int)m[kGsQnzCount + dp_col] : 0;
#pragma unroll
    for (int t = 0; t < kGsQnzMax; t++) {
      nz_off[t] = t < cnt ? (int)m[kGsQnzIdx + dp_col * kGsQnzMax + t] : 0;
      nz_v[t] = t < cnt ? m[kGsQnzVal + dp_col * kGsQnzMax + t] : 0.0;
    }
  }
#else
  if (sparse_dp) {
    for (int i = tid; i < 64 * kGsQnzMax; i += 256) {
      nz_val[i] = m[kGsQnzVal + i];
      nz_idx[i] = (uint8_t)m[kGsQnzIdx + i];
    }
    if (tid < 64) nz_cnt[tid] = (uint8_t)m[kGsQnzCount + tid];
  }
#endif
  // result registers -> row-major LDS: register r of lane 16 q + j holds row 4 r + q, column j
  auto to_lds = [&](const v4d acc[4]) {
#pragma unroll
    for (int nb = 0; nb < 4; nb++)
#pragma unroll
      for (int r = 0; r < 4; r++) Pl[(16 * w + 4 * r + kq) * kPld + 16 * nb + ii] = acc[nb][r];
  };

  // the exponentials of ALL the workgroup's jobs at once, two per thread (round 6; one job's 64 at a time before, by
  // one wave with the other three waiting at a barrier of their own per job: seven barriers fewer per workgroup)
  for (int i = tid; i < (job1 - job0) * 64; i += 256) {
    const int job = job0 + (i >> 6), br = job / C, c = job % C;
    const double time = branch[(size_t)tree * d.node_count + br] * m[kGsCatRate + c];
    e_all[i] = DetExp(m[kGsLambda + (i & 63)] * time);
  }
  for (int job = job0; job < job1; job++) {
    const int br = job / C, c = job % C;
    const double rate = m[kGsCatRate + c];
    const double* const e = e_all + (job - job0) * 64;
    __syncthreads();  // (the previous job's readers of Pl are done; the first job: Bl, the exponentials and the lists are in place)
    double* __restrict__ rec = imgs + (((size_t)blockIdx.y * NB + br) * C + c) * (3 * 4096);
    // rows 16 w .. 16 w + 15 of (V diag e) V^-1
    v4d acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; nb++) acc[nb] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int ks = 0; ks < 16; ks++) {
      const double a = v_row[ks] * e[4 * ks + kq];
      const v2d b01 = Bl[(ks * 2) * 64 + lane], b23 = Bl[(ks * 2 + 1) * 64 + lane];
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b01.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b01.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b23.x, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b23.y, acc[3], 0, 0, 0);
    }
    to_lds(acc);
    __syncthreads();

    if (br >= n) {
      // internal branch: image of P, and of P^T for the pre-order pass; 16-byte coalesced stores
      v2d* __restrict__ out0 = reinterpret_cast<v2d*>(rec);
      v2d* __restrict__ out2 = reinterpret_cast<v2d*>(rec + 8192);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int pos = i * 256 + tid;  // = (mb * 8 + ks2) * 64 + 16 kq' + ii'
        const int l = pos & 63, blk = pos >> 6, mb = blk >> 3, ks2 = blk & 7;
        const int row = 16 * mb + (l & 15), col = 8 * ks2 + (l >> 4);
        out0[pos] = v2d{Pl[row * kPld + col], Pl[row * kPld + col + 4]};
        if (want_gradient) out2[pos] = v2d{Pl[col * kPld + row], Pl[(col + 4) * kPld + row]};
      }
      continue;
    }
    // leaf branch: transposed look-up tables; position 16 m + 4 q + r of row s holds state 16 m + 4 r + q
    auto table = [&](double* __restrict__ dst, bool is_p) {
      v2d* __restrict__ out = reinterpret_cast<v2d*>(dst);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int pos = i * 256 + tid;  // pair index: row s = pos / 32, positions 2 (pos % 32), +1
        const int s = pos >> 5, at = (pos & 31) * 2;
        const int mm = at >> 4, q = (at >> 2) & 3, r = at & 3;  // r in {0, 2}
        const int st0 = 16 * mm + 4 * r + q, st1 = st0 + 4;
        v2d v{Pl[st0 * kPld + s], Pl[st1 * kPld + s]};
        if (s == S) v = is_p ? v2d{st0 < S ? 1.0 : 0.0, st1 < S ? 1.0 : 0.0} : v2d{0.0, 0.0};
        out[pos] = v;
      }
    };
    table(rec, true);
    if (!want_gradient) continue;
    // dP = P (r_c Q); deriv_mode 1: the site-model pass, r_c -> d r_c / d shape
    const double drate = deriv_mode ? m[kGsCatRateDeriv + c] : rate;
    if (sparse_dp) {
      // Q of a codon model has at most ten entries per column: the product on the vector ALU, straight into the table
      // positions, one fused multiply-add per nonzero entry in ascending row order -- bit for bit what the dense chain on
      // the matrix pipe gives (its other terms are exact zeros), at a tenth of the arithmetic and none of it on the pipe
      // this kernel and the traversal wait for
      v2d* __restrict__ out = reinterpret_cast<v2d*>(rec + 4096);
#if GS_DP_COLUMN
      double qv[kGsQnzMax];
#pragma unroll
      for (int t = 0; t < kGsQnzMax; t++) qv[t] = nz_v[t] * drate;
#pragma unroll 2
      for (int j = 0; j < 8; j++) {
        // positions at, at + 1 of row dp_col: states 16 mm + 4 r + q and the one four further (r + 1); a quartet of
        // threads writes 64 contiguous bytes per store
        const int at = 2 * (dp_q + 4 * j);
        const int mm = at >> 4, q = (at >> 2) & 3, r = at & 3;
        const double* __restrict__ p0 = Pl + (16 * mm + 4 * r + q) * kPld;
        const double* __restrict__ p1 = p0 + 4 * kPld;
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int t = 0; t < kGsQnzMax; t++) {
          d0 = __builtin_fma(p0[nz_off[t]], qv[t], d0);
          d1 = __builtin_fma(p1[nz_off[t]], qv[t], d1);
        }
        out[dp_col * 32 + dp_q + 4 * j] = v2d{d0, d1};
      }
#else
#pragma unroll 2
      for (int i = 0; i < 8; i++) {
        const int pos = i * 256 + tid;
        const int s = pos >> 5, at = (pos & 31) * 2;
        const int mm = at >> 4, q = (at >> 2) & 3, r = at & 3;
        const int st0 = 16 * mm + 4 * r + q, st1 = st0 + 4;
        double d0 = 0.0, d1 = 0.0;
        if (s != S) {
          const int cnt = nz_cnt[s];
          const uint8_t* idx = nz_idx + s * kGsQnzMax;
          const double* val = nz_val + s * kGsQnzMax;
          for (int t = 0; t < cnt; t++) {
            const int k = idx[t];
            const double qv = val[t] * drate;
            d0 = __builtin_fma(Pl[st0 * kPld + k], qv, d0);
            d1 = __builtin_fma(Pl[st1 * kPld + k], qv, d1);
          }
        }
        out[pos] = v2d{d0, d1};
      }
#endif
      continue;
    }
    // (a model whose Q is not that sparse: the second product on the matrix pipe, B = r_c Q from the model record)
    v4d accd[4];
#pragma unroll
    for (int nb = 0; nb < 4; nb++) accd[nb] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int ks = 0; ks < 16; ks++) {
      const double a = Pl[(16 * w + ii) * kPld + 4 * ks + kq];
      const double* brow = m + kGsQ + (4 * ks + kq) * 64 + ii;
#pragma unroll
      for (int nb = 0; nb < 4; nb++)
        accd[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, brow[16 * nb] * drate, accd[nb], 0, 0, 0);
    }
    __syncthreads();
    to_lds(accd);
    __syncthreads();
    table(rec + 4096, false);
  }
}

void LaunchGsMatrices(const BatchDims& d, int S, int tree0, int chunk, const double* branch,
                      const int32_t* model_index, const double* gs_model, double* imgs, int want_gradient,
                      int deriv_mode, hipStream_t stream) {
  const int jobs = (d.node_count - 1) * d.category_count;
  const dim3 grid((jobs + kGsMatJobs - 1) / kGsMatJobs, chunk);
#if GS_DP_COLUMN
  const size_t lds = (64 * kPld + 16 * 2 * 64 * 2 + kGsMatJobs * 64) * sizeof(double);
#else
  const size_t lds = (64 * kPld + 16 * 2 * 64 * 2 + kGsMatJobs * 64 + 64 * kGsQnzMax) * sizeof(double) + 64 * kGsQnzMax + 64;
#endif
  // (69.5 KB of LDS: two workgroups per CU; the attribute is per device, and a process may drive several)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gs_matrices_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipLaunchKernelGGL(gs_matrices_kernel, grid, dim3(256), lds, stream, d, S, tree0, branch, model_index, gs_model,
                     imgs, want_gradient, deriv_mode);
}

// --------------------------------------------------------------------------------------------
// Traversal.  A wave owns 16 consecutive site patterns of one tree; the four waves of a workgroup
// own four consecutive tiles of the SAME tree and therefore need the same matrix images in the same
// order.  A partial-likelihood vector of a wave's patterns (64 padded states x 16 patterns) lives in
// 32 VGPRs per lane in the MFMA's D layout: lane = 16 kq + pn holds, for block m and register r, state
// 16 m + 4 r + kq of pattern pn.  Stored vectors (post-order partials, overwritten in place by
// pre-order partials exactly as in walk_hbm_kernel) live in an HBM arena in that same layout,
// [m][lane][r]: 2 KB coalesced rows.
//
// Images go through LDS: the order in which a tree's images are needed is written once per tree by
// gs_schedule_kernel (a list of record numbers); while the workgroup contracts with image j out of
// one 32 KB LDS buffer, image j+1 is already on its way from L2 straight into the other buffer
// (global_load_lds_dwordx4 issued before the MFMAs) -- one workgroup barrier per image, L2 read once
// per workgroup instead of once per wave.

// (from here on the walk's own arithmetic: per pattern, free to use fused multiply-add -- model.hpp's file-scope
// contract(off) is meant for the set-up kernels above, which repeat it in their bodies)
#pragma clang fp contract(fast)

struct GsPlv {
  v4d b[4];
};

// Image order of one tree.  Post-order: per internal node and category, P of each internal child.
// Pre-order: per internal node (parents first) and category: the model's Q^T (entry -1; not for the root), then P^T
// of each internal child.  Other entries are record numbers (br * C + c) * 3 + which.
__global__ void __launch_bounds__(64)
gs_schedule_kernel(BatchDims d, const int32_t* __restrict__ children, int32_t* __restrict__ jobs, int stride) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= d.tree_count) return;
  const int n = d.taxon_count, N = d.node_count, C = d.category_count;
  const int32_t* ch = children + (size_t)t * (n - 1) * 2;
  int32_t* out = jobs + (size_t)t * stride;
  int at = 0;
  // child order of the walk: the chained child (the second child if internal, else the first) is
  // contracted first on the way up and last on the way down
  for (int node = n; node < N; node++) {
    const int c0 = ch[(node - n) * 2], c1 = ch[(node - n) * 2 + 1];
    const int first = c1 >= n ? c1 : c0, second = c1 >= n ? c0 : c1;
    for (int c = 0; c < C; c++) {
      if (first >= n) out[at++] = (first * C + c) * 3;
      if (second >= n) out[at++] = (second * C + c) * 3;
    }
  }
  for (int node = N - 1; node >= n; node--) {
    const int c0 = ch[(node - n) * 2], c1 = ch[(node - n) * 2 + 1];
    const int last = c1 >= n ? c1 : c0, first = c1 >= n ? c0 : c1;
    for (int c = 0; c < C; c++) {
#if GS_OWN_EDGE
      if (node != N - 1) out[at++] = -1;
      if (first >= n) out[at++] = (first * C + c) * 3 + 2;
      if (last >= n) out[at++] = (last * C + c) * 3 + 2;
#else
      if (first >= n) {
        out[at++] = (first * C + c) * 3 + 2;
        out[at++] = -1;
      }
      if (last >= n) {
        out[at++] = (last * C + c) * 3 + 2;
        out[at++] = -1;
      }
#endif
    }
  }
  const int tail = at ? out[at - 1] : 0;
  for (; at < stride; at++) out[at] = tail;
}

int GsScheduleStride(const BatchDims& d) { return (d.taxon_count - 1) * d.category_count * 8 + 1; }

void LaunchGsSchedule(const BatchDims& d, const DeviceBatch& b, hipStream_t stream) {
  hipLaunchKernelGGL(gs_schedule_kernel, dim3((d.tree_count + 63) / 64), dim3(64), 0, stream, d, b.children, b.sched,
                     GsScheduleStride(d));
}

// (non-temporal arena accesses, which pay in walk_hbm_cat_kernel, cost here: 59.5 against 56.8 ms per 4096 config-5
// trees -- the vectors ARE read again soon, messages by the same wave's pre-order pass out of L2 / MALL)
#ifndef GS_ARENA_NT
#define GS_ARENA_NT 0
#endif
// Stored vectors go through buffer instructions: the slot's address is wave-uniform (scalar registers: a descriptor
// built by scalar instructions), a lane adds ONE 32-bit offset that never changes (lane_bytes = 32 lane) and the
// instruction its constant -- no 64-bit per-lane address arithmetic on the vector ALU, which a wave gets one issue
// slot of in ~58 cycles while its SIMD's other wave issues FP64 matrix instructions (77 v_lshl_add_u64 in the kernel
// before).
typedef unsigned GsUInt4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t GsRsrc(const double* slot) {
  // (the address is wave-uniform by construction; said explicitly, or a descriptor the compiler happened to form on
  // the vector ALU costs a four-register readfirstlane loop around every access)
  const uintptr_t a = reinterpret_cast<uintptr_t>(slot);
  const uintptr_t u = (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a) |
                      ((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<double*>(u), 0, 8192, 0x00020000);
}
__device__ __forceinline__ void GsLoad(const double* __restrict__ slot, unsigned lane_bytes, GsPlv& x) {
#if GS_EXP_NOLOAD
  for (int m = 0; m < 4; m++) x.b[m] = v4d{1.0, 0.5, 0.25, 0.125};
  return;
#endif
  const __amdgpu_buffer_rsrc_t r = GsRsrc(slot);
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const GsUInt4 lo = __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes + (m & 1) * 2048, (m >> 1) * 4096, GS_ARENA_NT ? 2 : 0);
    const GsUInt4 hi = __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes + (m & 1) * 2048 + 16, (m >> 1) * 4096, GS_ARENA_NT ? 2 : 0);
    const v2d a = __builtin_bit_cast(v2d, lo), b = __builtin_bit_cast(v2d, hi);
    x.b[m] = v4d{a.x, a.y, b.x, b.y};
  }
}
__device__ __forceinline__ void GsStore(double* __restrict__ slot, unsigned lane_bytes, const GsPlv& x) {
#if GS_EXP_NOSTORE
  return;
#endif
  const __amdgpu_buffer_rsrc_t r = GsRsrc(slot);
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const v2d a{x.b[m][0], x.b[m][1]}, b{x.b[m][2], x.b[m][3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(GsUInt4, a), r, lane_bytes + (m & 1) * 2048, (m >> 1) * 4096, GS_ARENA_NT ? 2 : 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(GsUInt4, b), r, lane_bytes + (m & 1) * 2048 + 16, (m >> 1) * 4096, GS_ARENA_NT ? 2 : 0);
  }
}
// tip child: row `state` of the transposed table
__device__ __forceinline__ void GsTip(const double* __restrict__ table, int state, int kq, GsPlv& x) {
#pragma unroll
  for (int m = 0; m < 4; m++) x.b[m] = *reinterpret_cast<const v4d*>(table + state * 64 + 16 * m + 4 * kq);
}
__device__ __forceinline__ double GsDot(const GsPlv& a, const GsPlv& b) {
  double s = 0.0;
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int r = 0; r < 4; r++) s += a.b[m][r] * b.b[m][r];
  return s;
}
// sum over the four lanes that hold one pattern's states (kq = 0..3: lanes 16 apart), in each of them
__device__ __forceinline__ double GsPatternSum(double v) { return SwapSum32(SwapSum16(v)); }

// The workgroup's image pipeline (all 256 threads call every member together).
struct GsImagePipe {
  double* lds;                        // two 4096-double buffers
  const double* __restrict__ recs;    // the tree's records
  const double* __restrict__ qt;      // the model's Q^T image (entry -1 of the image order)
  const int32_t* __restrict__ jobs;   // the tree's image order
  int j;                              // image now in lds[(j & 1) * 4096]
  int tid, lane;

  __device__ __forceinline__ const double* Image(int entry) const {
    return entry < 0 ? qt : recs + (size_t)entry * 4096;
  }

  // global -> LDS without passing through registers (global_load_lds_dwordx4): every thread moves
  // 8 x 16 bytes of the 32 KB image; the LDS address is wave-uniform base (M0) + 16 * lane.
  // Written as assembly on purpose: the compiler cannot tell the buffer such a load fills from the buffer the
  // contraction reads, and put s_waitcnt vmcnt(0) in front of the first ds_read behind the builtin form -- the wave sat
  // out the whole transfer of image j + 1 before the first product with image j.  MatVec waits for these loads itself
  // (vmcnt(0) in front of its barrier).
  __device__ __forceinline__ void Fetch(int entry, int buffer) {
#if GS_ASM_FETCH
    // wave w moves bytes [w * 32 KB / waves, ...) of the image, 1 KB per instruction; the instruction's offset field
    // (0 .. 3072) counts for the global AND the LDS address, so four instructions share a base: no vector instruction
    // in a fetch (seven additions before)
    static_assert(GS_WG_WAVES == 4 || GS_WG_WAVES == 8, "eight or four loads of 1 KB per wave");
    const double* src = Image(entry);
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)src);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)src >> 32));
    const uint32_t wave_bytes = (tid >> 6) * (32768 / GS_WG_WAVES);
    uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds + buffer * 4096) + wave_bytes);
    const uint32_t v = wave_bytes + (tid & 63) * 16, v2 = v + 4096;  // (loop invariants: two registers for the whole walk)
#define GS_FETCH_4(V) "global_load_lds_dwordx4 %[" V "], %[s]\n global_load_lds_dwordx4 %[" V "], %[s] offset:1024\n" \
                      "global_load_lds_dwordx4 %[" V "], %[s] offset:2048\n global_load_lds_dwordx4 %[" V "], %[s] offset:3072\n"
#if GS_WG_WAVES == 4
#define GS_FETCH_REST "s_add_u32 m0, m0, 0x1000\n s_nop 0\n" GS_FETCH_4("v2")
#else
#define GS_FETCH_REST
#endif
    asm volatile("s_mov_b32 m0, %[m]\n s_nop 0\n" GS_FETCH_4("v") GS_FETCH_REST
                 :
                 : [v] "v"(v), [v2] "v"(v2), [m] "s"(m), [s] "s"(((uint64_t)hi << 32) | lo)
                 : "memory", "scc");  // (M0: the compiler has no other use for it in this kernel)
#undef GS_FETCH_4
#undef GS_FETCH_REST
#else
    const double* src = Image(entry) + tid * 2;
    double* dst = lds + buffer * 4096 + (tid & ~63) * 2;
#pragma unroll
    for (int i = 0; i < 32 / GS_WG_WAVES; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * (GS_WG_WAVES * 128)),
                                       (__attribute__((address_space(3))) void*)(dst + i * (GS_WG_WAVES * 128)), 16, 0, 0);
#endif
  }

  __device__ __forceinline__ void FetchWait() {
#if GS_ASM_FETCH
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  }

  __device__ __forceinline__ void Begin() {
    j = 0;
    Fetch(jobs[0], 0);
    FetchWait();
    __syncthreads();
  }

  // out = (image j) x; leaves image j+1 in the other buffer
  __device__ __forceinline__ void MatVec(const GsPlv& x, GsPlv& out) {
#if GS_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);  // keep the caller's loads from being hoisted across the contraction
#endif
#if GS_ASM_FETCH
    // x complete before the fetch is issued (the compiler's wait for x would otherwise count the fetch's loads as well);
    // an empty statement that only READS x: operands that were also written cost sixteen register moves per call
    asm volatile("" : : "v"(x.b[0]), "v"(x.b[1]), "v"(x.b[2]), "v"(x.b[3]));
#endif
    Fetch(__builtin_amdgcn_readfirstlane(jobs[j + 1]), (j + 1) & 1);
    const v2d* p = reinterpret_cast<const v2d*>(lds + (j & 1) * 4096) + lane;
    v4d acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; mb++) acc[mb] = v4d{0.0, 0.0, 0.0, 0.0};
    v2d a[4], an[4];
#pragma unroll
    for (int mb = 0; mb < 4; mb++) a[mb] = p[(mb * 8) * 64];
#pragma unroll
    for (int k2 = 0; k2 < (GS_EXP_NOMFMA ? 1 : 8); k2++) {
      if (k2 < 7) {
#pragma unroll
        for (int mb = 0; mb < 4; mb++) an[mb] = p[(mb * 8 + k2 + 1) * 64];
      }
      const double b0 = x.b[k2 >> 1][(k2 & 1) * 2], b1 = x.b[k2 >> 1][(k2 & 1) * 2 + 1];
#pragma unroll
      for (int mb = 0; mb < 4; mb++) acc[mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mb].x, b0, acc[mb], 0, 0, 0);
#pragma unroll
      for (int mb = 0; mb < 4; mb++) acc[mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mb].y, b1, acc[mb], 0, 0, 0);
#pragma unroll
      for (int mb = 0; mb < 4; mb++) a[mb] = an[mb];
    }
#pragma unroll
    for (int mb = 0; mb < 4; mb++) out.b[mb] = acc[mb];
    j++;
    FetchWait();
    __syncthreads();
#if GS_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
};

template <bool GRAD, bool RESCALE>
__global__ void __launch_bounds__(GS_WG_WAVES * 64, GS_WAVES)
gs_walk_kernel(BatchDims d, int S, int tree0, int chunk, int tiles, int sched_stride, int deriv_mode, const int32_t* __restrict__ children,
               const int32_t* __restrict__ sched, const double* __restrict__ imgs,
               const int32_t* __restrict__ model_index, const double* __restrict__ gs_model,
               const uint8_t* __restrict__ tip_states, const double* __restrict__ weights,
               double* __restrict__ arena, double* __restrict__ scale_arena, double* __restrict__ part_ll,
               double* __restrict__ part_grad) {
  extern __shared__ double lds[];  // two image buffers
  const int n = d.taxon_count, N = d.node_count, NI = n - 1, C = d.category_count, Ppad = d.pattern_stride;
  // (the wave's number as a scalar: everything derived from it -- its tile, its vectors' addresses -- is then scalar
  // arithmetic, and the vector loads and stores take a scalar base + one constant lane offset)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, kq = lane >> 4, pn = lane & 15;
  // a wave past the last tile repeats the last tile's work (it must take part in the image
  // pipeline) and stores nothing
  // Workgroups are dealt to the 8 XCDs round-robin by linear id: id % 8 picks the XCD.  All tile groups
  // of a tree get the same id % 8, so that a tree's images are fetched into one L2 only.
  const int groups = (tiles + GS_WG_WAVES - 1) / GS_WG_WAVES;
  const int tree_local = (int)(blockIdx.x / 8 / groups) * 8 + (int)(blockIdx.x % 8);
  const int group = (int)(blockIdx.x / 8) % groups;
  if (tree_local >= chunk) return;
  const unsigned lane_bytes = lane * 32;  // a lane's byte offset into a stored vector (GsLoad / GsStore)
  const bool active = group * GS_WG_WAVES + wave < tiles;
  const int tile = active ? group * GS_WG_WAVES + wave : tiles - 1;
  const int tree = tree0 + tree_local;
  const int p = tile * 16 + pn;
  const int32_t* __restrict__ ch = children + (size_t)tree * NI * 2;
  const double* __restrict__ model = gs_model + (size_t)model_index[tree] * kGsModelStride;
  const double* __restrict__ recs = imgs + (size_t)tree_local * (N - 1) * C * (3 * 4096);
  double* __restrict__ slots = arena + (size_t)tree_local * (GRAD ? 2 : 1) * NI * C * tiles * 1024;
  const uint8_t* __restrict__ tips = tip_states + p;
  const double weight = weights[p];
  auto slot = [&](int node, int c) { return slots + (((size_t)(node - n) * C + c) * tiles + tile) * 1024; };
  // message P x of an internal child, kept for the pre-order pass (second half of the tree's arena)
  auto mslot = [&](int node, int c) { return slots + (((size_t)(NI + node - n) * C + c) * tiles + tile) * 1024; };
  auto rec = [&](int br, int c, int which) { return recs + (((size_t)br * C + c) * 3 + which) * 4096; };
  // RESCALE && GRAD: reciprocal post-order scale factor of (node, pattern)
  auto inv_at = [&](int node) { return scale_arena + (((size_t)tree_local * NI + (node - n)) * tiles + tile) * 16 + pn; };
  GsImagePipe pipe{lds, recs, model + kGsQtImage, sched + (size_t)tree * sched_stride, 0, (int)threadIdx.x, lane};
  pipe.Begin();

  auto load_pi = [&](GsPlv& v) {  // stationary frequencies in the vector layout (root only)
    const double* mp = model + kGsPi + kq;
    asm volatile("" : "+v"(mp));  // opaque address: keeps these 32 registers from being hoisted out of the node loops
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int r = 0; r < 4; r++) v.b[m][r] = mp[16 * m + 4 * r];
  };

  // Ids are in post-order, so an internal node v with an internal child has child v - 1 (its second
  // child if that is internal, else its first): the CHAINED child.  With one rate category the
  // chained child's vector is still in registers when v is processed in the post-order pass (it is the
  // previous iteration's result), and in the pre-order pass -- which visits v - 1 right after v -- the
  // chained child's pre-order partial, produced last in v's step, is the next iteration's input.  That
  // removes one 8 KB load per internal node and pass.  The image order written by gs_schedule_kernel
  // follows the same child order: chained child first on the way up, last on the way down.
  const bool chain = C == 1;

  // ---- post-order: dest = (P_f x_f) . (P_s x_s) per category ------------------------------
  double site = 0.0, log_scale = 0.0;
  GsPlv a;  // the node's partial; survives into the next iteration
  for (int node = n; node < N; ++node) {
    const int c0 = __builtin_amdgcn_readfirstlane(ch[(node - n) * 2]);
    const int c1 = __builtin_amdgcn_readfirstlane(ch[(node - n) * 2 + 1]);
    const int cf = c1 >= n ? c1 : c0, cs = c1 >= n ? c0 : c1;  // chained child (if any) first
    const int sf = cf < n ? tips[(size_t)cf * Ppad] : 0;
    const int ss = cs < n ? tips[(size_t)cs * Ppad] : 0;
    double cat_max = 0.0;  // RESCALE with several categories: maximum over all of them
    for (int c = 0; c < C; c++) {
      GsPlv bb, x;
#if GS_SIBLING_EARLY
      // (the sibling's partial is requested before the first child's contraction, not behind it)
      GsPlv x2;
      if (cs >= n && cf >= n) GsLoad(slot(cs, c), lane_bytes, x2);
#endif
      if (cf < n) {
        GsTip(rec(cf, c, 0), sf, kq, a);
      } else {
        if (chain && cf == node - 1) {  // (always so when ids are in post-order)
          pipe.MatVec(a, a);  // (the contraction has read its operand before it hands out its result: no copy of a)
        } else {
          GsLoad(slot(cf, c), lane_bytes, x);
          pipe.MatVec(x, a);
        }
        if (GRAD && active) GsStore(mslot(cf, c), lane_bytes, a);
      }
      if (cs < n) {
        GsTip(rec(cs, c, 0), ss, kq, bb);
      } else {
#if GS_SIBLING_EARLY
        if (cf >= n) x = x2;
        else GsLoad(slot(cs, c), lane_bytes, x);
#else
        GsLoad(slot(cs, c), lane_bytes, x);
#endif
        pipe.MatVec(x, bb);
        if (GRAD && active) GsStore(mslot(cs, c), lane_bytes, bb);
      }
#pragma unroll
      for (int m = 0; m < 4; m++) a.b[m] *= bb.b[m];
      if (RESCALE && C == 1) {
        // BEAGLE manual scaling (src/fat_beagle.cpp:353-364): per pattern, divide by the maximum over
        // states (and categories) and accumulate its logarithm.  A pattern's 64 states sit in four lanes.
        double mx = 0.0;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int r = 0; r < 4; r++) mx = fmax(mx, a.b[m][r]);
        mx = fmax(mx, __shfl_xor(mx, 16));
        mx = fmax(mx, __shfl_xor(mx, 32));
        if (mx == 0.0) mx = 1.0;
        const double inv = 1.0 / mx;
#pragma unroll
        for (int m = 0; m < 4; m++) a.b[m] *= inv;
        log_scale += log(mx);
        if (GRAD && active && kq == 0) *inv_at(node) = inv;
      }
      if (RESCALE && C > 1) {
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int r = 0; r < 4; r++) cat_max = fmax(cat_max, a.b[m][r]);
      }
      if (node == N - 1) {
        load_pi(bb);
        site += model[kGsCatWeight + c] * GsDot(bb, a);
      } else if (active && ((GRAD && !GS_OWN_EDGE) || !chain || node + 1 >= N || (ch[(node + 1 - n) * 2] != node && ch[(node + 1 - n) * 2 + 1] != node))) {
        // (no copy in memory of a vector that is consumed from registers by the next node: the pre-order pass does
        // not read post-order partials either, it rebuilds them from the children's messages)
        GsStore(slot(node, c), lane_bytes, a);
      }
    }
    if (RESCALE && C > 1) {
      // several categories share one factor per pattern: the partials were stored unscaled, now the
      // maximum over all categories is known.  (The root is left unscaled: it is never stored.)
      double inv = 1.0;
      if (node != N - 1) {
        double mx = fmax(cat_max, __shfl_xor(cat_max, 16));
        mx = fmax(mx, __shfl_xor(mx, 32));
        if (mx == 0.0) mx = 1.0;
        inv = 1.0 / mx;
        log_scale += log(mx);
        for (int c = 0; c < C; c++) {
          GsPlv v;
          GsLoad(slot(node, c), lane_bytes, v);
#pragma unroll
          for (int m = 0; m < 4; m++) v.b[m] *= inv;
          if (active) GsStore(slot(node, c), lane_bytes, v);
        }
      }
      if (GRAD && active && kq == 0) *inv_at(node) = inv;
    }
  }
  site = GsPatternSum(site);
  const double ll = kq == 0 ? weight * (log(site) + log_scale) : 0.0;
  const double wll = WaveSum64(ll);
  if (lane == 0 && active) part_ll[(size_t)tree * tiles + tile] = wll;

  // ---- pre-order + edge derivatives: one step per internal node, parents first -------------
#if GS_OWN_EDGE
  // A step has the node's pre-order partial u (from its parent's step: in registers when the node is its parent's
  // chained child, else from the arena) and the messages a_f = P_f x_f, a_l = P_l x_l of its two children (kept by
  // the post-order pass, or a tip's table row).  From these alone:
  //   * the node's OWN edge: its post-order partial is x = a_f . a_l, so the reference's  pre^T (r_c Q) post  of the
  //     edge above it (src/fat_beagle.cpp:101-160) is (Q^T u) . a_f . a_l -- one contraction with Q^T; no post-order
  //     partial is read back, and the post-order pass keeps in memory only the partials a sibling's step loads;
  //   * the pattern's likelihood  den = u . a_f . a_l  (any positive per-pattern factor cancels in num / den);
  //   * the children's pre-order partials  q_f = P_f^T (u . a_l),  q_l = P_l^T (u . a_f): stored, or handed to the
  //     next step in registers; a tip child's edge derivative from its dP table row.
  // (Round 2 formed a child's edge derivative in the parent's step, (Q^T q_child) . x_child with x_child re-read:
  // per internal node one more 8 KB load and one more 8 KB store per pattern tile.  The form  w^T (r_c Q) (P x)  with
  // the child's MESSAGE is not an alternative: Q and the ROUNDED P of a codon model do not commute to better than
  // about 1e-6 relative, a thousand tolerances on short branches with large derivatives.)
  if (GRAD) {
    double* __restrict__ grow = part_grad + ((size_t)tree * tiles + tile) * N;
    if (lane == 0 && active) grow[N - 1] = 0.0;
    const double site_scale = weight / site;
    GsPlv u;              // the node's pre-order partial; the chained child's is formed in place and survives into the next iteration
    bool have_u = false;  // u is node's own pre-order partial already (wave-uniform)
    for (int node = N - 1; node >= n; --node) {
      const int c0 = __builtin_amdgcn_readfirstlane(ch[(node - n) * 2]);
      const int c1 = __builtin_amdgcn_readfirstlane(ch[(node - n) * 2 + 1]);
      const int cl = c1 >= n ? c1 : c0, cf = c1 >= n ? c0 : c1;  // chained child (if any) last
      const int sf = cf < n ? tips[(size_t)cf * Ppad] : 0;
      const int sl = cl < n ? tips[(size_t)cl * Ppad] : 0;
      double den = 0.0, numf = 0.0, numl = 0.0, numo = 0.0;
      // Rescaled pre-order partials: a child's partial is divided by this node's POST-order factor
      // (any positive per-pattern factor cancels in num / den; this one keeps the products O(1), see
      // walk_hbm_kernel)
      const double step_inv = RESCALE ? *inv_at(node) : 1.0;
      for (int c = 0; c < C; c++) {
        GsPlv af, al;
        if (node == N - 1) load_pi(u);
        else if (!have_u) GsLoad(slot(node, c), lane_bytes, u);
        if (cf < n) GsTip(rec(cf, c, 0), sf, kq, af);
        else GsLoad(mslot(cf, c), lane_bytes, af);
        if (cl < n) GsTip(rec(cl, c, 0), sl, kq, al);
        else GsLoad(mslot(cl, c), lane_bytes, al);
        const double wc = model[kGsCatWeight + c];
        const double rc = model[(deriv_mode ? kGsCatRateDeriv : kGsCatRate) + c];  // site-model pass: d r_c / d shape
        if (node != N - 1) {
          GsPlv t;
          pipe.MatVec(u, t);  // Q^T u
          double so = 0.0, sd = 0.0;
#pragma unroll
          for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const double x = af.b[m][r] * al.b[m][r];
              so += t.b[m][r] * x;
              if (RESCALE) sd += u.b[m][r] * x;
            }
          numo += wc * rc * so;
          den += wc * sd;
        } else if (RESCALE) {
          double sd = 0.0;
#pragma unroll
          for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) sd += u.b[m][r] * (af.b[m][r] * al.b[m][r]);
          den += wc * sd;
        }
        // what each child sees from above: w_f = u . a_l, w_l = u . a_f (in place)
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const v4d f = af.b[m];
          af.b[m] = u.b[m] * al.b[m];  // w_f
          al.b[m] = u.b[m] * f;        // w_l
        }
        if (cf < n) {
          GsPlv x;
          GsTip(rec(cf, c, 1), sf, kq, x);
          numf += wc * GsDot(af, x);
        } else {
          GsPlv y;
          pipe.MatVec(af, y);
          if (RESCALE) {
#pragma unroll
            for (int m = 0; m < 4; m++) y.b[m] *= step_inv;
          }
          if (active) GsStore(slot(cf, c), lane_bytes, y);
        }
        if (cl < n) {
          GsPlv x;
          GsTip(rec(cl, c, 1), sl, kq, x);
          numl += wc * GsDot(al, x);
        } else {
          pipe.MatVec(al, u);  // (u has had its last use: the child's partial takes its place)
          if (RESCALE) {
#pragma unroll
            for (int m = 0; m < 4; m++) u.b[m] *= step_inv;
          }
          // the chained child's pre-order partial goes to memory only if its own step will not take
          // it from registers (it always will with one category; the store is then not needed)
          if (active && !(chain && cl == node - 1)) GsStore(slot(cl, c), lane_bytes, u);
        }
      }
      have_u = chain && cl >= n && cl == node - 1;  // the next node's own pre-order partial is in u
      // den is a pattern's sum (its four lanes), the numerators stay per lane: sum over lanes of num_lane . w_p / den_p is
      // an edge's derivative.  Two edges' sums at once on the vector ALU (the first ends up in lane 31, the second in
      // lane 63): the tip children's edges, then the node's own
      // (Without rescaling u . a_f . a_l is the pattern's likelihood at EVERY node -- the post-order pass left it in
      // `site` --, so the walk takes weight / site, once, and forms no sum and no quotient per step.  With rescaling the
      // steps' sums differ by the scale factors between the node and the root.)
      const double scale = RESCALE ? weight / GsPatternSum(den) : site_scale;
      if (cf < n || cl < n) {
        const double g = PairSum(numf * scale, numl * scale);
        const int child = lane < 32 ? cf : cl;
        if (active && (lane & 31) == 31 && child < n) grow[child] = g;
      }
      if (node != N - 1) {
        const double g = PairSum(numo * scale, 0.0);
        if (active && lane == 31) grow[node] = g;
      }
    }
  }
#else
  if (GRAD) {
    double* __restrict__ grow = part_grad + ((size_t)tree * tiles + tile) * N;
    if (lane == 0 && active) grow[N - 1] = 0.0;
    GsPlv y;              // pre-order partial of the child processed last; survives into the next iteration
    bool have_u = false;  // y is node's own pre-order partial (wave-uniform)
    for (int node = N - 1; node >= n; --node) {
      const int c0 = __builtin_amdgcn_readfirstlane(ch[(node - n) * 2]);
      const int c1 = __builtin_amdgcn_readfirstlane(ch[(node - n) * 2 + 1]);
      const int cl = c1 >= n ? c1 : c0, cf = c1 >= n ? c0 : c1;  // chained child (if any) last
      const int sf = cf < n ? tips[(size_t)cf * Ppad] : 0;
      const int sl = cl < n ? tips[(size_t)cl * Ppad] : 0;
      double den = 0.0, numf = 0.0, numl = 0.0;
      // Rescaled pre-order partials: a child's partial is divided by this node's POST-order factor
      // (any positive per-pattern factor cancels in num / den; this one keeps the products O(1), see
      // walk_hbm_kernel)
      const double step_inv = RESCALE ? *inv_at(node) : 1.0;
      for (int c = 0; c < C; c++) {
        GsPlv u, wf, wl, x;
        if (node == N - 1) {
          load_pi(u);
        } else if (have_u) {
          u = y;
        } else {
          GsLoad(slot(node, c), lane_bytes, u);
        }
        // child messages: a_f -> w_l = u . a_f (what the LAST child sees), a_l -> w_f = u . a_l
        if (cf < n) GsTip(rec(cf, c, 0), sf, kq, wl);
        else GsLoad(mslot(cf, c), lane_bytes, wl);
        if (cl < n) GsTip(rec(cl, c, 0), sl, kq, wf);
        else GsLoad(mslot(cl, c), lane_bytes, wf);
        const double wc = model[kGsCatWeight + c];
        {
          double sden = 0.0;
#pragma unroll
          for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const double af = wl.b[m][r], al = wf.b[m][r], uu = u.b[m][r];
              wf.b[m][r] = uu * al;
              wl.b[m][r] = uu * af;
              sden += wf.b[m][r] * af;
            }
          den += wc * sden;
        }
        // per child: pre-order partial q = P^T w (stored in place of the child's post-order partial
        // x), and the edge derivative in the reference's own form  pre^T (r_c Q) post = (Q^T q) . x r_c
        // (src/fat_beagle.cpp:101-160); a tip child reads dP's column instead.
        // (Tried and withdrawn: w^T (r_c Q) (P x) with the child's MESSAGE P x, which the pass has in hand anyway --
        // x need not be read again, chained partials need not be stored at all: 56.8 -> 54.2 ms per 4096 config-5
        // trees.  Q and the ROUNDED P of a codon model do not commute to better than about 1e-6 relative (V e^{Lt} V^-1
        // at 61 states), and on short branches with large derivatives that is a thousand tolerances away from the
        // reference's form: seeded sweep cases 104 and 119 of tests/test_gpu_fuzz.py.)
        const double rc = model[(deriv_mode ? kGsCatRateDeriv : kGsCatRate) + c];  // site-model pass: d r_c / d shape
        if (cf < n) {
          GsTip(rec(cf, c, 1), sf, kq, x);
          numf += wc * GsDot(wf, x);
        } else {
#if GS_X_EARLY
          GsLoad(slot(cf, c), lane_bytes, x);  // (x is read before q takes its place; in flight behind the two contractions)
          pipe.MatVec(wf, y);
          pipe.MatVec(y, wf);
#else
          pipe.MatVec(wf, y);
          pipe.MatVec(y, wf);
          GsLoad(slot(cf, c), lane_bytes, x);  // (x is read before q takes its place)
#endif
          numf += wc * rc * GsDot(wf, x);
          if (RESCALE) {
#pragma unroll
            for (int m = 0; m < 4; m++) y.b[m] *= step_inv;
          }
          if (active) GsStore(slot(cf, c), lane_bytes, y);
        }
        if (cl < n) {
          GsTip(rec(cl, c, 1), sl, kq, x);
          numl += wc * GsDot(wl, x);
        } else {
#if GS_X_EARLY
          GsLoad(slot(cl, c), lane_bytes, x);
          pipe.MatVec(wl, y);
          pipe.MatVec(y, wl);
#else
          pipe.MatVec(wl, y);
          pipe.MatVec(y, wl);
          GsLoad(slot(cl, c), lane_bytes, x);
#endif
          numl += wc * rc * GsDot(wl, x);
          if (RESCALE) {
#pragma unroll
            for (int m = 0; m < 4; m++) y.b[m] *= step_inv;
          }
          // the chained child's pre-order partial goes to memory only if its own step will not take
          // it from registers (it always will with one category; the store is then not needed)
          if (active && !(chain && cl == node - 1)) GsStore(slot(cl, c), lane_bytes, y);
        }
      }
      have_u = chain && cl >= n && cl == node - 1;  // the next node's own pre-order partial is in y
      // den is a pattern's sum (its four lanes), the numerators stay per lane: sum over lanes of num_lane . w_p / den_p is
      // the edge's derivative.  Both edges' sums at once on the vector ALU (the sum of the first ends up in lane 31, of the
      // second in lane 63) -- as 36 dependent ds_bpermute round trips these sums were a sixth of a step
      den = GsPatternSum(den);
      const double scale = weight / den;
      const double g = PairSum(numf * scale, numl * scale);
      if (active && (lane & 31) == 31) grow[lane < 32 ? cf : cl] = g;
    }
  }
#endif
}

size_t GsArenaDoublesPerTree(const BatchDims& d, int tiles, int want_gradient) {
  return (size_t)(want_gradient ? 2 : 1) * (d.taxon_count - 1) * d.category_count * tiles * 1024;
}
size_t GsImageDoublesPerTree(const BatchDims& d) {
  return (size_t)(d.node_count - 1) * d.category_count * 3 * 4096;
}

void LaunchGsWalk(const BatchDims& d, int S, const DeviceBatch& b, const int32_t* model_index,
                  const double* gs_model, int tree0, int chunk, int tiles, int want_gradient, int rescaling,
                  int deriv_mode, hipStream_t stream) {
  const dim3 grid((unsigned)((chunk + 7) / 8 * 8 * ((tiles + GS_WG_WAVES - 1) / GS_WG_WAVES))), block(GS_WG_WAVES * 64);
  const size_t lds = 2 * 4096 * sizeof(double);
  const int stride = GsScheduleStride(d);
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, block, lds, stream, d, S, tree0, chunk, tiles, stride, deriv_mode, b.children, b.sched,
                       b.images, model_index, gs_model, b.tip_states, b.weights, b.arena, b.scale_arena, b.part_ll,
                       b.part_grad);
  };
  if (want_gradient) {
    if (rescaling) launch(gs_walk_kernel<true, true>);
    else launch(gs_walk_kernel<true, false>);
  } else {
    if (rescaling) launch(gs_walk_kernel<false, true>);
    else launch(gs_walk_kernel<false, false>);
  }
}

}  // namespace bito_amd
