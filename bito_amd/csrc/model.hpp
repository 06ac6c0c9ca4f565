// model.hpp -- substitution / site model arithmetic shared by the host-side
// validation code and the per-tree device set-up kernel.
//
// What is computed (not how) follows the reference:
//   JC69/HKY/GTR rate matrix and eigendecomposition  src/substitution_model.cpp:20-187
//   Weibull-median category rates and d rate/d shape  src/site_model.cpp:37-62
//   parameter row layout                               src/block_specification.cpp:14-53
// Everything is FP64; the functions are __host__ __device__ so the same code
// validates on the host and runs one-thread-per-tree on the GPU.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

// The set-up arithmetic (rate matrix, eigensystem, category rates) is evaluated
// without fused multiply-add contraction and in a fixed operation order.  Errors in
// P(t) are coherent across all site patterns, so they are the part of the
// computation that decides whether two FP64 implementations agree to 1e-10 on a
// log-likelihood of magnitude 1e4 ("Numerical contract").
#pragma clang fp contract(off)

namespace bito_amd {

constexpr int kStates = 4;
constexpr int kMaxCategories = 16;

enum SubstitutionKind : int32_t { kJC69 = 0, kHKY = 1, kGTR = 2, kGY94 = 3 };

// Layout of one row of phylo_model_params_ (reference src/phylo_model.cpp:6-31):
// [substitution: frequencies(4) | rates] [site: Weibull_shape] [clock: clock_rate].
struct ModelSpec {
  int32_t substitution;
  int32_t category_count;
  int32_t weibull;       // 0 = "constant"
  int32_t strict_clock;  // 0 = "none"
  int32_t freq_start, rates_start, rates_len, shape_start, clock_start;
  int32_t param_count;
  int32_t state_count;  // 4, or 61 for the codon model "GY94" (general-state kernels, gs_kernels.hip)
  // log(-log(1 - (2i+1)/(2C))) of the Weibull category medians: depends on the category count only,
  // evaluated once on the host (general-state kernels)
  double weibull_log_l[16];
};

// exp() with a fixed operation sequence (argument reduction by ln 2, degree-13 Taylor polynomial in
// Horner form, every step an IEEE multiply or a correctly rounded fma, exact scaling by 2^k): the same
// bits on any IEEE machine.  About 1 ulp.  The general-state set-up uses it for exp(lambda t r_c) and
// the category rates: the O(t^2) entries of a codon P(t) are ill-conditioned,
// so results must not depend on which math library evaluated exp.
__host__ __device__ inline double DetExp(double x) {
  if (x < -745.0) return 0.0;
  if (x > 709.0) return HUGE_VAL;
  const double k = rint(x * 1.4426950408889634);
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}

// Per-tree model state produced by the set-up kernel and consumed by the
// transition-matrix and traversal kernels.  Row-major 4x4.
struct TreeModel {
  double V[16];
  double Vinv[16];
  double Q[16];
  double lambda[4];
  double pi[4];
  double cat_rate[kMaxCategories];
  double cat_weight[kMaxCategories];
  double cat_rate_deriv[kMaxCategories];  // d r_c / d shape
};

// Q_ij = r pi_j (i<j order AC,AG,AT,CG,CT,GT), Q_ji = r pi_i, rows sum to zero,
// normalised to one expected substitution per unit time.
__host__ __device__ inline void BuildQ(const double r[6], const double pi[4], double Q[16]) {
  Q[0 * 4 + 1] = r[0] * pi[1]; Q[1 * 4 + 0] = r[0] * pi[0];
  Q[0 * 4 + 2] = r[1] * pi[2]; Q[2 * 4 + 0] = r[1] * pi[0];
  Q[0 * 4 + 3] = r[2] * pi[3]; Q[3 * 4 + 0] = r[2] * pi[0];
  Q[1 * 4 + 2] = r[3] * pi[2]; Q[2 * 4 + 1] = r[3] * pi[1];
  Q[1 * 4 + 3] = r[4] * pi[3]; Q[3 * 4 + 1] = r[4] * pi[1];
  Q[2 * 4 + 3] = r[5] * pi[3]; Q[3 * 4 + 2] = r[5] * pi[2];
  double total = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double row = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (i != j) row += Q[i * 4 + j];
    Q[i * 4 + i] = -row;
    total += row * pi[i];
  }
#pragma unroll
  for (int i = 0; i < 16; i++) Q[i] /= total;
}

// One Jacobi rotation on the (p,q) plane of symmetric A, accumulated into U.
template <int p, int q>
__host__ __device__ inline void JacobiRotate(double A[16], double U[16]) {
  const double apq = A[p * 4 + q];
  if (apq == 0.0) return;
  const double theta = (A[q * 4 + q] - A[p * 4 + p]) / (2.0 * apq);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double akp = A[k * 4 + p], akq = A[k * 4 + q];
    A[k * 4 + p] = c * akp - s * akq;
    A[k * 4 + q] = s * akp + c * akq;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double apk = A[p * 4 + k], aqk = A[q * 4 + k];
    A[p * 4 + k] = c * apk - s * aqk;
    A[q * 4 + k] = s * apk + c * aqk;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double ukp = U[k * 4 + p], ukq = U[k * 4 + q];
    U[k * 4 + p] = c * ukp - s * ukq;
    U[k * 4 + q] = s * ukp + c * ukq;
  }
}

template <int i, int j>
__host__ __device__ inline void SortPair(double w[4], double U[16]) {
  if (w[j] < w[i]) {
    const double tw = w[i];
    w[i] = w[j];
    w[j] = tw;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double tu = U[k * 4 + i];
      U[k * 4 + i] = U[k * 4 + j];
      U[k * 4 + j] = tu;
    }
  }
}

// Symmetric 4x4 eigensolve (cyclic Jacobi, statically indexed so it stays in
// registers on the device).  Eigenvalues ascending; A is destroyed.
__host__ __device__ inline void SymmetricEigen4(double A[16], double U[16], double w[4]) {
#pragma unroll
  for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 32; sweep++) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[3] * A[3] + A[6] * A[6] + A[7] * A[7] +
                       A[11] * A[11];
    // Converged when every off-diagonal entry is below 1e-20 of the diagonal's scale: a rotation by such an entry
    // has c = 1 exactly and s * x below half an ulp of anything it is added to, so further sweeps (the iteration
    // converges quadratically: 1e-20, 1e-40, 1e-80, ... until the old bound of 1e-150) change neither the eigenvalues
    // nor the eigenvectors by a bit -- they cost a third of the set-up kernel's time.
    const double diag = A[0] * A[0] + A[5] * A[5] + A[10] * A[10] + A[15] * A[15];
    if (off < 1e-300 || off < 1e-40 * diag) break;
    JacobiRotate<0, 1>(A, U);
    JacobiRotate<0, 2>(A, U);
    JacobiRotate<0, 3>(A, U);
    JacobiRotate<1, 2>(A, U);
    JacobiRotate<1, 3>(A, U);
    JacobiRotate<2, 3>(A, U);
  }
  w[0] = A[0]; w[1] = A[5]; w[2] = A[10]; w[3] = A[15];
  SortPair<0, 1>(w, U); SortPair<0, 2>(w, U); SortPair<0, 3>(w, U);
  SortPair<1, 2>(w, U); SortPair<1, 3>(w, U); SortPair<2, 3>(w, U);
}

// Reversible-model eigendecomposition through the symmetrised matrix
// D^{1/2} Q D^{-1/2}: V = D^{-1/2} U, V^-1 = U^T D^{1/2}.
__host__ __device__ inline void EigenReversible(const double Q[16], const double pi[4],
                                                double V[16], double Vinv[16], double lam[4]) {
  double sq[4], A[16], U[16];
#pragma unroll
  for (int i = 0; i < 4; i++) sq[i] = sqrt(pi[i]);
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) A[i * 4 + j] = sq[i] * Q[i * 4 + j] / sq[j];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = i + 1; j < 4; j++) A[i * 4 + j] = A[j * 4 + i];
  SymmetricEigen4(A, U, lam);
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      V[i * 4 + j] = U[i * 4 + j] / sq[i];
      Vinv[i * 4 + j] = U[j * 4 + i] * sq[j];
    }
}

// Fills every field of TreeModel from one parameter row.  No validation here:
// the host checks the row before it is uploaded (ValidateParams).
// (two halves that share nothing, so that the set-up kernel can run them in different waves: rate matrix +
// eigensystem, and the site model's category rates)
__host__ __device__ inline void SetupSiteRates(const ModelSpec& spec, const double* row, TreeModel* m);

__host__ __device__ inline void SetupSubstitution(const ModelSpec& spec, const double* row, TreeModel* m) {
  if (spec.substitution == kJC69) {
    const double v[16] = {1.0, 2.0, 0.0, 0.5, 1.0, -2.0, 0.5, 0.0,
                          1.0, 2.0, 0.0, -0.5, 1.0, -2.0, -0.5, 0.0};
    const double vi[16] = {0.25, 0.25, 0.25, 0.25, 0.125, -0.125, 0.125, -0.125,
                           0.0, 1.0, 0.0, -1.0, 1.0, 0.0, -1.0, 0.0};
#pragma unroll
    for (int i = 0; i < 16; i++) {
      m->V[i] = v[i];
      m->Vinv[i] = vi[i];
      m->Q[i] = (i % 5 == 0) ? -1.0 : 1.0 / 3.0;
    }
    m->lambda[0] = 0.0;
    m->lambda[1] = m->lambda[2] = m->lambda[3] = -1.3333333333333333;
#pragma unroll
    for (int i = 0; i < 4; i++) m->pi[i] = 0.25;
  } else {
    double pi[4], Q[16];
#pragma unroll
    for (int i = 0; i < 4; i++) pi[i] = m->pi[i] = row[spec.freq_start + i];
    if (spec.substitution == kGTR) {
      double r[6];
#pragma unroll
      for (int i = 0; i < 6; i++) r[i] = row[spec.rates_start + i];
      BuildQ(r, pi, Q);
      EigenReversible(Q, pi, m->V, m->Vinv, m->lambda);
    } else {
      // HKY85: closed-form decomposition (Hasegawa, Kishino & Yano 1985).
      const double kappa = row[spec.rates_start];
      const double r[6] = {1.0, kappa, 1.0, 1.0, kappa, 1.0};
      BuildQ(r, pi, Q);
      const double pa = pi[0], pc = pi[1], pg = pi[2], pt = pi[3];
      const double pr = pa + pg, py = pc + pt;
      const double beta = -1.0 / (2.0 * (pr * py + kappa * (pa * pg + pc * pt)));
      m->lambda[0] = 0;
      m->lambda[1] = beta;
      m->lambda[2] = beta * (1 + py * (kappa - 1));
      m->lambda[3] = beta * (1 + pr * (kappa - 1));
#pragma unroll
      for (int i = 0; i < 16; i++) m->V[i] = m->Vinv[i] = 0.0;
      m->Vinv[0] = pa; m->Vinv[1] = pc; m->Vinv[2] = pg; m->Vinv[3] = pt;
      m->Vinv[4] = pa * py; m->Vinv[5] = -pc * pr; m->Vinv[6] = pg * py; m->Vinv[7] = -pt * pr;
      m->Vinv[9] = 1; m->Vinv[11] = -1;
      m->Vinv[12] = 1; m->Vinv[14] = -1;
      m->V[0] = m->V[4] = m->V[8] = m->V[12] = 1.0;
      m->V[1] = 1. / pr; m->V[5] = -1. / py; m->V[9] = 1. / pr; m->V[13] = -1. / py;
      m->V[6] = pt / py; m->V[14] = -pc / py;
      m->V[3] = pg / pr; m->V[11] = -pa / pr;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) m->Q[i] = Q[i];
  }
}

__host__ __device__ inline void SetupSiteRates(const ModelSpec& spec, const double* row, TreeModel* m) {
  const int C = spec.category_count;
  if (spec.weibull) {
    // Discretised Weibull, median of each equiprobable bin, scale 1, normalised
    // to mean rate 1; derivative of the normalised rate wrt the shape.
    const double shape = row[spec.shape_start];
    double mean = 0, dmean = 0;
    double du[kMaxCategories];
    for (int i = 0; i < C; i++) {
      const double quantile = (2.0 * i + 1.0) / (2.0 * C);
      const double l = -log(1.0 - quantile);
      const double r = pow(l, 1.0 / shape);
      m->cat_rate[i] = r;
      mean += r;
      du[i] = -r * log(l) / (shape * shape);
      dmean += du[i];
    }
    mean /= C;
    dmean /= C;
    for (int i = 0; i < C; i++) {
      m->cat_rate_deriv[i] = (du[i] * mean - m->cat_rate[i] * dmean) / (mean * mean);
      m->cat_rate[i] /= mean;
      m->cat_weight[i] = 1.0 / C;
    }
  } else {
    m->cat_rate[0] = 1.0;
    m->cat_weight[0] = 1.0;
    m->cat_rate_deriv[0] = 0.0;
  }
}

__host__ __device__ inline void SetupTreeModel(const ModelSpec& spec, const double* row, TreeModel* m) {
  SetupSubstitution(spec, row, m);
  SetupSiteRates(spec, row, m);
}

}  // namespace bito_amd
