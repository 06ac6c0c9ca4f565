// gp_engine.hip -- GPU executor for bito's GPOperation streams (include/bito_amd_gp.h).
//
// Arena in HBM: PLV i is [4][Ppad] doubles (state-major rows, so a wave touches 64
// consecutive patterns), with one int32 rescaling count per (PLV, pattern).  The reference
// keeps ONE rescaling count per PLV decided from the whole-PLV maximum
// (src/gp_engine.cpp:583-597); here the same rule is applied per pattern column, which
// removes the only cross-pattern coupling of the PLV ops -- log-likelihoods are invariant
// to how often a column was rescaled (the reference's own invariance test,
// src/gp_doctest.cpp:348-360), so results agree to rounding.  With that, a whole run of
// ops is one kernel launch: each thread owns a pattern and interprets the op stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <limits>
#include <cstdio>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>
#include <algorithm>

#include <cstring>
#include <list>
#include <memory>

#include "../../include/bito_amd.h"
#include "../../include/bito_amd_gp.h"
#include "gp_schedule.hpp"

// No FMA contraction anywhere in this file (round 5).  An operation's arithmetic is inlined into several kernels -- the
// stream interpreter, the levelled interpreter, the optimiser's evaluations -- and the executor promises that a scheduled
// sweep is bit for bit the sequential one (gp_schedule.hpp; tests/test_gp.py holds it on the device): with contraction
// left to the backend, where a multiply meets an add could differ from one inlining context to the next.  (The tests'
// CPU restatement is compiled the same way: one rounding per operation.)  ONE fused multiply-add is written out, and
// therefore the same in every context: the per-pattern likelihood A + B e of EdgeFunction (fma(B, e, A), the form the
// device ran in rounds 3-4; the restatement rounds the product first -- half an ulp of a value whose logarithm is held
// to 1e-10).  These kernels are bound by dependent launches
// and latency, not by the vector ALU: the unfused multiply-adds cost nothing that can be measured.
#pragma clang fp contract(off)

// a stream the executor has scheduled before (gp_schedule.hpp), found again by content
struct GpCachedSchedule {
  uint64_t hash = 0;
  bool reorder = true;
  std::vector<bito_amd_gp_op> ops;  // the stream as the caller gave it
  std::vector<uint64_t> side;
  bito_amd_gp_schedule::Schedule schedule;
  // its image on the device (scheduled operations, level offsets, side array), uploaded when the schedule is first run:
  // a replayed stream -- the three schedules of GPInstance::EstimateBranchLengths alternate until convergence -- crosses
  // PCIe once
  int device = 0;
  bito_amd_gp_op* d_ops = nullptr;
  int64_t* d_levels = nullptr;
  uint64_t* d_side = nullptr;
  bool on_device = false;
  size_t device_bytes = 0;
  ~GpCachedSchedule() {
    if (d_ops || d_levels || d_side) (void)hipSetDevice(device);
    if (d_ops) (void)hipFree(d_ops);
    if (d_levels) (void)hipFree(d_levels);
    if (d_side) (void)hipFree(d_side);
  }
};

constexpr size_t kScheduleCacheBytes = 64u << 20;  // device memory the cached schedules of one engine may hold

struct bito_amd_gp_engine {
  int device = 0, n = 0, P = 0, Ppad = 0, nodes = 0, gpcsps = 0, plvs = 0;
  int64_t spare_plvs = 0, spare_gpcsps = 0;  // GrowSparePLVs / GrowSpareGPCSPs: ids behind the DAG's own
  double threshold = 1e-40, log_threshold = 0;
  double *plv = nullptr, *weights = nullptr, *bl = nullptr, *q = nullptr, *ll = nullptr, *marginal = nullptr;
  double* scratch = nullptr;
  double *diff = nullptr, *coef = nullptr;  // DAGBranchHandler differences_; per-pattern optimiser coefficients
  int method = 0, significant_digits = 10, optimization_count = 0;
  int opt_waves = 4;  // waves per optimiser workgroup of the scheduled launches (BITO_AMD_GP_OPT_WAVES: 1, 2, 4, 8 or 16)
  int* counts = nullptr;
  bito_amd_gp_op* d_ops = nullptr;
  uint64_t* d_side = nullptr;
  int64_t* d_offsets = nullptr;
  size_t ops_cap = 0, side_cap = 0, offsets_cap = 0, coef_blocks = 1;  // coef holds coef_blocks x [2][Ppad]
  std::list<std::shared_ptr<GpCachedSchedule>> schedules;  // most recently used first
  // diagnostics: every function evaluation of the Brent optimisers as rows (edge, x, f, kind), see bito_amd_gp.h
  double* trace_rows = nullptr;
  unsigned long long* trace_cursor = nullptr;
  int64_t trace_capacity = 0;
  std::string err;
  ~bito_amd_gp_engine() {
    (void)hipSetDevice(device);
    for (void* p : {(void*)plv, (void*)weights, (void*)bl, (void*)q, (void*)ll, (void*)marginal, (void*)scratch, (void*)diff, (void*)coef,
                    (void*)counts, (void*)d_ops, (void*)d_side, (void*)d_offsets, (void*)trace_rows, (void*)trace_cursor})
      if (p) (void)hipFree(p);
  }
};

namespace {

__constant__ double cV[16] = {1.0, 2.0, 0.0, 0.5, 1.0, -2.0, 0.5, 0.0, 1.0, 2.0, 0.0, -0.5, 1.0, -2.0, -0.5, 0.0};
__constant__ double cVi[16] = {0.25, 0.25, 0.25, 0.25, 0.125, -0.125, 0.125, -0.125,
                               0.0, 1.0, 0.0, -1.0, 1.0, 0.0, -1.0, 0.0};
constexpr double kLam = -1.3333333333333333;

// P(t), and optionally P'(t), P''(t), through the eigensystem like the reference
// (src/gp_engine.cpp:341-364); fixed operation order, no FMA contraction (see model.hpp).
__device__ inline void Matrices(double t, double* M, double* dM, double* ddM) {
#pragma clang fp contract(off)
  const double e = exp(kLam * t);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0, d = 0, dd = 0;
      for (int k = 0; k < 4; k++) {
        const double ek = k == 0 ? 1.0 : e, lk = k == 0 ? 0.0 : kLam;
        s += cV[i * 4 + k] * ek * cVi[k * 4 + j];
        d += cV[i * 4 + k] * (lk * ek) * cVi[k * 4 + j];
        dd += cV[i * 4 + k] * (lk * lk * ek) * cVi[k * 4 + j];
      }
      M[i * 4 + j] = s;
      if (dM) dM[i * 4 + j] = d;
      if (ddM) ddM[i * 4 + j] = dd;
    }
}

__device__ inline double LogAdd(double x, double y) {
  if (y > x) { const double t = x; x = y; y = t; }
  if (x == -INFINITY) return x;
  const double nd = y - x;
  if (nd < -36.04365338911715) return x;
  return x + log(1.0 + exp(nd));
}

// One per-pattern operation for pattern p (everything in GPOperation except the optimiser and the SBN update)
__device__ inline void PatternOp(const bito_amd_gp_op& op, int p, const uint64_t* __restrict__ side,
                                 double* __restrict__ plv, int* __restrict__ counts, const double* __restrict__ bl,
                                 const double* __restrict__ q, double* __restrict__ ll, double* __restrict__ marginal,
                                 int Ppad, double threshold, double log_threshold) {
  auto cell = [&](uint64_t idx, int i) -> double& { return plv[((size_t)idx * 4 + i) * Ppad + p]; };
  auto cnt = [&](uint64_t idx) -> int& { return counts[(size_t)idx * Ppad + p]; };
  switch (op.opcode) {
    case BITO_AMD_GP_ZERO_PLV:
      for (int i = 0; i < 4; i++) cell(op.a, i) = 0.0;
      cnt(op.a) = 0;
      break;
    case BITO_AMD_GP_SET_TO_STATIONARY_DISTRIBUTION:
      for (int i = 0; i < 4; i++) cell(op.a, i) = q[op.b] * 0.25;
      cnt(op.a) = 0;
      break;
    case BITO_AMD_GP_INCREMENT_WITH_WEIGHTED_EVOLVED_PLV: {
      double M[16];
      Matrices(bl[op.b], M, nullptr, nullptr);
      const int diff = cnt(op.c) - cnt(op.a);
      const double f = (diff == 0 ? 1.0 : pow(threshold, (double)diff)) * q[op.b];
      double s[4];
      for (int i = 0; i < 4; i++) s[i] = cell(op.c, i);
      for (int i = 0; i < 4; i++)
        cell(op.a, i) += f * (M[i * 4] * s[0] + M[i * 4 + 1] * s[1] + M[i * 4 + 2] * s[2] + M[i * 4 + 3] * s[3]);
      break;
    }
    case BITO_AMD_GP_MULTIPLY: {
      double d[4], mx = 0;
      for (int i = 0; i < 4; i++) {
        d[i] = cell(op.b, i) * cell(op.c, i);
        mx = fmax(mx, d[i]);
      }
      int c = cnt(op.b) + cnt(op.c);
      if (mx != 0) {  // RescalePLVIfNeeded, per pattern column
        int extra = 0;
        while (mx < threshold) {
          mx /= threshold;
          extra++;
        }
        if (extra) {
          const double f = pow(threshold, (double)extra);
          for (int i = 0; i < 4; i++) d[i] /= f;
          c += extra;
        }
      }
      for (int i = 0; i < 4; i++) cell(op.a, i) = d[i];
      cnt(op.a) = c;
      break;
    }
    case BITO_AMD_GP_LIKELIHOOD: {
      double M[16];
      Matrices(bl[op.a], M, nullptr, nullptr);
      double s = 0;
      for (int i = 0; i < 4; i++)
        s += cell(op.c, i) * (M[i * 4] * cell(op.b, 0) + M[i * 4 + 1] * cell(op.b, 1) + M[i * 4 + 2] * cell(op.b, 2) +
                              M[i * 4 + 3] * cell(op.b, 3));
      ll[(size_t)op.a * Ppad + p] = log(s) + (cnt(op.b) + cnt(op.c)) * log_threshold;
      break;
    }
    case BITO_AMD_GP_RESET_MARGINAL_LIKELIHOOD:
      marginal[p] = -INFINITY;
      break;
    case BITO_AMD_GP_INCREMENT_MARGINAL_LIKELIHOOD: {
      double s = 0;
      for (int i = 0; i < 4; i++) s += cell(op.a, i) * cell(op.c, i);
      const double row = log(s) + cnt(op.c) * log_threshold;
      marginal[p] = LogAdd(marginal[p], row);
      ll[(size_t)op.b * Ppad + p] = row - log(q[op.b]);
      break;
    }
    case BITO_AMD_GP_PREP_FOR_MARGINALIZATION: {
      int mn = cnt(side[op.b]);
      for (uint32_t k = 1; k < op.count; k++) mn = min(mn, cnt(side[op.b + k]));
      cnt(op.a) = mn;
      break;
    }
    default:
      break;
  }
}

__global__ void __launch_bounds__(64)
gp_ops_kernel(const bito_amd_gp_op* __restrict__ ops, int64_t op_count, const int64_t* __restrict__ offsets,
              const uint64_t* __restrict__ side, double* __restrict__ plv, int* __restrict__ counts,
              const double* __restrict__ bl, const double* __restrict__ q, double* __restrict__ ll,
              double* __restrict__ marginal, int P, int Ppad, double threshold, double log_threshold) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  if (offsets) {
    // independent sub-streams side by side: sub-stream y owns [offsets[y], offsets[y + 1]); op_count is the
    // number of sub-streams, grid.y (capped at 65535 by the launcher) strides over them
    for (int64_t y = blockIdx.y; y < op_count; y += gridDim.y) {
      const bito_amd_gp_op* mine = ops + offsets[y];
      const int64_t n = offsets[y + 1] - offsets[y];
      for (int64_t o = 0; o < n; o++)
        PatternOp(mine[o], p, side, plv, counts, bl, q, ll, marginal, Ppad, threshold, log_threshold);
    }
    return;
  }
  for (int64_t o = 0; o < op_count; o++)
    PatternOp(ops[o], p, side, plv, counts, bl, q, ll, marginal, Ppad, threshold, log_threshold);
}

// A per-pattern segment whose operations have been sorted into dependency levels (LevelSegment below): one
// workgroup per tile of 64 patterns, kLevelWaves waves; wave w takes operations w, w + kLevelWaves, ... of the
// level, a workgroup barrier separates levels.  The dependent chain of a pass is then as long as the
// schedule is DEEP (DS1 ten-tree DAG: 57 levels for 1082 PopulatePLVs operations), not as long as it has
// operations, and nothing is launched per level.
constexpr int kLevelWaves = 16;

__global__ void __launch_bounds__(64 * kLevelWaves)
gp_levels_kernel(const bito_amd_gp_op* __restrict__ ops, const int64_t* __restrict__ level_offsets, int level_count,
                 const uint64_t* __restrict__ side, double* __restrict__ plv, int* __restrict__ counts,
                 const double* __restrict__ bl, const double* __restrict__ q, double* __restrict__ ll,
                 double* __restrict__ marginal, int P, int Ppad, double threshold, double log_threshold) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  const bool live = p < P;
  for (int level = 0; level < level_count; level++) {
    const int64_t first = level_offsets[level], last = level_offsets[level + 1];
    if (live)
      for (int64_t o = first + threadIdx.y; o < last; o += kLevelWaves)
        PatternOp(ops[o], p, side, plv, counts, bl, q, ll, marginal, Ppad, threshold, log_threshold);
    __threadfence_block();
    __syncthreads();  // the level's results are visible to every wave of the tile
  }
}

// dst[map[row]] = src[row] for rows of row_len elements (GrowPLVs / GrowGPCSPs with a reindexer)
template <typename T>
__global__ void __launch_bounds__(256)
gp_permute_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, const int64_t* __restrict__ map, size_t row_len) {
  // rows on grid.x (up to 2^31 - 1 of them), column chunks on grid.y
  const size_t row = blockIdx.x;
  const T* from = src + row * row_len;
  T* to = dst + (size_t)map[row] * row_len;
  for (size_t i = (size_t)blockIdx.y * blockDim.x + threadIdx.x; i < row_len; i += (size_t)gridDim.y * blockDim.x) to[i] = from[i];
}

// Launches the row permutation (nothing to do for zero rows) and reports a rejected launch: the
// caller frees the source buffers afterwards, so a silent no-op would lose every row.
template <typename T>
static hipError_t PermuteRows(const T* src, T* dst, const int64_t* map, size_t rows, size_t row_len, unsigned chunks) {
  if (rows == 0 || row_len == 0) return hipSuccess;
  if (rows > 0x7fffffffu) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gp_permute_rows_kernel<T>, dim3((unsigned)rows, chunks), dim3(256), 0, 0, src, dst, map, row_len);
  return hipGetLastError();
}

// block per row: out[row] = sum_p w_p * rows[row][p]
__global__ void __launch_bounds__(256)
gp_weighted_rows_kernel(const double* __restrict__ rows, const double* __restrict__ weights, int P, int Ppad,
                        double* __restrict__ out) {
  __shared__ double sh[4];
  double acc = 0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) acc += weights[p] * rows[(size_t)blockIdx.x * Ppad + p];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void __launch_bounds__(256)
gp_derivatives_kernel(const double* __restrict__ plv, const int* __restrict__ counts, const double* __restrict__ weights,
                      double t, uint64_t rootward, uint64_t leafward, int P, int Ppad, double log_threshold,
                      double* __restrict__ out) {
  __shared__ double sh[3][4];
  double M[16], dM[16], ddM[16];
  Matrices(t, M, dM, ddM);
  double a0 = 0, a1 = 0, a2 = 0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    double r[4], x[4];
    for (int i = 0; i < 4; i++) {
      r[i] = plv[((size_t)rootward * 4 + i) * Ppad + p];
      x[i] = plv[((size_t)leafward * 4 + i) * Ppad + p];
    }
    double l = 0, a = 0, b = 0;
    for (int i = 0; i < 4; i++) {
      l += r[i] * (M[i * 4] * x[0] + M[i * 4 + 1] * x[1] + M[i * 4 + 2] * x[2] + M[i * 4 + 3] * x[3]);
      a += r[i] * (dM[i * 4] * x[0] + dM[i * 4 + 1] * x[1] + dM[i * 4 + 2] * x[2] + dM[i * 4 + 3] * x[3]);
      b += r[i] * (ddM[i * 4] * x[0] + ddM[i * 4 + 1] * x[1] + ddM[i * 4 + 2] * x[2] + ddM[i * 4 + 3] * x[3]);
    }
    const double resc = (counts[(size_t)rootward * Ppad + p] + counts[(size_t)leafward * Ppad + p]) * log_threshold;
    a0 += weights[p] * (log(l) + resc);
    a1 += weights[p] * (a / l);
    a2 += weights[p] * ((b * l - a * a) / (l * l));
  }
  double v[3] = {a0, a1, a2};
  for (int k = 0; k < 3; k++) {
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 3) out[threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

// ---- OptimizeBranchLength on the device --------------------------------------
// One workgroup runs the whole one-dimensional optimisation of an edge: the per-pattern
// likelihood of the edge as a function of its length t is l_p(t) = A_p + B_p exp(lambda t) under
// JC69 (A, B from the rootward and leafward PLVs through the eigenbasis), so after one pass over
// the PLVs every function evaluation of Brent / Newton / gradient ascent is a block reduction
// over 2 doubles per pattern -- no host round trip per evaluation as in the reference
// (src/dag_branch_handler.cpp:123-300 calling back into GPEngine, src/gp_engine.cpp:603-661).
// The optimisers restate src/optimization.hpp:71-417; constants from src/dag_branch_handler.hpp:266-295.
struct OptSettings {
  int method, significant_digits, check_convergence;
  // diagnostics (bito_amd_gp_set_optimizer_trace): rows of (edge, x, f, kind), appended by thread 0 of the workgroup
  double* trace_rows;
  unsigned long long* trace_cursor;
  long long trace_capacity;
};

constexpr double kMinLogBl = -13.9, kMaxLogBl = 1.1, kNewtonEps = 1e-10, kStep = 5e-4, kLogStep = 1.0005,
                 kDiffThreshold = 1e-15;
constexpr int kOptMaxIter = 1000;
// Waves of the workgroup that optimises an edge.  Every function evaluation of an optimiser is a reduction over the
// patterns -- a logarithm and two divisions per pattern -- and an optimisation is a dependent chain of some thirty of
// them, an edge at a time: what counts is the latency of ONE evaluation, and that is two workgroup barriers and the
// cross-wave sum as much as the arithmetic.  Measured on the DS1 ten-tree DAG (bench.py --workload gp, 118 optimised edges
// of 934 patterns per sweep): four waves 4.98 ms per sweep (42 us per edge), sixteen waves -- one pattern per thread --
// 7.22 ms.
// Round 6: the number of waves is the LAUNCH's (blockDim.x / 64; scratch sized for sixteen).  Four is the default and the
// only form a device has timed.  Sixteen -- a pattern per thread at DS1's size -- lost in round 4, when an evaluation was
// two barriers and three cross-wave sums; since round 5 Brent's trial points are one barrier and one sum, and what that
// changes only a device can say: BITO_AMD_GP_OPT_WAVES = 1, 2, 8 or 16 (read when the engine is created; one wave: no
// cross-wave sum at all) for scripts/gpu_round6.sh.
// (The wave count is part of the summation order: other counts give other last bits, each held to the checker's bars.)
constexpr int kOptWaves = 4, kOptThreads = 64 * kOptWaves, kOptMaxWaves = 16;

struct EdgeFunction {
  const double* A;
  const double* B;
  const double* weights;
  double* sh;  // [2][3][kOptMaxWaves] block-reduction scratch in LDS, the two halves used by alternate evaluations
  double resc;
  int P;
  // diagnostics: the optimiser's evaluations are recorded when a trace buffer is set
  double* trace_rows;
  unsigned long long* trace_cursor;
  long long trace_capacity;
  double edge;
  // Evaluations alternate between the two halves of sh, so ONE barrier per evaluation is enough (round 5; two before):
  // evaluation k + 1 writes the half evaluation k - 1 was read from, and no thread gets to that write before it has
  // passed evaluation k's barrier -- which every thread reaches only after it has read evaluation k - 1's sums.
  int phase;

  // log-likelihood and its first two derivatives in t; every thread of the block gets the values
  __device__ void operator()(double t, double out[3]) {
    const double e = exp(kLam * t);
    double a0 = 0, a1 = 0, a2 = 0;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
      const double be = B[p] * e;
      const double l = fma(B[p], e, A[p]), d = kLam * be, dd = kLam * kLam * be;
      const double w = weights[p];
      a0 += w * log(l);
      a1 += w * (d / l);
      a2 += w * ((dd * l - d * d) / (l * l));
    }
    double v[3] = {a0, a1, a2};
    const int waves = (int)(blockDim.x >> 6);
    double* const buf = sh + (phase++ & 1) * 3 * kOptMaxWaves;
    for (int k = 0; k < 3; k++) {
      for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
      if ((threadIdx.x & 63) == 0) buf[k * kOptMaxWaves + (threadIdx.x >> 6)] = v[k];
    }
    __syncthreads();
    double total[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++)
      for (int w = 0; w < waves; w++) total[k] += buf[k * kOptMaxWaves + w];  // (fixed order: every thread the same bits)
    out[0] = total[0] + resc;
    out[1] = total[1];
    out[2] = total[2];
  }
  // The log-likelihood alone: what Brent asks for at every trial point (brent_nongrad_func, src/gp_engine.cpp:605-612).
  // The same per-pattern terms in the same order and the same reduction tree as out[0] above -- the same bits -- without
  // the two derivative sums (two divisions per pattern, two of the three cross-wave sums).
  __device__ double Value(double t) {
    const double e = exp(kLam * t);
    double a0 = 0;
    for (int p = threadIdx.x; p < P; p += blockDim.x) a0 += weights[p] * log(fma(B[p], e, A[p]));
    const int waves = (int)(blockDim.x >> 6);
    double* const buf = sh + (phase++ & 1) * 3 * kOptMaxWaves;
    for (int o = 32; o > 0; o >>= 1) a0 += __shfl_xor(a0, o);
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = a0;
    __syncthreads();
    double total = 0;
    for (int w = 0; w < waves; w++) total += buf[w];
    return total + resc;
  }
  // brent_nongrad_func: x is the LOG branch length.  kind (trace only): 0 the handler's evaluation of the current
  // length, 1 Brent's first point, 2 a trial point, 3 the gradient variant's second trial.
  __device__ void Trace(double x, double f, int kind) const {
    if (trace_rows && threadIdx.x == 0) {
      const unsigned long long at = atomicAdd(trace_cursor, 1ull);
      if ((long long)at < trace_capacity) {
        double* row = trace_rows + 4 * at;
        row[0] = edge; row[1] = x; row[2] = f; row[3] = (double)kind;
      }
    }
  }
  __device__ double NegLL(double x, int kind) {
    const double f = -Value(exp(x));
    Trace(x, f, kind);
    return f;
  }
};

// Optimization::BrentMinimize / BrentMinimizeWithGradients (src/optimization.hpp:71-331)
// f_guess = f(guess): the handler has just evaluated the current length (src/dag_branch_handler.cpp:160-163) and Brent's
// first point is that length again (src/optimization.hpp:93-94) -- the same function of the same argument, so the value
// is handed over instead of being summed a second time (the trace still shows the reference's two evaluations).
__device__ void BrentMinimize(EdgeFunction& f, bool with_gradients, double guess, double f_guess, double mn, double mx,
                              int significant_digits, int max_iter, double step_size, double* x_out, double* fx_out) {
  const double tolerance = ldexp(1.0, 1 - significant_digits);
  const double golden = 0.3819660f;
  double x, w, v, u, delta, delta2, fu, fv, fw, fx, mid, fract1, fract2;
  w = v = x = guess;
  fw = fv = fx = f_guess;
  f.Trace(x, fx, 1);
  delta2 = delta = 0;
  int count = max_iter;
  do {
    mid = (mn + mx) / 2;
    fract1 = tolerance * fabs(x) + tolerance / 4;
    fract2 = 2 * fract1;
    if (fabs(x - mid) <= (fract2 - (mx - mn) / 2)) break;
    bool use_bisection = true;
    if (fabs(delta2) > fract1) {
      double r = (x - w) * (fx - fv);
      double q = (x - v) * (fx - fw);
      double p = (x - v) * q - (x - w) * r;
      q = 2 * (q - r);
      if (q > 0) p = -p;
      q = fabs(q);
      const double td = delta2;
      delta2 = delta;
      if (!(fabs(p) >= fabs(q * td / 2)) && !(p <= q * (mn - x)) && !(p >= q * (mx - x))) {
        delta = p / q;
        u = x + delta;
        if (((u - mn) < fract2) || ((mx - u) < fract2)) delta = (mid - x) < 0 ? -fabs(fract1) : fabs(fract1);
        use_bisection = false;
      }
    }
    if (use_bisection) {
      delta2 = (x >= mid) ? mn - x : mx - x;
      delta = golden * delta2;
    }
    u = (fabs(delta) >= fract1) ? x + delta : (delta > 0 ? x + fabs(fract1) : x - fabs(fract1));
    fu = f.NegLL(u, 2);
    bool accepted = false;
    if (fu <= fx) {
      if (u >= x) mn = x; else mx = x;
      v = w; w = x; x = u;
      fv = fw; fw = fx; fx = fu;
      accepted = true;
    } else if (with_gradients) {
      double o[3];
      const double t = exp(x);
      f(t, o);
      const double u2 = x - step_size * (-t * o[1]);
      const double fu2 = f.NegLL(u2, 3);
      if (fu2 <= fx) {
        if (u2 >= x) mn = x; else mx = x;
        v = w; w = x; x = u2;
        fv = fw; fw = fx; fx = fu2;
        accepted = true;
      }
    }
    if (!accepted) {
      if (u < x) mn = u; else mx = u;
      if ((fu <= fw) || (w == x)) {
        v = w; w = u;
        fv = fw; fw = fu;
      } else if ((fu <= fv) || (v == x) || (v == w)) {
        v = u;
        fv = fu;
      }
    }
  } while (--count);
  *x_out = x;
  *fx_out = fx;
}

// The whole optimisation of one edge by one workgroup (four waves, or what the launch has); sh[2][3 kOptMaxWaves] and sh_resc[kOptMaxWaves] are LDS scratch,
// coef holds 2 * Ppad doubles private to the workgroup.
__device__ void OptimizeEdge(const bito_amd_gp_op& op, const double* __restrict__ plv, const int* __restrict__ counts,
                             const double* __restrict__ weights, double* __restrict__ bl, double* __restrict__ diff,
                             double* __restrict__ coef, double* sh, double* sh_resc, int P, int Ppad,
                             double log_threshold, const OptSettings& cfg) {
  // a = leafward_, b = rootward_, c = gpcsp_ (src/gp_operation.hpp:118-127)
  const uint64_t leafward = op.a, rootward = op.b, edge = op.c;
  __syncthreads();  // bl / diff writes of the previous op are visible; coef may be overwritten
  if (cfg.check_convergence && diff[edge] < kDiffThreshold) return;  // dag_branch_handler.cpp:127-131
  // eigenbasis coefficients: l_p(t) = sum_k (r^T V)_k (V^-1 x)_k exp(lambda_k t)
  double resc = 0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    double r[4], x[4];
    for (int i = 0; i < 4; i++) {
      r[i] = plv[((size_t)rootward * 4 + i) * Ppad + p];
      x[i] = plv[((size_t)leafward * 4 + i) * Ppad + p];
    }
    double c[4];
    for (int k = 0; k < 4; k++) {
      const double rv = r[0] * cV[k] + r[1] * cV[4 + k] + r[2] * cV[8 + k] + r[3] * cV[12 + k];
      const double vx = cVi[k * 4] * x[0] + cVi[k * 4 + 1] * x[1] + cVi[k * 4 + 2] * x[2] + cVi[k * 4 + 3] * x[3];
      c[k] = rv * vx;
    }
    coef[p] = c[0];
    coef[Ppad + p] = c[1] + c[2] + c[3];
    resc += weights[p] * ((counts[(size_t)rootward * Ppad + p] + counts[(size_t)leafward * Ppad + p]) * log_threshold);
  }
  for (int s = 32; s > 0; s >>= 1) resc += __shfl_xor(resc, s);
  if ((threadIdx.x & 63) == 0) sh_resc[threadIdx.x >> 6] = resc;
  __syncthreads();
  double resc_total = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) resc_total += sh_resc[w];
  EdgeFunction f{coef, coef + Ppad, weights, sh, resc_total, P, cfg.trace_rows, cfg.trace_cursor, cfg.trace_capacity, (double)edge, 0};
  const double current = bl[edge];
  double result = current;
  switch (cfg.method) {
    case 0:
    case 1: {  // BrentOptimization(WithGradients), dag_branch_handler.cpp:150-211
      const double cur_log = log(current);
      const double cur_nll = f.NegLL(cur_log, 0);
      double x, fx;
      BrentMinimize(f, cfg.method == 1, cur_log, cur_nll, kMinLogBl, kMaxLogBl, cfg.significant_digits, kOptMaxIter, kLogStep,
                    &x, &fx);
      result = fx > cur_nll ? exp(cur_log) : exp(x);
      break;
    }
    case 2: {  // GradientAscent (optimization.hpp:333-347); the floor is the handler's min LOG length, as there
      const double tolerance = pow(10.0, -cfg.significant_digits);
      double x = current;
      for (int iter = 0;; iter++) {
        double v[3];
        f(x, v);
        x = fmax(x + v[1] * kStep, kMinLogBl);
        if (fabs(v[1]) < fabs(v[0]) * tolerance || iter >= kOptMaxIter) break;
      }
      result = x;
      break;
    }
    case 3: {  // LogSpaceGradientAscent (optimization.hpp:349-367)
      const double tolerance = pow(10.0, -cfg.significant_digits), min_x = exp(kMinLogBl);
      double x = current;
      for (int iter = 0;; iter++) {
        double v[3];
        const double y = log(x);
        f(x, v);
        x = fmax(exp(y + x * v[1] * kLogStep), min_x);
        if (fabs(v[1]) < fabs(v[0]) * tolerance || iter >= kOptMaxIter) break;
      }
      result = x;
      break;
    }
    default: {  // NewtonRaphsonOptimization in the log length (optimization.hpp:369-405, gp_engine.cpp:643-655)
      const double tolerance = pow(10.0, -cfg.significant_digits);
      double x = log(current);
      for (int iter = 0;; iter++) {
        double v[3];
        const double t = exp(x);
        f(t, v);
        const double f1 = t * v[1], f2 = f1 + t * t * v[2];
        if (fabs(f2) < kNewtonEps) break;
        double new_x = x - f1 / f2;
        if (new_x < kMinLogBl) new_x = x - 0.5 * (x - kMinLogBl);
        if (new_x > kMaxLogBl) new_x = x - 0.5 * (x - kMaxLogBl);
        const double delta = fabs(x - new_x);
        if (delta < tolerance || fabs(f1) < fabs(v[0]) * tolerance || iter == kOptMaxIter) break;
        x = new_x;
      }
      result = exp(x);
      break;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bl[edge] = result;
    diff[edge] = fabs(current - result);
  }
}

// One workgroup per optimisation of the launch.  The optimisations of a launch are independent of one another
// (gp_schedule.hpp: equal optimiser depth, so no PLV or branch length one writes is read or written by another); each has
// its own coefficient block.
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
gp_optimize_kernel(const bito_amd_gp_op* __restrict__ ops, const double* __restrict__ plv,
                   const int* __restrict__ counts, const double* __restrict__ weights, double* __restrict__ bl,
                   double* __restrict__ diff, double* __restrict__ coef, int P, int Ppad, double log_threshold,
                   OptSettings cfg) {
  __shared__ double sh[2 * 3 * kOptMaxWaves];
  __shared__ double sh_resc[kOptMaxWaves];
  OptimizeEdge(ops[blockIdx.x], plv, counts, weights, bl, diff, coef + (size_t)blockIdx.x * 2 * Ppad, sh, sh_resc, P, Ppad,
               log_threshold, cfg);
}

// A workgroup interprets a whole sub-stream, per-pattern ops and optimiser ops alike: thread t owns the
// patterns t, t + kOptThreads, ... in every op, so per-pattern ops need no barrier between them; an optimiser op
// is a block-wide reduction and publishes the new branch length to the whole workgroup.  Grid = one
// workgroup per independent sub-stream (NNI proposals with optimize_new_edges).
__global__ void __launch_bounds__(kOptThreads)
gp_block_stream_kernel(const bito_amd_gp_op* __restrict__ ops, const int64_t* __restrict__ offsets,
                       const uint64_t* __restrict__ side, double* __restrict__ plv, int* __restrict__ counts,
                       const double* __restrict__ weights, double* __restrict__ bl, const double* __restrict__ q,
                       double* __restrict__ ll, double* __restrict__ marginal, double* __restrict__ diff,
                       double* __restrict__ coef, int P, int Ppad, double threshold, double log_threshold,
                       OptSettings cfg) {
  __shared__ double sh[2 * 3 * kOptMaxWaves];
  __shared__ double sh_resc[kOptMaxWaves];
  const int64_t first = offsets[blockIdx.x], last = offsets[blockIdx.x + 1];
  double* my_coef = coef + (size_t)blockIdx.x * 2 * Ppad;
  for (int64_t o = first; o < last; o++) {
    const bito_amd_gp_op op = ops[o];
    if (op.opcode == BITO_AMD_GP_OPTIMIZE_BRANCH_LENGTH) {
      OptimizeEdge(op, plv, counts, weights, bl, diff, my_coef, sh, sh_resc, P, Ppad, log_threshold, cfg);
      __threadfence_block();
      __syncthreads();  // bl[edge] of thread 0 is visible to every thread's next op
    } else {
      for (int p = threadIdx.x; p < P; p += blockDim.x)
        PatternOp(op, p, side, plv, counts, bl, q, ll, marginal, Ppad, threshold, log_threshold);
    }
  }
}

int Fail(bito_amd_gp_engine* e, int code, const std::string& msg) {
  e->err = msg;
  return code;
}

#define GP_TRY(e, call)                                                                                     \
  do {                                                                                                      \
    hipError_t rc_ = (call);                                                                                \
    if (rc_ != hipSuccess) return Fail(e, BITO_AMD_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(rc_)); \
  } while (0)

}  // namespace

extern "C" {

int bito_amd_gp_create(int32_t device_id, int32_t taxon_count, int32_t pattern_count, const int32_t* patterns,
                       const double* weights, int32_t node_count, int32_t gpcsp_count, double rescaling_threshold,
                       bito_amd_gp_engine** out, char* err, size_t err_len) {
  auto report = [&](int code, const std::string& msg) {
    if (err && err_len) std::snprintf(err, err_len, "%s", msg.c_str());
    return code;
  };
  if (!out) return report(BITO_AMD_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (taxon_count < 2 || pattern_count < 1 || node_count < taxon_count || gpcsp_count < 1 || !patterns || !weights)
    return report(BITO_AMD_ERR_BAD_ARG, "bad sizes or NULL arrays");
  if (!(rescaling_threshold > 0 && rescaling_threshold < 1))
    return report(BITO_AMD_ERR_BAD_ARG, "rescaling threshold must be in (0,1)");
  int devices = 0;
  if (hipGetDeviceCount(&devices) != hipSuccess || devices <= 0 || device_id < 0 || device_id >= devices)
    return report(BITO_AMD_ERR_DEVICE, "no HIP device available: the GP executor needs an MI355X (no CPU fallback)");
  auto e = new bito_amd_gp_engine();
  e->device = device_id; e->n = taxon_count; e->P = pattern_count; e->Ppad = (pattern_count + 63) / 64 * 64;
  e->nodes = node_count; e->gpcsps = gpcsp_count; e->plvs = 6 * node_count;
  e->threshold = rescaling_threshold; e->log_threshold = std::log(rescaling_threshold);
  if (const char* w = std::getenv("BITO_AMD_GP_OPT_WAVES")) {
    const int v = std::atoi(w);
    e->opt_waves = (v == 1 || v == 2 || v == 8 || v == 16) ? v : kOptWaves;
  }
  (void)hipSetDevice(device_id);
  const size_t plv_bytes = (size_t)e->plvs * 4 * e->Ppad * sizeof(double);
  bool ok = hipMalloc((void**)&e->plv, plv_bytes) == hipSuccess &&
            hipMalloc((void**)&e->counts, (size_t)e->plvs * e->Ppad * sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&e->weights, e->Ppad * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&e->bl, gpcsp_count * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&e->q, gpcsp_count * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&e->ll, (size_t)gpcsp_count * e->Ppad * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&e->marginal, e->Ppad * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&e->scratch, (size_t)(gpcsp_count + 4) * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&e->diff, gpcsp_count * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&e->coef, (size_t)2 * e->Ppad * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&e->d_side, sizeof(uint64_t)) == hipSuccess;
  if (!ok) { delete e; return report(BITO_AMD_ERR_DEVICE, "hipMalloc failed for the PLV arena"); }
  e->side_cap = 1;
  (void)hipMemset(e->plv, 0, plv_bytes);
  (void)hipMemset(e->counts, 0, (size_t)e->plvs * e->Ppad * sizeof(int));
  (void)hipMemset(e->ll, 0, (size_t)gpcsp_count * e->Ppad * sizeof(double));
  (void)hipMemset(e->marginal, 0, e->Ppad * sizeof(double));
  (void)hipMemset(e->diff, 0, gpcsp_count * sizeof(double));  // init_default_difference_ (dag_branch_handler.hpp:269)
  // InitializePLVsWithSitePatterns: leaf P-PLVs (type 0): one-hot, all ones for a gap
  std::vector<double> leaf((size_t)taxon_count * 4 * e->Ppad, 0.0);
  for (int t = 0; t < taxon_count; t++)
    for (int p = 0; p < pattern_count; p++) {
      const int s = patterns[(size_t)t * pattern_count + p];
      for (int i = 0; i < 4; i++) leaf[((size_t)t * 4 + i) * e->Ppad + p] = (s >= 4 || s == i) ? 1.0 : 0.0;
    }
  std::vector<double> w(e->Ppad, 0.0), ones(gpcsp_count, 1.0), zeros(gpcsp_count, 0.0);
  for (int p = 0; p < pattern_count; p++) w[p] = weights[p];
  ok = hipMemcpy(e->plv, leaf.data(), leaf.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(e->weights, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(e->q, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(e->bl, zeros.data(), zeros.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { delete e; return report(BITO_AMD_ERR_DEVICE, "hipMemcpy failed"); }
  *out = e;
  return BITO_AMD_OK;
}

void bito_amd_gp_destroy(bito_amd_gp_engine* e) { delete e; }
const char* bito_amd_gp_last_error(const bito_amd_gp_engine* e) { return e ? e->err.c_str() : ""; }

int bito_amd_gp_set_branch_lengths(bito_amd_gp_engine* e, const double* v) {
  if (!e || !v) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  GP_TRY(e, hipMemcpy(e->bl, v, e->gpcsps * sizeof(double), hipMemcpyHostToDevice));
  return BITO_AMD_OK;
}
int bito_amd_gp_get_branch_lengths(bito_amd_gp_engine* e, double* out) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  GP_TRY(e, hipMemcpy(out, e->bl, e->gpcsps * sizeof(double), hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}
int bito_amd_gp_set_sbn_parameters(bito_amd_gp_engine* e, const double* v) {
  if (!e || !v) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  GP_TRY(e, hipMemcpy(e->q, v, e->gpcsps * sizeof(double), hipMemcpyHostToDevice));
  return BITO_AMD_OK;
}
int bito_amd_gp_get_sbn_parameters(bito_amd_gp_engine* e, double* out) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  GP_TRY(e, hipMemcpy(out, e->q, e->gpcsps * sizeof(double), hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}

int bito_amd_gp_get_branch_length_differences(bito_amd_gp_engine* e, double* out) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  GP_TRY(e, hipMemcpy(out, e->diff, e->gpcsps * sizeof(double), hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}
int bito_amd_gp_set_optimization_method(bito_amd_gp_engine* e, int32_t method) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (method < 0 || method > 4)
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "DAGBranchHandler::Optimization(): Invalid OptimizationMethod given.");
  e->method = method;
  return BITO_AMD_OK;
}
int bito_amd_gp_set_significant_digits_for_optimization(bito_amd_gp_engine* e, int32_t digits) {
  if (!e || digits < 1) return BITO_AMD_ERR_BAD_ARG;
  e->significant_digits = digits;
  return BITO_AMD_OK;
}
int bito_amd_gp_reset_optimization_count(bito_amd_gp_engine* e) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  e->optimization_count = 0;
  return BITO_AMD_OK;
}
int bito_amd_gp_increment_optimization_count(bito_amd_gp_engine* e) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  e->optimization_count++;
  return BITO_AMD_OK;
}

int bito_amd_gp_per_gpcsp_log_likelihoods(bito_amd_gp_engine* e, double* out) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(gp_weighted_rows_kernel, dim3(e->gpcsps), dim3(256), 0, 0, e->ll, e->weights, e->P, e->Ppad,
                     e->scratch);
  GP_TRY(e, hipMemcpy(out, e->scratch, e->gpcsps * sizeof(double), hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}

int bito_amd_gp_log_marginal_likelihood(bito_amd_gp_engine* e, double* out) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(gp_weighted_rows_kernel, dim3(1), dim3(256), 0, 0, e->marginal, e->weights, e->P, e->Ppad,
                     e->scratch);
  GP_TRY(e, hipMemcpy(out, e->scratch, sizeof(double), hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}

int bito_amd_gp_log_likelihood_matrix(bito_amd_gp_engine* e, double* out) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  GP_TRY(e, hipMemcpy2D(out, e->P * sizeof(double), e->ll, e->Ppad * sizeof(double), e->P * sizeof(double), e->gpcsps,
                        hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}

// ids are validated once on the host, then the stream and its side array go to the device
static int ValidateOps(bito_amd_gp_engine* e, const bito_amd_gp_op* ops, int64_t op_count, const uint64_t* side,
                       int64_t side_count, bool batched) {
  const uint64_t plv_limit = (uint64_t)e->plvs + (uint64_t)e->spare_plvs;
  const uint64_t gp_limit = (uint64_t)e->gpcsps + (uint64_t)e->spare_gpcsps;
  for (int64_t o = 0; o < op_count; o++) {
    const bito_amd_gp_op& op = ops[o];
    auto plv_ok = [&](uint64_t id) { return id < plv_limit; };
    auto gp_ok = [&](uint64_t id) { return id < gp_limit; };
    bool ok = true;
    switch (op.opcode) {
      case BITO_AMD_GP_ZERO_PLV: ok = plv_ok(op.a); break;
      case BITO_AMD_GP_SET_TO_STATIONARY_DISTRIBUTION: ok = plv_ok(op.a) && gp_ok(op.b); break;
      case BITO_AMD_GP_INCREMENT_WITH_WEIGHTED_EVOLVED_PLV: ok = plv_ok(op.a) && gp_ok(op.b) && plv_ok(op.c); break;
      case BITO_AMD_GP_MULTIPLY: ok = plv_ok(op.a) && plv_ok(op.b) && plv_ok(op.c); break;
      case BITO_AMD_GP_LIKELIHOOD: ok = gp_ok(op.a) && plv_ok(op.b) && plv_ok(op.c); break;
      case BITO_AMD_GP_UPDATE_SBN_PROBABILITIES: ok = !batched && op.a < op.b && op.b <= (uint64_t)e->gpcsps; break;
      case BITO_AMD_GP_RESET_MARGINAL_LIKELIHOOD: ok = !batched; break;
      case BITO_AMD_GP_INCREMENT_MARGINAL_LIKELIHOOD: ok = !batched && plv_ok(op.a) && gp_ok(op.b) && plv_ok(op.c); break;
      case BITO_AMD_GP_PREP_FOR_MARGINALIZATION:
        ok = plv_ok(op.a) && op.count > 0 && side && side_count > 0 && op.b <= (uint64_t)side_count &&
             (uint64_t)op.count <= (uint64_t)side_count - op.b;  // (no sum: b near 2^64 must not wrap)
        for (uint32_t k = 0; ok && k < op.count; k++) ok = plv_ok(side[op.b + k]);
        break;
      case BITO_AMD_GP_OPTIMIZE_BRANCH_LENGTH: ok = plv_ok(op.a) && plv_ok(op.b) && gp_ok(op.c); break;
      default:
        return Fail(e, BITO_AMD_ERR_BAD_ARG, "unknown GP opcode " + std::to_string(op.opcode));
    }
    if (!ok)
      return Fail(e, BITO_AMD_ERR_BAD_ARG,
                  "GP op " + std::to_string(o) + " has an index out of range" +
                      (batched ? " or is not allowed in a batch of independent sub-streams" : ""));
  }
  return BITO_AMD_OK;
}

// the stream (or the scheduled image of it) and its side array go to the device
static int UploadOps(bito_amd_gp_engine* e, const bito_amd_gp_op* image, int64_t op_count, const uint64_t* side,
                     int64_t side_count) {
  if (side_count > 0 && side) {
    if ((size_t)side_count > e->side_cap) {
      (void)hipFree(e->d_side);
      e->side_cap = 0;
      GP_TRY(e, hipMalloc((void**)&e->d_side, side_count * sizeof(uint64_t)));
      e->side_cap = side_count;
    }
    GP_TRY(e, hipMemcpy(e->d_side, side, side_count * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
  if ((size_t)op_count > e->ops_cap) {
    if (e->d_ops) (void)hipFree(e->d_ops);
    e->ops_cap = 0;
    GP_TRY(e, hipMalloc((void**)&e->d_ops, op_count * sizeof(bito_amd_gp_op)));
    e->ops_cap = op_count;
  }
  if (op_count > 0) GP_TRY(e, hipMemcpy(e->d_ops, image, op_count * sizeof(bito_amd_gp_op), hipMemcpyHostToDevice));
  return BITO_AMD_OK;
}

static int ValidateAndUpload(bito_amd_gp_engine* e, const bito_amd_gp_op* ops, int64_t op_count, const uint64_t* side,
                             int64_t side_count, bool batched) {
  if (int rc = ValidateOps(e, ops, op_count, side, side_count, batched)) return rc;
  return UploadOps(e, ops, op_count, side, side_count);
}

// Runs of per-pattern operations shorter than this stay with the one-thread-per-pattern interpreter (gp_ops_kernel walks
// them in image order, which respects their levels); longer ones go level by level through gp_levels_kernel.
constexpr int64_t kLevelMinOps = 48;

static OptSettings Settings(const bito_amd_gp_engine* e) {
  return OptSettings{e->method, e->significant_digits, e->optimization_count != 0,  // !IsFirstOptimization()
                     e->trace_rows, e->trace_cursor, (long long)e->trace_capacity};
}

static uint64_t HashStream(const bito_amd_gp_op* ops, int64_t op_count, const uint64_t* side, int64_t side_count, bool reorder) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)op_count ^ ((uint64_t)side_count << 32) ^ (reorder ? 1 : 0);
  auto mix = [&](uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
  };
  for (int64_t o = 0; o < op_count; o++) {
    mix(((uint64_t)ops[o].opcode << 32) | ops[o].count);
    mix(ops[o].a);
    mix(ops[o].b);
    mix(ops[o].c);
  }
  for (int64_t k = 0; k < side_count; k++) mix(side[k]);
  return h;
}

// The schedule of a stream (gp_schedule.hpp), kept by content: a sweep's schedule is replayed many times
// (GPInstance::EstimateBranchLengths alternates the same three streams until convergence, src/gp_instance.cpp:241-308).
static std::shared_ptr<GpCachedSchedule> ScheduleOf(bito_amd_gp_engine* e, const bito_amd_gp_op* ops, int64_t op_count,
                                                    const uint64_t* side, int64_t side_count, bool reorder) {
  const uint64_t h = HashStream(ops, op_count, side, side_count, reorder);
  for (auto it = e->schedules.begin(); it != e->schedules.end(); ++it) {
    const GpCachedSchedule& c = **it;
    if (c.hash == h && c.reorder == reorder && (int64_t)c.ops.size() == op_count && (int64_t)c.side.size() == side_count &&
        (op_count == 0 || std::memcmp(c.ops.data(), ops, (size_t)op_count * sizeof(bito_amd_gp_op)) == 0) &&
        (side_count == 0 || std::memcmp(c.side.data(), side, (size_t)side_count * sizeof(uint64_t)) == 0)) {
      std::shared_ptr<GpCachedSchedule> hit = *it;
      e->schedules.erase(it);
      e->schedules.push_front(hit);
      return hit;
    }
  }
  auto fresh = std::make_shared<GpCachedSchedule>();
  fresh->hash = h;
  fresh->reorder = reorder;
  fresh->ops.assign(ops, ops + op_count);
  if (side_count > 0) fresh->side.assign(side, side + side_count);
  bito_amd_gp_schedule::ScheduleStream(ops, op_count, side, reorder, &fresh->schedule);
  e->schedules.push_front(fresh);
  if (e->schedules.size() > 16) e->schedules.pop_back();
  return fresh;
}

int bito_amd_gp_schedule_operations(const bito_amd_gp_op* ops, int64_t op_count, const uint64_t* side, int64_t side_count,
                                    int32_t reorder, bito_amd_gp_op* out_ops, int32_t* out_launch, int32_t* out_level,
                                    int32_t* out_launch_kinds, int64_t* out_launch_count) {
  if (op_count < 0 || (op_count > 0 && (!ops || !out_ops)) || !out_launch_count) return BITO_AMD_ERR_BAD_ARG;
  for (int64_t o = 0; o < op_count; o++) {
    // (needs no engine, so ids cannot be held to a DAG here; what the read / write sets index by must still be sane:
    // a known opcode, and a side range that lies inside the side array -- compared without a sum that could wrap)
    if (ops[o].opcode > BITO_AMD_GP_PREP_FOR_MARGINALIZATION) return BITO_AMD_ERR_BAD_ARG;
    if (ops[o].opcode == BITO_AMD_GP_PREP_FOR_MARGINALIZATION &&
        (!side || side_count <= 0 || ops[o].b > (uint64_t)side_count || (uint64_t)ops[o].count > (uint64_t)side_count - ops[o].b))
      return BITO_AMD_ERR_BAD_ARG;
  }
  bito_amd_gp_schedule::Schedule S;
  bito_amd_gp_schedule::ScheduleStream(ops, op_count, side, reorder != 0, &S);
  if ((int64_t)S.image.size() != op_count) return BITO_AMD_ERR_STATE;
  for (int64_t o = 0; o < op_count; o++) {
    out_ops[o] = S.image[o];
    if (out_launch) out_launch[o] = S.launch_of[o];
    if (out_level) out_level[o] = S.level_of[o];
  }
  if (out_launch_kinds)
    for (size_t k = 0; k < S.launches.size(); k++) out_launch_kinds[k] = S.launches[k].kind;
  *out_launch_count = (int64_t)S.launches.size();
  return BITO_AMD_OK;
}

int bito_amd_gp_process_operations(bito_amd_gp_engine* e, const bito_amd_gp_op* ops, int64_t op_count,
                                   const uint64_t* side, int64_t side_count) {
  if (!e || (op_count > 0 && !ops)) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  // The stream is executed in the order gp_schedule.hpp gives it: per-pattern operations sorted into dependency levels,
  // the optimisations of equal optimiser depth as concurrent workgroups of one launch -- the sequential arithmetic,
  // bit for bit (tests/test_gp.py holds the two orders to the same bits).  BITO_AMD_GP_SCHEDULE=0: the optimisations one
  // launch each in stream order (read per call: the tests compare both routes).
  const char* env = std::getenv("BITO_AMD_GP_SCHEDULE");
  const bool reorder = !(env != nullptr && env[0] == '0');
  // (ids are validated on the stream as given, before anything is indexed by them)
  if (int rc = ValidateOps(e, ops, op_count, side, side_count, false)) return rc;
  const int64_t side_used = side && side_count > 0 ? side_count : 0;
  const std::shared_ptr<GpCachedSchedule> cached = ScheduleOf(e, ops, op_count, side, side_used, reorder);
  const bito_amd_gp_schedule::Schedule& S = cached->schedule;
  if (!cached->on_device) {
    // (allocated into locals and handed to the cache entry only when all three stand: a failure half way leaves nothing
    // behind for the next call to overwrite)
    bito_amd_gp_op* up_ops = nullptr;
    int64_t* up_levels = nullptr;
    uint64_t* up_side = nullptr;
    auto upload = [&]() -> hipError_t {
      hipError_t st = hipSuccess;
      if (op_count > 0) {
        if ((st = hipMalloc((void**)&up_ops, (size_t)op_count * sizeof(bito_amd_gp_op))) != hipSuccess) return st;
        if ((st = hipMemcpy(up_ops, S.image.data(), (size_t)op_count * sizeof(bito_amd_gp_op), hipMemcpyHostToDevice)) != hipSuccess) return st;
      }
      if (!S.level_offsets.empty()) {
        if ((st = hipMalloc((void**)&up_levels, S.level_offsets.size() * sizeof(int64_t))) != hipSuccess) return st;
        if ((st = hipMemcpy(up_levels, S.level_offsets.data(), S.level_offsets.size() * sizeof(int64_t), hipMemcpyHostToDevice)) != hipSuccess) return st;
      }
      // (never a null pointer in a kernel's argument list: a stream without PrepForMarginalization gets one entry)
      if ((st = hipMalloc((void**)&up_side, (size_t)std::max<int64_t>(side_used, 1) * sizeof(uint64_t))) != hipSuccess) return st;
      if (side_used > 0) st = hipMemcpy(up_side, side, (size_t)side_used * sizeof(uint64_t), hipMemcpyHostToDevice);
      return st;
    };
    const hipError_t st = upload();
    if (st != hipSuccess) {
      if (up_ops) (void)hipFree(up_ops);
      if (up_levels) (void)hipFree(up_levels);
      if (up_side) (void)hipFree(up_side);
      GP_TRY(e, st);
    }
    cached->device = e->device;
    cached->d_ops = up_ops;
    cached->d_levels = up_levels;
    cached->d_side = up_side;
    cached->device_bytes = (size_t)op_count * sizeof(bito_amd_gp_op) + S.level_offsets.size() * sizeof(int64_t) +
                           (size_t)std::max<int64_t>(side_used, 1) * sizeof(uint64_t);
    cached->on_device = true;
    // the cache is bounded by what it holds on the device as well as by its sixteen entries: the oldest schedules leave
    // until the resident images fit kScheduleCacheBytes (the one in use always stays)
    size_t held = 0;
    for (const auto& c : e->schedules) held += c->device_bytes;
    while (held > kScheduleCacheBytes && e->schedules.size() > 1 && e->schedules.back() != cached) {
      held -= e->schedules.back()->device_bytes;
      e->schedules.pop_back();
    }
  }
  bito_amd_gp_op* const d_ops = cached->d_ops;
  const uint64_t* const d_side = cached->d_side;
  if ((size_t)S.max_concurrent_optimisers > e->coef_blocks) {
    (void)hipFree(e->coef);
    e->coef = nullptr;
    e->coef_blocks = 0;
    GP_TRY(e, hipMalloc((void**)&e->coef, (size_t)S.max_concurrent_optimisers * 2 * e->Ppad * sizeof(double)));
    e->coef_blocks = (size_t)S.max_concurrent_optimisers;
  }
  for (const bito_amd_gp_schedule::Launch& L : S.launches) {
    switch (L.kind) {
      case bito_amd_gp_schedule::kPatternOps:
        if (L.count >= kLevelMinOps && L.level_count > 1) {
          hipLaunchKernelGGL(gp_levels_kernel, dim3((e->P + 63) / 64), dim3(64, kLevelWaves), 0, 0, d_ops,
                             (const int64_t*)(cached->d_levels + L.level_first), L.level_count, d_side, e->plv, e->counts,
                             e->bl, e->q, e->ll, e->marginal, e->P, e->Ppad, e->threshold, e->log_threshold);
          GP_TRY(e, hipGetLastError());
        } else if (L.count > 0) {
          hipLaunchKernelGGL(gp_ops_kernel, dim3((e->P + 63) / 64), dim3(64), 0, 0, d_ops + L.first, L.count,
                             (const int64_t*)nullptr, d_side, e->plv, e->counts, e->bl, e->q, e->ll, e->marginal, e->P,
                             e->Ppad, e->threshold, e->log_threshold);
          GP_TRY(e, hipGetLastError());
        }
        break;
      case bito_amd_gp_schedule::kOptimisers:
        // DAGBranchHandler::OptimizeBranchLength for every edge of the launch, a workgroup each
#define BITO_GP_OPT(THREADS)                                                                                              \
  hipLaunchKernelGGL(gp_optimize_kernel<THREADS>, dim3((unsigned)L.count), dim3(THREADS), 0, 0, d_ops + L.first, e->plv, \
                     e->counts, e->weights, e->bl, e->diff, e->coef, e->P, e->Ppad, e->log_threshold, Settings(e))
        switch (e->opt_waves) {
          case 1: BITO_GP_OPT(64); break;
          case 2: BITO_GP_OPT(128); break;
          case 8: BITO_GP_OPT(512); break;
          case 16: BITO_GP_OPT(1024); break;
          default: BITO_GP_OPT(kOptThreads); break;
        }
#undef BITO_GP_OPT
        GP_TRY(e, hipGetLastError());
        break;
      default: {
        // UpdateSBNProbabilities (src/gp_engine.cpp:297-321): softmax over sibling GPCSPs of the
        // weighted per-GPCSP log-likelihood plus the log prior.
        const uint64_t a = S.image[L.first].a, b = S.image[L.first].b;
        const int len = (int)(b - a);
        std::vector<double> qv(e->gpcsps);
        GP_TRY(e, hipMemcpy(qv.data(), e->q, e->gpcsps * sizeof(double), hipMemcpyDeviceToHost));
        if (len == 1) {
          qv[a] = 1.0;
        } else {
          std::vector<double> per(e->gpcsps);
          if (int rc = bito_amd_gp_per_gpcsp_log_likelihoods(e, per.data())) return rc;
          double norm = -INFINITY;
          std::vector<double> lu(len);
          for (int k = 0; k < len; k++) {
            lu[k] = per[a + k] + std::log(qv[a + k]);
            const double x = std::max(norm, lu[k]), y = std::min(norm, lu[k]);
            norm = (x == -INFINITY || y - x < -36.04365338911715) ? x : x + std::log(1.0 + std::exp(y - x));
          }
          for (int k = 0; k < len; k++) qv[a + k] = std::exp(lu[k] - norm);
        }
        GP_TRY(e, hipMemcpy(e->q, qv.data(), e->gpcsps * sizeof(double), hipMemcpyHostToDevice));
        break;
      }
    }
  }
  GP_TRY(e, hipDeviceSynchronize());
  return BITO_AMD_OK;
}

int bito_amd_gp_set_optimizer_trace(bito_amd_gp_engine* e, int64_t capacity_rows) {
  if (!e || capacity_rows < 0) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  GP_TRY(e, hipDeviceSynchronize());
  if (e->trace_rows) (void)hipFree(e->trace_rows);
  e->trace_rows = nullptr;
  e->trace_capacity = 0;
  if (capacity_rows == 0) return BITO_AMD_OK;
  if (!e->trace_cursor) GP_TRY(e, hipMalloc((void**)&e->trace_cursor, sizeof(unsigned long long)));
  GP_TRY(e, hipMemset(e->trace_cursor, 0, sizeof(unsigned long long)));
  GP_TRY(e, hipMalloc((void**)&e->trace_rows, (size_t)capacity_rows * 4 * sizeof(double)));
  GP_TRY(e, hipMemset(e->trace_rows, 0, (size_t)capacity_rows * 4 * sizeof(double)));
  e->trace_capacity = capacity_rows;
  return BITO_AMD_OK;
}

int bito_amd_gp_get_optimizer_trace(bito_amd_gp_engine* e, double* rows, int64_t capacity_rows, int64_t* row_count) {
  if (!e || !row_count || capacity_rows < 0 || (capacity_rows > 0 && !rows)) return BITO_AMD_ERR_BAD_ARG;
  if (!e->trace_rows) return Fail(e, BITO_AMD_ERR_STATE, "no optimiser trace is being recorded");
  GP_TRY(e, hipSetDevice(e->device));
  unsigned long long made = 0;
  GP_TRY(e, hipMemcpy(&made, e->trace_cursor, sizeof(made), hipMemcpyDeviceToHost));
  *row_count = (int64_t)made;
  const int64_t have = std::min<int64_t>(std::min<int64_t>((int64_t)made, e->trace_capacity), capacity_rows);
  if (have > 0) GP_TRY(e, hipMemcpy(rows, e->trace_rows, (size_t)have * 4 * sizeof(double), hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}

int bito_amd_gp_process_operation_batches(bito_amd_gp_engine* e, const bito_amd_gp_op* ops, int64_t op_count,
                                          const uint64_t* side, int64_t side_count, const int64_t* offsets,
                                          int64_t batch_count) {
  if (!e || (op_count > 0 && !ops) || !offsets || batch_count < 0) return BITO_AMD_ERR_BAD_ARG;
  if (batch_count == 0) return BITO_AMD_OK;
  if (offsets[0] != 0 || offsets[batch_count] != op_count)
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "offsets must run from 0 to op_count");
  for (int64_t b = 0; b < batch_count; b++)
    if (offsets[b + 1] < offsets[b]) return Fail(e, BITO_AMD_ERR_BAD_ARG, "offsets must be non-decreasing");
  GP_TRY(e, hipSetDevice(e->device));
  if (int rc = ValidateAndUpload(e, ops, op_count, side, side_count, true)) return rc;
  if ((size_t)batch_count + 1 > e->offsets_cap) {
    if (e->d_offsets) (void)hipFree(e->d_offsets);
    e->offsets_cap = 0;
    GP_TRY(e, hipMalloc((void**)&e->d_offsets, (batch_count + 1) * sizeof(int64_t)));
    e->offsets_cap = batch_count + 1;
  }
  GP_TRY(e, hipMemcpy(e->d_offsets, offsets, (batch_count + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  bool has_optimiser = false;
  for (int64_t o = 0; o < op_count && !has_optimiser; o++) has_optimiser = ops[o].opcode == BITO_AMD_GP_OPTIMIZE_BRANCH_LENGTH;
  if (!has_optimiser) {
    const unsigned rows = (unsigned)(batch_count < 65535 ? batch_count : 65535);  // grid.y limit
    hipLaunchKernelGGL(gp_ops_kernel, dim3((e->P + 63) / 64, rows), dim3(64), 0, 0, e->d_ops,
                       (int64_t)batch_count, (const int64_t*)e->d_offsets, e->d_side, e->plv, e->counts, e->bl, e->q, e->ll,
                       e->marginal, e->P, e->Ppad, e->threshold, e->log_threshold);
  } else {
    // optimiser ops reduce over all patterns: one workgroup per sub-stream interprets everything
    if ((size_t)batch_count > e->coef_blocks) {
      (void)hipFree(e->coef);
      e->coef = nullptr;
      e->coef_blocks = 0;
      GP_TRY(e, hipMalloc((void**)&e->coef, (size_t)batch_count * 2 * e->Ppad * sizeof(double)));
      e->coef_blocks = batch_count;
    }
    const OptSettings cfg = Settings(e);
    hipLaunchKernelGGL(gp_block_stream_kernel, dim3((unsigned)batch_count), dim3(kOptThreads), 0, 0, e->d_ops,
                       (const int64_t*)e->d_offsets, e->d_side, e->plv, e->counts, e->weights, e->bl, e->q, e->ll,
                       e->marginal, e->diff, e->coef, e->P, e->Ppad, e->threshold, e->log_threshold, cfg);
  }
  GP_TRY(e, hipGetLastError());
  GP_TRY(e, hipDeviceSynchronize());
  return BITO_AMD_OK;
}

int bito_amd_gp_grow_spare(bito_amd_gp_engine* e, int64_t spare_plv_count, int64_t spare_gpcsp_count) {
  if (!e || spare_plv_count < 0 || spare_gpcsp_count < 0) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  auto grow = [&](auto*& ptr, size_t old_count, size_t new_count) -> hipError_t {
    using T = std::remove_pointer_t<std::remove_reference_t<decltype(ptr)>>;
    T* fresh = nullptr;
    hipError_t rc = hipMalloc((void**)&fresh, new_count * sizeof(T));
    if (rc != hipSuccess) return rc;
    if ((rc = hipMemset(fresh, 0, new_count * sizeof(T))) != hipSuccess) return rc;
    if ((rc = hipMemcpy(fresh, ptr, old_count * sizeof(T), hipMemcpyDeviceToDevice)) != hipSuccess) return rc;
    (void)hipFree(ptr);
    ptr = fresh;
    return hipSuccess;
  };
  if (spare_plv_count > e->spare_plvs) {
    const size_t have = (size_t)e->plvs + e->spare_plvs, want = (size_t)e->plvs + spare_plv_count;
    GP_TRY(e, grow(e->plv, have * 4 * e->Ppad, want * 4 * e->Ppad));
    GP_TRY(e, grow(e->counts, have * e->Ppad, want * e->Ppad));
    e->spare_plvs = spare_plv_count;
  }
  if (spare_gpcsp_count > e->spare_gpcsps) {
    const size_t have = (size_t)e->gpcsps + e->spare_gpcsps, want = (size_t)e->gpcsps + spare_gpcsp_count;
    GP_TRY(e, grow(e->bl, have, want));
    GP_TRY(e, grow(e->q, have, want));
    GP_TRY(e, grow(e->diff, have, want));
    GP_TRY(e, grow(e->ll, have * e->Ppad, want * e->Ppad));
    GP_TRY(e, grow(e->scratch, have + 4, want + 4));
    e->spare_gpcsps = spare_gpcsp_count;
  }
  return BITO_AMD_OK;
}

int bito_amd_gp_grow(bito_amd_gp_engine* e, int32_t new_node_count, int32_t new_gpcsp_count,
                     const int64_t* node_reindexer, const int64_t* gpcsp_reindexer) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (new_node_count < e->nodes || new_gpcsp_count < e->gpcsps)
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "GrowPLVs / GrowGPCSPs: the engine only grows");
  // a reindexer is a permutation of [0, new count): old index -> new index (Reindexer::GetNewIndexByOldIndex)
  auto check = [&](const int64_t* map, int64_t count) {
    if (!map) return true;
    std::vector<char> seen(count, 0);
    for (int64_t i = 0; i < count; i++) {
      if (map[i] < 0 || map[i] >= count || seen[map[i]]) return false;
      seen[map[i]] = 1;
    }
    return true;
  };
  if (!check(node_reindexer, new_node_count)) return Fail(e, BITO_AMD_ERR_BAD_ARG, "Node Reindexer is not valid.");
  if (!check(gpcsp_reindexer, new_gpcsp_count))
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "GPCSP Reindexer is not valid for GPEngine size.");
  GP_TRY(e, hipSetDevice(e->device));
  GP_TRY(e, hipDeviceSynchronize());
  const int old_nodes = e->nodes, old_gpcsps = e->gpcsps;
  auto device_copy = [&](const std::vector<int64_t>& host, int64_t** out) -> hipError_t {
    hipError_t rc = hipMalloc((void**)out, host.size() * sizeof(int64_t));
    if (rc != hipSuccess) return rc;
    return hipMemcpy(*out, host.data(), host.size() * sizeof(int64_t), hipMemcpyHostToDevice);
  };
  if (new_node_count != old_nodes || node_reindexer) {
    // PLV (type, node) moves to (type, reindexer[node]); new nodes and the spare slots start zeroed
    const size_t new_plvs = (size_t)6 * new_node_count, total = new_plvs + e->spare_plvs;
    std::vector<int64_t> map((size_t)6 * old_nodes);
    for (int t = 0; t < 6; t++)
      for (int v = 0; v < old_nodes; v++)
        map[(size_t)t * old_nodes + v] = (int64_t)t * new_node_count + (node_reindexer ? node_reindexer[v] : v);
    int64_t* d_map = nullptr;
    double* plv = nullptr;
    int* counts = nullptr;
    GP_TRY(e, device_copy(map, &d_map));
    GP_TRY(e, hipMalloc((void**)&plv, total * 4 * e->Ppad * sizeof(double)));
    GP_TRY(e, hipMalloc((void**)&counts, total * e->Ppad * sizeof(int)));
    GP_TRY(e, hipMemset(plv, 0, total * 4 * e->Ppad * sizeof(double)));
    GP_TRY(e, hipMemset(counts, 0, total * e->Ppad * sizeof(int)));
    GP_TRY(e, PermuteRows(e->plv, plv, d_map, map.size(), (size_t)4 * e->Ppad, 4));
    GP_TRY(e, PermuteRows(e->counts, counts, d_map, map.size(), (size_t)e->Ppad, 1));
    GP_TRY(e, hipDeviceSynchronize());
    (void)hipFree(d_map);
    (void)hipFree(e->plv);
    (void)hipFree(e->counts);
    e->plv = plv;
    e->counts = counts;
    e->nodes = new_node_count;
    e->plvs = (int)new_plvs;
  }
  if (new_gpcsp_count != old_gpcsps || gpcsp_reindexer) {
    const size_t total = (size_t)new_gpcsp_count + e->spare_gpcsps;
    // per-GPCSP scalars on the host: new edges start with the default branch length, q = 1 and no difference
    std::vector<double> bl(old_gpcsps), q(old_gpcsps), diff(old_gpcsps);
    GP_TRY(e, hipMemcpy(bl.data(), e->bl, old_gpcsps * sizeof(double), hipMemcpyDeviceToHost));
    GP_TRY(e, hipMemcpy(q.data(), e->q, old_gpcsps * sizeof(double), hipMemcpyDeviceToHost));
    GP_TRY(e, hipMemcpy(diff.data(), e->diff, old_gpcsps * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> nbl(total, 0.0), nq(total, 0.0), ndiff(total, 0.0);
    std::vector<int64_t> map(old_gpcsps);
    for (int64_t i = 0; i < new_gpcsp_count; i++) {
      const int64_t to = gpcsp_reindexer ? gpcsp_reindexer[i] : i;
      if (i < old_gpcsps) {
        nbl[to] = bl[i]; nq[to] = q[i]; ndiff[to] = diff[i];
        map[i] = to;
      } else {
        nbl[to] = 0.1;  // DAGBranchHandler::init_default_branch_length_
        nq[to] = 1.0;   // GPEngine::GrowGPCSPs: q_[i] = 1
      }
    }
    double *dbl = nullptr, *dq = nullptr, *ddiff = nullptr, *ll = nullptr, *scratch = nullptr;
    int64_t* d_map = nullptr;
    GP_TRY(e, device_copy(map, &d_map));
    GP_TRY(e, hipMalloc((void**)&dbl, total * sizeof(double)));
    GP_TRY(e, hipMalloc((void**)&dq, total * sizeof(double)));
    GP_TRY(e, hipMalloc((void**)&ddiff, total * sizeof(double)));
    GP_TRY(e, hipMalloc((void**)&ll, total * e->Ppad * sizeof(double)));
    GP_TRY(e, hipMalloc((void**)&scratch, (total + 4) * sizeof(double)));
    GP_TRY(e, hipMemcpy(dbl, nbl.data(), total * sizeof(double), hipMemcpyHostToDevice));
    GP_TRY(e, hipMemcpy(dq, nq.data(), total * sizeof(double), hipMemcpyHostToDevice));
    GP_TRY(e, hipMemcpy(ddiff, ndiff.data(), total * sizeof(double), hipMemcpyHostToDevice));
    GP_TRY(e, hipMemset(ll, 0, total * e->Ppad * sizeof(double)));
    GP_TRY(e, PermuteRows(e->ll, ll, d_map, (size_t)old_gpcsps, (size_t)e->Ppad, 1));
    GP_TRY(e, hipDeviceSynchronize());
    (void)hipFree(d_map);
    for (double* p : {e->bl, e->q, e->diff, e->ll, e->scratch}) (void)hipFree(p);
    e->bl = dbl; e->q = dq; e->diff = ddiff; e->ll = ll; e->scratch = scratch;
    e->gpcsps = new_gpcsp_count;
  }
  return BITO_AMD_OK;
}

int bito_amd_gp_get_plv(bito_amd_gp_engine* e, int64_t plv, double* out) {
  if (!e || !out || plv < 0 || plv >= (int64_t)e->plvs + e->spare_plvs) return BITO_AMD_ERR_BAD_ARG;
  GP_TRY(e, hipSetDevice(e->device));
  GP_TRY(e, hipMemcpy2D(out, e->P * sizeof(double), e->plv + (size_t)plv * 4 * e->Ppad, e->Ppad * sizeof(double),
                        e->P * sizeof(double), 4, hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}

// The reference keeps ONE rescaling count per PLV, decided from the whole-PLV maximum (RescalePLVIfNeeded,
// src/gp_engine.cpp:583-597); the executor keeps one per (PLV, pattern).  Both are the number of divisions by the
// threshold that brought a value into [threshold, 1): a count never exceeds what the magnitude of its values asks
// for, so the whole-PLV count -- decided by the LARGEST entry -- is the smallest of the per-pattern counts (over the
// patterns that are not identically zero: a zero does not take part in a maximum), and the reference's stored value
// of pattern p is the executor's times threshold^(count_p - count).  These two calls hand out that view.
static int ReferenceView(bito_amd_gp_engine* e, int64_t plv, std::vector<double>* values, std::vector<int>* cnt, int* count) {
  GP_TRY(e, hipSetDevice(e->device));
  values->assign((size_t)4 * e->P, 0.0);
  cnt->assign((size_t)e->P, 0);
  GP_TRY(e, hipMemcpy2D(values->data(), e->P * sizeof(double), e->plv + (size_t)plv * 4 * e->Ppad, e->Ppad * sizeof(double),
                        e->P * sizeof(double), 4, hipMemcpyDeviceToHost));
  GP_TRY(e, hipMemcpy(cnt->data(), e->counts + (size_t)plv * e->Ppad, (size_t)e->P * sizeof(int), hipMemcpyDeviceToHost));
  int lowest = std::numeric_limits<int>::max(), lowest_any = std::numeric_limits<int>::max();
  for (int p = 0; p < e->P; p++) {
    double mx = 0;
    for (int i = 0; i < 4; i++) mx = std::max(mx, (*values)[(size_t)i * e->P + p]);
    lowest_any = std::min(lowest_any, (*cnt)[p]);
    if (mx > 0) lowest = std::min(lowest, (*cnt)[p]);
  }
  *count = lowest == std::numeric_limits<int>::max() ? lowest_any : lowest;
  return BITO_AMD_OK;
}

int bito_amd_gp_rescaling_counts(bito_amd_gp_engine* e, int64_t first, int64_t count, int32_t* out) {
  if (!e || !out || first < 0 || count < 0 || first + count > (int64_t)e->plvs + e->spare_plvs) return BITO_AMD_ERR_BAD_ARG;
  std::vector<double> values;
  std::vector<int> cnt;
  for (int64_t k = 0; k < count; k++) {
    int c = 0;
    if (int rc = ReferenceView(e, first + k, &values, &cnt, &c)) return rc;
    out[k] = c;
  }
  return BITO_AMD_OK;
}

int bito_amd_gp_get_plv_as_reference(bito_amd_gp_engine* e, int64_t plv, double* out, int32_t* out_count) {
  if (!e || !out || plv < 0 || plv >= (int64_t)e->plvs + e->spare_plvs) return BITO_AMD_ERR_BAD_ARG;
  std::vector<double> values;
  std::vector<int> cnt;
  int c = 0;
  if (int rc = ReferenceView(e, plv, &values, &cnt, &c)) return rc;
  for (int p = 0; p < e->P; p++) {
    const double f = cnt[p] == c ? 1.0 : std::pow(e->threshold, (double)(cnt[p] - c));
    for (int i = 0; i < 4; i++) out[(size_t)i * e->P + p] = values[(size_t)i * e->P + p] * f;
  }
  if (out_count) *out_count = c;
  return BITO_AMD_OK;
}

int bito_amd_gp_copy_gpcsp_data(bito_amd_gp_engine* e, const int64_t* src, const int64_t* dst, int64_t count) {
  if (!e || count < 0 || (count > 0 && (!src || !dst))) return BITO_AMD_ERR_BAD_ARG;
  if (count == 0) return BITO_AMD_OK;
  const int64_t limit = (int64_t)e->gpcsps + e->spare_gpcsps;
  for (int64_t i = 0; i < count; i++)
    if (src[i] < 0 || src[i] >= limit || dst[i] < 0 || dst[i] >= limit)
      return Fail(e, BITO_AMD_ERR_BAD_ARG, "Cannot copy GPCSP data with src or dest index out-of-range.");
  GP_TRY(e, hipSetDevice(e->device));
  std::vector<double> bl(limit), q(limit);
  GP_TRY(e, hipMemcpy(bl.data(), e->bl, limit * sizeof(double), hipMemcpyDeviceToHost));
  GP_TRY(e, hipMemcpy(q.data(), e->q, limit * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < count; i++) {  // in order, like repeated CopyGPCSPData calls
    bl[dst[i]] = bl[src[i]];
    q[dst[i]] = q[src[i]];
  }
  GP_TRY(e, hipMemcpy(e->bl, bl.data(), limit * sizeof(double), hipMemcpyHostToDevice));
  GP_TRY(e, hipMemcpy(e->q, q.data(), limit * sizeof(double), hipMemcpyHostToDevice));
  return BITO_AMD_OK;
}

int bito_amd_gp_per_gpcsp_log_likelihoods_range(bito_amd_gp_engine* e, int64_t first, int64_t count, double* out) {
  if (!e || !out || first < 0 || count < 0 || first + count > (int64_t)e->gpcsps + e->spare_gpcsps)
    return BITO_AMD_ERR_BAD_ARG;
  if (count == 0) return BITO_AMD_OK;
  GP_TRY(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(gp_weighted_rows_kernel, dim3((unsigned)count), dim3(256), 0, 0, e->ll + (size_t)first * e->Ppad,
                     e->weights, e->P, e->Ppad, e->scratch);
  GP_TRY(e, hipMemcpy(out, e->scratch, count * sizeof(double), hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}

int bito_amd_gp_branch_lengths_range(bito_amd_gp_engine* e, int64_t first, int64_t count, double* out) {
  if (!e || !out || first < 0 || count < 0 || first + count > (int64_t)e->gpcsps + e->spare_gpcsps)
    return BITO_AMD_ERR_BAD_ARG;
  if (count == 0) return BITO_AMD_OK;
  GP_TRY(e, hipSetDevice(e->device));
  GP_TRY(e, hipMemcpy(out, e->bl + first, count * sizeof(double), hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}

int bito_amd_gp_log_likelihood_and_first_two_derivatives(bito_amd_gp_engine* e, int64_t gpcsp, int64_t rootward,
                                                         int64_t leafward, double* out) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  if (gpcsp < 0 || gpcsp >= e->gpcsps + e->spare_gpcsps || rootward < 0 || rootward >= e->plvs + e->spare_plvs ||
      leafward < 0 || leafward >= e->plvs + e->spare_plvs)
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "index out of range");
  GP_TRY(e, hipSetDevice(e->device));
  double t = 0;
  GP_TRY(e, hipMemcpy(&t, e->bl + gpcsp, sizeof(double), hipMemcpyDeviceToHost));
  hipLaunchKernelGGL(gp_derivatives_kernel, dim3(1), dim3(256), 0, 0, e->plv, e->counts, e->weights, t,
                     (uint64_t)rootward, (uint64_t)leafward, e->P, e->Ppad, e->log_threshold, e->scratch);
  GP_TRY(e, hipMemcpy(out, e->scratch, 3 * sizeof(double), hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}

}  // extern "C"
