// time_tree.hip -- the time parameterisation of rooted trees and the gradient post-transforms
// FatBeagle::Gradient(RootedTree) applies to the branch gradient (SURVEY.md 8f, row f2;
// reference src/rooted_tree.cpp:36-121, src/rooted_gradient_transforms.cpp:19-256).
//
// These are O(n) recursions per tree with parent->child dependencies, so the parallel axis is
// the tree: one thread per tree, the whole batch in one launch, operating on buffers that are
// already resident (the branch gradient never leaves the device between the walk kernel and the
// ratio transform).  Node ids are post-order (children < parent, root = 2n-2), so "post-order"
// is an ascending loop and "pre-order" a descending one; child contributions are pushed to the
// parent, which needs no child lists.
//
// Vectors per tree: node vectors [2n-1], internal-node vectors [n-1] (entry id-n, root last).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace bito_amd {

namespace {

constexpr int kTimeTreeBlock = 64;

__device__ inline int TreeIndex() { return blockIdx.x * kTimeTreeBlock + threadIdx.x; }

// RootedTree::SetNodeBoundsUsingDates (rooted_tree.cpp:46-60)
__device__ void NodeBounds(int n, const int32_t* par, const double* tip_dates, double* bounds) {
  const int N = 2 * n - 1;
  for (int i = 0; i < n; i++) bounds[i] = tip_dates[i];
  for (int i = n; i < N; i++) bounds[i] = -INFINITY;
  for (int i = 0; i < N - 1; i++) bounds[par[i]] = fmax(bounds[par[i]], bounds[i]);
}

// InitializeTimeTreeUsingBranchLengths (rooted_tree.cpp:62-99); max_diff[t] = largest
// disagreement between the two children's heights, checked by the host against 1e-4.
__global__ __launch_bounds__(kTimeTreeBlock) void time_tree_from_branch_lengths_kernel(
    int T, int n, const int32_t* __restrict__ parent_ids, const double* __restrict__ branch_lengths,
    const double* __restrict__ tip_dates, double* __restrict__ bounds_all, double* __restrict__ heights_all,
    double* __restrict__ ratios_all, double* __restrict__ max_diff) {
  const int t = TreeIndex();
  if (t >= T) return;
  const int N = 2 * n - 1;
  const int32_t* par = parent_ids + (size_t)t * (N - 1);
  const double* bl = branch_lengths + (size_t)t * N;
  double* bounds = bounds_all + (size_t)t * N;
  double* h = heights_all + (size_t)t * N;
  double* ratios = ratios_all + (size_t)t * (n - 1);
  NodeBounds(n, par, tip_dates, bounds);
  for (int i = 0; i < n; i++) h[i] = tip_dates[i];
  for (int i = n; i < N; i++) h[i] = -INFINITY;
  double worst = 0;
  for (int i = 0; i < N - 1; i++) {  // the lower-id child sets the height, the other is checked
    const int p = par[i];
    const double via = h[i] + bl[i];
    if (h[p] == -INFINITY) h[p] = via;
    else worst = fmax(worst, fabs(via - h[p]));
  }
  ratios[N - 1 - n] = h[N - 1];
  for (int v = N - 2; v >= n; v--) ratios[v - n] = (h[v] - bounds[v]) / (h[par[v]] - bounds[v]);
  max_diff[t] = worst;
}

// InitializeTimeTreeUsingHeightRatios (rooted_tree.cpp:101-121)
__global__ __launch_bounds__(kTimeTreeBlock) void time_tree_from_ratios_kernel(
    int T, int n, const int32_t* __restrict__ parent_ids, const double* __restrict__ bounds_all,
    const double* __restrict__ ratios_all, double* __restrict__ heights_all, double* __restrict__ branch_all) {
  const int t = TreeIndex();
  if (t >= T) return;
  const int N = 2 * n - 1;
  const int32_t* par = parent_ids + (size_t)t * (N - 1);
  const double* bounds = bounds_all + (size_t)t * N;
  const double* ratios = ratios_all + (size_t)t * (n - 1);
  double* h = heights_all + (size_t)t * N;
  double* bl = branch_all + (size_t)t * N;
  h[N - 1] = ratios[N - 1 - n];
  bl[N - 1] = 0.0;
  for (int v = N - 2; v >= 0; v--) {
    const double hp = h[par[v]];
    if (v >= n) h[v] = bounds[v] + ratios[v - n] * (hp - bounds[v]);
    else h[v] = bounds[v];  // a leaf's height is its date
    bl[v] = hp - h[v];
  }
}

// LogDetJacobianHeightTransform (rooted_gradient_transforms.cpp:243-256); when `add_to` is
// given the value is also added to add_to[t] (the include_log_det_jacobian_likelihood flag,
// fat_beagle.cpp:93-96).
__global__ __launch_bounds__(kTimeTreeBlock) void log_det_jacobian_kernel(
    int T, int n, const int32_t* __restrict__ parent_ids, const double* __restrict__ heights_all,
    const double* __restrict__ bounds_all, double* __restrict__ out, double* __restrict__ add_to) {
  const int t = TreeIndex();
  if (t >= T) return;
  const int N = 2 * n - 1;
  const int32_t* par = parent_ids + (size_t)t * (N - 1);
  const double* bounds = bounds_all + (size_t)t * N;
  const double* h = heights_all + (size_t)t * N;
  double s = 0;
  for (int v = N - 2; v >= n; v--) s += log(h[par[v]] - bounds[v]);
  if (out) out[t] = s;
  if (add_to) add_to[t] += s;
}

// UpdateGradientUnWeightedLogDensity + UpdateHeightParameterGradientUnweightedLogDensity
// (rooted_gradient_transforms.cpp:43-146): `hg` (height gradient, [n-1]) -> `out` ([n-1]);
// `mult` is scratch [n-1].
__device__ void RatioGradientOfHeightGradient(int n, const int32_t* par, const double* h, const double* bounds,
                                              const double* ratios, const double* hg, double* mult,
                                              double* out) {
  const int N = 2 * n - 1;
  for (int v = n; v < N - 1; v++) out[v - n] = (h[v] - bounds[v]) / ratios[v - n] * hg[v - n];
  out[N - 1 - n] = 0;
  for (int c = n; c < N - 1; c++) {  // ascending: out[c] is final when it is pushed to its parent
    const int p = par[c];
    if (p == N - 1) continue;
    double add;
    if (bounds[p] == bounds[c]) add = out[c - n] * ratios[c - n] / ratios[p - n];  // same epoch
    else add = out[c - n] * ratios[c - n] / (h[p] - bounds[c]) * ((h[p] - bounds[p]) / ratios[p - n]);
    out[p - n] += add;
  }
  mult[N - 1 - n] = 1.0;
  for (int v = N - 2; v >= n; v--) mult[v - n] = ratios[v - n] * mult[par[v] - n];
  double s = 0;
  for (int i = 0; i < n - 1; i++) s += hg[i] * mult[i];
  out[N - 1 - n] = s;
}

// mode 0: RatioGradientOfHeightGradient of a given height gradient (`in` = [T][n-1]).
// mode 1: GradientLogDeterminantJacobian (rooted_gradient_transforms.cpp:148-168).
// mode 2: RatioGradientOfBranchGradient (:186-241): `in` = branch gradient [T][in_stride],
//         rates [T][2n-2] or NULL (= 1); bit 2 of `mode` (value 4) adds the log-det-Jacobian
//         gradient (include_log_det_jacobian_gradient).
// work: [T][3(n-1)] scratch.
__global__ __launch_bounds__(kTimeTreeBlock) void ratio_gradient_kernel(
    int T, int n, int mode, const int32_t* __restrict__ parent_ids, const double* __restrict__ heights_all,
    const double* __restrict__ bounds_all, const double* __restrict__ ratios_all, const double* __restrict__ in,
    int in_stride, const double* __restrict__ rates_all, double* __restrict__ work_all, double* __restrict__ out_all) {
  const int t = TreeIndex();
  if (t >= T) return;
  const int N = 2 * n - 1, I = n - 1;
  const int32_t* par = parent_ids + (size_t)t * (N - 1);
  const double* bounds = bounds_all + (size_t)t * N;
  const double* h = heights_all + (size_t)t * N;
  const double* ratios = ratios_all + (size_t)t * I;
  double* hg = work_all + (size_t)t * 3 * I;
  double* mult = hg + I;
  double* tmp = mult + I;
  double* out = out_all + (size_t)t * I;
  const int what = mode & 3;
  if (what == 0) {
    RatioGradientOfHeightGradient(n, par, h, bounds, ratios, in + (size_t)t * in_stride, mult, out);
    return;
  }
  if (what == 2) {
    // HeightGradient (:19-41)
    const double* bg = in + (size_t)t * in_stride;
    const double* rates = rates_all ? rates_all + (size_t)t * (N - 1) : nullptr;
    for (int i = 0; i < I; i++) hg[i] = 0;
    for (int v = n; v < N - 1; v++) hg[v - n] = -bg[v] * (rates ? rates[v] : 1.0);
    for (int i = 0; i < N - 1; i++) hg[par[i] - n] += bg[i] * (rates ? rates[i] : 1.0);
    RatioGradientOfHeightGradient(n, par, h, bounds, ratios, hg, mult, out);
    if (!(mode & 4)) return;
  }
  // log-det-Jacobian gradient: GetLogTimeArray then the same transform, minus 1/ratio
  for (int i = 0; i < I - 1; i++) hg[i] = 1.0 / (h[n + i] - bounds[n + i]);
  hg[I - 1] = 0;
  double* dst = what == 1 ? out : tmp;
  RatioGradientOfHeightGradient(n, par, h, bounds, ratios, hg, mult, dst);
  for (int i = 0; i < I - 1; i++) dst[i] -= 1.0 / ratios[i];
  if (what == 2)
    for (int i = 0; i < I; i++) out[i] += tmp[i];
}

// ClockGradient (fat_beagle.cpp:379-399): per-branch gradient times the tree's own (time)
// branch length; summed for a strict clock (rate_count 1), per branch otherwise.
__global__ __launch_bounds__(kTimeTreeBlock) void clock_gradient_kernel(
    int T, int N, int rate_count, const double* __restrict__ branch_grad, const double* __restrict__ branch_lengths,
    int bl_stride, double* __restrict__ out) {
  const int t = TreeIndex();
  if (t >= T) return;
  const double* g = branch_grad + (size_t)t * N;
  const double* bl = branch_lengths + (size_t)t * bl_stride;
  if (rate_count == 1) {
    double s = 0;
    for (int i = 0; i < N - 1; i++) s += g[i] * bl[i];
    out[t] = s;
  } else {
    for (int i = 0; i < N - 1; i++) out[(size_t)t * (N - 1) + i] = g[i] * bl[i];
  }
}

inline dim3 TimeTreeGrid(int T) { return dim3((T + kTimeTreeBlock - 1) / kTimeTreeBlock); }

}  // namespace

void LaunchTimeTreeFromBranchLengths(int T, int n, const int32_t* parent_ids, const double* branch_lengths,
                                     const double* tip_dates, double* bounds, double* heights, double* ratios,
                                     double* max_diff, hipStream_t stream) {
  hipLaunchKernelGGL(time_tree_from_branch_lengths_kernel, TimeTreeGrid(T), dim3(kTimeTreeBlock), 0, stream, T, n,
                     parent_ids, branch_lengths, tip_dates, bounds, heights, ratios, max_diff);
}

void LaunchTimeTreeFromRatios(int T, int n, const int32_t* parent_ids, const double* bounds, const double* ratios,
                              double* heights, double* branch_lengths, hipStream_t stream) {
  hipLaunchKernelGGL(time_tree_from_ratios_kernel, TimeTreeGrid(T), dim3(kTimeTreeBlock), 0, stream, T, n,
                     parent_ids, bounds, ratios, heights, branch_lengths);
}

void LaunchLogDetJacobian(int T, int n, const int32_t* parent_ids, const double* heights, const double* bounds,
                          double* out, double* add_to, hipStream_t stream) {
  hipLaunchKernelGGL(log_det_jacobian_kernel, TimeTreeGrid(T), dim3(kTimeTreeBlock), 0, stream, T, n, parent_ids,
                     heights, bounds, out, add_to);
}

void LaunchRatioGradient(int T, int n, int mode, const int32_t* parent_ids, const double* heights,
                         const double* bounds, const double* ratios, const double* in, int in_stride,
                         const double* rates, double* work, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(ratio_gradient_kernel, TimeTreeGrid(T), dim3(kTimeTreeBlock), 0, stream, T, n, mode,
                     parent_ids, heights, bounds, ratios, in, in_stride, rates, work, out);
}

void LaunchClockGradient(int T, int N, int rate_count, const double* branch_grad, const double* branch_lengths,
                         int bl_stride, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(clock_gradient_kernel, TimeTreeGrid(T), dim3(kTimeTreeBlock), 0, stream, T, N, rate_count,
                     branch_grad, branch_lengths, bl_stride, out);
}

}  // namespace bito_amd
