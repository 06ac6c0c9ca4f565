/*
 * time_tree_oracle.c -- CPU ORACLE for the time-tree parameterisation of bito's RootedTree and
 * the gradient post-transforms FatBeagle::Gradient(RootedTree) applies (SURVEY.md section 8f,
 * row f2).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see bito_oracle.h for who may load it).
 * Plain-C restatement of reference src/rooted_tree.cpp:36-121 and
 * src/rooted_gradient_transforms.cpp:19-256; every function cites the lines it follows.
 *
 * Parity status: PINNED by tests/test_oracle_golden.py against RootedTree::Example's exact node
 * heights / bounds / ratios / branch lengths (src/rooted_tree.hpp:133-168) and against fluA's
 * log-det-Jacobian -9.25135166 and 68 ratio / root-height gradients
 * (src/rooted_sbn_instance.hpp:277-307).
 *
 * A tree is a parent-id vector over N = 2n-1 nodes (leaves 0..n-1, internal ids in post-order,
 * root N-1); vectors indexed "by internal node" have n-1 entries, entry id-n.
 */
#include <math.h>
#include <stdlib.h>

#include "bito_oracle.h"

static void children_of(int n, const int *parent_ids, int *child) {
  const int N = 2 * n - 1;
  for (int i = 0; i < 2 * (n - 1); i++) child[i] = -1;
  for (int i = 0; i < N - 1; i++) {
    int *slot = child + 2 * (parent_ids[i] - n);
    slot[slot[0] < 0 ? 0 : 1] = i;
  }
}

/* RootedTree::SetNodeBoundsUsingDates (rooted_tree.cpp:46-60) */
void oracle_time_tree_bounds(int n, const int *parent_ids, const double *tip_dates, double *bounds) {
  const int N = 2 * n - 1;
  for (int i = 0; i < N; i++) bounds[i] = i < n ? tip_dates[i] : -INFINITY;
  for (int i = 0; i < N - 1; i++)
    if (bounds[i] > bounds[parent_ids[i]]) bounds[parent_ids[i]] = bounds[i];
}

/* RootedTree::SetTipDates + InitializeTimeTreeUsingBranchLengths (rooted_tree.cpp:36-99);
 * ORACLE_ERR_BAD_TREE when the branch lengths are not those of a time tree (tolerance 1e-4). */
int oracle_time_tree_from_branch_lengths(int n, const int *parent_ids, const double *branch_lengths,
                                         const double *tip_dates, double *bounds, double *heights,
                                         double *ratios) {
  const int N = 2 * n - 1;
  int *child = (int *)malloc(sizeof(int) * 2 * (n - 1));
  children_of(n, parent_ids, child);
  oracle_time_tree_bounds(n, parent_ids, tip_dates, bounds);
  for (int i = 0; i < n; i++) heights[i] = tip_dates[i];
  int rc = ORACLE_OK;
  for (int v = n; v < N; v++) {
    const int c0 = child[2 * (v - n)], c1 = child[2 * (v - n) + 1];
    heights[v] = heights[c0] + branch_lengths[c0];
    if (fabs(heights[c1] + branch_lengths[c1] - heights[v]) > 1e-4) rc = ORACLE_ERR_BAD_TREE;
  }
  ratios[N - 1 - n] = heights[N - 1];
  for (int v = N - 2; v >= n; v--)
    ratios[v - n] = (heights[v] - bounds[v]) / (heights[parent_ids[v]] - bounds[v]);
  free(child);
  return rc;
}

/* RootedTree::InitializeTimeTreeUsingHeightRatios (rooted_tree.cpp:101-121); heights of the
 * leaves and bounds are inputs, internal heights and all branch lengths are outputs. */
void oracle_time_tree_from_height_ratios(int n, const int *parent_ids, const double *bounds,
                                         const double *ratios, double *heights, double *branch_lengths) {
  const int N = 2 * n - 1;
  heights[N - 1] = ratios[N - 1 - n];
  for (int v = N - 2; v >= 0; v--) {
    const int p = parent_ids[v];
    if (v >= n) heights[v] = bounds[v] + ratios[v - n] * (heights[p] - bounds[v]);
    branch_lengths[v] = heights[p] - heights[v];
  }
}

/* RootedGradientTransforms::LogDetJacobianHeightTransform (rooted_gradient_transforms.cpp:243-256) */
double oracle_log_det_jacobian(int n, const int *parent_ids, const double *heights, const double *bounds) {
  const int N = 2 * n - 1;
  double s = 0;
  for (int v = N - 2; v >= n; v--) s += log(heights[parent_ids[v]] - bounds[v]);
  return s;
}

/* HeightGradient (rooted_gradient_transforms.cpp:19-41) */
void oracle_height_gradient(int n, const int *parent_ids, const double *rates, const double *branch_gradient,
                            double *out) {
  const int N = 2 * n - 1;
  for (int i = 0; i < n - 1; i++) out[i] = 0;
  for (int v = n; v < N - 1; v++) out[v - n] = -branch_gradient[v] * rates[v];
  for (int i = 0; i < N - 1; i++) out[parent_ids[i] - n] += branch_gradient[i] * rates[i];
}

/* UpdateGradientUnWeightedLogDensity + UpdateHeightParameterGradientUnweightedLogDensity
 * (rooted_gradient_transforms.cpp:43-146,170-184): ratio gradient of a height gradient,
 * root entry = gradient with respect to the root height. */
void oracle_ratio_gradient_of_height_gradient(int n, const int *parent_ids, const double *heights,
                                              const double *bounds, const double *ratios,
                                              const double *height_gradient, double *out) {
  const int N = 2 * n - 1;
  int *child = (int *)malloc(sizeof(int) * 2 * (n - 1));
  double *mult = (double *)malloc(sizeof(double) * (n - 1));
  children_of(n, parent_ids, child);
  for (int i = 0; i < n - 1; i++) out[i] = 0;
  for (int v = n; v < N - 1; v++) {  /* post-order: children before parents */
    const double partial = (heights[v] - bounds[v]) / ratios[v - n];  /* GetNodePartial */
    double g = partial * height_gradient[v - n];
    for (int k = 0; k < 2; k++) {
      const int c = child[2 * (v - n) + k];
      if (c < n) continue;
      if (bounds[v] == bounds[c]) g += out[c - n] * ratios[c - n] / ratios[v - n];
      else g += out[c - n] * ratios[c - n] / (heights[v] - bounds[c]) * partial;
    }
    out[v - n] = g;
  }
  mult[N - 1 - n] = 1.0;
  for (int v = N - 2; v >= n; v--) mult[v - n] = ratios[v - n] * mult[parent_ids[v] - n];
  double s = 0;
  for (int i = 0; i < n - 1; i++) s += height_gradient[i] * mult[i];
  out[N - 1 - n] = s;
  free(child);
  free(mult);
}

/* GradientLogDeterminantJacobian (rooted_gradient_transforms.cpp:148-168) */
void oracle_gradient_log_det_jacobian(int n, const int *parent_ids, const double *heights, const double *bounds,
                                      const double *ratios, double *out) {
  double *log_time = (double *)calloc((size_t)(n - 1), sizeof(double));
  for (int i = 0; i < n - 2; i++) log_time[i] = 1.0 / (heights[n + i] - bounds[n + i]);  /* GetLogTimeArray */
  oracle_ratio_gradient_of_height_gradient(n, parent_ids, heights, bounds, ratios, log_time, out);
  for (int i = 0; i < n - 2; i++) out[i] -= 1.0 / ratios[i];
  free(log_time);
}

/* RatioGradientOfBranchGradient (rooted_gradient_transforms.cpp:186-241) */
void oracle_ratio_gradient_of_branch_gradient(int n, const int *parent_ids, const double *heights,
                                              const double *bounds, const double *ratios, const double *rates,
                                              const double *branch_gradient, int include_log_det_jacobian,
                                              double *out) {
  double *hg = (double *)malloc(sizeof(double) * (n - 1));
  oracle_height_gradient(n, parent_ids, rates, branch_gradient, hg);
  oracle_ratio_gradient_of_height_gradient(n, parent_ids, heights, bounds, ratios, hg, out);
  if (include_log_det_jacobian) {
    oracle_gradient_log_det_jacobian(n, parent_ids, heights, bounds, ratios, hg);
    for (int i = 0; i < n - 1; i++) out[i] += hg[i];
  }
  free(hg);
}
