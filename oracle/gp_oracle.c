/*
 * gp_oracle.c -- CPU ORACLE for bito's GPEngine (generalized pruning on the subsplit DAG).
 * TEST INFRASTRUCTURE ONLY (see bito_oracle.h for the rules).
 *
 * Restates, operation by operation, what GPEngine does to its PLV arena when it visits a
 * GPOperationVector (reference src/gp_engine.cpp:213-339, src/gp_engine.hpp:248-282,
 * src/gp_operation.hpp:24-170): JC69, one rate category, 4 x P column-major PLVs, 6 PLVs per
 * DAG node indexed type*node_count + node (src/pv_handler.hpp:487-490), whole-PLV rescaling
 * by powers of the threshold (src/gp_engine.cpp:564-601), per-edge log-likelihood matrix,
 * log-add marginal over rootsplits (src/numerical_utils.hpp:35-52).
 *
 * OptimizeBranchLength restates the five one-dimensional optimisers of
 * src/optimization.hpp:71-417 as DAGBranchHandler drives them
 * (src/dag_branch_handler.cpp:123-300, src/gp_engine.cpp:603-661).
 *
 * Parity status: PINNED by tests/test_gp.py against the reference's known answers for this
 * path (src/gp_doctest.cpp:119-131,257-346; src/gp_engine.hpp:382-393).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum {
  GP_ZERO_PLV = 0,
  GP_SET_TO_STATIONARY = 1,
  GP_INCREMENT_WITH_WEIGHTED_EVOLVED_PLV = 2,
  GP_MULTIPLY = 3,
  GP_LIKELIHOOD = 4,
  GP_OPTIMIZE_BRANCH_LENGTH = 5,
  GP_UPDATE_SBN_PROBABILITIES = 6,
  GP_RESET_MARGINAL_LIKELIHOOD = 7,
  GP_INCREMENT_MARGINAL_LIKELIHOOD = 8,
  GP_PREP_FOR_MARGINALIZATION = 9
};

typedef struct {
  uint32_t opcode, count;
  uint64_t a, b, c;
} gp_op;

typedef struct {
  int P, plv_count, gpcsp_count;
  int spare_plvs, spare_gpcsps; /* GrowSparePLVs / GrowSpareGPCSPs: slots behind the DAG's own ids */
  double threshold, log_threshold;
  double *plv;       /* [plv_count][P][4]  (column-major 4 x P per PLV) */
  int *counts;       /* [plv_count] */
  double *weights;   /* [P] */
  double *bl, *q;    /* [gpcsp_count] */
  double *ll;        /* [gpcsp_count][P] */
  double *marginal;  /* [P] */
  /* DAGBranchHandler state (src/dag_branch_handler.hpp:255-296) */
  double *diff;      /* [gpcsp_count] branch length differences of the last optimisation */
  int method;        /* OptimizationMethod, src/optimization.hpp:28-34 */
  int opt_count;     /* optimization_count_ */
  int significant_digits;
  /* test instruments (not part of the reference): a record of every function evaluation the Brent optimiser makes --
   * rows of 6: (edge, x = log branch length, f = negative log-likelihood, kind: 0 the handler's own evaluation of the
   * current length, 1 Brent's first point, 2 a trial point u, 3 the gradient variant's second trial; then how far from a
   * tie the decisions around this evaluation were: [4] the comparisons that CHOSE the point (convergence test, parabola
   * against golden section, clamps), smallest |lhs - rhs| / max(|lhs|, |rhs|), [5] the comparisons of its VALUE with the
   * best three points so far, smallest |difference|) -- and a relative perturbation of every evaluation's value, to
   * measure what rounding noise of a given size does to the iterates */
  double *trace;
  int trace_capacity, trace_rows;
  double noise;
  uint64_t noise_state;
} gp_oracle;

enum { OPT_BRENT = 0, OPT_BRENT_WITH_GRADIENTS = 1, OPT_GRADIENT_ASCENT = 2, OPT_LOGSPACE_GRADIENT_ASCENT = 3,
       OPT_NEWTON = 4 };
/* src/dag_branch_handler.hpp:266-295 */
static const double kMinLogBl = -13.9, kMaxLogBl = 1.1, kNewtonEps = 1e-10, kStep = 5e-4, kLogStep = 1.0005,
                    kDiffThreshold = 1e-15;
static const int kMaxIter = 1000;

/* JC69 eigensystem exactly as the reference builds it (src/substitution_model.cpp:20-26,
 * src/gp_engine.cpp:341-364). */
static const double kV[16] = {1.0, 2.0, 0.0, 0.5, 1.0, -2.0, 0.5, 0.0, 1.0, 2.0, 0.0, -0.5, 1.0, -2.0, -0.5, 0.0};
static const double kVi[16] = {0.25, 0.25, 0.25, 0.25, 0.125, -0.125, 0.125, -0.125,
                               0.0, 1.0, 0.0, -1.0, 1.0, 0.0, -1.0, 0.0};
static const double kLam[4] = {0.0, -1.3333333333333333, -1.3333333333333333, -1.3333333333333333};

static void matrices(double t, double *P, double *dP, double *ddP) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0, d = 0, dd = 0;
      for (int k = 0; k < 4; k++) {
        const double e = exp(kLam[k] * t), vv = kV[i * 4 + k];
        s += vv * e * kVi[k * 4 + j];
        d += vv * (kLam[k] * e) * kVi[k * 4 + j];
        dd += vv * (kLam[k] * kLam[k] * e) * kVi[k * 4 + j];
      }
      P[i * 4 + j] = s;
      if (dP) dP[i * 4 + j] = d;
      if (ddP) ddP[i * 4 + j] = dd;
    }
}

gp_oracle *gp_oracle_create(int taxon_count, int pattern_count, const int *patterns, const double *weights,
                            int node_count, int gpcsp_count, double threshold) {
  gp_oracle *g = (gp_oracle *)calloc(1, sizeof(*g));
  g->P = pattern_count;
  g->plv_count = 6 * node_count;
  g->gpcsp_count = gpcsp_count;
  g->threshold = threshold;
  g->log_threshold = log(threshold);
  g->plv = (double *)calloc((size_t)g->plv_count * pattern_count * 4, sizeof(double));
  g->counts = (int *)calloc(g->plv_count, sizeof(int));
  g->weights = (double *)malloc(sizeof(double) * pattern_count);
  memcpy(g->weights, weights, sizeof(double) * pattern_count);
  g->bl = (double *)calloc(gpcsp_count, sizeof(double));
  g->q = (double *)malloc(sizeof(double) * gpcsp_count);
  for (int i = 0; i < gpcsp_count; i++) g->q[i] = 1.0;
  g->ll = (double *)calloc((size_t)gpcsp_count * pattern_count, sizeof(double));
  g->marginal = (double *)calloc(pattern_count, sizeof(double));
  g->diff = (double *)calloc(gpcsp_count, sizeof(double));
  g->method = OPT_BRENT; /* GPEngine default: UseGradientOptimization(false) */
  g->opt_count = 0;
  g->significant_digits = 10;
  /* InitializePLVsWithSitePatterns (src/gp_engine.cpp:544-562): leaf P-PLVs (type 0) */
  for (int t = 0; t < taxon_count; t++)
    for (int p = 0; p < pattern_count; p++) {
      const int s = patterns[(size_t)t * pattern_count + p];
      double *col = g->plv + ((size_t)t * pattern_count + p) * 4;
      for (int i = 0; i < 4; i++) col[i] = (s >= 4 || s == i) ? 1.0 : 0.0;
    }
  return g;
}

void gp_oracle_destroy(gp_oracle *g) {
  if (!g) return;
  free(g->plv); free(g->counts); free(g->weights); free(g->bl); free(g->q); free(g->ll); free(g->marginal); free(g->diff);
  free(g);
}

void gp_oracle_set_branch_lengths(gp_oracle *g, const double *bl) { memcpy(g->bl, bl, sizeof(double) * g->gpcsp_count); }
void gp_oracle_set_sbn_parameters(gp_oracle *g, const double *q) { memcpy(g->q, q, sizeof(double) * g->gpcsp_count); }
void gp_oracle_get_sbn_parameters(const gp_oracle *g, double *q) { memcpy(q, g->q, sizeof(double) * g->gpcsp_count); }

static double *PLV(gp_oracle *g, uint64_t idx) { return g->plv + (size_t)idx * g->P * 4; }

static double log_add(double x, double y) { /* src/numerical_utils.hpp:35-52 */
  if (y > x) { double t = x; x = y; y = t; }
  if (x == -INFINITY) return x;
  const double neg_diff = y - x;
  if (neg_diff < -36.04365338911715) return x; /* LOG_EPS = log(DBL_EPSILON), numerical_utils.hpp:15-19 */
  return x + log(1.0 + exp(neg_diff));
}

/* RescalePLVIfNeeded (src/gp_engine.cpp:583-597): one decision for the whole PLV. */
static void rescale_if_needed(gp_oracle *g, uint64_t idx) {
  double *v = PLV(g, idx), mx = 0;
  for (int k = 0; k < g->P * 4; k++)
    if (v[k] > mx) mx = v[k];
  if (mx == 0) return;
  int count = 0;
  while (mx < g->threshold) {
    mx /= g->threshold;
    count++;
  }
  if (count == 0) return;
  const double f = pow(g->threshold, (double)count);
  for (int k = 0; k < g->P * 4; k++) v[k] /= f;
  g->counts[idx] += count;
}

/* per-pattern r^T M p */
static double bilinear(const double *r, const double *M, const double *p) {
  double s = 0;
  for (int i = 0; i < 4; i++) s += r[i] * (M[i * 4] * p[0] + M[i * 4 + 1] * p[1] + M[i * 4 + 2] * p[2] + M[i * 4 + 3] * p[3]);
  return s;
}

static void optimize_branch_length(gp_oracle *g, int edge, uint64_t rootward, uint64_t leafward);

int gp_oracle_process(gp_oracle *g, const gp_op *ops, int op_count, const uint64_t *side) {
  const int P = g->P;
  for (int o = 0; o < op_count; o++) {
    const gp_op *op = &ops[o];
    switch (op->opcode) {
      case GP_ZERO_PLV:
        memset(PLV(g, op->a), 0, sizeof(double) * P * 4);
        g->counts[op->a] = 0;
        break;
      case GP_SET_TO_STATIONARY: {
        double *v = PLV(g, op->a);
        for (int p = 0; p < P; p++)
          for (int i = 0; i < 4; i++) v[p * 4 + i] = g->q[op->b] * 0.25;
        g->counts[op->a] = 0;
        break;
      }
      case GP_INCREMENT_WITH_WEIGHTED_EVOLVED_PLV: {
        double M[16];
        matrices(g->bl[op->b], M, NULL, NULL);
        const int diff = g->counts[op->c] - g->counts[op->a];
        if (diff < 0) return -1;
        const double f = (diff == 0 ? 1.0 : pow(g->threshold, (double)diff)) * g->q[op->b];
        double *d = PLV(g, op->a);
        const double *s = PLV(g, op->c);
        for (int p = 0; p < P; p++)
          for (int i = 0; i < 4; i++)
            d[p * 4 + i] += f * (M[i * 4] * s[p * 4] + M[i * 4 + 1] * s[p * 4 + 1] + M[i * 4 + 2] * s[p * 4 + 2] +
                                 M[i * 4 + 3] * s[p * 4 + 3]);
        break;
      }
      case GP_MULTIPLY: {
        double *d = PLV(g, op->a);
        const double *x = PLV(g, op->b), *y = PLV(g, op->c);
        for (int k = 0; k < P * 4; k++) d[k] = x[k] * y[k];
        g->counts[op->a] = g->counts[op->b] + g->counts[op->c];
        rescale_if_needed(g, op->a);
        break;
      }
      case GP_LIKELIHOOD: { /* dest = gpcsp, b = child_ (r-PLV of the parent), c = parent_ (p-PLV of the child):
                               field order of GPOperations::Likelihood{dest, child, parent} as GPDAG fills it
                               (src/gp_dag.cpp:183-188): child_ <- RPLV(parent node), parent_ <- P(child node). */
        double M[16];
        matrices(g->bl[op->a], M, NULL, NULL);
        const double *r = PLV(g, op->c), *pp = PLV(g, op->b);
        const double resc = (g->counts[op->b] + g->counts[op->c]) * g->log_threshold;
        for (int p = 0; p < P; p++) g->ll[(size_t)op->a * P + p] = log(bilinear(r + p * 4, M, pp + p * 4)) + resc;
        break;
      }
      case GP_RESET_MARGINAL_LIKELIHOOD:
        for (int p = 0; p < P; p++) g->marginal[p] = -INFINITY;
        break;
      case GP_INCREMENT_MARGINAL_LIKELIHOOD: { /* a = stationary_times_prior, b = rootsplit, c = p */
        if (g->counts[op->a] != 0) return -2;
        const double *r = PLV(g, op->a), *pp = PLV(g, op->c);
        const double resc = g->counts[op->c] * g->log_threshold;
        for (int p = 0; p < P; p++) {
          double s = 0;
          for (int i = 0; i < 4; i++) s += r[p * 4 + i] * pp[p * 4 + i];
          const double row = log(s) + resc;
          g->marginal[p] = log_add(g->marginal[p], row);
          g->ll[(size_t)op->b * P + p] = row - log(g->q[op->b]);
        }
        break;
      }
      case GP_UPDATE_SBN_PROBABILITIES: { /* a = start, b = stop (src/gp_engine.cpp:297-321) */
        const int len = (int)(op->b - op->a);
        if (len == 1) {
          g->q[op->a] = 1.0;
          break;
        }
        double *lu = (double *)malloc(sizeof(double) * len), norm = -INFINITY;
        for (int k = 0; k < len; k++) {
          double s = 0;
          for (int p = 0; p < P; p++) s += g->ll[(size_t)(op->a + k) * P + p] * g->weights[p];
          lu[k] = s + log(g->q[op->a + k]);
          norm = log_add(norm, lu[k]);
        }
        for (int k = 0; k < len; k++) g->q[op->a + k] = exp(lu[k] - norm);
        free(lu);
        break;
      }
      case GP_PREP_FOR_MARGINALIZATION: { /* a = dest, b = offset into side, count sources */
        int mn = g->counts[side[op->b]];
        for (uint32_t k = 1; k < op->count; k++)
          if (g->counts[side[op->b + k]] < mn) mn = g->counts[side[op->b + k]];
        g->counts[op->a] = mn;
        break;
      }
      case GP_OPTIMIZE_BRANCH_LENGTH: /* a = leafward_, b = rootward_, c = gpcsp_ (src/gp_operation.hpp:118-127) */
        optimize_branch_length(g, (int)op->c, op->b, op->a);
        break;
      default:
        return -3;
    }
  }
  return 0;
}

double gp_oracle_log_marginal_likelihood(const gp_oracle *g) { /* src/gp_engine.cpp:413-415 */
  double s = 0;
  for (int p = 0; p < g->P; p++) s += g->marginal[p] * g->weights[p];
  return s;
}

void gp_oracle_per_gpcsp_log_likelihoods(const gp_oracle *g, double *out) { /* :437-440 */
  for (int e = 0; e < g->gpcsp_count; e++) {
    double s = 0;
    for (int p = 0; p < g->P; p++) s += g->ll[(size_t)e * g->P + p] * g->weights[p];
    out[e] = s;
  }
}

/* LogLikelihoodAndFirstTwoDerivatives (src/gp_engine.cpp:505-542) */
void gp_oracle_derivatives(gp_oracle *g, int gpcsp, uint64_t rootward, uint64_t leafward, double out[3]) {
  double M[16], dM[16], ddM[16];
  matrices(g->bl[gpcsp], M, dM, ddM);
  const double *r = PLV(g, rootward), *pp = PLV(g, leafward);
  const double resc = (g->counts[rootward] + g->counts[leafward]) * g->log_threshold;
  double ll = 0, d1 = 0, d2 = 0;
  for (int p = 0; p < g->P; p++) {
    const double l = bilinear(r + p * 4, M, pp + p * 4);
    const double a = bilinear(r + p * 4, dM, pp + p * 4);
    const double b = bilinear(r + p * 4, ddM, pp + p * 4);
    ll += g->weights[p] * (log(l) + resc);
    d1 += g->weights[p] * (a / l);
    d2 += g->weights[p] * ((b * l - a * a) / (l * l));
  }
  out[0] = ll;
  out[1] = d1;
  out[2] = d2;
}

/* GPEngine::GrowSparePLVs / GrowSpareGPCSPs (src/gp_engine.cpp:196-211): spare PLV ids start at
 * plv_count, spare GPCSP ids at gpcsp_count; contents are kept, counts only grow. */
static void *grow_zeroed(void *ptr, size_t old_bytes, size_t new_bytes) {
  char *fresh = (char *)realloc(ptr, new_bytes);
  memset(fresh + old_bytes, 0, new_bytes - old_bytes);
  return fresh;
}
void gp_oracle_grow_spare(gp_oracle *g, int spare_plvs, int spare_gpcsps) {
  if (spare_plvs > g->spare_plvs) {
    const size_t have = (size_t)g->plv_count + g->spare_plvs, want = (size_t)g->plv_count + spare_plvs;
    g->plv = (double *)grow_zeroed(g->plv, have * g->P * 4 * sizeof(double), want * g->P * 4 * sizeof(double));
    g->counts = (int *)grow_zeroed(g->counts, have * sizeof(int), want * sizeof(int));
    g->spare_plvs = spare_plvs;
  }
  if (spare_gpcsps > g->spare_gpcsps) {
    const size_t have = (size_t)g->gpcsp_count + g->spare_gpcsps, want = (size_t)g->gpcsp_count + spare_gpcsps;
    g->bl = (double *)grow_zeroed(g->bl, have * sizeof(double), want * sizeof(double));
    g->q = (double *)grow_zeroed(g->q, have * sizeof(double), want * sizeof(double));
    g->diff = (double *)grow_zeroed(g->diff, have * sizeof(double), want * sizeof(double));
    g->ll = (double *)grow_zeroed(g->ll, have * g->P * sizeof(double), want * g->P * sizeof(double));
    g->spare_gpcsps = spare_gpcsps;
  }
}
/* GPEngine::GrowPLVs + GrowGPCSPs with reindexers (src/gp_engine.cpp:64-193): old index -> new index,
 * NULL = identity; new nodes zeroed, new GPCSPs with branch length 0.1 and q = 1; spare slots emptied. */
void gp_oracle_grow(gp_oracle *g, int new_nodes, int new_gpcsps, const int64_t *node_map, const int64_t *gpcsp_map) {
  const int old_nodes = g->plv_count / 6, old_gpcsps = g->gpcsp_count;
  const size_t cell = (size_t)g->P * 4;
  {
    const size_t total = (size_t)6 * new_nodes + g->spare_plvs;
    double *plv = (double *)calloc(total * cell, sizeof(double));
    int *counts = (int *)calloc(total, sizeof(int));
    for (int t = 0; t < 6; t++)
      for (int v = 0; v < old_nodes; v++) {
        const size_t from = (size_t)t * old_nodes + v, to = (size_t)t * new_nodes + (node_map ? node_map[v] : v);
        memcpy(plv + to * cell, g->plv + from * cell, cell * sizeof(double));
        counts[to] = g->counts[from];
      }
    free(g->plv); free(g->counts);
    g->plv = plv; g->counts = counts; g->plv_count = 6 * new_nodes;
  }
  {
    const size_t total = (size_t)new_gpcsps + g->spare_gpcsps;
    double *bl = (double *)calloc(total, sizeof(double)), *q = (double *)calloc(total, sizeof(double));
    double *diff = (double *)calloc(total, sizeof(double)), *ll = (double *)calloc(total * g->P, sizeof(double));
    for (int i = 0; i < new_gpcsps; i++) {
      const size_t to = gpcsp_map ? (size_t)gpcsp_map[i] : (size_t)i;
      if (i < old_gpcsps) {
        bl[to] = g->bl[i]; q[to] = g->q[i]; diff[to] = g->diff[i];
        memcpy(ll + to * g->P, g->ll + (size_t)i * g->P, sizeof(double) * g->P);
      } else {
        bl[to] = 0.1; q[to] = 1.0;
      }
    }
    free(g->bl); free(g->q); free(g->diff); free(g->ll);
    g->bl = bl; g->q = q; g->diff = diff; g->ll = ll; g->gpcsp_count = new_gpcsps;
  }
}
/* GetPLV: out[4][P], one row per state (the arena is column-major 4 x P per PLV) */
void gp_oracle_get_plv(const gp_oracle *g, int plv, double *out) {
  for (int p = 0; p < g->P; p++)
    for (int i = 0; i < 4; i++) out[(size_t)i * g->P + p] = g->plv[((size_t)plv * g->P + p) * 4 + i];
}

/* rescaling_counts_ (src/gp_engine.hpp:300-330): one count per PLV */
void gp_oracle_rescaling_counts(const gp_oracle *g, int first, int count, int *out) {
  for (int k = 0; k < count; k++) out[k] = g->counts[first + k];
}

/* GPEngine::CopyGPCSPData (src/gp_engine.cpp:401-409) */
void gp_oracle_copy_gpcsp_data(gp_oracle *g, int src, int dst) {
  g->bl[dst] = g->bl[src];
  g->q[dst] = g->q[src];
}
/* GetPerGPCSPLogLikelihoods(start, length) / GetBranchLengths(start, length) (src/gp_engine.cpp:421-456) */
void gp_oracle_per_gpcsp_log_likelihoods_range(const gp_oracle *g, int first, int count, double *out) {
  for (int e = 0; e < count; e++) {
    double s = 0;
    for (int p = 0; p < g->P; p++) s += g->ll[(size_t)(first + e) * g->P + p] * g->weights[p];
    out[e] = s;
  }
}
void gp_oracle_branch_lengths_range(const gp_oracle *g, int first, int count, double *out) {
  memcpy(out, g->bl + first, sizeof(double) * count);
}

void gp_oracle_transition_matrix(double t, double *P16) { matrices(t, P16, NULL, NULL); }

void gp_oracle_get_branch_lengths(const gp_oracle *g, double *out) { memcpy(out, g->bl, sizeof(double) * g->gpcsp_count); }
void gp_oracle_get_branch_length_differences(const gp_oracle *g, double *out) {
  memcpy(out, g->diff, sizeof(double) * g->gpcsp_count);
}
void gp_oracle_set_optimization_method(gp_oracle *g, int method) { g->method = method; }
void gp_oracle_set_significant_digits(gp_oracle *g, int digits) { g->significant_digits = digits; }
void gp_oracle_reset_optimization_count(gp_oracle *g) { g->opt_count = 0; }
void gp_oracle_increment_optimization_count(gp_oracle *g) { g->opt_count++; }

/* ---- branch-length optimisation ----------------------------------------------------------- */

typedef struct {
  gp_oracle *g;
  int edge;
  uint64_t rootward, leafward;
} opt_ctx;

/* log-likelihood (and derivatives) of one edge at branch length t; does not touch g->bl */
static void eval_at(const opt_ctx *c, double t, double out[3]) {
  const double keep = c->g->bl[c->edge];
  c->g->bl[c->edge] = t;
  gp_oracle_derivatives(c->g, c->edge, c->rootward, c->leafward, out);
  c->g->bl[c->edge] = keep;
}

void gp_oracle_set_trace(gp_oracle *g, double *rows, int capacity) {
  g->trace = rows;
  g->trace_capacity = rows ? capacity : 0;
  g->trace_rows = 0;
}
int gp_oracle_trace_rows(const gp_oracle *g) { return g->trace_rows; }
void gp_oracle_set_eval_noise(gp_oracle *g, double relative, uint64_t seed) {
  g->noise = relative;
  g->noise_state = seed * 0x9E3779B97F4A7C15ull + 1;
}
static double g_trace_pre_margin = INFINITY; /* set by brent_minimize ahead of the evaluation it belongs to */
static double *g_trace_last_row = NULL;      /* the row of the last evaluation: its value margin is filled in afterwards */
static void trace_row(const opt_ctx *c, double x, double f, int kind) {
  gp_oracle *g = c->g;
  g_trace_last_row = NULL;
  if (g->trace && g->trace_rows < g->trace_capacity) {
    double *row = g->trace + 6 * (size_t)g->trace_rows;
    row[0] = c->edge; row[1] = x; row[2] = f; row[3] = kind; row[4] = g_trace_pre_margin; row[5] = INFINITY;
    g_trace_last_row = row;
  }
  if (g->trace) g->trace_rows++;
  g_trace_pre_margin = INFINITY;
}
static void margin_of(double lhs, double rhs) { /* one comparison lhs ? rhs of the iteration in progress */
  const double scale = fmax(fabs(lhs), fabs(rhs));
  if (!(scale > 0)) return; /* 0 against 0 (the first parabola: v = w = x) falls the same way whatever the rounding */
  const double m = fabs(lhs - rhs) / scale;
  if (m < g_trace_pre_margin) g_trace_pre_margin = m;
}
static void value_margin(double a, double b) {
  if (g_trace_last_row && fabs(a - b) < g_trace_last_row[5]) g_trace_last_row[5] = fabs(a - b);
}

/* brent_nongrad_func / brent_grad_func (src/gp_engine.cpp:605-625): x is the LOG branch length */
static int g_trace_kind = 2;
static double neg_ll(const opt_ctx *c, double x) {
  double o[3];
  eval_at(c, exp(x), o);
  double f = -o[0];
  if (c->g->noise != 0) { /* (test instrument) xorshift64*: f (1 + noise u), u uniform in [-1, 1) */
    uint64_t z = c->g->noise_state;
    z ^= z >> 12; z ^= z << 25; z ^= z >> 27;
    c->g->noise_state = z;
    const double u = (double)((z * 0x2545F4914F6CDD1Dull) >> 11) / 4503599627370496.0 - 1.0;
    f *= 1.0 + c->g->noise * u;
  }
  trace_row(c, x, f, g_trace_kind);
  return f;
}
static void neg_ll_and_derivative(const opt_ctx *c, double x, double *f, double *df) {
  double o[3];
  const double t = exp(x);
  eval_at(c, t, o);
  *f = -o[0];
  *df = -t * o[1];
}

/* Optimization::BrentMinimize / BrentMinimizeWithGradients (src/optimization.hpp:71-331) */
static void brent_minimize(const opt_ctx *c, int with_gradients, double guess, double min, double max,
                           int significant_digits, int max_iter, double step_size, double *x_out, double *fx_out) {
  const double tolerance = ldexp(1.0, 1 - significant_digits);
  const double golden = 0.3819660f;
  double x, w, v, u, delta, delta2, fu, fv, fw, fx, mid, fract1, fract2;
  w = v = x = guess;
  g_trace_kind = 1;
  fw = fv = fx = neg_ll(c, x);
  g_trace_kind = 2;
  delta2 = delta = 0;
  int count = max_iter;
  do {
    mid = (min + max) / 2;
    fract1 = tolerance * fabs(x) + tolerance / 4;
    fract2 = 2 * fract1;
    margin_of(fabs(x - mid), fract2 - (max - min) / 2);
    if (fabs(x - mid) <= (fract2 - (max - min) / 2)) {
      /* (no evaluation follows: the margin of the decision to stop goes with the last evaluation made) */
      if (g_trace_last_row && g_trace_pre_margin < g_trace_last_row[4]) g_trace_last_row[4] = g_trace_pre_margin;
      g_trace_pre_margin = INFINITY;
      break;
    }
    int use_bisection = 1, clamped = 0;
    margin_of(fabs(delta2), fract1);
    if (fabs(delta2) > fract1) {
      double r = (x - w) * (fx - fv);
      double q = (x - v) * (fx - fw);
      double p = (x - v) * q - (x - w) * r;
      q = 2 * (q - r);
      if (q > 0) p = -p;
      q = fabs(q);
      const double td = delta2;
      delta2 = delta;
      margin_of(fabs(p), fabs(q * td / 2));
      if (!(fabs(p) >= fabs(q * td / 2))) {
        margin_of(p, q * (min - x));
        if (!(p <= q * (min - x))) margin_of(p, q * (max - x));
      }
      if (!(fabs(p) >= fabs(q * td / 2)) && !(p <= q * (min - x)) && !(p >= q * (max - x))) {
        delta = p / q;
        u = x + delta;
        margin_of(u - min, fract2);
        margin_of(max - u, fract2);
        if (((u - min) < fract2) || ((max - u) < fract2)) {
          margin_of(mid, x);
          delta = (mid - x) < 0 ? -fabs(fract1) : fabs(fract1);
          clamped = 1; /* (|delta| = fract1 exactly: the comparison below is no decision) */
        }
        use_bisection = 0;
      }
    }
    if (use_bisection) {
      margin_of(x, mid);
      delta2 = (x >= mid) ? min - x : max - x;
      delta = golden * delta2;
    }
    if (!clamped) margin_of(fabs(delta), fract1);
    u = (fabs(delta) >= fract1) ? x + delta : (delta > 0 ? x + fabs(fract1) : x - fabs(fract1));
    fu = neg_ll(c, u);
    value_margin(fu, fx);
    int accepted = 0;
    if (fu <= fx) {
      if (u >= x) min = x; else max = x;
      v = w; w = x; x = u;
      fv = fw; fw = fx; fx = fu;
      accepted = 1;
    } else if (with_gradients) {
      double f0, df;
      neg_ll_and_derivative(c, x, &f0, &df);
      const double u2 = x - step_size * df;
      g_trace_kind = 3;
      const double fu2 = neg_ll(c, u2);
      g_trace_kind = 2;
      value_margin(fu2, fx);
      if (fu2 <= fx) {
        if (u2 >= x) min = x; else max = x;
        v = w; w = x; x = u2;
        fv = fw; fw = fx; fx = fu2;
        accepted = 1;
      }
    }
    if (!accepted) {
      if (u < x) min = u; else max = u;
      if (!(w == x)) value_margin(fu, fw);
      if ((fu <= fw) || (w == x)) {
        v = w; w = u;
        fv = fw; fw = fu;
      } else {
        if (!(v == x) && !(v == w)) value_margin(fu, fv);
        if ((fu <= fv) || (v == x) || (v == w)) {
          v = u;
          fv = fu;
        }
      }
    }
  } while (--count);
  *x_out = x;
  *fx_out = fx;
}

/* DAGBranchHandler::OptimizeBranchLength (src/dag_branch_handler.cpp:123-300) */
static void optimize_branch_length(gp_oracle *g, int edge, uint64_t rootward, uint64_t leafward) {
  const int check_convergence = g->opt_count != 0; /* !IsFirstOptimization() */
  if (check_convergence && g->diff[edge] < kDiffThreshold) return;
  opt_ctx c = {g, edge, rootward, leafward};
  const double current = g->bl[edge];
  switch (g->method) {
    case OPT_BRENT:
    case OPT_BRENT_WITH_GRADIENTS: {
      const double cur_log = log(current);
      g_trace_kind = 0;
      const double cur_nll = neg_ll(&c, cur_log);
      double x, fx;
      brent_minimize(&c, g->method == OPT_BRENT_WITH_GRADIENTS, cur_log, kMinLogBl, kMaxLogBl, g->significant_digits,
                     kMaxIter, kLogStep, &x, &fx);
      g->bl[edge] = fx > cur_nll ? exp(cur_log) : exp(x);
      g->diff[edge] = fabs(exp(cur_log) - g->bl[edge]);
      break;
    }
    case OPT_GRADIENT_ASCENT: { /* Optimization::GradientAscent (optimization.hpp:333-347); min_x is the
                                   handler's min LOG branch length, as the reference passes it */
      const double tolerance = pow(10, -g->significant_digits);
      double x = current;
      for (int iter = 0;; iter++) {
        double o[3];
        eval_at(&c, x, o);
        const double new_x = x + o[1] * kStep;
        x = new_x > kMinLogBl ? new_x : kMinLogBl;
        if (fabs(o[1]) < fabs(o[0]) * tolerance || iter >= kMaxIter) break;
      }
      g->bl[edge] = x;
      g->diff[edge] = fabs(current - x);
      break;
    }
    case OPT_LOGSPACE_GRADIENT_ASCENT: { /* optimization.hpp:349-367 */
      const double tolerance = pow(10, -g->significant_digits), min_x = exp(kMinLogBl);
      double x = current;
      for (int iter = 0;; iter++) {
        double o[3];
        const double y = log(x);
        eval_at(&c, x, o);
        const double new_x = exp(y + x * o[1] * kLogStep);
        x = new_x > min_x ? new_x : min_x;
        if (fabs(o[1]) < fabs(o[0]) * tolerance || iter >= kMaxIter) break;
      }
      g->bl[edge] = x;
      g->diff[edge] = fabs(current - x);
      break;
    }
    case OPT_NEWTON: { /* optimization.hpp:369-405 in the log branch length (gp_engine.cpp:643-655) */
      const double tolerance = pow(10, -g->significant_digits);
      double x = log(current);
      for (int iter = 0;; iter++) {
        double o[3];
        const double t = exp(x);
        eval_at(&c, t, o);
        const double f1 = t * o[1], f2 = f1 + t * t * o[2];
        if (fabs(f2) < kNewtonEps) break;
        double new_x = x - f1 / f2;
        if (new_x < kMinLogBl) new_x = x - 0.5 * (x - kMinLogBl);
        if (new_x > kMaxLogBl) new_x = x - 0.5 * (x - kMaxLogBl);
        const double delta = fabs(x - new_x);
        if (delta < tolerance || fabs(f1) < fabs(o[0]) * tolerance || iter == kMaxIter) break;
        x = new_x;
      }
      g->bl[edge] = exp(x);
      g->diff[edge] = fabs(current - g->bl[edge]);
      break;
    }
    default:
      break;
  }
}
