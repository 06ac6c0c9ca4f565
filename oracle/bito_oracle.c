/*
 * bito_oracle.c -- CPU ORACLE (test infrastructure; see bito_oracle.h).
 *
 * Plain C, FP64, no fast-math.  Each function cites the reference lines whose
 * behaviour it restates.  The arithmetic that bito delegates to BEAGLE
 * (beagle-dev/beagle-lib @ origin/hmc-clock, not vendored in the reference) is
 * restated from BEAGLE's published CPU algorithm for 4-state double precision
 * with manual scaling, driven through the same buffer map and operation lists
 * that src/fat_beagle.cpp builds.
 */
#include "bito_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define S 4 /* DNA; src/site_pattern.cpp:118, src/mmapped_plv.hpp:14 */
#define OP_NONE (-1) /* BEAGLE_OP_NONE */

enum { SUB_JC69 = 0, SUB_HKY = 1, SUB_GTR = 2 };

/* ------------------------------------------------------------------------ */
/* Model specification and block layout                                      */
/* src/phylo_model.cpp:6-31, src/block_specification.cpp:14-53: parameter row
 * = [substitution | site | clock]; inside a model, blocks are laid out in
 * std::map (alphabetical) key order: "substitution_model_frequencies" before
 * "substitution_model_rates" (src/substitution_model.hpp:38-39,82,102). */
typedef struct {
  int substitution;
  int category_count;
  int weibull; /* 0 = "constant" */
  int strict_clock;
  int freq_start, freq_len, rates_start, rates_len; /* within the full row */
  int shape_start, clock_start;
  int param_count;
} model_spec;

static int spec_parse(const char *sub, const char *site, const char *clock,
                      model_spec *m, char *err, int err_len) {
  memset(m, 0, sizeof(*m));
  /* src/substitution_model.cpp:6-18 */
  if (strcmp(sub, "JC69") == 0) {
    m->substitution = SUB_JC69;
  } else if (strcmp(sub, "HKY") == 0) {
    m->substitution = SUB_HKY;
  } else if (strcmp(sub, "GTR") == 0) {
    m->substitution = SUB_GTR;
  } else {
    snprintf(err, err_len, "Substitution model not known: %s", sub);
    return ORACLE_ERR_BAD_MODEL;
  }
  /* src/site_model.cpp:10-25 */
  if (strcmp(site, "constant") == 0) {
    m->weibull = 0;
    m->category_count = 1;
  } else if (strncmp(site, "weibull", 7) == 0) {
    const char *plus = strchr(site, '+');
    m->weibull = 1;
    m->category_count = plus ? atoi(plus + 1) : 4;
    if (m->category_count < 1) {
      snprintf(err, err_len, "Site model not known: %s", site);
      return ORACLE_ERR_BAD_MODEL;
    }
  } else {
    snprintf(err, err_len, "Site model not known: %s", site);
    return ORACLE_ERR_BAD_MODEL;
  }
  /* src/clock_model.cpp:6-15 */
  if (strcmp(clock, "none") == 0) {
    m->strict_clock = 0;
  } else if (strcmp(clock, "strict") == 0) {
    m->strict_clock = 1;
  } else {
    snprintf(err, err_len, "Clock model not known: %s", clock);
    return ORACLE_ERR_BAD_MODEL;
  }
  int at = 0;
  m->freq_start = m->rates_start = m->shape_start = m->clock_start = -1;
  if (m->substitution != SUB_JC69) {
    m->freq_start = at;
    m->freq_len = 4;
    at += 4;
    m->rates_start = at;
    m->rates_len = (m->substitution == SUB_GTR) ? 6 : 1;
    at += m->rates_len;
  }
  if (m->weibull) {
    m->shape_start = at;
    at += 1;
  }
  if (m->strict_clock) {
    m->clock_start = at;
    at += 1;
  }
  m->param_count = at;
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* Substitution models                                                       */

/* Cyclic Jacobi for a symmetric 4x4 (stands in for
 * Eigen::SelfAdjointEigenSolver<Matrix4d>, src/substitution_model.cpp:172).
 * Eigenvalues ascending like Eigen; eigenvector signs are immaterial because
 * only V diag(.) V^-1 products are ever formed. */
static void jacobi4(double A[4][4], double U[4][4], double w[4]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) U[i][j] = (i == j);
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0;
    for (int i = 0; i < 4; i++)
      for (int j = i + 1; j < 4; j++) off += A[i][j] * A[i][j];
    if (off < 1e-300) break;
    for (int p = 0; p < 4; p++)
      for (int q = p + 1; q < 4; q++) {
        if (A[p][q] == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) /
                   (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; k++) {
          double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 4; k++) {
          double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 4; k++) {
          double ukp = U[k][p], ukq = U[k][q];
          U[k][p] = c * ukp - s * ukq;
          U[k][q] = s * ukp + c * ukq;
        }
      }
  }
  for (int i = 0; i < 4; i++) w[i] = A[i][i];
  for (int i = 0; i < 4; i++) /* ascending selection sort, columns follow */
    for (int j = i + 1; j < 4; j++)
      if (w[j] < w[i]) {
        double tw = w[i];
        w[i] = w[j];
        w[j] = tw;
        for (int k = 0; k < 4; k++) {
          double tu = U[k][i];
          U[k][i] = U[k][j];
          U[k][j] = tu;
        }
      }
}

/* GTRModel::UpdateQMatrix / HKYModel::UpdateQMatrix
 * (src/substitution_model.cpp:49-76,141-166): upper-triangle order AC,AG,AT,
 * CG,CT,GT; Q_ij = r pi_j, Q_ji = r pi_i; diagonal = -rowsum; normalised to
 * unit expected rate. */
static void build_q(const double r6[6], const double pi[4], double Q[16]) {
  int k = 0;
  for (int i = 0; i < 4; i++)
    for (int j = i + 1; j < 4; j++) {
      double rate = r6[k++];
      Q[i * 4 + j] = rate * pi[j];
      Q[j * 4 + i] = rate * pi[i];
    }
  double total = 0;
  for (int i = 0; i < 4; i++) {
    double row = 0;
    for (int j = 0; j < 4; j++)
      if (i != j) row += Q[i * 4 + j];
    Q[i * 4 + i] = -row;
    total += row * pi[i];
  }
  for (int i = 0; i < 16; i++) Q[i] /= total;
}

/* DNAModel::UpdateEigendecomposition (src/substitution_model.cpp:168-182). */
static void eigen_reversible(const double Q[16], const double pi[4],
                             double V[16], double Vinv[16], double lam[4]) {
  double sq[4], A[4][4], U[4][4];
  for (int i = 0; i < 4; i++) sq[i] = sqrt(pi[i]);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) A[i][j] = sq[i] * Q[i * 4 + j] / sq[j];
  /* symmetrise away rounding asymmetry (Eigen reads the lower triangle) */
  for (int i = 0; i < 4; i++)
    for (int j = i + 1; j < 4; j++) A[i][j] = A[j][i];
  jacobi4(A, U, lam);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      V[i * 4 + j] = U[i][j] / sq[i];
      Vinv[i * 4 + j] = U[j][i] * sq[j];
    }
}

/* Returns 0 or ORACLE_ERR_BAD_PARAMS with message. `sub_params` points at the
 * substitution block: [freqs(4) | rates]. */
static int substitution_setup(int kind, const double *sub_params, double Q[16],
                              double V[16], double Vinv[16], double lam[4],
                              double pi[4], char *err, int err_len) {
  if (kind == SUB_JC69) {
    /* src/substitution_model.cpp:20-31 (closed form) */
    static const double v[16] = {1.0, 2.0,  0.0, 0.5,  1.0, -2.0, 0.5,  0.0,
                                 1.0, 2.0,  0.0, -0.5, 1.0, -2.0, -0.5, 0.0};
    static const double vi[16] = {0.25, 0.25, 0.25, 0.25, 0.125, -0.125,
                                  0.125, -0.125, 0.0,  1.0,  0.0,   -1.0,
                                  1.0,  0.0,  -1.0, 0.0};
    memcpy(V, v, sizeof(v));
    memcpy(Vinv, vi, sizeof(vi));
    lam[0] = 0.0;
    lam[1] = lam[2] = lam[3] = -1.3333333333333333;
    for (int i = 0; i < 4; i++) {
      pi[i] = 0.25;
      for (int j = 0; j < 4; j++) Q[i * 4 + j] = (i == j) ? -1.0 : 1.0 / 3.0;
    }
    return ORACLE_OK;
  }
  const double *freqs = sub_params;
  const double *rates = sub_params + 4;
  double fsum = freqs[0] + freqs[1] + freqs[2] + freqs[3];
  if (fabs(fsum - 1.) >= 0.001) { /* substitution_model.cpp:37-44,124-131 */
    snprintf(err, err_len, "%s frequencies do not sum to 1 +/- 0.001!",
             kind == SUB_GTR ? "GTR" : "HKY");
    return ORACLE_ERR_BAD_PARAMS;
  }
  for (int i = 0; i < 4; i++) pi[i] = freqs[i];
  if (kind == SUB_GTR) {
    double rsum = 0;
    for (int i = 0; i < 6; i++) rsum += rates[i];
    if (fabs(rsum - 1.) >= 0.001) { /* substitution_model.cpp:132-138 */
      snprintf(err, err_len, "GTR rates do not sum to 1 +/- 0.001!");
      return ORACLE_ERR_BAD_PARAMS;
    }
    build_q(rates, pi, Q);
    eigen_reversible(Q, pi, V, Vinv, lam);
    return ORACLE_OK;
  }
  /* HKY: substitution_model.cpp:49-118 (analytic decomposition) */
  double kappa = rates[0];
  double r6[6] = {1.0, kappa, 1.0, 1.0, kappa, 1.0};
  build_q(r6, pi, Q);
  double pa = pi[0], pc = pi[1], pg = pi[2], pt = pi[3];
  double pr = pa + pg, py = pc + pt;
  double beta = -1.0 / (2.0 * (pr * py + kappa * (pa * pg + pc * pt)));
  lam[0] = 0;
  lam[1] = beta;
  lam[2] = beta * (1 + py * (kappa - 1));
  lam[3] = beta * (1 + pr * (kappa - 1));
  memset(V, 0, 16 * sizeof(double));
  memset(Vinv, 0, 16 * sizeof(double));
  Vinv[0] = pa, Vinv[1] = pc, Vinv[2] = pg, Vinv[3] = pt;
  Vinv[4] = pa * py, Vinv[5] = -pc * pr, Vinv[6] = pg * py, Vinv[7] = -pt * pr;
  Vinv[2 * 4 + 1] = 1, Vinv[2 * 4 + 3] = -1;
  Vinv[3 * 4 + 0] = 1, Vinv[3 * 4 + 2] = -1;
  for (int i = 0; i < 4; i++) V[i * 4 + 0] = 1.0;
  V[0 * 4 + 1] = 1. / pr, V[1 * 4 + 1] = -1. / py, V[2 * 4 + 1] = 1. / pr,
          V[3 * 4 + 1] = -1. / py;
  V[1 * 4 + 2] = pt / py, V[3 * 4 + 2] = -pc / py;
  V[0 * 4 + 3] = pg / pr, V[2 * 4 + 3] = -pa / pr;
  return ORACLE_OK;
}

int oracle_substitution_model(const char *substitution, const double *params,
                              double *Q16, double *V16, double *Vinv16,
                              double *lambda4, double *pi4) {
  model_spec m;
  char err[128];
  int rc = spec_parse(substitution, "constant", "none", &m, err, sizeof(err));
  if (rc) return rc;
  return substitution_setup(m.substitution, params, Q16, V16, Vinv16, lambda4,
                            pi4, err, sizeof(err));
}

/* WeibullSiteModel::UpdateRates (src/site_model.cpp:37-62). */
int oracle_weibull_rates(int C, double shape, double *rates, double *props,
                         double *derivs) {
  if (C < 1) return ORACLE_ERR_BAD_ARG;
  double mean = 0, dmean = 0;
  double *du = (double *)malloc(sizeof(double) * C);
  for (int i = 0; i < C; i++) {
    double quantile = (2.0 * i + 1.0) / (2.0 * C);
    rates[i] = pow(-log(1.0 - quantile), 1.0 / shape);
    mean += rates[i];
    du[i] = -rates[i] * log(-log(1.0 - quantile)) / (shape * shape);
    dmean += du[i];
  }
  mean /= C;
  dmean /= C;
  for (int i = 0; i < C; i++) {
    if (derivs) derivs[i] = (du[i] * mean - rates[i] * dmean) / (mean * mean);
    rates[i] /= mean;
    if (props) props[i] = 1.0 / C;
  }
  free(du);
  return ORACLE_OK;
}

void oracle_transition_matrix(const double *V, const double *Vinv,
                              const double *lam, double t, double *P) {
  double e[4];
  for (int k = 0; k < 4; k++) e[k] = exp(lam[k] * t);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += V[i * 4 + k] * e[k] * Vinv[k * 4 + j];
      P[i * 4 + j] = s;
    }
}

/* ------------------------------------------------------------------------ */
/* Bifurcating tree with bito ids (leaves 0..n-1, root = 2n-2)               */

typedef struct {
  int n, N, root;
  int *c0, *c1; /* -1 for leaves */
  double *bl;   /* indexed by node id, length N */
} otree;

static void otree_free(otree *t) {
  free(t->c0);
  free(t->c1);
  free(t->bl);
  memset(t, 0, sizeof(*t));
}

/* Node::OfParentIdVector (src/node.cpp:511-551): children of a node are in
 * ascending child-id order; leaf count = smallest parent id.  For unrooted
 * input the trifurcating root is resolved exactly like
 * UnrootedTree::Detrifurcate (src/unrooted_tree.cpp:27-37): children 1 and 2
 * are joined under a new node that re-uses the old root id with branch length
 * 0, and a new root (id+1, branch 0) joins child 0 with it. */
static int otree_build(otree *t, int rooted, int node_count,
                       const int *parent_ids, const double *bl, char *err,
                       int err_len) {
  memset(t, 0, sizeof(*t));
  if (node_count < 3) {
    snprintf(err, err_len, "tree too small");
    return ORACLE_ERR_BAD_TREE;
  }
  int M = node_count, n = M;
  for (int i = 0; i < M - 1; i++)
    if (parent_ids[i] < n) n = parent_ids[i];
  int expect = rooted ? 2 * n - 1 : 2 * n - 2;
  if (M != expect || n < 2) {
    snprintf(err, err_len,
             "node_count %d inconsistent with leaf count %d for a%s tree", M, n,
             rooted ? " rooted" : "n unrooted");
    return ORACLE_ERR_BAD_TREE;
  }
  int N = 2 * n - 1;
  t->n = n;
  t->N = N;
  t->root = N - 1;
  t->c0 = (int *)malloc(sizeof(int) * N);
  t->c1 = (int *)malloc(sizeof(int) * N);
  t->bl = (double *)calloc(N, sizeof(double));
  int *c2 = (int *)malloc(sizeof(int) * N);
  for (int i = 0; i < N; i++) t->c0[i] = t->c1[i] = c2[i] = -1;
  int rc = ORACLE_OK;
  for (int child = 0; child < M - 1 && rc == ORACLE_OK; child++) {
    int p = parent_ids[child];
    if (p <= child || p >= M || p < n) {
      snprintf(err, err_len, "parent id %d of node %d is not a valid internal id", p,
               child);
      rc = ORACLE_ERR_BAD_TREE;
    } else if (t->c0[p] < 0) {
      t->c0[p] = child;
    } else if (t->c1[p] < 0) {
      t->c1[p] = child;
    } else if (!rooted && p == M - 1 && c2[p] < 0) {
      c2[p] = child;
    } else {
      snprintf(err, err_len, "node %d has too many children", p);
      rc = ORACLE_ERR_BAD_TREE;
    }
  }
  for (int i = n; i < M && rc == ORACLE_OK; i++)
    if (t->c0[i] < 0 || t->c1[i] < 0 || (!rooted && i == M - 1 && c2[i] < 0)) {
      snprintf(err, err_len, "internal node %d has too few children", i);
      rc = ORACLE_ERR_BAD_TREE;
    }
  if (rc != ORACLE_OK) {
    free(c2);
    otree_free(t);
    return rc;
  }
  for (int i = 0; i < M; i++) t->bl[i] = bl[i];
  if (!rooted) {
    int r = M - 1; /* old root id = 2n-3 */
    int a = t->c0[r], b = t->c1[r], c = c2[r];
    t->c0[r] = b; /* root12 = Join(children[1], children[2], our_id) */
    t->c1[r] = c;
    t->bl[r] = 0.;
    t->c0[r + 1] = a; /* Join(children[0], root12, our_id + 1) */
    t->c1[r + 1] = r;
    t->bl[r + 1] = 0.;
  }
  free(c2);
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* The BEAGLE-instance equivalent                                            */

typedef struct {
  int dest, scale_write, scale_read, child1, child1_mat, child2, child2_mat;
} bop; /* BeagleOperation field order: src/fat_beagle.cpp:345-352 */

typedef struct {
  model_spec spec;
  int n, N, P, C;
  int use_tip_states, rescaling;
  const int *patterns; /* [n][P], shared, owned by engine */
  const double *weights;
  /* model state (PhyloModel + what UpdatePhyloModelInBeagle uploads) */
  double Q[16], V[16], Vinv[16], lam[4], pi[4];
  double *cat_rates, *cat_weights, *cat_rate_derivs;
  double *params; /* last row set */
  /* buffers, indexed like FatBeagle::CreateInstance (fat_beagle.cpp:218-267) */
  double **partials; /* 2N slots: [0,n) tips, [n,N) post, [N+id] pre */
  double *matrices;  /* 2N * C * 16 */
  double *scale;     /* (2N+1) * P log-scalers; 0 = cumulative */
  bop *ops;
  int *stack;
  char err[256];
} ofb;

static size_t plv_len(const ofb *f) { return (size_t)f->C * f->P * S; }

static double *partial_buf(ofb *f, int idx) {
  if (!f->partials[idx])
    f->partials[idx] = (double *)malloc(sizeof(double) * plv_len(f));
  return f->partials[idx];
}

/* SitePattern::GetPartials (src/site_pattern.cpp:117-131) tiled over
 * categories, as beagleSetTipPartials does. */
static void set_tip_partials(ofb *f) {
  for (int tip = 0; tip < f->n; tip++) {
    double *buf = partial_buf(f, tip);
    for (int c = 0; c < f->C; c++)
      for (int p = 0; p < f->P; p++) {
        int st = f->patterns[(size_t)tip * f->P + p];
        double *x = buf + ((size_t)c * f->P + p) * S;
        for (int i = 0; i < S; i++) x[i] = (st >= S || st == i) ? 1.0 : 0.0;
      }
  }
}

static void ofb_free(ofb *f) {
  if (!f) return;
  if (f->partials)
    for (int i = 0; i < 2 * f->N; i++) free(f->partials[i]);
  free(f->partials);
  free(f->matrices);
  free(f->scale);
  free(f->cat_rates);
  free(f->cat_weights);
  free(f->cat_rate_derivs);
  free(f->params);
  free(f->ops);
  free(f->stack);
  free(f);
}

/* PhyloModel::SetParameters + FatBeagle::UpdatePhyloModelInBeagle
 * (src/phylo_model.cpp:26-31, src/fat_beagle.cpp:42-45,284-311). */
static int ofb_set_parameters(ofb *f, const double *row) {
  const model_spec *m = &f->spec;
  if (m->param_count > 0) memcpy(f->params, row, sizeof(double) * m->param_count);
  const double *sub = (m->freq_start >= 0) ? row + m->freq_start : NULL;
  int rc = substitution_setup(m->substitution, sub, f->Q, f->V, f->Vinv, f->lam,
                              f->pi, f->err, sizeof(f->err));
  if (rc) return rc;
  if (m->weibull) {
    double shape = row[m->shape_start];
    oracle_weibull_rates(f->C, shape, f->cat_rates, f->cat_weights,
                         f->cat_rate_derivs);
  } else {
    f->cat_rates[0] = 1.0; /* ConstantSiteModel */
    f->cat_weights[0] = 1.0;
    f->cat_rate_derivs[0] = 0.0;
  }
  return ORACLE_OK;
}

static ofb *ofb_create(const model_spec *spec, int n, int P, const int *patterns,
                       const double *weights, int use_tip_states) {
  ofb *f = (ofb *)calloc(1, sizeof(ofb));
  f->spec = *spec;
  f->n = n;
  f->N = 2 * n - 1;
  f->P = P;
  f->C = spec->category_count;
  f->use_tip_states = use_tip_states;
  f->patterns = patterns;
  f->weights = weights;
  f->partials = (double **)calloc(2 * f->N, sizeof(double *));
  f->matrices = (double *)calloc((size_t)2 * f->N * f->C * 16, sizeof(double));
  f->scale = (double *)calloc((size_t)(2 * f->N + 1) * P, sizeof(double));
  f->cat_rates = (double *)calloc(f->C, sizeof(double));
  f->cat_weights = (double *)calloc(f->C, sizeof(double));
  f->cat_rate_derivs = (double *)calloc(f->C, sizeof(double));
  f->params = (double *)calloc(spec->param_count + 1, sizeof(double));
  f->ops = (bop *)malloc(sizeof(bop) * 2 * f->N);
  f->stack = (int *)malloc(sizeof(int) * 4 * f->N);
  if (!use_tip_states) set_tip_partials(f);
  /* default model = the model's default parameters (JC69 / rates all equal):
   * callers always SetParameters before use (fat_beagle.hpp:177). */
  return f;
}

/* beagleUpdateTransitionMatrices: P = V diag(exp(lam * t * r_c)) V^-1 for each
 * listed branch and category (call site fat_beagle.cpp:315-325). */
static void update_transition_matrices(ofb *f, const int *idx, const double *t,
                                       int count) {
  for (int b = 0; b < count; b++)
    for (int c = 0; c < f->C; c++)
      oracle_transition_matrix(f->V, f->Vinv, f->lam, t[b] * f->cat_rates[c],
                               f->matrices + ((size_t)idx[b] * f->C + c) * 16);
}

/* out[i] = sum_j M[i][j] x[j], or column lookup for a compact (tip-state)
 * child: state >= S means gap => 1 (BEAGLE's extra all-ones column). */
static inline void apply_child(const ofb *f, int child, const double *M, int c,
                               int p, double out[S]) {
  if (f->use_tip_states && child < f->n) {
    int st = f->patterns[(size_t)child * f->P + p];
    if (st >= S) {
      out[0] = out[1] = out[2] = out[3] = 1.0;
    } else {
      for (int i = 0; i < S; i++) out[i] = M[i * 4 + st];
    }
  } else {
    const double *x = f->partials[child] + ((size_t)c * f->P + p) * S;
    for (int i = 0; i < S; i++)
      out[i] = M[i * 4 + 0] * x[0] + M[i * 4 + 1] * x[1] + M[i * 4 + 2] * x[2] +
               M[i * 4 + 3] * x[3];
  }
}

/* BEAGLE manual rescaling: per pattern, max over categories and states;
 * divide; store log(max); accumulate into the cumulative buffer. */
static void rescale_partials(ofb *f, double *dest, int scale_write,
                             int cumulative) {
  double *sw = f->scale + (size_t)scale_write * f->P;
  double *cum = cumulative >= 0 ? f->scale + (size_t)cumulative * f->P : NULL;
  for (int p = 0; p < f->P; p++) {
    double mx = 0;
    for (int c = 0; c < f->C; c++) {
      const double *x = dest + ((size_t)c * f->P + p) * S;
      for (int i = 0; i < S; i++)
        if (x[i] > mx) mx = x[i];
    }
    if (mx == 0) mx = 1.0;
    double inv = 1.0 / mx;
    for (int c = 0; c < f->C; c++) {
      double *x = dest + ((size_t)c * f->P + p) * S;
      for (int i = 0; i < S; i++) x[i] *= inv;
    }
    sw[p] = log(mx);
    if (cum) cum[p] += sw[p];
  }
}

/* beagleUpdatePartials (call sites fat_beagle.cpp:59-62,133-135). */
static void update_partials(ofb *f, const bop *ops, int count, int cumulative) {
  for (int o = 0; o < count; o++) {
    const bop *op = &ops[o];
    double *dest = partial_buf(f, op->dest);
    for (int c = 0; c < f->C; c++) {
      const double *M1 = f->matrices + ((size_t)op->child1_mat * f->C + c) * 16;
      const double *M2 = f->matrices + ((size_t)op->child2_mat * f->C + c) * 16;
      for (int p = 0; p < f->P; p++) {
        double a[S], b[S];
        apply_child(f, op->child1, M1, c, p, a);
        apply_child(f, op->child2, M2, c, p, b);
        double *d = dest + ((size_t)c * f->P + p) * S;
        for (int i = 0; i < S; i++) d[i] = a[i] * b[i];
      }
    }
    if (op->scale_write >= 0) rescale_partials(f, dest, op->scale_write, cumulative);
  }
}

/* beagleUpdatePrePartials (call site fat_beagle.cpp:143-145, op layout
 * :355-373): child1 = pre-order partial of the parent, matrix1 =
 * this node's matrix (applied transposed), child2 = sibling's post-order
 * partial, matrix2 = sibling's matrix. */
static void update_pre_partials(ofb *f, const bop *ops, int count,
                                int cumulative) {
  for (int o = 0; o < count; o++) {
    const bop *op = &ops[o];
    double *dest = partial_buf(f, op->dest);
    const double *par = f->partials[op->child1];
    for (int c = 0; c < f->C; c++) {
      const double *M1 = f->matrices + ((size_t)op->child1_mat * f->C + c) * 16;
      const double *M2 = f->matrices + ((size_t)op->child2_mat * f->C + c) * 16;
      for (int p = 0; p < f->P; p++) {
        double sib[S], u[S];
        apply_child(f, op->child2, M2, c, p, sib);
        const double *pp = par + ((size_t)c * f->P + p) * S;
        for (int i = 0; i < S; i++) u[i] = pp[i] * sib[i];
        double *d = dest + ((size_t)c * f->P + p) * S;
        for (int j = 0; j < S; j++)
          d[j] = M1[0 * 4 + j] * u[0] + M1[1 * 4 + j] * u[1] +
                 M1[2 * 4 + j] * u[2] + M1[3 * 4 + j] * u[3];
      }
    }
    if (op->scale_write >= 0) rescale_partials(f, dest, op->scale_write, cumulative);
  }
}

/* beagleCalculateRootLogLikelihoods (fat_beagle.cpp:63-68). */
static double root_log_likelihood(ofb *f, int root_buf, int cumulative) {
  const double *root = f->partials[root_buf];
  const double *cum = cumulative >= 0 ? f->scale + (size_t)cumulative * f->P : NULL;
  double total = 0;
  for (int p = 0; p < f->P; p++) {
    double site = 0;
    for (int c = 0; c < f->C; c++) {
      const double *x = root + ((size_t)c * f->P + p) * S;
      site += f->cat_weights[c] * (f->pi[0] * x[0] + f->pi[1] * x[1] +
                                   f->pi[2] * x[2] + f->pi[3] * x[3]);
    }
    double lp = log(site);
    if (cum) lp += cum[p];
    total += f->weights[p] * lp;
  }
  return total;
}

/* beagleCalculateEdgeDerivatives (fat_beagle.cpp:151-160):
 * g_b = sum_p w_p * [sum_c w_c pre^T dQ_c post] / [sum_c w_c pre^T post]. */
static void edge_derivatives(ofb *f, const int *post_idx, const int *pre_idx,
                             const double *dQ, int count, double *out_sum) {
  for (int b = 0; b < count; b++) {
    int post = post_idx[b];
    const double *pre = f->partials[pre_idx[b]];
    int compact = f->use_tip_states && post < f->n;
    double g = 0;
    for (int p = 0; p < f->P; p++) {
      double num = 0, den = 0;
      for (int c = 0; c < f->C; c++) {
        const double *D = dQ + (size_t)c * 16;
        const double *u = pre + ((size_t)c * f->P + p) * S;
        double x[S];
        if (compact) {
          int st = f->patterns[(size_t)post * f->P + p];
          for (int i = 0; i < S; i++) x[i] = (st >= S || st == i) ? 1.0 : 0.0;
        } else {
          const double *xp = f->partials[post] + ((size_t)c * f->P + p) * S;
          for (int i = 0; i < S; i++) x[i] = xp[i];
        }
        double nc = 0, dc = 0;
        for (int i = 0; i < S; i++) {
          double qx = D[i * 4 + 0] * x[0] + D[i * 4 + 1] * x[1] +
                      D[i * 4 + 2] * x[2] + D[i * 4 + 3] * x[3];
          nc += u[i] * qx;
          dc += u[i] * x[i];
        }
        num += f->cat_weights[c] * nc;
        den += f->cat_weights[c] * dc;
      }
      g += f->weights[p] * (num / den);
    }
    out_sum[b] = g;
  }
}

/* ------------------------------------------------------------------------ */
/* Traversals (src/node.cpp:150-173,232-304)                                 */

/* Node::BinaryIdPostorder: children in order, then the node. */
static int postorder_ops(ofb *f, const otree *t, bop *ops) {
  int *st = f->stack, sp = 0, count = 0;
  st[sp++] = t->root * 2; /* low bit = visited */
  while (sp) {
    int top = st[--sp], node = top >> 1;
    if (t->c0[node] < 0) continue;
    if (top & 1) {
      /* AddLowerPartialOperation (fat_beagle.cpp:338-353) */
      bop op = {node,
                f->rescaling ? node - t->n + 1 : OP_NONE,
                OP_NONE,
                t->c0[node],
                t->c0[node],
                t->c1[node],
                t->c1[node]};
      ops[count++] = op;
    } else {
      st[sp++] = node * 2 + 1;
      st[sp++] = t->c1[node] * 2;
      st[sp++] = t->c0[node] * 2;
    }
  }
  return count;
}

/* Node::TripleIdPreorderBifurcating (src/node.cpp:266-304): (child0, child1,
 * node), child0's subtree, (child1, child0, node), child1's subtree. */
static int preorder_ops(ofb *f, const otree *t, bop *ops) {
  int *st = f->stack, sp = 0, count = 0;
  int internal_count = t->n - 1;
  if (t->c0[t->root] < 0) return 0;
  st[sp++] = t->root * 2;
  while (sp) {
    int top = st[--sp], node = top >> 1, visited = top & 1;
    int a = visited ? t->c1[node] : t->c0[node];
    int sis = visited ? t->c0[node] : t->c1[node];
    /* AddUpperPartialOperation (fat_beagle.cpp:355-373) */
    bop op = {a + t->N,
              f->rescaling ? a + 1 + internal_count : OP_NONE,
              OP_NONE,
              node + t->N,
              a,
              sis,
              sis};
    ops[count++] = op;
    if (!visited) st[sp++] = node * 2 + 1;
    if (t->c0[a] >= 0) st[sp++] = a * 2;
  }
  return count;
}

/* FatBeagle::LogLikelihoodInternals (src/fat_beagle.cpp:49-69). */
static double log_likelihood_internals(ofb *f, const otree *t, const double *bl) {
  int cumulative = f->rescaling ? 0 : OP_NONE;
  memset(f->scale, 0, sizeof(double) * f->P); /* beagleResetScaleFactors(0) */
  int nops = postorder_ops(f, t, f->ops);
  int *idx = f->stack;
  for (int i = 0; i < t->N - 1; i++) idx[i] = i;
  update_transition_matrices(f, idx, bl, t->N - 1);
  update_partials(f, f->ops, nops, cumulative);
  return root_log_likelihood(f, t->root, cumulative);
}

/* FatBeagle::BranchGradientInternals (src/fat_beagle.cpp:113-169).
 * `scalers` are the per-category multipliers of Q (BuildDifferentialMatrices,
 * :101-111).  gradient has N entries, root entry stays 0. */
static double branch_gradient_internals(ofb *f, const otree *t, const double *bl,
                                        const double *scalers, double *gradient) {
  int cumulative = f->rescaling ? 0 : OP_NONE;
  int N = t->N;
  memset(f->scale, 0, sizeof(double) * f->P);
  int *idx = f->stack;
  for (int i = 0; i < N - 1; i++) idx[i] = i;
  update_transition_matrices(f, idx, bl, N - 1);
  /* SetRootPreorderPartialsToStateFrequencies (:327-336) */
  double *rootpre = partial_buf(f, t->root + N);
  for (size_t k = 0; k < (size_t)f->C * f->P; k++)
    for (int i = 0; i < S; i++) rootpre[k * S + i] = f->pi[i];
  double *dQ = (double *)malloc(sizeof(double) * f->C * 16);
  for (int c = 0; c < f->C; c++)
    for (int i = 0; i < 16; i++) dQ[c * 16 + i] = f->Q[i] * scalers[c];
  int nops = postorder_ops(f, t, f->ops);
  update_partials(f, f->ops, nops, cumulative);
  nops = preorder_ops(f, t, f->ops);
  update_pre_partials(f, f->ops, nops, OP_NONE);
  int *post_idx = (int *)malloc(sizeof(int) * 2 * N);
  int *pre_idx = post_idx + N;
  for (int i = 0; i < N - 1; i++) {
    post_idx[i] = i;
    pre_idx[i] = N + i;
  }
  for (int i = 0; i < N; i++) gradient[i] = 0.;
  edge_derivatives(f, post_idx, pre_idx, dQ, N - 1, gradient);
  free(post_idx);
  free(dQ);
  return root_log_likelihood(f, t->root, cumulative);
}

/* ------------------------------------------------------------------------ */
/* Per-tree entry points (FatBeagle::LogLikelihood / Gradient)               */

typedef struct {
  int rooted, node_count;
  const int *parent_ids;
  const double *bl, *rates;
} tree_in;

/* FatBeagle::LogLikelihood(UnrootedTree) :71-76 and (RootedTree) :83-98. */
static int fb_log_likelihood(ofb *f, const tree_in *in, double *out) {
  otree t;
  int rc = otree_build(&t, in->rooted, in->node_count, in->parent_ids, in->bl,
                       f->err, sizeof(f->err));
  if (rc) return rc;
  if (in->rooted && in->rates)
    for (int i = 0; i < t.N - 1; i++) t.bl[i] *= in->rates[i];
  *out = log_likelihood_internals(f, &t, t.bl);
  otree_free(&t);
  return ORACLE_OK;
}

/* StickBreakingTransform (src/stick_breaking_transform.cpp:10-44), the Stan
 * simplex transform: x = T(y), y = T^-1(x). */
static void stick_forward(const double *y, int K, double *x) {
  double stick = 1.0;
  for (int k = 0; k < K - 1; k++) {
    double z = 1.0 / (1 + exp(-(y[k] - log((double)(K - k - 1)))));
    x[k] = stick * z;
    stick -= x[k];
  }
  x[K - 1] = stick;
}
static void stick_inverse(const double *x, int K, double *y) {
  double sum = 0;
  for (int k = 0; k < K - 1; k++) {
    double z = x[k] / (1.0 - sum);
    y[k] = log(z / (1.0 - z)) + log((double)(K - k - 1));
    sum += x[k];
  }
}

/* SubstitutionModelGradientFiniteDifference (fat_beagle.cpp:422-460): central
 * differences on each entry of one block, in the reparameterised space when
 * the stick-breaking transform is requested (then the block has K-1 free
 * entries), otherwise on the raw entries without renormalisation. */
static int subst_block_fd(ofb *f, const tree_in *in, double *row, int start, int len,
                          int stick, double delta, double *out, int *nout) {
  double x[8], y[8];
  int rc = ORACLE_OK, dim = stick ? len - 1 : len;
  memcpy(x, row + start, sizeof(double) * len);
  if (stick)
    stick_inverse(x, len, y);
  else
    memcpy(y, x, sizeof(double) * len);
  for (int i = 0; i < dim && !rc; i++) {
    double orig = y[i], lp = 0, lm = 0;
    y[i] = orig + delta;
    if (stick) stick_forward(y, len, row + start); else row[start + i] = y[i];
    if ((rc = ofb_set_parameters(f, row))) break;
    if ((rc = fb_log_likelihood(f, in, &lp))) break;
    y[i] = orig - delta;
    if (stick) stick_forward(y, len, row + start); else row[start + i] = y[i];
    if ((rc = ofb_set_parameters(f, row))) break;
    if ((rc = fb_log_likelihood(f, in, &lm))) break;
    out[i] = (lp - lm) / (2. * delta);
    y[i] = orig;
    memcpy(row + start, x, sizeof(double) * len);
  }
  *nout = dim;
  return rc;
}

/* FatBeagle::SubstitutionModelGradient (fat_beagle.cpp:462-508): frequencies
 * use the stick-breaking transform when requested; rates only when there are
 * six of them (GTR).  Output: rates first, then frequencies (:528-531). */
static int subst_model_fd(ofb *f, const tree_in *in, int stick, double delta,
                          double *out, int *out_len) {
  const model_spec *m = &f->spec;
  double *row = (double *)malloc(sizeof(double) * (m->param_count + 1));
  memcpy(row, f->params, sizeof(double) * m->param_count);
  int nr = 0, nf = 0;
  int rc = subst_block_fd(f, in, row, m->rates_start, m->rates_len,
                          stick && m->rates_len == 6, delta, out, &nr);
  if (!rc)
    rc = subst_block_fd(f, in, row, m->freq_start, m->freq_len, stick, delta,
                        out + nr, &nf);
  memcpy(row, f->params, sizeof(double) * m->param_count);
  int rc2 = ofb_set_parameters(f, row);
  free(row);
  if (out_len) *out_len = nr + nf;
  return rc ? rc : rc2;
}

/* FatBeagle::Gradient(UnrootedTree) :510-557 and (RootedTree) :559-619. */
static int fb_gradient(ofb *f, const tree_in *in, int flags, double fd_delta,
                       double *out_ll, double *out_branch, double *out_site,
                       double *out_subst, double *out_clock) {
  otree t;
  int rc = otree_build(&t, in->rooted, in->node_count, in->parent_ids, in->bl,
                       f->err, sizeof(f->err));
  if (rc) return rc;
  int fixed_node = t.c1[t.root], root_child = t.c0[t.root];
  if (!in->rooted) {
    /* Tree::SlideRootPosition (src/tree.cpp:82-88) */
    t.bl[root_child] += t.bl[fixed_node];
    t.bl[fixed_node] = 0.;
  } else if (in->rates) {
    for (int i = 0; i < t.N - 1; i++) t.bl[i] *= in->rates[i];
  }
  *out_ll = branch_gradient_internals(f, &t, t.bl, f->cat_rates, out_branch);
  if ((flags & ORACLE_GRAD_SUBSTITUTION_MODEL) && out_subst && f->spec.rates_len > 0)
    rc = subst_model_fd(f, in, (flags & ORACLE_GRAD_STICKBREAKING) != 0,
                        fd_delta > 0 ? fd_delta : 1.e-6, out_subst, NULL);
  if (!rc && (flags & ORACLE_GRAD_SITE_MODEL) && out_site && f->C > 1) {
    /* second pass with dQ = Q * d r_c / d shape; then sum_b g_b t_b
     * (DiscreteSiteModelGradient, fat_beagle.cpp:401-410) */
    double *g2 = (double *)malloc(sizeof(double) * t.N);
    branch_gradient_internals(f, &t, t.bl, f->cat_rate_derivs, g2);
    double s = 0;
    for (int i = 0; i < t.N - 1; i++) s += g2[i] * t.bl[i];
    *out_site = s;
    free(g2);
  }
  if (!in->rooted) {
    out_branch[fixed_node] = 0.; /* :553 */
  } else if (!rc && (flags & ORACLE_GRAD_CLOCK_MODEL) && out_clock) {
    /* ClockGradient, strict clock (fat_beagle.cpp:379-399): uses the tree's
     * own (time) branch lengths, not the rate-scaled ones. */
    double s = 0;
    for (int i = 0; i < t.N - 1; i++) s += out_branch[i] * in->bl[i];
    *out_clock = s;
  }
  otree_free(&t);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* Engine + FatBeagleParallelize + TaskProcessor                             */
/* (src/engine.cpp:10-31, src/fat_beagle.hpp:151-184, task_processor.hpp)    */

struct oracle_engine {
  model_spec spec;
  int thread_count, n, P;
  int *patterns;
  double *weights;
  ofb **fbs;
  char err[256];
};

oracle_engine *oracle_engine_create(const char *substitution, const char *site,
                                    const char *clock, int thread_count,
                                    int use_tip_states, int taxon_count,
                                    int pattern_count, const int *patterns,
                                    const double *weights, char *err,
                                    int err_len) {
  model_spec spec;
  if (spec_parse(substitution, site, clock, &spec, err, err_len)) return NULL;
  if (thread_count <= 0) { /* engine.cpp:14-16 */
    snprintf(err, err_len, "Thread count needs to be strictly positive.");
    return NULL;
  }
  if (taxon_count < 2 || pattern_count < 1) {
    snprintf(err, err_len, "Need at least 2 taxa and 1 pattern.");
    return NULL;
  }
  oracle_engine *e = (oracle_engine *)calloc(1, sizeof(*e));
  e->spec = spec;
  e->thread_count = thread_count;
  e->n = taxon_count;
  e->P = pattern_count;
  size_t np = (size_t)taxon_count * pattern_count;
  e->patterns = (int *)malloc(sizeof(int) * np);
  memcpy(e->patterns, patterns, sizeof(int) * np);
  e->weights = (double *)malloc(sizeof(double) * pattern_count);
  memcpy(e->weights, weights, sizeof(double) * pattern_count);
  e->fbs = (ofb **)calloc(thread_count, sizeof(ofb *));
  for (int i = 0; i < thread_count; i++)
    e->fbs[i] = ofb_create(&spec, taxon_count, pattern_count, e->patterns,
                           e->weights, use_tip_states);
  return e;
}

void oracle_engine_destroy(oracle_engine *e) {
  if (!e) return;
  for (int i = 0; i < e->thread_count; i++) ofb_free(e->fbs[i]);
  free(e->fbs);
  free(e->patterns);
  free(e->weights);
  free(e);
}

int oracle_engine_param_count(const oracle_engine *e) { return e->spec.param_count; }
int oracle_engine_category_count(const oracle_engine *e) {
  return e->spec.category_count;
}
const char *oracle_engine_last_error(const oracle_engine *e) { return e->err; }

int oracle_engine_block_count(const oracle_engine *e) {
  const model_spec *m = &e->spec;
  int k = 1; /* "entire" */
  if (m->freq_start >= 0) k += 3;
  if (m->weibull) k += 2;
  if (m->strict_clock) k += 2;
  return k;
}

int oracle_engine_block(const oracle_engine *e, int idx, char *name, int name_len,
                        int *start, int *len) {
  const model_spec *m = &e->spec;
  const char *names[8];
  int starts[8], lens[8], k = 0;
  if (m->strict_clock) {
    names[k] = "clock_rate", starts[k] = m->clock_start, lens[k++] = 1;
    names[k] = "entire_clock", starts[k] = m->clock_start, lens[k++] = 1;
  }
  if (m->weibull) {
    names[k] = "Weibull_shape", starts[k] = m->shape_start, lens[k++] = 1;
    names[k] = "entire_site", starts[k] = m->shape_start, lens[k++] = 1;
  }
  if (m->freq_start >= 0) {
    names[k] = "substitution_model_frequencies", starts[k] = m->freq_start,
    lens[k++] = 4;
    names[k] = "substitution_model_rates", starts[k] = m->rates_start,
    lens[k++] = m->rates_len;
    names[k] = "entire_substitution", starts[k] = m->freq_start,
    lens[k++] = 4 + m->rates_len;
  }
  names[k] = "entire", starts[k] = 0, lens[k++] = m->param_count;
  if (idx < 0 || idx >= k) return ORACLE_ERR_BAD_ARG;
  snprintf(name, name_len, "%s", names[idx]);
  *start = starts[idx];
  *len = lens[idx];
  return ORACLE_OK;
}

typedef struct {
  oracle_engine *e;
  int tree_count, rooted, node_count, rescaling, want_gradient, flags;
  double fd_delta;
  const int *parent_ids;
  const double *bl, *rates, *params;
  double *out_ll, *out_branch, *out_site, *out_subst, *out_clock;
  pthread_mutex_t mu;
  int next, rc;
} job;

typedef struct {
  job *j;
  ofb *f;
} worker_arg;

static void *worker(void *argp) {
  worker_arg *wa = (worker_arg *)argp;
  job *j = wa->j;
  ofb *f = wa->f;
  const model_spec *m = &j->e->spec;
  int N = 2 * j->e->n - 1;
  int sub_len = m->rates_len + m->freq_len;
  for (;;) {
    pthread_mutex_lock(&j->mu);
    int i = (j->rc == ORACLE_OK && j->next < j->tree_count) ? j->next++ : -1;
    pthread_mutex_unlock(&j->mu);
    if (i < 0) break;
    tree_in in = {j->rooted, j->node_count,
                  j->parent_ids + (size_t)i * (j->node_count - 1),
                  j->bl + (size_t)i * j->node_count,
                  j->rates ? j->rates + (size_t)i * (j->node_count - 1) : NULL};
    /* fat_beagle.hpp:177-178 */
    int rc = ofb_set_parameters(f, j->params + (size_t)i * m->param_count);
    f->rescaling = j->rescaling;
    if (!rc) {
      if (j->want_gradient)
        rc = fb_gradient(f, &in, j->flags, j->fd_delta, &j->out_ll[i],
                         j->out_branch + (size_t)i * N,
                         j->out_site ? j->out_site + i : NULL,
                         j->out_subst ? j->out_subst + (size_t)i * sub_len : NULL,
                         j->out_clock ? j->out_clock + i : NULL);
      else
        rc = fb_log_likelihood(f, &in, &j->out_ll[i]);
    }
    if (rc) {
      pthread_mutex_lock(&j->mu);
      if (j->rc == ORACLE_OK) {
        j->rc = rc;
        snprintf(j->e->err, sizeof(j->e->err), "%s", f->err);
      }
      pthread_mutex_unlock(&j->mu);
    }
  }
  return NULL;
}

static int run_job(job *j) {
  oracle_engine *e = j->e;
  pthread_mutex_init(&j->mu, NULL);
  j->next = 0;
  j->rc = ORACLE_OK;
  e->err[0] = 0;
  int nt = e->thread_count;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
  worker_arg *wa = (worker_arg *)malloc(sizeof(worker_arg) * nt);
  for (int i = 0; i < nt; i++) {
    wa[i].j = j;
    wa[i].f = e->fbs[i];
    if (nt == 1)
      worker(&wa[i]);
    else
      pthread_create(&th[i], NULL, worker, &wa[i]);
  }
  if (nt > 1)
    for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
  free(th);
  free(wa);
  pthread_mutex_destroy(&j->mu);
  return j->rc;
}

int oracle_engine_log_likelihoods(oracle_engine *e, int tree_count, int rooted,
                                  int node_count, const int *parent_ids,
                                  const double *branch_lengths,
                                  const double *rates, const double *params,
                                  int rescaling, double *out) {
  job j;
  memset(&j, 0, sizeof(j));
  j.e = e, j.tree_count = tree_count, j.rooted = rooted, j.node_count = node_count;
  j.rescaling = rescaling, j.parent_ids = parent_ids, j.bl = branch_lengths;
  j.rates = rates, j.params = params, j.out_ll = out;
  return run_job(&j);
}

int oracle_engine_gradients(oracle_engine *e, int tree_count, int rooted,
                            int node_count, const int *parent_ids,
                            const double *branch_lengths, const double *rates,
                            const double *params, int rescaling, int flags,
                            double fd_delta, double *out_ll, double *out_branch,
                            double *out_site, double *out_subst,
                            double *out_clock) {
  job j;
  memset(&j, 0, sizeof(j));
  j.e = e, j.tree_count = tree_count, j.rooted = rooted, j.node_count = node_count;
  j.rescaling = rescaling, j.parent_ids = parent_ids, j.bl = branch_lengths;
  j.rates = rates, j.params = params, j.out_ll = out_ll, j.out_branch = out_branch;
  j.out_site = out_site, j.out_subst = out_subst, j.out_clock = out_clock;
  j.want_gradient = 1, j.flags = flags, j.fd_delta = fd_delta;
  return run_job(&j);
}
