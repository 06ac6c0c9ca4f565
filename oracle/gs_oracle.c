/*
 * gs_oracle.c -- CPU ORACLE for the general-state-count likelihood path (BASELINE config 5:
 * 61-state codon model).  TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/,
 * __graft_entry__.smoke() and bench/measurement scripts' cpu_baseline legs may load it.
 *
 * Plain C, FP64.  It restates, for an arbitrary state count S <= 64, the same algorithm
 * bito_oracle.c restates for S = 4, i.e. what bito's FatBeagle drives through BEAGLE:
 *   rate matrix -> eigendecomposition of the symmetrised matrix      src/substitution_model.cpp:120-187
 *   P(t r_c) = V diag(exp(lambda t r_c)) V^-1 per branch and category src/fat_beagle.cpp:315-325
 *   post-order partials (tip-state children = column look-up)        src/fat_beagle.cpp:54-62,338-353
 *   root log-likelihood                                              src/fat_beagle.cpp:63-68
 *   pre-order partials at the bottom of every branch                 src/fat_beagle.cpp:138-145,327-373
 *   edge derivatives  pre^T (r_c Q) post / pre^T post                src/fat_beagle.cpp:101-160
 *   unrooted trees detrifurcated as UnrootedTree::Detrifurcate       src/unrooted_tree.cpp:27-37
 *
 * Parity status.  The reference has NO model with more than 4 states, so the
 * codon model itself ("GY94": Goldman & Yang 1994 rate matrix with F1x4 codon frequencies, the
 * standard genetic code) is defined by this build and S = 61 is "parity unpinned" against the
 * reference.  What IS pinned: this file's state-count-generic code path run at S = 4 with the
 * reference's GTR matrix reproduces the reference's DS1 / fluA golden log-likelihoods and
 * gradients (tests/test_gs_oracle.py), so every routine below except the GY94 matrix builder is
 * checked against the reference's known answers; the GY94 builder is checked by its defining
 * properties (rows sum to zero, detailed balance, unit mean rate, kappa/omega placement) and the
 * gradients by central finite differences.
 *
 * Set-up arithmetic is written so that a GPU can reproduce it bit for bit (errors in P(t) are
 * coherent across site patterns): no implicit FMA contraction
 * (-ffp-contract=off), a fixed operation order, a round-robin Jacobi ordering whose rotations
 * within a round are independent, and P(t) accumulated as an explicit fma() chain in ascending
 * eigenvalue index.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MS 64 /* padded state count: every matrix is MS x MS, real states first */

enum { GS_GTR = 0, GS_GY94 = 1 };

typedef struct {
  int kind, S, C, weibull;
  int freq_start, rates_start, rates_len, shape_start, param_count;
} gs_spec;

/* ---- GY94 ------------------------------------------------------------------------------- */
/* Standard genetic code in TCAG order; nucleotides here are bito's A,C,G,T = 0..3
 * (src/site_pattern.cpp:16-46).  Sense codons are numbered in lexicographic ACGT order with the
 * three stop codons (TAA, TAG, TGA) removed: 61 states. */
static const char kCodeTCAG[65] = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
static const int kToTCAG[4] = {2, 1, 3, 0};

static char codon_aa(int a, int b, int c) {
  return kCodeTCAG[16 * kToTCAG[a] + 4 * kToTCAG[b] + kToTCAG[c]];
}

/* state -> (n1,n2,n3); returns the number of sense codons (61) */
int gs_codon_table(int *nuc /* [61][3] */) {
  int s = 0;
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++)
      for (int c = 0; c < 4; c++)
        if (codon_aa(a, b, c) != '*') {
          if (nuc) {
            nuc[s * 3] = a;
            nuc[s * 3 + 1] = b;
            nuc[s * 3 + 2] = c;
          }
          s++;
        }
  return s;
}

/* Codon state of three nucleotide symbols (each 0..3, anything else = ambiguous): 0..60, or 61
 * (gap / missing) for ambiguous or stop codons. */
int gs_codon_state(int a, int b, int c) {
  if (a < 0 || a > 3 || b < 0 || b > 3 || c < 0 || c > 3 || codon_aa(a, b, c) == '*') return 61;
  int s = 0;
  for (int x = 0; x < 4; x++)
    for (int y = 0; y < 4; y++)
      for (int z = 0; z < 4; z++) {
        if (x == a && y == b && z == c) return s;
        if (codon_aa(x, y, z) != '*') s++;
      }
  return 61;
}

static int is_transition(int x, int y) { return (x ^ y) == 2; } /* A<->G (0,2), C<->T (1,3) */

/* Q_ij = pi_j * (kappa if the single differing position is a transition) * (omega if the amino
 * acid changes); 0 when the codons differ at more than one position; rows sum to zero; scaled
 * to one expected substitution per unit time.  pi = F1x4 from nucleotide frequencies f. */
static void build_gy94(const double f[4], double kappa, double omega, double *Q, double *pi) {
  int nuc[61 * 3];
  const int S = gs_codon_table(nuc);
  double tot = 0;
  for (int j = 0; j < S; j++) {
    pi[j] = (f[nuc[j * 3]] * f[nuc[j * 3 + 1]]) * f[nuc[j * 3 + 2]];
    tot += pi[j];
  }
  for (int j = 0; j < S; j++) pi[j] /= tot;
  for (int i = 0; i < MS * MS; i++) Q[i] = 0.0;
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) {
      if (i == j) continue;
      int diff = 0, pos = -1;
      for (int k = 0; k < 3; k++)
        if (nuc[i * 3 + k] != nuc[j * 3 + k]) {
          diff++;
          pos = k;
        }
      if (diff != 1) continue;
      double q = pi[j];
      if (is_transition(nuc[i * 3 + pos], nuc[j * 3 + pos])) q *= kappa;
      if (codon_aa(nuc[i * 3], nuc[i * 3 + 1], nuc[i * 3 + 2]) != codon_aa(nuc[j * 3], nuc[j * 3 + 1], nuc[j * 3 + 2]))
        q *= omega;
      Q[i * MS + j] = q;
    }
  double total = 0;
  for (int i = 0; i < S; i++) {
    double row = 0;
    for (int j = 0; j < S; j++)
      if (j != i) row += Q[i * MS + j];
    Q[i * MS + i] = -row;
    total += row * pi[i];
  }
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) Q[i * MS + j] /= total;
}

/* GTRModel::UpdateQMatrix (src/substitution_model.cpp:141-166), same arithmetic as
 * bito_oracle.c:build_q, written into the padded layout. */
static void build_gtr(const double r[6], const double pi4[4], double *Q, double *pi) {
  for (int i = 0; i < MS * MS; i++) Q[i] = 0.0;
  for (int i = 0; i < 4; i++) pi[i] = pi4[i];
  int k = 0;
  for (int i = 0; i < 4; i++)
    for (int j = i + 1; j < 4; j++, k++) {
      Q[i * MS + j] = r[k] * pi[j];
      Q[j * MS + i] = r[k] * pi[i];
    }
  double total = 0;
  for (int i = 0; i < 4; i++) {
    double row = 0;
    for (int j = 0; j < 4; j++)
      if (i != j) row += Q[i * MS + j];
    Q[i * MS + i] = -row;
    total += row * pi[i];
  }
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) Q[i * MS + j] /= total;
}

/* ---- symmetric eigensolver -------------------------------------------------------------- */
/* Round r of the round-robin ordering on MS indices: MS/2 disjoint pairs. */
static void jacobi_round_pairs(int r, int *p, int *q) {
  const int m = MS;
  for (int k = 0; k < m / 2; k++) {
    int a, b;
    if (k == 0) {
      a = m - 1;
      b = r;
    } else {
      a = (r + k) % (m - 1);
      b = (r - k + (m - 1)) % (m - 1);
    }
    p[k] = a < b ? a : b;
    q[k] = a < b ? b : a;
  }
}

/* Jacobi eigenvalue iteration on the padded symmetric matrix A (destroyed); U accumulates the
 * rotations (stands in for Eigen::SelfAdjointEigenSolver, src/substitution_model.cpp:172; the
 * eigenvalue order is immaterial because only V f(Lambda) V^-1 products are formed).  Within a
 * round all rotation angles are taken from the matrix as it stands at the start of the round,
 * then all column updates (A and U), then all row updates, then the annihilated entries are set to
 * exactly zero. */
static int jacobi_padded(double *A, double *U, double *w) {
  for (int i = 0; i < MS; i++)
    for (int j = 0; j < MS; j++) U[i * MS + j] = (i == j) ? 1.0 : 0.0;
  int sweeps = 0;
  for (; sweeps < 60; sweeps++) {
    double maxoff = 0;
    for (int i = 0; i < MS; i++)
      for (int j = 0; j < MS; j++)
        if (i != j && fabs(A[i * MS + j]) > maxoff) maxoff = fabs(A[i * MS + j]);
    if (maxoff < 1e-20) break;
    for (int r = 0; r < MS - 1; r++) {
      int p[MS / 2], q[MS / 2];
      double cs[MS / 2], sn[MS / 2];
      jacobi_round_pairs(r, p, q);
      for (int k = 0; k < MS / 2; k++) {
        const double apq = A[p[k] * MS + q[k]];
        if (apq == 0.0) {
          cs[k] = 1.0;
          sn[k] = 0.0;
          continue;
        }
        const double theta = (A[q[k] * MS + q[k]] - A[p[k] * MS + p[k]]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        cs[k] = 1.0 / sqrt(t * t + 1.0);
        sn[k] = t * cs[k];
      }
      for (int k = 0; k < MS / 2; k++) {
        if (sn[k] == 0.0 && cs[k] == 1.0) continue;
        const double c = cs[k], s = sn[k];
        for (int i = 0; i < MS; i++) {
          const double aip = A[i * MS + p[k]], aiq = A[i * MS + q[k]];
          A[i * MS + p[k]] = c * aip - s * aiq;
          A[i * MS + q[k]] = s * aip + c * aiq;
          const double uip = U[i * MS + p[k]], uiq = U[i * MS + q[k]];
          U[i * MS + p[k]] = c * uip - s * uiq;
          U[i * MS + q[k]] = s * uip + c * uiq;
        }
      }
      for (int k = 0; k < MS / 2; k++) {
        if (sn[k] == 0.0 && cs[k] == 1.0) continue;
        const double c = cs[k], s = sn[k];
        for (int j = 0; j < MS; j++) {
          const double apj = A[p[k] * MS + j], aqj = A[q[k] * MS + j];
          A[p[k] * MS + j] = c * apj - s * aqj;
          A[q[k] * MS + j] = s * apj + c * aqj;
        }
      }
      for (int k = 0; k < MS / 2; k++) /* the rotation annihilates its own pair */
        A[p[k] * MS + q[k]] = A[q[k] * MS + p[k]] = 0.0;
    }
  }
  for (int i = 0; i < MS; i++) w[i] = A[i * MS + i];
  return sweeps;
}

/* DNAModel::UpdateEigendecomposition (src/substitution_model.cpp:168-182) for S states:
 * A = D^{1/2} Q D^{-1/2}, V = D^{-1/2} U, V^-1 = U^T D^{1/2}.  Padded states get pi = 1. */
static void eigen_reversible(int S, const double *Q, const double *pi, double *V, double *Vinv, double *lam) {
  double *A = (double *)malloc(sizeof(double) * MS * MS);
  double *U = (double *)malloc(sizeof(double) * MS * MS);
  double sq[MS];
  for (int i = 0; i < MS; i++) sq[i] = i < S ? sqrt(pi[i]) : 1.0;
  for (int i = 0; i < MS; i++)
    for (int j = 0; j < MS; j++) A[i * MS + j] = (j <= i) ? sq[i] * Q[i * MS + j] / sq[j] : 0.0;
  for (int i = 0; i < MS; i++)
    for (int j = i + 1; j < MS; j++) A[i * MS + j] = A[j * MS + i];
  jacobi_padded(A, U, lam);
  for (int i = 0; i < MS; i++)
    for (int j = 0; j < MS; j++) {
      V[i * MS + j] = U[i * MS + j] / sq[i];
      Vinv[i * MS + j] = U[j * MS + i] * sq[j];
    }
  free(A);
  free(U);
}

/* exp() with a fixed operation sequence (argument reduction by ln 2, degree-13 Taylor polynomial in
 * Horner form, every step an IEEE multiply or a correctly rounded fma, exact scaling by 2^k): the
 * same bits on any IEEE machine, CPU or GPU.  Accuracy about 1 ulp.  Errors in P(t) are coherent
 * across site patterns and its O(t^2) entries are ill-conditioned, so the
 * set-up arithmetic must not depend on whose libm is linked. */
static double det_exp(double x) {
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
double gs_det_exp(double x) { return det_exp(x); }

/* P = V diag(exp(lam t)) V^-1: W = V diag(e) rounded, then an fma chain over ascending k. */
static void transition_matrix(const double *V, const double *Vinv, const double *lam, double t, double *P) {
  double e[MS], W[MS];
  for (int k = 0; k < MS; k++) e[k] = det_exp(lam[k] * t);
  for (int i = 0; i < MS; i++) {
    for (int k = 0; k < MS; k++) W[k] = V[i * MS + k] * e[k];
    for (int j = 0; j < MS; j++) {
      double s = 0.0;
      for (int k = 0; k < MS; k++) s = fma(W[k], Vinv[k * MS + j], s);
      P[i * MS + j] = s;
    }
  }
}

/* WeibullSiteModel::UpdateRates (src/site_model.cpp:37-62), as bito_oracle.c */
static void weibull_rates(int C, double shape, double *rates, double *weights, double *derivs) {
  double mean = 0, dmean = 0, du[16];
  for (int i = 0; i < C; i++) {
    const double quantile = (2.0 * i + 1.0) / (2.0 * C);
    const double log_l = log(-log(1.0 - quantile)); /* depends on (i, C) only */
    rates[i] = det_exp(log_l / shape);              /* = pow(-log(1 - quantile), 1 / shape) */
    mean += rates[i];
    du[i] = -rates[i] * log_l / (shape * shape);
    dmean += du[i];
  }
  mean /= C;
  dmean /= C;
  for (int i = 0; i < C; i++) {
    derivs[i] = (du[i] * mean - rates[i] * dmean) / (mean * mean);
    rates[i] /= mean;
    weights[i] = 1.0 / C;
  }
}

/* ---- engine ------------------------------------------------------------------------------- */
typedef struct gs_engine {
  gs_spec spec;
  int n, P, threads;
  int *patterns; /* [n][P], >= S = gap */
  double *weights;
  char err[256];
} gs_engine;

static int parse_spec(const char *sub, const char *site, gs_spec *m, char *err, int err_len) {
  memset(m, 0, sizeof(*m));
  if (strcmp(sub, "GTR") == 0) {
    m->kind = GS_GTR;
    m->S = 4;
    m->rates_len = 6;
  } else if (strcmp(sub, "GY94") == 0) {
    m->kind = GS_GY94;
    m->S = 61;
    m->rates_len = 2; /* kappa, omega */
  } else {
    snprintf(err, err_len, "Substitution model not known: %s", sub);
    return -1;
  }
  if (strcmp(site, "constant") == 0) {
    m->C = 1;
  } else if (strncmp(site, "weibull", 7) == 0) {
    const char *plus = strchr(site, '+');
    m->weibull = 1;
    m->C = plus ? atoi(plus + 1) : 4;
    if (m->C < 1 || m->C > 16) {
      snprintf(err, err_len, "Site model not known: %s", site);
      return -1;
    }
  } else {
    snprintf(err, err_len, "Site model not known: %s", site);
    return -1;
  }
  /* row layout as BlockSpecification (src/block_specification.cpp:14-53): frequencies, rates, shape */
  m->freq_start = 0;
  m->rates_start = 4;
  m->shape_start = m->weibull ? 4 + m->rates_len : -1;
  m->param_count = 4 + m->rates_len + (m->weibull ? 1 : 0);
  return 0;
}

gs_engine *gs_engine_create(const char *substitution, const char *site, int thread_count, int taxon_count,
                            int pattern_count, const int *patterns, const double *weights, char *err,
                            int err_len) {
  gs_spec spec;
  if (parse_spec(substitution, site, &spec, err, err_len)) return NULL;
  gs_engine *e = (gs_engine *)calloc(1, sizeof(gs_engine));
  e->spec = spec;
  e->n = taxon_count;
  e->P = pattern_count;
  e->threads = thread_count < 1 ? 1 : thread_count;
  e->patterns = (int *)malloc(sizeof(int) * (size_t)taxon_count * pattern_count);
  memcpy(e->patterns, patterns, sizeof(int) * (size_t)taxon_count * pattern_count);
  e->weights = (double *)malloc(sizeof(double) * pattern_count);
  memcpy(e->weights, weights, sizeof(double) * pattern_count);
  return e;
}

void gs_engine_destroy(gs_engine *e) {
  if (!e) return;
  free(e->patterns);
  free(e->weights);
  free(e);
}

int gs_engine_param_count(const gs_engine *e) { return e->spec.param_count; }
int gs_engine_state_count(const gs_engine *e) { return e->spec.S; }
int gs_engine_category_count(const gs_engine *e) { return e->spec.C; }
const char *gs_engine_last_error(const gs_engine *e) { return e->err; }

/* Q, V, V^-1 (padded MS x MS row-major), lambda[MS], pi[MS] of one parameter row. */
static int model_setup(const gs_spec *m, const double *row, double *Q, double *V, double *Vinv, double *lam,
                       double *pi, char *err, int err_len) {
  const double *f = row + m->freq_start, *r = row + m->rates_start;
  if (fabs(f[0] + f[1] + f[2] + f[3] - 1.) >= 0.001) {
    snprintf(err, err_len, "%s frequencies do not sum to 1 +/- 0.001!", m->kind == GS_GTR ? "GTR" : "GY94");
    return -2;
  }
  for (int i = 0; i < MS; i++) pi[i] = 0.0;
  if (m->kind == GS_GTR) {
    double sum = 0;
    for (int i = 0; i < 6; i++) sum += r[i];
    if (fabs(sum - 1.) >= 0.001) {
      snprintf(err, err_len, "GTR rates do not sum to 1 +/- 0.001!");
      return -2;
    }
    build_gtr(r, f, Q, pi);
  } else {
    if (!(r[0] > 0) || !(r[1] > 0)) {
      snprintf(err, err_len, "GY94 kappa and omega must be positive");
      return -2;
    }
    build_gy94(f, r[0], r[1], Q, pi);
  }
  eigen_reversible(m->S, Q, pi, V, Vinv, lam);
  return 0;
}

int gs_substitution_model(const char *substitution, const double *params, double *Q, double *V, double *Vinv,
                          double *lam, double *pi) {
  gs_spec m;
  char err[128];
  if (parse_spec(substitution, "constant", &m, err, sizeof(err))) return -1;
  return model_setup(&m, params, Q, V, Vinv, lam, pi, err, sizeof(err));
}

void gs_transition_matrix(const double *V, const double *Vinv, const double *lam, double t, double *P) {
  transition_matrix(V, Vinv, lam, t, P);
}

/* One tree: log-likelihood and (if grad != NULL) branch gradient [2n-1].  deriv_mode 1 is the
 * site-model pass of FatBeagle::Gradient (src/fat_beagle.cpp:538-550): the differential matrices are
 * scaled by d r_c / d shape instead of r_c, and *out_site (if given) receives sum_b g_b t_b. */
static int tree_eval(const gs_engine *e, int rooted, int M, const int *parent_ids, const double *bl_in,
                     const double *rates, const double *row, int rescaling, double *out_ll, double *grad,
                     int deriv_mode, double *out_site, char *err, int err_len) {
  const gs_spec *m = &e->spec;
  const int S = m->S, C = m->C, n = e->n, P = e->P, N = 2 * n - 1;
  if (M != (rooted ? N : N - 1)) {
    snprintf(err, err_len, "node_count %d does not match %d taxa", M, n);
    return -3;
  }
  /* topology: children in ascending id order (Node::OfParentIdVector, src/node.cpp:511-551) */
  int *c0 = (int *)malloc(sizeof(int) * 3 * N), *c1 = c0 + N, *c2 = c0 + 2 * N;
  double *bl = (double *)calloc(N, sizeof(double));
  for (int i = 0; i < N; i++) c0[i] = c1[i] = c2[i] = -1;
  int rc = 0;
  for (int child = 0; child < M - 1 && !rc; child++) {
    const int p = parent_ids[child];
    if (p <= child || p >= M || p < n) rc = -3;
    else if (c0[p] < 0) c0[p] = child;
    else if (c1[p] < 0) c1[p] = child;
    else if (!rooted && p == M - 1 && c2[p] < 0) c2[p] = child;
    else rc = -3;
  }
  for (int i = n; i < M && !rc; i++)
    if (c0[i] < 0 || c1[i] < 0 || (!rooted && i == M - 1 && c2[i] < 0)) rc = -3;
  if (rc) {
    snprintf(err, err_len, "parent-id vector is not a valid bito topology");
    free(c0);
    free(bl);
    return rc;
  }
  for (int i = 0; i < M; i++) bl[i] = bl_in[i];
  if (!rooted) { /* UnrootedTree::Detrifurcate (src/unrooted_tree.cpp:27-37) */
    const int r = M - 1, a = c0[r], b = c1[r], c = c2[r];
    c0[r] = b;
    c1[r] = c;
    bl[r] = 0.;
    c0[r + 1] = a;
    c1[r + 1] = r;
    bl[r + 1] = 0.;
  } else if (rates) {
    for (int i = 0; i < N - 1; i++) bl[i] *= rates[i]; /* src/fat_beagle.cpp:86-90 */
  }
  /* model */
  const size_t MM = (size_t)MS * MS;
  double *Q = (double *)malloc(sizeof(double) * MM * 3), *V = Q + MM, *Vinv = Q + 2 * MM;
  double lam[MS], pi[MS], cat_rate[16], cat_w[16], cat_dr[16];
  rc = model_setup(m, row, Q, V, Vinv, lam, pi, err, err_len);
  if (rc) {
    free(Q);
    free(c0);
    free(bl);
    return rc;
  }
  if (m->weibull) {
    weibull_rates(C, row[m->shape_start], cat_rate, cat_w, cat_dr);
  } else {
    cat_rate[0] = cat_w[0] = 1.0;
    cat_dr[0] = 0.0;
  }
  /* transition matrices [branch][c] (beagleUpdateTransitionMatrices) */
  double *mats = (double *)malloc(sizeof(double) * MM * (size_t)(N - 1) * C);
  for (int b = 0; b < N - 1; b++)
    for (int c = 0; c < C; c++) transition_matrix(V, Vinv, lam, bl[b] * cat_rate[c], mats + ((size_t)b * C + c) * MM);
  /* partials: post[node] for internal nodes, pre[node] for all; layout [c][p][S] */
  const size_t plv = (size_t)C * P * S;
  double *post = (double *)calloc(plv * N, sizeof(double));
  double *cum = (double *)calloc(P, sizeof(double));
  /* message of child `ch` into category c, pattern p: M x (column look-up for tips, gap = ones) */
#define CHILD_MSG(ch, c, p, out)                                                         \
  do {                                                                                   \
    const double *Mx = mats + ((size_t)(ch)*C + (c)) * MM;                               \
    if ((ch) < n) {                                                                      \
      const int st = e->patterns[(size_t)(ch)*P + (p)];                                  \
      for (int i_ = 0; i_ < S; i_++) (out)[i_] = st >= S ? 1.0 : Mx[i_ * MS + st];       \
    } else {                                                                             \
      const double *x_ = post + plv * (ch) + ((size_t)(c)*P + (p)) * S;                  \
      for (int i_ = 0; i_ < S; i_++) {                                                   \
        double s_ = 0;                                                                   \
        for (int j_ = 0; j_ < S; j_++) s_ += Mx[i_ * MS + j_] * x_[j_];                  \
        (out)[i_] = s_;                                                                  \
      }                                                                                  \
    }                                                                                    \
  } while (0)
  double a[MS], b[MS];
  for (int node = n; node < N; node++) { /* ids are in post-order (src/node.cpp:383-402) */
    double *dest = post + plv * node;
    for (int c = 0; c < C; c++)
      for (int p = 0; p < P; p++) {
        CHILD_MSG(c0[node], c, p, a);
        CHILD_MSG(c1[node], c, p, b);
        double *d = dest + ((size_t)c * P + p) * S;
        for (int i = 0; i < S; i++) d[i] = a[i] * b[i];
      }
    if (rescaling) /* BEAGLE manual scaling: per pattern max over categories and states */
      for (int p = 0; p < P; p++) {
        double mx = 0;
        for (int c = 0; c < C; c++)
          for (int i = 0; i < S; i++)
            if (dest[((size_t)c * P + p) * S + i] > mx) mx = dest[((size_t)c * P + p) * S + i];
        if (mx == 0) mx = 1.0;
        for (int c = 0; c < C; c++)
          for (int i = 0; i < S; i++) dest[((size_t)c * P + p) * S + i] *= 1.0 / mx;
        cum[p] += log(mx);
      }
  }
  double total = 0;
  for (int p = 0; p < P; p++) {
    double site = 0;
    for (int c = 0; c < C; c++) {
      const double *x = post + plv * (N - 1) + ((size_t)c * P + p) * S;
      double s = 0;
      for (int i = 0; i < S; i++) s += pi[i] * x[i];
      site += cat_w[c] * s;
    }
    total += e->weights[p] * (log(site) + cum[p]);
  }
  *out_ll = total;
  if (grad) {
    double *pre = (double *)calloc(plv * N, sizeof(double));
    for (size_t k = 0; k < (size_t)C * P; k++)
      for (int i = 0; i < S; i++) pre[plv * (N - 1) + k * S + i] = pi[i];
    /* parents before children: descending ids; pre[child] = P_child^T (pre[parent] . msg(sister)) */
    for (int node = N - 1; node >= n; node--)
      for (int side = 0; side < 2; side++) {
        const int ch = side ? c1[node] : c0[node], sis = side ? c0[node] : c1[node];
        const double *Mc = NULL;
        for (int c = 0; c < C; c++) {
          Mc = mats + ((size_t)ch * C + c) * MM;
          for (int p = 0; p < P; p++) {
            CHILD_MSG(sis, c, p, a);
            const double *u = pre + plv * node + ((size_t)c * P + p) * S;
            for (int i = 0; i < S; i++) b[i] = u[i] * a[i];
            double *d = pre + plv * ch + ((size_t)c * P + p) * S;
            for (int j = 0; j < S; j++) {
              double s = 0;
              for (int i = 0; i < S; i++) s += Mc[i * MS + j] * b[i];
              d[j] = s;
            }
          }
        }
      }
    /* edge derivatives: sum_p w_p [sum_c w_c pre^T (r_c Q) post] / [sum_c w_c pre^T post] */
    for (int i = 0; i < N; i++) grad[i] = 0.0;
    for (int br = 0; br < N - 1; br++) {
      double g = 0;
      for (int p = 0; p < P; p++) {
        double num = 0, den = 0;
        for (int c = 0; c < C; c++) {
          const double *u = pre + plv * br + ((size_t)c * P + p) * S;
          double x[MS];
          if (br < n) {
            const int st = e->patterns[(size_t)br * P + p];
            for (int i = 0; i < S; i++) x[i] = (st >= S || st == i) ? 1.0 : 0.0;
          } else {
            for (int i = 0; i < S; i++) x[i] = post[plv * br + ((size_t)c * P + p) * S + i];
          }
          double nc = 0, dc = 0;
          for (int i = 0; i < S; i++) {
            double qx = 0;
            for (int j = 0; j < S; j++) qx += (Q[i * MS + j] * (deriv_mode ? cat_dr[c] : cat_rate[c])) * x[j];
            nc += u[i] * qx;
            dc += u[i] * x[i];
          }
          num += cat_w[c] * nc;
          den += cat_w[c] * dc;
        }
        g += e->weights[p] * (num / den);
      }
      grad[br] = g;
    }
    if (!rooted) grad[N - 2] = 0.0; /* the fixed node (src/fat_beagle.cpp:148,553) */
    if (out_site) {
      double sum = 0;
      for (int i = 0; i < N - 1; i++) sum += grad[i] * bl[i];
      *out_site = sum;
    }
    free(pre);
  }
#undef CHILD_MSG
  free(post);
  free(cum);
  free(mats);
  free(Q);
  free(c0);
  free(bl);
  return 0;
}

typedef struct {
  const gs_engine *e;
  int T, rooted, M, rescaling, next, rc;
  const int *parent_ids;
  const double *bl, *rates, *params;
  double *out_ll, *out_grad, *out_site;
  pthread_mutex_t mu;
  char err[256];
} gs_job;

static void *gs_worker(void *arg) {
  gs_job *j = (gs_job *)arg;
  const int N = 2 * j->e->n - 1;
  for (;;) {
    pthread_mutex_lock(&j->mu);
    const int t = j->next++;
    pthread_mutex_unlock(&j->mu);
    if (t >= j->T) break;
    char err[256] = "";
    double ll = 0;
    const int rc = tree_eval(j->e, j->rooted, j->M, j->parent_ids + (size_t)t * (j->M - 1), j->bl + (size_t)t * j->M,
                             j->rates ? j->rates + (size_t)t * (j->M - 1) : NULL,
                             j->params + (size_t)t * j->e->spec.param_count, j->rescaling, &ll,
                             j->out_grad ? j->out_grad + (size_t)t * N : NULL, 0, NULL, err, sizeof(err));
    j->out_ll[t] = ll;
    if (!rc && j->out_site) {
      double *g2 = (double *)malloc(sizeof(double) * N), ll2;
      tree_eval(j->e, j->rooted, j->M, j->parent_ids + (size_t)t * (j->M - 1), j->bl + (size_t)t * j->M,
                j->rates ? j->rates + (size_t)t * (j->M - 1) : NULL, j->params + (size_t)t * j->e->spec.param_count,
                j->rescaling, &ll2, g2, 1, &j->out_site[t], err, sizeof(err));
      free(g2);
    }
    if (rc) {
      pthread_mutex_lock(&j->mu);
      if (!j->rc) {
        j->rc = rc;
        snprintf(j->err, sizeof(j->err), "%s [tree %d]", err, t);
      }
      pthread_mutex_unlock(&j->mu);
    }
  }
  return NULL;
}

/* Engine::LogLikelihoods / Engine::Gradients (src/engine.cpp:58-110): out_grad and out_site_model
 * ([tree_count], the "site_model" gradient) may be NULL. */
int gs_engine_evaluate(gs_engine *e, int tree_count, int rooted, int node_count, const int *parent_ids,
                       const double *branch_lengths, const double *rates, const double *params, int rescaling,
                       double *out_ll, double *out_grad, double *out_site_model) {
  gs_job j;
  memset(&j, 0, sizeof(j));
  j.e = e;
  j.T = tree_count;
  j.rooted = rooted;
  j.M = node_count;
  j.rescaling = rescaling;
  j.parent_ids = parent_ids;
  j.bl = branch_lengths;
  j.rates = rates;
  j.params = params;
  j.out_ll = out_ll;
  j.out_grad = out_grad;
  j.out_site = out_site_model;
  pthread_mutex_init(&j.mu, NULL);
  int nt = e->threads < tree_count ? e->threads : tree_count;
  if (nt <= 1) {
    gs_worker(&j);
  } else {
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
    for (int i = 0; i < nt; i++) pthread_create(&th[i], NULL, gs_worker, &j);
    for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
    free(th);
  }
  pthread_mutex_destroy(&j.mu);
  if (j.rc) snprintf(e->err, sizeof(e->err), "%s", j.err);
  return j.rc;
}
