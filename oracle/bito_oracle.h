/*
 * bito_oracle.h -- CPU ORACLE for the bito FatBeagle/Engine likelihood path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a plain-C, FP64,
 * single-source restatement of the algorithm that bito's per-tree likelihood
 * path executes through BEAGLE (reference: src/fat_beagle.cpp, src/engine.cpp,
 * src/substitution_model.cpp, src/site_model.cpp, src/node.cpp; BEAGLE itself
 * -- beagle-dev/beagle-lib, branch hmc-clock, CMakeLists.txt:51-60 of the
 * reference -- is an un-vendored dependency, so its published CPU algorithm is
 * restated here from the call sites and the BEAGLE API semantics).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (bito_amd/) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this oracle
 * against every golden log-likelihood / gradient the reference's own tests
 * hold for this path (reference
 * src/unrooted_sbn_instance.hpp:236-365, src/rooted_sbn_instance.hpp:277-430,
 * vip/test/test_burrito.py:33-51, src/substitution_model.hpp:117-168,
 * src/site_model.hpp:83-107).
 */
#ifndef BITO_ORACLE_H
#define BITO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_OK 0
#define ORACLE_ERR_BAD_MODEL -1
#define ORACLE_ERR_BAD_PARAMS -2
#define ORACLE_ERR_BAD_TREE -3
#define ORACLE_ERR_BAD_ARG -4

/* Gradient request flags (reference: src/phylo_flags.hpp, the subset that
 * changes what the hot path computes; fat_beagle.cpp:524,538,607,613). */
#define ORACLE_GRAD_SUBSTITUTION_MODEL 1
#define ORACLE_GRAD_SITE_MODEL 2
#define ORACLE_GRAD_CLOCK_MODEL 4
/* use_stickbreaking_transform (reference default: on; fat_beagle.cpp:482-505):
 * the substitution-model output then has (rates-1)+(freqs-1) entries for GTR,
 * rates+(freqs-1) for HKY; the output row stride stays rates+freqs. */
#define ORACLE_GRAD_STICKBREAKING 8

typedef struct oracle_engine oracle_engine;

/* Engine::Engine (src/engine.cpp:10-31): `thread_count` independent
 * FatBeagle-equivalents that share the compressed alignment.
 * patterns: row-major [taxon_count][pattern_count], symbols 0..3, >=4 = gap
 * (src/site_pattern.cpp:16-46).  Returns NULL and fills err on failure. */
oracle_engine *oracle_engine_create(const char *substitution, const char *site,
                                    const char *clock, int thread_count,
                                    int use_tip_states, int taxon_count,
                                    int pattern_count, const int *patterns,
                                    const double *weights, char *err,
                                    int err_len);
void oracle_engine_destroy(oracle_engine *e);

/* BlockSpecification::ParameterCount (src/block_specification.hpp:62). */
int oracle_engine_param_count(const oracle_engine *e);
int oracle_engine_category_count(const oracle_engine *e);
/* Block layout: name -> (start,len); idx in [0,oracle_engine_block_count). */
int oracle_engine_block_count(const oracle_engine *e);
int oracle_engine_block(const oracle_engine *e, int idx, char *name,
                        int name_len, int *start, int *len);
const char *oracle_engine_last_error(const oracle_engine *e);

/*
 * Engine::LogLikelihoods (src/engine.cpp:58-74).
 *   rooted == 0: every tree has node_count = 2n-2 ids with a trifurcating root
 *                (UnrootedTree); detrifurcated inside (unrooted_tree.cpp:27-37).
 *   rooted == 1: node_count = 2n-1, bifurcating root; branch lengths are
 *                multiplied by rates[tree][i] (fat_beagle.cpp:83-91).  rates may
 *                be NULL (= all 1).
 * parent_ids: [tree_count][node_count-1] (Node::OfParentIdVector, node.cpp:511-551)
 * branch_lengths: [tree_count][node_count], indexed by child node id.
 * params: [tree_count][param_count] row-major (phylo_model_params_).
 */
int oracle_engine_log_likelihoods(oracle_engine *e, int tree_count, int rooted,
                                  int node_count, const int *parent_ids,
                                  const double *branch_lengths,
                                  const double *rates, const double *params,
                                  int rescaling, double *out_log_likelihoods);

/*
 * Engine::Gradients (src/engine.cpp:94-110) -> FatBeagle::Gradient
 * (fat_beagle.cpp:510-619).  out_branch_gradients: [tree_count][2n-1].
 * Optional outputs (may be NULL; honoured only when the flag bit is set):
 *   out_site_model:  [tree_count]           (fat_beagle.cpp:538-550)
 *   out_subst_model: [tree_count][rates+freqs] rates first (fat_beagle.cpp:528-531)
 *   out_clock_model: [tree_count][rate_count] strict (1) only here (fat_beagle.cpp:379-399)
 */
int oracle_engine_gradients(oracle_engine *e, int tree_count, int rooted,
                            int node_count, const int *parent_ids,
                            const double *branch_lengths, const double *rates,
                            const double *params, int rescaling, int flags,
                            double fd_delta, double *out_log_likelihoods,
                            double *out_branch_gradients, double *out_site_model,
                            double *out_subst_model, double *out_clock_model);

/* Model pieces exposed for known-answer tests. */
int oracle_substitution_model(const char *substitution, const double *params,
                              double *Q16, double *V16, double *Vinv16,
                              double *lambda4, double *pi4);
int oracle_weibull_rates(int category_count, double shape, double *rates,
                         double *proportions, double *rate_derivs);
/* P(t) = V diag(exp(lambda t)) V^-1, row-major 4x4. */
void oracle_transition_matrix(const double *V16, const double *Vinv16,
                              const double *lambda4, double t, double *P16);

/*
 * Time-tree parameterisation of RootedTree and the rooted gradient post-transforms
 * (time_tree_oracle.c; reference src/rooted_tree.cpp:36-121,
 * src/rooted_gradient_transforms.cpp:19-256).  One tree per call: parent_ids[2n-2],
 * node vectors [2n-1], internal-node vectors [n-1] (entry id-n, root last).
 */
void oracle_time_tree_bounds(int n, const int *parent_ids, const double *tip_dates, double *bounds);
int oracle_time_tree_from_branch_lengths(int n, const int *parent_ids, const double *branch_lengths,
                                         const double *tip_dates, double *bounds, double *heights,
                                         double *ratios);
void oracle_time_tree_from_height_ratios(int n, const int *parent_ids, const double *bounds,
                                         const double *ratios, double *heights, double *branch_lengths);
double oracle_log_det_jacobian(int n, const int *parent_ids, const double *heights, const double *bounds);
void oracle_height_gradient(int n, const int *parent_ids, const double *rates, const double *branch_gradient,
                            double *out);
void oracle_ratio_gradient_of_height_gradient(int n, const int *parent_ids, const double *heights,
                                              const double *bounds, const double *ratios,
                                              const double *height_gradient, double *out);
void oracle_gradient_log_det_jacobian(int n, const int *parent_ids, const double *heights, const double *bounds,
                                      const double *ratios, double *out);
void oracle_ratio_gradient_of_branch_gradient(int n, const int *parent_ids, const double *heights,
                                              const double *bounds, const double *ratios, const double *rates,
                                              const double *branch_gradient, int include_log_det_jacobian,
                                              double *out);

#ifdef __cplusplus
}
#endif
#endif
