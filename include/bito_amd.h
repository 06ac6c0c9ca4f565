/*
 * bito_amd.h -- C ABI of the MI355X-native likelihood engine that replaces
 * bito's FatBeagle/Engine path (seam 2).
 *
 * Plain C: opaque handle, pointers and sizes only.  One engine = one or more GPUs of
 * one node driven from the calling thread (bito_amd_engine_spec.device_count); one process
 * per GPU with one single-device engine each works as well (bench.py).  Every entry point below names the reference
 * interface it stands in for; bito's own host code above this line (tree
 * collections, SBN instances, pybind11 module) stays as it is -- see
 * INTEGRATION.md for the binding a bito maintainer would add.
 *
 * All calls are blocking unless stated otherwise, return 0 on success and a
 * negative BITO_AMD_ERR_* code on failure; the message is kept on the handle
 * (bito_amd_engine_last_error) so a C++ caller can rethrow it as
 * std::runtime_error exactly like Failwith (reference src/sugar.hpp:119-130).
 * An engine is used by one host thread at a time; distinct engines may run
 * concurrently (the reference's contract for FatBeagle instances,
 * src/task_processor.hpp:96-138).
 */
#ifndef BITO_AMD_H
#define BITO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BITO_AMD_OK 0
#define BITO_AMD_ERR_BAD_MODEL (-1)  /* unknown model string (substitution_model.cpp:17, site_model.cpp:24) */
#define BITO_AMD_ERR_BAD_PARAMS (-2) /* GTR/HKY sums off by >= 1e-3 (substitution_model.cpp:37-44,124-139) */
#define BITO_AMD_ERR_BAD_TREE (-3)   /* parent-id vector is not a valid bito topology */
#define BITO_AMD_ERR_BAD_ARG (-4)
#define BITO_AMD_ERR_DEVICE (-5)     /* HIP runtime failure / no gfx950 device */
#define BITO_AMD_ERR_STATE (-6)      /* call sequence error (e.g. run before upload) */

/* Gradient request bits: the PhyloGradientFlagOptions that change what the hot
 * path computes (reference src/phylo_flags.hpp:322-354, fat_beagle.cpp:524-616). */
#define BITO_AMD_GRAD_SUBSTITUTION_MODEL 1
#define BITO_AMD_GRAD_SITE_MODEL 2
#define BITO_AMD_GRAD_CLOCK_MODEL 4
#define BITO_AMD_GRAD_STICKBREAKING 8 /* use_stickbreaking_transform (reference default) */
#define BITO_AMD_GRAD_RATIOS_ROOT_HEIGHT 16        /* ratios_root_height (time trees only) */
#define BITO_AMD_GRAD_LOG_DET_JACOBIAN_GRADIENT 32 /* include_log_det_jacobian_gradient */

/* Kernel selection (diagnostics / benchmarking).  AUTO picks walk_pipe_kernel (partial-likelihood messages in
 * LDS, matrix images in registers) for trees of up to 38 taxa -- up to 64 when the batch's shortest branch times its
 * smallest off-diagonal rate (over 0.2) is 9e-7 or more -- with 1, 2 or 4 rate categories and no rescaling, the
 * HBM-arena walk otherwise. */
#define BITO_AMD_KERNEL_AUTO 0
#define BITO_AMD_KERNEL_HBM_ARENA 1 /* walk_hbm_cat_kernel (up to 4 rate categories: one wave per category), walk_hbm_kernel */
#define BITO_AMD_KERNEL_LDS 2      /* walk_lds_kernel: one wave per SIMD, images from L2 */
#define BITO_AMD_KERNEL_LDS_TREE 3 /* walk_tree_kernel: two waves per SIMD, images staged in LDS */
#define BITO_AMD_KERNEL_GENERAL 4  /* gs_walk_kernel: the general-state-count kernels (any model; the only
                                      choice for the 61-state codon model) */
#define BITO_AMD_KERNEL_LDS_PIPE 5 /* walk_pipe_kernel: walk_lds_kernel's mapping, child messages in LDS, both
                                      tree loops hand-scheduled (software-pipelined) gfx950 assembly; one wave per SIMD */
#define BITO_AMD_KERNEL_LDS_PIPE2 6 /* walk_pipe_kernel with two waves per SIMD (up to 28 taxa; trees that do not hold the
                                       reversible-form guard, or keep too many vectors, run on the one-wave form behind
                                       the others: at most a quarter of the batch) */

typedef struct bito_amd_engine bito_amd_engine;

/* Replaces EngineSpecification (reference src/engine.hpp:20-24).  The reference's one parallel axis lives inside
 * Engine -- thread_count FatBeagle instances behind one queue of trees (src/engine.cpp:10-31,
 * src/fat_beagle.hpp:151-184); here it is device_count GPUs behind one engine: every blocking call shards its
 * tree collection contiguously over the devices, drives all of them from the calling thread (streams, no extra
 * process) and gathers the results into the caller's arrays. */
typedef struct {
  int32_t device_id;      /* HIP ordinal of the first device */
  int32_t use_tip_states; /* accepted for API parity (engine.hpp:23); tips are always
                             held as compact states, which is what BEAGLE's
                             tip-state path computes (fat_beagle.cpp:269-275) */
  uint64_t arena_bytes;   /* cap on the HBM PLV arena per device; 0 = default (3/4 of free HBM: the engine owns its GPUs) */
  int32_t device_count;   /* GPUs this engine drives; 0 = default (one); negative: "Thread count needs to be strictly
                             positive." (src/engine.cpp:14-16).  Every ordinal named is checked against the
                             machine's device count when the engine is created. */
  int32_t host_threads;   /* host threads a blocking call may use for its own share of the work -- checking the
                             wire-format rows, packing them into pinned memory, copying results out -- as the reference
                             gives every FatBeagle instance a thread (EngineSpecification::thread_count_,
                             src/engine.hpp:20-24).  0 = default: min(8, CPUs this process may use); 1 = the calling
                             thread alone.  The helpers sleep between calls, but keep polling (a core each) for 8 ms after a
                             large call has woken them and 1.5 ms after every job, so that a loop of calls never pays a
                             wake-up: a process that runs several engines or ranks should divide its CPUs among them
                             here (bench.py does, by LOCAL_WORLD_SIZE). */
  const int32_t *devices; /* device_count HIP ordinals, or NULL: device_id, device_id + 1, ...  A device may be named
                             more than once (each entry is served like a device of its own): that is how the
                             multi-device path is exercised on a one-GPU machine. */
} bito_amd_engine_spec;

/*
 * Engine::Engine (reference src/engine.cpp:10-31) + FatBeagle ctor
 * (src/fat_beagle.cpp:12-28): uploads the compressed alignment once.
 *   substitution: "JC69" | "HKY" | "GTR"      (src/substitution_model.cpp:6-18)
 *                 | "GY94": 61-state codon model (BASELINE config 5).  The reference has no
 *                 model with more than four states; this one is defined here: sense codons of
 *                 the standard genetic code in lexicographic A,C,G,T order (stop codons removed),
 *                 Q_ij = pi_j [kappa if transition] [omega if the amino acid changes] for codons
 *                 that differ at one position, F1x4 frequencies from the four nucleotide
 *                 frequencies, one expected substitution per unit time.  Parameter row:
 *                 substitution_model_frequencies (4) | substitution_model_rates (kappa, omega).
 *                 patterns then hold codon states 0..60, >= 61 = gap.
 *   site:         "constant" | "weibull+K"    (src/site_model.cpp:10-25)
 *   clock:        "none" | "strict"           (src/clock_model.cpp:6-15)
 *   patterns: row-major [taxon_count][pattern_count], 0..3 = ACGT, >= 4 = gap (codon model: see above)
 *             (SitePattern::GetPatterns, src/site_pattern.cpp:16-115)
 *   weights:  [pattern_count]                 (SitePattern::GetWeights)
 * On failure *out is NULL and err (if given) holds the message.
 */
int bito_amd_engine_create(const bito_amd_engine_spec *spec, const char *substitution,
                           const char *site, const char *clock, int32_t taxon_count,
                           int32_t pattern_count, const int32_t *patterns,
                           const double *weights, bito_amd_engine **out, char *err,
                           size_t err_len);

/* ~Engine / beagleFinalizeInstance (src/fat_beagle.cpp:30-36). */
void bito_amd_engine_destroy(bito_amd_engine *e);

const char *bito_amd_engine_last_error(const bito_amd_engine *e);

/* Engine::GetPhyloModelBlockSpecification (src/engine.cpp:52-56):
 * BlockSpecification::ParameterCount and GetMap (src/block_specification.hpp:60-62).
 * Row layout [substitution | site | clock], keys alphabetical inside a model. */
int32_t bito_amd_engine_param_count(const bito_amd_engine *e);
int32_t bito_amd_engine_category_count(const bito_amd_engine *e);
int32_t bito_amd_engine_state_count(const bito_amd_engine *e); /* 4, or 61 for "GY94" */
int32_t bito_amd_engine_block_count(const bito_amd_engine *e);
int32_t bito_amd_engine_device_count(const bito_amd_engine *e); /* device slots of this engine */
int bito_amd_engine_block(const bito_amd_engine *e, int32_t idx, char *name,
                          size_t name_len, int32_t *start, int32_t *len);

/*
 * Engine::LogLikelihoods (reference src/engine.cpp:58-74) ->
 * FatBeagle::LogLikelihood (src/fat_beagle.cpp:71-98).
 *   rooted == 0: UnrootedTree, node_count = 2n-2, trifurcating root; resolved
 *                on the device as UnrootedTree::Detrifurcate does
 *                (src/unrooted_tree.cpp:27-37).
 *   rooted == 1: RootedTree, node_count = 2n-1; branch lengths are multiplied by
 *                rates[tree][branch] (fat_beagle.cpp:86-90); rates may be NULL.
 *   parent_ids:     [tree_count][node_count-1]  Node::ParentIdVector (src/node.hpp:182-185)
 *   branch_lengths: [tree_count][node_count]    by child id (src/tree.cpp:16-30)
 *   params:         [tree_count][param_count]   one row per tree (fat_beagle.hpp:177)
 *   rescaling:      Engine's `rescaling` argument (BEAGLE manual scaling)
 * Host pointers.  out_log_likelihoods: [tree_count].  tree_count == 0 (an empty collection) returns
 * BITO_AMD_OK and writes nothing, as FatBeagleParallelize does over no trees; so does _gradients.
 */
int bito_amd_engine_log_likelihoods(bito_amd_engine *e, int32_t tree_count, int32_t rooted,
                                    int32_t node_count, const int32_t *parent_ids,
                                    const double *branch_lengths, const double *rates,
                                    const double *params, int32_t rescaling,
                                    double *out_log_likelihoods);

/*
 * Engine::Gradients (reference src/engine.cpp:94-110) -> FatBeagle::Gradient
 * (src/fat_beagle.cpp:510-619).  out_branch_gradients: [tree_count][2n-1]
 * (PhyloGradient "branch_lengths"; root entry 0, unrooted fixed node 0).
 * Optional outputs, honoured when non-NULL and the flag bit is set:
 *   out_site_model:  [tree_count]                 "site_model"
 *   out_subst_model: [tree_count][rates+freqs]    "substitution_model", rates first
 *                    (row stride stays rates+freqs with the stick-breaking transform,
 *                    which yields (rates-1 if GTR else rates)+(freqs-1) entries)
 *   out_clock_model: [tree_count]                 "clock_model", strict clock
 */
int bito_amd_engine_gradients(bito_amd_engine *e, int32_t tree_count, int32_t rooted,
                              int32_t node_count, const int32_t *parent_ids,
                              const double *branch_lengths, const double *rates,
                              const double *params, int32_t rescaling, int32_t flags,
                              double fd_delta, double *out_log_likelihoods,
                              double *out_branch_gradients, double *out_site_model,
                              double *out_subst_model, double *out_clock_model);

/* ---- time trees: RootedTree's height-ratio parameterisation and the rooted gradient
 * post-transforms.  A time tree is a rooted tree (node_count = 2n-1) with
 *   node_bounds   [tree_count][2n-1]  latest tip date below each node (rooted_tree.cpp:46-60)
 *   node_heights  [tree_count][2n-1]
 *   height_ratios [tree_count][n-1]   entry id-n; the root's entry is the root height
 * All per-tree recursions run on the device, one thread per tree. ---- */

/* RootedTree::SetTipDates + InitializeTimeTreeUsingBranchLengths (src/rooted_tree.cpp:36-99).
 * tip_dates: [n] (already relative to the latest tip, taxon_name_munging.cpp:46-56).
 * BITO_AMD_ERR_BAD_TREE when a tree's branch lengths disagree with the dates by more than 1e-4. */
int bito_amd_engine_time_trees_from_branch_lengths(bito_amd_engine *e, int32_t tree_count,
                                                   const int32_t *parent_ids, const double *branch_lengths,
                                                   const double *tip_dates, double *out_node_bounds,
                                                   double *out_node_heights, double *out_height_ratios);

/* RootedTree::InitializeTimeTreeUsingHeightRatios (src/rooted_tree.cpp:101-121):
 * out_node_heights [tree_count][2n-1], out_branch_lengths [tree_count][2n-1] (root entry 0). */
int bito_amd_engine_time_trees_from_height_ratios(bito_amd_engine *e, int32_t tree_count,
                                                  const int32_t *parent_ids, const double *node_bounds,
                                                  const double *height_ratios, double *out_node_heights,
                                                  double *out_branch_lengths);

/* Engine::LogDetJacobianHeightTransform (src/engine.cpp:85-92,
 * rooted_gradient_transforms.cpp:243-256): out [tree_count]. */
int bito_amd_engine_log_det_jacobian(bito_amd_engine *e, int32_t tree_count, const int32_t *parent_ids,
                                     const double *node_heights, const double *node_bounds, double *out);

/* Engine::GradientLogDeterminantJacobian (src/engine.cpp:112-119,
 * rooted_gradient_transforms.cpp:148-168): out [tree_count][n-1]. */
int bito_amd_engine_gradient_log_det_jacobian(bito_amd_engine *e, int32_t tree_count, const int32_t *parent_ids,
                                              const double *node_heights, const double *node_bounds,
                                              const double *height_ratios, double *out);

/* RootedGradientTransforms::RatioGradientOfHeightGradient (rooted_gradient_transforms.cpp:170-184;
 * pybito ratio_gradient_of_height_gradient): height_gradient, out [tree_count][n-1]. */
int bito_amd_engine_ratio_gradient_of_height_gradient(bito_amd_engine *e, int32_t tree_count,
                                                      const int32_t *parent_ids, const double *node_heights,
                                                      const double *node_bounds, const double *height_ratios,
                                                      const double *height_gradient, double *out);

/* Engine::LogLikelihoods(RootedTreeCollection) with the include_log_det_jacobian_likelihood flag
 * (src/fat_beagle.cpp:83-98): the tree log-likelihood plus, when the flag is non-zero, the
 * log-det-Jacobian of the height transform. */
int bito_amd_engine_time_tree_log_likelihoods(bito_amd_engine *e, int32_t tree_count, const int32_t *parent_ids,
                                              const double *branch_lengths, const double *rates,
                                              const double *node_heights, const double *node_bounds,
                                              const double *params, int32_t rescaling,
                                              int32_t include_log_det_jacobian, double *out_log_likelihoods);

/* Engine::Gradients(RootedTreeCollection) (src/fat_beagle.cpp:559-619) for time trees: everything
 * bito_amd_engine_gradients returns, plus
 *   out_clock_model:        [tree_count][rate_count], rate_count 1 (strict) or 2n-2 (one rate per
 *                           branch) as RootedTree::rate_count_ (fat_beagle.cpp:379-399)
 *   out_ratios_root_height: [tree_count][n-1] "ratios_root_height", with the log-det-Jacobian
 *                           gradient added when BITO_AMD_GRAD_LOG_DET_JACOBIAN_GRADIENT is set.
 * The branch gradient stays on the device between the traversal and the ratio transform. */
int bito_amd_engine_time_tree_gradients(bito_amd_engine *e, int32_t tree_count, const int32_t *parent_ids,
                                        const double *branch_lengths, const double *rates, int32_t rate_count,
                                        const double *node_heights, const double *node_bounds,
                                        const double *height_ratios, const double *params, int32_t rescaling,
                                        int32_t flags, double fd_delta, double *out_log_likelihoods,
                                        double *out_branch_gradients, double *out_site_model,
                                        double *out_subst_model, double *out_clock_model,
                                        double *out_ratios_root_height);

/* ---- HBM-resident batch interface ----------------------------------------
 * The two calls above are upload + run + download.  Callers that keep a batch
 * on the device across steps (vip's particle loop re-evaluates the same
 * topologies with new branch lengths every step, reference vip/burrito.py:84-117)
 * use the split form; bench.py times bito_amd_engine_run with inputs resident. */

/* Host -> HBM.  Same argument meaning as bito_amd_engine_log_likelihoods. */
int bito_amd_engine_upload(bito_amd_engine *e, int32_t tree_count, int32_t rooted,
                           int32_t node_count, const int32_t *parent_ids,
                           const double *branch_lengths, const double *rates,
                           const double *params);
/* Refresh only branch lengths / params of the resident batch (either may be NULL). */
int bito_amd_engine_update(bito_amd_engine *e, const double *branch_lengths,
                           const double *params);
/* Enqueue one pass of the hot path over the resident batch on the engine's HIP
 * stream (asynchronous): per-tree model setup + eigendecomposition, transition
 * matrices, partial-likelihood traversal, (gradient) pre-order pass and edge
 * derivatives, per-tree reductions.  want_gradient: 0 = LogLikelihoods, 1 = Gradients. */
int bito_amd_engine_run(bito_amd_engine *e, int32_t want_gradient, int32_t rescaling);
/* Wait for the stream; reports per-tree validation errors found on the device. */
int bito_amd_engine_sync(bito_amd_engine *e);
/* HBM -> caller.  Destinations may be host or device pointers (hipMemcpyDefault);
 * out_branch_gradients may be NULL. */
int bito_amd_engine_download(bito_amd_engine *e, double *out_log_likelihoods,
                             double *out_branch_gradients);
/* The same copies, enqueued on the engine's stream without waiting (destinations should be device or
 * pinned host memory), and the stream itself (a hipStream_t) so that a caller can order its own work
 * behind them -- record an event on it, or hand it to its framework as an external stream.  This is how a
 * multi-GPU caller reduces the summed log-likelihood over RCCL without stalling the next pass (bench.py). */
int bito_amd_engine_download_async(bito_amd_engine *e, double *out_log_likelihoods,
                                   double *out_branch_gradients);
void *bito_amd_engine_stream(bito_amd_engine *e);
/* No copy at all: makes `consumer_stream` (a hipStream_t) wait for the passes enqueued so far and hands out the
 * device addresses of the last pass's results.  Per-tree log-likelihoods live in a ring of four buffers (pass k
 * writes buffer k mod 4): the address stays valid, and its contents untouched, until three more passes have
 * been enqueued; the gradient buffer ([tree_count][2n-1]) is rewritten by the next pass that computes
 * gradients.  Nothing is enqueued on the engine's own stream, so a consumer that sums the log-likelihoods and
 * reduces them over RCCL costs the next pass nothing (bench.py with more than one rank). */
int bito_amd_engine_results_async(bito_amd_engine *e, void *consumer_stream,
                                  const double **log_likelihoods, const double **branch_gradients);

/* Diagnostics / benchmarking. */
int bito_amd_engine_set_kernel(bito_amd_engine *e, int32_t kernel);
/* How walk_pipe_kernel would walk a batch of that shape (host arithmetic only: no device is touched, so the
 * planner can be checked on a machine without a GPU).  min_cherries: fewest cherries of any tree of the batch.
 * plan[0..6] = pattern groups per wave (0: the kernel does not take this shape), patterns per workgroup, pattern
 * tiles, LDS bytes per workgroup, tiles per run-of-tiles unit, trees walked as whole-tree units, vectors a wave
 * keeps in LDS. */
int bito_amd_plan_pipe_walk(int32_t taxon_count, int32_t pattern_count, int32_t category_count,
                            int32_t tree_count, int32_t min_cherries, int32_t plan[7]);
/* Diagnostics / tests: per tree, the nodes walk_pipe_kernel keeps no LDS vector for -- cherries (internal nodes over two
 * tips, the root excepted) and, with fold != 0, the pitchforks (a tip and a cherry under one node) whose sibling is a tip
 * or a stored node, counted on the detrifurcated tree exactly as the kernel's step tables are built.  parent_ids
 * [tree_count][node_count - 1], node_count 2n-2 (unrooted) or 2n-1 (rooted); out [tree_count]. */
int bito_amd_count_unstored_nodes(int32_t taxon_count, int32_t tree_count, int32_t rooted, int32_t node_count,
                                  const int32_t *parent_ids, int32_t fold, int32_t *out);
/* Runs `steps` passes back to back with HIP events on the engine's stream.
 * total_ms: wall time of all steps; kernel_ms: summed duration of the dominant
 * (traversal) kernel only; kernel_launches: how many such launches that was. */
int bito_amd_engine_time_runs(bito_amd_engine *e, int32_t want_gradient, int32_t rescaling,
                              int32_t steps, double *total_ms, double *kernel_ms,
                              int32_t *kernel_launches);
/* Event timing of the traversal kernel inside ordinary bito_amd_engine_run calls:
 * enable, run any number of passes or blocking calls, then read the time the
 * traversal kernel was running since enabling (HIP events recorded on the launches'
 * streams around each launch; where the chunks of a blocking call overlap -- the next
 * chunk's workgroups move in as the previous chunk's leave -- the union of the spans,
 * not their sum) and the number of launches.  Reading resets the accumulation. */
int bito_amd_engine_kernel_timing(bito_amd_engine *e, int32_t enable);
int bito_amd_engine_kernel_elapsed(bito_amd_engine *e, double *kernel_ms, int32_t *kernel_launches);
/* ... and what a profiler's per-launch durations add up to for the launches of the last
 * bito_amd_engine_kernel_elapsed: the spans summed, overlaps counted in both launches. */
double bito_amd_engine_kernel_span_sum(const bito_amd_engine *e);
/* General-state kernels only: the per-tree model record the set-up kernel produced for `tree` of the
 * resident batch after a run -- V [64][64], V^-1 [64][64], Q [64][64], lambda [64], pi [64],
 * sqrt(pi) [64], then 16 category rates, 16 weights, 16 d rate / d shape (row-major, padded to 64
 * states); at most `capacity` doubles. */
int bito_amd_engine_read_general_model(bito_amd_engine *e, int32_t tree, double *out, size_t capacity);
/* Name of the traversal kernel the last run used (for matching rocprof rows). */
const char *bito_amd_engine_kernel_name(const bito_amd_engine *e);
/* how the last walk_pipe_kernel pass ran: trees, waves per SIMD and pattern groups per wave of each class ("" for the
 * other kernels); diagnostics */
const char *bito_amd_engine_kernel_form(const bito_amd_engine *e);
/* Library/device info string, e.g. "bito_amd 0.1 gfx950 256CU". */
const char *bito_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BITO_AMD_H */
