/*
 * bito_amd_gp.h -- C ABI of the generalized-pruning executor (seam 3):
 * the part of bito's GPEngine that owns the PLV arena and executes a GPOperationVector
 * (reference src/gp_engine.hpp:24-141, src/gp_engine.cpp:213-339).  The subsplit DAG and the
 * schedule generator (GPDAG) stay on the caller's side; what crosses the seam is the op stream.
 *
 * JC69, one rate category, 4 states (what the reference's GPEngine supports,
 * src/gp_engine.hpp:364-377).  PLV ids follow PLVHandler: type * node_count + node with types
 * {P, PHatRight, PHatLeft, RHat, RRight, RLeft} (src/pv_handler.hpp:26-34,487-490).
 * Returns 0 or a negative BITO_AMD_ERR_* code (bito_amd.h); message via bito_amd_gp_last_error.
 */
#ifndef BITO_AMD_GP_H
#define BITO_AMD_GP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* opcode = alternative index of the reference's std::variant GPOperation (src/gp_operation.hpp:162-167) */
enum {
  BITO_AMD_GP_ZERO_PLV = 0,                            /* a = dest */
  BITO_AMD_GP_SET_TO_STATIONARY_DISTRIBUTION = 1,      /* a = dest, b = root gpcsp */
  BITO_AMD_GP_INCREMENT_WITH_WEIGHTED_EVOLVED_PLV = 2, /* a = dest, b = gpcsp, c = src */
  BITO_AMD_GP_MULTIPLY = 3,                            /* a = dest, b = src1, c = src2 */
  BITO_AMD_GP_LIKELIHOOD = 4,                          /* a = dest gpcsp, b = child_, c = parent_ */
  BITO_AMD_GP_OPTIMIZE_BRANCH_LENGTH = 5,              /* a = leafward_, b = rootward_, c = gpcsp_ */
  BITO_AMD_GP_UPDATE_SBN_PROBABILITIES = 6,            /* a = start, b = stop */
  BITO_AMD_GP_RESET_MARGINAL_LIKELIHOOD = 7,
  BITO_AMD_GP_INCREMENT_MARGINAL_LIKELIHOOD = 8,       /* a = stationary_times_prior, b = rootsplit, c = p */
  BITO_AMD_GP_PREP_FOR_MARGINALIZATION = 9             /* a = dest, b = offset into side[], count sources */
};

typedef struct {
  uint32_t opcode;
  uint32_t count;
  uint64_t a, b, c;
} bito_amd_gp_op;

typedef struct bito_amd_gp_engine bito_amd_gp_engine;

/* GPEngine::GPEngine + InitializePLVsWithSitePatterns (src/gp_engine.hpp:26-29, gp_engine.cpp:544-562).
 * The arena holds 6 * node_count PLVs in HBM (the reference memory-maps a file). */
int bito_amd_gp_create(int32_t device_id, int32_t taxon_count, int32_t pattern_count, const int32_t *patterns,
                       const double *weights, int32_t node_count, int32_t gpcsp_count, double rescaling_threshold,
                       bito_amd_gp_engine **out, char *err, size_t err_len);
void bito_amd_gp_destroy(bito_amd_gp_engine *e);
const char *bito_amd_gp_last_error(const bito_amd_gp_engine *e);

/* SetBranchLengths / GetBranchLengths / SBN parameters q (src/gp_engine.hpp:88-101): [gpcsp_count] */
int bito_amd_gp_set_branch_lengths(bito_amd_gp_engine *e, const double *branch_lengths);
int bito_amd_gp_get_branch_lengths(bito_amd_gp_engine *e, double *out);
int bito_amd_gp_set_sbn_parameters(bito_amd_gp_engine *e, const double *q);
int bito_amd_gp_get_sbn_parameters(bito_amd_gp_engine *e, double *out);

/* DAGBranchHandler state behind GPEngine (src/gp_engine.hpp:103-121, src/dag_branch_handler.hpp:40-60):
 * GetBranchLengthDifferences out[gpcsp_count]; SetOptimizationMethod with the values of
 * Optimization::OptimizationMethod (src/optimization.hpp:28-34): 0 Brent, 1 Brent with gradients,
 * 2 gradient ascent, 3 log-space gradient ascent, 4 Newton; SetSignificantDigitsForOptimization;
 * Reset/IncrementOptimizationCount (the first sweep skips the per-edge convergence test). */
#define BITO_AMD_GP_OPT_BRENT 0
#define BITO_AMD_GP_OPT_BRENT_WITH_GRADIENTS 1
#define BITO_AMD_GP_OPT_GRADIENT_ASCENT 2
#define BITO_AMD_GP_OPT_LOGSPACE_GRADIENT_ASCENT 3
#define BITO_AMD_GP_OPT_NEWTON 4
int bito_amd_gp_get_branch_length_differences(bito_amd_gp_engine *e, double *out);
int bito_amd_gp_set_optimization_method(bito_amd_gp_engine *e, int32_t method);
int bito_amd_gp_set_significant_digits_for_optimization(bito_amd_gp_engine *e, int32_t digits);
int bito_amd_gp_reset_optimization_count(bito_amd_gp_engine *e);
int bito_amd_gp_increment_optimization_count(bito_amd_gp_engine *e);

/* GPEngine::ProcessOperations (src/gp_engine.hpp:74).  Every op except UpdateSBNProbabilities is
 * independent across site patterns, so a run of such ops is ONE kernel (one thread per pattern
 * interprets the stream); an UpdateSBNProbabilities op ends the run.  OptimizeBranchLength
 * (GPEngine::OptimizeBranchLength, src/gp_engine.cpp:663-666 -> DAGBranchHandler::OptimizeBranchLength,
 * src/dag_branch_handler.cpp:123-300) runs the whole one-dimensional optimisation of the edge in one
 * single-workgroup launch: no host round trip per function evaluation.  The stream is executed in the order
 * bito_amd_gp_schedule_operations describes. */
int bito_amd_gp_process_operations(bito_amd_gp_engine *e, const bito_amd_gp_op *ops, int64_t op_count,
                                   const uint64_t *side, int64_t side_count);

/* The order in which bito_amd_gp_process_operations executes a stream (host arithmetic only: no device is touched, so
 * the order can be checked on a machine without a GPU -- tests replay it through the CPU checker).  The reference runs
 * the operations one after the other (src/gp_engine.cpp:213-339); here a stream is placed by the dependency graph of
 * its read and write sets: per-pattern operations in dependency levels, the OptimizeBranchLength operations of equal
 * "optimiser depth" (longest chain of optimisations among an operation's predecessors) as ONE launch of concurrent
 * workgroups -- the sequential arithmetic, bit for bit.  reorder = 0: optimisations one launch each, in stream order.
 * out_ops[op_count]: the stream in execution order; out_launch / out_level [op_count] (may be NULL): the launch an
 * operation belongs to and, in a per-pattern launch, its level (operations of one launch and level are independent);
 * out_launch_kinds [>= launch count <= op_count] (may be NULL): 0 per-pattern operations, 1 concurrent optimisations,
 * 2 UpdateSBNProbabilities. */
int bito_amd_gp_schedule_operations(const bito_amd_gp_op *ops, int64_t op_count, const uint64_t *side,
                                    int64_t side_count, int32_t reorder, bito_amd_gp_op *out_ops, int32_t *out_launch,
                                    int32_t *out_level, int32_t *out_launch_kinds, int64_t *out_launch_count);

/* Diagnostics: a record of every function evaluation the Brent optimisers make (Optimization::BrentMinimize(WithGradients),
 * src/optimization.hpp:71-331, called from DAGBranchHandler::BrentOptimization, src/dag_branch_handler.cpp:150-211).
 * _set_optimizer_trace(capacity_rows) starts (capacity_rows > 0: the buffer is cleared) or stops (0) recording;
 * _get_optimizer_trace copies out up to capacity_rows rows of 4 doubles -- (gpcsp, x = log branch length, f = negative
 * log-likelihood, kind: 0 the handler's evaluation of the current length, 1 Brent's first point, 2 a trial point, 3 the
 * gradient variant's second trial) -- and the number of evaluations made (which may exceed the capacity).  Rows of
 * edges optimised concurrently interleave; the rows of one edge are in order. */
int bito_amd_gp_set_optimizer_trace(bito_amd_gp_engine *e, int64_t capacity_rows);
int bito_amd_gp_get_optimizer_trace(bito_amd_gp_engine *e, double *rows, int64_t capacity_rows, int64_t *row_count);

/* Spare slots behind the DAG's own ids -- GPEngine::GrowSparePLVs / GrowSpareGPCSPs
 * (src/gp_engine.hpp:56-57, src/gp_engine.cpp:196-211): after the call PLV ids
 * [6 * node_count, 6 * node_count + spare_plv_count) and GPCSP ids [gpcsp_count, gpcsp_count +
 * spare_gpcsp_count) are valid in op streams (GetSparePVIndex / GetSpareGPCSPIndex).  Contents of the
 * arena are kept; counts only grow.  Spare GPCSPs start with branch length 0 and q = 0: fill them
 * with bito_amd_gp_copy_gpcsp_data. */
int bito_amd_gp_grow_spare(bito_amd_gp_engine *e, int64_t spare_plv_count, int64_t spare_gpcsp_count);

/* GPEngine::GrowPLVs(node_count, node_reindexer) + GrowGPCSPs(gpcsp_count, gpcsp_reindexer)
 * (src/gp_engine.hpp:44-51, src/gp_engine.cpp:64-193): the DAG has grown (AddNodePair).  A reindexer is a
 * permutation of [0, new count) giving old index -> new index (Reindexer::GetNewIndexByOldIndex); NULL =
 * identity.  PLV (type, node) moves to (type, reindexer[node]) with its rescaling counts, per-GPCSP data
 * (branch length, q, difference, log-likelihood row) to reindexer[gpcsp]; new nodes start zeroed, new
 * GPCSPs with the default branch length 0.1 and q = 1.  Spare slots are kept in number, not in content. */
int bito_amd_gp_grow(bito_amd_gp_engine *e, int32_t new_node_count, int32_t new_gpcsp_count,
                     const int64_t *node_reindexer, const int64_t *gpcsp_reindexer);
/* GetPLV(plv_index) (src/gp_engine.hpp:145-150): out[4][pattern_count], one row per state. */
int bito_amd_gp_get_plv(bito_amd_gp_engine *e, int64_t plv, double *out);

/* The reference's view of a PLV's rescaling.  GPEngine keeps ONE count per PLV, decided from the whole-PLV maximum
 * (rescaling_counts_, RescalePLVIfNeeded, src/gp_engine.cpp:564-601, src/gp_engine.hpp:300-330); this executor keeps
 * one per (PLV, pattern).  A count is the number of divisions by the threshold that brought values into
 * [threshold, 1), and the whole-PLV decision follows the PLV's largest entry, so the reference's count is the
 * smallest per-pattern count (over patterns that are not identically zero) and its stored values are the
 * executor's times threshold^(count_p - count).
 *   _rescaling_counts:    out[count] = the reference's rescaling_counts_ for PLVs first .. first + count - 1
 *   _get_plv_as_reference: GetPLV(plv) with the values the reference would hold, out[4][pattern_count], and its count
 * (what code that copies PLVs together with their counts -- the NNI engine -- reads). */
int bito_amd_gp_rescaling_counts(bito_amd_gp_engine *e, int64_t first, int64_t count, int32_t *out);
int bito_amd_gp_get_plv_as_reference(bito_amd_gp_engine *e, int64_t plv, double *out, int32_t *out_count);

/* GPEngine::CopyGPCSPData(src, dest) (src/gp_engine.cpp:401-409) for count pairs, applied in order:
 * branch length and q of src[i] are copied to dst[i]. */
int bito_amd_gp_copy_gpcsp_data(bito_amd_gp_engine *e, const int64_t *src, const int64_t *dst, int64_t count);

/* ProcessOperations for batch_count INDEPENDENT sub-streams laid side by side on the device:
 * sub-stream b is ops[offsets[b] .. offsets[b + 1]) (offsets[0] = 0, offsets[batch_count] = op_count).
 * The caller guarantees that no sub-stream writes a PLV or GPCSP slot another one reads or writes --
 * e.g. one sub-stream per proposed NNI, each on its own spare slots (the reference runs
 * NNIEvalEngineViaGP::ComputeAdjacentNNILikelihood once per NNI, src/nni_evaluation_engine.cpp:206-461).
 * Per-pattern PLV ops, Likelihood and OptimizeBranchLength are allowed (no marginal or SBN ops, which
 * share state across sub-streams).  Without optimiser ops the grid is pattern tiles x sub-streams; with
 * them one workgroup per sub-stream interprets its whole list (an optimiser op reduces over all patterns). */
int bito_amd_gp_process_operation_batches(bito_amd_gp_engine *e, const bito_amd_gp_op *ops, int64_t op_count,
                                          const uint64_t *side, int64_t side_count, const int64_t *offsets,
                                          int64_t batch_count);

/* GetPerGPCSPLogLikelihoods(start, length) and GetBranchLengths / GetSpareBranchLengths(start, length)
 * (src/gp_engine.cpp:421-456): ranges may reach into the spare GPCSPs. */
int bito_amd_gp_per_gpcsp_log_likelihoods_range(bito_amd_gp_engine *e, int64_t first, int64_t count, double *out);
int bito_amd_gp_branch_lengths_range(bito_amd_gp_engine *e, int64_t first, int64_t count, double *out);

/* GetLogMarginalLikelihood (gp_engine.cpp:413-415): out[1];
 * GetPerGPCSPLogLikelihoods (:437-440): out[gpcsp_count];
 * GetLogLikelihoodMatrix: out[gpcsp_count][pattern_count]. */
int bito_amd_gp_log_marginal_likelihood(bito_amd_gp_engine *e, double *out);
int bito_amd_gp_per_gpcsp_log_likelihoods(bito_amd_gp_engine *e, double *out);
int bito_amd_gp_log_likelihood_matrix(bito_amd_gp_engine *e, double *out);

/* LogLikelihoodAndFirstTwoDerivatives(gpcsp, rootward_pv, leafward_pv) (gp_engine.cpp:505-542): out[3] */
int bito_amd_gp_log_likelihood_and_first_two_derivatives(bito_amd_gp_engine *e, int64_t gpcsp, int64_t rootward,
                                                         int64_t leafward, double *out);

#ifdef __cplusplus
}
#endif
#endif
