// worker.hpp -- one GPU worker of the engine: a HIP device, its streams, the compressed alignment in HBM and one
// resident batch of trees with everything the kernels derive from it.  The public C ABI (engine.cpp) owns one or
// more workers per device: a blocking call cuts its tree collection into chunks, one per worker, so that the host
// side of chunk k+1 (validation, staging) and the copies of chunk k-1 overlap the traversal of chunk k; an engine
// over several devices shards the collection across them first (reference src/engine.cpp:10-31,
// src/fat_beagle.hpp:151-184: N FatBeagle instances behind one Engine).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <limits>
#include <string>
#include <vector>

#include "../../include/bito_amd.h"
#include "cpu_pause.hpp"
#include "kernels.hpp"
#include "model.hpp"

namespace bito_amd {

struct Block {
  std::string name;
  int32_t start, len;
};

template <typename T>
struct DeviceBuffer {
  T* ptr = nullptr;
  size_t capacity = 0;  // elements
  hipError_t Reserve(size_t count) {
    if (count <= capacity) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    capacity = 0;
    hipError_t rc = hipMalloc(reinterpret_cast<void**>(&ptr), count * sizeof(T));
    if (rc == hipSuccess) capacity = count;
    return rc;
  }
  void Free() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    capacity = 0;
  }
};

// a typed window into another allocation (the input block)
template <typename T>
struct DeviceView {
  T* ptr = nullptr;
};

// page-locked host memory (hipHostMalloc): copies to and from it are asynchronous and run at PCIe speed
struct PinnedBuffer {
  void* ptr = nullptr;
  size_t capacity = 0;  // bytes
  hipError_t Reserve(size_t bytes) {
    if (bytes <= capacity) return hipSuccess;
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    capacity = 0;
    bytes += bytes / 2;  // (grown with room to spare: allocating pinned memory takes milliseconds)
    hipError_t rc = hipHostMalloc(&ptr, bytes, hipHostMallocDefault);
    if (rc == hipSuccess) capacity = bytes;
    return rc;
  }
  void Free() {
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    capacity = 0;
  }
};

struct Worker {
  ModelSpec spec{};
  std::vector<Block> blocks;
  int device = 0;
  int n = 0, P = 0, Ppad = 0;
  uint64_t arena_limit = 0;
  bool arena_auto = true;  // arena_limit is the default (a share of the free HBM), not a caller's cap
  hipStream_t stream = nullptr;
  // Set-up pipeline of the LDS kernels: the set-up kernels of pass k+1 (topology, model, matrix images,
  // step tables) run on prep_stream while earlier passes' traversals are still on `stream`; they write into the
  // next of kSets buffer sets.  Events order the two streams: a set is not rewritten before the traversal that
  // read it has finished, and a traversal does not start before its set is ready.  Three sets: the set-up of
  // pass k+1 may start when pass k-2 has finished, a whole pass before it is needed -- walk_pipe_kernel keeps
  // every CU until its queue of work is empty, so the set-up kernels mostly run in the tail of a traversal,
  // and with two sets the next traversal waited for them there.
  hipStream_t prep_stream = nullptr;
  int serial_setup = 0;  // BITO_AMD_SERIAL_SETUP (measurements): 1 = the set-up kernels run on `stream`, in front of the traversal;
                         // 2 = no set-up and no events after the first kSets passes (the buffer sets keep what they hold)
  static constexpr int kSets = 3;
  hipEvent_t ev_prep_done[kSets] = {nullptr, nullptr, nullptr}, ev_walk_done[kSets] = {nullptr, nullptr, nullptr};
  unsigned run_counter = 0;
  std::string err;
  int kernel_choice = BITO_AMD_KERNEL_AUTO;
  std::string kernel_name = "none";

  // alignment
  DeviceBuffer<uint8_t> tip_states;
  DeviceBuffer<double> weights;
  // resident batch
  bool resident = false;
  BatchDims dims{};
  bool has_rates = false;
  // the wire-format inputs of the resident batch are ONE device allocation, filled by one copy from the pinned
  // staging buffer that has the same layout: branch lengths [T][M], parameter rows [T][pc], rates [T][M-1]
  // (rooted trees with rates only), then the parent ids [T][M-1] as int32
  DeviceBuffer<double> in_block;
  DeviceView<int32_t> parent_ids;
  DeviceView<double> branch_in, rates, params;
  PinnedBuffer pin_order;  // staging of the two-class tree order (walk_pipe_kernel)
  PinnedBuffer pin_in, pin_out;  // host staging: inputs as laid out above; results [ll T][gradient T*N][site T]
  hipEvent_t ev_inputs = nullptr;   // recorded behind the copy of a batch's inputs (the set-up stream waits for it)
  hipEvent_t ev_results = nullptr;  // recorded behind the copies of a pass's results into pin_out
  // Blocking calls: the final-sums kernel writes a chunk's results straight into pin_out (pinned host memory is
  // device-accessible), and a one-thread kernel behind it stores the pass's ticket into pin_flag, which the host
  // polls: no device-to-host copy commands and no event wake-up on the way back (measured: 45 us per call).
  PinnedBuffer pin_flag;
  uint64_t ticket = 0;           // of the last pass whose results go to pin_out
  DeviceBuffer<int32_t> done_counter;  // workgroups of the final-sums kernel that have finished (it leaves 0)
  bool signalled = false;        // the last pass's final-sums kernel stores the ticket itself (no kernel behind it)
  bool results_on_host = false;  // the resident pass's results are in pin_out, not in the device buffers
  bool inputs_pending = false;      // the copy of the resident batch's inputs may still be in flight
  bool inputs_on_host = false;      // ... or has not been made: the next pass's set-up kernel reads pin_in and makes it
  // Blocking calls (engine.cpp): this worker walks ONE chunk of the call, so its set-up kernels have no earlier
  // traversal of its own to hide behind and run on `stream`, in front of the traversal -- no cross-stream events.
  // 1: the GPU is otherwise idle (set-up kernels spread out); 2: another worker's traversal is running (packed).
  int one_shot = 0;
  // (with one_shot) the streams the chunk's device slot lends it, or null: the worker's own `stream` for both
  hipStream_t lent_setup = nullptr, lent_walk = nullptr;
  hipStream_t last_walk = nullptr;  // the stream the last pass's traversal went to
  int reserve_cus = 0;  // (with one_shot) CUs the traversal leaves to the set-up kernels of the call's next chunk
  bool busy = false;       // something may still be in flight on the worker's streams
  bool prep_used = false;  // ... on the set-up stream (pipelined passes since the last synchronisation)
  int id_offset = 0;  // index of the resident block's first tree in the caller's collection (error messages)
  // the block being staged between WorkerStageBegin and WorkerStageEnd: the caller's arrays and the layout of pin_in
  struct Staging {
    bool open = false;
    int tree_count = 0, rooted = 0, node_count = 0, wait = 0;
    const int32_t* parent_ids = nullptr;
    const double *branch_lengths = nullptr, *rates = nullptr, *params = nullptr;
    size_t off_params = 0, off_rates = 0, off_pid = 0, bytes = 0;
    std::vector<double> row0;  // tree 0's parameter row, when the call's set-up may leave its model in model_cache
  } staging;
  DeviceBuffer<int32_t> children, sched, children2, sched2, children3, sched3;
  DeviceBuffer<int32_t> pipe_masks;  // packed tip masks per pattern tile (walk_pipe_kernel): a function of the alignment and the plan
  long long pipe_masks_key = -1;     // groups | tiles << 8 the masks were built for
  int hbm_fold_of_set[3] = {1, 1, 1};  // HbmFoldLevel() when a buffer set's step records were last written (walk_hbm_cat_kernel)
  DeviceBuffer<int32_t> pipe_queue;  // walk_pipe_kernel's unit queue (the kernel leaves it zeroed)
  DeviceBuffer<uint8_t> pipe_done;   // per tree: its whole-tree unit wrote the final results itself (kernels.hpp: DeviceBatch::pipe_done)
  DeviceBuffer<int32_t> pipe_tree_units;  // per tree: run-of-tiles units that have counted themselves (BITO_AMD_PIPE_LAST_UNIT=1; zero between launches)
  bool pipe_last_unit = false;       // BITO_AMD_PIPE_LAST_UNIT=1: a tree's last unit forms its final sums, no final-sums launch (round 6; off until a device has run it)
  bool pipe_direct = true;           // (BITO_AMD_PIPE_DIRECT=0 when the worker is created: everything through the final-sums kernel)
  bool small_prepare = false;  // BITO_AMD_SMALL_PREPARE=1: a small batch on an idle engine gets set-up + step tables + images as ONE launch (round 6; off until a device has run it)
  // walk_pipe_kernel in two launches: the trees of the resident batch that keep few enough vectors for four
  // pattern groups per wave (class A), and the others (class B) -- one tree with few cherries would otherwise
  // halve the groups of the whole batch
  std::vector<int32_t> tree_cherries;  // per tree of the resident batch: nodes walk_pipe_kernel keeps no vector for -- cherries and folded pitchforks (counted while it is validated)
  // Two waves per SIMD (round 4, trees of up to 28 taxa): that form keeps one image per branch, i.e. the reversible form
  // of the pre-order recursion, which a tree may take only when its shortest branch times its smallest off-diagonal
  // rate clears the bound of DESIGN.md section 3 -- decided PER TREE while the batch is staged (the trees that do not, and
  // the ones that keep too many vectors, form class B and run on the one-wave kernel with (P, P^T) pairs).
  // AUTO does not take it: measured on config 3 it is 13 % SLOWER than one wave per SIMD with four groups (4.23 against
  // 3.75 ms per 6400 trees; profiles/r4_pipe_two_waves.md has the numbers and the reasons) -- it runs when the kernel is
  // pinned (BITO_AMD_KERNEL_LDS_PIPE2) or with BITO_AMD_PIPE_TWO=1 in the environment.
  // walk_pipe_kernel folds pitchforks (a tip and a cherry under one node) into their parents' steps (round 4;
  // BITO_AMD_PIPE_FOLD=0: every pitchfork a step of its own, as before)
  int pipe_fold = 1;
  int pipe_two = 0;                    // 1: AUTO takes the two-wave form where it applies (BITO_AMD_PIPE_TWO)
  std::vector<uint8_t> tree_rev_ok;    // per tree of the resident batch: the guard holds (empty: not evaluated)
  struct PipeSplit {
    bool built = false, active = false;
    bool all_two = false;  // the whole batch runs in the two-wave form (plan_a is its plan)
    int count_a = 0, count_b = 0, slots_a = 0, groups_a = 4, layout_a = 0;
    bool flagged = false;  // class A is given by a per-tree flag array (behind the order list), not by the slot count alone
    LdsPlan plan_a{}, plan_b{};
    std::vector<int32_t> order_host;
  } pipe_split;
  std::string kernel_form;             // how the last walk_pipe_kernel pass ran (waves per SIMD, groups, classes)
  DeviceBuffer<int32_t> pipe_order;    // class A's tree ids, then class B's (then, as bytes, a flag per tree: class A)
  DeviceBuffer<int32_t> pipe_masks_a;  // packed tip masks for class A's plan
  long long pipe_masks_a_key = -1;
  DeviceBuffer<double> branch, mats, mats2, mats3, images, arena, part_ll, part_grad,
      out_grad, out_site, scale_arena, branch2, images2, branch3, images3;
  // per-tree log-likelihoods: a ring, pass k writes slot k mod kOutRing, so that a consumer on another stream
  // may still be reading a pass's values while the next passes run (WorkerResultsAsync)
  static constexpr int kOutRing = 4;
  DeviceBuffer<double> out_ll_ring[kOutRing];
  unsigned out_slot = 0;
  hipEvent_t last_pass_done = nullptr;  // recorded behind the last pass enqueued (one of ev_walk_done)
  double* cur_ll() { return out_ll_ring[out_slot % kOutRing].ptr; }
  bool site_ready = false;  // out_site holds the site-model gradient of the resident pass
  double min_rate = 1.0;    // smallest off-diagonal entry of the batch's normalised rate matrices (39 taxa and more only)
  double min_branch = 0.0;  // smallest branch length of the resident batch (known for 39 taxa and more only, else 0)
  DeviceBuffer<TreeModel> model, model2, model3;
  // the model of tree 0 of the last small blocking call whose set-up left it (DeviceBatch::model_cache), the parameter
  // row it was formed from, and whether the call now staged may copy it (every row equal to that one)
  DeviceBuffer<TreeModel> model_cache;
  std::vector<double> model_cache_row;
  bool model_cache_valid = false, model_reuse_next = false;
  DeviceBuffer<double> gs_model;  // general-state path: per-model V, V^-1, Q, lambda, pi, category rates
  DeviceBuffer<int32_t> gs_model_index;  // [T] first tree with the same parameter row
  // general-state path: the parameter rows the index and the models on the device were formed from (host copy), and
  // whether those models still stand (UploadModelIndex, RunResidentGeneral)
  std::vector<double> gs_rows;
  int gs_rows_trees = 0;
  bool gs_models_fresh = false;
  const double* gs_model_seen = nullptr;
  bool gs_index_valid = false;           // the index was built from the parameter rows that are resident now
  // time-tree transforms (row f2): staging for host inputs, scratch and results
  DeviceBuffer<int32_t> tt_parents;
  DeviceBuffer<double> tt_heights, tt_bounds, tt_ratios, tt_in, tt_work, tt_out, tt_aux;
  // host mirrors for the composed gradients
  std::vector<double> h_params;
  // timing
  bool timing = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;

  ~Worker() {
    (void)hipSetDevice(device);
    for (auto ev : ev_pool) (void)hipEventDestroy(ev);
    tip_states.Free(); weights.Free(); in_block.Free(); children.Free(); pin_in.Free(); pin_out.Free(); pin_order.Free(); pin_flag.Free(); done_counter.Free();
    if (ev_inputs) (void)hipEventDestroy(ev_inputs);
    if (ev_results) (void)hipEventDestroy(ev_results);
    branch.Free(); mats.Free(); mats2.Free(); mats3.Free(); images.Free(); arena.Free(); scale_arena.Free(); part_ll.Free();
    part_grad.Free(); out_grad.Free();
    for (auto& r : out_ll_ring) r.Free(); model.Free(); gs_model.Free(); gs_model_index.Free(); sched.Free();
    tt_parents.Free(); tt_heights.Free(); tt_bounds.Free(); tt_ratios.Free(); tt_in.Free(); tt_work.Free();
    tt_out.Free(); tt_aux.Free();
    children2.Free(); sched2.Free(); pipe_masks.Free(); pipe_queue.Free(); pipe_done.Free(); pipe_tree_units.Free(); pipe_order.Free(); pipe_masks_a.Free(); branch2.Free(); images2.Free(); model2.Free(); out_site.Free();
    children3.Free(); sched3.Free(); branch3.Free(); images3.Free(); model3.Free();
    for (int i = 0; i < kSets; i++) {
      if (ev_prep_done[i]) (void)hipEventDestroy(ev_prep_done[i]);
      if (ev_walk_done[i]) (void)hipEventDestroy(ev_walk_done[i]);
    }
    if (prep_stream) (void)hipStreamDestroy(prep_stream);
    if (stream) (void)hipStreamDestroy(stream);
  }
};


// ---- worker.cpp ----------------------------------------------------------------------------------------
int WorkerCreate(int32_t device_id, uint64_t arena_bytes, const char* substitution, const char* site,
                 const char* clock, int32_t taxon_count, int32_t pattern_count, const int32_t* patterns,
                 const double* weights, Worker** out, std::string* err);
void WorkerDestroy(Worker* e);
const char* WorkerLastError(const Worker* e);
int32_t WorkerParamCount(const Worker* e);
int32_t WorkerCategoryCount(const Worker* e);
int32_t WorkerStateCount(const Worker* e);
int32_t WorkerBlockCount(const Worker* e);
int WorkerBlock(const Worker* e, int32_t idx, char* name, size_t name_len, int32_t* start, int32_t* len);
int WorkerStage(Worker* e, int32_t tree_count, int32_t rooted, int32_t node_count, const int32_t* parent_ids,
                const double* branch_lengths, const double* rates, const double* params, int wait);
// ... in three steps (worker.cpp): Begin and End on the calling thread, Fill for disjoint ranges of trees that cover
// the block on any host threads in between (no HIP call, no write outside the range's own rows)
struct StagePart {
  int code = 0;          // BITO_AMD_OK or the error the range's first bad tree gives
  std::string message;
  int first_tree = 0, end_tree = 0;
  int min_cherries = 1 << 30, min_unstored = 1 << 30;
  bool pitchforks_counted = false;  // tree_cherries of [first_tree, end_tree) hold cherries + folded pitchforks, not cherries alone
  double min_branch = std::numeric_limits<double>::infinity(), min_rate = std::numeric_limits<double>::infinity();
};
int WorkerStageBegin(Worker* e, int32_t tree_count, int32_t rooted, int32_t node_count, const int32_t* parent_ids,
                     const double* branch_lengths, const double* rates, const double* params, int wait);
void WorkerStageFill(Worker* e, int32_t t0, int32_t t1, StagePart* out);
int WorkerStageEnd(Worker* e, const StagePart* parts, int part_count);
int WorkerUpload(Worker* e, int32_t tree_count, int32_t rooted, int32_t node_count, const int32_t* parent_ids,
                 const double* branch_lengths, const double* rates, const double* params);
int WorkerValidate(Worker* e, int32_t tree_count, int32_t rooted, int32_t node_count, const int32_t* parent_ids,
                   const double* params);
int WorkerUpdate(Worker* e, const double* branch_lengths, const double* params);
int WorkerRun(Worker* e, int32_t want_gradient, int32_t rescaling);
// the pass with everything RunResident takes: deriv_mode 1 = the site-model pass (d r_c / d shape in place of r_c),
// want_site = also produce the site-model gradient when the traversal kernel can (Worker::site_ready says so)
int WorkerRunPass(Worker* e, int want_gradient, int rescaling, int deriv_mode, int want_site);
int WorkerSync(Worker* e);
int WorkerFetchResults(Worker* e, int want_gradient, int want_site);
int WorkerResults(Worker* e, const double** ll, const double** grad, const double** site);
bool WorkerResultsReady(Worker* e);  // the results WorkerFetchResults asked for have arrived (never blocks)
int WorkerDownloadAsync(Worker* e, double* out_ll, double* out_grad);
int WorkerResultsAsync(Worker* e, void* consumer_stream, const double** out_ll, const double** out_grad);
int WorkerDownload(Worker* e, double* out_ll, double* out_grad);
void* WorkerStream(Worker* e);
int WorkerLogLikelihoods(Worker* e, int32_t tree_count, int32_t rooted, int32_t node_count,
                         const int32_t* parent_ids, const double* branch_lengths, const double* rates,
                         const double* params, int32_t rescaling, double* out);
int WorkerGradients(Worker* e, int32_t tree_count, int32_t rooted, int32_t node_count, const int32_t* parent_ids,
                    const double* branch_lengths, const double* rates, const double* params, int32_t rescaling,
                    int32_t flags, double fd_delta, double* out_ll, double* out_branch, double* out_site,
                    double* out_subst, double* out_clock);
// the site-model gradient by a second traversal with dQ = Q d r_c / d shape (kernels that do not produce it in
// the main pass); leaves the main pass's results resident again
int WorkerSiteGradientSecondPass(Worker* e, int32_t rooted, int32_t node_count, const double* branch_lengths,
                                 const double* rates, int32_t rescaling, double* out_site);
int WorkerTimeTreesFromBranchLengths(Worker* e, int32_t tree_count, const int32_t* parent_ids,
                                     const double* branch_lengths, const double* tip_dates, double* out_node_bounds,
                                     double* out_node_heights, double* out_height_ratios);
int WorkerTimeTreesFromHeightRatios(Worker* e, int32_t tree_count, const int32_t* parent_ids,
                                    const double* node_bounds, const double* height_ratios, double* out_node_heights,
                                    double* out_branch_lengths);
int WorkerLogDetJacobian(Worker* e, int32_t tree_count, const int32_t* parent_ids, const double* node_heights,
                         const double* node_bounds, double* out);
int WorkerGradientLogDetJacobian(Worker* e, int32_t tree_count, const int32_t* parent_ids, const double* node_heights,
                                 const double* node_bounds, const double* height_ratios, double* out);
int WorkerRatioGradientOfHeightGradient(Worker* e, int32_t tree_count, const int32_t* parent_ids,
                                        const double* node_heights, const double* node_bounds,
                                        const double* height_ratios, const double* height_gradient, double* out);
int WorkerTimeTreeLogLikelihoods(Worker* e, int32_t tree_count, const int32_t* parent_ids,
                                 const double* branch_lengths, const double* rates, const double* node_heights,
                                 const double* node_bounds, const double* params, int32_t rescaling,
                                 int32_t include_log_det_jacobian, double* out);
int WorkerTimeTreeGradients(Worker* e, int32_t tree_count, const int32_t* parent_ids, const double* branch_lengths,
                            const double* rates, int32_t rate_count, const double* node_heights,
                            const double* node_bounds, const double* height_ratios, const double* params,
                            int32_t rescaling, int32_t flags, double fd_delta, double* out_ll, double* out_branch,
                            double* out_site, double* out_subst, double* out_clock, double* out_ratios);
int WorkerSetKernel(Worker* e, int32_t kernel);
int WorkerKernelTiming(Worker* e, int32_t enable);
int WorkerReadGeneralModel(Worker* e, int32_t tree, double* out, size_t capacity);
const char* WorkerKernelName(const Worker* e);
int WorkerTimeRuns(Worker* e, int32_t want_gradient, int32_t rescaling, int32_t steps, double* total_ms,
                   double* kernel_ms, int32_t* kernel_launches);
// FatBeagle::SubstitutionModelGradient (reference src/fat_beagle.cpp:412-508) as one batch of 2 x (#parameters)
// perturbed copies of every tree, evaluated by `log_likelihoods(tree_count, parent_ids, branch_lengths, rates,
// params, out)` -- one worker's blocking call, or the engine's over all of its workers.
using LogLikelihoodFn = std::function<int(int32_t, const int32_t*, const double*, const double*, const double*, double*)>;
int SubstitutionGradientsVia(const ModelSpec& m, int T, int rooted, int node_count, const int32_t* parent_ids,
                             const double* branch_lengths, const double* rates, const double* params, bool stick,
                             double delta, double* out_subst, const LogLikelihoodFn& log_likelihoods);
bool TreeFitsReversibleForm(const ModelSpec& m, int32_t rooted, int32_t node_count, const double* branch_lengths,
                            const double* rates, const double* params);
// host-only pieces the engine level shares with the worker level
void StickForward(const double* y, int K, double* x);
void StickInverse(const double* x, int K, double* y);

}  // namespace bito_amd
