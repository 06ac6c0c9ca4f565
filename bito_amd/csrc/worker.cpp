// worker.cpp -- one GPU worker behind the C ABI of include/bito_amd.h (the ABI itself is engine.cpp).
//
// A worker owns a HIP device's streams, the compressed alignment in HBM, and one
// resident batch of trees.  It does what one FatBeagle instance does in the
// reference (src/fat_beagle.cpp:49-169,510-619) for a whole block of trees at once:
// the per-tree work (model set-up, transition matrices, traversal) is
// done by kernels, the host only validates inputs and moves buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "worker.hpp"

using namespace bito_amd;

namespace {

int Fail(Worker* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  return code;
}

#define HIP_TRY(e, call)                                                              \
  do {                                                                                \
    hipError_t rc_ = (call);                                                          \
    if (rc_ != hipSuccess)                                                            \
      return Fail(e, BITO_AMD_ERR_DEVICE,                                             \
                  std::string(#call) + " failed: " + hipGetErrorString(rc_));         \
  } while (0)

// The two streams of a pass.  Passes over a resident batch: the worker's own pair (set-up of pass k+1 on
// prep_stream beside the traversal of pass k on stream).  A blocking call's chunk (one_shot): the pair its
// device slot lends it (engine.cpp) -- every chunk's traversal on ONE stream, in order, every later chunk's copy and
// set-up kernels on ONE other stream -- or, for the slot's first chunk, everything in line on the worker's stream.
hipStream_t SetupStream(const Worker* e) {
  if (e->one_shot) return e->lent_setup ? e->lent_setup : e->stream;
  return e->serial_setup ? e->stream : e->prep_stream;
}
hipStream_t WalkStream(const Worker* e) { return (e->one_shot && e->lent_walk) ? e->lent_walk : e->stream; }

// A blocking call's chunk whose final-sums kernel is the last kernel of the pass: that kernel stores the completion
// flag itself.  (Otherwise WorkerFetchResults enqueues a one-thread kernel for it.)
ReduceDone DoneByReduce(Worker* e) {
  ReduceDone done{};
  e->signalled = false;
  if (e->results_on_host && e->done_counter.ptr != nullptr) {
    done.flag = static_cast<unsigned long long*>(e->pin_flag.ptr);
    done.ticket = ++e->ticket;
    done.counter = e->done_counter.ptr;
    e->signalled = true;
  }
  return done;
}

// PhyloModel::OfSpecification + BlockSpecification layout
// (reference src/phylo_model.cpp:6-24, src/block_specification.cpp:14-53).
int ParseSpec(const char* sub, const char* site, const char* clock, ModelSpec* m,
              std::vector<Block>* blocks, std::string* err) {
  std::memset(m, 0, sizeof(*m));
  const std::string s(sub ? sub : ""), si(site ? site : ""), cl(clock ? clock : "");
  if (s == "JC69") m->substitution = kJC69;
  else if (s == "HKY") m->substitution = kHKY;
  else if (s == "GTR") m->substitution = kGTR;
  else if (s == "GY94") m->substitution = kGY94;  // 61-state codon model: defined by this build (bito_amd.h)
  else { *err = "Substitution model not known: " + s; return BITO_AMD_ERR_BAD_MODEL; }
  if (si == "constant") {
    m->weibull = 0;
    m->category_count = 1;
  } else if (si.rfind("weibull", 0) == 0) {
    m->weibull = 1;
    m->category_count = 4;
    const auto plus = si.find('+');
    if (plus != std::string::npos) m->category_count = std::atoi(si.c_str() + plus + 1);
    if (m->category_count < 1 || m->category_count > 8) {
      *err = "Site model '" + si + "': the GPU engine supports 1..8 rate categories.";
      return BITO_AMD_ERR_BAD_MODEL;
    }
  } else { *err = "Site model not known: " + si; return BITO_AMD_ERR_BAD_MODEL; }
  if (cl == "none") m->strict_clock = 0;
  else if (cl == "strict") m->strict_clock = 1;
  else { *err = "Clock model not known: " + cl; return BITO_AMD_ERR_BAD_MODEL; }
  int at = 0;
  m->freq_start = m->rates_start = m->shape_start = m->clock_start = -1;
  blocks->clear();
  if (m->substitution != kJC69) {
    m->freq_start = at;
    at += 4;
    m->rates_start = at;
    m->rates_len = (m->substitution == kGTR) ? 6 : (m->substitution == kGY94 ? 2 : 1);
    at += m->rates_len;
    blocks->push_back({"substitution_model_frequencies", m->freq_start, 4});
    blocks->push_back({"substitution_model_rates", m->rates_start, m->rates_len});
    blocks->push_back({"entire_substitution", m->freq_start, 4 + m->rates_len});
  }
  if (m->weibull) {
    m->shape_start = at++;
    blocks->push_back({"Weibull_shape", m->shape_start, 1});
    blocks->push_back({"entire_site", m->shape_start, 1});
  }
  if (m->strict_clock) {
    m->clock_start = at++;
    blocks->push_back({"clock_rate", m->clock_start, 1});
    blocks->push_back({"entire_clock", m->clock_start, 1});
  }
  m->param_count = at;
  m->state_count = (m->substitution == kGY94) ? 61 : 4;
  for (int i = 0; i < m->category_count && i < 16; i++)
    m->weibull_log_l[i] = std::log(-std::log(1.0 - (2.0 * i + 1.0) / (2.0 * m->category_count)));
  blocks->push_back({"entire", 0, at});
  return BITO_AMD_OK;
}

// GTRModel/HKYModel::SetParameters checks (reference src/substitution_model.cpp:33-47,120-139) of trees
// [t0, t1).  Touches nothing but *msg: the host threads of a blocking call check disjoint ranges side by side.
int ValidateParamsRange(const Worker* e, int t0, int t1, const double* params, std::string* msg) {
  const ModelSpec& m = e->spec;
  if (m.substitution == kJC69) return BITO_AMD_OK;
  const char* name = m.substitution == kGTR ? "GTR" : (m.substitution == kGY94 ? "GY94" : "HKY");
  for (int t = t0; t < t1; t++) {
    const double* row = params + (size_t)t * m.param_count;
    const double* f = row + m.freq_start;
    if (std::fabs(f[0] + f[1] + f[2] + f[3] - 1.) >= 0.001) {
      char buf[256];
      std::snprintf(buf, sizeof(buf),
                    "%s frequencies do not sum to 1 +/- 0.001! frequency vector: (%g,%g,%g,%g) [tree %d]",
                    name, f[0], f[1], f[2], f[3], t + e->id_offset);
      *msg = buf;
      return BITO_AMD_ERR_BAD_PARAMS;
    }
    if (m.substitution == kGY94) {
      const double* r = row + m.rates_start;
      if (!(r[0] > 0.) || !(r[1] > 0.)) {
        char buf[256];
        std::snprintf(buf, sizeof(buf), "GY94 kappa and omega must be positive: (%g,%g) [tree %d]", r[0], r[1], t + e->id_offset);
        *msg = buf;
        return BITO_AMD_ERR_BAD_PARAMS;
      }
    }
    if (m.substitution == kGTR) {
      const double* r = row + m.rates_start;
      double sum = 0;
      for (int i = 0; i < 6; i++) sum += r[i];
      if (std::fabs(sum - 1.) >= 0.001) {
        char buf[256];
        std::snprintf(buf, sizeof(buf),
                      "GTR rates do not sum to 1 +/- 0.001! rate vector: (%g,%g,%g,%g,%g,%g) [tree %d]",
                      r[0], r[1], r[2], r[3], r[4], r[5], t + e->id_offset);
        *msg = buf;
        return BITO_AMD_ERR_BAD_PARAMS;
      }
    }
  }
  return BITO_AMD_OK;
}

int ValidateParams(Worker* e, int tree_count, const double* params) {
  std::string msg;
  const int rc = ValidateParamsRange(e, 0, tree_count, params, &msg);
  return rc ? Fail(e, rc, msg) : rc;
}

// The parent-id vector must describe a bito topology: leaves 0..n-1, internal ids
// in post-order (every parent id larger than its children), bifurcating except for
// the trifurcating root of an unrooted tree (reference src/node.cpp:383-402,511-551;
// src/unrooted_tree.cpp:46-52).
int ValidateTreeShape(Worker* e, int rooted, int node_count) {
  const int n = e->n, M = node_count;
  if (M != (rooted ? 2 * n - 1 : 2 * n - 2)) {
    char buf[200];
    std::snprintf(buf, sizeof(buf), "node_count %d does not match %d taxa for a%s tree (expected %d)",
                  M, n, rooted ? " rooted" : "n unrooted", rooted ? 2 * n - 1 : 2 * n - 2);
    return Fail(e, BITO_AMD_ERR_BAD_TREE, buf);
  }
  if (M < 3) return Fail(e, BITO_AMD_ERR_BAD_TREE, "tree too small");
  return BITO_AMD_OK;
}

// Nodes of one tree that walk_pipe_kernel keeps no vector for, counted exactly as its step tables are built
// (walk_pipe.hip, PipeSchedulePart): on the DETRIFURCATED tree (SetupTopologyCore: an unrooted tree's root keeps children 1
// and 2 of the trifurcation, a new root joins child 0 with it) the cherries -- internal nodes over two tips, the root
// excepted -- and, with `fold`, the pitchforks (a tip and a cherry under one node) whose sibling is a tip or a stored
// node; of two pitchforks under one node the one with the lower id.  The parent-id row has passed the range checks.
int UnstoredNodes(int n, int M, int rooted, const int32_t* par, bool fold, std::vector<int>* kids_buf) {
  const int N = 2 * n - 1, NI = n - 1;
  // (one scratch array: child lists [2 NI], then the parents [N]; written in full below, so no clearing pass)
  std::vector<int>& ch = *kids_buf;
  if ((int)ch.size() < 2 * NI + N + 2) ch.resize((size_t)2 * NI + N + 2);
  int* const up = ch.data() + 2 * NI;
  for (int j = 0; j < 2 * NI; j++) ch[j] = -1;
  int third = -1;
  for (int child = 0; child < M - 1; child++) {
    const int k = par[child] - n;
    if (ch[2 * k] < 0) ch[2 * k] = child;
    else if (ch[2 * k + 1] < 0) ch[2 * k + 1] = child;
    else third = child;
  }
  if (!rooted) {
    const int r = M - 1, a = ch[2 * (r - n)], bb = ch[2 * (r - n) + 1];
    ch[2 * (r - n)] = bb;
    ch[2 * (r - n) + 1] = third;
    ch[2 * (r + 1 - n)] = a;
    ch[2 * (r + 1 - n) + 1] = r;
  }
  for (int j = 0; j < NI; j++) {
    if (ch[2 * j] < 0 || ch[2 * j + 1] < 0) return 0;  // (not a bifurcating tree: the shape check reports it)
    up[ch[2 * j]] = n + j;
    up[ch[2 * j + 1]] = n + j;
  }
  int cherries = 0;
  auto cherry = [&](int c) { return c >= n && c != N - 1 && ch[2 * (c - n)] < n && ch[2 * (c - n) + 1] < n; };
  if (!fold) {
    for (int c = n; c < N - 1; c++) cherries += cherry(c);
    return cherries;
  }
  auto fork = [&](int c) {
    if (c < n || c == N - 1) return false;
    const int a = ch[2 * (c - n)], b = ch[2 * (c - n) + 1];
    return (a < n) != (b < n) && cherry(a < n ? b : a);
  };
  int unstored = 0;
  for (int c = n; c < N - 1; c++) {
    const int a = ch[2 * (c - n)], b = ch[2 * (c - n) + 1];
    if (a < n && b < n) {
      unstored++;
    } else if ((a < n) != (b < n) && cherry(a < n ? b : a)) {
      const int p = up[c];
      const int sib = ch[2 * (p - n)] == c ? ch[2 * (p - n) + 1] : ch[2 * (p - n)];
      if (sib < n || (!cherry(sib) && (!fork(sib) || c < sib))) unstored++;
    }
  }
  return unstored;
}

// Trees [t0, t1) of a block whose shape ValidateTreeShape has accepted; writes the range's rows of cherries_of, the
// range's fewest cherries and, on failure, *msg -- nothing else (see ValidateParamsRange).
// does walk_pipe_kernel fold pitchforks for this engine's shape, and would a tree with `fewest` unstored nodes NOT fit beside
// the most pattern groups a wave can carry at this size?  (Then the pitchforks are worth counting.)
static bool FoldingApplies(const Worker* e) {
  const int C = e->spec.category_count;
  return e->pipe_fold != 0 && e->spec.state_count == 4 && e->n <= 64 && (C == 1 || C == 2 || C == 4);
}
static bool CherriesAloneDoNotFit(const Worker* e, int fewest) {
  BatchDims probe{};
  probe.taxon_count = e->n;
  probe.node_count = 2 * e->n - 1;
  probe.category_count = e->spec.category_count;
  const int room = PipeMaxSlots(probe, e->n <= kPipeExactTaxa ? 4 : 2);  // (four groups per wave up to 38 taxa, two beyond)
  return PipeSlotsOfTree(probe, fewest) > room;
}

int ValidateTreesRange(const Worker* e, int t0, int t1, int rooted, int node_count, const int32_t* parent_ids,
                       int* fewest_out, int32_t* cherries_of, std::string* msg, int* fewest_unstored_out = nullptr,
                       bool* pitchforks_counted = nullptr) {
  const int n = e->n, M = node_count;
  std::vector<int> count(M), tip_children(M);
  std::vector<int> kids;  // (the detrifurcated tree, for the pitchfork count)
  int fewest = M, fewest_unstored = M;
  // (pitchforks are folded by walk_pipe_kernel alone: up to 64 taxa, four states, 1 / 2 / 4 rate categories)
  const bool fold_counts = FoldingApplies(e);
  if (pitchforks_counted) *pitchforks_counted = false;
  for (int t = t0; t < t1; t++) {
    const int32_t* par = parent_ids + (size_t)t * (M - 1);
    std::fill(count.begin(), count.end(), 0);
    std::fill(tip_children.begin(), tip_children.end(), 0);
    for (int child = 0; child < M - 1; child++) {
      const int p = par[child];
      if (p < n || p >= M || p <= child) {
        char buf[200];
        std::snprintf(buf, sizeof(buf), "tree %d: parent id %d of node %d is not a valid internal id",
                      t + e->id_offset, p, child);
        *msg = buf;
        return BITO_AMD_ERR_BAD_TREE;
      }
      count[p]++;
      if (child < n) tip_children[p]++;
    }
    // cherries of the (detrifurcated) tree: non-root internal nodes over two tips; the node that
    // re-uses an unrooted tree's old root id joins children 1 and 2 of the trifurcation
    int cherries = 0;
    for (int i = n; i < M - 1; i++) cherries += tip_children[i] == 2;
    if (!rooted) cherries += tip_children[M - 1] == 3;
    fewest = std::min(fewest, cherries);
    if (cherries_of) cherries_of[t] = cherries;
    for (int i = n; i < M; i++) {
      const int want = (!rooted && i == M - 1) ? 3 : 2;
      if (count[i] != want) {
        char buf[200];
        std::snprintf(buf, sizeof(buf), "tree %d: node %d has %d children, expected %d", t + e->id_offset, i,
                      count[i], want);
        *msg = buf;
        return BITO_AMD_ERR_BAD_TREE;
      }
    }
  }
  // What walk_pipe_kernel keeps no vector for: the cherries and, with folding, the pitchforks of the tree as it is walked.
  // The cherries alone are a safe count (fewer unstored nodes only ask for more LDS slots than the tree uses), so the
  // pitchforks are counted -- a second pass, 0.1 us per tree -- only when it matters: when the range's trees, by their
  // cherries, would NOT all fit beside the most pattern groups a wave can carry at this size (config 3 fits by its
  // cherries: its calls, the 100-tree ones included, pay nothing for the folding).
  fewest_unstored = fewest;
  // (A range of a blocking call decides by its own trees here; WorkerStageEnd, which sees the whole batch, counts the
  // pitchforks of the ranges that did not when another range's trees make the batch need them: the plan does not depend
  // on how the call was cut into ranges.)
  if (fold_counts && (cherries_of || fewest_unstored_out) && CherriesAloneDoNotFit(e, fewest)) {
    fewest_unstored = M;
    for (int t = t0; t < t1; t++) {
      const int unstored = UnstoredNodes(n, M, rooted, parent_ids + (size_t)t * (M - 1), true, &kids);
      fewest_unstored = std::min(fewest_unstored, unstored);
      if (cherries_of) cherries_of[t] = unstored;
    }
    if (pitchforks_counted) *pitchforks_counted = true;
  }
  if (fewest_out) *fewest_out = fewest;
  if (fewest_unstored_out) *fewest_unstored_out = fewest_unstored;
  return BITO_AMD_OK;
}

int ValidateTrees(Worker* e, int tree_count, int rooted, int node_count,
                  const int32_t* parent_ids, int* min_cherries = nullptr, std::vector<int32_t>* cherries_of = nullptr) {
  if (int rc = ValidateTreeShape(e, rooted, node_count)) return rc;
  std::string msg;
  const int rc = ValidateTreesRange(e, 0, tree_count, rooted, node_count, parent_ids, min_cherries,
                                    cherries_of ? cherries_of->data() : nullptr, &msg);
  return rc ? Fail(e, rc, msg) : rc;
}

// General-state path: trees whose parameter rows are bit-identical share one model record (rate
// matrix, eigensystem); index[t] = first tree carrying t's row.
// BITO_AMD_MODEL_CACHE=0: every call forms every tree's eigensystem and category rates again
static bool ModelCacheOn() {
  static const bool on = [] { const char* v = std::getenv("BITO_AMD_MODEL_CACHE"); return !(v && v[0] == '0'); }();
  return on;
}
int UploadModelIndex(Worker* e, int tree_count, const double* params) {
  const int pc = e->spec.param_count;
  // the same rows as the batch before (a loop of calls over fixed model parameters): the index on the device and the
  // models formed from it -- rate matrices, 64 x 64 eigensystems, 0.94 ms per config-5 batch -- still stand
  // (BITO_AMD_MODEL_CACHE=0: formed again on every pass, as until round 4)
  const size_t doubles = (size_t)tree_count * (size_t)pc;  // (no parameters: `params` is one placeholder, never read)
  if (ModelCacheOn() && e->gs_models_fresh && e->gs_rows.size() == doubles && e->gs_rows_trees == tree_count &&
      std::memcmp(e->gs_rows.data(), params, doubles * sizeof(double)) == 0) {
    e->gs_index_valid = true;
    return BITO_AMD_OK;
  }
  e->gs_models_fresh = false;
  e->gs_rows.assign(params, params + doubles);
  e->gs_rows_trees = tree_count;
  std::vector<int32_t> index(tree_count);
  std::unordered_map<std::string, int32_t> first;
  for (int t = 0; t < tree_count; t++) {
    std::string key(reinterpret_cast<const char*>(params + (size_t)t * pc), pc * sizeof(double));
    index[t] = first.emplace(std::move(key), t).first->second;
  }
  HIP_TRY(e, e->gs_model_index.Reserve(tree_count));
  HIP_TRY(e, hipMemcpyAsync(e->gs_model_index.ptr, index.data(), tree_count * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  e->gs_index_valid = true;
  return BITO_AMD_OK;
}

// smallest effective branch length of a batch in wire format ([T][M], the last column is the root's)
double MinBranchLength(const double* branch_lengths, const double* rates, size_t T, size_t M) {
  double m = std::numeric_limits<double>::infinity();
  for (size_t t = 0; t < T; t++)
    for (size_t i = 0; i + 1 < M; i++) {
      const double bl = branch_lengths[t * M + i] * (rates ? rates[t * (M - 1) + i] : 1.0);
      m = bl < m ? bl : m;
    }
  return m;
}

// Smallest off-diagonal entry of the normalised rate matrices of a batch's parameter rows (JC69: 1/3).  The
// one-image-per-branch form of walk_pipe_kernel needs every off-diagonal entry of every P(t) ~ t Q_ij to stand well
// above its own rounding error, so what it asks of a batch is a bound on t_min * Q_min, not on t_min alone (a large
// kappa or a rare nucleotide makes some Q_ij a hundred times smaller than the matrices the bound was measured on).
double MinOffDiagonalRate(const ModelSpec& m, const double* params, size_t T) {
  if (m.substitution == kJC69 || m.state_count != 4 || params == nullptr) return 1.0 / 3.0;
  double lowest = std::numeric_limits<double>::infinity();
  for (size_t t = 0; t < T; t++) {
    const double* row = params + t * m.param_count;
    double r[6] = {1, 1, 1, 1, 1, 1}, Q[16];
    if (m.substitution == kGTR)
      for (int i = 0; i < 6; i++) r[i] = row[m.rates_start + i];
    else
      r[1] = r[4] = row[m.rates_start];  // HKY: kappa on the two transitions
    BuildQ(r, row + m.freq_start, Q);
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++)
        if (i != j) lowest = std::min(lowest, Q[i * 4 + j]);
  }
  return lowest;
}

// Blocking calls: a later chunk of a call (a traversal is running beside it) of this many trees and more gets its
// inputs by a copy command, which needs no CU and lands during that traversal, instead of the set-up kernel reading the
// pinned buffer once it has found a CU (BITO_AMD_INPUTS_COPY_MIN; 0 = never.  scripts/gpu_host_threads_sweep.sh, 6400
// config-3 trees as chunks of 1024 + 5376: 4.19 ms per call, 4.12 with the second chunk's inputs copied.  The same for
// the results -- final sums into HBM, copied out with the completion flag stored behind the copy -- changed nothing:
// the device writes host memory at about 20 GB/s by kernel and by copy engine alike.)
int InputsCopyMin() {
  static const int v = [] {
    const char* s = std::getenv("BITO_AMD_INPUTS_COPY_MIN");
    return s ? std::atoi(s) : 2000;
  }();
  return v;
}

double PipeReversibleMinBranch() {  // (BITO_AMD_PIPE_MIN_BRANCH: measurements of that bound)
  static const double bound = [] {
    const char* v = std::getenv("BITO_AMD_PIPE_MIN_BRANCH");
    return v ? std::atof(v) : kPipeReversibleMinBranch;
  }();
  return bound;
}

}  // namespace

namespace bito_amd {
// May one tree of 39 taxa or more take walk_pipe_kernel's one-image-per-branch form?  (The engine level sorts a
// collection's trees by this before it hands them to workers: a worker decides for its whole block.)
bool TreeFitsReversibleForm(const ModelSpec& m, int32_t rooted, int32_t node_count, const double* branch_lengths,
                            const double* rates, const double* params) {
  const double t_min = MinBranchLength(branch_lengths, rooted ? rates : nullptr, 1, (size_t)node_count);
  const double q_min = MinOffDiagonalRate(m, params, 1);
  return t_min * std::min(q_min / kPipeReversibleRateScale, 1.0) >= PipeReversibleMinBranch();
}
}  // namespace bito_amd

namespace {

// Does the batch being staged want the reversible-form guard evaluated tree by tree?  (The two-wave form of
// walk_pipe_kernel: up to 28 taxa, 1 / 2 / 4 rate categories; kernels.hpp.)
bool TracksTreeGuard(const Worker* e, int rooted, const double* rates) {
  (void)rooted;
  (void)rates;
  BatchDims probe{};
  probe.taxon_count = e->n;
  probe.category_count = e->spec.category_count;
  return e->spec.state_count == 4 && PipeTwoApplies(probe) &&
         ((e->pipe_two != 0 && e->kernel_choice == BITO_AMD_KERNEL_AUTO) || e->kernel_choice == BITO_AMD_KERNEL_LDS_PIPE2);
}

// ok[t] = may tree t take the one-image-per-branch form (TreeFitsReversibleForm), for t in [t0, t1); consecutive trees
// with the same parameter row share the rate matrix's figure
void TreeGuardRange(const ModelSpec& m, const double* branch_lengths, const double* rates, const double* params, size_t t0,
                    size_t t1, size_t M, uint8_t* ok) {
  const size_t pc = (size_t)m.param_count;
  const double bound = PipeReversibleMinBranch();
  const double* last_row = nullptr;
  double q_factor = 1.0;
  for (size_t t = t0; t < t1; t++) {
    const double* row = (params && pc) ? params + t * pc : nullptr;
    if (t == t0 || (row && std::memcmp(row, last_row, pc * sizeof(double)) != 0)) {
      q_factor = std::min(MinOffDiagonalRate(m, row, 1) / kPipeReversibleRateScale, 1.0);
      last_row = row;
    }
    const double t_min = MinBranchLength(branch_lengths + t * M, rates ? rates + t * (M - 1) : nullptr, 1, M);
    ok[t] = t_min * q_factor >= bound;
  }
}

// Bytes a pass may spend on its PLV arena.  A caller's cap (bito_amd_engine_spec.arena_bytes) holds as given.  The
// default -- 3/4 of the HBM that was free when the worker was created -- is each worker's own figure, and an engine
// keeps up to eight lane workers per device slot (and may name one GPU in several slots): when the pass wants more
// than this worker holds already, the budget is also held to what the device has free NOW, so that the workers of a
// device shrink their chunks instead of running it out of memory.
size_t ArenaBudget(Worker* e, size_t wanted_bytes) {
  const size_t held = (e->arena.capacity + e->images.capacity) * sizeof(double);
  if (!e->arena_auto || wanted_bytes <= held) return (size_t)e->arena_limit;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)e->arena_limit;
  return std::max<size_t>(std::min<size_t>((size_t)e->arena_limit, held + free_b / 4 * 3), (size_t)1 << 28);
}

// behind a set-up launch: what the model cache holds from now on (kernels.hpp, DeviceBatch::model_cache)
void NoteModelCache(Worker* e, const DeviceBatch& b) {
  if (b.model_cache == nullptr || b.model_reuse != nullptr) return;  // (not written, or it holds these values already)
  e->model_cache_row = e->staging.row0;
  e->model_cache_valid = !e->staging.row0.empty();
}

DeviceBatch MakeBatch(Worker* e, int set = 0) {
  DeviceBatch b{};
  b.hbm_fold = e->hbm_fold_of_set[set];
  b.parent_ids = e->parent_ids.ptr;
  b.branch_in = e->branch_in.ptr;
  b.rates = e->has_rates ? e->rates.ptr : nullptr;
  b.params = e->params.ptr;
  if (e->inputs_on_host) {
    // a blocking call's chunk, first pass: the set-up kernel reads the pinned staging buffer (same layout as the
    // device block) and leaves the device copies
    const char* stage = static_cast<const char*>(e->pin_in.ptr);
    const char* block = reinterpret_cast<const char*>(e->in_block.ptr);
    b.copy_parent_ids = e->parent_ids.ptr;
    b.copy_branch_in = e->branch_in.ptr;
    b.copy_rates = e->has_rates ? e->rates.ptr : nullptr;
    b.copy_params = e->params.ptr;
    b.parent_ids = reinterpret_cast<const int32_t*>(stage + (reinterpret_cast<const char*>(e->parent_ids.ptr) - block));
    b.branch_in = reinterpret_cast<const double*>(stage);
    b.rates = e->has_rates ? reinterpret_cast<const double*>(stage + (reinterpret_cast<const char*>(e->rates.ptr) - block)) : nullptr;
    b.params = reinterpret_cast<const double*>(stage + (reinterpret_cast<const char*>(e->params.ptr) - block));
    // (the staging set-up kernel of a small blocking call: tree 0's model kept for the next call, or the last call's
    // copied when WorkerStageEnd found every parameter row of this call equal to the one it was formed from)
    b.model_cache = e->model_cache.ptr;
    b.model_reuse = (e->model_reuse_next && e->model_cache.ptr != nullptr) ? e->model_cache.ptr : nullptr;
  }
  b.tip_states = e->tip_states.ptr;
  b.weights = e->weights.ptr;
  b.children = (set == 0 ? e->children : set == 1 ? e->children2 : e->children3).ptr;
  b.branch = (set == 0 ? e->branch : set == 1 ? e->branch2 : e->branch3).ptr;
  b.model = (set == 0 ? e->model : set == 1 ? e->model2 : e->model3).ptr;
  b.mats = (set == 0 ? e->mats : set == 1 ? e->mats2 : e->mats3).ptr;
  b.images = (set == 0 ? e->images : set == 1 ? e->images2 : e->images3).ptr;
  b.sched = (set == 0 ? e->sched : set == 1 ? e->sched2 : e->sched3).ptr;
  b.pipe_masks = reinterpret_cast<const uint32_t*>(e->pipe_masks.ptr);
  b.pipe_queue = e->pipe_queue.ptr;
  b.pipe_done = nullptr;  // (RunResident sets it for walk_pipe_kernel passes)
  b.arena = e->arena.ptr;
  b.scale_arena = e->scale_arena.ptr;
  b.part_ll = e->part_ll.ptr;
  b.part_grad = e->part_grad.ptr;
  if (e->one_shot) {
    // a blocking call's chunk: results straight into the pinned staging buffer, [ll T][gradient T*N][site T]
    const size_t T = e->dims.tree_count, N = e->dims.node_count;
    double* out = static_cast<double*>(e->pin_out.ptr);
    b.out_ll = out;
    b.out_grad = out + T;
    b.out_site = out + T + T * N;
  } else {
    b.out_ll = e->cur_ll();
    b.out_grad = e->out_grad.ptr;
    b.out_site = e->out_site.ptr;
  }
  return b;
}

hipEvent_t NextEvent(Worker* e) {
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t ev;
    (void)hipEventCreate(&ev);
    e->ev_pool.push_back(ev);
  }
  return e->ev_pool[e->ev_used++];
}

// General-state-count path (gs_kernels.hip): the codon model, or a 4-state model when the
// general kernels are selected explicitly.  Trees are processed in chunks sized so that a chunk's
// matrix records and PLV arena fit the arena budget.
int RunResidentGeneral(Worker* e, int want_gradient, int rescaling, int deriv_mode) {
  HIP_TRY(e, hipSetDevice(e->device));
  e->site_ready = false;
  const BatchDims& d = e->dims;
  const int T = d.tree_count, S = e->spec.state_count;
  if (!e->gs_index_valid) {
    // The batch (or its parameter rows) arrived while another kernel family was selected, so no index was
    // built for it: rebuild from the resident rows rather than reuse one that belongs to an earlier batch.
    const int pc = e->spec.param_count;
    std::vector<double> rows((size_t)T * std::max(pc, 1), 0.0);
    if (pc > 0) {
      HIP_TRY(e, hipStreamSynchronize(e->stream));
      HIP_TRY(e, hipMemcpy(rows.data(), e->params.ptr, rows.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    static const double none = 0.0;
    if (int rc = UploadModelIndex(e, T, pc > 0 ? rows.data() : &none)) return rc;
  }
  const int tiles = GsTiles(d.pattern_count);
  const size_t img_per_tree = GsImageDoublesPerTree(d), arena_per_tree = GsArenaDoublesPerTree(d, tiles, want_gradient);
  HIP_TRY(e, e->part_ll.Reserve((size_t)T * tiles));
  if (want_gradient) HIP_TRY(e, e->part_grad.Reserve((size_t)T * tiles * d.node_count));
  // (Not pipelined like the LDS path: the traversal fills the register file -- two 256-VGPR waves per
  // SIMD -- so set-up kernels of the next pass cannot co-reside with it; measured +2 % for twice the
  // matrix records.  Round 4, inside one pass: the batch in 2 / 4 / 8 chunks over two image buffers, chunk k + 1's
  // image kernel on the set-up stream beside chunk k's walk -- 57.1 / 57.3 / 58.3 ms per 4096 config-5 trees against
  // 56.7 in one piece: the dispatcher gives the image kernel its CUs when a walk's workgroups leave, i.e. it takes
  // turns with the walk rather than filling its gaps.)
  // serial on `stream`, buffer set 0, trees in chunks sized to the budget
  HIP_TRY(e, hipStreamSynchronize(e->prep_stream));
  const size_t per_tree = (img_per_tree + arena_per_tree) * sizeof(double);
  size_t chunk = std::max<size_t>(1, std::min<size_t>((size_t)T, ArenaBudget(e, (size_t)T * per_tree) / per_tree));
  chunk = std::min<size_t>(chunk, 65535);
  HIP_TRY(e, e->gs_model.Reserve((size_t)T * kGsModelStride));
  HIP_TRY(e, e->images.Reserve(chunk * img_per_tree));
  HIP_TRY(e, e->arena.Reserve(chunk * arena_per_tree));
  if (want_gradient && rescaling) HIP_TRY(e, e->scale_arena.Reserve(chunk * (size_t)(d.taxon_count - 1) * tiles * 16));
  HIP_TRY(e, e->sched.Reserve((size_t)T * GsScheduleStride(d)));
  const DeviceBatch b = MakeBatch(e);
  if (e->gs_model.ptr != e->gs_model_seen) e->gs_models_fresh = false;  // (the buffer grew: its contents went with the old one)
  LaunchGsSetup(d, e->spec, b, e->gs_model_index.ptr, e->gs_model.ptr, e->stream, /*models_stand=*/e->gs_models_fresh);
  if (!e->gs_models_fresh) {
    e->gs_model_seen = e->gs_model.ptr;
    e->gs_models_fresh = ModelCacheOn() && (int)e->gs_rows_trees == T;
  }
  LaunchGsSchedule(d, b, e->stream);
  for (int t0 = 0; t0 < T; t0 += (int)chunk) {
    const int ct = std::min<int>((int)chunk, T - t0);
    LaunchGsMatrices(d, S, t0, ct, e->branch.ptr, e->gs_model_index.ptr, e->gs_model.ptr, e->images.ptr, want_gradient, deriv_mode, e->stream);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (e->timing) {
      ev0 = NextEvent(e);
      ev1 = NextEvent(e);
      HIP_TRY(e, hipEventRecord(ev0, e->stream));
    }
    LaunchGsWalk(d, S, b, e->gs_model_index.ptr, e->gs_model.ptr, t0, ct, tiles, want_gradient, rescaling, deriv_mode, e->stream);
    if (e->timing) HIP_TRY(e, hipEventRecord(ev1, e->stream));
  }
  e->kernel_name = "gs_walk_kernel";
  e->last_walk = e->stream;
  LaunchReduce(d, b, tiles, want_gradient, e->stream, 0, DoneByReduce(e));
  HIP_TRY(e, hipEventRecord(e->ev_walk_done[0], e->stream));
  e->last_pass_done = e->ev_walk_done[0];
  HIP_TRY(e, hipGetLastError());
  return BITO_AMD_OK;
}

int RunResident(Worker* e, int want_gradient, int rescaling, int deriv_mode = 0, int want_site = 0) {
  if (!e->resident) return Fail(e, BITO_AMD_ERR_STATE, "no batch is resident: call WorkerUpload first");
  HIP_TRY(e, hipSetDevice(e->device));
  e->busy = true;
  e->results_on_host = e->one_shot != 0;
  const BatchDims& d = e->dims;
  const int T = d.tree_count;
  const size_t NB = (size_t)d.node_count - 1;
  e->out_slot++;  // this pass's log-likelihoods go to the next buffer of the ring
  if (e->spec.state_count != 4 || e->kernel_choice == BITO_AMD_KERNEL_GENERAL)
    return RunResidentGeneral(e, want_gradient, rescaling, deriv_mode);
  // Kernel choice: the LDS-resident MFMA walk when the tree fits in LDS and no rescaling
  // is requested, otherwise the HBM-arena walk.
  LdsPlan plan = PlanLds(d);
  const LdsPlan pplan = PlanPipe(d);
  const TreePlan tplan = PlanTree(d);
  bool use_tree = tplan.waves > 0 && !rescaling;
  bool use_lds = plan.groups > 0 && !rescaling;
  bool use_pipe = false;
  // (39 to 64 taxa: walk_pipe_kernel's one-image-per-branch form holds to rounding only while no transition
  // matrix entry is all rounding error, i.e. no branch is shorter than 9e-7; see walk_pipe.hip)
  const double min_branch_needed = PipeReversibleMinBranch();
  // (kPipeReversibleMinBranch was measured on matrices whose smallest off-diagonal entry is about 0.2,
  // scripts/gpu_pipe_reversible_bound.py: what is held is the product)
  const bool pipe_branches_ok = d.taxon_count <= kPipeExactTaxa ||
                                e->min_branch * std::min(e->min_rate / kPipeReversibleRateScale, 1.0) >= min_branch_needed;
  switch (e->kernel_choice) {
    case BITO_AMD_KERNEL_HBM_ARENA: use_tree = use_lds = false; break;
    case BITO_AMD_KERNEL_LDS_PIPE:
    case BITO_AMD_KERNEL_LDS_PIPE2:
      use_tree = false;
      use_pipe = pplan.groups > 0 && !rescaling && pipe_branches_ok;
      if (!use_pipe) return Fail(e, BITO_AMD_ERR_STATE, "the pipelined LDS kernel was forced but cannot run this batch (needs 1, 2 or 4 rate categories, no rescaling, a tree whose stored vectors fit in 160 KB of LDS, and from 39 taxa on branch lengths of 9e-7 and more)");
      break;
    case BITO_AMD_KERNEL_LDS:
      use_tree = false;
      if (!use_lds) return Fail(e, BITO_AMD_ERR_STATE, "the LDS kernel was forced but cannot run this batch (needs 1, 2 or 4 rate categories, no rescaling, and a tree whose PLVs fit in 160 KB of LDS)");
      break;
    case BITO_AMD_KERNEL_LDS_TREE:
      if (!use_tree) return Fail(e, BITO_AMD_ERR_STATE, "the LDS tree kernel was forced but cannot run this batch (needs 1, 2 or 4 rate categories, no rescaling, and a tree whose images + PLVs fit in 160 KB of LDS)");
      break;
    default:
      // AUTO: the hand-scheduled LDS walk where it applies (up to 38 taxa: every branch's images in the AGPR
      // file), measured 1.44 ms against walk_lds_kernel's 2.00 ms per 1600 config-3 trees
      use_pipe = pplan.groups > 0 && !rescaling && pipe_branches_ok && d.taxon_count <= kPipeAutoTaxa;
      // ... except a log-likelihood-only pass with one rate category: walk_hbm_kernel never stores a partial there
      // (each node's is forwarded in registers to its parent) and runs 0.17 ms per 1600 DS1 JC69 trees
      // against 0.32 ms (walk_pipe_kernel) and 0.37 ms (walk_lds_kernel); scripts/gpu_config2.py
      if (!want_gradient && d.category_count == 1) use_pipe = false;
      // Everything else goes to the HBM-arena walk since round 2: with one wave per rate category
      // (walk_hbm_cat_kernel) it beats walk_lds_kernel wherever walk_pipe_kernel does not apply -- 1600 trees of
      // 41 / 50 / 64 taxa, 1000 patterns, four categories: 3.75 / 4.50 / 5.85 ms against 4.79 / 5.72 / 11.8 ms
      // (scripts/gpu_midsize.py) -- so walk_lds_kernel and walk_tree_kernel run only when asked for.
      use_lds = use_tree = false;
      break;
  }
  // Measured on config 3 (profiles/): walk_lds_kernel 2.1 ms, walk_tree_kernel 3.8 ms per 1600
  // trees -- the single-wave software pipeline beats two latency-bound waves per SIMD, so the
  // tree-resident variant is only used when forced or when walk_lds cannot run.
  if (e->kernel_choice != BITO_AMD_KERNEL_LDS_TREE && use_lds) use_tree = false;
  if (use_tree) use_lds = false;
  if (use_pipe) {  // same launch sequence as the LDS kernel, with its own images, tables and plan
    use_lds = true;
    plan = pplan;
  }
  // walk_pipe_kernel: when the batch as a whole cannot have four pattern groups per wave (its tree with the
  // fewest cherries keeps too many vectors) but many of its trees could, they are walked in a launch of their own
  Worker::PipeSplit& split = e->pipe_split;
  // Two waves per SIMD (up to 28 taxa): the trees that hold the reversible-form guard and keep few enough vectors for
  // two pattern groups per wave beside eight waves form class A; if that is every tree the batch is one launch of that
  // form, otherwise the rest runs on the one-wave kernel behind it.
  const bool two_wanted = use_pipe && PipeTwoApplies(d) && (int)e->tree_rev_ok.size() == T && (int)e->tree_cherries.size() == T &&
                          ((e->pipe_two != 0 && e->kernel_choice == BITO_AMD_KERNEL_AUTO) || e->kernel_choice == BITO_AMD_KERNEL_LDS_PIPE2);
  if (use_pipe && !split.built && two_wanted) {
    // (two pattern groups per wave when three quarters of the trees and more leave room for them, else one)
    for (int groups : {2, 1}) {
      const int room = PipeMaxSlots(d, groups, kPipePlanTwoWaves);
      if (room <= 0) continue;
      std::vector<int32_t> a, bb;
      int need_a = 1, need_b = 1;
      for (int t = 0; t < T; t++) {
        const int need = PipeSlotsOfTree(d, e->tree_cherries[t]);
        const bool in_a = need <= room && e->tree_rev_ok[t];
        (in_a ? a : bb).push_back(t);
        (in_a ? need_a : need_b) = std::max(in_a ? need_a : need_b, need);
      }
      if ((int)a.size() * 4 < 3 * T) continue;
      const LdsPlan pa = PlanPipeClass(d, (int)a.size(), need_a, groups, kPipePlanTwoWaves);
      if (pa.groups != groups) continue;
      if (bb.empty()) {
        split.built = true;
        split.all_two = true;
        split.plan_a = pa;
        break;
      }
      const LdsPlan pb = PlanPipeClass(d, (int)bb.size(), need_b, 0);
      if (pb.groups <= 0) continue;
      split.built = split.active = split.flagged = true;
      split.count_a = (int)a.size();
      split.count_b = (int)bb.size();
      split.slots_a = room;
      split.groups_a = pa.groups;
      split.layout_a = kPipePlanTwoWaves;
      split.plan_a = pa;
      split.plan_b = pb;
      split.order_host = a;
      split.order_host.insert(split.order_host.end(), bb.begin(), bb.end());
      // the order list, and behind it one byte per tree: class A (the step tables are built per class)
      const size_t ints = (size_t)T + ((size_t)T + 3) / 4;
      if (ints > e->pipe_order.capacity) HIP_TRY(e, hipStreamSynchronize(WalkStream(e)));
      HIP_TRY(e, e->pipe_order.Reserve(ints));
      HIP_TRY(e, e->pin_order.Reserve(ints * sizeof(int32_t)));
      std::memcpy(e->pin_order.ptr, split.order_host.data(), (size_t)T * sizeof(int32_t));
      uint8_t* flags = reinterpret_cast<uint8_t*>(static_cast<int32_t*>(e->pin_order.ptr) + T);
      std::memset(flags, 0, (size_t)T);
      for (int32_t t : a) flags[t] = 1;
      HIP_TRY(e, hipMemcpyAsync(e->pipe_order.ptr, e->pin_order.ptr, ints * sizeof(int32_t), hipMemcpyHostToDevice, SetupStream(e)));
      break;
    }
  }
  if (e->kernel_choice == BITO_AMD_KERNEL_LDS_PIPE2 && use_pipe && PipeTwoApplies(d) && (int)e->tree_rev_ok.size() != T)
    return Fail(e, BITO_AMD_ERR_STATE, "the two-wave form of the pipelined LDS kernel was selected after this batch was uploaded: the reversible-form guard is evaluated tree by tree at upload, so upload the batch again after selecting the form");
  if (e->kernel_choice == BITO_AMD_KERNEL_LDS_PIPE2 && use_pipe && !(split.all_two || (split.active && split.layout_a == kPipePlanTwoWaves)))
    return Fail(e, BITO_AMD_ERR_STATE, "the two-wave form of the pipelined LDS kernel was forced but cannot run this batch (needs up to 28 taxa, 1, 2 or 4 rate categories, no rescaling, and for at least three quarters of the trees: branch lengths that hold the reversible-form guard and stored vectors that fit beside two pattern groups per wave)");
  if (use_pipe && !split.built) {
    split.built = true;
    split.active = false;
    static const bool no_split = std::getenv("BITO_AMD_PIPE_NO_SPLIT") != nullptr;
    // class A: the most pattern groups per wave -- 4, else 2 -- that at least a quarter of the trees leave room for
    // and that the batch as a whole (its tree with the fewest cherries) does not
    int groups_a = 0, slots_a = 0;
    if (!no_split && T >= 32 && (int)e->tree_cherries.size() == T)
      for (int g : {4, 2}) {
        const int room = PipeMaxSlots(d, g);
        if (g <= plan.groups || room <= 0) continue;
        int fit = 0;
        for (int t = 0; t < T; t++) fit += PipeSlotsOfTree(d, e->tree_cherries[t]) <= room;
        if (fit * 4 >= T) {
          groups_a = g;
          slots_a = room;
          break;
        }
      }
    if (groups_a > 0) {
      std::vector<int32_t> a, bb;
      int need_a = 1, need_b = 1;
      for (int t = 0; t < T; t++) {
        const int need = PipeSlotsOfTree(d, e->tree_cherries[t]);
        (need <= slots_a ? a : bb).push_back(t);
        (need <= slots_a ? need_a : need_b) = std::max(need <= slots_a ? need_a : need_b, need);
      }
      if (!bb.empty() && (int)a.size() * 4 >= T) {
        const LdsPlan pa = PlanPipeClass(d, (int)a.size(), need_a, groups_a), pb = PlanPipeClass(d, (int)bb.size(), need_b, 0);
        if (pa.groups == groups_a && pb.groups > 0 && pb.groups < groups_a) {
          split.active = true;
          split.count_a = (int)a.size();
          split.count_b = (int)bb.size();
          split.slots_a = slots_a;
          split.groups_a = groups_a;
          split.plan_a = pa;
          split.plan_b = pb;
          split.order_host = a;
          split.order_host.insert(split.order_host.end(), bb.begin(), bb.end());
          split.flagged = true;  // (the step tables take a tree's class from the host's list, never from a count of their own)
          // (through pinned memory, in stream order: an earlier traversal that reads the list has finished by then,
          // and the staging copy is rewritten only by the next batch, which waits for the worker to be idle)
          const size_t ints = (size_t)T + ((size_t)T + 3) / 4;  // the order list, then one byte per tree: class A
          if (ints > e->pipe_order.capacity) HIP_TRY(e, hipStreamSynchronize(WalkStream(e)));
          HIP_TRY(e, e->pipe_order.Reserve(ints));
          HIP_TRY(e, e->pin_order.Reserve(ints * sizeof(int32_t)));
          std::memcpy(e->pin_order.ptr, split.order_host.data(), (size_t)T * sizeof(int32_t));
          uint8_t* flags = reinterpret_cast<uint8_t*>(static_cast<int32_t*>(e->pin_order.ptr) + T);
          std::memset(flags, 0, (size_t)T);
          for (int32_t t : a) flags[t] = 1;
          HIP_TRY(e, hipMemcpyAsync(e->pipe_order.ptr, e->pin_order.ptr, ints * sizeof(int32_t), hipMemcpyHostToDevice, SetupStream(e)));
        }
      }
    }
  }
  const bool two_classes = use_pipe && split.active;
  if (two_classes) plan = split.plan_b;  // (class B's plan is the one the shared buffers and tables are sized by)
  if (use_pipe && split.all_two) plan = split.plan_a;
  const uint8_t* class_flags = (two_classes && split.flagged) ? reinterpret_cast<const uint8_t*>(e->pipe_order.ptr + T) : nullptr;
  const int tiles = use_tree ? tplan.tiles : (use_lds ? std::max(plan.tiles, two_classes ? split.plan_a.tiles : 0) : HbmWalkTiles(d));
  HIP_TRY(e, e->part_ll.Reserve((size_t)T * tiles));
  // partial gradient rows per tree: one per tile (LDS kernels), per run of tiles (pipelined LDS kernel), per wave (HBM kernel)
  const int pipe_rows = two_classes ? std::max(split.plan_a.grad_rows, split.plan_b.grad_rows) : plan.grad_rows;
  const int grad_rows = (use_pipe && pipe_rows > 0) ? pipe_rows : (use_tree || use_lds) ? tiles : HbmWalkGradRows(d);
  if (want_gradient) HIP_TRY(e, e->part_grad.Reserve((size_t)T * grad_rows * d.node_count));
  if (use_tree || use_lds) {
    // pipelined: this pass's set-up goes to prep_stream and into buffer set (run_counter mod kSets)
    const int set = (int)(e->run_counter++ % (unsigned)Worker::kSets);
    HIP_TRY(e, (set == 0 ? e->images : set == 1 ? e->images2 : e->images3).Reserve((size_t)T * NB * kImgStride));
    if (use_lds)
      HIP_TRY(e, (set == 0 ? e->sched : set == 1 ? e->sched2 : e->sched3).Reserve(use_pipe ? PipeScheduleInts(d) : LdsScheduleInts(d)));
    bool build_masks = false;
    if (use_pipe && e->pipe_queue.capacity == 0) {
      HIP_TRY(e, e->pipe_queue.Reserve(2));
      HIP_TRY(e, hipMemsetAsync(e->pipe_queue.ptr, 0, 2 * sizeof(int32_t), SetupStream(e)));
    }
    if (use_pipe) {  // the tile masks depend on the alignment and the plan only: built once
      const long long key = (long long)plan.groups | ((long long)plan.layout << 4) | ((long long)plan.tiles << 8);
      if (e->pipe_masks_key != key) {
        HIP_TRY(e, hipStreamSynchronize(WalkStream(e)));  // (a traversal may still be reading the old ones)
        HIP_TRY(e, e->pipe_masks.Reserve(PipeMaskInts(d, plan)));
        e->pipe_masks_key = key;
        build_masks = true;
      }
    }
    bool build_masks_a = false;
    if (two_classes) {
      const long long key = (long long)split.plan_a.groups | ((long long)split.plan_a.layout << 4) | ((long long)split.plan_a.tiles << 8);
      if (e->pipe_masks_a_key != key) {
        HIP_TRY(e, hipStreamSynchronize(WalkStream(e)));
        HIP_TRY(e, e->pipe_masks_a.Reserve(PipeMaskInts(d, split.plan_a)));
        e->pipe_masks_a_key = key;
        build_masks_a = true;
      }
    }
    DeviceBatch b = MakeBatch(e, set);
    hipStream_t prep = SetupStream(e);
    const hipStream_t walk = WalkStream(e);
    const bool in_line = prep == walk;  // set-up in front of the traversal, on its stream
    e->last_walk = walk;
    const bool bare = e->serial_setup == 2 && e->run_counter > (unsigned)Worker::kSets;
    bool inputs_event_due = false;
    if (!bare) {
      if (!in_line) {
        e->prep_used = e->prep_used || !e->one_shot;  // (a lent stream is not this worker's to wait for)
        HIP_TRY(e, hipStreamWaitEvent(prep, e->ev_walk_done[set], 0));
        if (e->inputs_pending && !e->one_shot) HIP_TRY(e, hipStreamWaitEvent(prep, e->ev_inputs, 0));  // (one_shot: the copy is on `prep` itself)
      }
      // (packed into few workgroups when a traversal is still running beside it; spread out -- 30 us sooner --
      // when the engine is idle, as it is for a caller that waits for every pass)
      const bool busy = e->one_shot ? e->one_shot == 2
                                    : (!e->serial_setup && e->last_pass_done != nullptr && hipEventQuery(e->last_pass_done) == hipErrorNotReady);
      // a small batch on an idle engine: set-up, step tables and images as ONE launch (round 6; walk_pipe.hip,
      // pipe_small_prepare_kernel -- the same functions, the same bits; BITO_AMD_SMALL_PREPARE=1, off by default until
      // a device has run it)
      const bool fused_prepare = use_pipe && !busy && e->small_prepare && T <= 2048 && SetupReadsHostInputs(d, e->spec) &&
                                 PipeSmallPrepareApplies(d, e->spec);
      if (fused_prepare)
        LaunchPipeSmallPrepare(d, e->spec, b, plan, prep, two_classes ? split.slots_a : 0, two_classes ? split.groups_a : 4,
                               two_classes ? split.layout_a : kPipePlanAuto, class_flags);
      else
        LaunchSetup(d, e->spec, b, want_gradient, prep, /*beside_traversal=*/busy);
      NoteModelCache(e, b);
      // (that kernel made the device copies of the inputs: later passes wait for it.  The event is recorded behind the
      // kernels that follow on the same stream, not between the set-up and the image kernel: there the marker costs the
      // 6 us it takes the queue to retire it before the next kernel starts -- a 100-tree call's set-up and image
      // kernels are 26 and 20 us)
      if (e->inputs_on_host) {
        inputs_event_due = true;
        e->inputs_pending = true;
        e->inputs_on_host = false;
      }
      if (use_pipe) {
        if (!fused_prepare)
          LaunchPipePrepare(d, b, plan, prep, busy, two_classes ? split.slots_a : 0, two_classes ? split.groups_a : 4,
                            two_classes ? split.layout_a : kPipePlanAuto, class_flags);
        if (build_masks) LaunchPipeMasks(d, b, plan, reinterpret_cast<uint32_t*>(e->pipe_masks.ptr), prep);
        if (build_masks_a) LaunchPipeMasks(d, b, split.plan_a, reinterpret_cast<uint32_t*>(e->pipe_masks_a.ptr), prep);
      } else {
        LaunchMatrixImages(d, b, want_gradient, deriv_mode, prep);
        if (use_lds) LaunchLdsSchedule(d, b, plan, prep);
      }
      if (!in_line) {
        if (inputs_event_due) {
          HIP_TRY(e, hipEventRecord(e->ev_inputs, prep));
          inputs_event_due = false;
        }
        HIP_TRY(e, hipEventRecord(e->ev_prep_done[set], prep));
        HIP_TRY(e, hipStreamWaitEvent(walk, e->ev_prep_done[set], 0));
      }
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (e->timing) {
      ev0 = NextEvent(e);
      ev1 = NextEvent(e);
      HIP_TRY(e, hipEventRecord(ev0, walk));
    }
    bool last_unit = false;
    if (use_tree) LaunchWalkTree(d, b, tplan, want_gradient, walk);
    else if (use_pipe) {
      const int site = want_site && want_gradient && deriv_mode == 0;
      // whole-tree units write their trees' results themselves (BITO_AMD_PIPE_DIRECT=0: all through the final-sums kernel)
      if (e->pipe_direct && !site) {
        HIP_TRY(e, e->pipe_done.Reserve((size_t)T));
        b.pipe_done = e->pipe_done.ptr;
      }
      // (BITO_AMD_PIPE_LAST_UNIT=1: a one-class launch finishes every tree itself -- whole-tree units as ever, the others by
      // the last of their runs of tiles -- and stores the chunk's completion flag: no final-sums launch behind it)
      last_unit = e->pipe_last_unit && e->pipe_direct && !site && !two_classes;
      PipeClass whole{T, nullptr, b.pipe_masks, grad_rows, e->reserve_cus};
      if (last_unit) {
        const size_t had = e->pipe_tree_units.capacity;
        HIP_TRY(e, e->pipe_tree_units.Reserve((size_t)T));
        if (e->pipe_tree_units.capacity != had)  // (the counters are zero between launches: each tree's last unit resets its own)
          HIP_TRY(e, hipMemsetAsync(e->pipe_tree_units.ptr, 0, e->pipe_tree_units.capacity * sizeof(int32_t), walk));
        whole.tree_units = e->pipe_tree_units.ptr;
        whole.done = DoneByReduce(e);
      }
      if (two_classes) {
        LaunchWalkPipe(d, b, split.plan_a, want_gradient, site, deriv_mode, walk,
                       PipeClass{split.count_a, e->pipe_order.ptr, reinterpret_cast<const uint32_t*>(e->pipe_masks_a.ptr), grad_rows, e->reserve_cus});
        LaunchWalkPipe(d, b, split.plan_b, want_gradient, site, deriv_mode, walk,
                       PipeClass{split.count_b, e->pipe_order.ptr + split.count_a, b.pipe_masks, grad_rows, e->reserve_cus});
      } else {
        LaunchWalkPipe(d, b, plan, want_gradient, site, deriv_mode, walk, whole);
      }
    }
    else LaunchWalkLds(d, b, plan, want_gradient, want_site && want_gradient && deriv_mode == 0, walk);
    if (e->timing) HIP_TRY(e, hipEventRecord(ev1, walk));
    e->kernel_name = use_tree ? "walk_tree_kernel" : (use_pipe ? "walk_pipe_kernel" : "walk_lds_kernel");
    if (use_pipe) {
      auto form = [](const LdsPlan& p, int trees) {
        return std::to_string(trees) + " trees " + (p.layout == kPipePlanTwoWaves ? "two waves" : "one wave") + " per SIMD x " +
               std::to_string(p.groups) + " pattern groups, " + std::to_string(p.slots) + " vectors per wave";
      };
      e->kernel_form = two_classes ? form(split.plan_a, split.count_a) + " + " + form(split.plan_b, split.count_b) : form(plan, T);
    } else {
      e->kernel_form.clear();
    }
    e->site_ready = use_lds && want_gradient && deriv_mode == 0 && want_site;
    // (walk_pipe_kernel's partial log-likelihoods are per run of tiles as well)
    // (one-class launches: the whole-tree units are trees 0 .. whole_trees-1, all written by the traversal)
    const int reduce_from = (use_pipe && b.pipe_done != nullptr && !two_classes) ? plan.whole_trees : 0;
    if (!last_unit)
      LaunchReduce(d, b, use_pipe ? grad_rows : tiles, want_gradient, walk, grad_rows, DoneByReduce(e),
                   use_pipe ? b.pipe_done : nullptr, reduce_from);
    if (inputs_event_due) HIP_TRY(e, hipEventRecord(e->ev_inputs, walk));  // (in line: the set-up ran on this stream)
    if (!bare) HIP_TRY(e, hipEventRecord(e->ev_walk_done[set], walk));
    e->last_pass_done = bare ? nullptr : e->ev_walk_done[set];
    HIP_TRY(e, hipGetLastError());
    return BITO_AMD_OK;
  }
  // HBM-arena walk: the same set-up pipeline (tree set-up and transition matrices of this pass on prep_stream,
  // into the next buffer set, while earlier passes' traversals run)
  e->site_ready = false;
  const int set = (int)(e->run_counter++ % (unsigned)Worker::kSets);
  // scratch sized for this run
  HIP_TRY(e, (set == 0 ? e->mats : set == 1 ? e->mats2 : e->mats3).Reserve((size_t)T * NB * d.category_count * kMatStride));
  if (HbmCatKernelApplies(d))  // the steps' order (hbm_order_kernel)
    HIP_TRY(e, (set == 0 ? e->sched : set == 1 ? e->sched2 : e->sched3).Reserve(HbmOrderInts(d)));
  const size_t per_tree = HbmArenaBytesPerTree(d);
  size_t chunk = std::max<size_t>(1, std::min<size_t>((size_t)T, ArenaBudget(e, (size_t)T * per_tree) / per_tree));
  // grid.y limit
  chunk = std::min<size_t>(chunk, 65535);
  HIP_TRY(e, e->arena.Reserve(chunk * per_tree / sizeof(double)));
  if (want_gradient && rescaling)
    HIP_TRY(e, e->scale_arena.Reserve(chunk * (size_t)(d.taxon_count - 1) * d.pattern_stride));
  // (the fold level of walk_hbm_cat_kernel is this pass's: its step records are written for it below, unless the pass
  // re-uses the set's records -- serial_setup 2 -- and with them the level they were written for)
  if (!(e->serial_setup == 2 && e->run_counter > (unsigned)Worker::kSets)) e->hbm_fold_of_set[set] = HbmFoldLevel();
  const DeviceBatch b = MakeBatch(e, set);
  const hipStream_t walk = WalkStream(e);
  e->last_walk = walk;
  {
    hipStream_t prep = SetupStream(e);
    const bool in_line = prep == walk;
    if (!in_line) {
      e->prep_used = e->prep_used || !e->one_shot;
      HIP_TRY(e, hipStreamWaitEvent(prep, e->ev_walk_done[set], 0));
      if (e->inputs_pending && !e->one_shot) HIP_TRY(e, hipStreamWaitEvent(prep, e->ev_inputs, 0));
    }
    LaunchSetup(d, e->spec, b, want_gradient, prep);
    NoteModelCache(e, b);
    if (e->inputs_on_host) {
      HIP_TRY(e, hipEventRecord(e->ev_inputs, prep));
      e->inputs_pending = true;
      e->inputs_on_host = false;
    }
    LaunchMatrices(d, b, want_gradient, deriv_mode, prep, /*hot_only=*/HbmCatKernelApplies(d));
    if (HbmCatKernelApplies(d)) LaunchHbmOrder(d, b, prep);
    if (!in_line) {
      HIP_TRY(e, hipEventRecord(e->ev_prep_done[set], prep));
      HIP_TRY(e, hipStreamWaitEvent(walk, e->ev_prep_done[set], 0));
    }
  }
  for (int t0 = 0; t0 < T; t0 += (int)chunk) {
    const int ct = std::min<int>((int)chunk, T - t0);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (e->timing) {
      ev0 = NextEvent(e);
      ev1 = NextEvent(e);
      HIP_TRY(e, hipEventRecord(ev0, walk));
    }
    LaunchWalkHbm(d, b, t0, ct, want_gradient, rescaling, walk, deriv_mode);
    if (e->timing) HIP_TRY(e, hipEventRecord(ev1, walk));
  }
  e->kernel_name = WalkHbmKernelName(d.category_count, want_gradient, rescaling);
  const bool site_kernel = want_site && want_gradient && deriv_mode == 0 && d.category_count > 1 && HbmCatKernelApplies(d);
  e->signalled = false;
  LaunchReduce(d, b, tiles, want_gradient, walk, grad_rows, site_kernel ? ReduceDone{} : DoneByReduce(e));
  if (site_kernel) {
    // (walk_hbm_cat_kernel's gradient rows are per rate category: the site-model gradient needs no second pass)
    LaunchSiteFromCategoryRows(d, b, grad_rows, walk);
    e->site_ready = true;
  }
  HIP_TRY(e, hipEventRecord(e->ev_walk_done[set], walk));
  e->last_pass_done = e->ev_walk_done[set];
  HIP_TRY(e, hipGetLastError());
  return BITO_AMD_OK;
}

}  // namespace

extern "C" const char* bito_amd_version(void) {
  static std::string v;
  if (v.empty()) {
    v = "bito_amd 0.1 (gfx950)";
    int count = 0;
    if (hipGetDeviceCount(&count) == hipSuccess && count > 0) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, 0) == hipSuccess) {
        char buf[256];
        std::snprintf(buf, sizeof(buf), "bito_amd 0.1 %s %dCU %.0fGB", prop.gcnArchName,
                      prop.multiProcessorCount, prop.totalGlobalMem / 1e9);
        v = buf;
      }
    }
  }
  return v.c_str();
}

namespace bito_amd {

int WorkerCreate(int32_t device_id, uint64_t arena_bytes, const char* substitution, const char* site,
                 const char* clock, int32_t taxon_count, int32_t pattern_count, const int32_t* patterns,
                 const double* weights, Worker** out, std::string* err) {
  auto report = [&](int code, const std::string& msg) {
    if (err) *err = msg;
    return code;
  };
  if (!out) return report(BITO_AMD_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  auto e = new Worker();
  std::string msg;
  int rc = ParseSpec(substitution, site, clock, &e->spec, &e->blocks, &msg);
  if (rc) { delete e; return report(rc, msg); }
  if (taxon_count < 2 || pattern_count < 1 || !patterns || !weights) {
    delete e;
    return report(BITO_AMD_ERR_BAD_ARG, "need at least 2 taxa, 1 site pattern and non-NULL arrays");
  }
  e->device = device_id;
  int count = 0;
  hipError_t hrc = hipGetDeviceCount(&count);
  if (hrc != hipSuccess || count <= 0) {
    delete e;
    return report(BITO_AMD_ERR_DEVICE, "no HIP device available: the bito_amd engine needs an MI355X (gfx950); there is no CPU fallback");
  }
  if (e->device < 0 || e->device >= count) {
    delete e;
    return report(BITO_AMD_ERR_DEVICE, "device_id out of range");
  }
  auto dev_fail = [&](const char* what, hipError_t c) {
    std::string m = std::string(what) + " failed: " + hipGetErrorString(c);
    delete e;
    return report(BITO_AMD_ERR_DEVICE, m);
  };
  if ((hrc = hipSetDevice(e->device)) != hipSuccess) return dev_fail("hipSetDevice", hrc);
  if ((hrc = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess)
    return dev_fail("hipStreamCreate", hrc);
  // The set-up stream gets its own priority level: the runtime keeps a separate pool of hardware queues per
  // priority, so the two streams can never be folded onto ONE hardware queue (which serialises them) however
  // many streams the process already holds.  Measured: with an RCCL process group created first, two
  // normal-priority streams shared a queue and the set-up overlap was gone (2.24 ms per step against 2.17).
  {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if ((hrc = hipStreamCreateWithPriority(&e->prep_stream, hipStreamNonBlocking, least)) != hipSuccess)
      return dev_fail("hipStreamCreate", hrc);
  }
  if (const char* serial = std::getenv("BITO_AMD_SERIAL_SETUP")) e->serial_setup = std::atoi(serial);
  if (const char* direct = std::getenv("BITO_AMD_PIPE_DIRECT")) e->pipe_direct = std::atoi(direct) != 0;
  if (const char* fused = std::getenv("BITO_AMD_SMALL_PREPARE")) e->small_prepare = std::atoi(fused) != 0;
  if (const char* last = std::getenv("BITO_AMD_PIPE_LAST_UNIT")) e->pipe_last_unit = std::atoi(last) != 0;
  if (const char* two = std::getenv("BITO_AMD_PIPE_TWO")) e->pipe_two = std::atoi(two);
  if (const char* fold = std::getenv("BITO_AMD_PIPE_FOLD")) e->pipe_fold = std::atoi(fold);
  for (int i = 0; i < Worker::kSets; i++) {
    if ((hrc = hipEventCreateWithFlags(&e->ev_prep_done[i], hipEventDisableTiming)) != hipSuccess ||
        (hrc = hipEventCreateWithFlags(&e->ev_walk_done[i], hipEventDisableTiming)) != hipSuccess)
      return dev_fail("hipEventCreate", hrc);
  }
  if ((hrc = hipEventCreateWithFlags(&e->ev_inputs, hipEventDisableTiming)) != hipSuccess ||
      (hrc = hipEventCreateWithFlags(&e->ev_results, hipEventDisableTiming)) != hipSuccess)
    return dev_fail("hipEventCreate", hrc);
  e->n = taxon_count;
  e->P = pattern_count;
  // padded so that every kernel's last tile (at most 512 patterns wide) stays in bounds
  e->Ppad = (pattern_count + 512 + kHbmBlock - 1) / kHbmBlock * kHbmBlock;
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  e->arena_limit = arena_bytes ? arena_bytes : std::max<size_t>(free_b / 4 * 3, (size_t)1 << 28);
  e->arena_auto = arena_bytes == 0;
  // Compact tip states, gap for every symbol >= 4 and for the padding columns
  // (SitePattern symbol table, reference src/site_pattern.cpp:16-46).
  const int S = e->spec.state_count;
  std::vector<uint8_t> tips((size_t)e->n * e->Ppad, (uint8_t)S);
  for (int t = 0; t < e->n; t++)
    for (int p = 0; p < e->P; p++) {
      const int32_t s = patterns[(size_t)t * e->P + p];
      if (s < 0) { delete e; return report(BITO_AMD_ERR_BAD_ARG, "negative pattern symbol"); }
      tips[(size_t)t * e->Ppad + p] = (uint8_t)(s >= S ? S : s);
    }
  std::vector<double> w(e->Ppad, 0.0);
  std::copy(weights, weights + e->P, w.begin());
  if ((hrc = e->tip_states.Reserve(tips.size())) != hipSuccess) return dev_fail("hipMalloc", hrc);
  if ((hrc = e->weights.Reserve(w.size())) != hipSuccess) return dev_fail("hipMalloc", hrc);
  if ((hrc = hipMemcpy(e->tip_states.ptr, tips.data(), tips.size(), hipMemcpyHostToDevice)) != hipSuccess)
    return dev_fail("hipMemcpy", hrc);
  if ((hrc = hipMemcpy(e->weights.ptr, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess)
    return dev_fail("hipMemcpy", hrc);
  *out = e;
  return BITO_AMD_OK;
}

void WorkerDestroy(Worker* e) { delete e; }

const char* WorkerLastError(const Worker* e) { return e ? e->err.c_str() : ""; }

int32_t WorkerParamCount(const Worker* e) { return e->spec.param_count; }
int32_t WorkerCategoryCount(const Worker* e) { return e->spec.category_count; }
int32_t WorkerStateCount(const Worker* e) { return e->spec.state_count; }
int32_t WorkerBlockCount(const Worker* e) { return (int32_t)e->blocks.size(); }

int WorkerBlock(const Worker* e, int32_t idx, char* name, size_t name_len,
                          int32_t* start, int32_t* len) {
  if (idx < 0 || idx >= (int32_t)e->blocks.size()) return BITO_AMD_ERR_BAD_ARG;
  const Block& b = e->blocks[idx];
  if (name && name_len) std::snprintf(name, name_len, "%s", b.name.c_str());
  if (start) *start = b.start;
  if (len) *len = b.len;
  return BITO_AMD_OK;
}

// Validates a block of trees, packs its wire-format inputs into the worker's pinned staging buffer and enqueues
// ONE copy into the device input block (same layout) on the worker's stream.  wait != 0: returns when the copy
// has landed (the classic upload); wait == 0: returns at once -- the worker's own streams are ordered behind the
// copy (ev_inputs), the staging buffer is not touched again before the next WorkerStage, which waits for the
// worker to be idle first.
int WorkerStage(Worker* e, int32_t tree_count, int32_t rooted, int32_t node_count, const int32_t* parent_ids,
                const double* branch_lengths, const double* rates, const double* params, int wait) {
  if (int rc = WorkerStageBegin(e, tree_count, rooted, node_count, parent_ids, branch_lengths, rates, params, wait)) return rc;
  StagePart part;
  WorkerStageFill(e, 0, tree_count, &part);
  return WorkerStageEnd(e, &part, 1);
}

// WorkerStage in three steps, so that the engine level can have several host threads check and pack disjoint ranges
// of a large block (WorkerStageFill makes no HIP call and writes only its own rows): Begin and End on the calling
// thread, Fill(t0, t1) for ranges that cover [0, tree_count) on any threads in between.
int WorkerStageBegin(Worker* e, int32_t tree_count, int32_t rooted, int32_t node_count, const int32_t* parent_ids,
                     const double* branch_lengths, const double* rates, const double* params, int wait) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  e->resident = false;
  if (tree_count < 1 || !parent_ids || !branch_lengths)
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "need at least one tree and non-NULL parent_ids / branch_lengths");
  if (e->spec.param_count > 0 && !params)
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "params is NULL but the model has parameters");
  if (int rc = ValidateTreeShape(e, rooted, node_count)) return rc;
  e->tree_cherries.assign((size_t)tree_count, 0);
  if (TracksTreeGuard(e, rooted, rates)) e->tree_rev_ok.assign((size_t)tree_count, 0);
  else e->tree_rev_ok.clear();
  e->pipe_split = Worker::PipeSplit{};
  HIP_TRY(e, hipSetDevice(e->device));
  // A set-up kernel of an earlier, still running pass may be reading the input buffers, an earlier copy the staging
  // buffer.  (Only when something may be in flight: a stream synchronisation is not free even on an idle stream --
  // the runtime multiplexes streams onto a few hardware queues, and the marker it waits for can queue up behind
  // ANOTHER worker's traversal: measured 0.77 ms in a five-chunk call.)
  if (e->busy) {
    HIP_TRY(e, hipStreamSynchronize(e->prep_stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->last_walk && e->last_walk != e->stream) HIP_TRY(e, hipStreamSynchronize(e->last_walk));
    e->prep_used = false;
  }
  e->busy = true;
  const int n = e->n, N = 2 * n - 1, M = node_count;
  const size_t T = tree_count;
  const size_t pc = (size_t)e->spec.param_count;
  e->has_rates = rooted && rates != nullptr;
  // layout of the input block, in doubles (the parent ids close it, as int32)
  Worker::Staging& st = e->staging;
  st = Worker::Staging{};
  st.tree_count = tree_count;
  st.rooted = rooted;
  st.node_count = node_count;
  st.wait = wait;
  st.parent_ids = parent_ids;
  st.branch_lengths = branch_lengths;
  st.rates = e->has_rates ? rates : nullptr;
  st.params = pc > 0 ? params : nullptr;
  st.off_params = T * M;
  st.off_rates = st.off_params + T * std::max<size_t>(pc, 1);
  st.off_pid = st.off_rates + (e->has_rates ? T * (M - 1) : 0);
  st.bytes = st.off_pid * sizeof(double) + T * (M - 1) * sizeof(int32_t);
  HIP_TRY(e, e->in_block.Reserve((st.bytes + sizeof(double) - 1) / sizeof(double)));
  HIP_TRY(e, e->pin_in.Reserve(st.bytes));
  HIP_TRY(e, e->children.Reserve(T * (n - 1) * 2));
  HIP_TRY(e, e->branch.Reserve(T * N));
  HIP_TRY(e, e->model.Reserve(T));
  HIP_TRY(e, e->children2.Reserve(T * (n - 1) * 2));
  HIP_TRY(e, e->branch2.Reserve(T * N));
  HIP_TRY(e, e->model2.Reserve(T));
  HIP_TRY(e, e->children3.Reserve(T * (n - 1) * 2));
  HIP_TRY(e, e->branch3.Reserve(T * N));
  HIP_TRY(e, e->model3.Reserve(T));
  for (auto& r : e->out_ll_ring) HIP_TRY(e, r.Reserve(T));
  HIP_TRY(e, e->out_grad.Reserve(T * N));
  HIP_TRY(e, e->out_site.Reserve(T));
  HIP_TRY(e, e->pin_out.Reserve(T * (N + 2) * sizeof(double)));
  if (!e->pin_flag.ptr) {
    HIP_TRY(e, e->pin_flag.Reserve(64));
    std::memset(e->pin_flag.ptr, 0, 64);
    HIP_TRY(e, e->done_counter.Reserve(1));
    HIP_TRY(e, hipMemset(e->done_counter.ptr, 0, sizeof(int32_t)));
  }
  e->branch_in.ptr = e->in_block.ptr;
  e->params.ptr = e->in_block.ptr + st.off_params;
  e->rates.ptr = e->has_rates ? e->in_block.ptr + st.off_rates : nullptr;
  e->parent_ids.ptr = reinterpret_cast<int32_t*>(e->in_block.ptr + st.off_pid);
  st.open = true;
  return BITO_AMD_OK;
}

void WorkerStageFill(Worker* e, int32_t t0, int32_t t1, StagePart* out) {
  const Worker::Staging& st = e->staging;
  const size_t M = (size_t)st.node_count, pc = (size_t)e->spec.param_count;
  const size_t a = (size_t)t0, count = (size_t)(t1 - t0);
  *out = StagePart{};
  out->first_tree = t0;
  out->end_tree = t1;
  out->code = ValidateTreesRange(e, t0, t1, st.rooted, st.node_count, st.parent_ids, &out->min_cherries,
                                 e->tree_cherries.data(), &out->message, &out->min_unstored, &out->pitchforks_counted);
  if (!out->code && st.params) out->code = ValidateParamsRange(e, t0, t1, st.params, &out->message);
  if (out->code) return;
  double* stage = static_cast<double*>(e->pin_in.ptr);
  std::memcpy(stage + a * M, st.branch_lengths + a * M, count * M * sizeof(double));
  if (pc > 0) std::memcpy(stage + st.off_params + a * pc, st.params + a * pc, count * pc * sizeof(double));
  if (st.rates) std::memcpy(stage + st.off_rates + a * (M - 1), st.rates + a * (M - 1), count * (M - 1) * sizeof(double));
  std::memcpy(reinterpret_cast<int32_t*>(stage + st.off_pid) + a * (M - 1), st.parent_ids + a * (M - 1),
              count * (M - 1) * sizeof(int32_t));
  if (e->n > kPipeExactTaxa) {
    out->min_branch = MinBranchLength(st.branch_lengths + a * M, st.rooted && st.rates ? st.rates + a * (M - 1) : nullptr, count, M);
    out->min_rate = MinOffDiagonalRate(e->spec, st.params ? st.params + a * pc : nullptr, count);
  }
  if (!e->tree_rev_ok.empty())  // (small trees: the two-wave form, tree by tree)
    TreeGuardRange(e->spec, st.branch_lengths, st.rooted && st.rates ? st.rates : nullptr, st.params, a, a + count, M,
                   e->tree_rev_ok.data());
}

int WorkerStageEnd(Worker* e, const StagePart* parts, int part_count) {
  Worker::Staging& st = e->staging;
  if (!st.open) return Fail(e, BITO_AMD_ERR_STATE, "WorkerStageEnd without WorkerStageBegin");
  st.open = false;
  // the error a serial pass over the trees would have met first
  const StagePart* bad = nullptr;
  for (int i = 0; i < part_count; i++)
    if (parts[i].code && (!bad || parts[i].first_tree < bad->first_tree)) bad = &parts[i];
  if (bad) return Fail(e, bad->code, bad->message);
  int min_cherries = st.node_count, min_unstored = st.node_count;
  double min_branch = std::numeric_limits<double>::infinity(), min_rate = std::numeric_limits<double>::infinity();
  for (int i = 0; i < part_count; i++) {
    min_cherries = std::min(min_cherries, parts[i].min_cherries);
    min_unstored = std::min(min_unstored, parts[i].min_unstored);
    min_branch = std::min(min_branch, parts[i].min_branch);
    min_rate = std::min(min_rate, parts[i].min_rate);
  }
  // Pitchforks: a range counts them when ITS trees, by their cherries, do not fit beside the most pattern groups; the batch
  // needs them when ANY range does.  The ranges that did not count are counted here (0.1 us per tree, in the mixed case
  // alone), so that the per-tree counts, the class split and the plan are those of a call checked in one range.
  if (FoldingApplies(e) && part_count > 1 && CherriesAloneDoNotFit(e, min_cherries)) {
    std::vector<int> kids;
    min_unstored = st.node_count;  // (from scratch: an uncounted range's minimum above is its cherries')
    for (int i = 0; i < part_count; i++) {
      if (parts[i].pitchforks_counted) {
        min_unstored = std::min(min_unstored, parts[i].min_unstored);
        continue;
      }
      for (int t = parts[i].first_tree; t < parts[i].end_tree; t++) {
        const int unstored = UnstoredNodes(e->n, st.node_count, st.rooted, st.parent_ids + (size_t)t * (st.node_count - 1), true, &kids);
        e->tree_cherries[(size_t)t] = unstored;
        min_unstored = std::min(min_unstored, unstored);
      }
    }
  }
  const int tree_count = st.tree_count, rooted = st.rooted, wait = st.wait;
  const int n = e->n, N = 2 * n - 1, M = st.node_count, C = e->spec.category_count;
  const double* stage = static_cast<const double*>(e->pin_in.ptr);
  // A blocking call's chunk of trees small enough for the staging set-up kernel: no copy at all, that kernel reads
  // the pinned buffer over PCIe (measured: the copy engine's start-up and the hand-over to the kernel behind it cost
  // 30 us per call) and writes the device copies.  Otherwise one copy, on the stream the set-up kernels follow on
  // (the worker's own: with ev_inputs behind it for the set-up stream).
  BatchDims probe{};
  probe.taxon_count = n;
  probe.node_count = N;
  probe.in_node_count = M;
  probe.rooted = rooted;
  probe.tree_count = tree_count;
  e->inputs_on_host = e->one_shot && !wait && e->spec.state_count == 4 && e->kernel_choice != BITO_AMD_KERNEL_GENERAL &&
                      SetupReadsHostInputs(probe, e->spec) &&
                      !(e->one_shot == 2 && InputsCopyMin() > 0 && tree_count >= InputsCopyMin());
  // the model of the last small call's tree 0 serves this call when every parameter row equals the row it came from
  e->model_reuse_next = false;
  st.row0.clear();
  if (e->inputs_on_host && e->spec.param_count > 0 && ModelCacheOn()) {
    const size_t pc = (size_t)e->spec.param_count;
    HIP_TRY(e, e->model_cache.Reserve(1));
    st.row0.assign(st.params, st.params + pc);
    bool same = e->model_cache_valid && e->model_cache_row.size() == pc;
    for (int t = 0; same && t < tree_count; t++)
      same = std::memcmp(st.params + (size_t)t * pc, e->model_cache_row.data(), pc * sizeof(double)) == 0;
    e->model_reuse_next = same;
  }
  if (!e->inputs_on_host) {
    HIP_TRY(e, hipMemcpyAsync(e->in_block.ptr, stage, st.bytes, hipMemcpyHostToDevice, SetupStream(e)));
    HIP_TRY(e, hipEventRecord(e->ev_inputs, SetupStream(e)));
    e->inputs_pending = true;
  }
  e->min_branch = e->n > kPipeExactTaxa ? min_branch : 0.0;
  e->min_rate = e->n > kPipeExactTaxa ? min_rate : 1.0;
  e->gs_index_valid = false;
  if (e->spec.state_count != 4 || e->kernel_choice == BITO_AMD_KERNEL_GENERAL) {
    static const double none = 0.0;
    if (int rc = UploadModelIndex(e, tree_count, e->spec.param_count > 0 ? st.params : &none)) return rc;
  }
  if (wait) {
    HIP_TRY(e, hipStreamSynchronize(SetupStream(e)));
    e->inputs_pending = false;
  }
  e->dims.taxon_count = n;
  e->dims.node_count = N;
  e->dims.in_node_count = M;
  e->dims.rooted = rooted;
  e->dims.pattern_count = e->P;
  e->dims.pattern_stride = e->Ppad;
  e->dims.category_count = C;
  e->dims.tree_count = tree_count;
  e->dims.min_cherries = min_cherries;
  e->dims.min_unstored = std::max(min_unstored, min_cherries);
  e->dims.pipe_fold = e->pipe_fold;
  e->resident = true;
  return BITO_AMD_OK;
}

// the host-side checks of WorkerStage alone (messages name trees by their position in the arrays given)
int WorkerValidate(Worker* e, int32_t tree_count, int32_t rooted, int32_t node_count, const int32_t* parent_ids,
                   const double* params) {
  if (!e || !parent_ids) return BITO_AMD_ERR_BAD_ARG;
  e->id_offset = 0;
  int rc = ValidateTrees(e, tree_count, rooted, node_count, parent_ids);
  if (!rc && params) rc = ValidateParams(e, tree_count, params);
  return rc;
}

int WorkerUpload(Worker* e, int32_t tree_count, int32_t rooted, int32_t node_count, const int32_t* parent_ids,
                 const double* branch_lengths, const double* rates, const double* params) {
  return WorkerStage(e, tree_count, rooted, node_count, parent_ids, branch_lengths, rates, params, /*wait=*/1);
}

// Asks for the last pass's results in the pinned staging buffer -- [ll T][gradient T*N][site T].  A blocking
// call's chunk (one_shot) already wrote them there: only the completion flag is enqueued.  Otherwise the copies,
// with ev_results recorded behind them.  WorkerResults waits and hands out the three host addresses.
int WorkerFetchResults(Worker* e, int want_gradient, int want_site) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (!e->resident) return Fail(e, BITO_AMD_ERR_STATE, "no batch is resident");
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t T = e->dims.tree_count, N = e->dims.node_count;
  if (e->results_on_host) {
    if (!e->signalled) {
      LaunchSignal(static_cast<unsigned long long*>(e->pin_flag.ptr), ++e->ticket, e->last_walk ? e->last_walk : e->stream);
      HIP_TRY(e, hipGetLastError());
    }
    return BITO_AMD_OK;
  }
  HIP_TRY(e, e->pin_out.Reserve((T * (N + 2)) * sizeof(double)));
  double* out = static_cast<double*>(e->pin_out.ptr);
  HIP_TRY(e, hipMemcpyAsync(out, e->cur_ll(), T * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (want_gradient)
    HIP_TRY(e, hipMemcpyAsync(out + T, e->out_grad.ptr, T * N * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (want_site && e->site_ready)
    HIP_TRY(e, hipMemcpyAsync(out + T + T * N, e->out_site.ptr, T * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipEventRecord(e->ev_results, e->stream));
  return BITO_AMD_OK;
}

bool WorkerResultsReady(Worker* e) {
  if (e->results_on_host)
    return __atomic_load_n(static_cast<volatile uint64_t*>(e->pin_flag.ptr), __ATOMIC_ACQUIRE) == e->ticket;
  return hipEventQuery(e->ev_results) == hipSuccess;
}

int WorkerResults(Worker* e, const double** ll, const double** grad, const double** site) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  HIP_TRY(e, hipSetDevice(e->device));
  if (e->results_on_host) {
    // poll the flag the chunk's last kernel stores; now and then ask the stream whether it has failed
    volatile uint64_t* flag = static_cast<volatile uint64_t*>(e->pin_flag.ptr);
    for (unsigned spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != e->ticket; spins++) {
      CpuPause();
      if ((spins & 0xffff) == 0xffff) {
        const hipError_t state = hipStreamQuery(e->last_walk ? e->last_walk : e->stream);
        if (state != hipSuccess && state != hipErrorNotReady)
          return Fail(e, BITO_AMD_ERR_DEVICE, std::string("the pass failed on the device: ") + hipGetErrorString(state));
        if (state == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != e->ticket)
          return Fail(e, BITO_AMD_ERR_DEVICE, "the stream is idle but the completion flag was never stored");
      }
    }
  } else {
    HIP_TRY(e, hipEventSynchronize(e->ev_results));
  }
  e->inputs_pending = false;
  e->busy = e->prep_used;  // (the flag / ev_results is the last thing on `stream`; the set-up stream is idle unless a pipelined pass used it)
  const size_t T = e->dims.tree_count, N = e->dims.node_count;
  const double* out = static_cast<const double*>(e->pin_out.ptr);
  if (ll) *ll = out;
  if (grad) *grad = out + T;
  if (site) *site = out + T + T * N;
  return BITO_AMD_OK;
}

int WorkerUpdate(Worker* e, const double* branch_lengths, const double* params) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (!e->resident) return Fail(e, BITO_AMD_ERR_STATE, "no batch is resident: call WorkerUpload first");
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipStreamSynchronize(e->prep_stream));  // (see WorkerUpload)
  const size_t T = e->dims.tree_count;
  bool h_params_keep = false;
  if (params && e->spec.param_count > 0) {
    int rc = ValidateParams(e, (int)T, params);
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->params.ptr, params, T * e->spec.param_count * sizeof(double), hipMemcpyHostToDevice, e->stream));
    e->gs_index_valid = false;
    if (e->spec.state_count != 4 || e->kernel_choice == BITO_AMD_KERNEL_GENERAL)
      if ((rc = UploadModelIndex(e, (int)T, params))) return rc;
    if (e->n > kPipeExactTaxa) e->min_rate = MinOffDiagonalRate(e->spec, params, T);
    h_params_keep = true;
  }
  if (branch_lengths) {
    HIP_TRY(e, hipMemcpyAsync(e->branch_in.ptr, branch_lengths, T * e->dims.in_node_count * sizeof(double), hipMemcpyHostToDevice, e->stream));
    // (rates, when a batch has them, stay on the device: unknown here, so no claim about the effective lengths)
    e->min_branch = (e->n > kPipeExactTaxa && !e->has_rates) ? MinBranchLength(branch_lengths, nullptr, T, (size_t)e->dims.in_node_count) : 0.0;
  }
  if (!e->tree_rev_ok.empty() && (branch_lengths || h_params_keep)) {
    // the two-wave form's classes follow the new values.  (Only the array that came with this call is known here: with
    // new branch lengths AND the batch's parameter rows the guard is re-evaluated tree by tree; otherwise -- rates on
    // the device, or one of the two arrays missing -- no tree is claimed to hold it and the batch runs on the one-wave kernel.)
    if (branch_lengths && !e->has_rates && (params || e->spec.param_count == 0))
      TreeGuardRange(e->spec, branch_lengths, nullptr, params, 0, T, (size_t)e->dims.in_node_count, e->tree_rev_ok.data());
    else
      std::fill(e->tree_rev_ok.begin(), e->tree_rev_ok.end(), (uint8_t)0);
    e->pipe_split = Worker::PipeSplit{};
  }
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return BITO_AMD_OK;
}

int WorkerRun(Worker* e, int32_t want_gradient, int32_t rescaling) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  return RunResident(e, want_gradient != 0, rescaling != 0);
}

int WorkerRunPass(Worker* e, int want_gradient, int rescaling, int deriv_mode, int want_site) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  return RunResident(e, want_gradient, rescaling, deriv_mode, want_site);
}

int WorkerSync(Worker* e) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (e->last_walk && e->last_walk != e->stream) HIP_TRY(e, hipStreamSynchronize(e->last_walk));
  e->inputs_pending = false;
  return BITO_AMD_OK;
}

int WorkerDownloadAsync(Worker* e, double* out_ll, double* out_grad) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (!e->resident) return Fail(e, BITO_AMD_ERR_STATE, "no batch is resident");
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t T = e->dims.tree_count;
  // (a blocking call's chunk left its results in the pinned staging buffer, which a device reads as well)
  const double* ll = e->results_on_host ? static_cast<const double*>(e->pin_out.ptr) : e->cur_ll();
  const double* grad = e->results_on_host ? static_cast<const double*>(e->pin_out.ptr) + T : e->out_grad.ptr;
  if (out_ll) HIP_TRY(e, hipMemcpyAsync(out_ll, ll, T * sizeof(double), hipMemcpyDefault, e->stream));
  if (out_grad)
    HIP_TRY(e, hipMemcpyAsync(out_grad, grad, T * e->dims.node_count * sizeof(double), hipMemcpyDefault, e->stream));
  return BITO_AMD_OK;
}

int WorkerResultsAsync(Worker* e, void* consumer_stream, const double** out_ll,
                                  const double** out_grad) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (!e->resident) return Fail(e, BITO_AMD_ERR_STATE, "no batch is resident");
  HIP_TRY(e, hipSetDevice(e->device));
  hipStream_t consumer = static_cast<hipStream_t>(consumer_stream);
  if (e->last_pass_done) {
    HIP_TRY(e, hipStreamWaitEvent(consumer, e->last_pass_done, 0));
  } else {  // (nothing recorded behind the last pass: an event of its own)
    hipEvent_t ev = NextEvent(e);
    HIP_TRY(e, hipEventRecord(ev, e->stream));
    HIP_TRY(e, hipStreamWaitEvent(consumer, ev, 0));
  }
  if (out_ll) *out_ll = e->results_on_host ? static_cast<const double*>(e->pin_out.ptr) : e->cur_ll();
  if (out_grad) *out_grad = e->results_on_host ? static_cast<const double*>(e->pin_out.ptr) + e->dims.tree_count : e->out_grad.ptr;
  return BITO_AMD_OK;
}

int WorkerDownload(Worker* e, double* out_ll, double* out_grad) {
  if (int rc = WorkerDownloadAsync(e, out_ll, out_grad)) return rc;
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return BITO_AMD_OK;
}

void* WorkerStream(Worker* e) { return e ? (void*)e->stream : nullptr; }

int WorkerLogLikelihoods(Worker* e, int32_t tree_count, int32_t rooted,
                                    int32_t node_count, const int32_t* parent_ids,
                                    const double* branch_lengths, const double* rates,
                                    const double* params, int32_t rescaling, double* out) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  // an empty collection is not an error: FatBeagleParallelize over no trees returns an empty vector
  // (reference src/fat_beagle.hpp:160-181)
  if (tree_count == 0) return BITO_AMD_OK;
  int rc = WorkerUpload(e, tree_count, rooted, node_count, parent_ids, branch_lengths, rates, params);
  if (rc) return rc;
  if ((rc = RunResident(e, 0, rescaling != 0))) return rc;
  return WorkerDownload(e, out, nullptr);
}

// StickBreakingTransform (reference src/stick_breaking_transform.cpp:10-44): the Stan
// simplex transform; x = T(y) has K entries, y has K-1.
void StickForward(const double* y, int K, double* x) {
  double stick = 1.0;
  for (int k = 0; k < K - 1; k++) {
    const double z = 1.0 / (1 + std::exp(-(y[k] - std::log((double)(K - k - 1)))));
    x[k] = stick * z;
    stick -= x[k];
  }
  x[K - 1] = stick;
}
void StickInverse(const double* x, int K, double* y) {
  double sum = 0;
  for (int k = 0; k < K - 1; k++) {
    const double z = x[k] / (1.0 - sum);
    y[k] = std::log(z / (1.0 - z)) + std::log((double)(K - k - 1));
    sum += x[k];
  }
}

// FatBeagle::SubstitutionModelGradient (reference src/fat_beagle.cpp:412-508): central finite
// differences of the tree log-likelihood in every free substitution-model parameter, rates
// first then frequencies, in stick-breaking coordinates when requested (frequencies always,
// rates only for GTR's six).  The 2 x (#parameters) perturbed evaluations of every tree are
// run as ONE batch of log-likelihood-only passes on the device.
int SubstitutionGradientsVia(const ModelSpec& m, int T, int rooted, int node_count, const int32_t* parent_ids,
                             const double* branch_lengths, const double* rates, const double* params, bool stick,
                             double delta, double* out_subst, const LogLikelihoodFn& log_likelihoods) {
  const int pc = m.param_count;
  struct Dir { int start, len, index; bool stick; };
  std::vector<Dir> dirs;
  const bool rates_stick = stick && m.rates_len == 6;
  for (int i = 0; i < (rates_stick ? 5 : m.rates_len); i++) dirs.push_back({m.rates_start, m.rates_len, i, rates_stick});
  for (int i = 0; i < (stick ? 3 : 4); i++) dirs.push_back({m.freq_start, 4, i, stick});
  const int K = (int)dirs.size(), M = node_count;
  const size_t big = (size_t)T * 2 * K;
  std::vector<int32_t> pid(big * (M - 1));
  std::vector<double> bl(big * M), par(big * pc), rt;
  if (rooted && rates) rt.resize(big * (M - 1));
  for (int t = 0; t < T; t++)
    for (int j = 0; j < 2 * K; j++) {
      const size_t r = (size_t)t * 2 * K + j;
      std::copy(parent_ids + (size_t)t * (M - 1), parent_ids + (size_t)(t + 1) * (M - 1), pid.begin() + r * (M - 1));
      std::copy(branch_lengths + (size_t)t * M, branch_lengths + (size_t)(t + 1) * M, bl.begin() + r * M);
      if (!rt.empty()) std::copy(rates + (size_t)t * (M - 1), rates + (size_t)(t + 1) * (M - 1), rt.begin() + r * (M - 1));
      double* row = par.data() + r * pc;
      std::copy(params + (size_t)t * pc, params + (size_t)(t + 1) * pc, row);
      const Dir& dr = dirs[j / 2];
      const double sign = (j % 2 == 0) ? 1.0 : -1.0;
      double y[8];
      if (dr.stick) {
        StickInverse(row + dr.start, dr.len, y);
        y[dr.index] += sign * delta;
        StickForward(y, dr.len, row + dr.start);
      } else {
        row[dr.start + dr.index] += sign * delta;
      }
    }
  std::vector<double> ll(big);
  int rc = log_likelihoods((int32_t)big, pid.data(), bl.data(), rt.empty() ? nullptr : rt.data(), par.data(), ll.data());
  if (rc) return rc;
  const int stride = m.rates_len + 4;
  for (int t = 0; t < T; t++)
    for (int k = 0; k < K; k++)
      out_subst[(size_t)t * stride + k] = (ll[((size_t)t * K + k) * 2] - ll[((size_t)t * K + k) * 2 + 1]) / (2. * delta);
  return BITO_AMD_OK;
}

// The site-model gradient for kernels that do not produce it in the main pass: a second gradient pass with
// dQ = Q * d r_c / d shape, then sum_b g_b t_b over the effective branch lengths (DiscreteSiteModelGradient,
// reference src/fat_beagle.cpp:401-410,538-550).  The arrays are the resident batch's own rows.
int WorkerSiteGradientSecondPass(Worker* e, int32_t rooted, int32_t node_count, const double* branch_lengths,
                                 const double* rates, int32_t rescaling, double* out_site) {
  const int tree_count = e->dims.tree_count, N = 2 * e->n - 1;
  int rc;
  if ((rc = RunResident(e, 1, rescaling != 0, /*deriv_mode=*/1))) return rc;
  std::vector<double> g2((size_t)tree_count * N);
  if ((rc = WorkerDownload(e, nullptr, g2.data()))) return rc;
  for (int t = 0; t < tree_count; t++) {
    double s = 0;
    for (int i = 0; i < node_count - 1; i++) {
      double bl = branch_lengths[(size_t)t * node_count + i];
      if (rooted && rates) bl *= rates[(size_t)t * (node_count - 1) + i];
      s += g2[(size_t)t * N + i] * bl;
    }
    out_site[t] = s;  // unrooted: the two extra nodes of the detrifurcated tree have branch length 0
  }
  // leave the device results of the main pass in place for WorkerDownload
  if ((rc = RunResident(e, 1, rescaling != 0))) return rc;
  return WorkerSync(e);
}

int WorkerGradients(Worker* e, int32_t tree_count, int32_t rooted,
                              int32_t node_count, const int32_t* parent_ids,
                              const double* branch_lengths, const double* rates,
                              const double* params, int32_t rescaling, int32_t flags,
                              double fd_delta, double* out_ll, double* out_branch,
                              double* out_site, double* out_subst, double* out_clock) {
  if (!e || !out_ll || !out_branch) return BITO_AMD_ERR_BAD_ARG;
  if (tree_count == 0) return BITO_AMD_OK;  // empty collection, empty result (as WorkerLogLikelihoods)
  int rc;
  // the finite-difference batch first: the main batch must be the resident one on return
  if ((flags & BITO_AMD_GRAD_SUBSTITUTION_MODEL) && out_subst && e->spec.rates_len > 0) {
    rc = SubstitutionGradientsVia(
        e->spec, tree_count, rooted, node_count, parent_ids, branch_lengths, rates, params,
        (flags & BITO_AMD_GRAD_STICKBREAKING) != 0, fd_delta > 0 ? fd_delta : 1e-6, out_subst,
        [&](int32_t big, const int32_t* pid, const double* bl, const double* rt, const double* par, double* ll) {
          return WorkerLogLikelihoods(e, big, rooted, node_count, pid, bl, rt, par, rescaling, ll);
        });
    if (rc) return rc;
  }
  rc = WorkerUpload(e, tree_count, rooted, node_count, parent_ids, branch_lengths, rates, params);
  if (rc) return rc;
  const int want_site = (flags & BITO_AMD_GRAD_SITE_MODEL) && out_site && e->spec.category_count > 1;
  if ((rc = RunResident(e, 1, rescaling != 0, /*deriv_mode=*/0, want_site))) return rc;
  if ((rc = WorkerDownload(e, out_ll, out_branch))) return rc;
  const int N = 2 * e->n - 1;
  if ((flags & BITO_AMD_GRAD_SITE_MODEL) && out_site && e->spec.category_count > 1 && e->site_ready) {
    // the traversal produced it in the same pass (per-category edge sums: walk_lds_kernel, walk_pipe_kernel,
    // walk_hbm_cat_kernel)
    HIP_TRY(e, hipMemcpyAsync(out_site, e->out_site.ptr, (size_t)tree_count * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  } else if ((flags & BITO_AMD_GRAD_SITE_MODEL) && out_site && e->spec.category_count > 1) {
    if ((rc = WorkerSiteGradientSecondPass(e, rooted, node_count, branch_lengths, rates, rescaling, out_site))) return rc;
  }
  if (rooted && (flags & BITO_AMD_GRAD_CLOCK_MODEL) && out_clock) {
    // ClockGradient, strict clock (reference src/fat_beagle.cpp:379-399): sum of
    // branch gradient times the tree's own (time) branch length.
    for (int t = 0; t < tree_count; t++) {
      double s = 0;
      for (int i = 0; i < N - 1; i++) s += out_branch[(size_t)t * N + i] * branch_lengths[(size_t)t * node_count + i];
      out_clock[t] = s;
    }
  }
  return BITO_AMD_OK;
}

// ---- time trees -------------------------------------------------------------
namespace {

template <typename T>
int ToDevice(Worker* e, DeviceBuffer<T>& buf, const T* host, size_t count) {
  HIP_TRY(e, buf.Reserve(count));
  HIP_TRY(e, hipMemcpyAsync(buf.ptr, host, count * sizeof(T), hipMemcpyHostToDevice, e->stream));
  return BITO_AMD_OK;
}

int ToHost(Worker* e, double* host, const double* dev, size_t count) {
  HIP_TRY(e, hipMemcpyAsync(host, dev, count * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return BITO_AMD_OK;
}

// common front end of the stand-alone transforms: validate, stage the topologies
int StageTimeTrees(Worker* e, int32_t tree_count, const int32_t* parent_ids) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (tree_count < 1 || !parent_ids) return Fail(e, BITO_AMD_ERR_BAD_ARG, "need at least one tree and parent_ids");
  const int N = 2 * e->n - 1;
  int rc = ValidateTrees(e, tree_count, 1, N, parent_ids);
  if (rc) return rc;
  HIP_TRY(e, hipSetDevice(e->device));
  return ToDevice(e, e->tt_parents, parent_ids, (size_t)tree_count * (N - 1));
}

}  // namespace

int WorkerTimeTreesFromBranchLengths(Worker* e, int32_t tree_count,
                                                   const int32_t* parent_ids, const double* branch_lengths,
                                                   const double* tip_dates, double* out_node_bounds,
                                                   double* out_node_heights, double* out_height_ratios) {
  int rc = StageTimeTrees(e, tree_count, parent_ids);
  if (rc) return rc;
  if (!branch_lengths || !tip_dates || !out_node_bounds || !out_node_heights || !out_height_ratios)
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "NULL argument");
  const int n = e->n, N = 2 * n - 1;
  const size_t T = tree_count;
  if ((rc = ToDevice(e, e->tt_in, branch_lengths, T * N))) return rc;
  if ((rc = ToDevice(e, e->tt_aux, tip_dates, (size_t)n))) return rc;
  HIP_TRY(e, e->tt_bounds.Reserve(T * N));
  HIP_TRY(e, e->tt_heights.Reserve(T * N));
  HIP_TRY(e, e->tt_ratios.Reserve(T * (n - 1)));
  HIP_TRY(e, e->tt_out.Reserve(T));
  LaunchTimeTreeFromBranchLengths(tree_count, n, e->tt_parents.ptr, e->tt_in.ptr, e->tt_aux.ptr, e->tt_bounds.ptr,
                                  e->tt_heights.ptr, e->tt_ratios.ptr, e->tt_out.ptr, e->stream);
  HIP_TRY(e, hipGetLastError());
  std::vector<double> diff(T);
  if ((rc = ToHost(e, diff.data(), e->tt_out.ptr, T))) return rc;
  for (size_t t = 0; t < T; t++)
    if (!(diff[t] <= 1e-4)) {  // BRANCH_LENGTH_TOLERANCE, rooted_tree.cpp:7
      char buf[200];
      std::snprintf(buf, sizeof(buf),
                    "Tree isn't time-calibrated in RootedTree::InitializeTimeTreeUsingBranchLengths. "
                    "Height difference: %f (tree %zu)", diff[t], t + (size_t)e->id_offset);
      return Fail(e, BITO_AMD_ERR_BAD_TREE, buf);
    }
  if ((rc = ToHost(e, out_node_bounds, e->tt_bounds.ptr, T * N))) return rc;
  if ((rc = ToHost(e, out_node_heights, e->tt_heights.ptr, T * N))) return rc;
  return ToHost(e, out_height_ratios, e->tt_ratios.ptr, T * (n - 1));
}

int WorkerTimeTreesFromHeightRatios(Worker* e, int32_t tree_count,
                                                  const int32_t* parent_ids, const double* node_bounds,
                                                  const double* height_ratios, double* out_node_heights,
                                                  double* out_branch_lengths) {
  int rc = StageTimeTrees(e, tree_count, parent_ids);
  if (rc) return rc;
  if (!node_bounds || !height_ratios || !out_node_heights || !out_branch_lengths)
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "NULL argument");
  const int n = e->n, N = 2 * n - 1;
  const size_t T = tree_count;
  if ((rc = ToDevice(e, e->tt_bounds, node_bounds, T * N))) return rc;
  if ((rc = ToDevice(e, e->tt_ratios, height_ratios, T * (n - 1)))) return rc;
  HIP_TRY(e, e->tt_heights.Reserve(T * N));
  HIP_TRY(e, e->tt_in.Reserve(T * N));
  LaunchTimeTreeFromRatios(tree_count, n, e->tt_parents.ptr, e->tt_bounds.ptr, e->tt_ratios.ptr, e->tt_heights.ptr,
                           e->tt_in.ptr, e->stream);
  HIP_TRY(e, hipGetLastError());
  if ((rc = ToHost(e, out_node_heights, e->tt_heights.ptr, T * N))) return rc;
  return ToHost(e, out_branch_lengths, e->tt_in.ptr, T * N);
}

int WorkerLogDetJacobian(Worker* e, int32_t tree_count, const int32_t* parent_ids,
                                     const double* node_heights, const double* node_bounds, double* out) {
  int rc = StageTimeTrees(e, tree_count, parent_ids);
  if (rc) return rc;
  if (!node_heights || !node_bounds || !out) return Fail(e, BITO_AMD_ERR_BAD_ARG, "NULL argument");
  const int n = e->n, N = 2 * n - 1;
  const size_t T = tree_count;
  if ((rc = ToDevice(e, e->tt_heights, node_heights, T * N))) return rc;
  if ((rc = ToDevice(e, e->tt_bounds, node_bounds, T * N))) return rc;
  HIP_TRY(e, e->tt_out.Reserve(T));
  LaunchLogDetJacobian(tree_count, n, e->tt_parents.ptr, e->tt_heights.ptr, e->tt_bounds.ptr, e->tt_out.ptr, nullptr,
                       e->stream);
  HIP_TRY(e, hipGetLastError());
  return ToHost(e, out, e->tt_out.ptr, T);
}

// shared by the two stand-alone ratio-space transforms (mode 0 / 1 of ratio_gradient_kernel)
static int RatioTransform(Worker* e, int mode, int32_t tree_count, const int32_t* parent_ids,
                          const double* node_heights, const double* node_bounds, const double* height_ratios,
                          const double* height_gradient, double* out) {
  int rc = StageTimeTrees(e, tree_count, parent_ids);
  if (rc) return rc;
  if (!node_heights || !node_bounds || !height_ratios || !out || (mode == 0 && !height_gradient))
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "NULL argument");
  const int n = e->n, N = 2 * n - 1;
  const size_t T = tree_count;
  if ((rc = ToDevice(e, e->tt_heights, node_heights, T * N))) return rc;
  if ((rc = ToDevice(e, e->tt_bounds, node_bounds, T * N))) return rc;
  if ((rc = ToDevice(e, e->tt_ratios, height_ratios, T * (n - 1)))) return rc;
  if (mode == 0 && (rc = ToDevice(e, e->tt_in, height_gradient, T * (n - 1)))) return rc;
  HIP_TRY(e, e->tt_work.Reserve(T * 3 * (n - 1)));
  HIP_TRY(e, e->tt_out.Reserve(T * (n - 1)));
  LaunchRatioGradient(tree_count, n, mode, e->tt_parents.ptr, e->tt_heights.ptr, e->tt_bounds.ptr, e->tt_ratios.ptr,
                      e->tt_in.ptr, n - 1, nullptr, e->tt_work.ptr, e->tt_out.ptr, e->stream);
  HIP_TRY(e, hipGetLastError());
  return ToHost(e, out, e->tt_out.ptr, T * (n - 1));
}

int WorkerGradientLogDetJacobian(Worker* e, int32_t tree_count, const int32_t* parent_ids,
                                              const double* node_heights, const double* node_bounds,
                                              const double* height_ratios, double* out) {
  return RatioTransform(e, 1, tree_count, parent_ids, node_heights, node_bounds, height_ratios, nullptr, out);
}

int WorkerRatioGradientOfHeightGradient(Worker* e, int32_t tree_count,
                                                      const int32_t* parent_ids, const double* node_heights,
                                                      const double* node_bounds, const double* height_ratios,
                                                      const double* height_gradient, double* out) {
  return RatioTransform(e, 0, tree_count, parent_ids, node_heights, node_bounds, height_ratios, height_gradient,
                        out);
}

int WorkerTimeTreeLogLikelihoods(Worker* e, int32_t tree_count, const int32_t* parent_ids,
                                              const double* branch_lengths, const double* rates,
                                              const double* node_heights, const double* node_bounds,
                                              const double* params, int32_t rescaling,
                                              int32_t include_log_det_jacobian, double* out) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  const int n = e->n, N = 2 * n - 1;
  if (include_log_det_jacobian && (!node_heights || !node_bounds))
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "node_heights / node_bounds are needed for the log-det-Jacobian");
  int rc = WorkerUpload(e, tree_count, 1, N, parent_ids, branch_lengths, rates, params);
  if (rc) return rc;
  if ((rc = RunResident(e, 0, rescaling != 0))) return rc;
  if (include_log_det_jacobian) {
    const size_t T = tree_count;
    if ((rc = ToDevice(e, e->tt_heights, node_heights, T * N))) return rc;
    if ((rc = ToDevice(e, e->tt_bounds, node_bounds, T * N))) return rc;
    LaunchLogDetJacobian(tree_count, n, e->parent_ids.ptr, e->tt_heights.ptr, e->tt_bounds.ptr, nullptr,
                         e->cur_ll(), e->stream);
    HIP_TRY(e, hipGetLastError());
  }
  return WorkerDownload(e, out, nullptr);
}

int WorkerTimeTreeGradients(Worker* e, int32_t tree_count, const int32_t* parent_ids,
                                        const double* branch_lengths, const double* rates, int32_t rate_count,
                                        const double* node_heights, const double* node_bounds,
                                        const double* height_ratios, const double* params, int32_t rescaling,
                                        int32_t flags, double fd_delta, double* out_ll, double* out_branch,
                                        double* out_site, double* out_subst, double* out_clock,
                                        double* out_ratios) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  const int n = e->n, N = 2 * n - 1;
  const size_t T = tree_count;
  const bool want_clock = (flags & BITO_AMD_GRAD_CLOCK_MODEL) && out_clock;
  const bool want_ratios = (flags & BITO_AMD_GRAD_RATIOS_ROOT_HEIGHT) && out_ratios;
  if (want_clock && rate_count != 1 && rate_count != N - 1)
    return Fail(e, BITO_AMD_ERR_BAD_ARG,
                "The number of rates should be equal to 1 (i.e. strict clock) or equal to the number of branches.");
  if (want_ratios && (!node_heights || !node_bounds || !height_ratios))
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "time trees are not initialised: node_heights / node_bounds / height_ratios");
  // everything except the clock and ratio outputs; leaves the main pass resident on the device
  int rc = WorkerGradients(e, tree_count, 1, N, parent_ids, branch_lengths, rates, params, rescaling,
                                     flags & ~BITO_AMD_GRAD_CLOCK_MODEL, fd_delta, out_ll, out_branch, out_site,
                                     out_subst, nullptr);
  if (rc) return rc;
  if (want_clock) {
    const size_t count = T * (rate_count == 1 ? 1 : N - 1);
    HIP_TRY(e, e->tt_out.Reserve(count));
    LaunchClockGradient(tree_count, N, rate_count, e->out_grad.ptr, e->branch_in.ptr, N, e->tt_out.ptr, e->stream);
    HIP_TRY(e, hipGetLastError());
    if ((rc = ToHost(e, out_clock, e->tt_out.ptr, count))) return rc;
  }
  if (want_ratios) {
    if ((rc = ToDevice(e, e->tt_heights, node_heights, T * N))) return rc;
    if ((rc = ToDevice(e, e->tt_bounds, node_bounds, T * N))) return rc;
    if ((rc = ToDevice(e, e->tt_ratios, height_ratios, T * (n - 1)))) return rc;
    HIP_TRY(e, e->tt_work.Reserve(T * 3 * (n - 1)));
    HIP_TRY(e, e->tt_out.Reserve(T * (n - 1)));
    const int mode = 2 | ((flags & BITO_AMD_GRAD_LOG_DET_JACOBIAN_GRADIENT) ? 4 : 0);
    LaunchRatioGradient(tree_count, n, mode, e->parent_ids.ptr, e->tt_heights.ptr, e->tt_bounds.ptr, e->tt_ratios.ptr,
                        e->out_grad.ptr, N, e->has_rates ? e->rates.ptr : nullptr, e->tt_work.ptr, e->tt_out.ptr,
                        e->stream);
    HIP_TRY(e, hipGetLastError());
    if ((rc = ToHost(e, out_ratios, e->tt_out.ptr, T * (n - 1)))) return rc;
  }
  return BITO_AMD_OK;
}

}  // namespace bito_amd

extern "C" int bito_amd_plan_pipe_walk(int32_t taxon_count, int32_t pattern_count, int32_t category_count, int32_t tree_count,
                            int32_t min_cherries, int32_t plan[7]) {
  if (!plan || taxon_count < 3 || pattern_count < 1 || tree_count < 1) return BITO_AMD_ERR_BAD_ARG;
  BatchDims d{};
  d.taxon_count = taxon_count;
  d.node_count = 2 * taxon_count - 1;
  d.in_node_count = 2 * taxon_count - 2;
  d.pattern_count = pattern_count;
  d.pattern_stride = (pattern_count + 512 + kHbmBlock - 1) / kHbmBlock * kHbmBlock;
  d.category_count = category_count;
  d.tree_count = tree_count;
  d.min_cherries = min_cherries;
  d.min_unstored = min_cherries;
  d.pipe_fold = 0;
  const LdsPlan p = PlanPipe(d);
  const int32_t out[7] = {p.groups, p.patterns_per_block, p.tiles, (int32_t)p.lds_bytes, p.tile_run, p.whole_trees, p.slots};
  for (int k = 0; k < 7; k++) plan[k] = out[k];
  return BITO_AMD_OK;
}

// nodes of each tree that walk_pipe_kernel keeps no vector for (cherries, and with `fold` the pitchforks it folds): the
// host's count, which sizes a tree's LDS slots and must be the step tables' own (tests hold it to a restatement)
extern "C" int bito_amd_count_unstored_nodes(int32_t taxon_count, int32_t tree_count, int32_t rooted, int32_t node_count,
                                             const int32_t* parent_ids, int32_t fold, int32_t* out) {
  if (!parent_ids || !out || taxon_count < 3 || tree_count < 1) return BITO_AMD_ERR_BAD_ARG;
  if (node_count != (rooted ? 2 * taxon_count - 1 : 2 * taxon_count - 2)) return BITO_AMD_ERR_BAD_ARG;
  // the rows come from the caller: the same range and shape checks as every other entry point (ValidateTreesRange)
  // before UnstoredNodes indexes by them -- parent ids in [n, M) above their child, two children per internal node,
  // three at an unrooted tree's root
  const int n = taxon_count, M = node_count;
  std::vector<int> kids, count(M);
  for (int t = 0; t < tree_count; t++) {
    const int32_t* par = parent_ids + (size_t)t * (M - 1);
    std::fill(count.begin(), count.end(), 0);
    for (int child = 0; child < M - 1; child++) {
      const int p = par[child];
      if (p < n || p >= M || p <= child) return BITO_AMD_ERR_BAD_TREE;
      count[p]++;
    }
    for (int i = n; i < M; i++)
      if (count[i] != ((!rooted && i == M - 1) ? 3 : 2)) return BITO_AMD_ERR_BAD_TREE;
    out[t] = UnstoredNodes(n, M, rooted, par, fold != 0, &kids);
  }
  return BITO_AMD_OK;
}

namespace bito_amd {

int WorkerSetKernel(Worker* e, int32_t kernel) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  // a resident batch's class split (which trees take the two-wave form, which share a launch plan) was decided under the
  // old choice: the next pass decides again, as it does after an update of the batch's values
  if (kernel != e->kernel_choice) e->pipe_split = Worker::PipeSplit{};
  e->kernel_choice = kernel;
  return BITO_AMD_OK;
}

int WorkerKernelTiming(Worker* e, int32_t enable) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  e->timing = enable != 0;
  e->ev_used = 0;
  return BITO_AMD_OK;
}

int WorkerReadGeneralModel(Worker* e, int32_t tree, double* out, size_t capacity) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  const double* gs_model = e->gs_model.ptr;
  if (!e->resident || tree < 0 || tree >= e->dims.tree_count || !gs_model || !e->gs_model_index.ptr)
    return Fail(e, BITO_AMD_ERR_STATE, "no general-state model is resident for that tree");
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  const size_t count = std::min<size_t>(capacity, (size_t)kGsModelStride);
  int32_t slot = tree;
  HIP_TRY(e, hipMemcpy(&slot, e->gs_model_index.ptr + tree, sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(e, hipMemcpy(out, gs_model + (size_t)slot * kGsModelStride, count * sizeof(double), hipMemcpyDeviceToHost));
  return BITO_AMD_OK;
}

const char* WorkerKernelName(const Worker* e) { return e ? e->kernel_name.c_str() : ""; }

int WorkerTimeRuns(Worker* e, int32_t want_gradient, int32_t rescaling,
                              int32_t steps, double* total_ms, double* kernel_ms,
                              int32_t* kernel_launches) {
  if (!e || steps < 1) return BITO_AMD_ERR_BAD_ARG;
  if (!e->resident) return Fail(e, BITO_AMD_ERR_STATE, "no batch is resident");
  HIP_TRY(e, hipSetDevice(e->device));
  e->timing = true;
  e->ev_used = 0;
  hipEvent_t t0 = NextEvent(e), t1 = NextEvent(e);
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  HIP_TRY(e, hipEventRecord(t0, e->stream));
  int rc = BITO_AMD_OK;
  for (int s = 0; s < steps && !rc; s++) rc = RunResident(e, want_gradient != 0, rescaling != 0);
  e->timing = false;
  if (rc) return rc;
  HIP_TRY(e, hipEventRecord(t1, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  float ms = 0;
  HIP_TRY(e, hipEventElapsedTime(&ms, t0, t1));
  if (total_ms) *total_ms = ms;
  double k = 0;
  int launches = 0;
  for (size_t i = 2; i + 1 < e->ev_used; i += 2) {
    float kms = 0;
    HIP_TRY(e, hipEventElapsedTime(&kms, e->ev_pool[i], e->ev_pool[i + 1]));
    k += kms;
    launches++;
  }
  if (kernel_ms) *kernel_ms = k;
  if (kernel_launches) *kernel_launches = launches;
  return BITO_AMD_OK;
}

}  // namespace bito_amd
