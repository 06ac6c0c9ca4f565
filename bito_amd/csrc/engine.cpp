// engine.cpp -- the C ABI declared in include/bito_amd.h.
//
// One engine drives one or more GPUs from one host thread, as the reference's Engine drives
// thread_count FatBeagle instances (src/engine.cpp:10-31, src/fat_beagle.hpp:151-184): a tree
// collection is cut into contiguous blocks, one per device, and every block into a few chunks
// that travel through the device one behind the other -- each chunk on a worker of its own
// (worker.cpp: streams, pinned staging, resident buffers), so that while chunk k is being
// traversed the host validates and stages chunk k+1 and the results of chunk k-1 come back.
// Results land in the caller's arrays at the trees' own positions; nothing depends on which
// chunk or device a tree went to except the order of the pattern-tile sums (rounding level,
// see INTEGRATION.md).
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "host_pool.hpp"
#include "worker.hpp"

using namespace bito_amd;

namespace {

// a block of the resident batch: trees [t0, t0 + count) of the caller's collection live on worker (slot, lane)
struct Shard {
  int slot, lane;
  int t0, count;
};

constexpr int kMaxLanes = 8;

}  // namespace

struct bito_amd_engine {
  std::vector<int> devices;  // HIP ordinal per device slot (a list may name a device twice: two slots on one GPU)
  std::vector<std::vector<std::unique_ptr<Worker>>> workers;  // [slot][lane]; lanes beyond 0 are created on first use
  // what a worker is created from
  std::string substitution, site, clock;
  int n = 0, P = 0;
  std::vector<int32_t> patterns;
  std::vector<double> weights;
  uint64_t arena_bytes = 0;
  int kernel_choice = BITO_AMD_KERNEL_AUTO;
  bool timing = false;
  // the resident batch
  bool resident = false;
  int rooted = 0, node_count = 0, tree_count = 0;
  std::vector<Shard> shards;
  std::string err;
  // chunking of blocking calls (BITO_AMD_CHUNK_FIRST / _GROWTH / _CAP / _LANES / _RESERVE: measurements;
  // scripts/gpu_chunk_sweep.sh: 4.24 ms per 6400 config-3 trees with these, 4.9 with first = 128, growth = 1.5)
  int chunk_first = 512, chunk_cap = 2048, max_lanes = 8, reserve_cus = 0;
  double chunk_growth = 3.0;
  int walk_streams = 2, chunk_taper = 1;
  // host threads of a blocking call (host_pool.hpp): chunks of par_min_trees trees and more are checked and packed
  // in ranges, one per thread, and large result blocks copied out the same way.  Created on first use.
  // (BITO_AMD_HOST_THREADS: 1 = the calling thread alone; default min(8, CPUs this process may use))
  int host_threads = 0, par_min_trees = 1024;
  size_t par_min_bytes = (size_t)512 << 10;
  std::unique_ptr<HostPool> pool;
  // one issuing thread per device slot beyond the first (the reference runs a thread per FatBeagle instance,
  // src/task_processor.hpp:43-140): a call over several devices hands every slot's block to its own thread, so that no
  // device waits for the host to be done with another one's.  Created on the first call that uses several slots.
  std::unique_ptr<HostPool> slot_pool;
  // ... and the helper threads those issuing threads share (host_pool.hpp, SharedPool): a slot's chunk of a thousand
  // trees and more is checked and packed in ranges that the helpers and the slot's own thread claim one by one, a large
  // result block copied out the same way.  host_threads - 1 helpers; created with slot_pool.
  std::unique_ptr<SharedPool> shared_pool;
  double span_sum_ms = 0;  // of the last bito_amd_engine_kernel_elapsed: the launches' spans added up (overlaps counted twice)
};

namespace {

int Fail(bito_amd_engine* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  return code;
}

Worker* Primary(const bito_amd_engine* e) { return e->workers[0][0].get(); }

// worker (slot, lane), created on first use
int GetWorker(bito_amd_engine* e, int slot, int lane, Worker** out) {
  auto& lanes = e->workers[slot];
  if ((int)lanes.size() <= lane) lanes.resize(lane + 1);
  if (!lanes[lane]) {
    Worker* w = nullptr;
    std::string msg;
    const int rc = WorkerCreate(e->devices[slot], e->arena_bytes, e->substitution.c_str(), e->site.c_str(),
                                e->clock.c_str(), e->n, e->P, e->patterns.data(), e->weights.data(), &w, &msg);
    if (rc) return Fail(e, rc, msg);
    WorkerSetKernel(w, e->kernel_choice);
    if (e->timing) WorkerKernelTiming(w, 1);
    lanes[lane].reset(w);
  }
  *out = lanes[lane].get();
  return BITO_AMD_OK;
}

Worker* ShardWorker(const bito_amd_engine* e, const Shard& s) { return e->workers[s.slot][s.lane].get(); }

int Propagate(bito_amd_engine* e, Worker* w, int rc) {
  if (rc) e->err = WorkerLastError(w);
  return rc;
}

// CPUs this process may use: the affinity mask and the cgroup's quota (a container on a 256-thread host is typically
// given far fewer), not std::thread::hardware_concurrency().
int UsableCpus() {
  int cpus = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32];
    long period = 0;
    if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0)
      cpus = std::min<int>(cpus, (int)((std::atol(quota) + period / 2) / period));
    std::fclose(f);
  }
  return std::max(1, cpus);
}

HostPool* Pool(bito_amd_engine* e) {
  if (!e->pool) {
    e->pool = std::make_unique<HostPool>(std::max(0, e->host_threads - 1));
  }
  return e->pool.get();
}

SharedPool* Shared(bito_amd_engine* e) {
  // (helpers poll beside one issuing thread per device slot: together they must not outnumber the CPUs this process
  // may use, or the helpers take the time slices of the very threads they are there to relieve)
  if (!e->shared_pool)
    e->shared_pool = std::make_unique<SharedPool>(
        std::max(0, std::min(e->host_threads - 1, UsableCpus() - (int)e->devices.size())));
  return e->shared_pool.get();
}

// fn(part, begin, end) over [0, count) in `parts` contiguous ranges, claimed one by one by the calling thread and the
// shared helpers (several issuing threads may be in here at once)
void SharedRanges(bito_amd_engine* e, size_t count, int parts, const std::function<void(int, size_t, size_t)>& fn) {
  Shared(e)->Run(parts, [&](int part) {
    const size_t a = count * (size_t)part / (size_t)parts, b = count * ((size_t)part + 1) / (size_t)parts;
    if (b > a) fn(part, a, b);
  });
}

// fn(begin, end) over [0, count) in one contiguous range per host thread
void ParallelRanges(bito_amd_engine* e, size_t count, const std::function<void(int, size_t, size_t)>& fn) {
  HostPool* pool = Pool(e);
  const size_t parts = (size_t)pool->parts();
  pool->Run([&](int part) {
    const size_t a = count * (size_t)part / parts, b = count * ((size_t)part + 1) / parts;
    if (b > a) fn(part, a, b);
  });
}

// fn(slot, first tree, trees, the slot's first worker) for every device slot's contiguous block of a collection of
// tree_count trees, each slot on a host thread of its own (slot 0 on the caller's).  fn returns a C-ABI code and, when
// it fails, leaves the message with its worker; what is reported is the failure of the slot with the lowest trees -- the
// one a serial pass would have met first.  Blocks: slot s takes trees [T s / D, T (s + 1) / D).
int RunPerSlot(bito_amd_engine* e, int tree_count, const std::function<int(int, int, int, Worker*)>& fn) {
  const int D = (int)e->devices.size();
  for (int s = 0; s < D; s++) {
    Worker* w = nullptr;
    if (int rc = GetWorker(e, s, 0, &w)) return rc;
  }
  if (!e->slot_pool) e->slot_pool = std::make_unique<HostPool>(D - 1);
  std::vector<int> codes((size_t)D, BITO_AMD_OK);
  e->slot_pool->Run([&](int slot) {
    const int t0 = (int)((long long)tree_count * slot / D), t1 = (int)((long long)tree_count * (slot + 1) / D);
    if (t1 == t0) return;
    Worker* w = e->workers[slot][0].get();
    (void)hipSetDevice(e->devices[slot]);
    w->id_offset = t0;
    codes[(size_t)slot] = fn(slot, t0, t1 - t0, w);
  });
  for (int s = 0; s < D; s++)
    if (codes[(size_t)s]) return Propagate(e, e->workers[s][0].get(), codes[(size_t)s]);
  return BITO_AMD_OK;
}

// the resident batch after a call that left one block on every slot's first worker
void SetSlotResident(bito_amd_engine* e, int rooted, int node_count, int tree_count) {
  const int D = (int)e->devices.size();
  e->shards.clear();
  for (int s = 0; s < D; s++) {
    const int t0 = (int)((long long)tree_count * s / D), t1 = (int)((long long)tree_count * (s + 1) / D);
    if (t1 > t0) e->shards.push_back({s, 0, t0, t1 - t0});
  }
  e->resident = true;
  e->rooted = rooted;
  e->node_count = node_count;
  e->tree_count = tree_count;
}

// Chunk sizes for `count` trees on one device.  The first chunk is small, so that the device starts early; the
// following ones grow, so that the host -- which validates and stages about five times faster than the device
// traverses -- stays a chunk ahead and the set-up kernels of chunk k+1 are in the queue before the traversal of
// chunk k needs every CU.  Heavy trees (large n x P x C, or the 61-state model) go as one chunk: their traversal
// takes so long per tree that there is nothing to hide, and one launch fills the PLV arena best.
std::vector<int> PlanChunks(const bito_amd_engine* e, int count, bool single) {
  std::vector<int> sizes;
  const Worker* w = Primary(e);
  const size_t work = (size_t)e->n * e->P * w->spec.category_count;
  if (single || w->spec.state_count != 4 || work > ((size_t)1 << 20) || count < 2 * e->chunk_first) {
    sizes.push_back(count);
    return sizes;
  }
  int remaining = count;
  double c = e->chunk_first;
  while (remaining > 0) {
    int take = std::min<int>((int)c, remaining);
    if ((int)sizes.size() == e->max_lanes - 1 || remaining - take < take / 2) take = remaining;
    sizes.push_back(take);
    remaining -= take;
    c = std::min<double>(c * e->chunk_growth, e->chunk_cap);
  }
  // ... and the last chunk small again: what the caller waits for at the very end -- the final sums of the last chunk
  // written over PCIe, and the host's copy of them into the caller's arrays -- is in proportion to its size
  if (e->chunk_taper && sizes.size() >= 2 && sizes.back() >= 3 * e->chunk_first && (int)sizes.size() < e->max_lanes) {
    sizes.back() -= e->chunk_first;
    sizes.push_back(e->chunk_first);
  }
  return sizes;
}

// The blocks of a collection: contiguous per device slot, then chunks per slot; listed in the order they are
// issued (chunk 0 of every slot, then chunk 1 of every slot, ...), so that every device starts early.
int PlanShards(bito_amd_engine* e, int tree_count, bool single, std::vector<Shard>* out) {
  out->clear();
  const int D = single ? 1 : (int)e->devices.size();
  std::vector<std::vector<Shard>> per_slot(D);
  size_t most = 0;
  for (int s = 0; s < D; s++) {
    const int t0 = (int)((long long)tree_count * s / D), t1 = (int)((long long)tree_count * (s + 1) / D);
    if (t1 == t0) continue;
    int at = t0, lane = 0;
    for (int size : PlanChunks(e, t1 - t0, single)) {
      per_slot[s].push_back({s, lane++, at, size});
      at += size;
    }
    most = std::max(most, per_slot[s].size());
  }
  for (size_t k = 0; k < most; k++)
    for (int s = 0; s < D; s++)
      if (k < per_slot[s].size()) out->push_back(per_slot[s][k]);
  for (const Shard& s : *out) {
    Worker* w = nullptr;
    if (int rc = GetWorker(e, s.slot, s.lane, &w)) return rc;
  }
  return BITO_AMD_OK;
}

// The blocking evaluation behind bito_amd_engine_log_likelihoods / _gradients: stage, run and fetch every chunk
// without waiting, copy results out as they arrive.
int Evaluate(bito_amd_engine* e, int32_t tree_count, int32_t rooted, int32_t node_count, const int32_t* parent_ids,
             const double* branch_lengths, const double* rates, const double* params, int rescaling,
             int want_gradient, int want_site, double* out_ll, double* out_grad, double* out_site, bool single) {
  e->resident = false;
  if (tree_count < 1 || !parent_ids || !branch_lengths)
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "need at least one tree and non-NULL parent_ids / branch_lengths");
  const Worker* w0 = Primary(e);
  const size_t pc = (size_t)w0->spec.param_count, M = (size_t)node_count, N = 2 * (size_t)e->n - 1;
  if (pc > 0 && !params) return Fail(e, BITO_AMD_ERR_BAD_ARG, "params is NULL but the model has parameters");
  if (int rc = PlanShards(e, tree_count, single, &e->shards)) return rc;
  const bool has_rates = rooted && rates != nullptr;
  // One issuing thread per device slot (the reference runs a thread per FatBeagle instance,
  // src/task_processor.hpp:43-140): with several slots every slot's chunks are staged, issued and drained by the slot's
  // own thread, so that no device waits for the host to be done with another one's -- slot 7 of an eight-GPU engine
  // would otherwise see its first chunk issued behind seven other stage + launch sequences.  A one-slot call (the
  // headline path) stays on the calling thread and cuts large chunks into ranges over the helper threads instead.
  // (BITO_AMD_SLOT_THREADS=0: every slot from the calling thread, chunk k of every slot before chunk k + 1 of any.)
  static const bool slot_threads = [] {
    const char* v = std::getenv("BITO_AMD_SLOT_THREADS");
    return v == nullptr || std::atoi(v) != 0;
  }();
  std::vector<std::vector<size_t>> lists;
  {
    int slots_used = 0;
    std::vector<char> seen(e->devices.size(), 0);
    for (const Shard& s : e->shards)
      if (!seen[(size_t)s.slot]) {
        seen[(size_t)s.slot] = 1;
        slots_used++;
      }
    if (slots_used > 1 && slot_threads) {
      lists.resize(e->devices.size());
      for (size_t k = 0; k < e->shards.size(); k++) lists[(size_t)e->shards[k].slot].push_back(k);
    } else {
      lists.resize(1);
      for (size_t k = 0; k < e->shards.size(); k++) lists[0].push_back(k);
    }
  }
  const bool threaded = lists.size() > 1;
  // (BITO_AMD_TRACE_CALL=1: host-side time line of the call on stderr -- when each chunk was issued and when its
  // results had arrived)
  static const bool trace = std::getenv("BITO_AMD_TRACE_CALL") != nullptr;
  const auto call_start = std::chrono::steady_clock::now();
  auto stamp = [&](const char* what, size_t k) {
    if (trace)
      std::fprintf(stderr, "  %8.1f us  %s chunk %zu (slot %d lane %d, %d trees)\n",
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - call_start).count(), what, k,
                   e->shards[k].slot, e->shards[k].lane, e->shards[k].count);
  };
  auto big_block = [&](const Shard& s) {
    return e->host_threads != 1 &&
           (size_t)s.count * (want_gradient && out_grad ? N + 1 : 1) * sizeof(double) >= e->par_min_bytes;
  };
  auto big_copy = [&](const Shard& s) { return !threaded && big_block(s); };
  int par_min_trees = e->par_min_trees;  // (this call's threshold: lowered below while the one-slot helpers still poll)
  // several issuing threads: ranges of half the threshold (512 trees), claimed by the thread itself and the shared helpers
  auto shared_parts = [&](int count) {
    return std::max(1, std::min(Shared(e)->helpers() + 1, count / std::max(1, par_min_trees / 2)));
  };
  auto drain = [&](size_t k) -> int {
    const Shard& s = e->shards[k];
    Worker* w = ShardWorker(e, s);
    const double *ll = nullptr, *grad = nullptr, *site = nullptr;
    // (a large block: the helper threads are woken now and poll for the copy while this thread polls for the results)
    if (big_copy(s) && !WorkerResultsReady(w)) Pool(e)->Arm();
    if (int rc = WorkerResults(w, &ll, &grad, &site)) return rc;
    stamp("results of", k);
    const bool with_site = want_site && out_site && w->site_ready;
    auto copy = [&](size_t a, size_t b) {  // trees [a, b) of the chunk
      std::memcpy(out_ll + s.t0 + a, ll + a, (b - a) * sizeof(double));
      if (want_gradient && out_grad) std::memcpy(out_grad + ((size_t)s.t0 + a) * N, grad + a * N, (b - a) * N * sizeof(double));
      if (with_site) std::memcpy(out_site + s.t0 + a, site + a, (b - a) * sizeof(double));
    };
    if (big_copy(s))
      ParallelRanges(e, (size_t)s.count, [&](int, size_t a, size_t b) { copy(a, b); });
    else if (threaded && big_block(s) && shared_parts(s.count) > 1)
      SharedRanges(e, (size_t)s.count, shared_parts(s.count), [&](int, size_t a, size_t b) { copy(a, b); });
    else
      copy(0, (size_t)s.count);
    stamp("copied out", k);
    return BITO_AMD_OK;
  };
  // a call that will hand ranges to the helper threads wakes them now: they are up by the time the first chunk is staged.
  // (In a loop of calls they still poll from the call before: then smaller chunks -- the first one of a large call --
  // are worth cutting up as well.)
  if (!threaded && e->host_threads != 1 && tree_count >= e->par_min_trees) {
    if (Pool(e)->Hot()) par_min_trees = std::min(par_min_trees, 256);
    Pool(e)->Arm();
  }
  // (several issuing threads: a shorter poll -- the staging of a call's large chunks is over within 2 ms, and every
  // published range re-arms the helpers for kLingerNs)
  if (threaded && e->host_threads != 1) Shared(e)->Arm(std::chrono::microseconds(2000));
  // what one issuing thread does with its chunks, in order; returns the first failure (the failing worker holds the message)
  struct Outcome {
    int rc = BITO_AMD_OK;
    Worker* worker = nullptr;
    int t0 = 0;
  };
  auto run_list = [&](const std::vector<size_t>& mine, Outcome* oc) {
    size_t drained = 0;
    std::vector<char> slot_busy(e->devices.size(), 0);
    auto fail = [&](int rc, size_t k, size_t issued) {
      oc->rc = rc;
      oc->worker = ShardWorker(e, e->shards[k]);
      oc->t0 = e->shards[k].t0;
      for (size_t i = 0; i < issued && i < mine.size(); i++) (void)WorkerSync(ShardWorker(e, e->shards[mine[i]]));
    };
    for (size_t i = 0; i < mine.size(); i++) {
      const size_t k = mine[i];
      const Shard& s = e->shards[k];
      Worker* w = ShardWorker(e, s);
      if (threaded) (void)hipSetDevice(e->devices[(size_t)s.slot]);
      w->one_shot = slot_busy[s.slot] ? 2 : 1;
      // The slot's stream pair: every chunk's traversal, final sums and completion flag on the first worker's
      // stream, in order; the copies and set-up kernels of the later chunks on its (low-priority) set-up stream,
      // beside the traversal of the chunk before.  Two streams per device, whatever the number of chunks: the
      // runtime multiplexes streams onto four hardware queues, and with a stream per chunk the chunks' commands
      // queued up behind one another's traversals (measured: copies waiting a millisecond).
      // (walk_streams == 2: the chunks' traversals alternate between the first two workers' streams, so that chunk
      // k+1's resident workgroups move in as chunk k's leave -- the ragged end of one launch filled by the next)
      Worker* first = e->workers[s.slot][0].get();
      Worker* second = (e->walk_streams > 1 && e->workers[s.slot].size() > 1) ? e->workers[s.slot][1].get() : first;
      w->lent_walk = s.lane > 0 ? ((s.lane & 1) ? second->stream : first->stream) : nullptr;
      w->lent_setup = s.lane > 0 ? first->prep_stream : nullptr;
      w->id_offset = s.t0;
      slot_busy[s.slot] = 1;
      // a traversal holds every CU it is given until its queue of work is empty: while a later chunk of this device
      // still has its set-up kernels to run, it leaves them one CU per XCD
      w->reserve_cus = 0;
      for (size_t later = k + 1; later < e->shards.size(); later++)
        if (e->shards[later].slot == s.slot) w->reserve_cus = e->reserve_cus;
      int rc = WorkerStageBegin(w, s.count, rooted, node_count, parent_ids + (size_t)s.t0 * (M - 1),
                                branch_lengths + (size_t)s.t0 * M, has_rates ? rates + (size_t)s.t0 * (M - 1) : nullptr,
                                pc > 0 ? params + (size_t)s.t0 * pc : nullptr, /*wait=*/0);
      if (!rc) {
        // the host's share of the chunk -- checks, one pack into pinned memory -- in ranges over the host threads
        if (!threaded && e->host_threads != 1 && s.count >= par_min_trees) {
          std::vector<StagePart> parts((size_t)Pool(e)->parts());
          ParallelRanges(e, (size_t)s.count, [&](int part, size_t a, size_t b) { WorkerStageFill(w, (int32_t)a, (int32_t)b, &parts[(size_t)part]); });
          rc = WorkerStageEnd(w, parts.data(), (int)parts.size());
        } else if (threaded && e->host_threads != 1 && s.count >= par_min_trees && shared_parts(s.count) > 1) {
          // (round 5: a slot's thread no longer packs its large chunks alone -- eight slots x 5376 trees were 0.33 ms of
          // one thread each, the second chunks issued 0.74-0.78 ms into the call)
          std::vector<StagePart> parts((size_t)shared_parts(s.count));
          SharedRanges(e, (size_t)s.count, (int)parts.size(),
                       [&](int part, size_t a, size_t b) { WorkerStageFill(w, (int32_t)a, (int32_t)b, &parts[(size_t)part]); });
          rc = WorkerStageEnd(w, parts.data(), (int)parts.size());
        } else {
          StagePart part;
          WorkerStageFill(w, 0, s.count, &part);
          rc = WorkerStageEnd(w, &part, 1);
        }
      }
      stamp("staged", k);
      if (!rc) rc = WorkerRunPass(w, want_gradient, rescaling, 0, want_site);
      if (!rc) rc = WorkerFetchResults(w, want_gradient, want_site);
      stamp("issued", k);
      if (rc) return fail(rc, k, i + 1);
      // results that have arrived meanwhile (in order: the chunks finish in the order they were issued, near enough)
      while (drained < i && WorkerResultsReady(ShardWorker(e, e->shards[mine[drained]]))) {
        const size_t kd = mine[drained++];
        if (int rc2 = drain(kd)) return fail(rc2, kd, i + 1);
      }
    }
    for (; drained < mine.size(); drained++)
      if (int rc = drain(mine[drained])) return fail(rc, mine[drained], mine.size());
  };
  std::vector<Outcome> outcomes(lists.size());
  if (!threaded) {
    run_list(lists[0], &outcomes[0]);
  } else {
    if (!e->slot_pool) e->slot_pool = std::make_unique<HostPool>((int)e->devices.size() - 1);
    e->slot_pool->Run([&](int slot) {
      if (!lists[(size_t)slot].empty()) run_list(lists[(size_t)slot], &outcomes[(size_t)slot]);
    });
  }
  {
    const Outcome* bad = nullptr;  // the failure a serial pass over the trees would have met first
    for (const Outcome& oc : outcomes)
      if (oc.rc && (!bad || oc.t0 < bad->t0)) bad = &oc;
    if (bad) return Propagate(e, bad->worker, bad->rc);
  }
  if (want_site && out_site) {
    // kernels that do not produce the site-model gradient in the main pass: a second traversal per block.  Every
    // chunk has drained by now, so each worker runs the pass on its OWN streams and waits for it there: the streams a
    // chunk was lent for the call belong to other workers, and a download ordered behind this worker's stream alone
    // would read pin_out before the lent stream's traversal has written it.
    // With several slots every slot's blocks go from the slot's own thread (round 5: they went slot after slot from the
    // calling thread -- eight GPUs took turns at a pass each could run at once).
    auto second_pass = [&](const std::vector<size_t>& mine, Outcome* oc) {
      for (size_t k : mine) {
        const Shard& s = e->shards[k];
        Worker* w = ShardWorker(e, s);
        if (w->site_ready) continue;
        if (threaded) (void)hipSetDevice(e->devices[(size_t)s.slot]);
        w->one_shot = 0;
        w->lent_walk = w->lent_setup = nullptr;
        w->reserve_cus = 0;
        stamp("second pass (site-model gradient) of", k);
        const int rc = WorkerSiteGradientSecondPass(w, rooted, node_count, branch_lengths + (size_t)s.t0 * M,
                                                    has_rates ? rates + (size_t)s.t0 * (M - 1) : nullptr, rescaling,
                                                    out_site + s.t0);
        stamp("second pass done of", k);
        if (rc) {
          oc->rc = rc;
          oc->worker = w;
          oc->t0 = s.t0;
          return;
        }
      }
    };
    std::vector<Outcome> passes(lists.size());
    if (!threaded)
      second_pass(lists[0], &passes[0]);
    else
      e->slot_pool->Run([&](int slot) {
        if (!lists[(size_t)slot].empty()) second_pass(lists[(size_t)slot], &passes[(size_t)slot]);
      });
    const Outcome* bad = nullptr;
    for (const Outcome& oc : passes)
      if (oc.rc && (!bad || oc.t0 < bad->t0)) bad = &oc;
    if (bad) return Propagate(e, bad->worker, bad->rc);
  }
  e->resident = true;  // (only once nothing can fail any more)
  e->rooted = rooted;
  e->node_count = node_count;
  e->tree_count = tree_count;
  return BITO_AMD_OK;
}

// Trees of 39 to 64 taxa take walk_pipe_kernel only in its one-image-per-branch form, which needs every branch of a
// block well above the rounding error of its transition matrix (worker.cpp) -- and a worker
// decides for its whole block, so ONE tree with a near-zero branch would send a collection of thousands to the
// HBM-arena walk at half the speed.  Collections that mix the two kinds are therefore evaluated as two: the trees that
// fit, and the others; inputs gathered, results scattered back by position.  (Such a call leaves no batch resident.)
int EvaluateByKind(bito_amd_engine* e, int32_t tree_count, int32_t rooted, int32_t node_count, const int32_t* parent_ids,
                   const double* branch_lengths, const double* rates, const double* params, int rescaling,
                   int want_gradient, int want_site, double* out_ll, double* out_grad, double* out_site, bool single) {
  const Worker* w0 = Primary(e);
  const ModelSpec& m = w0->spec;
  const int C = m.category_count;
  const bool candidates = !single && !rescaling && m.state_count == 4 && (C == 1 || C == 2 || C == 4) &&
                          e->n > kPipeExactTaxa && e->n <= kPipeAutoTaxa && tree_count >= 2 && e->kernel_choice == BITO_AMD_KERNEL_AUTO &&
                          parent_ids && branch_lengths && (m.param_count == 0 || params) && !(want_gradient == 0 && C == 1);
  if (!candidates)
    return Evaluate(e, tree_count, rooted, node_count, parent_ids, branch_lengths, rates, params, rescaling, want_gradient,
                    want_site, out_ll, out_grad, out_site, single);
  const size_t M = (size_t)node_count, pc = (size_t)m.param_count, N = 2 * (size_t)e->n - 1;
  const bool has_rates = rooted && rates != nullptr;
  std::vector<int32_t> kind[2];  // [0]: trees that fit the form, [1]: the others
  for (int32_t t = 0; t < tree_count; t++)
    kind[TreeFitsReversibleForm(m, rooted, node_count, branch_lengths + t * M, has_rates ? rates + t * (M - 1) : nullptr,
                                pc ? params + t * pc : nullptr) ? 0 : 1].push_back(t);
  if (kind[0].empty() || kind[1].empty())
    return Evaluate(e, tree_count, rooted, node_count, parent_ids, branch_lengths, rates, params, rescaling, want_gradient,
                    want_site, out_ll, out_grad, out_site, single);
  // (checked as one collection first, so that an error names a tree by its position in the caller's arrays)
  {
    Worker* w = e->workers[0][0].get();
    if (int rc = WorkerValidate(w, tree_count, rooted, node_count, parent_ids, pc ? params : nullptr)) return Propagate(e, w, rc);
  }
  for (const auto& ids : kind) {
    const size_t K = ids.size();
    std::vector<int32_t> pid(K * (M - 1));
    std::vector<double> bl(K * M), par(K * std::max<size_t>(pc, 1)), rt(has_rates ? K * (M - 1) : 0);
    std::vector<double> ll(K), grad(want_gradient ? K * N : 0), site(want_site ? K : 0);
    for (size_t k = 0; k < K; k++) {
      const size_t t = (size_t)ids[k];
      std::copy(parent_ids + t * (M - 1), parent_ids + (t + 1) * (M - 1), pid.begin() + k * (M - 1));
      std::copy(branch_lengths + t * M, branch_lengths + (t + 1) * M, bl.begin() + k * M);
      if (pc) std::copy(params + t * pc, params + (t + 1) * pc, par.begin() + k * pc);
      if (has_rates) std::copy(rates + t * (M - 1), rates + (t + 1) * (M - 1), rt.begin() + k * (M - 1));
    }
    const int rc = Evaluate(e, (int32_t)K, rooted, node_count, pid.data(), bl.data(), has_rates ? rt.data() : nullptr,
                            pc ? par.data() : nullptr, rescaling, want_gradient, want_site, ll.data(),
                            want_gradient ? grad.data() : nullptr, want_site ? site.data() : nullptr, single);
    if (rc) return rc;
    for (size_t k = 0; k < K; k++) {
      const size_t t = (size_t)ids[k];
      out_ll[t] = ll[k];
      if (want_gradient && out_grad) std::copy(grad.begin() + k * N, grad.begin() + (k + 1) * N, out_grad + t * N);
      if (want_site && out_site) out_site[t] = site[k];
    }
  }
  e->resident = false;
  e->shards.clear();
  return BITO_AMD_OK;
}

int LogLikelihoods(bito_amd_engine* e, int32_t tree_count, int32_t rooted, int32_t node_count,
                   const int32_t* parent_ids, const double* branch_lengths, const double* rates, const double* params,
                   int32_t rescaling, double* out, bool single) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  // an empty collection is not an error: FatBeagleParallelize over no trees returns an empty vector
  // (reference src/fat_beagle.hpp:160-181)
  if (tree_count == 0) return BITO_AMD_OK;
  return EvaluateByKind(e, tree_count, rooted, node_count, parent_ids, branch_lengths, rates, params, rescaling != 0, 0, 0,
                        out, nullptr, nullptr, single);
}

// the calls that work on ONE worker's resident batch (time trees, stream hand-off, event timing)
int SingleShard(bito_amd_engine* e, Worker** w) {
  if (!e->resident) return Fail(e, BITO_AMD_ERR_STATE, "no batch is resident: call bito_amd_engine_upload first");
  if (e->shards.size() != 1)
    return Fail(e, BITO_AMD_ERR_STATE, "the resident batch is spread over several devices or chunks: this call needs it on one (a single-device engine and bito_amd_engine_upload)");
  *w = ShardWorker(e, e->shards[0]);
  return BITO_AMD_OK;
}

void SetSingleResident(bito_amd_engine* e, int rooted, int node_count, int tree_count) {
  e->shards.assign(1, Shard{0, 0, 0, tree_count});
  e->resident = true;
  e->rooted = rooted;
  e->node_count = node_count;
  e->tree_count = tree_count;
}

bool OneSlot(const bito_amd_engine* e, int tree_count) { return e->devices.size() == 1 || tree_count < (int)e->devices.size(); }
template <typename T>
const T* At(const T* p, size_t offset) { return p ? p + offset : nullptr; }
template <typename T>
T* At(T* p, size_t offset) { return p ? p + offset : nullptr; }
}  // namespace

extern "C" {

int bito_amd_engine_create(const bito_amd_engine_spec* spec, const char* substitution, const char* site,
                           const char* clock, int32_t taxon_count, int32_t pattern_count, const int32_t* patterns,
                           const double* weights, bito_amd_engine** out, char* err, size_t err_len) {
  auto report = [&](int code, const std::string& msg) {
    if (err && err_len) std::snprintf(err, err_len, "%s", msg.c_str());
    return code;
  };
  if (!out) return report(BITO_AMD_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (taxon_count < 2 || pattern_count < 1 || !patterns || !weights)
    return report(BITO_AMD_ERR_BAD_ARG, "need at least 2 taxa, 1 site pattern and non-NULL arrays");
  // "Thread count needs to be strictly positive." (reference src/engine.cpp:14-16): here, devices
  // (0 = the default, one device, like host_threads and arena_bytes: a zero-initialised spec, or one written for the
  // three-field struct of earlier headers, keeps working)
  const int device_count = (spec && spec->device_count != 0) ? spec->device_count : 1;
  if (device_count < 1) return report(BITO_AMD_ERR_BAD_ARG, "Device count needs to be strictly positive.");
  {
    int present = 0;
    if (hipGetDeviceCount(&present) == hipSuccess && present > 0)
      for (int k = 0; k < device_count; k++) {
        const int id = spec && spec->devices ? spec->devices[k] : (spec ? spec->device_id : 0) + k;
        if (id < 0 || id >= present)
          return report(BITO_AMD_ERR_BAD_ARG, "device " + std::to_string(id) + " (slot " + std::to_string(k) + ") does not exist: this machine has " + std::to_string(present));
      }
  }
  auto e = std::make_unique<bito_amd_engine>();
  for (int k = 0; k < device_count; k++)
    e->devices.push_back(spec && spec->devices ? spec->devices[k] : (spec ? spec->device_id : 0) + k);
  e->workers.resize(device_count);
  e->substitution = substitution ? substitution : "";
  e->site = site ? site : "";
  e->clock = clock ? clock : "";
  e->n = taxon_count;
  e->P = pattern_count;
  e->patterns.assign(patterns, patterns + (size_t)taxon_count * pattern_count);
  e->weights.assign(weights, weights + pattern_count);
  e->arena_bytes = spec ? spec->arena_bytes : 0;
  e->host_threads = spec ? std::max(0, spec->host_threads) : 0;
  if (const char* v = std::getenv("BITO_AMD_HOST_THREADS")) e->host_threads = std::max(1, std::atoi(v));
  if (e->host_threads == 0) e->host_threads = std::min(8, UsableCpus());
  if (e->host_threads > 1) {
    // With helper threads the host stages 6400 config-3 trees in 0.08 ms instead of 0.33: no need for several chunks to
    // hide it.  Two chunks: a first one that gets the device going while the rest is staged, then everything else
    // (scripts/gpu_host_threads_sweep.sh: 4.17-4.20 ms per 6400 trees against 4.32 with one thread and five chunks).
    e->chunk_first = 1024;
    e->chunk_growth = 1e9;
    e->chunk_cap = 1 << 30;
    e->chunk_taper = 0;
    // ... and the first chunk's traversal leaves 32 CUs (four per XCD) to the second chunk's set-up and image kernels,
    // which otherwise run in the gap between the two traversals (0.11 ms): 4.08 -> 3.99 ms per 6400 trees, 2.14 -> 2.06
    // per 3200 on a warm box (16 CUs: no gain; 64: 4.01).  With five chunks every traversal but the last gave up CUs
    // and the call lost more than it gained; now 16 % of the trees do.
    e->reserve_cus = 32;
  }
  if (const char* v = std::getenv("BITO_AMD_CHUNK_FIRST")) e->chunk_first = std::max(1, std::atoi(v));
  if (const char* v = std::getenv("BITO_AMD_CHUNK_CAP")) e->chunk_cap = std::max(1, std::atoi(v));
  if (const char* v = std::getenv("BITO_AMD_CHUNK_GROWTH")) e->chunk_growth = std::max(1.0, std::atof(v));
  if (const char* v = std::getenv("BITO_AMD_CHUNK_TAPER")) e->chunk_taper = std::atoi(v);
  if (const char* v = std::getenv("BITO_AMD_CHUNK_WALK_STREAMS")) e->walk_streams = std::atoi(v);
  if (const char* v = std::getenv("BITO_AMD_CHUNK_RESERVE")) e->reserve_cus = std::max(0, std::atoi(v));
  if (const char* v = std::getenv("BITO_AMD_HOST_MIN_TREES")) e->par_min_trees = std::max(1, std::atoi(v));
  if (const char* v = std::getenv("BITO_AMD_CHUNK_LANES")) e->max_lanes = std::min(kMaxLanes, std::max(1, std::atoi(v)));
  // the first worker of every device now (model strings, device ordinals and the alignment are checked here);
  // further lanes when a call first needs them
  for (int s = 0; s < device_count; s++) {
    Worker* w = nullptr;
    if (int rc = GetWorker(e.get(), s, 0, &w)) return report(rc, e->err);
  }
  *out = e.release();
  return BITO_AMD_OK;
}

void bito_amd_engine_destroy(bito_amd_engine* e) { delete e; }

const char* bito_amd_engine_last_error(const bito_amd_engine* e) { return e ? e->err.c_str() : ""; }

int32_t bito_amd_engine_param_count(const bito_amd_engine* e) { return WorkerParamCount(Primary(e)); }
int32_t bito_amd_engine_category_count(const bito_amd_engine* e) { return WorkerCategoryCount(Primary(e)); }
int32_t bito_amd_engine_state_count(const bito_amd_engine* e) { return WorkerStateCount(Primary(e)); }
int32_t bito_amd_engine_block_count(const bito_amd_engine* e) { return WorkerBlockCount(Primary(e)); }
int32_t bito_amd_engine_device_count(const bito_amd_engine* e) { return e ? (int32_t)e->devices.size() : 0; }

int bito_amd_engine_block(const bito_amd_engine* e, int32_t idx, char* name, size_t name_len, int32_t* start,
                          int32_t* len) {
  return WorkerBlock(Primary(e), idx, name, name_len, start, len);
}

int bito_amd_engine_log_likelihoods(bito_amd_engine* e, int32_t tree_count, int32_t rooted, int32_t node_count,
                                    const int32_t* parent_ids, const double* branch_lengths, const double* rates,
                                    const double* params, int32_t rescaling, double* out) {
  return LogLikelihoods(e, tree_count, rooted, node_count, parent_ids, branch_lengths, rates, params, rescaling, out,
                        /*single=*/false);
}

int bito_amd_engine_gradients(bito_amd_engine* e, int32_t tree_count, int32_t rooted, int32_t node_count,
                              const int32_t* parent_ids, const double* branch_lengths, const double* rates,
                              const double* params, int32_t rescaling, int32_t flags, double fd_delta,
                              double* out_ll, double* out_branch, double* out_site, double* out_subst,
                              double* out_clock) {
  if (!e || !out_ll || !out_branch) return BITO_AMD_ERR_BAD_ARG;
  if (tree_count == 0) return BITO_AMD_OK;  // empty collection, empty result (as bito_amd_engine_log_likelihoods)
  const ModelSpec& m = Primary(e)->spec;
  // the finite-difference batch first: the main batch must be the resident one on return
  if ((flags & BITO_AMD_GRAD_SUBSTITUTION_MODEL) && out_subst && m.rates_len > 0) {
    if (!parent_ids || !branch_lengths || !params) return Fail(e, BITO_AMD_ERR_BAD_ARG, "NULL argument");
    const int rc = SubstitutionGradientsVia(
        m, tree_count, rooted, node_count, parent_ids, branch_lengths, rates, params,
        (flags & BITO_AMD_GRAD_STICKBREAKING) != 0, fd_delta > 0 ? fd_delta : 1e-6, out_subst,
        [&](int32_t big, const int32_t* pid, const double* bl, const double* rt, const double* par, double* ll) {
          return LogLikelihoods(e, big, rooted, node_count, pid, bl, rt, par, rescaling, ll, false);
        });
    if (rc) return rc;
  }
  const int want_site = (flags & BITO_AMD_GRAD_SITE_MODEL) && out_site && m.category_count > 1;
  int rc = EvaluateByKind(e, tree_count, rooted, node_count, parent_ids, branch_lengths, rates, params, rescaling != 0, 1,
                          want_site, out_ll, out_branch, out_site, /*single=*/false);
  if (rc) return rc;
  if (rooted && (flags & BITO_AMD_GRAD_CLOCK_MODEL) && out_clock) {
    // ClockGradient, strict clock (reference src/fat_beagle.cpp:379-399): sum of
    // branch gradient times the tree's own (time) branch length.
    const int N = 2 * e->n - 1;
    for (int t = 0; t < tree_count; t++) {
      double s = 0;
      for (int i = 0; i < N - 1; i++) s += out_branch[(size_t)t * N + i] * branch_lengths[(size_t)t * node_count + i];
      out_clock[t] = s;
    }
  }
  return BITO_AMD_OK;
}

// ---- time trees: Engine::LogLikelihoods / Gradients(RootedTreeCollection) and the height-ratio transforms run over
// every FatBeagle of the engine (reference src/engine.cpp:76-119); here over every device slot: the collection is cut
// into one contiguous block per slot, each block handled by the slot's first worker on a host thread of its own ----

int bito_amd_engine_time_trees_from_branch_lengths(bito_amd_engine* e, int32_t tree_count, const int32_t* parent_ids,
                                                   const double* branch_lengths, const double* tip_dates,
                                                   double* out_node_bounds, double* out_node_heights,
                                                   double* out_height_ratios) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (OneSlot(e, tree_count)) {
    Worker* w = Primary(e);
    w->id_offset = 0;
    return Propagate(e, w, WorkerTimeTreesFromBranchLengths(w, tree_count, parent_ids, branch_lengths, tip_dates,
                                                            out_node_bounds, out_node_heights, out_height_ratios));
  }
  const size_t N = 2 * (size_t)e->n - 1, R = (size_t)e->n - 1;
  return RunPerSlot(e, tree_count, [&](int, int t0, int count, Worker* w) {
    const size_t t = (size_t)t0;
    return WorkerTimeTreesFromBranchLengths(w, count, At(parent_ids, t * (N - 1)), At(branch_lengths, t * N), tip_dates,
                                            At(out_node_bounds, t * N), At(out_node_heights, t * N), At(out_height_ratios, t * R));
  });
}

int bito_amd_engine_time_trees_from_height_ratios(bito_amd_engine* e, int32_t tree_count, const int32_t* parent_ids,
                                                  const double* node_bounds, const double* height_ratios,
                                                  double* out_node_heights, double* out_branch_lengths) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (OneSlot(e, tree_count)) {
    Worker* w = Primary(e);
    w->id_offset = 0;
    return Propagate(e, w, WorkerTimeTreesFromHeightRatios(w, tree_count, parent_ids, node_bounds, height_ratios,
                                                           out_node_heights, out_branch_lengths));
  }
  const size_t N = 2 * (size_t)e->n - 1, R = (size_t)e->n - 1;
  return RunPerSlot(e, tree_count, [&](int, int t0, int count, Worker* w) {
    const size_t t = (size_t)t0;
    return WorkerTimeTreesFromHeightRatios(w, count, At(parent_ids, t * (N - 1)), At(node_bounds, t * N), At(height_ratios, t * R),
                                           At(out_node_heights, t * N), At(out_branch_lengths, t * N));
  });
}

int bito_amd_engine_log_det_jacobian(bito_amd_engine* e, int32_t tree_count, const int32_t* parent_ids,
                                     const double* node_heights, const double* node_bounds, double* out) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (OneSlot(e, tree_count)) {
    Worker* w = Primary(e);
    w->id_offset = 0;
    return Propagate(e, w, WorkerLogDetJacobian(w, tree_count, parent_ids, node_heights, node_bounds, out));
  }
  const size_t N = 2 * (size_t)e->n - 1;
  return RunPerSlot(e, tree_count, [&](int, int t0, int count, Worker* w) {
    const size_t t = (size_t)t0;
    return WorkerLogDetJacobian(w, count, At(parent_ids, t * (N - 1)), At(node_heights, t * N), At(node_bounds, t * N), At(out, t));
  });
}

int bito_amd_engine_gradient_log_det_jacobian(bito_amd_engine* e, int32_t tree_count, const int32_t* parent_ids,
                                              const double* node_heights, const double* node_bounds,
                                              const double* height_ratios, double* out) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (OneSlot(e, tree_count)) {
    Worker* w = Primary(e);
    w->id_offset = 0;
    return Propagate(e, w, WorkerGradientLogDetJacobian(w, tree_count, parent_ids, node_heights, node_bounds,
                                                        height_ratios, out));
  }
  const size_t N = 2 * (size_t)e->n - 1, R = (size_t)e->n - 1;
  return RunPerSlot(e, tree_count, [&](int, int t0, int count, Worker* w) {
    const size_t t = (size_t)t0;
    return WorkerGradientLogDetJacobian(w, count, At(parent_ids, t * (N - 1)), At(node_heights, t * N), At(node_bounds, t * N),
                                        At(height_ratios, t * R), At(out, t * R));
  });
}

int bito_amd_engine_ratio_gradient_of_height_gradient(bito_amd_engine* e, int32_t tree_count,
                                                      const int32_t* parent_ids, const double* node_heights,
                                                      const double* node_bounds, const double* height_ratios,
                                                      const double* height_gradient, double* out) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (OneSlot(e, tree_count)) {
    Worker* w = Primary(e);
    w->id_offset = 0;
    return Propagate(e, w, WorkerRatioGradientOfHeightGradient(w, tree_count, parent_ids, node_heights, node_bounds,
                                                               height_ratios, height_gradient, out));
  }
  const size_t N = 2 * (size_t)e->n - 1, R = (size_t)e->n - 1;
  return RunPerSlot(e, tree_count, [&](int, int t0, int count, Worker* w) {
    const size_t t = (size_t)t0;
    return WorkerRatioGradientOfHeightGradient(w, count, At(parent_ids, t * (N - 1)), At(node_heights, t * N), At(node_bounds, t * N),
                                               At(height_ratios, t * R), At(height_gradient, t * R), At(out, t * R));
  });
}

int bito_amd_engine_time_tree_log_likelihoods(bito_amd_engine* e, int32_t tree_count, const int32_t* parent_ids,
                                              const double* branch_lengths, const double* rates,
                                              const double* node_heights, const double* node_bounds,
                                              const double* params, int32_t rescaling,
                                              int32_t include_log_det_jacobian, double* out) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  e->resident = false;
  if (OneSlot(e, tree_count)) {
    Worker* w = Primary(e);
    w->one_shot = 0;
    w->reserve_cus = 0;
    w->id_offset = 0;
    const int rc = WorkerTimeTreeLogLikelihoods(w, tree_count, parent_ids, branch_lengths, rates, node_heights,
                                                node_bounds, params, rescaling, include_log_det_jacobian, out);
    if (rc) return Propagate(e, w, rc);
    SetSingleResident(e, 1, 2 * e->n - 1, tree_count);
    return BITO_AMD_OK;
  }
  const size_t N = 2 * (size_t)e->n - 1, pc = (size_t)Primary(e)->spec.param_count;
  const int rc = RunPerSlot(e, tree_count, [&](int, int t0, int count, Worker* w) {
    const size_t t = (size_t)t0;
    w->one_shot = 0;
    w->reserve_cus = 0;
    return WorkerTimeTreeLogLikelihoods(w, count, At(parent_ids, t * (N - 1)), At(branch_lengths, t * N), At(rates, t * (N - 1)),
                                        At(node_heights, t * N), At(node_bounds, t * N), At(params, t * pc), rescaling,
                                        include_log_det_jacobian, At(out, t));
  });
  if (rc) return rc;
  SetSlotResident(e, 1, 2 * e->n - 1, tree_count);
  return BITO_AMD_OK;
}

int bito_amd_engine_time_tree_gradients(bito_amd_engine* e, int32_t tree_count, const int32_t* parent_ids,
                                        const double* branch_lengths, const double* rates, int32_t rate_count,
                                        const double* node_heights, const double* node_bounds,
                                        const double* height_ratios, const double* params, int32_t rescaling,
                                        int32_t flags, double fd_delta, double* out_ll, double* out_branch,
                                        double* out_site, double* out_subst, double* out_clock, double* out_ratios) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  e->resident = false;
  if (OneSlot(e, tree_count)) {
    Worker* w = Primary(e);
    w->one_shot = 0;
    w->reserve_cus = 0;
    w->id_offset = 0;
    const int rc = WorkerTimeTreeGradients(w, tree_count, parent_ids, branch_lengths, rates, rate_count, node_heights,
                                           node_bounds, height_ratios, params, rescaling, flags, fd_delta, out_ll,
                                           out_branch, out_site, out_subst, out_clock, out_ratios);
    if (rc) return Propagate(e, w, rc);
    SetSingleResident(e, 1, 2 * e->n - 1, tree_count);
    return BITO_AMD_OK;
  }
  const ModelSpec& m = Primary(e)->spec;
  const size_t N = 2 * (size_t)e->n - 1, R = (size_t)e->n - 1, pc = (size_t)m.param_count;
  const size_t subst_stride = (size_t)m.rates_len + 4, clock_stride = rate_count == 1 ? 1 : N - 1;
  const int rc = RunPerSlot(e, tree_count, [&](int, int t0, int count, Worker* w) {
    const size_t t = (size_t)t0;
    w->one_shot = 0;
    w->reserve_cus = 0;
    return WorkerTimeTreeGradients(w, count, At(parent_ids, t * (N - 1)), At(branch_lengths, t * N), At(rates, t * (N - 1)),
                                   rate_count, At(node_heights, t * N), At(node_bounds, t * N), At(height_ratios, t * R),
                                   At(params, t * pc), rescaling, flags, fd_delta, At(out_ll, t), At(out_branch, t * N),
                                   At(out_site, t), At(out_subst, t * subst_stride), At(out_clock, t * clock_stride),
                                   At(out_ratios, t * R));
  });
  if (rc) return rc;
  SetSlotResident(e, 1, 2 * e->n - 1, tree_count);
  return BITO_AMD_OK;
}

// ---- HBM-resident batch interface: one block per device, passes pipelined on each device's first worker ----

int bito_amd_engine_upload(bito_amd_engine* e, int32_t tree_count, int32_t rooted, int32_t node_count,
                           const int32_t* parent_ids, const double* branch_lengths, const double* rates,
                           const double* params) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  e->resident = false;
  if (tree_count < 1 || !parent_ids || !branch_lengths)
    return Fail(e, BITO_AMD_ERR_BAD_ARG, "need at least one tree and non-NULL parent_ids / branch_lengths");
  const size_t pc = (size_t)Primary(e)->spec.param_count, M = (size_t)node_count;
  if (pc > 0 && !params) return Fail(e, BITO_AMD_ERR_BAD_ARG, "params is NULL but the model has parameters");
  const int D = (int)e->devices.size();
  e->shards.clear();
  for (int s = 0; s < D; s++) {
    const int t0 = (int)((long long)tree_count * s / D), t1 = (int)((long long)tree_count * (s + 1) / D);
    if (t1 > t0) e->shards.push_back({s, 0, t0, t1 - t0});
  }
  const bool has_rates = rooted && rates != nullptr;
  for (const Shard& s : e->shards) {
    Worker* w = ShardWorker(e, s);
    w->one_shot = 0;
  w->reserve_cus = 0;
    w->id_offset = s.t0;
    const int rc = WorkerStage(w, s.count, rooted, node_count, parent_ids + (size_t)s.t0 * (M - 1),
                               branch_lengths + (size_t)s.t0 * M, has_rates ? rates + (size_t)s.t0 * (M - 1) : nullptr,
                               pc > 0 ? params + (size_t)s.t0 * pc : nullptr, /*wait=*/0);
    if (rc) return Propagate(e, w, rc);
  }
  for (const Shard& s : e->shards)
    if (int rc = WorkerSync(ShardWorker(e, s))) return Propagate(e, ShardWorker(e, s), rc);
  e->resident = true;
  e->rooted = rooted;
  e->node_count = node_count;
  e->tree_count = tree_count;
  return BITO_AMD_OK;
}

int bito_amd_engine_update(bito_amd_engine* e, const double* branch_lengths, const double* params) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (!e->resident) return Fail(e, BITO_AMD_ERR_STATE, "no batch is resident: call bito_amd_engine_upload first");
  const size_t pc = (size_t)Primary(e)->spec.param_count, M = (size_t)e->node_count;
  for (const Shard& s : e->shards) {
    Worker* w = ShardWorker(e, s);
    w->id_offset = s.t0;
    const int rc = WorkerUpdate(w, branch_lengths ? branch_lengths + (size_t)s.t0 * M : nullptr,
                                (params && pc > 0) ? params + (size_t)s.t0 * pc : nullptr);
    if (rc) return Propagate(e, w, rc);
  }
  return BITO_AMD_OK;
}

int bito_amd_engine_run(bito_amd_engine* e, int32_t want_gradient, int32_t rescaling) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (!e->resident) return Fail(e, BITO_AMD_ERR_STATE, "no batch is resident: call bito_amd_engine_upload first");
  for (const Shard& s : e->shards) {
    Worker* w = ShardWorker(e, s);
    w->one_shot = 0;  // passes over a resident batch are pipelined: the set-up of pass k+1 beside the traversal of pass k
    w->reserve_cus = 0;
    if (int rc = WorkerRun(w, want_gradient, rescaling)) return Propagate(e, w, rc);
  }
  return BITO_AMD_OK;
}

int bito_amd_engine_sync(bito_amd_engine* e) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  for (auto& lanes : e->workers)
    for (auto& w : lanes)
      if (w)
        if (int rc = WorkerSync(w.get())) return Propagate(e, w.get(), rc);
  return BITO_AMD_OK;
}

int bito_amd_engine_download_async(bito_amd_engine* e, double* out_ll, double* out_grad) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  if (!e->resident) return Fail(e, BITO_AMD_ERR_STATE, "no batch is resident");
  const size_t N = 2 * (size_t)e->n - 1;
  for (const Shard& s : e->shards) {
    Worker* w = ShardWorker(e, s);
    const int rc = WorkerDownloadAsync(w, out_ll ? out_ll + s.t0 : nullptr, out_grad ? out_grad + (size_t)s.t0 * N : nullptr);
    if (rc) return Propagate(e, w, rc);
  }
  return BITO_AMD_OK;
}

int bito_amd_engine_download(bito_amd_engine* e, double* out_ll, double* out_grad) {
  if (int rc = bito_amd_engine_download_async(e, out_ll, out_grad)) return rc;
  for (const Shard& s : e->shards)
    if (int rc = WorkerSync(ShardWorker(e, s))) return Propagate(e, ShardWorker(e, s), rc);
  return BITO_AMD_OK;
}

int bito_amd_engine_results_async(bito_amd_engine* e, void* consumer_stream, const double** out_ll,
                                  const double** out_grad) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  Worker* w = nullptr;
  if (int rc = SingleShard(e, &w)) return rc;
  return Propagate(e, w, WorkerResultsAsync(w, consumer_stream, out_ll, out_grad));
}

void* bito_amd_engine_stream(bito_amd_engine* e) { return e ? WorkerStream(Primary(e)) : nullptr; }

// ---- diagnostics / benchmarking ----

int bito_amd_engine_set_kernel(bito_amd_engine* e, int32_t kernel) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  e->kernel_choice = kernel;
  for (auto& lanes : e->workers)
    for (auto& w : lanes)
      if (w) WorkerSetKernel(w.get(), kernel);
  return BITO_AMD_OK;
}

int bito_amd_engine_kernel_timing(bito_amd_engine* e, int32_t enable) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  e->timing = enable != 0;
  for (auto& lanes : e->workers)
    for (auto& w : lanes)
      if (w) WorkerKernelTiming(w.get(), enable);
  return BITO_AMD_OK;
}

int bito_amd_engine_kernel_elapsed(bito_amd_engine* e, double* kernel_ms, int32_t* kernel_launches) {
  if (!e) return BITO_AMD_ERR_BAD_ARG;
  // Time the traversal kernels were running, per device slot the UNION of the launches' spans: the chunks of a blocking
  // call run on two streams, and the second chunk's workgroups move in while the first chunk's are still leaving (its
  // set-up ran beside the first traversal) -- the sum of the spans would count that stretch twice.
  double total = 0, span_sum = 0;
  int launches = 0;
  for (auto& lanes : e->workers) {
    hipEvent_t base = nullptr;
    std::vector<std::pair<double, double>> spans;
    for (auto& w : lanes) {
      if (!w) continue;
      if (hipSetDevice(w->device) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess ||
          (w->last_walk && w->last_walk != w->stream && hipStreamSynchronize(w->last_walk) != hipSuccess))
        return Fail(e, BITO_AMD_ERR_DEVICE, "kernel timing: the worker's streams could not be synchronised");
      for (size_t i = 0; i + 1 < w->ev_used; i += 2) {
        if (!base) base = w->ev_pool[i];
        float t0 = 0, t1 = 0;
        if (hipEventElapsedTime(&t0, base, w->ev_pool[i]) != hipSuccess ||
            hipEventElapsedTime(&t1, base, w->ev_pool[i + 1]) != hipSuccess)
          return Fail(e, BITO_AMD_ERR_DEVICE, "kernel timing: hipEventElapsedTime failed");
        spans.emplace_back((double)t0, (double)t1);
      }
    }
    std::sort(spans.begin(), spans.end());
    for (const auto& s : spans) span_sum += s.second - s.first;
    double covered_to = -1e300;
    for (const auto& s : spans) {
      const double from = std::max(s.first, covered_to);
      if (s.second > from) total += s.second - from;
      covered_to = std::max(covered_to, s.second);
    }
    launches += (int)spans.size();
    for (auto& w : lanes)
      if (w) w->ev_used = 0;
  }
  e->span_sum_ms = span_sum;
  if (kernel_ms) *kernel_ms = total;
  if (kernel_launches) *kernel_launches = launches;
  return BITO_AMD_OK;
}

double bito_amd_engine_kernel_span_sum(const bito_amd_engine* e) { return e ? e->span_sum_ms : 0.0; }

int bito_amd_engine_read_general_model(bito_amd_engine* e, int32_t tree, double* out, size_t capacity) {
  if (!e || !out) return BITO_AMD_ERR_BAD_ARG;
  if (!e->resident) return Fail(e, BITO_AMD_ERR_STATE, "no general-state model is resident for that tree");
  for (const Shard& s : e->shards)
    if (tree >= s.t0 && tree < s.t0 + s.count) {
      Worker* w = ShardWorker(e, s);
      return Propagate(e, w, WorkerReadGeneralModel(w, tree - s.t0, out, capacity));
    }
  return Fail(e, BITO_AMD_ERR_STATE, "no general-state model is resident for that tree");
}

const char* bito_amd_engine_kernel_form(const bito_amd_engine* e) {
  if (!e) return "";
  const Worker* w = e->resident && !e->shards.empty() ? ShardWorker(e, e->shards[0]) : Primary(e);
  return w ? w->kernel_form.c_str() : "";
}

const char* bito_amd_engine_kernel_name(const bito_amd_engine* e) {
  if (!e) return "";
  return WorkerKernelName(e->resident && !e->shards.empty() ? ShardWorker(e, e->shards[0]) : Primary(e));
}

int bito_amd_engine_time_runs(bito_amd_engine* e, int32_t want_gradient, int32_t rescaling, int32_t steps,
                              double* total_ms, double* kernel_ms, int32_t* kernel_launches) {
  if (!e || steps < 1) return BITO_AMD_ERR_BAD_ARG;
  Worker* w = nullptr;
  if (int rc = SingleShard(e, &w)) return rc;
  w->one_shot = 0;
  w->reserve_cus = 0;
  return Propagate(e, w, WorkerTimeRuns(w, want_gradient, rescaling, steps, total_ms, kernel_ms, kernel_launches));
}

}  // extern "C"
