// gp_schedule.hpp -- the order in which the GP executor runs a GPOperation stream (host arithmetic only, no HIP).
//
// The reference executes a GPOperationVector one operation after the other (GPEngine::ProcessOperations,
// src/gp_engine.cpp:213-339).  Every operation except OptimizeBranchLength and UpdateSBNProbabilities is independent
// across site patterns and cheap; an OptimizeBranchLength is a whole one-dimensional optimisation (some fifteen to
// thirty function evaluations, each a reduction over all patterns: 39 us in one workgroup) -- so a branch-length sweep
// (GPDAG::BranchLengthOptimization, src/gp_dag.cpp:78-121) costs what its optimisations cost ONE AFTER THE OTHER.
// But the sweep's operations form a dependency graph, not a chain: in a DAG with several subsplits per clade, the
// edges below different children of a clade are optimised from PLVs that do not depend on one another.
//
// ScheduleStream builds that graph from the operations' read and write sets -- PLVs (with their rescaling counts),
// per-GPCSP log-likelihood rows, the marginal row, and per-GPCSP scalars (branch length + difference) -- with
// read-after-write, write-after-read and write-after-write edges, so ANY order that respects it computes every
// value from the same inputs by the same arithmetic: bit for bit the sequential result.  Operations are then placed by
// their OPTIMISER DEPTH d = the largest number of OptimizeBranchLength operations on a path of predecessors:
//
//     pattern group 0 | optimiser launch 0 | pattern group 1 | optimiser launch 1 | ...
//
// group d = the per-pattern operations of depth d, sorted into dependency levels (operations of one level touch
// disjoint results); launch d = the optimisations of depth d, mutually independent (no path joins two operations of
// equal depth), run as concurrent workgroups.  The number of optimiser launches is the longest chain of optimisations
// in the graph -- DS1 ten-tree DAG: 56 instead of 118; 20 seeded topologies: 156 instead of 950.
// UpdateSBNProbabilities reads every log-likelihood row and writes q: it stays a barrier between scheduled regions.
#pragma once

#include <algorithm>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "../../include/bito_amd_gp.h"

namespace bito_amd_gp_schedule {

enum LaunchKind : int32_t { kPatternOps = 0, kOptimisers = 1, kSbnUpdate = 2 };

struct Launch {
  int32_t kind;
  int64_t first, count;         // [first, first + count) of the scheduled image
  int64_t level_first;          // kPatternOps: its levels are level_offsets[level_first .. level_first + level_count]
  int32_t level_count;
};

struct Schedule {
  std::vector<bito_amd_gp_op> image;   // the stream's operations in execution order
  std::vector<Launch> launches;
  std::vector<int64_t> level_offsets;  // absolute indices into image, per pattern launch level_count + 1 entries
  std::vector<int32_t> launch_of, level_of;  // per operation of the image (diagnostics, tests)
  int64_t max_concurrent_optimisers = 0;
};

namespace detail {

// resources: PLV id | per-GPCSP row (bit 62) | per-GPCSP scalars: branch length + difference (bit 61) | marginal (bit 63)
inline uint64_t Plv(uint64_t id) { return id; }
inline uint64_t Row(uint64_t id) { return id | (1ull << 62); }
inline uint64_t Edge(uint64_t id) { return id | (1ull << 61); }
constexpr uint64_t kMarginal = 1ull << 63;

struct Sets {
  std::vector<uint64_t> reads, writes;
};

inline void ReadWriteSets(const bito_amd_gp_op& op, const uint64_t* side, Sets* s) {
  s->reads.clear();
  s->writes.clear();
  auto R = [&](uint64_t r) { s->reads.push_back(r); };
  auto W = [&](uint64_t w) { s->writes.push_back(w); };
  switch (op.opcode) {
    case BITO_AMD_GP_ZERO_PLV: W(Plv(op.a)); break;
    case BITO_AMD_GP_SET_TO_STATIONARY_DISTRIBUTION: W(Plv(op.a)); break;  // (q is written by the SBN update only: a barrier)
    case BITO_AMD_GP_INCREMENT_WITH_WEIGHTED_EVOLVED_PLV: W(Plv(op.a)); R(Plv(op.a)); R(Plv(op.c)); R(Edge(op.b)); break;
    case BITO_AMD_GP_MULTIPLY: W(Plv(op.a)); R(Plv(op.b)); R(Plv(op.c)); break;
    case BITO_AMD_GP_LIKELIHOOD: W(Row(op.a)); R(Plv(op.b)); R(Plv(op.c)); R(Edge(op.a)); break;
    case BITO_AMD_GP_OPTIMIZE_BRANCH_LENGTH: W(Edge(op.c)); R(Edge(op.c)); R(Plv(op.a)); R(Plv(op.b)); break;
    case BITO_AMD_GP_RESET_MARGINAL_LIKELIHOOD: W(kMarginal); break;
    case BITO_AMD_GP_INCREMENT_MARGINAL_LIKELIHOOD:
      W(kMarginal); W(Row(op.b)); R(kMarginal); R(Plv(op.a)); R(Plv(op.c)); break;
    case BITO_AMD_GP_PREP_FOR_MARGINALIZATION:
      W(Plv(op.a)); R(Plv(op.a));
      for (uint32_t j = 0; j < op.count; j++) R(Plv(side[op.b + j]));
      break;
    default: break;
  }
}

// One region [first, first + count) without an SBN update, appended to the schedule.
inline void ScheduleRegion(const bito_amd_gp_op* ops, int64_t first, int64_t count, const uint64_t* side, Schedule* out) {
  if (count <= 0) return;
  struct Use {
    int writer_depth = -1;                    // depth a reader of this resource has at least (writer's depth + is-optimiser)
    int readers_depth = -1;                   // the same over the readers since the last write
    int writer_group = -1, writer_level = -1;  // pass 2: the last writer's group and level inside it
    int readers_group = -1, readers_level = -1;  // pass 2: the highest group among the readers since, deepest level in it
  };
  std::unordered_map<uint64_t, Use> use;
  use.reserve((size_t)count * 2);
  std::vector<int> depth(count), level(count, 0);
  Sets sets;
  // pass 1: optimiser depth
  int deepest = 0;
  for (int64_t k = 0; k < count; k++) {
    const bito_amd_gp_op& op = ops[first + k];
    ReadWriteSets(op, side, &sets);
    int d = 0;
    for (uint64_t r : sets.reads) {
      auto it = use.find(r);
      if (it != use.end()) d = std::max(d, it->second.writer_depth);
    }
    for (uint64_t w : sets.writes) {
      auto it = use.find(w);
      if (it != use.end()) d = std::max(d, std::max(it->second.writer_depth, it->second.readers_depth));
    }
    depth[k] = d;
    deepest = std::max(deepest, d);
    const int after = d + (op.opcode == BITO_AMD_GP_OPTIMIZE_BRANCH_LENGTH ? 1 : 0);
    for (uint64_t r : sets.reads) {
      Use& u = use[r];
      u.readers_depth = std::max(u.readers_depth, after);
    }
    for (uint64_t w : sets.writes) {
      Use& u = use[w];
      u.writer_depth = after;
      u.readers_depth = -1;
    }
  }
  // pass 2: dependency levels of the per-pattern operations inside their group (predecessors of the same group only:
  // everything of a lower group, and every optimiser a pattern operation depends on, has run before the group starts)
  for (auto& kv : use) kv.second = Use{};
  std::vector<int> levels_in_group(deepest + 1, 0);
  for (int64_t k = 0; k < count; k++) {
    const bito_amd_gp_op& op = ops[first + k];
    ReadWriteSets(op, side, &sets);
    const bool is_opt = op.opcode == BITO_AMD_GP_OPTIMIZE_BRANCH_LENGTH;
    const int g = depth[k];
    int lv = 0;
    if (!is_opt) {
      for (uint64_t r : sets.reads) {
        const Use& u = use[r];
        if (u.writer_group == g) lv = std::max(lv, u.writer_level + 1);
      }
      for (uint64_t w : sets.writes) {
        const Use& u = use[w];
        if (u.writer_group == g) lv = std::max(lv, u.writer_level + 1);
        if (u.readers_group == g) lv = std::max(lv, u.readers_level + 1);
      }
      level[k] = lv;
      levels_in_group[g] = std::max(levels_in_group[g], lv + 1);
    }
    // an optimiser never shares a group with an operation that depends on it (that one is a group deeper): it is
    // recorded with a group of its own that no pattern operation has
    const int my_group = is_opt ? -2 : g, my_level = is_opt ? -1 : lv;
    for (uint64_t r : sets.reads) {
      Use& u = use[r];
      if (my_group > u.readers_group) { u.readers_group = my_group; u.readers_level = my_level; }
      else if (my_group == u.readers_group) u.readers_level = std::max(u.readers_level, my_level);
    }
    for (uint64_t w : sets.writes) {
      Use& u = use[w];
      u.writer_group = my_group;
      u.writer_level = my_level;
      u.readers_group = -1;
      u.readers_level = -1;
    }
  }
  // placement: group d by level (stable), then the optimisers of depth d (stream order)
  std::vector<std::vector<int64_t>> pattern_ops(deepest + 1), optimisers(deepest + 1);
  for (int64_t k = 0; k < count; k++)
    (ops[first + k].opcode == BITO_AMD_GP_OPTIMIZE_BRANCH_LENGTH ? optimisers : pattern_ops)[depth[k]].push_back(k);
  for (int d = 0; d <= deepest; d++) {
    std::vector<int64_t>& group = pattern_ops[d];
    if (!group.empty()) {
      std::stable_sort(group.begin(), group.end(), [&](int64_t a, int64_t b) { return level[a] < level[b]; });
      Launch L{kPatternOps, (int64_t)out->image.size(), (int64_t)group.size(), (int64_t)out->level_offsets.size(), 0};
      int current = -1;
      for (int64_t k : group) {
        if (level[k] != current) {
          out->level_offsets.push_back((int64_t)out->image.size());
          current = level[k];
          L.level_count++;
        }
        out->image.push_back(ops[first + k]);
        out->launch_of.push_back((int32_t)out->launches.size());
        out->level_of.push_back(L.level_count - 1);
      }
      out->level_offsets.push_back((int64_t)out->image.size());
      out->launches.push_back(L);
    }
    if (!optimisers[d].empty()) {
      Launch L{kOptimisers, (int64_t)out->image.size(), (int64_t)optimisers[d].size(), 0, 0};
      for (int64_t k : optimisers[d]) {
        out->image.push_back(ops[first + k]);
        out->launch_of.push_back((int32_t)out->launches.size());
        out->level_of.push_back(0);
      }
      out->max_concurrent_optimisers = std::max<int64_t>(out->max_concurrent_optimisers, L.count);
      out->launches.push_back(L);
    }
  }
}

}  // namespace detail

// The whole stream.  `reorder` false: the stream as given, cut into the same kinds of launches (runs of per-pattern
// operations in stream order as one level each operation -- i.e. sequential --, every optimisation a launch of its own).
inline void ScheduleStream(const bito_amd_gp_op* ops, int64_t op_count, const uint64_t* side, bool reorder, Schedule* out) {
  out->image.clear();
  out->launches.clear();
  out->level_offsets.clear();
  out->launch_of.clear();
  out->level_of.clear();
  out->max_concurrent_optimisers = 0;
  out->image.reserve((size_t)op_count);
  int64_t start = 0;
  auto flush_sequential = [&](int64_t from, int64_t to) {
    // per-pattern runs between optimisations, in stream order; still sorted into levels (the sort keeps chains in order)
    int64_t seg = from;
    for (int64_t o = from; o <= to; o++) {
      const bool is_opt = o < to && ops[o].opcode == BITO_AMD_GP_OPTIMIZE_BRANCH_LENGTH;
      if (o < to && !is_opt) continue;
      if (o > seg) {
        // a run without an optimiser is one region of depth 0: ScheduleRegion places it as ONE pattern launch
        detail::ScheduleRegion(ops, seg, o - seg, side, out);
      }
      if (is_opt) {
        Launch L{kOptimisers, (int64_t)out->image.size(), 1, 0, 0};
        out->image.push_back(ops[o]);
        out->launch_of.push_back((int32_t)out->launches.size());
        out->level_of.push_back(0);
        out->max_concurrent_optimisers = std::max<int64_t>(out->max_concurrent_optimisers, 1);
        out->launches.push_back(L);
      }
      seg = o + 1;
    }
  };
  for (int64_t o = 0; o <= op_count; o++) {
    if (o < op_count && ops[o].opcode != BITO_AMD_GP_UPDATE_SBN_PROBABILITIES) continue;
    if (reorder) detail::ScheduleRegion(ops, start, o - start, side, out);
    else flush_sequential(start, o);
    if (o < op_count) {
      Launch L{kSbnUpdate, (int64_t)out->image.size(), 1, 0, 0};
      out->image.push_back(ops[o]);
      out->launch_of.push_back((int32_t)out->launches.size());
      out->level_of.push_back(0);
      out->launches.push_back(L);
    }
    start = o + 1;
  }
}

}  // namespace bito_amd_gp_schedule
