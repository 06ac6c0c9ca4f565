// ref_shim.cpp -- a C ABI around the REFERENCE's own code, for the parts of the path that compile from its own source
// files without BEAGLE or Eigen (test infrastructure: only tests/ loads the library this builds).
//
// oracle/Makefile compiles this file together with the reference's sources WHERE THEY LIE under /root/reference/src --
// alignment.cpp, site_pattern.cpp, node.cpp, tree.cpp, unrooted_tree.cpp, bitset.cpp and the header-only optimisers
// (optimization.hpp) -- into oracle/_ref/libbito_ref.so (git-ignored; nothing of the reference is copied into this
// repository).  What it pins, by running the reference itself:
//   SURVEY 8a A1  SitePattern::Compress / GetPatterns / GetWeights          (src/site_pattern.cpp:16-131)
//   SURVEY 8a A6  Node::Polish (which ids a topology's nodes get), Node::OfParentIdVector / ParentIdVector,
//                 UnrootedTree::Detrifurcate                                (src/node.cpp:383-551, src/unrooted_tree.cpp:27-37)
//   SURVEY 8b (3) the alternative index of every GPOperation in the reference's std::variant (src/gp_operation.hpp:162-167):
//                 the opcodes of include/bito_amd_gp.h
//   SURVEY 8f f1  Optimization::BrentMinimize(WithGradients), GradientAscent, LogSpaceGradientAscent,
//                 NewtonRaphsonOptimization                                  (src/optimization.hpp:71-405)
// Not buildable here, and therefore not in it: the Newick / Nexus parser (driver.cpp calls into taxon_name_munging.cpp,
// which includes Eigen through numerical_utils.hpp) and the likelihood arithmetic itself (BEAGLE, Eigen) -- those stay
// with the restatements, pinned to the reference's golden values.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "alignment.hpp"
#include "gp_operation.hpp"
#include "optimization.hpp"
#include "site_pattern.hpp"
#include "unrooted_tree.hpp"

namespace {

template <typename F>
int Guarded(F&& body) {
  try {
    body();
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}

}  // namespace

extern "C" {

// A topology given as nested children -- node k of `count` nodes has children child_ids[child_start[k] ..
// child_start[k + 1]) in that order (none: a leaf carrying taxon id leaf_taxon[k]), root = node `root` -- built with
// Node::Leaf / Node::Join, then Node::Polish: out[count - 1] = Node::ParentIdVector() of the polished topology.
int ref_polished_parent_ids(int count, int root, const int32_t* child_start, const int32_t* child_ids, const int32_t* leaf_taxon,
                            int64_t* out) {
  return Guarded([&] {
    std::vector<Node::NodePtr> built((size_t)count);
    // children before parents: post-order over the given structure
    std::vector<std::pair<int, bool>> stack{{root, false}};
    while (!stack.empty()) {
      auto [k, done] = stack.back();
      stack.pop_back();
      const int first = child_start[k], last = child_start[k + 1];
      if (first == last) {
        built[(size_t)k] = Node::Leaf((uint32_t)leaf_taxon[k]);
      } else if (done) {
        Node::NodePtrVec children;
        for (int c = first; c < last; c++) children.push_back(built[(size_t)child_ids[c]]);
        built[(size_t)k] = Node::Join(children);
      } else {
        stack.push_back({k, true});
        for (int c = last - 1; c >= first; c--) stack.push_back({child_ids[c], false});
      }
    }
    built[(size_t)root]->Polish();
    const auto ids = built[(size_t)root]->ParentIdVector();
    for (size_t k = 0; k < ids.size(); k++) out[k] = (int64_t)ids[k];
  });
}
// Node::OfParentIdVector(ids)->ParentIdVector(): the round trip the reference's own test makes (src/node.hpp:343)
int ref_parent_id_round_trip(const int64_t* ids, int count, int64_t* out) {
  return Guarded([&] {
    const auto back = Node::OfParentIdVector(std::vector<size_t>(ids, ids + count))->ParentIdVector();
    for (size_t k = 0; k < back.size(); k++) out[k] = (int64_t)back[k];
  });
}
// UnrootedTree(Node::OfParentIdVector(ids), branch_lengths).Detrifurcate(): parent ids (count + 1 entries: one node more
// than the unrooted tree) and branch lengths (count + 2)
int ref_detrifurcate(const int64_t* ids, int count, const double* branch_lengths, int64_t* out_ids, double* out_lengths) {
  return Guarded([&] {
    const UnrootedTree unrooted(Node::OfParentIdVector(std::vector<size_t>(ids, ids + count)),
                                Tree::BranchLengthVector(branch_lengths, branch_lengths + count + 1));
    const Tree rooted = unrooted.Detrifurcate();
    const auto back = rooted.ParentIdVector();
    for (size_t k = 0; k < back.size(); k++) out_ids[k] = (int64_t)back[k];
    std::memcpy(out_lengths, rooted.BranchLengths().data(), rooted.BranchLengths().size() * sizeof(double));
  });
}

// SitePattern(Alignment::ReadFasta(fasta), {PackInts(id, 1): names[id]})
void* ref_site_pattern(const char* fasta, const char* const* names, int taxon_count) {
  try {
    TagStringMap tags;
    for (int i = 0; i < taxon_count; i++) tags[PackInts((uint32_t)i, 1)] = names[i];
    return new SitePattern(Alignment::ReadFasta(fasta), tags);
  } catch (const std::exception&) {
    return nullptr;
  }
}
void ref_free_site_pattern(void* h) { delete static_cast<SitePattern*>(h); }
int ref_pattern_count(void* h) { return (int)static_cast<SitePattern*>(h)->PatternCount(); }
int ref_sequence_count(void* h) { return (int)static_cast<SitePattern*>(h)->SequenceCount(); }
void ref_patterns(void* h, int32_t* out) {  // [sequence = taxon id][pattern]
  const auto& p = static_cast<SitePattern*>(h)->GetPatterns();
  size_t at = 0;
  for (const auto& row : p)
    for (auto symbol : row) out[at++] = (int32_t)symbol;
}
void ref_weights(void* h, double* out) {
  const auto& w = static_cast<SitePattern*>(h)->GetWeights();
  std::memcpy(out, w.data(), w.size() * sizeof(double));
}

// The reference's GPOperation for each opcode of include/bito_amd_gp.h, built with recognisable field values
// (a = 11, b = 22, c = 33 in the order of bito_amd_gp_op's comment), and handed back as (variant index, a, b, c, count)
// through the visitor a binding would use: the opcode numbering and the field order of the seam are the reference's.
int ref_gp_operation(int opcode, uint64_t out[5]) {
  using namespace GPOperations;
  GPOperation op = ZeroPLV{0};
  switch (opcode) {
    case 0: op = ZeroPLV{11}; break;
    case 1: op = SetToStationaryDistribution{11, 22}; break;
    case 2: op = IncrementWithWeightedEvolvedPLV{11, 22, 33}; break;
    case 3: op = Multiply{11, 22, 33}; break;
    case 4: op = Likelihood{11, 22, 33}; break;
    case 5: op = OptimizeBranchLength{11, 22, 33}; break;
    case 6: op = UpdateSBNProbabilities{11, 22}; break;
    case 7: op = ResetMarginalLikelihood{}; break;
    case 8: op = IncrementMarginalLikelihood{11, 22, 33}; break;
    case 9: op = PrepForMarginalization{11, {5, 6, 7}}; break;
    default: return -1;
  }
  struct Fields {
    uint64_t* o;
    void operator()(const ZeroPLV& x) { o[1] = x.dest_; }
    void operator()(const SetToStationaryDistribution& x) { o[1] = x.dest_; o[2] = x.root_gpcsp_idx_; }
    void operator()(const IncrementWithWeightedEvolvedPLV& x) { o[1] = x.dest_; o[2] = x.gpcsp_; o[3] = x.src_; }
    void operator()(const Multiply& x) { o[1] = x.dest_; o[2] = x.src1_; o[3] = x.src2_; }
    void operator()(const Likelihood& x) { o[1] = x.dest_; o[2] = x.child_; o[3] = x.parent_; }
    void operator()(const OptimizeBranchLength& x) { o[1] = x.leafward_; o[2] = x.rootward_; o[3] = x.gpcsp_; }
    void operator()(const UpdateSBNProbabilities& x) { o[1] = x.start_; o[2] = x.stop_; }
    void operator()(const ResetMarginalLikelihood&) {}
    void operator()(const IncrementMarginalLikelihood& x) { o[1] = x.stationary_times_prior_; o[2] = x.rootsplit_; o[3] = x.p_; }
    void operator()(const PrepForMarginalization& x) { o[1] = x.dest_; o[4] = x.src_vector_.size(); }
  };
  out[0] = op.index();
  out[1] = out[2] = out[3] = out[4] = 0;
  std::visit(Fields{out}, op);
  return 0;
}

// the optimisers, on a function handed in by the caller
typedef double (*ref_value_fn)(double x, void* ctx);
typedef void (*ref_derivative_fn)(double x, void* ctx, double* out);  // out[0] = f, out[1] = f', out[2] = f''
void ref_brent_minimize(ref_value_fn f, void* ctx, double guess, double min, double max, int significant_digits,
                        uint64_t max_iter, double step_size, double* x, double* fx) {
  const auto [rx, rfx] = Optimization::BrentMinimize<double>([&](double v) { return f(v, ctx); }, guess, min, max,
                                                              significant_digits, (size_t)max_iter, step_size);
  *x = rx;
  *fx = rfx;
}
void ref_brent_minimize_with_gradients(ref_derivative_fn f, void* ctx, double guess, double min, double max,
                                       int significant_digits, uint64_t max_iter, double step_size, double* x, double* fx) {
  const auto [rx, rfx] = Optimization::BrentMinimizeWithGradients<double>(
      [&](double v) {
        double o[3];
        f(v, ctx, o);
        return std::make_pair(o[0], o[1]);
      },
      guess, min, max, significant_digits, (size_t)max_iter, step_size);
  *x = rx;
  *fx = rfx;
}
double ref_gradient_ascent(ref_derivative_fn f, void* ctx, double x, int significant_digits, double step_size, double min_x,
                           uint64_t max_iter) {
  return Optimization::GradientAscent(
      [&](double v) {
        double o[3];
        f(v, ctx, o);
        return std::make_pair(o[0], o[1]);
      },
      x, significant_digits, step_size, min_x, (size_t)max_iter);
}
double ref_logspace_gradient_ascent(ref_derivative_fn f, void* ctx, double x, int significant_digits, double step_size,
                                    double min_x, uint64_t max_iter) {
  return Optimization::LogSpaceGradientAscent(
      [&](double v) {
        double o[3];
        f(v, ctx, o);
        return std::make_pair(o[0], o[1]);
      },
      x, significant_digits, step_size, min_x, (size_t)max_iter);
}
double ref_newton(ref_derivative_fn f, void* ctx, double x, int significant_digits, double epsilon, double min_x, double max_x,
                  uint64_t max_iter) {
  return Optimization::NewtonRaphsonOptimization(
      [&](double v) {
        double o[3];
        f(v, ctx, o);
        return std::make_tuple(o[0], o[1], o[2]);
      },
      x, significant_digits, epsilon, min_x, max_x, (size_t)max_iter);
}

}  // extern "C"
