// beagle_client.cpp -- a C++ translation unit that sees ONLY include/bito_amd_beagle.h (no HIP, no torch, plain g++) and
// drives the 17 BEAGLE entry points of libbito_amd.so in the order, and with the buffer / matrix / scale-buffer index
// arithmetic, that bito's FatBeagle uses (SURVEY.md section 8b seam 1; reference src/fat_beagle.cpp:218-373 for the
// instance set-up, :49-69 log-likelihood, :113-169 gradient).  Written from that description -- it is the proof that the
// header compiles as C++ and that the symbols link and behave when called the reference's way; src/fat_beagle.cpp itself
// cannot be compiled here (Eigen is not in the image).
//
// Buffer map of an instance for n taxa (N = 2n - 1 nodes): partials buffer of node v is v (tips 0..n-1 hold compact
// states, or tip partials with use_tip_states = 0: then n more buffers); pre-order partials of node v at v + N; the
// transition matrix of the branch above node v at v, the differential matrix at N - 1; scale buffer of internal node v
// at v - n + 1 (post-order) and v + 1 + (n - 1) (pre-order), cumulative scale buffer 0.
//
// usage: beagle_client <case file> [use_tip_states = 1] [rescaling = 0]
// case file (whitespace separated): n P C | patterns [n][P] | weights [P] | V [16] Vinv [16] lambda [4] pi [4] Q [16] |
//   category rates [C] | category weights [C] | parent ids [2n-3] (unrooted, trifurcating root 2n-3) | branch lengths [2n-2]
// prints: "impl <name>", "ll <value>", "ll_from_gradient <value>", "gradient <2n-1 values>"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../include/bito_amd_beagle.h"

namespace {

struct Case {
  int n = 0, P = 0, C = 0;
  std::vector<int> patterns;
  std::vector<double> weights, V, Vinv, lambda, pi, Q, rates, props, branch_lengths;
  std::vector<int> parent_ids;
};

template <typename T>
void Read(std::istream& in, std::vector<T>& v, size_t count) {
  v.resize(count);
  for (auto& x : v)
    if (!(in >> x)) throw std::runtime_error("case file is short");
}

void Check(int rc, const char* what) {
  if (rc != BEAGLE_SUCCESS) throw std::runtime_error(std::string(what) + " returned " + std::to_string(rc));
}

// one BEAGLE instance and the model state kept beside it
class Client {
 public:
  Client(const Case& c, bool use_tip_states) : c_(c), n_(c.n), N_(2 * c.n - 1) {
    const int partials = 3 * n_ - 2 + (use_tip_states ? 0 : n_);
    BeagleInstanceDetails info{};
    inst_ = beagleCreateInstance(n_, partials, use_tip_states ? n_ : 0, 4, c.P, 1, 2 * N_, c.C, partials + 1, nullptr, 0,
                                 BEAGLE_FLAG_VECTOR_SSE, BEAGLE_FLAG_SCALING_MANUAL, &info);
    if (inst_ < 0) throw std::runtime_error("beagleCreateInstance returned " + std::to_string(inst_));
    if (!(info.flags & (BEAGLE_FLAG_PROCESSOR_CPU | BEAGLE_FLAG_PROCESSOR_GPU)))
      throw std::runtime_error("the instance reports neither a CPU nor a GPU implementation");
    impl_ = info.implName ? info.implName : "";
    for (int tip = 0; tip < n_; tip++) {
      if (use_tip_states) {
        Check(beagleSetTipStates(inst_, tip, c.patterns.data() + (size_t)tip * c.P), "beagleSetTipStates");
      } else {
        std::vector<double> part((size_t)c.P * 4, 0.0);
        for (int p = 0; p < c.P; p++) {
          const int s = c.patterns[(size_t)tip * c.P + p];
          for (int i = 0; i < 4; i++) part[(size_t)p * 4 + i] = (s >= 4 || s == i) ? 1.0 : 0.0;
        }
        Check(beagleSetTipPartials(inst_, tip, part.data()), "beagleSetTipPartials");
      }
    }
    Check(beagleSetPatternWeights(inst_, c.weights.data()), "beagleSetPatternWeights");
    Check(beagleSetCategoryWeights(inst_, 0, c.props.data()), "beagleSetCategoryWeights");
    Check(beagleSetCategoryRates(inst_, c.rates.data()), "beagleSetCategoryRates");
    Check(beagleSetStateFrequencies(inst_, 0, c.pi.data()), "beagleSetStateFrequencies");
    Check(beagleSetEigenDecomposition(inst_, 0, c.V.data(), c.Vinv.data(), c.lambda.data()), "beagleSetEigenDecomposition");
  }
  ~Client() { (void)beagleFinalizeInstance(inst_); }
  const std::string& impl() const { return impl_; }

  double LogLikelihood(bool rescaling) {
    Detrifurcate();
    Check(beagleResetScaleFactors(inst_, 0), "beagleResetScaleFactors");
    UpdateMatrices();
    const auto ops = PostOrderOperations(rescaling);
    Check(beagleUpdatePartials(inst_, ops.data(), (int)ops.size(), rescaling ? 0 : BEAGLE_OP_NONE), "beagleUpdatePartials");
    return RootLogLikelihood(rescaling);
  }

  std::pair<double, std::vector<double>> Gradient(bool rescaling) {
    Detrifurcate();
    const int root = N_ - 1, fixed = kids_[root].second;
    Check(beagleResetScaleFactors(inst_, 0), "beagleResetScaleFactors");
    UpdateMatrices();
    // the root's pre-order partials are the stationary frequencies at every (category, pattern)
    std::vector<double> root_pre((size_t)c_.C * c_.P * 4);
    for (size_t k = 0; k < root_pre.size(); k++) root_pre[k] = c_.pi[k % 4];
    Check(beagleSetPartials(inst_, root + N_, root_pre.data()), "beagleSetPartials");
    // the differential matrix r_c Q of every category, in the slot behind the branches' matrices
    std::vector<double> dq((size_t)c_.C * 16);
    for (int cat = 0; cat < c_.C; cat++)
      for (int k = 0; k < 16; k++) dq[(size_t)cat * 16 + k] = c_.Q[k] * c_.rates[cat];
    const int dmat = N_ - 1;
    Check(beagleSetDifferentialMatrix(inst_, dmat, dq.data()), "beagleSetDifferentialMatrix");
    const auto post = PostOrderOperations(rescaling);
    Check(beagleUpdatePartials(inst_, post.data(), (int)post.size(), rescaling ? 0 : BEAGLE_OP_NONE), "beagleUpdatePartials");
    const auto pre = PreOrderOperations(rescaling);
    Check(beagleUpdatePrePartials(inst_, pre.data(), (int)pre.size(), BEAGLE_OP_NONE), "beagleUpdatePrePartials");
    std::vector<int> post_idx(N_ - 1), pre_idx(N_ - 1), dmat_idx(N_ - 1, dmat);
    for (int v = 0; v < N_ - 1; v++) {
      post_idx[v] = v;
      pre_idx[v] = v + N_;
    }
    const int zero = 0;
    std::vector<double> gradient(N_, 0.0);
    Check(beagleCalculateEdgeDerivatives(inst_, post_idx.data(), pre_idx.data(), dmat_idx.data(), &zero, N_ - 1, nullptr,
                                         gradient.data(), nullptr),
          "beagleCalculateEdgeDerivatives");
    const double ll = RootLogLikelihood(rescaling);
    gradient[fixed] = 0.0;  // (the branch the detrifurcation pins to zero)
    return {ll, gradient};
  }

 private:
  // the unrooted tree's trifurcation at node 2n - 3 becomes two bifurcations: children 1 and 2 stay under the old root id
  // (branch length 0), a new root 2n - 2 joins child 0 with it
  void Detrifurcate() {
    const int M = 2 * n_ - 2, r = M - 1;
    std::map<int, std::vector<int>> lists;
    for (int child = 0; child < M - 1; child++) lists[c_.parent_ids[child]].push_back(child);
    kids_.clear();
    for (auto& kv : lists)
      if (kv.first != r) {
        if (kv.second.size() != 2) throw std::runtime_error("not a bifurcating node");
        kids_[kv.first] = {kv.second[0], kv.second[1]};
      }
    const auto& top = lists[r];
    if (top.size() != 3) throw std::runtime_error("the root is not a trifurcation");
    kids_[r] = {top[1], top[2]};
    kids_[r + 1] = {top[0], r};
    lengths_.assign(N_, 0.0);
    for (int v = 0; v < M; v++) lengths_[v] = c_.branch_lengths[v];
    lengths_[r] = 0.0;
  }

  void UpdateMatrices() {
    std::vector<int> idx(N_ - 1);
    for (int v = 0; v < N_ - 1; v++) idx[v] = v;
    Check(beagleUpdateTransitionMatrices(inst_, 0, idx.data(), nullptr, nullptr, lengths_.data(), N_ - 1),
          "beagleUpdateTransitionMatrices");
  }

  // children before parents, a node's first child's subtree first
  void PostOrder(int node, std::vector<int>* out) const {
    auto it = kids_.find(node);
    if (it == kids_.end()) return;
    PostOrder(it->second.first, out);
    PostOrder(it->second.second, out);
    out->push_back(node);
  }
  std::vector<BeagleOperation> PostOrderOperations(bool rescaling) const {
    std::vector<int> order;
    PostOrder(N_ - 1, &order);
    std::vector<BeagleOperation> ops;
    for (int node : order) {
      const auto& ch = kids_.at(node);
      ops.push_back({node, rescaling ? node - n_ + 1 : BEAGLE_OP_NONE, BEAGLE_OP_NONE, ch.first, ch.first, ch.second, ch.second});
    }
    return ops;
  }
  // parents before children: for every non-root node (node, sister, parent), a node's first child, that child's
  // subtree, then the second child
  void PreOrder(int node, std::vector<BeagleOperation>* ops, bool rescaling) const {
    auto it = kids_.find(node);
    if (it == kids_.end()) return;
    const int c0 = it->second.first, c1 = it->second.second;
    const int order[2][2] = {{c0, c1}, {c1, c0}};
    for (const auto& pair : order) {
      const int child = pair[0], sister = pair[1];
      ops->push_back({child + N_, rescaling ? child + 1 + (n_ - 1) : BEAGLE_OP_NONE, BEAGLE_OP_NONE, node + N_, child, sister, sister});
      PreOrder(child, ops, rescaling);
    }
  }
  std::vector<BeagleOperation> PreOrderOperations(bool rescaling) const {
    std::vector<BeagleOperation> ops;
    PreOrder(N_ - 1, &ops, rescaling);
    return ops;
  }
  double RootLogLikelihood(bool rescaling) {
    const int root = N_ - 1, zero = 0, cumulative = rescaling ? 0 : BEAGLE_OP_NONE;
    double out = 0.0;
    Check(beagleCalculateRootLogLikelihoods(inst_, &root, &zero, &zero, &cumulative, 1, &out), "beagleCalculateRootLogLikelihoods");
    return out;
  }

  const Case& c_;
  int n_, N_, inst_ = -1;
  std::string impl_;
  std::map<int, std::pair<int, int>> kids_;
  std::vector<double> lengths_;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <case file> [use_tip_states] [rescaling]\n", argv[0]);
    return 2;
  }
  try {
    std::ifstream in(argv[1]);
    if (!in) throw std::runtime_error("cannot read the case file");
    Case c;
    if (!(in >> c.n >> c.P >> c.C)) throw std::runtime_error("case file is short");
    Read(in, c.patterns, (size_t)c.n * c.P);
    Read(in, c.weights, (size_t)c.P);
    Read(in, c.V, 16);
    Read(in, c.Vinv, 16);
    Read(in, c.lambda, 4);
    Read(in, c.pi, 4);
    Read(in, c.Q, 16);
    Read(in, c.rates, (size_t)c.C);
    Read(in, c.props, (size_t)c.C);
    Read(in, c.parent_ids, (size_t)(2 * c.n - 3));
    Read(in, c.branch_lengths, (size_t)(2 * c.n - 2));
    const bool use_tip_states = argc < 3 || std::atoi(argv[2]) != 0, rescaling = argc > 3 && std::atoi(argv[3]) != 0;
    Client client(c, use_tip_states);
    std::printf("impl %s\n", client.impl().c_str());
    std::printf("ll %.17g\n", client.LogLikelihood(rescaling));
    const auto result = client.Gradient(rescaling);
    std::printf("ll_from_gradient %.17g\ngradient", result.first);
    for (double g : result.second) std::printf(" %.17g", g);
    std::printf("\n");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "beagle_client: %s\n", e.what());
    return 1;
  }
}
