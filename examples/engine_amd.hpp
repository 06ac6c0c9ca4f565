// engine_amd.hpp -- header-only C++ wrapper over the C ABI of libbito_amd.so with the shape of the
// reference's Engine (src/engine.hpp:26-68): construct from a model specification and a compressed
// alignment, then LogLikelihoods / Gradients over a whole tree collection in wire format.  Errors
// become std::runtime_error like the reference's Failwith (src/sugar.hpp:119-130).  This is the
// code INTEGRATION.md's src/engine_amd.cpp would be built from, minus bito's own container types;
// tests/cabi_client.cpp drives it on the GPU box (plain g++, no HIP headers needed).
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../include/bito_amd.h"

namespace bito_amd_cpp {

struct PhyloModelSpecification {  // src/phylo_model.hpp:13-17
  std::string substitution_, site_, clock_;
};

struct TreeBatch {                 // a tree collection in wire format
  int32_t tree_count = 0, node_count = 0;
  bool rooted = false;
  std::vector<int32_t> parent_ids;     // [tree_count][node_count-1]  Node::ParentIdVector
  std::vector<double> branch_lengths;  // [tree_count][node_count]    by child id
  std::vector<double> rates;           // rooted: [tree_count][node_count-1], may be empty
};

struct PhyloGradient {             // src/phylo_gradient.hpp:10-35
  double log_likelihood_ = 0;
  std::map<std::string, std::vector<double>> gradient_;
};

class Engine {
 public:
  Engine(const PhyloModelSpecification& spec, int32_t taxon_count, int32_t pattern_count,
         const std::vector<int32_t>& patterns, const std::vector<double>& weights, int32_t device_id = 0,
         int32_t device_count = 1, const std::vector<int32_t>& devices = {}) {
    // device_count stands where the reference has thread_count (src/engine.hpp:20-24): how many FatBeagle
    // instances -- here GPUs -- serve one call
    // ("Thread count needs to be strictly positive.", src/engine.cpp:14-16 -- the C ABI itself reads a 0 as its
    // default, one device, so that a zero-initialised spec works)
    if (device_count < 1) throw std::runtime_error("Device count needs to be strictly positive.");
    bito_amd_engine_spec es{device_id, 1, 0, device_count, /*host_threads=*/0, devices.empty() ? nullptr : devices.data()};
    char err[512] = {0};
    const int rc = bito_amd_engine_create(&es, spec.substitution_.c_str(), spec.site_.c_str(), spec.clock_.c_str(),
                                          taxon_count, pattern_count, patterns.data(), weights.data(), &e_, err,
                                          sizeof(err));
    if (rc != BITO_AMD_OK) throw std::runtime_error(err);
    n_ = taxon_count;
    for (int32_t i = 0; i < bito_amd_engine_block_count(e_); i++) {
      char name[64];
      int32_t start = 0, len = 0;
      bito_amd_engine_block(e_, i, name, sizeof(name), &start, &len);
      blocks_[name] = {start, len};
    }
  }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  ~Engine() { bito_amd_engine_destroy(e_); }

  int32_t ParameterCount() const { return bito_amd_engine_param_count(e_); }
  int32_t DeviceCount() const { return bito_amd_engine_device_count(e_); }
  const std::map<std::string, std::pair<int32_t, int32_t>>& BlockMap() const { return blocks_; }

  // Engine::LogLikelihoods (src/engine.cpp:58-74); params is [tree_count][ParameterCount()] row-major
  std::vector<double> LogLikelihoods(const TreeBatch& t, const std::vector<double>& params, bool rescaling) const {
    std::vector<double> out(t.tree_count);
    Check(bito_amd_engine_log_likelihoods(e_, t.tree_count, t.rooted, t.node_count, t.parent_ids.data(),
                                          t.branch_lengths.data(), t.rates.empty() ? nullptr : t.rates.data(),
                                          params.empty() ? nullptr : params.data(), rescaling, out.data()));
    return out;
  }

  // Engine::Gradients (src/engine.cpp:94-110)
  std::vector<PhyloGradient> Gradients(const TreeBatch& t, const std::vector<double>& params, bool rescaling,
                                       int32_t flags = 0) const {
    const size_t T = t.tree_count, N = 2 * (size_t)n_ - 1;
    std::vector<double> ll(T), branch(T * N), site(T), clock(T);
    const int32_t sub_len = blocks_.count("entire_substitution") ? blocks_.at("entire_substitution").second : 0;
    std::vector<double> subst(T * (size_t)(sub_len > 0 ? sub_len : 1));
    Check(bito_amd_engine_gradients(e_, t.tree_count, t.rooted, t.node_count, t.parent_ids.data(),
                                    t.branch_lengths.data(), t.rates.empty() ? nullptr : t.rates.data(),
                                    params.empty() ? nullptr : params.data(), rescaling, flags, 1e-6, ll.data(),
                                    branch.data(), site.data(), subst.data(), clock.data()));
    std::vector<PhyloGradient> out(T);
    for (size_t i = 0; i < T; i++) {
      out[i].log_likelihood_ = ll[i];
      out[i].gradient_["branch_lengths"].assign(branch.begin() + i * N, branch.begin() + (i + 1) * N);
      if ((flags & BITO_AMD_GRAD_SITE_MODEL) && bito_amd_engine_category_count(e_) > 1)
        out[i].gradient_["site_model"] = {site[i]};
      if ((flags & BITO_AMD_GRAD_CLOCK_MODEL) && t.rooted) out[i].gradient_["clock_model"] = {clock[i]};
    }
    return out;
  }

 private:
  void Check(int rc) const {
    if (rc != BITO_AMD_OK) throw std::runtime_error(bito_amd_engine_last_error(e_));
  }
  bito_amd_engine* e_ = nullptr;
  int32_t n_ = 0;
  std::map<std::string, std::pair<int32_t, int32_t>> blocks_;
};

}  // namespace bito_amd_cpp
