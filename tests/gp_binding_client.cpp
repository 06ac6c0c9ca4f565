// gp_binding_client.cpp -- seam 3 as a bito maintainer would bind it, compiled against the REFERENCE's own operation types:
// this translation unit includes the reference's src/gp_operation.hpp (it needs neither BEAGLE nor Eigen) and
// include/bito_amd_gp.h, builds a GPOperationVector out of the reference's structs, flattens it with the visitor of
// INTEGRATION.md ("Seam 3") and hands it to bito_amd_gp_process_operations -- what GPEngine::ProcessOperations
// (src/gp_engine.hpp:74, src/gp_engine.cpp:213-339) becomes.  Built by oracle/Makefile (target ref) where the reference is
// present, into oracle/_ref/gp_binding_client.bin; no HIP, no torch, plain g++.
//
// usage: gp_binding_client <case file>
// case file (whitespace separated): taxa P nodes gpcsps | patterns [taxa][P] | weights [P] | branch lengths [gpcsps] |
//   q [gpcsps] | stream count | per stream: op count, then per operation: opcode a b c [count sources...]
// prints: "marginal <value>", "per_gpcsp <gpcsps values>", "branch_lengths <gpcsps values>"
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "gp_operation.hpp"

#include "../include/bito_amd_gp.h"

namespace {

// INTEGRATION.md, "Seam 3": one record per operation, opcode = the alternative index of the reference's variant
struct Flatten {
  std::vector<bito_amd_gp_op>& ops;
  std::vector<uint64_t>& side;
  void operator()(const GPOperations::ZeroPLV& o) { ops.push_back({0, 0, o.dest_, 0, 0}); }
  void operator()(const GPOperations::SetToStationaryDistribution& o) { ops.push_back({1, 0, o.dest_, o.root_gpcsp_idx_, 0}); }
  void operator()(const GPOperations::IncrementWithWeightedEvolvedPLV& o) { ops.push_back({2, 0, o.dest_, o.gpcsp_, o.src_}); }
  void operator()(const GPOperations::Multiply& o) { ops.push_back({3, 0, o.dest_, o.src1_, o.src2_}); }
  void operator()(const GPOperations::Likelihood& o) { ops.push_back({4, 0, o.dest_, o.child_, o.parent_}); }
  void operator()(const GPOperations::OptimizeBranchLength& o) { ops.push_back({5, 0, o.leafward_, o.rootward_, o.gpcsp_}); }
  void operator()(const GPOperations::UpdateSBNProbabilities& o) { ops.push_back({6, 0, o.start_, o.stop_, 0}); }
  void operator()(const GPOperations::ResetMarginalLikelihood&) { ops.push_back({7, 0, 0, 0, 0}); }
  void operator()(const GPOperations::IncrementMarginalLikelihood& o) {
    ops.push_back({8, 0, o.stationary_times_prior_, o.rootsplit_, o.p_});
  }
  void operator()(const GPOperations::PrepForMarginalization& o) {
    ops.push_back({9, (uint32_t)o.src_vector_.size(), o.dest_, side.size(), 0});
    side.insert(side.end(), o.src_vector_.begin(), o.src_vector_.end());
  }
};

void Check(bito_amd_gp_engine* e, int rc, const char* what) {
  if (rc) throw std::runtime_error(std::string(what) + ": " + (e ? bito_amd_gp_last_error(e) : "") + " (" + std::to_string(rc) + ")");
}

// GPEngine::ProcessOperations over the C ABI
void ProcessOperations(bito_amd_gp_engine* e, const GPOperationVector& operations) {
  std::vector<bito_amd_gp_op> ops;
  std::vector<uint64_t> side;
  for (const auto& op : operations) std::visit(Flatten{ops, side}, op);
  Check(e, bito_amd_gp_process_operations(e, ops.data(), (int64_t)ops.size(), side.data(), (int64_t)side.size()), "ProcessOperations");
}

template <typename T>
std::vector<T> Read(std::istream& in, size_t count) {
  std::vector<T> v(count);
  for (auto& x : v)
    if (!(in >> x)) throw std::runtime_error("case file is short");
  return v;
}

// the reference's operation for one record of the case file
GPOperation Operation(std::istream& in) {
  using namespace GPOperations;
  size_t opcode = 0, a = 0, b = 0, c = 0;
  if (!(in >> opcode >> a >> b >> c)) throw std::runtime_error("case file is short");
  switch (opcode) {
    case 0: return ZeroPLV{a};
    case 1: return SetToStationaryDistribution{a, b};
    case 2: return IncrementWithWeightedEvolvedPLV{a, b, c};
    case 3: return Multiply{a, b, c};
    case 4: return Likelihood{a, b, c};
    case 5: return OptimizeBranchLength{a, b, c};
    case 6: return UpdateSBNProbabilities{a, b};
    case 7: return ResetMarginalLikelihood{};
    case 8: return IncrementMarginalLikelihood{a, b, c};
    case 9: {
      size_t count = 0;
      in >> count;
      return PrepForMarginalization{a, Read<size_t>(in, count)};
    }
    default: throw std::runtime_error("unknown opcode " + std::to_string(opcode));
  }
}

}  // namespace

int main(int argc, char** argv) {
  try {
    if (argc < 2) throw std::runtime_error("usage: gp_binding_client <case file>");
    std::ifstream in(argv[1]);
    int taxa = 0, P = 0, nodes = 0, gpcsps = 0;
    if (!(in >> taxa >> P >> nodes >> gpcsps)) throw std::runtime_error("cannot read the case file");
    const auto patterns = Read<int32_t>(in, (size_t)taxa * P);
    const auto weights = Read<double>(in, (size_t)P);
    const auto lengths = Read<double>(in, (size_t)gpcsps);
    const auto q = Read<double>(in, (size_t)gpcsps);
    bito_amd_gp_engine* e = nullptr;
    char err[512] = "";
    const int rc = bito_amd_gp_create(0, taxa, P, patterns.data(), weights.data(), nodes, gpcsps, 1e-40, &e, err, sizeof(err));
    if (rc) throw std::runtime_error(std::string("bito_amd_gp_create: ") + err);
    Check(e, bito_amd_gp_set_branch_lengths(e, lengths.data()), "SetBranchLengths");
    Check(e, bito_amd_gp_set_sbn_parameters(e, q.data()), "SetSBNParameters");
    size_t streams = 0;
    in >> streams;
    for (size_t s = 0; s < streams; s++) {
      size_t count = 0;
      in >> count;
      GPOperationVector operations;
      for (size_t k = 0; k < count; k++) operations.push_back(Operation(in));
      ProcessOperations(e, operations);
    }
    double marginal = 0;
    std::vector<double> per((size_t)gpcsps), after((size_t)gpcsps);
    Check(e, bito_amd_gp_log_marginal_likelihood(e, &marginal), "GetLogMarginalLikelihood");
    Check(e, bito_amd_gp_per_gpcsp_log_likelihoods(e, per.data()), "GetPerGPCSPLogLikelihoods");
    Check(e, bito_amd_gp_get_branch_lengths(e, after.data()), "GetBranchLengths");
    std::printf("marginal %.17g\nper_gpcsp", marginal);
    for (double v : per) std::printf(" %.17g", v);
    std::printf("\nbranch_lengths");
    for (double v : after) std::printf(" %.17g", v);
    std::printf("\n");
    bito_amd_gp_destroy(e);
    return 0;
  } catch (const std::exception& err) {
    std::fprintf(stderr, "gp_binding_client: %s\n", err.what());
    return 1;
  }
}
