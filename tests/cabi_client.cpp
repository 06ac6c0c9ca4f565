// cabi_client.cpp -- a C++ program that uses libbito_amd.so through examples/engine_amd.hpp and
// nothing else (no Python, no torch, no HIP headers): the proof that the drop-in boundary is a
// plain C ABI.  Reads a text case written by tests/test_cabi_client.py:
//   substitution site clock
//   n P
//   patterns (n*P ints)   weights (P doubles)
//   rooted T node_count   parent ids (T*(node_count-1))   branch lengths (T*node_count)
//   param_count           params (T*param_count)
// and prints "ll <value>" per tree followed by "grad <2n-1 values>" per tree with %.17g.
// A second argument is the number of device slots the engine is created over (EngineSpecification's thread_count
// in the reference, src/engine.hpp:20-24); every slot names GPU 0, so a one-GPU machine runs the multi-device path.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "../examples/engine_amd.hpp"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s case.txt\n", argv[0]);
    return 2;
  }
  std::ifstream in(argv[1]);
  bito_amd_cpp::PhyloModelSpecification spec;
  int n = 0, P = 0;
  in >> spec.substitution_ >> spec.site_ >> spec.clock_ >> n >> P;
  std::vector<int32_t> patterns((size_t)n * P);
  std::vector<double> weights(P);
  for (auto& v : patterns) in >> v;
  for (auto& v : weights) in >> v;
  bito_amd_cpp::TreeBatch trees;
  int rooted = 0;
  in >> rooted >> trees.tree_count >> trees.node_count;
  trees.rooted = rooted != 0;
  trees.parent_ids.resize((size_t)trees.tree_count * (trees.node_count - 1));
  trees.branch_lengths.resize((size_t)trees.tree_count * trees.node_count);
  for (auto& v : trees.parent_ids) in >> v;
  for (auto& v : trees.branch_lengths) in >> v;
  int pc = 0;
  in >> pc;
  std::vector<double> params((size_t)trees.tree_count * pc);
  for (auto& v : params) in >> v;
  if (!in) {
    std::fprintf(stderr, "malformed case file\n");
    return 2;
  }
  try {
    const int device_count = argc > 2 ? std::atoi(argv[2]) : 1;
    bito_amd_cpp::Engine engine(spec, n, P, patterns, weights, /*device_id=*/0, device_count,
                                std::vector<int32_t>(device_count > 0 ? device_count : 0, 0));
    std::printf("devices %d\n", engine.DeviceCount());
    if (engine.ParameterCount() != pc) throw std::runtime_error("parameter count mismatch");
    const auto ll = engine.LogLikelihoods(trees, params, false);
    const auto grads = engine.Gradients(trees, params, false);
    for (size_t t = 0; t < ll.size(); t++) std::printf("ll %.17g\n", ll[t]);
    for (const auto& g : grads) {
      std::printf("grad");
      for (double v : g.gradient_.at("branch_lengths")) std::printf(" %.17g", v);
      std::printf("\n");
    }
    // error path: a bad tree must surface as an exception carrying the engine's message
    bito_amd_cpp::TreeBatch bad = trees;
    bad.parent_ids[0] = 0;
    try {
      engine.LogLikelihoods(bad, params, false);
      std::printf("error-path MISSING\n");
    } catch (const std::runtime_error& e) {
      std::printf("error-path ok: %s\n", e.what());
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
