// probe_mfma16.hip -- measures, on the device, two facts gs_kernels.hip relies on:
//  (1) the result lane layout of v_mfma_f64_16x16x4 given A lane = 16 k + i, B lane = 16 k + j:
//      prints, for every (lane / 16, register), the row of D it holds;
//  (2) whether a chain of these instructions over 16 k-steps rounds exactly like a sequential
//      fma() chain over k = 0..63 (it decides if P(t) may be built on the matrix pipe while staying
//      bit-identical to the fixed-order CPU arithmetic).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A, const double* B, double* D) {  // A [16][64], B [64][16], D [64 lanes][4]
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  v4d acc = {0, 0, 0, 0};
  for (int ks = 0; ks < 16; ks++) {
    const double a = A[i * 64 + 4 * ks + q];        // A[i][k], k = 4 ks + q
    const double b = B[(4 * ks + q) * 16 + i];      // B[k][j], j = lane & 15
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; r++) D[lane * 4 + r] = acc[r];
}

int main() {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  std::vector<double> A(16 * 64), B(64 * 16), D(256), ref(256);
  for (auto& x : A) x = u(rng) * std::exp(8 * u(rng));
  for (auto& x : B) x = u(rng) * std::exp(8 * u(rng));
  double *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 8);
  hipMalloc(&dB, B.size() * 8);
  hipMalloc(&dD, D.size() * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> mag(256);
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      double s = 0, m = 0;
      for (int k = 0; k < 64; k++) {
        s = std::fma(A[i * 64 + k], B[k * 16 + j], s);
        m += std::fabs(A[i * 64 + k] * B[k * 16 + j]);
      }
      ref[i * 16 + j] = s;
      mag[i * 16 + j] = m;
    }
  // where does each (lane, register) come from?
  int exact = 0, found = 0;
  int row_of[4][4];
  bool consistent = true;
  for (int lane = 0; lane < 64; lane++)
    for (int r = 0; r < 4; r++) {
      const double got = D[lane * 4 + r];
      int hit = -1;
      for (int e = 0; e < 256; e++)
        if (std::fabs(got - ref[e]) <= 1e-13 * mag[e]) hit = e;
      if (hit < 0) continue;
      found++;
      if (std::memcmp(&got, &ref[hit], 8) == 0) exact++;
      const int i = hit / 16, j = hit % 16;
      if (j != (lane & 15)) consistent = false;
      if ((lane & 15) == 0) row_of[lane >> 4][r] = i;
      else if (row_of[lane >> 4][r] != i) consistent = false;
    }
  std::printf("{\"matched\": %d, \"of\": 256, \"column_is_lane_mod_16\": %s, \"bit_equal_to_sequential_fma_chain\": %d,\n"
              " \"row_of[lane/16][register]\": [", found, consistent ? "true" : "false", exact);
  for (int q = 0; q < 4; q++)
    std::printf("[%d,%d,%d,%d]%s", row_of[q][0], row_of[q][1], row_of[q][2], row_of[q][3], q < 3 ? "," : "]}\n");
  return 0;
}
