// microbench.hip -- gfx950 FP64 pipe probes that size the traversal kernels:
//   * v_fma_f64 issue rate
//   * v_mfma_f64_4x4x4_4b / v_mfma_f64_16x16x4 issue rate and dependent latency
//   * whether the FP64 MFMA and FP64 VALU pipes overlap (same wave / sibling waves)
//   * the lane <-> (block, row, col) layout of v_mfma_f64_4x4x4_4b operands
//   * LDS ds_read_b64 / ds_write_b64 streaming rate from a 1-wave-per-SIMD kernel
// Prints one JSON object.  Build: hipcc --offload-arch=gfx950 -O3 microbench.hip -o microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t rc_ = (x);                                                        \
    if (rc_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(rc_));            \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void __launch_bounds__(256) fma_kernel(double* out, int iters, double a, double b) {
  double acc[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) acc[i] = threadIdx.x * 1e-9 + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += acc[i];
  if (s == 12345.678) out[0] = s;
}

template <int CHAINS>
__global__ void __launch_bounds__(256) mfma4_kernel(double* out, int iters, double a, double b) {
  double acc[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) acc[i] = threadIdx.x * 1e-9 + i;
  double av = a + threadIdx.x * 1e-12, bv = b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += acc[i];
  if (s == 12345.678) out[0] = s;
}

template <int CHAINS>
__global__ void __launch_bounds__(256) mfma16_kernel(double* out, int iters, double a, double b) {
  double4_t acc[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) acc[i] = double4_t{threadIdx.x * 1e-9 + i, 0, 0, 0};
  double av = a + threadIdx.x * 1e-12, bv = b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}

// MFMA and VALU FP64 in the same wave, interleaved: M mfma + F fma per iteration.
template <int M, int F>
__global__ void __launch_bounds__(256) mixed_kernel(double* out, int iters, double a, double b) {
  double macc[M > 0 ? M : 1], facc[F > 0 ? F : 1];
#pragma unroll
  for (int i = 0; i < M; i++) macc[i] = threadIdx.x * 1e-9 + i;
#pragma unroll
  for (int i = 0; i < F; i++) facc[i] = threadIdx.x * 1e-9 - i;
  double av = a + threadIdx.x * 1e-12, bv = b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < (M > F ? M : F); i++) {
      if (i < M) macc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, macc[i], 0, 0, 0);
      if (i < F) facc[i] = __builtin_fma(facc[i], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < M; i++) s += macc[i];
#pragma unroll
  for (int i = 0; i < F; i++) s += facc[i];
  if (s == 12345.678) out[0] = s;
}

// Half the waves of a block run MFMA only, the other half FMA only.
__global__ void __launch_bounds__(512) split_kernel(double* out, int iters, double a, double b) {
  const int wave = threadIdx.x >> 6;
  double acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = threadIdx.x * 1e-9 + i;
  double av = a + threadIdx.x * 1e-12, bv = b;
  if (wave & 1) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_fma(acc[i], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i];
  if (s == 12345.678) out[0] = s;
}

__global__ void layout_kernel(int* out) {
  // For every (la, lb): A = unit at lane la, B = unit at lane lb; record which lane of D is 1.
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 64 + lb] = m ? __ffsll((long long)m) - 1 : -1;
    }
}

// LDS streaming: each lane reads 8 doubles / writes 4 doubles of its own column per iteration.
__global__ void __launch_bounds__(256) lds_kernel(double* out, int iters, int rows) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < rows * nt; i += nt) lds[i] = i * 1e-9;
  __syncthreads();
  double acc = 0;
  int r = 0;
  for (int it = 0; it < iters; it++) {
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = lds[((r + k) % rows) * nt + tid];
    const double y = (x[0] + x[1]) * (x[2] + x[3]) + (x[4] + x[5]) * (x[6] + x[7]);
#pragma unroll
    for (int k = 0; k < 4; k++) lds[((r + 8 + k) % rows) * nt + tid] = y + k;
    acc += y;
    r = (r + 12) % rows;
  }
  if (acc == 12345.678) out[0] = acc;
}

template <typename F>
static double TimeMs(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < reps; r++) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double* out;
  CHECK(hipMalloc(&out, 1024));
  const int iters = 20000;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d,\n", prop.gcnArchName, cus, prop.clockRate / 1000);

  // blocks: 4 per CU of 256 threads => 4 waves per SIMD
  auto tf = [&](double flops, double ms) { return flops / (ms * 1e-3) / 1e12; };
  for (int wps : {1, 2, 4}) {
    const dim3 grid(cus * wps), block(256);
    const double waves = (double)cus * wps * 4;
    double ms = TimeMs([&] { hipLaunchKernelGGL(fma_kernel<8>, grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    printf(" \"fma_f64_8chains_wps%d_tflops\": %.2f,\n", wps, tf(waves * 64 * 2.0 * 8 * iters, ms));
    ms = TimeMs([&] { hipLaunchKernelGGL(mfma4_kernel<8>, grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    printf(" \"mfma_f64_4x4x4_8chains_wps%d_tflops\": %.2f,\n", wps, tf(waves * 512.0 * 8 * iters, ms));
    printf(" \"mfma_f64_4x4x4_8chains_wps%d_cycles_per_inst_at_2400\": %.2f,\n", wps,
           ms * 1e-3 * 2.4e9 / (8.0 * iters) / wps);
    ms = TimeMs([&] { hipLaunchKernelGGL(mfma16_kernel<4>, grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    printf(" \"mfma_f64_16x16x4_4chains_wps%d_tflops\": %.2f,\n", wps, tf(waves * 2048.0 * 4 * iters, ms));
  }
  {
    const dim3 grid(cus), block(256);
    double ms = TimeMs([&] { hipLaunchKernelGGL(mfma4_kernel<1>, grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    printf(" \"mfma_f64_4x4x4_dependent_cycles_at_2400\": %.2f,\n", ms * 1e-3 * 2.4e9 / iters);
    ms = TimeMs([&] { hipLaunchKernelGGL(fma_kernel<1>, grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    printf(" \"fma_f64_dependent_cycles_at_2400\": %.2f,\n", ms * 1e-3 * 2.4e9 / iters);
    ms = TimeMs([&] { hipLaunchKernelGGL(fma_kernel<2>, grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    printf(" \"fma_f64_2chains_cycles_per_inst_at_2400\": %.2f,\n", ms * 1e-3 * 2.4e9 / iters / 2);
    ms = TimeMs([&] { hipLaunchKernelGGL(mfma4_kernel<2>, grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    printf(" \"mfma_f64_4x4x4_2chains_cycles_per_inst_at_2400\": %.2f,\n", ms * 1e-3 * 2.4e9 / iters / 2);
    ms = TimeMs([&] { hipLaunchKernelGGL(mfma4_kernel<4>, grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    printf(" \"mfma_f64_4x4x4_4chains_cycles_per_inst_at_2400\": %.2f,\n", ms * 1e-3 * 2.4e9 / iters / 4);
  }
  {
    // one wave per SIMD: same-wave interleave of 4 mfma + 4 fma chains vs each alone
    const dim3 grid(cus), block(256);
    double m = TimeMs([&] { hipLaunchKernelGGL((mixed_kernel<4, 0>), grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    double f = TimeMs([&] { hipLaunchKernelGGL((mixed_kernel<0, 4>), grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    double mf = TimeMs([&] { hipLaunchKernelGGL((mixed_kernel<4, 4>), grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    double mf2 = TimeMs([&] { hipLaunchKernelGGL((mixed_kernel<4, 8>), grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    printf(" \"same_wave_ms\": {\"mfma4\": %.3f, \"fma4\": %.3f, \"mfma4_plus_fma4\": %.3f, \"mfma4_plus_fma8\": %.3f},\n", m, f, mf, mf2);
    const dim3 grid2(cus), block2(512);
    double sp = TimeMs([&] { hipLaunchKernelGGL(split_kernel, grid2, block2, 0, 0, out, iters, 1.0000001, 1e-9); });
    double m8 = TimeMs([&] { hipLaunchKernelGGL(mfma4_kernel<8>, grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    double f8 = TimeMs([&] { hipLaunchKernelGGL(fma_kernel<8>, grid, block, 0, 0, out, iters, 1.0000001, 1e-9); });
    printf(" \"sibling_waves_ms\": {\"mfma8_alone\": %.3f, \"fma8_alone\": %.3f, \"both_on_each_simd\": %.3f},\n", m8, f8, sp);
  }
  {
    for (int threads : {64, 128, 192, 256}) {
      const int rows = 72;
      const size_t bytes = (size_t)rows * threads * 8;
      const int it2 = 4000;
      CHECK(hipFuncSetAttribute((const void*)lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      double ms = TimeMs([&] { hipLaunchKernelGGL(lds_kernel, dim3(cus), dim3(threads), bytes, 0, out, it2, rows); });
      const double per_cu_bytes = (double)threads * 12 * 8 * it2;
      printf(" \"lds_rw_%dthr_bytes_per_clk_per_cu_at_2400\": %.1f,\n", threads, per_cu_bytes / (ms * 1e-3 * 2.4e9));
    }
  }
  {
    int* d_map;
    CHECK(hipMalloc(&d_map, 64 * 64 * sizeof(int)));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, d_map);
    CHECK(hipDeviceSynchronize());
    std::vector<int> map(64 * 64);
    CHECK(hipMemcpy(map.data(), d_map, map.size() * sizeof(int), hipMemcpyDeviceToHost));
    printf(" \"mfma_f64_4x4x4_layout_rows_la_cols_lb_value_ld\": [\n");
    for (int la = 0; la < 64; la++) {
      printf("  [");
      for (int lb = 0; lb < 64; lb++) printf("%d%s", map[la * 64 + lb], lb == 63 ? "" : ",");
      printf("]%s\n", la == 63 ? "" : ",");
    }
    printf(" ]\n");
  }
  printf("}\n");
  return 0;
}
