// racetest.hip -- the ThreadSanitizer build of the stand-in runtime against two kernels: one that reads LDS another WAVE wrote
// without a barrier between (a data race on the device; the serial fibers run it in ONE of the possible orders and say nothing),
// and the same kernel with its __syncthreads().  tests/test_hip_emu.py expects a report for the first and none for the
// second.  Test infrastructure (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

template <bool BARRIER>
__global__ void __launch_bounds__(128) exchange_kernel(const double* in, double* out) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  lds[tid] = in[tid] * 2.0;
  if (BARRIER) __syncthreads();
  out[tid] = lds[(tid + 64) % 128];  // (the other wave's value)
}

int main(int argc, char** argv) {
  const bool with_barrier = argc > 1 && std::atoi(argv[1]) != 0;
  double *in, *out;
  hipMalloc(&in, 128 * 8);
  hipMalloc(&out, 128 * 8);
  for (int i = 0; i < 128; i++) in[i] = i;
  if (with_barrier) hipLaunchKernelGGL(exchange_kernel<true>, dim3(1), dim3(128), 128 * 8, 0, in, out);
  else hipLaunchKernelGGL(exchange_kernel<false>, dim3(1), dim3(128), 128 * 8, 0, in, out);
  std::printf("out[0] %g out[64] %g\n", out[0], out[64]);
  return 0;
}
