// selftest.hip -- the stand-in runtime's own semantics, exercised by tests/test_hip_emu.py (test infrastructure).
// Each kernel writes what it computed; the host prints one line per check: "name ok" or "name BAD ...".
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "wave_sums.hpp"

using namespace bito_amd;

__global__ void __launch_bounds__(256) block_sum_kernel(const double* in, double* out, int n) {
  extern __shared__ double partial[];  // [waves]
  double v = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v += in[i];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x >= 64) return;  // (three of the four waves leave before the next barrier)
  __syncthreads();
  if (threadIdx.x == 0) {
    double total = 0;
    for (unsigned w = 0; w < blockDim.x / 64; w++) total += partial[w];
    out[blockIdx.x] = total;
  }
}

__global__ void __launch_bounds__(64) wave_ops_kernel(const double* a, const double* b, double* out, unsigned long long* masks) {
  const int lane = threadIdx.x;
  const double s = PairSum(a[lane], b[lane]);  // lane 31: sum of a, lane 63: sum of b
  out[lane] = s;
  out[64 + lane] = SwapSum32(a[lane]);
  out[128 + lane] = SwapSum16(a[lane]);
  out[192 + lane] = __shfl(a[lane], 7);
  out[256 + lane] = __builtin_amdgcn_readfirstlane(a[lane]);
  masks[lane] = __ballot(lane % 3 == 0);
  const unsigned long long whole = (unsigned long long)__all(lane < 64) | ((unsigned long long)__any(lane > 62) << 1) | ((unsigned long long)__all(lane < 63) << 2);
  if (lane == 0) masks[64] = whole;
  // in a divergent branch a vote counts the lanes that take it (the hardware's EXEC mask): lanes 0-9 only
  if (lane < 10) {
    const unsigned long long part = (unsigned long long)__all(lane < 10) | ((unsigned long long)__any(lane > 9) << 1) | (__ballot(1) << 2);
    if (lane == 0) masks[65] = part;
  }
}

__global__ void counter_kernel(unsigned long long* counter, int* order) {
  const unsigned long long at = atomicAdd(counter, 1ull);
  order[at] = (int)(blockIdx.x * blockDim.x + threadIdx.x);
}

// The assembly interpreter (gfx950_asm.hpp) against the builtin it stands beside, and its hazard check against small
// programs that keep and break the rules: variant 0 the 4x4x4 matrix instruction, 1 an LDS read behind its s_waitcnt,
// 2 the same read used without the wait, 3 s_movrels straight behind a write of M0, 4 the same with its wait state,
// 5 lgkmcnt(1) with a scalar load in flight (which may return first: the LDS read is NOT guaranteed), 6-9 below.
__global__ void __launch_bounds__(64) asm_kernel(const double* a, const double* b, double* out, int variant) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  double x = a[lane], y = b[lane], r = -1.0;
  unsigned addr = (unsigned)lane * 8u;
  const double* table = a;
  lds[lane] = x;
  __syncthreads();
  using hip_emu::Op;
  if (variant == 0)
    hip_emu::RunAsm("v_mfma_f64_4x4x4_4b_f64 %[r], %[x], %[y], 0\n", {Op("r", "=v", r)}, {Op("x", "v", x), Op("y", "v", y)});
  if (variant == 1)
    hip_emu::RunAsm("ds_read_b64 v[2:3], %[addr]\ns_waitcnt lgkmcnt(0)\nv_add_f64 %[r], v[2:3], %[y]\n", {Op("r", "=v", r)},
                    {Op("addr", "v", addr), Op("y", "v", y)});
  if (variant == 2)
    hip_emu::RunAsm("ds_read_b64 v[2:3], %[addr]\nv_add_f64 %[r], v[2:3], %[y]\ns_waitcnt lgkmcnt(0)\n", {Op("r", "=v", r)},
                    {Op("addr", "v", addr), Op("y", "v", y)});
  if (variant == 3)
    hip_emu::RunAsm("s_mov_b32 s21, 7\ns_mov_b32 m0, 1\ns_movrels_b32 s10, s20\nv_mov_b32 %[r], s10\n", {Op("r", "=v", r)}, {});
  if (variant == 4)
    hip_emu::RunAsm("s_mov_b32 s21, 7\ns_mov_b32 m0, 1\ns_nop 0\ns_movrels_b32 s10, s20\nv_mov_b32 %[r], s10\n", {Op("r", "=v", r)}, {});
  if (variant == 5)
    hip_emu::RunAsm("s_load_dwordx2 s[30:31], %[table], 0x0\nds_read_b64 v[2:3], %[addr]\ns_waitcnt lgkmcnt(1)\n"
                    "v_add_f64 %[r], v[2:3], %[y]\ns_waitcnt lgkmcnt(0)\n",
                    {Op("r", "=v", r)}, {Op("addr", "v", addr), Op("y", "v", y), Op("table", "s", table)});
  // wait states around the matrix instruction: 6 a vector read straight behind it, 7 the same behind s_nop 5, 8 a vector
  // result as a matrix operand one state later, 9 the same two states later
  if (variant == 6)
    hip_emu::RunAsm("v_mfma_f64_4x4x4_4b_f64 v[4:5], %[x], %[y], 0\nv_add_f64 %[r], v[4:5], 0\n", {Op("r", "=v", r)}, {Op("x", "v", x), Op("y", "v", y)});
  if (variant == 7)
    hip_emu::RunAsm("v_mfma_f64_4x4x4_4b_f64 v[4:5], %[x], %[y], 0\ns_nop 5\nv_add_f64 %[r], v[4:5], 0\n", {Op("r", "=v", r)}, {Op("x", "v", x), Op("y", "v", y)});
  if (variant == 8)
    hip_emu::RunAsm("v_add_f64 v[6:7], %[x], 0\ns_nop 0\nv_mfma_f64_4x4x4_4b_f64 v[4:5], v[6:7], %[y], 0\ns_nop 5\nv_add_f64 %[r], v[4:5], 0\n",
                    {Op("r", "=v", r)}, {Op("x", "v", x), Op("y", "v", y)});
  if (variant == 9)
    hip_emu::RunAsm("v_add_f64 v[6:7], %[x], 0\ns_nop 1\nv_mfma_f64_4x4x4_4b_f64 v[4:5], v[6:7], %[y], 0\ns_nop 5\nv_add_f64 %[r], v[4:5], 0\n",
                    {Op("r", "=v", r)}, {Op("x", "v", x), Op("y", "v", y)});
  out[lane] = (variant == 0 || variant >= 6) ? r - __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, 0.0, 0, 0, 0) : r;
}

int main() {
  int bad = 0;
  {
    std::vector<double> a(64), b(64);
    for (int l = 0; l < 64; l++) {
      a[l] = 0.3 + l * 0.37;
      b[l] = 1.0 / (1 + l);
    }
    double *d_a, *d_b, *d_out;
    hipMalloc(&d_a, 64 * 8);
    hipMalloc(&d_b, 64 * 8);
    hipMalloc(&d_out, 64 * 8);
    hipMemcpy(d_a, a.data(), 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_b, b.data(), 64 * 8, hipMemcpyHostToDevice);
    bool ok = true;
    // (variant, hazards it must raise, the value every lane must hold -- NaN: lane-dependent, checked below)
    const struct { int variant; long long hazards; } cases[] = {{0, 0}, {1, 0}, {2, 1}, {3, 1}, {4, 0}, {5, 1}, {6, 1}, {7, 0}, {8, 1}, {9, 0}};
    for (const auto& c : cases) {
      const long long before = hip_emu::Hazards().count;
      hipLaunchKernelGGL(asm_kernel, dim3(1), dim3(64), 64 * sizeof(double), 0, d_a, d_b, d_out, c.variant);
      const long long raised = hip_emu::Hazards().count - before;
      bool fine = (raised > 0) == (c.hazards > 0);
      for (int l = 0; l < 64 && fine; l++) {
        if (c.variant == 0 || c.variant >= 6) fine = d_out[l] == 0.0;
        else if (c.variant == 3 || c.variant == 4) { unsigned bits; float f = 0; (void)f; std::memcpy(&bits, &d_out[l], 4); fine = bits == 7u; }
        else fine = d_out[l] == a[l] + b[l];
      }
      if (!fine) std::printf("  asm variant %d: %lld hazards raised (expected %s), out[5] %g\n", c.variant, raised, c.hazards ? "some" : "none", d_out[5]);
      ok = ok && fine;
    }
    std::printf("asm_interpreter %s\n", ok ? "ok" : "BAD");
    bad += !ok;
  }
  {
    const int n = 1000, blocks = 3;
    std::vector<double> in(n);
    double want = 0;
    for (int i = 0; i < n; i++) in[i] = std::sin(i) * 1e3;
    double *d_in, *d_out;
    hipMalloc(&d_in, n * sizeof(double));
    hipMalloc(&d_out, blocks * sizeof(double));
    hipMemcpy(d_in, in.data(), n * sizeof(double), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(block_sum_kernel, dim3(blocks), dim3(256), 4 * sizeof(double), 0, d_in, d_out, n);
    // the kernel's own order: thread t sums i = t, t + 256, ...; xor tree inside a wave; waves in order
    double lanes[256];
    for (int t = 0; t < 256; t++) {
      lanes[t] = 0;
      for (int i = t; i < n; i += 256) lanes[t] += in[i];
    }
    for (int w = 0; w < 4; w++) {
      double v[64];
      for (int l = 0; l < 64; l++) v[l] = lanes[64 * w + l];
      for (int o = 32; o > 0; o >>= 1) {
        double nx[64];
        for (int l = 0; l < 64; l++) nx[l] = v[l] + v[l ^ o];
        for (int l = 0; l < 64; l++) v[l] = nx[l];
      }
      want += v[0];
    }
    const bool ok = d_out[0] == want && d_out[1] == want && d_out[2] == want;
    std::printf("block_sum %s\n", ok ? "ok" : "BAD");
    bad += !ok;
  }
  {
    std::vector<double> a(64), b(64);
    for (int l = 0; l < 64; l++) {
      a[l] = 1.0 + l * 0.25;
      b[l] = -3.0 + l * l * 0.5;
    }
    double *d_a, *d_b, *d_out;
    unsigned long long* d_masks;
    hipMalloc(&d_a, 64 * 8);
    hipMalloc(&d_b, 64 * 8);
    hipMalloc(&d_out, 320 * 8);
    hipMalloc(&d_masks, 66 * 8);
    hipMemcpy(d_a, a.data(), 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_b, b.data(), 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(wave_ops_kernel, dim3(1), dim3(64), 0, 0, d_a, d_b, d_out, d_masks);
    double sa = 0, sb = 0;
    for (int l = 0; l < 64; l++) {
      sa += a[l];
      sb += b[l];
    }
    // (the sums here are exact in double: quarter-integers and half-integers of modest size)
    bool ok = d_out[31] == sa && d_out[63] == sb;
    for (int l = 0; l < 64 && ok; l++) {
      ok = ok && d_out[64 + l] == a[l] + a[l ^ 32] && d_out[128 + l] == a[l] + a[l ^ 16] && d_out[192 + l] == a[7] && d_out[256 + l] == a[0];
      unsigned long long want = 0;
      for (int k = 0; k < 64; k += 3) want |= 1ull << k;
      ok = ok && d_masks[l] == want;
    }
    ok = ok && d_masks[64] == 3ull;  // all(lane < 64) and any(lane > 62), not all(lane < 63)
    ok = ok && d_masks[65] == (1ull | (0x3ffull << 2));  // among lanes 0-9: all, not any, ballot = ten bits
    if (!ok) std::printf("  pair sums %g %g (want %g %g); swap32[0] %g (want %g) swap16[0] %g (want %g) shfl %g readfirst %g mask %llx flags %llu\n",
                         d_out[31], d_out[63], sa, sb, d_out[64], a[0] + a[32], d_out[128], a[0] + a[16], d_out[192], d_out[256], d_masks[0], d_masks[64]);
    if (!ok) std::printf("  divergent votes %llx\n", d_masks[65]);
    std::printf("wave_ops %s\n", ok ? "ok" : "BAD");
    bad += !ok;
  }
  {
    unsigned long long* d_counter;
    int* d_order;
    hipMalloc(&d_counter, 8);
    hipMalloc(&d_order, 5 * 96 * sizeof(int));
    hipMemset(d_counter, 0, 8);
    hipLaunchKernelGGL(counter_kernel, dim3(5), dim3(96), 0, 0, d_counter, d_order);
    std::vector<char> seen(5 * 96, 0);
    bool ok = *d_counter == 5 * 96;
    for (int i = 0; i < 5 * 96 && ok; i++) {
      ok = d_order[i] >= 0 && d_order[i] < 5 * 96 && !seen[d_order[i]];
      if (ok) seen[d_order[i]] = 1;
    }
    std::printf("counter %s\n", ok ? "ok" : "BAD");
    bad += !ok;
  }
  return bad;
}
