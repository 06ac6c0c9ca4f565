// probe_gpr_idx.hip -- does the VGPR index mode (s_set_gpr_idx_on) reach AGPR operands on gfx950?  Four checks:
// v_mov from an indexed VGPR, v_mfma with an indexed VGPR A operand, v_mfma with an indexed AGPR A operand (what
// walk_pipe.hip relies on), v_accvgpr_read from an indexed AGPR; plus v_readlane with a scalar lane select.
// Output on the MI355X box (round 2): every error 0 / 1.85e-16 for index 0, 1, 2; readlane(37) = 1037.
#include <hip/hip_runtime.h>
#include <cstdio>
// tests of VGPR index mode (s_set_gpr_idx_on) with AGPR/VGPR sources
__global__ void k(const double* __restrict__ p, double* __restrict__ out, int idx) {
  const int lane = threadIdx.x;
  double a = p[lane], b = p[64 + lane];
  double r0, r1, r2, r3;
  asm volatile(
      // a[0:1] = 1.0*a-ish images: a[2k:2k+1] = p[lane] * (k+1), k = 0..3 ; v[100+2k] likewise
      "v_mul_f64 v[100:101], %[a], 1.0\n"
      "v_mul_f64 v[102:103], %[a], 2.0\n"
      "v_mul_f64 v[104:105], %[a], 4.0\n"
      "s_nop 4\n"
      "v_accvgpr_write_b32 a0, v100\n v_accvgpr_write_b32 a1, v101\n"
      "v_accvgpr_write_b32 a2, v102\n v_accvgpr_write_b32 a3, v103\n"
      "v_accvgpr_write_b32 a4, v104\n v_accvgpr_write_b32 a5, v105\n"
      "s_nop 4\n"
      "s_lshl_b32 s40, %[idx], 1\n"
      // test 1: v_mov with SRC0 indexing from VGPRs
      "s_set_gpr_idx_on s40, gpr_idx(SRC0)\n"
      "v_mov_b32 v110, v100\n"
      "v_mov_b32 v111, v101\n"
      "s_set_gpr_idx_off\n"
      "v_mov_b64 %[r0], v[110:111]\n"
      // test 2: mfma srcA VGPR indexed
      "s_set_gpr_idx_on s40, gpr_idx(SRC0)\n"
      "v_mfma_f64_4x4x4_4b_f64 v[112:113], v[100:101], %[b], 0\n"
      "s_set_gpr_idx_off\n"
      "s_nop 8\n"
      "v_mov_b64 %[r1], v[112:113]\n"
      // test 3: mfma srcA AGPR indexed
      "s_set_gpr_idx_on s40, gpr_idx(SRC0)\n"
      "v_mfma_f64_4x4x4_4b_f64 v[114:115], a[0:1], %[b], 0\n"
      "s_set_gpr_idx_off\n"
      "s_nop 8\n"
      "v_mov_b64 %[r2], v[114:115]\n"
      // test 4: accvgpr_read indexed
      "s_set_gpr_idx_on s40, gpr_idx(SRC0)\n"
      "v_accvgpr_read_b32 v116, a0\n"
      "v_accvgpr_read_b32 v117, a1\n"
      "s_set_gpr_idx_off\n"
      "s_nop 2\n"
      "v_mov_b64 %[r3], v[116:117]\n"
      : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3)
      : [a] "v"(a), [b] "v"(b), [idx] "s"(idx)
      : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v110", "v111", "v112", "v113", "v114", "v115", "v116",
        "v117", "a0", "a1", "a2", "a3", "a4", "a5", "s40", "m0");
  out[lane] = r0; out[64 + lane] = r1; out[128 + lane] = r2; out[192 + lane] = r3;
}
// descriptor-in-lanes: v_readlane with an SGPR lane select
__global__ void k2(const unsigned* __restrict__ tab, unsigned* __restrict__ out, int step) {
  unsigned v = tab[threadIdx.x];
  unsigned r;
  asm volatile("s_nop 4\n v_readlane_b32 s40, %[v], %[st]\n s_nop 4\n v_mov_b32 %[r], s40\n" : [r] "=v"(r) : [v] "v"(v), [st] "s"(step) : "s40");
  out[threadIdx.x] = r;
}
int main() {
  double h[128], o[256];
  for (int i = 0; i < 128; i++) h[i] = 1.0 + 0.01 * i;
  double *d, *dout; hipMalloc(&d, sizeof h); hipMalloc(&dout, sizeof o);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  for (int idx = 0; idx < 3; idx++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout, idx);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    const double scale = 1 << idx;
    // expected mfma for lane: D_b[i][j] = sum_k (scale*h[16k+4b+i]) * h[64+16k+4b+j]
    double e1 = 0, e2 = 0, e0 = 0, e3 = 0;
    for (int lane = 0; lane < 64; lane++) {
      int i = lane >> 4, b = (lane >> 2) & 3, j = lane & 3; double m = 0;
      for (int k2 = 0; k2 < 4; k2++) m += scale * h[16 * k2 + 4 * b + i] * h[64 + 16 * k2 + 4 * b + j];
      e0 = fmax(e0, fabs(o[lane] - scale * h[lane])); e1 = fmax(e1, fabs(o[64 + lane] - m) / m);
      e2 = fmax(e2, fabs(o[128 + lane] - m) / m); e3 = fmax(e3, fabs(o[192 + lane] - scale * h[lane]));
    }
    printf("idx %d: v_mov err %.2e | mfma vgpr-indexed relerr %.2e | mfma agpr-indexed relerr %.2e | accvgpr_read indexed err %.2e\n", idx, e0, e1, e2, e3);
  }
  unsigned t[64], ot[64]; for (int i = 0; i < 64; i++) t[i] = 1000 + i;
  unsigned *dt, *dot; hipMalloc(&dt, sizeof t); hipMalloc(&dot, sizeof ot); hipMemcpy(dt, t, sizeof t, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, dt, dot, 37); hipMemcpy(ot, dot, sizeof ot, hipMemcpyDeviceToHost);
  printf("readlane(37) = %u\n", ot[5]);
  return 0;
}
