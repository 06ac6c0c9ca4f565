// issue_bench.hip -- what ONE wave per SIMD pays per instruction on gfx950 (4 waves per CU, 150 KB of LDS
// per workgroup so that no second workgroup shares the CU): the cost model behind walk_pipe.hip's loops
// (scripts/gen_walk_pipe.py).  Every kernel repeats a short asm sequence 1024 times between two s_memtime
// reads and reports cycles per repetition.  Output of the MI355X box: profiles/r2_issue_costs.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define BODY(name, text)                                                                          \
  __global__ void __launch_bounds__(256) name(long long* out, double* sink) {                     \
    extern __shared__ double lds[];                                                               \
    double a = threadIdx.x * 0.5 + 1.0, b = 1.000001, c = 0.5, d = 2.0;                           \
    unsigned la = threadIdx.x * 8;                                                                \
    lds[threadIdx.x] = a;                                                                         \
    __syncthreads();                                                                              \
    long long t0 = clock64();                                                                     \
    for (int it = 0; it < 16; it++) asm volatile(REP64(text) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(la) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "s40", "s41", "s42", "s43", "memory"); \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");                                                \
    long long t1 = clock64();                                                                     \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                    \
    sink[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;                                         \
  }
BODY(k_salu, "s_add_u32 s40, s40, 1\n")
BODY(k_salu_indep, "s_add_u32 s40, s41, 1\n")
BODY(k_valu32, "v_add_u32 v100, v101, v102\n")
BODY(k_valu32_dep, "v_add_u32 v100, v100, v102\n")
BODY(k_mul64, "v_mul_f64 v[100:101], %1, %2\n")
BODY(k_mul64_dep, "v_mul_f64 %0, %0, %1\n")
BODY(k_mov64, "v_mov_b64 v[100:101], %1\n")
BODY(k_readlane, "v_readlane_b32 s40, v101, s41\n")
BODY(k_mfma, "v_mfma_f64_4x4x4_4b_f64 v[100:101], %1, %2, 0\n")
BODY(k_mfma_dep, "v_mfma_f64_4x4x4_4b_f64 %0, %0, %1, 0\n")
BODY(k_mfma_valu, "v_mfma_f64_4x4x4_4b_f64 v[100:101], %1, %2, 0\n v_add_u32 v104, v105, v106\n")
BODY(k_mfma_3valu, "v_mfma_f64_4x4x4_4b_f64 v[100:101], %1, %2, 0\n v_add_u32 v104, v105, v106\n v_add_u32 v105, v105, v106\n v_add_u32 v107, v105, v106\n")
BODY(k_mfma_mul64, "v_mfma_f64_4x4x4_4b_f64 v[100:101], %1, %2, 0\n v_mul_f64 v[104:105], %1, %2\n")
BODY(k_mfma_2salu, "v_mfma_f64_4x4x4_4b_f64 v[100:101], %1, %2, 0\n s_add_u32 s40, s41, 1\n s_add_u32 s42, s41, 1\n")
BODY(k_mul64_salu, "v_mul_f64 v[100:101], %1, %2\n s_add_u32 s40, s41, 1\n")
BODY(k_mul64_valu32, "v_mul_f64 v[100:101], %1, %2\n v_add_u32 v104, v105, v106\n")
BODY(k_dsread, "ds_read_b64 v[100:101], %4\n")
BODY(k_dsread2, "ds_read2st64_b64 v[100:103], %4 offset1:1\n")
BODY(k_dswrite2, "ds_write2st64_b64 %4, %1, %2 offset1:1\n")
BODY(k_gpridx, "s_set_gpr_idx_on s41, gpr_idx(SRC0)\n v_mov_b32 v100, v101\n s_set_gpr_idx_off\n")
BODY(k_nop0, "s_nop 0\n")
BODY(k_nop7, "s_nop 7\n")
BODY(k_branch, "s_branch 1f\n s_nop 0\n 1:\n")
BODY(k_cbranch_nt, "s_cmp_eq_u32 s41, s41\n s_cbranch_scc0 1f\n 1:\n")
BODY(k_exec, "s_cselect_b64 exec, -1, -1\n")
#define BODY2(name, text)                                                                          \
  __global__ void __launch_bounds__(256) name(long long* out, const unsigned* tab, double* sink) {\
    extern __shared__ double lds[];                                                               \
    double a = threadIdx.x * 0.5 + 1.0, b = 1.000001;                                             \
    unsigned la = threadIdx.x * 16;                                                               \
    lds[threadIdx.x] = a;                                                                         \
    __syncthreads();                                                                              \
    const unsigned* p = tab + (blockIdx.x % 64) * 256;                                            \
    long long t0 = clock64();                                                                     \
    for (int it = 0; it < 16; it++) asm volatile(REP64(text) : "+v"(a), "+v"(b) : "s"(p), "v"(la) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "memory"); \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");                                                \
    long long t1 = clock64();                                                                     \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                    \
    sink[blockIdx.x * 256 + threadIdx.x] = a + b;                                                 \
  }
BODY2(k_sload_wait, "s_load_dwordx8 s[40:47], %2, 0x0\n s_waitcnt lgkmcnt(0)\n")
BODY2(k_sload_wait_far, "s_load_dwordx8 s[40:47], %2, 0x40\n s_waitcnt lgkmcnt(0)\n s_load_dwordx8 s[40:47], %2, 0x100\n s_waitcnt lgkmcnt(0)\n")
BODY2(k_dsread128_wait, "ds_read_b128 v[100:103], %3\n s_waitcnt lgkmcnt(0)\n")
BODY2(k_dsread128, "ds_read_b128 v[100:103], %3\n")
BODY2(k_dswrite128, "ds_write_b128 %3, v[100:103]\n")
BODY2(k_dswrite128_wait, "ds_write_b128 %3, v[100:103]\n s_waitcnt lgkmcnt(0)\n")
BODY2(k_setpc, "s_getpc_b64 s[40:41]\n s_add_u32 s40, s40, 16\n s_addc_u32 s41, s41, 0\n s_setpc_b64 s[40:41]\n")
BODY2(k_mfma_agpr_idx, "s_set_gpr_idx_on s42, gpr_idx(SRC0)\n v_mfma_f64_4x4x4_4b_f64 v[100:101], a[0:1], %1, 0\n v_mfma_f64_4x4x4_4b_f64 v[102:103], a[0:1], %1, 0\n s_set_gpr_idx_off\n")
BODY2(k_mfma_then_mul, "v_mfma_f64_4x4x4_4b_f64 v[100:101], %0, %1, 0\n s_nop 5\n v_mul_f64 v[102:103], v[100:101], %1\n")
BODY2(k_4mfma_4mul, "v_mfma_f64_4x4x4_4b_f64 v[100:101], %0, %1, 0\n v_mfma_f64_4x4x4_4b_f64 v[102:103], %0, %1, 0\n v_mfma_f64_4x4x4_4b_f64 v[100:101], %0, %1, 0\n v_mfma_f64_4x4x4_4b_f64 v[102:103], %0, %1, 0\n v_mul_f64 v[100:101], %0, %1\n v_mul_f64 v[102:103], %0, %1\n v_mul_f64 v[100:101], %0, %1\n v_mul_f64 v[102:103], %0, %1\n")
#define MF "v_mfma_f64_4x4x4_4b_f64 v[100:101], %0, %1, 0\n"
BODY2(k_mfma_dsread128, MF "ds_read_b128 v[104:107], %3\n")
BODY2(k_4mfma_4dsread128, MF MF MF MF "ds_read_b128 v[104:107], %3\n ds_read_b128 v[104:107], %3\n ds_read_b128 v[104:107], %3\n ds_read_b128 v[104:107], %3\n")
BODY2(k_mfma_dswrite128, MF "ds_write_b128 %3, v[104:107]\n")
BODY2(k_4mfma_2dswrite128, MF MF MF MF "ds_write_b128 %3, v[104:107]\n ds_write_b128 %3, v[104:107]\n")
BODY2(k_mfma_4salu, MF "s_add_u32 s40, s41, 1\n s_add_u32 s42, s41, 1\n s_add_u32 s43, s41, 1\n s_add_u32 s44, s41, 1\n")
BODY2(k_mfma_salu_dsread, MF "s_add_u32 s40, s41, 1\n ds_read_b128 v[104:107], %3\n")
BODY2(k_4mfma, MF MF MF MF)

// Two waves per SIMD (512-thread workgroups: wave w runs on SIMD w % 4): waves 0-3 repeat textA, waves 4-7
// textB; each half reports its own time.  Answers whether one wave's vector / scalar / LDS work overlaps the
// other wave's matrix instructions.
#define PAIR(name, textA, textB)                                                                  \
  __global__ void __launch_bounds__(512) name(long long* out, double* sink) {                     \
    extern __shared__ double lds[];                                                               \
    double a = threadIdx.x * 0.5 + 1.0, b = 1.000001, c = 0.5, d = 2.0;                           \
    unsigned la = threadIdx.x * 16;                                                               \
    lds[threadIdx.x] = a;                                                                         \
    __syncthreads();                                                                              \
    long long t0 = clock64();                                                                     \
    if (threadIdx.x < 256) {                                                                      \
      for (int it = 0; it < 16; it++) asm volatile(REP64(textA) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(la) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "s40", "s41", "s42", "s43", "memory"); \
    } else {                                                                                      \
      for (int it = 0; it < 16; it++) asm volatile(REP64(textB) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(la) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "s40", "s41", "s42", "s43", "memory"); \
    }                                                                                             \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");                                                \
    long long t1 = clock64();                                                                     \
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) out[threadIdx.x / 256] = t1 - t0; \
    sink[blockIdx.x * 512 + threadIdx.x] = a + b + c + d;                                         \
  }
#define MFV "v_mfma_f64_4x4x4_4b_f64 v[100:101], %1, %2, 0\n"
PAIR(p_mfma_mfma, MFV, MFV)
PAIR(p_mfma_valu32, MFV, "v_add_u32 v100, v101, v102\n")
PAIR(p_mfma_mul64, MFV, "v_mul_f64 v[100:101], %1, %2\n")
PAIR(p_mfma_salu, MFV, "s_add_u32 s40, s41, 1\n")
PAIR(p_mfma_dsread128, MFV, "ds_read_b128 v[104:107], %4\n")
PAIR(p_mfma_dswrite128, MFV, "ds_write_b128 %4, v[104:107]\n")
PAIR(p_valu32_valu32, "v_add_u32 v100, v101, v102\n", "v_add_u32 v100, v101, v102\n")
PAIR(p_mul64_mul64, "v_mul_f64 v[100:101], %1, %2\n", "v_mul_f64 v[100:101], %1, %2\n")
PAIR(p_mul64_valu32, "v_mul_f64 v[100:101], %1, %2\n", "v_add_u32 v100, v101, v102\n")
PAIR(p_branch_branch, "s_branch 1f\n s_nop 0\n 1:\n", "s_branch 1f\n s_nop 0\n 1:\n")
PAIR(p_mfma_branch, MFV, "s_branch 1f\n s_nop 0\n 1:\n")
// the mix of a loop body (per matrix instruction: 1.7 vector, 0.9 scalar, 0.4 LDS), both waves the same
#define MIX MFV "v_mul_f64 v[104:105], %1, %2\n s_add_u32 s40, s41, 1\n" MFV "v_add_u32 v106, v107, v106\n ds_read_b128 v[100:103], %4\n" MFV "v_mul_f64 v[104:105], %1, %2\n v_add_u32 v106, v107, v106\n s_add_u32 s42, s41, 1\n"
PAIR(p_mix_mix, MIX, MIX)
PAIR(p_mix_idle, MIX, "s_nop 0\n")

// round 4: why two waves per SIMD did not help walk_pipe_kernel -- which instruction kinds of two sibling waves overlap
PAIR(p_salu_salu, "s_add_u32 s40, s41, 1\n", "s_add_u32 s40, s41, 1\n")
PAIR(p_nop_nop, "s_nop 0\n", "s_nop 0\n")
PAIR(p_nop3_nop3, "s_nop 3\n", "s_nop 3\n")
PAIR(p_mfma2salu_same, MFV "s_add_u32 s40, s41, 1\n s_add_u32 s42, s41, 1\n", MFV "s_add_u32 s40, s41, 1\n s_add_u32 s42, s41, 1\n")
PAIR(p_mfma4salu_same, MFV "s_add_u32 s40, s41, 1\n s_add_u32 s42, s41, 1\n s_add_u32 s40, s41, 1\n s_add_u32 s42, s41, 1\n", MFV "s_add_u32 s40, s41, 1\n s_add_u32 s42, s41, 1\n s_add_u32 s40, s41, 1\n s_add_u32 s42, s41, 1\n")
PAIR(p_mfmavalu_same, MFV "v_add_u32 v104, v105, v106\n", MFV "v_add_u32 v104, v105, v106\n")
PAIR(p_mfmamul_same, MFV "v_mul_f64 v[104:105], %1, %2\n", MFV "v_mul_f64 v[104:105], %1, %2\n")
PAIR(p_mfma2mul_same, MFV "v_mul_f64 v[104:105], %1, %2\n v_mul_f64 v[106:107], %1, %2\n", MFV "v_mul_f64 v[104:105], %1, %2\n v_mul_f64 v[106:107], %1, %2\n")
// dependent chains: matrix result -> wait states -> multiply -> wait states -> matrix operand
#define DEP "v_mfma_f64_4x4x4_4b_f64 v[100:101], %1, %2, 0\n s_nop 5\n v_mul_f64 v[102:103], v[100:101], %2\n s_nop 1\n v_mfma_f64_4x4x4_4b_f64 v[104:105], %1, v[102:103], 0\n s_nop 5\n v_mul_f64 v[106:107], v[104:105], %2\n s_nop 1\n"
PAIR(p_dep_dep, DEP, DEP)
PAIR(p_dep_idle, DEP, "s_nop 0\n")
// two groups' worth of the same chain interleaved (what G = 2 gives a wave)
#define DEP2 "v_mfma_f64_4x4x4_4b_f64 v[100:101], %1, %2, 0\n v_mfma_f64_4x4x4_4b_f64 v[102:103], %1, %3, 0\n s_nop 4\n v_mul_f64 v[100:101], v[100:101], %2\n v_mul_f64 v[102:103], v[102:103], %2\n s_nop 0\n v_mfma_f64_4x4x4_4b_f64 v[104:105], %1, v[100:101], 0\n v_mfma_f64_4x4x4_4b_f64 v[106:107], %1, v[102:103], 0\n s_nop 4\n v_mul_f64 v[104:105], v[104:105], %2\n v_mul_f64 v[106:107], v[106:107], %2\n s_nop 0\n"
PAIR(p_dep2_dep2, DEP2, DEP2)
PAIR(p_dep2_idle, DEP2, "s_nop 0\n")
PAIR(p_setpc_setpc, "s_getpc_b64 s[40:41]\n s_add_u32 s40, s40, 16\n s_addc_u32 s41, s41, 0\n s_setpc_b64 s[40:41]\n", "s_getpc_b64 s[40:41]\n s_add_u32 s40, s40, 16\n s_addc_u32 s41, s41, 0\n s_setpc_b64 s[40:41]\n")
PAIR(p_dsread_dsread, "ds_read_b128 v[104:107], %4\n", "ds_read_b128 v[104:107], %4\n")
PAIR(p_dswrite_dswrite, "ds_write_b128 %4, v[104:107]\n", "ds_write_b128 %4, v[104:107]\n")
// index mode around the matrix instructions, as the loops use it
#define IDX "s_mov_b32 s42, 0\n s_set_gpr_idx_on s42, gpr_idx(SRC0)\n v_mfma_f64_4x4x4_4b_f64 v[100:101], v[104:105], %1, 0\n v_mfma_f64_4x4x4_4b_f64 v[102:103], v[104:105], %1, 0\n s_set_gpr_idx_off\n"
PAIR(p_idx_idx, IDX, IDX)
PAIR(p_idx_idle, IDX, "s_nop 0\n")

int main() {
  long long* out; double* sink; hipMalloc(&out, 16); hipMalloc(&sink, 256 * 512 * 8);
#define RUN(k, n) { hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 150 * 1024, 0, out, sink); long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost); \
    printf("%-16s %7.2f cycles per repetition (%d instr)\n", #k, c / 1024.0, n); }
  RUN(k_salu, 1) RUN(k_salu_indep, 1) RUN(k_valu32, 1) RUN(k_valu32_dep, 1) RUN(k_mul64, 1) RUN(k_mul64_dep, 1) RUN(k_mov64, 1)
  RUN(k_readlane, 1) RUN(k_mfma, 1) RUN(k_mfma_dep, 1) RUN(k_mfma_valu, 2) RUN(k_mfma_3valu, 4) RUN(k_mfma_mul64, 2) RUN(k_mfma_2salu, 3)
  RUN(k_mul64_salu, 2) RUN(k_mul64_valu32, 2) RUN(k_dsread, 1) RUN(k_dsread2, 1) RUN(k_dswrite2, 1) RUN(k_gpridx, 3) RUN(k_nop0, 1) RUN(k_nop7, 1)
  RUN(k_branch, 1) RUN(k_cbranch_nt, 2) RUN(k_exec, 1)
  unsigned* tab; hipMalloc(&tab, 1 << 20); hipMemset(tab, 0, 1 << 20);
#define RUN2(k, n) { hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 150 * 1024, 0, out, tab, sink); long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost); \
    printf("%-20s %7.2f cycles per repetition (%d instr)\n", #k, c / 1024.0, n); }
  RUN2(k_sload_wait, 2) RUN2(k_sload_wait_far, 4) RUN2(k_dsread128_wait, 2) RUN2(k_dsread128, 1) RUN2(k_dswrite128, 1) RUN2(k_dswrite128_wait, 2) RUN2(k_setpc, 4)
  RUN2(k_mfma_agpr_idx, 4) RUN2(k_mfma_then_mul, 3) RUN2(k_4mfma_4mul, 8)
  RUN2(k_mfma_dsread128, 2) RUN2(k_4mfma_4dsread128, 8) RUN2(k_mfma_dswrite128, 2) RUN2(k_4mfma_2dswrite128, 6) RUN2(k_mfma_4salu, 5)
  RUN2(k_mfma_salu_dsread, 3) RUN2(k_4mfma, 4)
#define RUNP(k) { hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 150 * 1024, 0, out, sink); long long c[2]; hipMemcpy(c, out, 16, hipMemcpyDeviceToHost); \
    printf("%-20s %7.2f | %7.2f cycles per repetition (two waves per SIMD: first text | second text)\n", #k, c[0] / 1024.0, c[1] / 1024.0); }
  RUNP(p_salu_salu) RUNP(p_nop_nop) RUNP(p_nop3_nop3) RUNP(p_mfma2salu_same) RUNP(p_mfma4salu_same) RUNP(p_mfmavalu_same) RUNP(p_mfmamul_same) RUNP(p_mfma2mul_same)
  RUNP(p_dep_dep) RUNP(p_dep_idle) RUNP(p_dep2_dep2) RUNP(p_dep2_idle) RUNP(p_setpc_setpc) RUNP(p_dsread_dsread) RUNP(p_dswrite_dswrite) RUNP(p_idx_idx) RUNP(p_idx_idle)
  RUNP(p_mfma_mfma) RUNP(p_mfma_valu32) RUNP(p_mfma_mul64) RUNP(p_mfma_salu) RUNP(p_mfma_dsread128) RUNP(p_mfma_dswrite128)
  RUNP(p_valu32_valu32) RUNP(p_mul64_mul64) RUNP(p_mul64_valu32) RUNP(p_branch_branch) RUNP(p_mfma_branch) RUNP(p_mix_mix) RUNP(p_mix_idle)
    return 0;
}
