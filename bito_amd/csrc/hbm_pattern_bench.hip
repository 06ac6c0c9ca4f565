// hbm_pattern_bench.hip -- what HBM delivers for the HBM-arena walk's access pattern (measurement tool, not product):
// independent waves read and write CHUNKS at pseudo-random places of a large buffer, a chunk = `chunk_bytes` contiguous
// bytes moved by a wave in instructions of 64 lanes x 8 or 16 bytes, non-temporal.  walk_hbm_cat_kernel moves 2 KB
// chunks (four 512-byte instructions), the four waves of a workgroup adjacent ones (`group` = 4).
// usage: hbm_pattern_bench.bin [buffer GB = 16] [only the runs with this window: a workgroup's chunks inside its own MB (0: anywhere)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef unsigned UInt2 __attribute__((ext_vector_type(2)));
typedef unsigned UInt4 __attribute__((ext_vector_type(4)));

template <int LANE_BYTES, bool NT>
__global__ void __launch_bounds__(256) pattern_kernel(const char* __restrict__ src, char* __restrict__ dst, unsigned long long chunks,
                                                      int chunk_bytes, int iters, int reads, int writes, int group, double* sink, unsigned long long window) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long state = (unsigned long long)(blockIdx.x * (group == 4 ? 1 : 4) + (group == 4 ? 0 : wave)) * 0x9E3779B97F4A7C15ull + 12345;
  double acc = 0.0;
  const int per = chunk_bytes / (64 * LANE_BYTES);
  for (int it = 0; it < iters; it++) {
    for (int r = 0; r < reads; r++) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      unsigned long long chunk = window ? (blockIdx.x * window + (state >> 24) % window) % chunks : (state >> 24) % chunks;
      if (group == 4) chunk = (chunk & ~3ull) + wave;
      chunk = __builtin_amdgcn_readfirstlane((unsigned)chunk) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(chunk >> 32)) << 32);
      const char* p = src + chunk * (unsigned long long)chunk_bytes + lane * LANE_BYTES;
      for (int j = 0; j < per; j++) {
        if (LANE_BYTES == 8) {
          const double v = NT ? __builtin_nontemporal_load(reinterpret_cast<const double*>(p + (size_t)j * 64 * LANE_BYTES))
                              : *reinterpret_cast<const double*>(p + (size_t)j * 64 * LANE_BYTES);
          acc += v;
        } else {
          typedef double D2 __attribute__((ext_vector_type(2)));
          const D2 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const D2*>(p + (size_t)j * 64 * LANE_BYTES))
                          : *reinterpret_cast<const D2*>(p + (size_t)j * 64 * LANE_BYTES);
          acc += v.x + v.y;
        }
      }
    }
    for (int w = 0; w < writes; w++) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      unsigned long long chunk = window ? (blockIdx.x * window + (state >> 24) % window) % chunks : (state >> 24) % chunks;
      if (group == 4) chunk = (chunk & ~3ull) + wave;
      chunk = __builtin_amdgcn_readfirstlane((unsigned)chunk) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(chunk >> 32)) << 32);
      char* p = dst + chunk * (unsigned long long)chunk_bytes + lane * LANE_BYTES;
      for (int j = 0; j < per; j++) {
        if (LANE_BYTES == 8) {
          if (NT) __builtin_nontemporal_store(acc + j, reinterpret_cast<double*>(p + (size_t)j * 64 * LANE_BYTES));
          else *reinterpret_cast<double*>(p + (size_t)j * 64 * LANE_BYTES) = acc + j;
        } else {
          typedef double D2 __attribute__((ext_vector_type(2)));
          const D2 v{acc + j, acc - j};
          if (NT) __builtin_nontemporal_store(v, reinterpret_cast<D2*>(p + (size_t)j * 64 * LANE_BYTES));
          else *reinterpret_cast<D2*>(p + (size_t)j * 64 * LANE_BYTES) = v;
        }
      }
    }
  }
  if (acc == 123.456) sink[0] = acc;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const size_t gb = argc > 1 ? atoi(argv[1]) : 16;
  const size_t bytes = gb << 30;
  char *src, *dst;
  double* sink;
  CHECK(hipMalloc(&src, bytes));
  CHECK(hipMalloc(&dst, bytes));
  CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(src, 0, bytes));
  CHECK(hipMemset(dst, 0, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("buffers 2 x %zu GB; workgroups of 4 waves, 8 per CU; GB/s moved (reads + writes)\n", gb);
  printf("%8s %5s %3s %6s | %9s %9s %9s\n", "chunk", "lane", "nt", "group", "read", "write", "1r+1w");
  const int only_window = argc > 2 ? atoi(argv[2]) : -1;  // window per workgroup in MB (0: the whole buffer)
  for (int window_mb : {0, 8, 2})
  for (int group : {1, 4})
    for (int nt : {1, 0})
      for (int lane_bytes : {8, 16})
        for (int chunk : {512, 1024, 2048, 4096, 8192, 32768}) {
          if (chunk < 64 * lane_bytes) continue;
          if (only_window >= 0 && window_mb != only_window) continue;
          if (window_mb && (nt == 0 || chunk == 512 || chunk == 1024 || chunk == 4096)) continue;
          const unsigned long long window_chunks = (unsigned long long)window_mb * (1 << 20) / chunk;
          double res[3];
          for (int mode = 0; mode < 3; mode++) {
            const int reads = mode != 1, writes = mode != 0;
            const unsigned long long chunks = bytes / chunk;
            const int blocks = 256 * 8;
            const int iters = (int)((size_t)(24ull << 30) / ((size_t)blocks * 4 * chunk * (reads + writes)));
            auto launch = [&](int it) {
#define GO(LB, NTF) hipLaunchKernelGGL((pattern_kernel<LB, NTF>), dim3(blocks), dim3(256), 0, 0, src, dst, chunks, chunk, it, reads, writes, group, sink, window_chunks)
              if (lane_bytes == 8) { if (nt) GO(8, true); else GO(8, false); }
              else { if (nt) GO(16, true); else GO(16, false); }
            };
            launch(iters / 8 + 1);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            launch(iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            res[mode] = (double)blocks * 4 * chunk * (reads + writes) * iters / (ms * 1e-3) / 1e9;
          }
          printf("%8d %5d %3d %6d | %9.0f %9.0f %9.0f   window %d MB\n", chunk, lane_bytes, nt, group, res[0], res[1], res[2], window_mb);
          fflush(stdout);
        }
  return 0;
}
