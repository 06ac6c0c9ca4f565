// tests/hip_emu/hip/hip_runtime.h -- a functional stand-in for the HIP runtime and the device-side language, for
// running this repository's OWN .hip translation units on the CPU inside the test suite.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under bito_amd/ includes or links this; the product library is built by hipcc for
// gfx950 and has no CPU path (bito_amd/_capi.py fails loudly without it).  What this buys: in a round (or on a machine)
// without GPU access, the kernels and the host code around them can still be executed -- every workgroup's threads as
// cooperative fibers on one OS thread, barriers and wave shuffles with their real semantics -- so that logic errors
// (a wrong index, a missed dependency, a race that a barrier was meant to close in program order) show up in the CPU
// suite.  What it does not show: timing, the hardware's rounding under FMA contraction, memory-model effects between
// workgroups, anything written in gfx950 assembly (walk_pipe.hip is out of its reach).
//
// Execution model: a kernel launch runs at once, on the launching thread, block after block (one launch at a time,
// process-wide mutex); a block's threads are fibers (a dozen instructions of x86-64 context switch: the callee-saved
// registers and the stack pointer) switched round-robin at __syncthreads() and at the wave operations (a wave = 64
// consecutive threads of the block, as on CDNA).  Streams and events are accepted and ignored: in-order, synchronous
// execution is one of the orders the stream semantics allow.
#pragma once

#if !defined(__x86_64__)
#error "tests/hip_emu switches fibers with x86-64 System V assembly"
#endif

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __constant__ static
#define __shared__ static
#define HIP_EMULATION 1

// ---- host API -----------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef struct hipEmuStream* hipStream_t;
typedef struct hipEmuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t bytes) {
  *p = std::malloc(bytes ? bytes : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { std::memmove(dst, src, bytes); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind k, hipStream_t = nullptr) { return hipMemcpy(dst, src, bytes, k); }
inline hipError_t hipMemset(void* dst, int value, size_t bytes) { std::memset(dst, value, bytes); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t = nullptr) { return hipMemset(dst, value, bytes); }
inline hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
  for (size_t r = 0; r < height; r++) std::memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
  return hipSuccess;
}

// ---- the device side ----------------------------------------------------------------------------------------------
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hip_emu {

constexpr int kWave = 64;
constexpr size_t kStack = 256 << 10;

// saves the callee-saved registers of the System V ABI on the current stack, parks the stack pointer in *from, takes
// the one in *to and returns on that stack (a new fiber's stack is laid out as if it had called this from Trampoline)
__attribute__((naked, noinline)) static void SwitchStacks(void** /*from: rdi*/, void** /*to: rsi*/) {
  __asm__ volatile(
      "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
      "movq %rsp, (%rdi)\n\t"
      "movq (%rsi), %rsp\n\t"
      "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
      "ret\n\t");
}

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  unsigned tx = 0, ty = 0, tz = 0;
};

// stacks are kept from launch to launch (not cleared: a fiber's frame is written before it is read)
inline std::vector<char*>& StackPool() {
  static thread_local std::vector<char*> pool;
  return pool;
}

struct Block {
  std::vector<Fiber> fibers;
  void* scheduler = nullptr;
  const std::function<void()>* body = nullptr;
  int current = -1, live = 0;
  // the workgroup barrier
  int arrived = 0;
  unsigned generation = 0;
  // per wave: exchange slots of the shuffles and a barrier of the wave's live lanes
  struct Wave {
    double slot[kWave];
    uint64_t bits[kWave];
    int arrived = 0, live = 0;
    unsigned generation = 0;
  };
  std::vector<Wave> waves;
};

inline Block*& Current() {
  static thread_local Block* b = nullptr;
  return b;
}
inline std::mutex& LaunchMutex() {
  static std::mutex m;
  return m;
}

}  // namespace hip_emu

struct hipEmuIdx { unsigned x, y, z; };
inline thread_local hipEmuIdx threadIdx, blockIdx;
inline thread_local dim3 blockDim, gridDim;

namespace hip_emu {

inline void Yield() {
  Block* b = Current();
  Fiber& f = b->fibers[(size_t)b->current];
  SwitchStacks(&f.sp, &b->scheduler);
  threadIdx = {f.tx, f.ty, f.tz};  // (another fiber ran meanwhile)
}
inline int Linear() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)); }

inline void BlockBarrier() {
  Block* b = Current();
  const unsigned mine = b->generation;
  if (++b->arrived == b->live) {
    b->arrived = 0;
    b->generation++;
    return;
  }
  while (b->generation == mine) Yield();
}
inline void WaveBarrier() {
  Block* b = Current();
  Block::Wave& w = b->waves[(size_t)Linear() / kWave];
  const unsigned mine = w.generation;
  if (++w.arrived == w.live) {
    w.arrived = 0;
    w.generation++;
    return;
  }
  while (w.generation == mine) Yield();
}
// a thread that returns leaves its block's and its wave's barriers (as a wave that has ended does on the hardware)
inline void Retire() {
  Block* b = Current();
  b->live--;
  if (b->live > 0 && b->arrived == b->live) {
    b->arrived = 0;
    b->generation++;
  }
  Block::Wave& w = b->waves[(size_t)Linear() / kWave];
  w.live--;
  if (w.live > 0 && w.arrived == w.live) {
    w.arrived = 0;
    w.generation++;
  }
}

inline void Trampoline() {
  Block* b = Current();
  Fiber& f = b->fibers[(size_t)b->current];
  threadIdx = {f.tx, f.ty, f.tz};
  (*b->body)();
  Retire();
  f.done = true;
  SwitchStacks(&f.sp, &b->scheduler);
  std::abort();  // (a finished fiber is never resumed)
}

inline void RunBlock(const std::function<void()>& body, dim3 block) {
  const int count = (int)(block.x * block.y * block.z);
  Block b;
  b.body = &body;
  b.fibers.resize((size_t)count);
  b.live = count;
  b.waves.resize((size_t)(count + kWave - 1) / kWave);
  for (int t = 0; t < count; t++) b.waves[(size_t)t / kWave].live++;
  Current() = &b;
  std::vector<char*>& pool = StackPool();
  while ((int)pool.size() < count) pool.push_back(static_cast<char*>(std::malloc(kStack)));
  for (int t = 0; t < count; t++) {
    Fiber& f = b.fibers[(size_t)t];
    f.tx = (unsigned)t % block.x;
    f.ty = ((unsigned)t / block.x) % block.y;
    f.tz = (unsigned)t / (block.x * block.y);
    f.stack = pool[(size_t)t];
    // top of the stack, 16-byte aligned: [top - 8] a null return address for Trampoline, [top - 16] Trampoline itself
    // (what SwitchStacks returns to), below it the six registers it pops
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~(uintptr_t)15;
    void** slot = reinterpret_cast<void**>(top);
    slot[-1] = nullptr;
    slot[-2] = reinterpret_cast<void*>(&Trampoline);
    for (int r = 3; r <= 8; r++) slot[-r] = nullptr;
    f.sp = &slot[-8];
  }
  int remaining = count;
  long idle_rounds = 0;
  while (remaining > 0) {
    const int before = remaining;
    const unsigned gen_before = b.generation;
    for (int t = 0; t < count; t++) {
      Fiber& f = b.fibers[(size_t)t];
      if (f.done) continue;
      b.current = t;
      SwitchStacks(&b.scheduler, &f.sp);
      if (f.done) remaining--;
    }
    // (a block whose threads wait at barriers that can never fill: a divergent barrier in the kernel)
    idle_rounds = (remaining == before && b.generation == gen_before) ? idle_rounds + 1 : 0;
    if (idle_rounds > 1000000) {
      std::fprintf(stderr, "hip_emu: a workgroup makes no progress (divergent barrier?)\n");
      std::abort();
    }
  }
  Current() = nullptr;
}

template <typename F>
inline void Launch(F&& body_of_thread, dim3 grid, dim3 block) {
  std::lock_guard<std::mutex> lock(LaunchMutex());
  const std::function<void()> body = body_of_thread;
  gridDim = grid;
  blockDim = block;
  for (unsigned z = 0; z < grid.z; z++)
    for (unsigned y = 0; y < grid.y; y++)
      for (unsigned x = 0; x < grid.x; x++) {
        blockIdx = {x, y, z};
        RunBlock(body, block);
      }
}

}  // namespace hip_emu

#define hipLaunchKernelGGL(kernel, grid, block, shared_bytes, stream, ...) \
  hip_emu::Launch([=]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block))

inline void __syncthreads() { hip_emu::BlockBarrier(); }
inline void __threadfence() {}
inline void __threadfence_block() {}
inline void __threadfence_system() {}

template <typename T>
inline T __shfl_xor(T v, int mask) {
  static_assert(sizeof(T) <= 8, "shuffle of a wider type");
  hip_emu::Block* b = hip_emu::Current();
  const int lin = hip_emu::Linear(), lane = lin % hip_emu::kWave;
  hip_emu::Block::Wave& w = b->waves[(size_t)lin / hip_emu::kWave];
  uint64_t raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  w.bits[lane] = raw;
  hip_emu::WaveBarrier();
  const uint64_t got = w.bits[(lane ^ mask) & (hip_emu::kWave - 1)];
  hip_emu::WaveBarrier();
  T out;
  std::memcpy(&out, &got, sizeof(T));
  return out;
}
template <typename T>
inline T __shfl(T v, int src) {
  hip_emu::Block* b = hip_emu::Current();
  const int lin = hip_emu::Linear(), lane = lin % hip_emu::kWave;
  hip_emu::Block::Wave& w = b->waves[(size_t)lin / hip_emu::kWave];
  uint64_t raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  w.bits[lane] = raw;
  hip_emu::WaveBarrier();
  const uint64_t got = w.bits[src & (hip_emu::kWave - 1)];
  hip_emu::WaveBarrier();
  T out;
  std::memcpy(&out, &got, sizeof(T));
  return out;
}
template <typename T>
inline T __builtin_amdgcn_readfirstlane_emu(T v) { return __shfl(v, 0); }

template <typename T>
inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline double atomicAdd(double* p, double v) { const double old = *p; *p = old + v; return old; }
template <typename T>
inline T atomicMin(T* p, T v) { const T old = *p; if (v < old) *p = v; return old; }
template <typename T>
inline T atomicMax(T* p, T v) { const T old = *p; if (v > old) *p = v; return old; }

// math of the device library that <cmath> spells the same way is used as is; min / max come as overloads in HIP
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline long long min(long long a, long long b) { return a < b ? a : b; }
inline long long max(long long a, long long b) { return a > b ? a : b; }
