// tests/hip_emu/hip/hip_runtime.h -- a functional stand-in for the HIP runtime and the device-side language, for
// running this repository's OWN .hip translation units on the CPU inside the test suite.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under bito_amd/ includes or links this; the product library is built by hipcc for
// gfx950 and has no CPU path (bito_amd/_capi.py fails loudly without it).  What this buys: in a round (or on a machine)
// without GPU access, the kernels and the host code around them can still be executed -- every workgroup's threads as
// cooperative fibers on one OS thread, barriers and wave shuffles with their real semantics -- so that logic errors
// (a wrong index, a missed dependency, a race that a barrier was meant to close in program order) show up in the CPU
// suite.  What it does not show: timing, the hardware's rounding under FMA contraction, memory-model effects between
// workgroups.  gfx950 assembly: walk_pipe.hip's three asm statements are INTERPRETED (gfx950_asm.hpp, RunAsm below);
// walk_lds.hip's asm statements (scalar descriptor loads and the waits for them) become copies: every kernel of the product
// is in the emulated build.  Lanes of a wave do
// not run in lockstep here: code that relies on that without a wave operation in between (one place, pipe_prepare's
// exponentials handed round through LDS) gets a wave barrier in the prepared copy (prepare.py).
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

#include <time.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
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
  // (device memory starts out as garbage on the hardware too; HIP_EMU_POISON=1 makes a read of it show: all-ones words
  // are NaNs as doubles and -1 as indices)
  static const bool poison = std::getenv("HIP_EMU_POISON") != nullptr;
  if (*p && poison) std::memset(*p, 0xff, bytes ? bytes : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <typename T>
inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(p), bytes); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { std::memmove(dst, src, bytes); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind k, hipStream_t = nullptr) { return hipMemcpy(dst, src, bytes, k); }
inline hipError_t hipMemset(void* dst, int value, size_t bytes) { std::memset(dst, value, bytes); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t = nullptr) { return hipMemset(dst, value, bytes); }
inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned = 0) {
  *p = nullptr;
  return posix_memalign(p, 4096, bytes ? bytes : 1) == 0 ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes) {
  *free_bytes = (size_t)6 << 30;  // (what an arena may plan with on the CPU)
  *total_bytes = (size_t)8 << 30;
  return hipSuccess;
}
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t, int) { *value = 2; return hipSuccess; }  // (two "CUs")
struct hipDeviceProp_t {
  char gcnArchName[64];
  int multiProcessorCount;
  size_t totalGlobalMem;
};
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "cpu-emulation");
  p->multiProcessorCount = 1;
  p->totalGlobalMem = (size_t)8 << 30;
  return hipSuccess;
}
// streams and events: handles without behaviour (every command has completed when its call returns); an event keeps
// the time of its record for hipEventElapsedTime
struct hipEmuStream { int unused; };
struct hipEmuEvent { double ms; };
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = new hipEmuStream{0}; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEmuEvent{0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  e->ms = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  return hipSuccess;
}
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->ms - a->ms); return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <typename F>
inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }

inline hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
  for (size_t r = 0; r < height; r++) std::memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
  return hipSuccess;
}

// ---- the device side ----------------------------------------------------------------------------------------------
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ThreadSanitizer build (make race: -fsanitize=thread): every GPU thread is a TSan fiber, __syncthreads() and the wave
// operations are its only happens-before edges inside a workgroup, so an LDS or global access that two threads of a
// workgroup make without a barrier between them -- which the serial fibers above would never expose -- is REPORTED as a
// data race.  (Within a wave the hardware's lockstep orders what TSan cannot know about: such reports name two lanes of
// one wave and are read as "relies on lockstep".)  Workgroups and launches are ordered through the launching thread.
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define HIP_EMU_TSAN 1
#endif
#endif
#ifndef HIP_EMU_TSAN
#define HIP_EMU_TSAN 0
#endif
#if HIP_EMU_TSAN
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void* fiber);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
void __tsan_acquire(void* addr);
void __tsan_release(void* addr);
}
#define HIP_EMU_NO_TSAN __attribute__((no_sanitize("thread")))
#else
#define HIP_EMU_NO_TSAN
#endif

namespace hip_emu {

inline void TsanSwitch(void* fiber) {
#if HIP_EMU_TSAN
  __tsan_switch_to_fiber(fiber, 1u);  // (1: no happens-before edge between the fibers)
#else
  (void)fiber;
#endif
}
inline void TsanRelease(void* token) {
#if HIP_EMU_TSAN
  __tsan_release(token);
#else
  (void)token;
#endif
}
inline void TsanAcquire(void* token) {
#if HIP_EMU_TSAN
  __tsan_acquire(token);
#else
  (void)token;
#endif
}

constexpr int kWave = 64;
#if HIP_EMU_TSAN
constexpr size_t kStack = 1024 << 10;  // (instrumented frames are larger)
#else
constexpr size_t kStack = 256 << 10;
#endif

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
  void* tsan = nullptr;  // (ThreadSanitizer build: this thread's fiber)
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
  void* scheduler_tsan = nullptr;
  char start_token = 0, done_token = 0, barrier_token = 0;  // (ThreadSanitizer build: what the happens-before edges hang on)
  const std::function<void()>* body = nullptr;
  int current = -1, live = 0;
  // the workgroup barrier
  int arrived = 0;
  unsigned generation = 0;
  // s_barrier inside an interpreted asm statement: one participant per wave
  int asm_arrived = 0;
  unsigned asm_generation = 0;
  // per wave: exchange slots of the shuffles and a barrier of the wave's live lanes
  struct Wave {
    uint64_t bits[kWave];
    char token = 0;                               // (ThreadSanitizer build: the wave operations' happens-before edges)
    void* machine = nullptr;                      // gfx950_asm.hpp: the wave's registers, kept from one asm statement to the next
    std::vector<std::vector<uint64_t>>* staged = nullptr;  // ... and the operands of the statement being run
    void (*release)(Wave&) = nullptr;                       // frees the two at the end of the workgroup
    unsigned stamp[kWave];  // the exchange a lane last took part in: a wave operation sees the lanes that called it
    unsigned exchange = 0;  // (as the hardware's sees the lanes of the EXEC mask), not a lane's stale value
    int arrived = 0, live = 0, lanes = 0;  // lanes: threads the wave was launched with
    unsigned generation = 0;
  };
  std::vector<Wave> waves;
};

// the launch's dynamic LDS (`extern __shared__ T name[];` becomes `T* name = (T*)hip_emu::DynamicShared();` in the copies
// of the sources the emulated build compiles, tests/hip_emu/prepare.py)
inline std::vector<double>& DynamicStore() {
  static thread_local std::vector<double> store;
  return store;
}
inline void* DynamicShared() { return DynamicStore().data(); }

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

HIP_EMU_NO_TSAN inline void Yield() {
  Block* b = Current();
  Fiber& f = b->fibers[(size_t)b->current];
  TsanSwitch(b->scheduler_tsan);
  SwitchStacks(&f.sp, &b->scheduler);
  threadIdx = {f.tx, f.ty, f.tz};  // (another fiber ran meanwhile)
}
inline int Linear() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)); }

HIP_EMU_NO_TSAN inline void BlockBarrier() {
  Block* b = Current();
  const unsigned mine = b->generation;
  TsanRelease(&b->barrier_token);  // (everything this thread did so far happens before what any thread does behind the barrier)
  if (++b->arrived == b->live) {
    b->arrived = 0;
    b->generation++;
    TsanAcquire(&b->barrier_token);
    return;
  }
  while (b->generation == mine) Yield();
  TsanAcquire(&b->barrier_token);
}
HIP_EMU_NO_TSAN inline void WaveBarrier() {
  Block* b = Current();
  Block::Wave& w = b->waves[(size_t)Linear() / kWave];
  const unsigned mine = w.generation;
  TsanRelease(&w.token);
  if (++w.arrived == w.live) {
    w.arrived = 0;
    w.generation++;
    TsanAcquire(&w.token);
    return;
  }
  while (w.generation == mine) Yield();
  TsanAcquire(&w.token);
}
// a thread that returns leaves its block's and its wave's barriers (as a wave that has ended does on the hardware)
HIP_EMU_NO_TSAN inline void Retire() {
  Block* b = Current();
  TsanRelease(&b->barrier_token);
  TsanRelease(&b->waves[(size_t)Linear() / kWave].token);
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

HIP_EMU_NO_TSAN inline void Trampoline() {
  Block* b = Current();
  TsanAcquire(&b->start_token);  // (first: the workgroup's record itself was written by the launching thread)
  Fiber& f = b->fibers[(size_t)b->current];
  threadIdx = {f.tx, f.ty, f.tz};
  (*b->body)();
  Retire();
  f.done = true;
  TsanRelease(&b->done_token);  // (the launching thread acquires it behind the workgroup: last thing this fiber does)
  TsanSwitch(b->scheduler_tsan);
  SwitchStacks(&f.sp, &b->scheduler);
  std::abort();  // (a finished fiber is never resumed)
}

HIP_EMU_NO_TSAN inline void RunBlock(const std::function<void()>& body, dim3 block) {
  const int count = (int)(block.x * block.y * block.z);
  Block b;
  b.body = &body;
  b.fibers.resize((size_t)count);
  b.live = count;
  b.waves.resize((size_t)(count + kWave - 1) / kWave);
  for (int t = 0; t < count; t++) {
    b.waves[(size_t)t / kWave].live++;
    b.waves[(size_t)t / kWave].lanes++;
  }
  Current() = &b;
#if HIP_EMU_TSAN
  // (fibers are kept from workgroup to workgroup, like the stacks: this runtime gives out about a thousand per process)
  static thread_local std::vector<void*> tsan_fibers;  // (a fiber stays with the OS thread that made it)
  b.scheduler_tsan = __tsan_get_current_fiber();
  while ((int)tsan_fibers.size() < count) tsan_fibers.push_back(__tsan_create_fiber(0));
  for (int t = 0; t < count; t++) b.fibers[(size_t)t].tsan = tsan_fibers[(size_t)t];
#endif
  TsanRelease(&b.start_token);  // (what the launching thread did -- and the workgroups before this one -- happens before this workgroup)
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
      TsanSwitch(f.tsan);
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
  for (Block::Wave& w : b.waves)
    if (w.release) w.release(w);
  TsanAcquire(&b.done_token);
  Current() = nullptr;
}

void NoteLaunch(const char* name);  // (the matrix builtins' counts are kept per kernel: below, beside the builtins)
template <typename F>
inline void Launch(const char* name, F&& body_of_thread, dim3 grid, dim3 block, size_t shared_bytes) {
  std::lock_guard<std::mutex> lock(LaunchMutex());
  NoteLaunch(name);
  static const bool trace = std::getenv("HIP_EMU_TRACE_LAUNCH") != nullptr;  // (one line per launch: which kernel, what shape)
  if (trace) std::fprintf(stderr, "hip_emu: launch %s grid (%u, %u, %u) block (%u, %u, %u) lds %zu\n", name, grid.x, grid.y, grid.z, block.x, block.y, block.z, shared_bytes);
  const std::function<void()> body = body_of_thread;
  DynamicStore().assign(shared_bytes / sizeof(double) + 2, 0.0);
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
  hip_emu::Launch(#kernel, [=]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(shared_bytes))

inline void __syncthreads() { hip_emu::BlockBarrier(); }
inline void __threadfence() {}
inline void __threadfence_block() {}
inline void __threadfence_system() {}

template <typename T>
inline T __shfl(T v, int src);
template <typename T>
inline T __shfl_xor(T v, int mask);
// every lane of the wave publishes a 64-bit value; returns the wave's array (valid until the lane's next wave operation)
namespace hip_emu {
HIP_EMU_NO_TSAN inline const uint64_t* Publish(uint64_t mine) {
  Block* b = Current();
  const int lin = Linear(), lane = lin % kWave;
  Block::Wave& w = b->waves[(size_t)lin / kWave];
  WaveBarrier();  // (the readers of the exchange before are done)
  w.bits[lane] = mine;
  w.stamp[lane] = w.generation;  // (the same for every lane of this exchange: the barrier above has just released them)
  WaveBarrier();
  w.exchange = w.stamp[lane];
  return w.bits;
}
// did lane l take part in the exchange the caller has just returned from?
HIP_EMU_NO_TSAN inline bool Active(int l) {
  Block* b = Current();
  const Block::Wave& w = b->waves[(size_t)Linear() / kWave];
  return l < w.lanes && w.stamp[l] == w.stamp[Linear() % kWave];
}
inline int Lane() { return Linear() % kWave; }
template <typename T>
inline T ReadFirstLane(T v) {
  uint64_t raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  const uint64_t got = Publish(raw)[0];
  T out;
  std::memcpy(&out, &got, sizeof(T));
  return out;
}
}  // namespace hip_emu
template <typename T>
inline T __shfl(T v, int src) {
  static_assert(sizeof(T) <= 8, "shuffle of a wider type");
  uint64_t raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  const uint64_t got = hip_emu::Publish(raw)[src & (hip_emu::kWave - 1)];
  T out;
  std::memcpy(&out, &got, sizeof(T));
  return out;
}
template <typename T>
inline T __shfl_xor(T v, int mask) { return __shfl(v, hip_emu::Lane() ^ mask); }
#define __builtin_amdgcn_readfirstlane(x) hip_emu::ReadFirstLane(x)
#define __builtin_amdgcn_wave_barrier() hip_emu::WaveBarrier()  // (the lanes of a wave meet: one instruction stream on the device)
// v_readlane_b32: the value of lane `lane` (wave-uniform index); ds_bpermute_b32: of lane (byte index / 4) & 63, per lane
inline int __builtin_amdgcn_readlane(int v, int lane) { return (int)(uint32_t)hip_emu::Publish((uint64_t)(uint32_t)v)[lane & 63]; }
inline int __builtin_amdgcn_ds_bpermute(int byte_index, int v) {
  return (int)(uint32_t)hip_emu::Publish((uint64_t)(uint32_t)v)[(byte_index >> 2) & 63];
}
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_ldexp(x, e) std::ldexp((double)(x), (int)(e))
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
inline int hip_emu_frexp_exp(double x) {  // v_frexp_exp_i32_f64: 0 for zero, infinity and nan
  if (x == 0.0 || !std::isfinite(x)) return 0;
  int e = 0;
  std::frexp(x, &e);
  return e;
}
#define __builtin_amdgcn_frexp_exp(x) hip_emu_frexp_exp(x)

inline uint64_t __ballot(int predicate) {
  const uint64_t* all = hip_emu::Publish(predicate ? 1 : 0);
  uint64_t mask = 0;
  for (int l = 0; l < hip_emu::kWave; l++)
    if (hip_emu::Active(l) && all[l]) mask |= 1ull << l;
  return mask;
}
inline int __all(int predicate) {  // over the lanes that call it
  const uint64_t* all = hip_emu::Publish(predicate ? 1 : 0);
  for (int l = 0; l < hip_emu::kWave; l++)
    if (hip_emu::Active(l) && !all[l]) return 0;
  return 1;
}
inline int __any(int predicate) { return __ballot(predicate) != 0; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

inline int __double2loint(double x) { uint64_t r; std::memcpy(&r, &x, 8); return (int)(uint32_t)r; }
inline int __double2hiint(double x) { uint64_t r; std::memcpy(&r, &x, 8); return (int)(uint32_t)(r >> 32); }
inline double __hiloint2double(int hi, int lo) {
  const uint64_t r = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
  double x;
  std::memcpy(&x, &r, 8);
  return x;
}

// v_permlane32_swap / v_permlane16_swap (wave_sums.hpp, walk_lds.hip, walk_tree.hip): {new vdst, new vsrc} -- a pair
// of UNSIGNED dwords, as the device builtin's result (callers OR them into 64-bit patterns)
struct hip_emu_pair { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
inline hip_emu_pair __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned vsrc, bool, bool) {
  const uint64_t* all = hip_emu::Publish(((uint64_t)vdst << 32) | vsrc);
  const int lane = hip_emu::Lane();
  hip_emu_pair r;
  r.v[0] = lane < 32 ? vdst : (uint32_t)all[lane - 32];           // upper half of vdst <- lower half of vsrc
  r.v[1] = lane < 32 ? (uint32_t)(all[lane + 32] >> 32) : vsrc;   // lower half of vsrc <- upper half of vdst
  return r;
}
inline hip_emu_pair __builtin_amdgcn_permlane16_swap(unsigned vdst, unsigned vsrc, bool, bool) {
  const uint64_t* all = hip_emu::Publish(((uint64_t)vdst << 32) | vsrc);
  const int lane = hip_emu::Lane(), row = lane >> 4;
  hip_emu_pair r;
  r.v[0] = (row & 1) ? (uint32_t)all[lane - 16] : vdst;           // odd rows of vdst <- even rows of vsrc
  r.v[1] = (row & 1) ? vsrc : (uint32_t)(all[lane + 16] >> 32);   // even rows of vsrc <- odd rows of vdst
  return r;
}
// v_mov_b32 with DPP: row_shr:n (0x111-0x11f), row_bcast:15 (0x142) -- the controls wave_sums.hpp uses -- and row_ror:n
// (0x121-0x12f: walk_tree.hip); bank mask 0xf
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int, bool bound_ctrl) {
  const uint64_t* all = hip_emu::Publish((uint32_t)src);
  const int lane = hip_emu::Lane(), row = lane >> 4, in_row = lane & 15;
  if (!((row_mask >> row) & 1)) return old;
  if (ctrl >= 0x111 && ctrl <= 0x11f) {
    const int n = ctrl - 0x110;
    if (in_row >= n) return (int)(uint32_t)all[lane - n];
    return bound_ctrl ? 0 : old;
  }
  if (ctrl == 0x142) return row > 0 ? (int)(uint32_t)all[row * 16 - 1] : old;
  if (ctrl >= 0x121 && ctrl <= 0x12f) return (int)(uint32_t)all[row * 16 + ((in_row - (ctrl - 0x120)) & 15)];  // row_ror:n
  std::fprintf(stderr, "hip_emu: DPP control 0x%x is not emulated\n", ctrl);
  std::abort();
}

// raw buffer instructions: a resource is a base address and a size; reads beyond it return zero, writes are dropped
struct hip_emu_rsrc { char* base; uint32_t bytes; };
#define __amdgpu_buffer_rsrc_t hip_emu_rsrc
inline hip_emu_rsrc __builtin_amdgcn_make_buffer_rsrc(void* base, short, int num_records, int) {
  return hip_emu_rsrc{static_cast<char*>(base), (uint32_t)num_records};
}
typedef unsigned hip_emu_u2 __attribute__((ext_vector_type(2)));
typedef unsigned hip_emu_u4 __attribute__((ext_vector_type(4)));
template <typename T>
inline T hip_emu_buffer_load(hip_emu_rsrc r, unsigned voffset, unsigned soffset) {
  T out{};
  const uint64_t at = (uint64_t)voffset + soffset;
  if (at + sizeof(T) <= r.bytes) std::memcpy(&out, r.base + at, sizeof(T));
  return out;
}
template <typename T>
inline void hip_emu_buffer_store(T v, hip_emu_rsrc r, unsigned voffset, unsigned soffset) {
  const uint64_t at = (uint64_t)voffset + soffset;
  if (at + sizeof(T) <= r.bytes) std::memcpy(r.base + at, &v, sizeof(T));
}
#define __builtin_amdgcn_raw_buffer_load_b8(r, v, s, aux) hip_emu_buffer_load<unsigned char>(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b64(r, v, s, aux) hip_emu_buffer_load<hip_emu_u2>(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b128(r, v, s, aux) hip_emu_buffer_load<hip_emu_u4>(r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b64(x, r, v, s, aux) hip_emu_buffer_store<hip_emu_u2>(x, r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b128(x, r, v, s, aux) hip_emu_buffer_store<hip_emu_u4>(x, r, v, s)

// matrix instructions the compiled C++ issues through builtins, wave-level, over the process (HIP_EMU_ASM_COUNT=1 prints
// them at exit beside the interpreter's counts, gfx950_asm.hpp: what SQ_INSTS_MFMA counts on the device)
namespace hip_emu {
struct BuiltinCounts {
  long long mfma_16x16x4 = 0, mfma_4x4x4 = 0;
  std::map<std::string, long long> per_kernel;  // matrix builtins by the kernel that issued them (the name at the launch)
  long long scratch = 0;
  long long* current = &scratch;
  ~BuiltinCounts() {
    if (!std::getenv("HIP_EMU_ASM_COUNT")) return;
    std::fprintf(stderr, "{\"builtin_mfma_f64_16x16x4\": %lld, \"builtin_mfma_f64_4x4x4\": %lld}\n", mfma_16x16x4, mfma_4x4x4);
    std::string line = "{\"builtin_mfma_by_kernel\": {";
    bool first = true;
    for (const auto& [name, count] : per_kernel) {
      if (!count) continue;
      line += std::string(first ? "" : ", ") + "\"" + name + "\": " + std::to_string(count);
      first = false;
    }
    std::fprintf(stderr, "%s}}\n", line.c_str());
  }
};
inline BuiltinCounts& Builtins() {
  static BuiltinCounts c;
  return c;
}
inline void NoteLaunch(const char* name) { Builtins().current = &Builtins().per_kernel[name]; }
}  // namespace hip_emu

// v_mfma_f64_16x16x4: A lane = 16 k + i, B lane = 16 k + j; register r of lane 16 q + j holds D[4 r + q][j]; one
// instruction rounds like a sequential fma() chain over k = 0..3 (measured on the device: profiles/r1_mfma16_probe.json)
template <typename V4>
inline V4 hip_emu_mfma_f64_16x16x4(double a, double b, V4 c) {
  uint64_t ra, rb;
  std::memcpy(&ra, &a, 8);
  std::memcpy(&rb, &b, 8);
  uint64_t A[64], B[64];
  std::memcpy(A, hip_emu::Publish(ra), sizeof(A));
  std::memcpy(B, hip_emu::Publish(rb), sizeof(B));
  const int lane = hip_emu::Lane(), q = lane >> 4, j = lane & 15;
  if (lane == 0) {
    __atomic_fetch_add(&hip_emu::Builtins().mfma_16x16x4, 1, __ATOMIC_RELAXED);
    __atomic_fetch_add(hip_emu::Builtins().current, 1, __ATOMIC_RELAXED);
  }
  V4 d = c;
  for (int r = 0; r < 4; r++) {
    const int i = 4 * r + q;
    double acc = c[r];
    for (int k = 0; k < 4; k++) {
      double x, y;
      std::memcpy(&x, &A[16 * k + i], 8);
      std::memcpy(&y, &B[16 * k + j], 8);
      acc = std::fma(x, y, acc);
    }
    d[r] = acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) hip_emu_mfma_f64_16x16x4(a, b, c)
// global_load_lds: every lane moves `bytes` from its own global address to the wave's LDS base + lane * bytes
inline void hip_emu_global_load_lds(const __attribute__((address_space(1))) void* src, __attribute__((address_space(3))) void* dst, int bytes) {
  std::memcpy(reinterpret_cast<char*>((uintptr_t)dst) + (size_t)hip_emu::Lane() * bytes, reinterpret_cast<const void*>((uintptr_t)src), (size_t)bytes);
}
#define __builtin_amdgcn_global_load_lds(src, dst, bytes, offset, aux) hip_emu_global_load_lds(src, dst, bytes)

// v_mfma_f64_4x4x4_4b: A lane = 16 k + 4 b + i, B lane = 16 k + 4 b + j, D lane = 16 i + 4 b + j (walk_lds.hip)
inline double hip_emu_mfma_f64_4x4x4(double a, double b, double c) {
  uint64_t ra, rb;
  std::memcpy(&ra, &a, 8);
  std::memcpy(&rb, &b, 8);
  uint64_t A[64], B[64];
  std::memcpy(A, hip_emu::Publish(ra), sizeof(A));
  std::memcpy(B, hip_emu::Publish(rb), sizeof(B));
  const int lane = hip_emu::Lane(), i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3;
  if (lane == 0) {
    __atomic_fetch_add(&hip_emu::Builtins().mfma_4x4x4, 1, __ATOMIC_RELAXED);
    __atomic_fetch_add(hip_emu::Builtins().current, 1, __ATOMIC_RELAXED);
  }
  double acc = c;
  for (int k = 0; k < 4; k++) {
    double x, y;
    std::memcpy(&x, &A[16 * k + 4 * blk + i], 8);
    std::memcpy(&y, &B[16 * k + 4 * blk + j], 8);
    acc = std::fma(x, y, acc);
  }
  return acc;
}
#define __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, x, y, z) hip_emu_mfma_f64_4x4x4(a, b, c)
// v_mov_b32 with DPP row_ror:n (0x121-0x12f): lane i of a row takes the value of lane (i - n) mod 16 of that row
inline int __builtin_amdgcn_mov_dpp(int src, int ctrl, int, int, bool) {
  const uint64_t* all = hip_emu::Publish((uint32_t)src);
  const int lane = hip_emu::Lane(), row = lane & ~15, in_row = lane & 15;
  if (ctrl >= 0x121 && ctrl <= 0x12f) return (int)(uint32_t)all[row + ((in_row - (ctrl - 0x120)) & 15)];
  std::fprintf(stderr, "hip_emu: DPP control 0x%x is not emulated\n", ctrl);
  std::abort();
}
inline long long clock64() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 100000000ll + ts.tv_nsec / 10;
}

#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_store(ptr, value, order, scope) __atomic_store_n(ptr, value, order)

struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct double2 { double x, y; };
struct double4 { double x, y, z, w; };
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
inline double2 make_double2(double x, double y) { return double2{x, y}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }

template <typename T>
inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline double atomicAdd(double* p, double v) {
  uint64_t seen = __atomic_load_n(reinterpret_cast<uint64_t*>(p), __ATOMIC_RELAXED), want;
  double old;
  do {
    std::memcpy(&old, &seen, 8);
    const double sum = old + v;
    std::memcpy(&want, &sum, 8);
  } while (!__atomic_compare_exchange_n(reinterpret_cast<uint64_t*>(p), &seen, want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return old;
}
template <typename T>
inline T atomicMin(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <typename T>
inline T atomicMax(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}

// math of the device library that <cmath> spells the same way is used as is; min / max come as overloads in HIP
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline long long min(long long a, long long b) { return a < b ? a : b; }
inline long long max(long long a, long long b) { return a > b ? a : b; }

// ---- interpreted asm statements (gfx950_asm.hpp; prepare.py turns `asm volatile(text : outs : ins : clobbers)` of
// walk_pipe.hip into hip_emu::RunAsm(text, {outs}, {ins})) ---------------------------------------------------------------
#include "../gfx950_asm.hpp"

namespace hip_emu {

inline const Program& ProgramOf(const char* text) {
  static std::unordered_map<const char*, Program> cache;  // (launches are serialised: no lock)
  auto it = cache.find(text);
  if (it == cache.end()) it = cache.emplace(text, ParseProgram(text)).first;
  return it->second;
}

inline void RunAsm(const char* text, std::initializer_list<AsmOperand> outs, std::initializer_list<AsmOperand> ins) {
  const Program& P = ProgramOf(text);
  Block* b = Current();
  const int lin = Linear(), lane = lin % kWave;
  Block::Wave& w = b->waves[(size_t)lin / kWave];
  const size_t count = P.operand_names.size();
  std::vector<const AsmOperand*> bound(count, nullptr);
  for (const auto* list : {&outs, &ins})
    for (const AsmOperand& o : *list)
      for (size_t k = 0; k < count; k++)
        if (P.operand_names[k] == o.name) bound[k] = &o;
  WaveBarrier();  // (the statement before is over for every lane)
  if (lane == 0) {
    if (!w.machine) w.machine = new WaveMachine();
    w.release = [](Block::Wave& x) {
      delete static_cast<WaveMachine*>(x.machine);
      delete x.staged;
      x.machine = nullptr;
      x.staged = nullptr;
    };
    if (!w.staged) w.staged = new std::vector<std::vector<uint64_t>>();
    w.staged->assign(count, std::vector<uint64_t>(64, 0));
  }
  WaveBarrier();
  for (size_t k = 0; k < count; k++) {
    if (!bound[k]) { std::fprintf(stderr, "hip_emu: asm operand %s is not bound\n", P.operand_names[k].c_str()); std::abort(); }
    (*w.staged)[k][(size_t)lane] = bound[k]->value;
  }
  WaveBarrier();
  if (lane == 0) {
    std::vector<bool> scalar(count);
    for (size_t k = 0; k < count; k++) scalar[k] = bound[k]->scalar;
    AsmContext ctx;
    ctx.lds = static_cast<char*>(DynamicShared());
    ctx.lds_bytes = DynamicStore().size() * sizeof(double);
    ctx.barrier = [b] {
      const int waves = (int)b->waves.size();
      const unsigned mine = b->asm_generation;
      TsanRelease(&b->barrier_token);  // (s_barrier is a workgroup barrier like __syncthreads(): the same happens-before edges)
      if (++b->asm_arrived == waves) {
        b->asm_arrived = 0;
        b->asm_generation++;
      } else {
        while (b->asm_generation == mine) Yield();
      }
      TsanAcquire(&b->barrier_token);
    };
    static const bool digest = std::getenv("HIP_EMU_ASM_DIGEST") != nullptr;
    auto fnv = [](const void* data, size_t bytes, uint64_t h = 1469598103934665603ull) {
      const unsigned char* c = static_cast<const unsigned char*>(data);
      for (size_t i = 0; i < bytes; i++) h = (h ^ c[i]) * 1099511628211ull;
      return h;
    };
    WaveMachine& M = *static_cast<WaveMachine*>(w.machine);
    if (digest) {
      uint64_t ho = 0;
      for (size_t k = 0; k < count; k++) ho = fnv((*w.staged)[k].data(), scalar[k] ? 8 : 512, ho + k);
      std::fprintf(stderr, "asm in  block %u wave %d stmt %p: lds %016llx operands %016llx v %016llx a %016llx s %016llx\n", blockIdx.x,
                   lin / kWave, (const void*)text, (unsigned long long)fnv(ctx.lds, ctx.lds_bytes), (unsigned long long)ho,
                   (unsigned long long)fnv(M.v.data(), 256 * 64 * 4), (unsigned long long)fnv(M.a.data(), M.a.size() * 4),
                   (unsigned long long)fnv(M.s, 106 * 4));
    }
    Execute(P, M, *w.staged, scalar, ctx);
    if (digest) {
      uint64_t ho = 0;
      for (size_t k = 0; k < count; k++) ho = fnv((*w.staged)[k].data(), scalar[k] ? 8 : 512, ho + k);
      std::fprintf(stderr, "asm out block %u wave %d stmt %p: lds %016llx operands %016llx v %016llx a %016llx s %016llx\n", blockIdx.x,
                   lin / kWave, (const void*)text, (unsigned long long)fnv(ctx.lds, ctx.lds_bytes), (unsigned long long)ho,
                   (unsigned long long)fnv(M.v.data(), 256 * 64 * 4), (unsigned long long)fnv(M.a.data(), M.a.size() * 4),
                   (unsigned long long)fnv(M.s, 106 * 4));
    }
  }
  WaveBarrier();
  for (size_t k = 0; k < count; k++)
    if (bound[k]->output) std::memcpy(bound[k]->target, &(*w.staged)[k][(size_t)lane], (size_t)bound[k]->size);
}

}  // namespace hip_emu
