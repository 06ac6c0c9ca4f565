// host_pool.hpp -- the engine's helper threads for the host side of a blocking call.
//
// The reference's Engine runs one thread per FatBeagle instance (src/fat_beagle.hpp:160-181); here the device does
// the per-tree work and the host's share of a call is checking the wire-format rows and packing them into pinned
// memory (0.06 us per tree on one thread: 0.4 ms of a 4 ms call at 6400 trees) and copying results out.  Both are
// independent per tree, so a large chunk is cut into ranges, one per thread.  All HIP calls stay on the calling
// thread.
//
// The helpers sleep on a condition variable between bursts of calls.  A wake-up costs 30 to 150 us depending on the
// host, so it is kept off a call's critical path: Arm() wakes the helpers ahead of a job whose start time matters (at
// the start of a large call, while the caller stages the first chunk itself; ahead of the copy-out of the last chunk)
// and they poll for it for a few milliseconds; after a job they keep polling for 1.5 ms -- the next job of the call,
// or the next call of a loop, finds them awake.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "cpu_pause.hpp"

namespace bito_amd {

class HostPool {
 public:
  explicit HostPool(int helpers) {
    for (int i = 0; i < helpers; i++) threads_.emplace_back([this, i] { Loop(i + 1); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
      generation_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  HostPool(const HostPool&) = delete;
  HostPool& operator=(const HostPool&) = delete;

  int parts() const { return (int)threads_.size() + 1; }

  // the helpers are polling right now (a job or an Arm() less than the linger ago): a job starts at once
  bool Hot() const { return !threads_.empty() && Now() <= armed_until_.load(std::memory_order_relaxed); }

  // fn(part) for part = 0 .. parts() - 1, part 0 on the calling thread; returns when every part is done
  void Run(const std::function<void(int)>& fn) {
    if (threads_.empty()) {
      fn(0);
      return;
    }
    {
      std::lock_guard<std::mutex> lock(mu_);
      job_ = &fn;
      pending_.store((int)threads_.size(), std::memory_order_relaxed);
      generation_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    fn(0);
    while (pending_.load(std::memory_order_acquire) != 0) Pause();
    job_ = nullptr;
  }

  // wakes the helpers: they poll for the next Run for up to `linger` before sleeping again
  void Arm(std::chrono::microseconds linger = std::chrono::microseconds(8000)) {
    if (threads_.empty()) return;
    {
      std::lock_guard<std::mutex> lock(mu_);
      armed_until_.store(Now() + std::chrono::duration_cast<std::chrono::nanoseconds>(linger).count(), std::memory_order_relaxed);
      arm_generation_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
  }

 private:
  static long long Now() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  static void Pause() { CpuPause(); }

  void Loop(int part) {
    unsigned seen = 0, seen_arm = 0;
    for (;;) {
      // poll for the next job while armed (by Arm(), or by the job just done: calls come in bursts) ...
      while (generation_.load(std::memory_order_acquire) == seen &&
             Now() <= armed_until_.load(std::memory_order_relaxed))
        Pause();
      // ... then sleep until there is one, or somebody arms the pool
      if (generation_.load(std::memory_order_acquire) == seen) {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&] {
          return generation_.load(std::memory_order_acquire) != seen ||
                 arm_generation_.load(std::memory_order_acquire) != seen_arm;
        });
        seen_arm = arm_generation_.load(std::memory_order_acquire);
        if (generation_.load(std::memory_order_acquire) == seen) continue;  // (armed: back to polling)
      }
      seen = generation_.load(std::memory_order_acquire);
      if (stop_) return;
      const std::function<void(int)>* job = job_;
      if (job) (*job)(part);
      // the next job of this call, or the next call, is probably not far: stay awake for a while
      const long long until = Now() + kLingerAfterJobNs;
      if (armed_until_.load(std::memory_order_relaxed) < until) armed_until_.store(until, std::memory_order_relaxed);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }

  static constexpr long long kLingerAfterJobNs = 1500000;  // 1.5 ms

  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<unsigned> generation_{0}, arm_generation_{0};
  std::atomic<int> pending_{0};
  const std::function<void(int)>* job_ = nullptr;
  bool stop_ = false;
  std::atomic<long long> armed_until_{0};  // steady-clock nanoseconds
};

}  // namespace bito_amd
