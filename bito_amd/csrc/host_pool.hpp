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
#include <memory>
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

// SharedPool -- helper threads that SEVERAL issuing threads hand ranges to at once.
//
// An engine over several device slots drives every slot from a host thread of its own (engine.cpp; the reference runs a
// thread per FatBeagle instance, src/task_processor.hpp:43-140).  Each of those threads has chunks of a thousand trees
// and more to check and pack (0.06 us per tree: 0.33 ms for a 5376-tree chunk on one thread) while its GPU waits for
// the chunk -- HostPool above serves ONE caller at a time, so a slot thread packed its chunks alone.  Here every caller
// publishes a job of `count` independent items; the helpers and the caller itself claim items by an atomic counter until
// none is left, and the caller returns when all of its items are done.  A job lives in a shared_ptr that the helpers
// copy under the pool's mutex, so a helper that arrives late at a finished job touches nothing that is gone.
class SharedPool {
 public:
  explicit SharedPool(int helpers) {
    for (int i = 0; i < helpers; i++) threads_.emplace_back([this] { Loop(); });
  }
  ~SharedPool() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
      stopping_.store(true, std::memory_order_release);
      epoch_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  SharedPool(const SharedPool&) = delete;
  SharedPool& operator=(const SharedPool&) = delete;

  int helpers() const { return (int)threads_.size(); }

  // fn(item) for item = 0 .. count - 1, each exactly once, on the calling thread and whatever helpers are free;
  // returns when every item is done.  Safe to call from several threads at the same time.
  void Run(int count, const std::function<void(int)>& fn) {
    if (count <= 0) return;
    if (threads_.empty() || count == 1) {
      for (int i = 0; i < count; i++) fn(i);
      return;
    }
    auto job = std::make_shared<Job>();
    job->fn = &fn;
    job->count = count;
    {
      std::lock_guard<std::mutex> lock(mu_);
      jobs_.push_back(job);
      open_jobs_.fetch_add(1, std::memory_order_release);
      epoch_.fetch_add(1, std::memory_order_release);
      const long long until = Now() + kLingerNs;
      if (armed_until_.load(std::memory_order_relaxed) < until) armed_until_.store(until, std::memory_order_relaxed);
    }
    cv_.notify_all();
    Work(*job);
    {
      // (every item is claimed: the helpers need not look at this job again, whether or not the last items are done)
      std::lock_guard<std::mutex> lock(mu_);
      for (size_t k = 0; k < jobs_.size(); k++)
        if (jobs_[k] == job) {
          jobs_.erase(jobs_.begin() + (long)k);
          break;
        }
      open_jobs_.fetch_sub(1, std::memory_order_release);
    }
    while (job->done.load(std::memory_order_acquire) != count) CpuPause();
  }

  // wakes the helpers ahead of the first job of a call: they poll for work for `linger`
  void Arm(std::chrono::microseconds linger = std::chrono::microseconds(8000)) {
    if (threads_.empty()) return;
    {
      std::lock_guard<std::mutex> lock(mu_);
      const long long until = Now() + std::chrono::duration_cast<std::chrono::nanoseconds>(linger).count();
      if (armed_until_.load(std::memory_order_relaxed) < until) armed_until_.store(until, std::memory_order_relaxed);
      epoch_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
  }

 private:
  struct Job {
    const std::function<void(int)>* fn = nullptr;
    int count = 0;
    std::atomic<int> next{0}, done{0};
  };
  static long long Now() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  // claims items of the job until none is left; true when it ran at least one
  static bool Work(Job& job) {
    bool any = false;
    for (;;) {
      const int i = job.next.fetch_add(1, std::memory_order_relaxed);
      if (i >= job.count) return any;
      (*job.fn)(i);
      job.done.fetch_add(1, std::memory_order_release);
      any = true;
    }
  }
  std::shared_ptr<Job> Pick() {  // a job that still has unclaimed items
    std::lock_guard<std::mutex> lock(mu_);
    for (const auto& j : jobs_)
      if (j->next.load(std::memory_order_relaxed) < j->count) return j;
    return nullptr;
  }
  void Loop() {
    unsigned seen = epoch_.load(std::memory_order_acquire);
    unsigned spins = 0;
    for (;;) {
      // (polling helpers look at a counter, not at the list: the mutex stays free for the issuing threads)
      if (open_jobs_.load(std::memory_order_acquire) > 0) {
        if (std::shared_ptr<Job> job = Pick()) {
          Work(*job);
          spins = 0;
          continue;
        }
      }
      if (Now() <= armed_until_.load(std::memory_order_relaxed)) {  // armed: poll
        if (stopping_.load(std::memory_order_acquire)) return;
        // (every 64th look gives the CPU away if another thread wants it -- an issuing thread on a box with fewer CPUs
        // than threads; with a CPU to itself the call returns at once)
        if ((++spins & 63u) == 0) std::this_thread::yield();
        CpuPause();
        continue;
      }
      std::unique_lock<std::mutex> lock(mu_);
      if (stop_) return;
      cv_.wait(lock, [&] { return stop_ || epoch_.load(std::memory_order_acquire) != seen; });
      if (stop_) return;
      seen = epoch_.load(std::memory_order_acquire);
    }
  }
  static constexpr long long kLingerNs = 1500000;  // helpers keep polling 1.5 ms after the last job was published

  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<std::shared_ptr<Job>> jobs_;
  std::atomic<unsigned> epoch_{0};
  std::atomic<int> open_jobs_{0};  // jobs in the list (a job leaves it once all of its items are claimed)
  std::atomic<long long> armed_until_{0};
  std::atomic<bool> stopping_{false};
  bool stop_ = false;
};

}  // namespace bito_amd
