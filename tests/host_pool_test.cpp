// CPU-only exercise of bito_amd/csrc/host_pool.hpp (the helper threads of a blocking call): every part of every job
// runs exactly once, jobs never overlap, helpers that were armed, that lingered or that slept all pick the next job
// up.  Built with g++ (and -fsanitize=thread when the runtime is there) by tests/test_host.py.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "../bito_amd/csrc/host_pool.hpp"

int main() {
  using bito_amd::HostPool;
  int bad = 0;
  for (int helpers : {0, 1, 3, 7}) {
    HostPool pool(helpers);
    const int parts = pool.parts();
    if (parts != helpers + 1) bad++;
    std::vector<long> sums((size_t)parts, 0);
    std::atomic<int> inside{0};
    long expect = 0;
    for (int job = 0; job < 3000; job++) {
      if (job % 7 == 0) pool.Arm(std::chrono::microseconds(200));
      if (job % 501 == 500) std::this_thread::sleep_for(std::chrono::milliseconds(3));  // (the helpers fall asleep)
      std::vector<int> seen((size_t)parts, 0);
      pool.Run([&](int part) {
        inside.fetch_add(1);
        seen[(size_t)part]++;
        sums[(size_t)part] += job;
        inside.fetch_sub(1);
      });
      if (inside.load() != 0) bad++;
      for (int p = 0; p < parts; p++)
        if (seen[(size_t)p] != 1) bad++;
      expect += job;
    }
    for (int p = 0; p < parts; p++)
      if (sums[(size_t)p] != expect) bad++;
    if (helpers > 0) {
      pool.Arm();
      if (!pool.Hot()) bad++;
    } else if (pool.Hot()) {
      bad++;
    }
  }
  // SharedPool: several issuing threads hand jobs to the same helpers at once; every item of every job exactly once,
  // a caller returns only when all of its own items are done (helpers asleep, armed or lingering alike)
  for (int helpers : {0, 2, 5}) {
    bito_amd::SharedPool shared(helpers);
    const int clients = 4;
    std::atomic<int> wrong{0};
    std::vector<std::thread> issuers;
    for (int c = 0; c < clients; c++)
      issuers.emplace_back([&, c] {
        for (int job = 0; job < 1500; job++) {
          const int count = 1 + (job * 7 + c) % 23;
          std::vector<std::atomic<int>> hits((size_t)count);
          for (auto& h : hits) h.store(0);
          if (job % 11 == 0) shared.Arm(std::chrono::microseconds(100));
          if (job % 400 == 399) std::this_thread::sleep_for(std::chrono::milliseconds(3));  // (the helpers fall asleep)
          long sum = 0;
          std::atomic<long> total{0};
          shared.Run(count, [&](int item) {
            hits[(size_t)item].fetch_add(1);
            total.fetch_add(item + 1);
          });
          for (int i = 0; i < count; i++) {
            sum += i + 1;
            if (hits[(size_t)i].load() != 1) wrong.fetch_add(1);
          }
          if (total.load() != sum) wrong.fetch_add(1);
        }
      });
    for (auto& t : issuers) t.join();
    bad += wrong.load();
  }
  std::printf("host_pool_test: %d bad\n", bad);
  return bad ? 1 : 0;
}
