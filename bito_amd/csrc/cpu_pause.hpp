// cpu_pause.hpp -- one spin of a host-side polling loop: the x86 pause hint, elsewhere a yield.  Shared by the helper
// threads (host_pool.hpp, which a plain g++ test includes on its own) and the result polling of worker.cpp.
#pragma once
#include <thread>

namespace bito_amd {
inline void CpuPause() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}
}  // namespace bito_amd
