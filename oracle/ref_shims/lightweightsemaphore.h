// Test-infrastructure shim (NOT product code): stand-in for
// moodycamel::LightweightSemaphore (concurrentqueue 1.0.4, un-vendored in the
// reference).  Interface taken from the reference's call sites
// (envpool/core/action_buffer_queue.h:47-80, state_buffer.h:51,129,141,
// circular_buffer.h:38-75): wait() -> bool, tryWait() -> bool, signal(n).
// Spin-then-block, like upstream, so CPU-baseline timings are not penalised
// by a pure mutex/condvar implementation.
#ifndef ORACLE_SHIM_LIGHTWEIGHTSEMAPHORE_H_
#define ORACLE_SHIM_LIGHTWEIGHTSEMAPHORE_H_
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <mutex>
#include <thread>

namespace moodycamel {
class LightweightSemaphore {
 public:
  explicit LightweightSemaphore(std::ptrdiff_t initial = 0) : count_(initial) {}

  bool tryWait() {
    std::ptrdiff_t old = count_.load(std::memory_order_relaxed);
    while (old > 0) {
      if (count_.compare_exchange_weak(old, old - 1, std::memory_order_acquire,
                                       std::memory_order_relaxed)) {
        return true;
      }
    }
    return false;
  }

  bool wait() {
    for (int spin = 0; spin < 2000; ++spin) {
      if (tryWait()) return true;
      if ((spin & 63) == 63) std::this_thread::yield();
    }
    std::unique_lock<std::mutex> lk(mu_);
    ++sleepers_;
    cv_.wait(lk, [this] { return tryWait(); });
    --sleepers_;
    return true;
  }

  void signal(std::ptrdiff_t n = 1) {
    count_.fetch_add(n, std::memory_order_release);
    std::lock_guard<std::mutex> lk(mu_);
    if (sleepers_ > 0) {
      if (n == 1) cv_.notify_one(); else cv_.notify_all();
    }
  }

 private:
  std::atomic<std::ptrdiff_t> count_;
  std::mutex mu_;
  std::condition_variable cv_;
  int sleepers_{0};
};
}  // namespace moodycamel
#endif  // ORACLE_SHIM_LIGHTWEIGHTSEMAPHORE_H_
