// Host worker threads (no HIP in here: the CPU suite compiles this header and what is built on it under the sanitizers).
#pragma once
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace cp2i {

// Host worker threads with a FIFO of tasks; the destructor drains the queue and joins (no joinable thread is
// ever destroyed, whatever path leaves the owning scope).
class Workers {
 public:
  // nice_inc > 0: the threads give way to the process's other threads when cores are short (Linux: a thread's nice value is its own).
  // The streamed build's formatting threads run like that: what they produce is needed at the END of the build, while the fill
  // threads of the slot-file pipe and the building thread feed the device NOW.
  explicit Workers(int n, int nice_inc = 0) : nice_inc_(nice_inc) {
    if (n < 1) n = 1;
    for (int i = 0; i < n; ++i) th_.emplace_back([this] { run(); });
  }
  ~Workers() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void submit(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      q_.push_back(std::move(f));
    }
    cv_.notify_one();
  }
  void wait_idle() {
    std::unique_lock<std::mutex> lk(mu_);
    idle_.wait(lk, [this] { return q_.empty() && busy_ == 0; });
  }
  size_t size() const { return th_.size(); }

 private:
  void run() {
    if (nice_inc_ > 0) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), nice_inc_);
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;   // stop_ and drained
        f = std::move(q_.front());
        q_.pop_front();
        ++busy_;
      }
      try { f(); } catch (...) {}   // tasks report through their own status words
      {
        std::lock_guard<std::mutex> lk(mu_);
        --busy_;
      }
      idle_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, idle_;
  std::deque<std::function<void()>> q_;
  size_t busy_ = 0;
  bool stop_ = false;
  int nice_inc_ = 0;
};

}  // namespace cp2i
