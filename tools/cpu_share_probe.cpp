// How many cores does this process really get?  N busy threads for a fixed wall time, total iterations against one thread's:
// the ratio is the effective core count (a cgroup CPU quota shows up here, not in sched_getaffinity).  g++ -O2 -pthread
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double run(int n, double secs) {
  std::atomic<bool> stop{false};
  std::vector<unsigned long long> cnt(n, 0);
  std::vector<std::thread> th;
  for (int i = 0; i < n; ++i)
    th.emplace_back([&, i] {
      unsigned long long c = 0, x = 88172645463325252ULL + i;
      while (!stop.load(std::memory_order_relaxed)) {
        for (int k = 0; k < 4096; ++k) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; }
        c += 4096 + (x & 1 ? 0 : 0);
      }
      cnt[i] = c;
    });
  std::this_thread::sleep_for(std::chrono::duration<double>(secs));
  stop = true;
  for (auto& t : th) t.join();
  unsigned long long tot = 0;
  for (auto c : cnt) tot += c;
  return (double)tot / secs;
}
int main() {
  const double one = run(1, 0.5);
  for (int n : {4, 8, 16, 24, 32, 64}) std::printf("%3d busy threads: %.1f x one thread\n", n, run(n, 0.5) / one);
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) { char b[128] = {0}; if (std::fgets(b, sizeof b, f)) std::printf("cgroup cpu.max: %s", b); std::fclose(f); }
  return 0;
}
