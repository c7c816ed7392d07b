// Where the streamed build keeps the proof-input bodies of its slots: host memory up to a budget, private spill files beyond it.
// Not installed; included by proof_input.cpp alone.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "internal.hpp"

// JSON bodies (", \"cellData\": ... }") of the streamed build, one per local slot.  Bodies stay in host memory up to the
// context's budget (cp2_set_body_budget; 0.7 MB each at nSamples = 100, cellSize = 2048: 23 GB for 32 768 local slots if
// nothing bounded it); beyond it they go to files and are read back at export.  The bodies hold sampled cell data, so the
// files live in a PRIVATE directory made by mkdtemp (mode 0700, unpredictable name) under the spill directory, each file
// created with O_EXCL | O_NOFOLLOW and mode 0600: nothing planted in a shared /tmp is followed or overwritten, two processes
// with equal pids (containers sharing a spill volume) cannot meet, and nobody else can read them.  The directory and its
// files go with the dataset.  put() is called from the formatting workers, everything else from the owning thread.
struct BodyStore {
  std::vector<std::string> mem;
  std::vector<uint64_t> size;        // text length of every body, resident or spilled
  std::vector<uint8_t> spilled;
  std::string base, dir;             // base: the caller's spill directory; dir: the private directory, made at the first spill
  std::string error;                 // first spill failure, with the path (read by the owning thread after the workers are idle)
  size_t budget = 0;
  std::atomic<size_t> resident{0};
  std::atomic<size_t> n_spilled{0};
  std::mutex mu;
  ~BodyStore() {
    for (size_t s = 0; s < spilled.size(); ++s)
      if (spilled[s]) (void)unlink(file_of(s).c_str());
    if (!dir.empty()) (void)rmdir(dir.c_str());
  }
  void init(cp2_ctx* ctx, size_t n) {
    mem.assign(n, std::string());
    size.assign(n, 0);
    spilled.assign(n, 0);
    budget = ctx->body_budget;
    if (!budget) {
      const char* e = std::getenv("CP2_BODY_BUDGET_MB");
      const unsigned long long mb = e ? std::strtoull(e, nullptr, 10) : 0;
      budget = mb ? (size_t)mb << 20 : (size_t)4 << 30;
    }
    base = ctx->spill_dir;
    if (base.empty()) { const char* t = std::getenv("TMPDIR"); base = (t && *t) ? t : "/tmp"; }
  }
  std::string file_of(size_t s) const { return dir + "/body_" + std::to_string(s) + ".part"; }
  // the private directory, created once (any worker may be the first to spill)
  bool ensure_dir() {
    std::lock_guard<std::mutex> lk(mu);
    if (!dir.empty()) return true;
    std::string tmpl = base + "/cp2_bodies_XXXXXX";
    std::vector<char> buf(tmpl.begin(), tmpl.end());
    buf.push_back(0);
    if (!mkdtemp(buf.data())) {
      if (error.empty()) error = "cannot create a private spill directory under " + base + ": " + std::strerror(errno);
      return false;
    }
    dir = buf.data();
    return true;
  }
  int fail(const std::string& what) {
    std::lock_guard<std::mutex> lk(mu);
    if (error.empty()) error = what;
    return CP2_ERR_IO;
  }
  // takes a copy of exactly text.size() bytes (the caller's buffer is sized for the worst case and reused)
  int put(size_t s, const std::string& text) {
    size[s] = text.size();
    const size_t before = resident.fetch_add(text.size());
    if (before + text.size() <= budget) {
      mem[s].assign(text.data(), text.size());
      return CP2_OK;
    }
    resident.fetch_sub(text.size());
    if (!ensure_dir()) return CP2_ERR_IO;
    const std::string name = file_of(s);
    const int fd = open(name.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return fail("cannot create spill file " + name + ": " + std::strerror(errno));
    const char* p = text.data();
    size_t left = text.size();
    bool ok = true;
    while (ok && left) {
      const ssize_t w = write(fd, p, left);
      if (w <= 0) { ok = false; break; }
      p += w;
      left -= (size_t)w;
    }
    const int err = errno;
    if (close(fd) != 0) ok = false;
    if (!ok) {
      (void)unlink(name.c_str());
      return fail("cannot write spill file " + name + ": " + std::strerror(err));
    }
    spilled[s] = 1;
    n_spilled.fetch_add(1);
    return CP2_OK;
  }
  // appends the body of slot s to `out`
  int append(size_t s, std::string& out) const {
    if (!spilled[s]) { out.append(mem[s]); return CP2_OK; }
    FILE* f = open_spilled(s);
    if (!f) return CP2_ERR_IO;
    const size_t at = out.size();
    out.resize(at + size[s]);
    const bool ok = std::fread(&out[at], 1, size[s], f) == size[s];
    std::fclose(f);
    return ok ? CP2_OK : CP2_ERR_IO;
  }
  // writes the body of slot s to an open file (spilled bodies are copied through a bounded buffer)
  int write_to(size_t s, FILE* dst) const {
    if (!spilled[s]) return std::fwrite(mem[s].data(), 1, mem[s].size(), dst) == mem[s].size() ? CP2_OK : CP2_ERR_IO;
    FILE* f = open_spilled(s);
    if (!f) return CP2_ERR_IO;
    std::vector<char> buf((size_t)1 << 20);
    uint64_t left = size[s];
    bool ok = true;
    while (ok && left) {
      const size_t m = (size_t)std::min<uint64_t>(left, buf.size());
      ok = std::fread(buf.data(), 1, m, f) == m && std::fwrite(buf.data(), 1, m, dst) == m;
      left -= m;
    }
    std::fclose(f);
    return ok ? CP2_OK : CP2_ERR_IO;
  }

 private:
  FILE* open_spilled(size_t s) const {
    const int fd = open(file_of(s).c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return nullptr;
    FILE* f = fdopen(fd, "rb");
    if (!f) close(fd);
    return f;
  }
};

