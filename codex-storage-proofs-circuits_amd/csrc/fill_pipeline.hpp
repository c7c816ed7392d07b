// The host side of the ingestion pipe (csrc/slot_trees.cpp): filling a turn's buffer from slot files (pread) or from a host array
// (memcpy) on several threads.  No HIP in here: tests/host_check/fill_pipeline_check.cpp runs it against real files on the CPU, under
// AddressSanitizer + UBSan and again under ThreadSanitizer.
//
// Bytes [0, m * cell_size) of the turn [c0, c0 + m) of batch `g` go into `buf`, from the slot files "<base><slot>.dat" (dataset.nim:34),
// zero-filled past the end of a file (slot.nim:61-66).  The turn is cut into GRAINS of 4 MiB which the fill threads take from a shared
// counter (round 6; equal byte ranges, one per thread, joined per turn, before): the formatting threads of a streamed build compete for
// the same cores, and a fill thread that loses its core for a scheduler slice no longer holds up a whole turn.  Every thread walks the
// pieces of its grain (ingest_piece) and opens the files it needs itself: nothing is held open between turns, however many files a turn
// touches.
// A fill is POSTED and JOINED apart (begin / join), two turns deep: the building thread posts turn k + 1's fill before it joins turn
// k's, so a worker that finds no grain of turn k left goes straight on to turn k + 1 -- no thread waits at a turn's end for the slowest
// one (16 slots of 8 GiB: 0.96 -> 0.99 of the fake source's rate) -- and turn k's scheduling work (layer passes, the caller's sampling
// hook) runs on the building thread while the workers read.
// O_DIRECT (cp2_set_ingest_direct / CP2_INGEST_DIRECT=1): slot files that are not in the page cache are read straight into the pinned
// ring, whole 4 KiB blocks, without passing through (and evicting) the page cache; a piece whose file offset or buffer address is not
// block aligned, the last partial block of a piece, and a file system that refuses O_DIRECT (tmpfs) are read buffered.
#pragma once
#include <fcntl.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>

#include "ingest_turns.hpp"
#include "workers.hpp"

namespace cp2i {

inline std::string fill_slot_file_name(const std::string& base, uint64_t slot) { return base + std::to_string(slot) + ".dat"; }   // dataset.nim:34

class FillPipeline {
 public:
  static constexpr size_t DIRECT_ALIGN = 4096;   // offset, length and address granule of O_DIRECT reads
  explicit FillPipeline(int threads) : threads_(threads < 1 ? 1 : threads) {
    if (threads_ > 1) pool_.reset(new Workers(threads_ - 1));
  }
  ~FillPipeline() { join_all(); }                // (the workers write into the caller's buffers: none may outlive this)
  Workers* workers() { return pool_.get(); }
  bool idle() const { return jobs_.empty(); }

  // post the fill of the turn [c0, c0 + m) into `buf`: the workers start on it as soon as they run out of grains of the job before.
  // mem != nullptr: the turn's bytes are copied from host memory at `mem` (host arrays) instead of read from slot files.
  void begin(const IngestGeom& g, const std::string& base, size_t c0, size_t m, uint8_t* buf, bool want_direct, const uint8_t* mem = nullptr) {
    auto job = std::make_shared<Job>();
    job->g = g;
    job->base = base;
    job->c0 = c0;
    job->mem = mem;
    job->nbytes = m * g.cell_size;
    job->n_grains = ingest_grain_count(job->nbytes, INGEST_FILL_GRAIN);
    job->buf = buf;
    job->direct = want_direct;
    jobs_.push_back(job);
    if (pool_)
      for (size_t t = 1; t < (size_t)threads_ && t < job->n_grains; ++t) pool_->submit([job] { take_grains(job); });
  }
  // The OLDEST posted fill is complete (this thread takes grains of it too; the workers may already be on the next job).
  // false: a slot file could not be opened -- *bad names the one of the LOWEST slot (whichever thread met it); its bytes read as zeros.
  bool join(std::string* bad) {
    if (jobs_.empty()) return true;
    std::shared_ptr<Job> job = jobs_.front();
    jobs_.pop_front();
    take_grains(job);
    {
      std::unique_lock<std::mutex> lk(job->mu);
      job->cv.wait(lk, [&] { return job->done == job->n_grains; });
      if (!job->first_bad.empty()) {
        if (bad) *bad = job->first_bad;
        return false;
      }
    }
    return true;
  }
  void join_all() {
    while (!jobs_.empty()) (void)join(nullptr);
  }

 private:
  struct Job {
    IngestGeom g;
    std::string base;
    size_t c0 = 0, nbytes = 0, n_grains = 0;
    uint8_t* buf = nullptr;
    const uint8_t* mem = nullptr;
    bool direct = false;
    std::atomic<size_t> next{0};            // the next grain nobody has taken yet
    std::mutex mu;
    std::condition_variable cv;
    size_t done = 0;                        // grains completed (under mu)
    std::string first_bad;                  // (under mu)
    uint64_t first_bad_slot = ~0ULL;
  };

  static void fill_range(Job& job, size_t a, size_t b) {
    if (job.mem) { std::memcpy(job.buf + a, job.mem + a, b - a); return; }
    const IngestGeom& g = job.g;
    uint8_t* buf = job.buf;
    for (size_t p = a; p < b;) {
      const IngestPiece q = ingest_piece(g, job.c0, p, b);
      const std::string fname = fill_slot_file_name(job.base, q.slot);
      const int fd = open(fname.c_str(), O_RDONLY);
      if (fd < 0) {
        {
          std::lock_guard<std::mutex> lk(job.mu);
          if (q.slot < job.first_bad_slot) { job.first_bad_slot = q.slot; job.first_bad = fname; }
        }
        std::memset(buf + p, 0, q.len);
        p += q.len;
        continue;
      }
      size_t pos = 0;
      if (job.direct && q.len >= DIRECT_ALIGN && q.file_off % DIRECT_ALIGN == 0 && reinterpret_cast<uintptr_t>(buf + p) % DIRECT_ALIGN == 0) {
        const int dfd = open(fname.c_str(), O_RDONLY | O_DIRECT);
        if (dfd >= 0) {
          const size_t whole = q.len / DIRECT_ALIGN * DIRECT_ALIGN;
          while (pos < whole) {
            const ssize_t r = pread(dfd, buf + p + pos, whole - pos, (off_t)(q.file_off + pos));
            if (r <= 0) break;
            pos += (size_t)r;
            if ((size_t)r % DIRECT_ALIGN) break;   // short, unaligned: end of file (the buffered reads below see that too)
          }
          close(dfd);
        }
      }
      while (pos < q.len) {
        const ssize_t r = pread(fd, buf + p + pos, q.len - pos, (off_t)(q.file_off + pos));
        if (r <= 0) break;
        pos += (size_t)r;
      }
      if (pos < q.len) std::memset(buf + p + pos, 0, q.len - pos);
      close(fd);
      p += q.len;
    }
  }
  static void take_grains(const std::shared_ptr<Job>& job) {   // any thread: grains of this job until none is left
    size_t mine = 0;
    for (;;) {
      const size_t i = job->next.fetch_add(1, std::memory_order_relaxed);
      if (i >= job->n_grains) break;
      size_t a = 0, b = 0;
      ingest_grain(job->nbytes, INGEST_FILL_GRAIN, i, &a, &b);
      fill_range(*job, a, b);
      ++mine;
    }
    if (mine) {
      std::lock_guard<std::mutex> lk(job->mu);
      job->done += mine;
      if (job->done == job->n_grains) job->cv.notify_all();
    }
  }

  int threads_;
  std::unique_ptr<Workers> pool_;
  std::deque<std::shared_ptr<Job>> jobs_;   // posted, not yet joined: at most two (the turn about to be shipped and the one after it)
};

}  // namespace cp2i
