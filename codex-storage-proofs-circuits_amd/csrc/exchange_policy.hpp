// When may the multi-device exchange of slot roots (csrc/multi_gpu.cpp) be tried once more through host memory?  Plain logic, no
// HIP: the CPU suite compiles this header (tests/test_abi_cpu.py).
//
// Only in the automatic mode, only once, only with more than one shard -- and NEVER after a time-out.  A launch error or a failed
// verification leaves the participating streams drained: host memory then carries the same 1 MiB just as well.  A time-out leaves a
// collective (or a peer copy) queued on those very streams; the host path enqueues its downloads behind it and then waits without a
// bound, so the "bounded" exchange would never return after all.  A time-out is final: the error goes to the caller, the
// contexts are marked as taking no further work, their pooled buffers are dropped from the books rather than waited for.
#pragma once
#include <cstddef>

#include "../../include/codex_p2.h"

namespace cp2i {

inline bool exchange_may_retry_on_host(int status, bool timed_out, int gather_mode, size_t world, int attempt) {
  if (status == CP2_OK || timed_out) return false;
  return attempt == 0 && gather_mode == CP2_GATHER_AUTO && world > 1 && status == CP2_ERR_HIP;
}

}  // namespace cp2i
