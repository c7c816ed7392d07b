// The turns of the fake-data builder (trees_build_fake, csrc/slot_trees.cpp): how a batch of generated slots is cut into staging
// chunks, which of the two staging buffers / hashing streams a turn uses, and whether a second buffer is needed at all.  Plain
// arithmetic, no HIP: the CPU suite compiles this header and walks it over thousands of shapes (tests/host_check/
// turn_plan_check.cpp) -- round 5's soak found a GPU fault here: a dataset that fits ONE chunk but holds more than one residency
// of the hash kernel is cut into several turns by the ramp-down, and the second turn had no staging buffer.
#pragma once
#include <algorithm>
#include <cstddef>

namespace cp2i {

struct FakeTurnPlan {
  size_t total_cells = 0;
  size_t chunk = 0;        // cells per full turn
  bool ramp = false;       // groups: the last turns shrink down to one residency of the hash kernel
  size_t g_slots = 0;      // ramp: slots per full turn
  size_t g_min = 0;        // ramp: slots of the smallest turn
  bool two = false;        // more than one turn: turns alternate between two staging buffers (and hashing streams)
};

// One residency of k_hash_cells -- what the ramp-down halves the last groups down to: 256 CUs x 3 workgroups x 256 cells at full
// occupancy, x 2 workgroups when the launches leave room (group builds: launch_hash_cells' leave_room).  The builder passes the one
// it launches with.
constexpr size_t FAKE_RESIDENCY_CELLS = (size_t)768 * 256, FAKE_RESIDENCY_CELLS_WITH_ROOM = (size_t)512 * 256;

inline FakeTurnPlan fake_turn_plan(size_t n_slots, size_t n_cells, size_t cell_size, size_t stage_bytes, size_t group, bool ramp_allowed,
                                   size_t residency_cells = FAKE_RESIDENCY_CELLS) {
  FakeTurnPlan p;
  p.total_cells = n_slots * n_cells;
  // staging chunk: up to `stage_bytes` of generated cells, a whole number of slots when slots are smaller than that
  p.chunk = std::max<size_t>(1, std::min(p.total_cells, stage_bytes / cell_size));
  if (p.chunk > n_cells) p.chunk -= p.chunk % n_cells;
  if (group && p.chunk > group * n_cells) p.chunk = group * n_cells;
  // Groups (the streamed proof-input path): what follows a group on the host -- the JSON bodies of its slots -- overlaps the
  // hashing of the NEXT group, so the last group's formatting overlaps nothing.  When a chunk is a whole number of slots the
  // last groups are therefore halved down to one residency of the hash kernel: 256, 256, ..., 128, 64, 48 slots of 2^12 cells
  // instead of a final 256, and the un-overlapped tail shrinks from ~50 ms of formatting to ~10.
  p.ramp = group && ramp_allowed && p.chunk >= n_cells && p.chunk % n_cells == 0;
  p.g_slots = p.ramp ? p.chunk / n_cells : 0;
  p.g_min = p.ramp ? std::max<size_t>(1, std::min(p.g_slots, std::max<size_t>(1, residency_cells) / n_cells)) : 0;
  // a second staging buffer whenever there is a second turn: more cells than one chunk, or a ramp that cuts even a single
  // chunk into several turns
  p.two = p.total_cells > p.chunk || (p.ramp && n_slots > p.g_min);
  return p;
}

// cells of the turn that starts at cell c0 of the batch
inline size_t fake_turn_cells(const FakeTurnPlan& p, size_t n_cells, size_t c0) {
  size_t n = std::min(p.chunk, p.total_cells - c0);
  if (p.ramp) {
    const size_t left = (p.total_cells - c0) / n_cells;
    n = (left >= 2 * p.g_slots ? p.g_slots : (left > p.g_min ? std::max(p.g_min, (left + 1) / 2) : left)) * n_cells;
  }
  return n;
}

// staging buffer / hashing stream of turn number `turn`
inline int fake_turn_side(const FakeTurnPlan& p, size_t turn, bool serial) { return (serial || !p.two) ? 0 : (int)(turn & 1); }

}  // namespace cp2i
