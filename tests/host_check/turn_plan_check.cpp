// CPU check of the fake-data builder's turn plan (csrc/fake_turns.hpp, used by trees_build_fake): for thousands of shapes --
// slot counts, cells per slot, cell sizes, staging sizes, group sizes, ramp on / off, serial on / off -- walk the turns exactly as
// the builder does and check what a GPU run can only check by faulting: every turn is non-empty, fits the staging chunk, starts
// where the previous one ended, is a whole number of slots when the ramp is on; the turns add up to the batch; a turn on the
// second buffer only when the plan said there are two.  g++ -std=c++17 -fsanitize=address,undefined -I<csrc> turn_plan_check.cpp
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "fake_turns.hpp"

using namespace cp2i;

static uint64_t rng_state = 0x9e3779b97f4a7c15ULL;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static size_t residency = FAKE_RESIDENCY_CELLS;   // the builder passes 768 x 256 (full occupancy) or 512 x 256 (launches that leave room)

static long check_shape(size_t n_slots, size_t n_cells, size_t cell_size, size_t stage_bytes, size_t group, bool ramp_allowed, bool serial) {
  const FakeTurnPlan p = fake_turn_plan(n_slots, n_cells, cell_size, stage_bytes, group, ramp_allowed, residency);
  auto fail = [&](const char* what, size_t turn, size_t c0, size_t n) {
    std::printf("FAILED: %s  (n_slots %zu n_cells %zu cell_size %zu stage %zu group %zu ramp %d serial %d: turn %zu at cell %zu, %zu cells; chunk %zu two %d)\n", what, n_slots,
                n_cells, cell_size, stage_bytes, group, (int)ramp_allowed, (int)serial, turn, c0, n, p.chunk, (int)p.two);
    std::exit(1);
  };
  if (p.total_cells != n_slots * n_cells) fail("total", 0, 0, 0);
  if (p.chunk == 0 || p.chunk > p.total_cells) fail("chunk out of range", 0, 0, p.chunk);
  size_t turns = 0, used_second = 0;
  for (size_t c0 = 0, n = 0; c0 < p.total_cells; c0 += n, ++turns) {
    n = fake_turn_cells(p, n_cells, c0);
    const int s = fake_turn_side(p, turns, serial);
    if (n == 0) fail("empty turn (the builder's loop would never end)", turns, c0, n);
    if (n > p.chunk) fail("turn larger than the staging chunk", turns, c0, n);
    if (c0 + n > p.total_cells) fail("turn past the end of the batch", turns, c0, n);
    if (p.ramp && (n % n_cells || c0 % n_cells)) fail("ramp turn that is not whole slots", turns, c0, n);
    if (s != 0 && s != 1) fail("side", turns, c0, n);
    if (s == 1 && !p.two) fail("turn on the SECOND staging buffer, which the plan does not allocate", turns, c0, n);
    if (serial && s != 0) fail("serial order on the second stream", turns, c0, n);
    used_second += s == 1;
    if (turns > (size_t)1 << 22) fail("too many turns", turns, c0, n);   // (an empty-turn bug would otherwise spin)
  }
  if (!serial && turns > 1 && !p.two) fail("several turns on one buffer although two streams alternate", turns, 0, 0);
  if (!serial && p.two && turns > 1 && used_second == 0) fail("two buffers planned, second never used", turns, 0, 0);
  return (long)turns;
}

int main() {
  long shapes = 0, turns = 0, multi_turn_single_chunk = 0;
  // the case of round 5's GPU fault, literally: 100 slots of 2^12 cells of 2 KiB, default staging (2 GiB), default group (one chunk)
  {
    const FakeTurnPlan p = fake_turn_plan(100, 4096, 2048, (size_t)2048 << 20, 256, true);
    if (!(p.total_cells <= p.chunk && p.ramp && p.two)) { std::printf("FAILED: the round-5 case is not planned with two buffers\n"); return 1; }
  }
  const size_t cell_sizes[] = {31, 64, 100, 256, 2048, 16384};
  const size_t stages[] = {(size_t)1 << 20, (size_t)16 << 20, (size_t)96 << 20, (size_t)2048 << 20};
  for (int it = 0; it < 60000; ++it) {
    const size_t cs = cell_sizes[rnd() % 6];
    const size_t n_cells = (size_t)1 << (rnd() % 23);                       // 1 .. 2^22 cells per slot
    size_t n_slots = 1 + rnd() % ((rnd() % 4 == 0) ? 5000 : 300);
    while (n_slots * n_cells > ((size_t)1 << 34)) n_slots = n_slots / 2 + 1;   // keep the walk short
    const size_t stage = stages[rnd() % 4];
    const size_t group = (rnd() % 3 == 0) ? 0 : 1 + rnd() % ((rnd() % 2) ? 8 : 600);
    {   // (shapes of millions of turns prove nothing more: skip them)
      const FakeTurnPlan q = fake_turn_plan(n_slots, n_cells, cs, stage, group, true);
      if (q.total_cells / q.chunk > 100000) continue;
    }
    residency = (it & 1) ? FAKE_RESIDENCY_CELLS_WITH_ROOM : FAKE_RESIDENCY_CELLS;
    for (int ramp = 0; ramp < 2; ++ramp)
      for (int serial = 0; serial < 2; ++serial) {
        const long t = check_shape(n_slots, n_cells, cs, stage, group, ramp != 0, serial != 0);
        ++shapes;
        turns += t;
        const FakeTurnPlan p = fake_turn_plan(n_slots, n_cells, cs, stage, group, ramp != 0, residency);
        multi_turn_single_chunk += (p.total_cells <= p.chunk && t > 1);
      }
  }
  if (multi_turn_single_chunk == 0) { std::printf("FAILED: the walk never met the case of round 5's fault (a single chunk cut into several turns)\n"); return 1; }
  std::printf("turn plan ok: %ld shapes, %ld turns walked, %ld shapes where a single chunk is cut into several turns\n", shapes, turns, multi_turn_single_chunk);
  return 0;
}
