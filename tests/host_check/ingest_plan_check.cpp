// CPU check of the slot-file builder's arithmetic (csrc/ingest_turns.hpp, the header trees_build_files and LayerScheduler themselves
// use) and, beside it, of the layer scheduler over the fake-data builder's turns (csrc/fake_turns.hpp): for >= 10^5 shapes -- unit
// counts, cells per unit, cell sizes, ring-slot sizes, O_DIRECT granules, slots cut into units, pipes that have already served a batch,
// group sizes -- walk the turns exactly as the builder does and check what a GPU run can only check by faulting or by a wrong root:
//   turns    every turn is non-empty, fits its ring buffer, starts where the previous one ended, ends on a slot boundary whenever one
//            is in reach; the turns add up to the batch
//   pieces   the fill grains tile the turn's buffer; the pieces of every grain tile the grain; every piece lies inside
//            ONE unit's byte range of ONE file; every byte of the buffer comes from the file byte slot.nim:57-68 reads for that cell
//            (recomputed independently); small shapes are written into a real buffer of exactly the turn's size (AddressSanitizer
//            sees any byte outside it) and every byte is written exactly once
//   layers   every slot's layers are built exactly once, never before its last cell has been enqueued for hashing, all of them
//            by the end; a pass of a grouped build that does not follow the turns never exceeds the group
// g++ -std=c++17 -fsanitize=address,undefined -I<csrc> ingest_plan_check.cpp
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fake_turns.hpp"
#include "ingest_turns.hpp"

using namespace cp2i;

static uint64_t rng_state = 0x2545f4914f6cdd1dULL;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

struct Shape {
  IngestGeom g;
  size_t chunk_bytes, cell_multiple, pipe_turn0, group;
  bool take_all, ramp;
  int threads;
};

[[noreturn]] static void fail(const Shape& s, const char* what, size_t turn, size_t c0, size_t m) {
  std::printf("FAILED: %s  (units %zu x %zu cells of %zu B, first_unit %llu, units_per_slot %llu, chunk %zu B, cell_multiple %zu, pipe turn %zu, group %zu take_all %d ramp %d, threads %d: turn %zu at cell %zu, %zu cells)\n",
              what, s.g.n_units, s.g.n_cells, s.g.cell_size, (unsigned long long)s.g.first_unit, (unsigned long long)s.g.units_per_slot, s.chunk_bytes, s.cell_multiple,
              s.pipe_turn0, s.group, (int)s.take_all, (int)s.ramp, s.threads, turn, c0, m);
  std::exit(1);
}

// Layer passes over a sequence of turns: times[slot] counts how often a slot's layers were built.
struct LayerWalk {
  size_t n_slots, n_cells, group;
  bool take_all;
  size_t ramp_min = 0;
  size_t built = 0, passes = 0, max_pass = 0;
  std::vector<uint8_t> times;
  LayerWalk(size_t ns, size_t nc, size_t g, bool ta) : n_slots(ns), n_cells(nc), group(g), take_all(ta), times(ns, 0) {}
  // after a turn: cells [0, cells_hashed) are enqueued.  false on a violation.
  bool advance(size_t cells_hashed, bool final) {
    const size_t complete = cells_hashed / n_cells;
    for (;;) {
      const size_t take = layer_take(complete, built, group, take_all, final, n_slots, ramp_min);
      if (!take) return true;
      if (built + take > complete || built + take > n_slots) return false;        // a slot whose cells are not all enqueued yet
      for (size_t s = built; s < built + take; ++s) ++times[s];
      built += take;
      ++passes;
      if (take > max_pass) max_pass = take;
    }
  }
  bool all_once() const {
    for (uint8_t t : times)
      if (t != 1) return false;
    return built == n_slots;
  }
};

static long check_shape(const Shape& s, long* pieces_out, long* bytes_checked) {
  const IngestGeom& g = s.g;
  const size_t total = g.total_cells();
  const size_t chunk = ingest_chunk_cells(s.chunk_bytes, g.cell_size, total);
  if (chunk == 0 || chunk > total) fail(s, "ring-slot capacity out of range", 0, 0, chunk);
  const size_t cap_bytes = chunk * g.cell_size;
  LayerWalk layers(g.n_units, g.n_cells, s.group, s.take_all);
  layers.ramp_min = s.ramp ? std::max<size_t>(1, chunk / g.n_cells) : 0;       // as the slot-file builder sets it
  size_t turns = 0;
  long pieces = 0;
  for (size_t c0 = 0, m = 0; c0 < total; c0 += m, ++turns) {
    m = ingest_turn_cells(g, chunk, s.cell_multiple, s.pipe_turn0 + turns, c0);
    if (m == 0) fail(s, "empty turn (the builder's loop would never end)", turns, c0, m);
    if (m > chunk || m * g.cell_size > cap_bytes) fail(s, "turn larger than its ring buffer", turns, c0, m);
    if (c0 + m > total) fail(s, "turn past the end of the batch", turns, c0, m);
    if (c0 + m < total) {
      // a slot boundary inside (c0, c0 + m]: the turn must end on one
      const size_t last_boundary = (c0 + m) / g.n_cells * g.n_cells;
      if (last_boundary > c0 && last_boundary != c0 + m) fail(s, "turn that crosses a slot boundary without ending on one", turns, c0, m);
      if (last_boundary <= c0 && s.cell_multiple > 1 && m > s.cell_multiple && ((c0 + m) % g.n_cells) % s.cell_multiple != 0)
        fail(s, "turn inside a large slot that does not end on the O_DIRECT granule", turns, c0, m);
    }
    size_t u0 = 0, u1 = 0;
    ingest_turn_units(g, c0, m, &u0, &u1);
    if (u0 > u1 || u1 >= g.n_units || u0 != c0 / g.n_cells) fail(s, "turn's unit range", turns, c0, m);
    // ---- the fill: ranges tile the buffer, pieces tile the ranges
    const size_t nbytes = m * g.cell_size;
    // the product cuts a turn's bytes into grains taken from a shared counter (fill_begin / fill_grains); the walk also uses grains far
    // below the product's 4 MiB so that small shapes split
    const size_t align = (s.cell_multiple > 1) ? 4096 : 1;
    const size_t grain = std::max<size_t>(std::max<size_t>(align, (nbytes / 512 + align - 1) / align * align), (INGEST_FILL_GRAIN >> (rnd() % 14)) / align * align);   // (at most ~512 grains per turn: the walk stays short)
    const int nt = (int)ingest_grain_count(nbytes, grain);
    const bool real = nbytes <= ((size_t)1 << 14);
    std::vector<uint8_t> buf, hits;
    if (real) { buf.assign(nbytes, 0); hits.assign(nbytes, 0); }
    size_t prev_end = 0;
    if (INGEST_FILL_GRAIN % 4096) fail(s, "the fill grain is not a multiple of the O_DIRECT granule", turns, c0, m);
    for (int t = 0; t < nt; ++t) {
      size_t a = 0, b = 0;
      ingest_grain(nbytes, grain, (size_t)t, &a, &b);
      if (a != prev_end || b < a || b > nbytes) fail(s, "fill ranges do not tile the turn's buffer", turns, c0, m);
      if (t > 0 && a % align) fail(s, "inner fill boundary off the O_DIRECT granule", turns, c0, m);
      prev_end = b;
      for (size_t p = a; p < b;) {
        const IngestPiece q = ingest_piece(g, c0, p, b);
        ++pieces;
        if (q.len == 0 || p + q.len > b) fail(s, "piece empty or past its range", turns, c0, m);
        if (q.unit < u0 || q.unit > u1) fail(s, "piece outside the turn's units", turns, c0, m);
        const uint64_t u = g.first_unit + q.unit;
        if (q.slot != u / g.units_per_slot) fail(s, "piece in the wrong slot file", turns, c0, m);
        const size_t unit_lo = (size_t)(u % g.units_per_slot) * g.unit_bytes();
        if (q.file_off < unit_lo || q.file_off + q.len > unit_lo + g.unit_bytes()) fail(s, "piece outside its unit's byte range of the file", turns, c0, m);
        // independent restatement: buffer byte x holds byte (x % cell_size) of cell c0 + x / cell_size of the batch
        auto want_off = [&](size_t x, uint64_t* slot) {
          const size_t cell = c0 + x / g.cell_size, unit = cell / g.n_cells, in_unit = cell % g.n_cells;
          const uint64_t uu = g.first_unit + unit;
          *slot = uu / g.units_per_slot;
          return ((size_t)(uu % g.units_per_slot) * g.n_cells + in_unit) * g.cell_size + x % g.cell_size;
        };
        uint64_t sl = 0;
        if (want_off(p, &sl) != q.file_off || sl != q.slot) fail(s, "piece starts at the wrong file byte", turns, c0, m);
        if (want_off(p + q.len - 1, &sl) != q.file_off + q.len - 1 || sl != q.slot) fail(s, "piece ends at the wrong file byte", turns, c0, m);
        if (real) {
          std::memset(buf.data() + p, 0xA5, q.len);             // what the builder's pread / memset does: ASan traps a byte outside the buffer
          for (size_t x = p; x < p + q.len; ++x) {
            ++hits[x];
            if (want_off(x, &sl) != q.file_off + (x - p) || sl != q.slot) fail(s, "byte from the wrong file position", turns, c0, m);
          }
          *bytes_checked += (long)q.len;
        }
        p += q.len;
      }
    }
    if (prev_end != nbytes) fail(s, "fill ranges stop short of the turn's end", turns, c0, m);
    if (real)
      for (size_t x = 0; x < nbytes; ++x)
        if (hits[x] != 1) fail(s, "buffer byte not written exactly once", turns, c0, m);
    // ---- the layer passes that follow this turn
    if (!layers.advance(c0 + m, c0 + m == total)) fail(s, "layer pass over a slot whose cells are not all enqueued", turns, c0, m);
    if (turns > (size_t)1 << 22) fail(s, "too many turns", turns, c0, m);
  }
  if (!layers.all_once()) fail(s, "a slot's layers built never or twice", turns, 0, 0);
  if (s.group && !s.take_all && layers.max_pass > s.group) fail(s, "a pass of a grouped build larger than the group", turns, 0, layers.max_pass);
  *pieces_out += pieces;
  return (long)turns;
}

// the layer scheduler over the FAKE builder's turns (trees_build_fake: take_all = plan.ramp)
static void check_fake_layers(size_t n_slots, size_t n_cells, size_t cell_size, size_t stage, size_t group, bool ramp) {
  const FakeTurnPlan p = fake_turn_plan(n_slots, n_cells, cell_size, stage, group, ramp);
  LayerWalk layers(n_slots, n_cells, group, p.ramp);
  for (size_t c0 = 0, n = 0; c0 < p.total_cells; c0 += n) {
    n = fake_turn_cells(p, n_cells, c0);
    if (n == 0 || !layers.advance(c0 + n, c0 + n == p.total_cells)) {
      std::printf("FAILED: fake builder layer walk (%zu slots x %zu cells, group %zu ramp %d)\n", n_slots, n_cells, group, (int)ramp);
      std::exit(1);
    }
  }
  if (!layers.all_once()) {
    std::printf("FAILED: fake builder: a slot's layers built never or twice (%zu slots x %zu cells, group %zu ramp %d)\n", n_slots, n_cells, group, (int)ramp);
    std::exit(1);
  }
}

int main(int argc, char** argv) {
  const long want = argc > 1 ? std::atol(argv[1]) : 120000;
  long shapes = 0, turns = 0, pieces = 0, bytes = 0, multi_file_shapes = 0, split_large_shapes = 0, fake_shapes = 0;
  const size_t cell_sizes[] = {1, 31, 64, 100, 128, 256, 2047, 2048, 4096, 16384};
  const size_t chunk_bytes[] = {(size_t)1 << 12, (size_t)1 << 16, (size_t)1 << 20, (size_t)64 << 20, (size_t)384 << 20, (size_t)1 << 30};
  // the shapes the round's measurements run, literally: configs[3]'s scale-down and nominal slots, default ring slot
  {
    Shape s{};
    s.g.n_units = 4096; s.g.n_cells = 4096; s.g.cell_size = 2048; s.chunk_bytes = (size_t)768 << 20; s.cell_multiple = 1; s.group = 256; s.take_all = false; s.ramp = true; s.threads = 8;
    turns += check_shape(s, &pieces, &bytes); ++shapes;
    s.g.n_units = 4; s.g.n_cells = (size_t)1 << 22; s.cell_multiple = 2; s.group = 1;
    turns += check_shape(s, &pieces, &bytes); ++shapes;
  }
  while (shapes < want) {
    Shape s{};
    s.g.cell_size = cell_sizes[rnd() % 10];
    s.g.n_cells = (rnd() % 5 == 0) ? 1 + rnd() % 5000 : (size_t)1 << (rnd() % 23);
    s.g.n_units = 1 + rnd() % ((rnd() % 4 == 0) ? 5000 : 200);
    while (s.g.n_units * s.g.n_cells > ((size_t)1 << 30)) s.g.n_units = s.g.n_units / 2 + 1;
    s.g.units_per_slot = (rnd() % 3 == 0) ? (uint64_t)1 << (rnd() % 5) : 1;
    s.g.first_unit = (rnd() % 2) ? rnd() % 100000 : 0;
    s.chunk_bytes = chunk_bytes[rnd() % 6];
    if (s.chunk_bytes < s.g.cell_size) s.chunk_bytes = s.g.cell_size;
    {   // O_DIRECT granule as the builder derives it: 4096 / gcd(cell_size, 4096), or 1 (buffered)
      size_t a = s.g.cell_size, h = 4096;
      while (h) { size_t r = a % h; a = h; h = r; }
      s.cell_multiple = (rnd() % 2) ? 4096 / a : 1;
    }
    s.pipe_turn0 = (rnd() % 3 == 0) ? rnd() % 50 : 0;
    s.group = (rnd() % 3 == 0) ? 0 : 1 + rnd() % ((rnd() % 2) ? 8 : 600);
    s.take_all = s.group != 0 && (rnd() % 4 == 0);
    s.ramp = s.group != 0 && !s.take_all && (rnd() % 3 != 0);
    s.threads = 1 + (int)(rnd() % 16);
    const size_t chunk = ingest_chunk_cells(s.chunk_bytes, s.g.cell_size, s.g.total_cells());
    if (s.g.total_cells() / chunk > 400) continue;   // (shapes of very many turns prove nothing more: skip them)
    const long before = pieces;
    const long t = check_shape(s, &pieces, &bytes);
    ++shapes;
    turns += t;
    multi_file_shapes += (chunk >= 2 * s.g.n_cells && s.g.n_units > 2);
    split_large_shapes += (pieces - before > t && chunk < s.g.n_cells);
    if (shapes % 4 == 0 && s.g.units_per_slot == 1 && s.g.total_cells() <= ((size_t)1 << 26)) {
      check_fake_layers(s.g.n_units, s.g.n_cells, s.g.cell_size, s.chunk_bytes < 4096 ? 4096 : s.chunk_bytes, s.group, (rnd() % 2) != 0);
      ++fake_shapes;
    }
  }
  if (multi_file_shapes == 0 || split_large_shapes == 0) {
    std::printf("FAILED: the walk never met turns of many files (%ld) or turns inside one large slot split over threads (%ld)\n", multi_file_shapes, split_large_shapes);
    return 1;
  }
  std::printf("ingest plan ok: %ld shapes, %ld turns, %ld pieces, %ld buffer bytes written under the sanitizer; %ld shapes whose turns hold several files, %ld of turns inside one large slot; layer passes exactly once per slot in all of them and over %ld fake-builder plans\n",
              shapes, turns, pieces, bytes, multi_file_shapes, split_large_shapes, fake_shapes);
  return 0;
}
