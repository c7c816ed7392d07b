// The turns of the slot-file builder (trees_build_files, csrc/slot_trees.cpp) and the layer scheduler's decisions, as plain
// arithmetic, no HIP: how a batch of slot files ("<base><k>.dat", dataset.nim:34; cells read as slot.nim:57-68 reads them) is cut
// into ring turns, which bytes of which file a turn's buffer holds, and which slots' layer passes follow a turn.  The builder uses
// exactly these functions; the CPU suite compiles this header and walks it over >= 10^5 shapes under ASan/UBSan
// (tests/host_check/ingest_plan_check.cpp) -- round 5's only GPU fault lived in the sibling arithmetic of the fake-data builder
// (csrc/fake_turns.hpp) and was found by a random soak.
//
// Round 6: a turn is a range of the BATCH's cells, not of one slot's.  Before, a turn never crossed a file, so a dataset of small
// slots (configs[3]'s scale-down: 4096 slots of 8 MiB) was hashed 16 workgroups at a time on a device that holds 768 -- each launch
// as long as one workgroup's lifetime whatever its size.  Now a turn takes whole slots up to the ring slot's capacity (48 slots of
// 8 MiB in the default 384 MiB slot), or a piece of one large slot, or the end of one slot and the start of the next.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>

namespace cp2i {

// ---- layer scheduler: how many of the complete slots whose layers are not built yet go into the next layer pass
//   group == 0            one pass over everything at the end (`final`)
//   group, !take_all      passes of exactly `group` slots; what is left at the end goes as one shorter pass
//   group, take_all       every complete slot goes at once (passes follow the builder's turns)
//   group, !take_all, ramp_min   ... and the LAST passes shrink: once fewer than two groups are left (of `n_slots` in the batch) a pass
//                         takes half of what is left, never less than `ramp_min` slots -- what follows a pass on the host (the JSON
//                         bodies of its slots) overlaps the hashing of the next pass, so the last pass's formatting overlaps nothing
//                         and should be small.  The slot-file builder runs like this (ramp_min = the slots of one ring turn).
// Why not a pass per ring turn (round 6 tried): a pass is a chain of ~15 small dependent kernels on the third stream (the block- and
// slot-tree layers, sampling, gathers), each of which takes 0.7 ms BESIDE the hash launches against 0.1 ms alone -- about 10 ms per
// pass whatever its size.  A pass per 512 MiB turn (14 ms) keeps that stream three quarters busy, per 256 MiB turn it is the bottleneck
// (profiles/r06_streamed_files_ab.txt: 0.77 / 0.86 / 0.97 of the fake source's rate at 256 / 512 / 1024 MiB turns).
inline size_t layer_take(size_t complete, size_t built, size_t group, bool take_all, bool final, size_t n_slots = 0, size_t ramp_min = 0) {
  const size_t avail = complete > built ? complete - built : 0;
  if (group && take_all) return avail;
  if (group) {
    size_t want = group;
    if (ramp_min && n_slots > built) {
      const size_t left = n_slots - built;
      // (a remainder of up to half a ramp_min goes into the last pass rather than into one of its own: a pass has a fixed cost)
      if (left < 2 * group) want = left > ramp_min + ramp_min / 2 ? std::max(std::min(ramp_min, group), (left + 1) / 2) : left;
      if (want > group) want = group;
    }
    if (want && avail >= want) return want;
  }
  if (final) return avail;
  return 0;
}

// ---- the batch a slot-file build reads: `n_units` consecutive units of `n_cells` cells; unit i of the batch is unit first_unit + i
// of the dataset = cells [(u % units_per_slot) * n_cells, +n_cells) of the file of slot u / units_per_slot (units_per_slot 1: a unit
// is a whole slot)
struct IngestGeom {
  size_t n_units = 0, n_cells = 0, cell_size = 0;
  uint64_t first_unit = 0, units_per_slot = 1;
  size_t total_cells() const { return n_units * n_cells; }
  size_t unit_bytes() const { return n_cells * cell_size; }
};

// cells a ring slot holds: `chunk_bytes` of cells, never more than the batch, at least one cell
inline size_t ingest_chunk_cells(size_t chunk_bytes, size_t cell_size, size_t total_cells) {
  return std::max<size_t>(1, std::min(total_cells, chunk_bytes / cell_size));
}

// Cells of the turn that starts at cell c0 of the batch; `pipe_turn` counts the turns since the PIPE was set up (a pipe serves
// several batches of a transient build).
//   * the first three turns of a pipe are a quarter, a half, three quarters of a ring slot: the device starts hashing after a
//     quarter of the fill + upload latency;
//   * a turn ends on a slot (unit) boundary whenever one lies inside it -- slots smaller than the ring slot go in whole -- so the
//     layer passes that follow a turn cover whole slots and O_DIRECT pieces start on file offset 0;
//   * inside one large slot a turn ends on a multiple of `cell_multiple` cells of that slot (O_DIRECT: whole 4 KiB blocks).
inline size_t ingest_turn_cells(const IngestGeom& g, size_t chunk, size_t cell_multiple, size_t pipe_turn, size_t c0) {
  const size_t total = g.total_cells();
  size_t m = pipe_turn < 3 ? std::max<size_t>(chunk * (pipe_turn + 1) / 4, std::min<size_t>(chunk, 32768)) : chunk;
  m = std::min(m, total - c0);
  const size_t end = c0 + m;
  if (end == total) return m;
  const size_t boundary = end / g.n_cells * g.n_cells;          // the last unit boundary at or before the turn's end
  if (boundary > c0) return boundary - c0;
  if (cell_multiple > 1) {
    const size_t in_unit_end = end - boundary;                  // (boundary <= c0: the whole turn lies inside one unit)
    const size_t cut = in_unit_end % cell_multiple;
    if (cut < m) m -= cut;
  }
  return m;
}

// One piece of a turn's buffer: bytes [p, p + len) of the buffer = bytes [file_off, file_off + len) of the file of slot `slot`
// (unit `unit` of the batch).  ingest_piece gives the piece that STARTS at byte p, clipped to byte `limit` of the buffer and to the
// end of its unit.
struct IngestPiece {
  size_t unit = 0;        // index inside the batch
  uint64_t slot = 0;      // which slot file
  size_t file_off = 0, len = 0;
};
inline IngestPiece ingest_piece(const IngestGeom& g, size_t c0, size_t p, size_t limit) {
  IngestPiece q;
  const size_t ub = g.unit_bytes(), at = c0 * g.cell_size + p;   // byte position inside the batch
  q.unit = at / ub;
  const size_t in_unit = at - q.unit * ub;
  const uint64_t u = g.first_unit + q.unit;
  q.slot = u / g.units_per_slot;
  q.file_off = (size_t)(u % g.units_per_slot) * ub + in_unit;
  q.len = std::min(limit - p, ub - in_unit);
  return q;
}

// The fill of a turn (slot files: pread; host arrays: memcpy): grains of INGEST_FILL_GRAIN bytes (a multiple of the O_DIRECT granule),
// taken by the fill threads from a shared counter; grain i of a turn of n bytes is [a, b).
constexpr size_t INGEST_FILL_GRAIN = (size_t)4 << 20;
inline size_t ingest_grain_count(size_t n, size_t grain) { return (n + grain - 1) / grain; }
inline void ingest_grain(size_t n, size_t grain, size_t i, size_t* a, size_t* b) {
  *a = std::min(n, i * grain);
  *b = std::min(n, (i + 1) * grain);
}

// the units a turn of m cells starting at cell c0 touches: [first, last]
inline void ingest_turn_units(const IngestGeom& g, size_t c0, size_t m, size_t* first, size_t* last) {
  *first = c0 / g.n_cells;
  *last = (c0 + m - 1) / g.n_cells;
}

}  // namespace cp2i
