// CPU check of the ingestion pipe's HOST side (csrc/fill_pipeline.hpp: the very class IngestPipe uses): real slot files in a scratch
// directory, turns cut by csrc/ingest_turns.hpp, fills posted TWO TURNS DEEP on several threads into a ring of exactly-sized heap
// buffers, every turn's bytes compared with the reference's own way of reading a cell (slot.nim:57-68: seek cellSize * idx, read
// cellSize bytes, what the file does not hold is zero) -- short files, a missing file (reported by name, the lowest slot first),
// slots cut into units, O_DIRECT requested, and the host-array source (memcpy).  Built twice by the CPU suite: with
// -fsanitize=address,undefined (a byte outside a ring buffer, a use after a turn was joined) and with -fsanitize=thread (the grain
// counter, the completion count, the error slot: workers run on into the next turn while the building thread joins this one).
//   g++ -std=c++17 -pthread -fsanitize=... -I<csrc> fill_pipeline_check.cpp -o check && ./check <scratch dir> [shapes]
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fill_pipeline.hpp"

using namespace cp2i;

static uint64_t rng_state = 0x853c49e6748fea9bULL;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

// byte x of slot file `slot` as written below: a function of (slot, x), never zero, so that zero-fill is distinguishable
static uint8_t file_byte(uint64_t slot, size_t x) { return (uint8_t)(1 + ((slot * 131 + x * 7 + (x >> 9)) % 251)); }

struct Dataset {
  std::string base;
  std::vector<size_t> file_bytes;   // per slot; (size_t)-1: the file does not exist
};

static void write_files(Dataset& d, size_t n_slots, size_t slot_bytes, int short_every, int missing_slot) {
  d.file_bytes.assign(n_slots, 0);
  for (size_t s = 0; s < n_slots; ++s) {
    const std::string name = fill_slot_file_name(d.base, s);
    if ((int)s == missing_slot) { unlink(name.c_str()); d.file_bytes[s] = (size_t)-1; continue; }
    size_t len = slot_bytes;
    if (short_every && s % (size_t)short_every == 1) len = slot_bytes / 2 + (rnd() % 7);     // a file that ends early, not on a cell boundary
    std::vector<uint8_t> v(len);
    for (size_t x = 0; x < len; ++x) v[x] = file_byte(s, x);
    FILE* f = std::fopen(name.c_str(), "wb");
    if (!f || std::fwrite(v.data(), 1, len, f) != len) { std::printf("FAILED: cannot write %s\n", name.c_str()); std::exit(2); }
    std::fclose(f);
    d.file_bytes[s] = len;
  }
}

// the reference's read of one cell of a slot (slot.nim:57-68), from what the files hold
static void reference_cell(const Dataset& d, uint64_t slot, size_t cell_in_slot, size_t cell_size, uint8_t* out) {
  const size_t have = d.file_bytes[slot] == (size_t)-1 ? 0 : d.file_bytes[slot];
  for (size_t b = 0; b < cell_size; ++b) {
    const size_t x = cell_in_slot * cell_size + b;
    out[b] = x < have ? file_byte(slot, x) : 0;
  }
}

static long check_shape(const std::string& dir, size_t n_slots_files, size_t cells_per_slot, size_t cell_size, uint64_t units_per_slot, size_t chunk_bytes,
                        int threads, int ring, bool direct, int short_every, int missing_slot, uint64_t first_unit, size_t n_units, long* bytes) {
  Dataset d;
  d.base = dir + "/s";
  write_files(d, n_slots_files, cells_per_slot * cell_size, short_every, missing_slot);
  IngestGeom g;
  g.n_units = n_units; g.n_cells = cells_per_slot / units_per_slot; g.cell_size = cell_size; g.first_unit = first_unit; g.units_per_slot = units_per_slot;
  const size_t total = g.total_cells();
  const size_t chunk = ingest_chunk_cells(chunk_bytes, cell_size, total);
  const size_t cell_multiple = direct ? [&] { size_t a = cell_size, h = 4096; while (h) { size_t r = a % h; a = h; h = r; } return (size_t)4096 / a; }() : 1;
  auto fail = [&](const char* what, size_t turn, size_t at) {
    std::printf("FAILED: %s (files %zu x %zu cells of %zu B, units/slot %llu, first unit %llu, %zu units, chunk %zu B, threads %d, ring %d, direct %d, short every %d, missing %d: turn %zu, byte %zu)\n",
                what, n_slots_files, cells_per_slot, cell_size, (unsigned long long)units_per_slot, (unsigned long long)first_unit, n_units, chunk_bytes, threads, ring, (int)direct,
                short_every, missing_slot, turn, at);
    std::exit(1);
  };
  // which slot a missing-file report must name: the lowest missing slot among the files a turn touches
  std::vector<std::vector<uint8_t>> bufs((size_t)ring);
  struct Posted { size_t c0, m; int b; };
  std::deque<Posted> posted;
  long turns = 0;
  {
    FillPipeline fill(threads);
    size_t c_next = 0, turn_posted = 0;
    auto post = [&] {
      const size_t m = ingest_turn_cells(g, chunk, cell_multiple, turn_posted, c_next);
      if (m == 0 || m > chunk || c_next + m > total) fail("turn outside the batch or its buffer", turn_posted, 0);
      const int b = (int)(turn_posted % (size_t)ring);
      bufs[(size_t)b].assign(m * cell_size, 0xEE);                  // EXACTLY the turn's size: ASan sees a byte beyond it
      fill.begin(g, d.base, c_next, m, bufs[(size_t)b].data(), direct);
      posted.push_back({c_next, m, b});
      c_next += m;
      ++turn_posted;
    };
    post();
    while (!posted.empty()) {
      if (c_next < total && (int)posted.size() < ring && posted.size() < 2) post();   // two turns deep, like the builder (and never into a buffer still posted)
      const Posted p = posted.front();
      posted.pop_front();
      std::string bad;
      const bool ok = fill.join(&bad);
      // what the reference reads for these cells
      std::vector<uint8_t> want(cell_size);
      bool touches_missing = false;
      uint64_t lowest_missing = ~0ULL;
      for (size_t c = 0; c < p.m; ++c) {
        const size_t cell = p.c0 + c, unit = cell / g.n_cells, in_unit = cell % g.n_cells;
        const uint64_t u = g.first_unit + unit, slot = u / g.units_per_slot;
        const size_t cell_in_slot = (size_t)(u % g.units_per_slot) * g.n_cells + in_unit;
        if (d.file_bytes[slot] == (size_t)-1) { touches_missing = true; if (slot < lowest_missing) lowest_missing = slot; }
        reference_cell(d, slot, cell_in_slot, cell_size, want.data());
        if (std::memcmp(want.data(), bufs[(size_t)p.b].data() + c * cell_size, cell_size) != 0) fail("a cell's bytes differ from the reference's read of the slot file", (size_t)turns, c * cell_size);
      }
      if (ok == touches_missing) fail(ok ? "a turn that touches a missing file was not reported" : "a turn reported a missing file it does not touch", (size_t)turns, 0);
      if (!ok && bad != fill_slot_file_name(d.base, lowest_missing)) fail("the missing file reported is not the one of the lowest slot", (size_t)turns, 0);
      *bytes += (long)(p.m * cell_size);
      ++turns;
    }
    if (!fill.idle()) fail("fills left posted", (size_t)turns, 0);
  }
  for (size_t s = 0; s < n_slots_files; ++s) unlink(fill_slot_file_name(d.base, s).c_str());
  return turns;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: fill_pipeline_check <scratch dir> [shapes]\n"); return 2; }
  const std::string dir = argv[1];
  const long want = argc > 2 ? std::atol(argv[2]) : 300;
  long shapes = 0, turns = 0, bytes = 0, with_missing = 0, with_units = 0, multi_file = 0;
  const size_t cell_sizes[] = {31, 64, 100, 256, 2048, 4096};
  while (shapes < want) {
    const size_t cs = cell_sizes[rnd() % 6];
    const uint64_t ups = (rnd() % 4 == 0) ? (uint64_t)1 << (1 + rnd() % 2) : 1;
    size_t cells_per_slot = ((size_t)1 << (rnd() % 9)) * ups;                  // 1 .. 256 cells per unit
    const size_t n_files = 1 + rnd() % 24;
    const size_t all_units = n_files * ups;
    const uint64_t first_unit = (rnd() % 3 == 0) ? rnd() % all_units : 0;
    const size_t n_units = 1 + rnd() % (all_units - first_unit);
    const size_t data = n_units * (cells_per_slot / ups) * cs;
    // chunk sizes from a fraction of a unit to several files; grains are 4 MiB in the product, so most turns here are ONE grain:
    // every sixth shape is made large enough for several grains per turn (8 ... 40 MiB of data)
    size_t chunk_bytes = std::max<size_t>(cs, data / (1 + rnd() % 9));
    if (shapes % 6 == 5) { cells_per_slot = (((size_t)8 << 20) / cs / ups + 1) * ups; chunk_bytes = (size_t)13 << 20; }
    const int threads = 1 + (int)(rnd() % 8), ring = 2 + (int)(rnd() % 3);
    const bool direct = rnd() % 3 == 0;
    const int short_every = (rnd() % 3 == 0) ? 2 + (int)(rnd() % 3) : 0;
    const int missing = (rnd() % 5 == 0) ? (int)(rnd() % n_files) : -1;
    size_t units_now = n_units, first_now = (size_t)first_unit;
    if (shapes % 6 == 5) { units_now = std::min<size_t>(n_units, 3 * ups); first_now = 0; }
    turns += check_shape(dir, n_files, cells_per_slot, cs, ups, chunk_bytes, threads, ring, direct, short_every, missing, first_now, units_now, &bytes);
    ++shapes;
    with_missing += missing >= 0;
    with_units += ups > 1;
    multi_file += chunk_bytes >= 2 * (cells_per_slot / ups) * cs;
  }
  // the host-array source: memcpy from a caller's array, turns two deep
  {
    const size_t cs = 2048, n = 9000;
    std::vector<uint8_t> src(n * cs);
    for (size_t x = 0; x < src.size(); ++x) src[x] = (uint8_t)(x * 31 + (x >> 11));
    IngestGeom g;
    g.n_units = 1; g.n_cells = n; g.cell_size = cs;
    const size_t chunk = ingest_chunk_cells((size_t)5 << 20, cs, n);
    FillPipeline fill(6);
    std::vector<uint8_t> out(src.size(), 0), b0, b1;
    size_t c0 = 0, turn = 0;
    size_t m = ingest_turn_cells(g, chunk, 1, turn, c0);
    b0.assign(m * cs, 0);
    fill.begin(g, "", c0, m, b0.data(), false, src.data());
    while (c0 < n) {
      const size_t c1 = c0 + m;
      size_t m_next = 0;
      std::vector<uint8_t>& cur = (turn & 1) ? b1 : b0;
      std::vector<uint8_t>& nxt = (turn & 1) ? b0 : b1;
      if (c1 < n) {
        m_next = ingest_turn_cells(g, chunk, 1, turn + 1, c1);
        nxt.assign(m_next * cs, 0);
        fill.begin(g, "", c1, m_next, nxt.data(), false, src.data() + c1 * cs);
      }
      if (!fill.join(nullptr)) { std::printf("FAILED: host-array fill reported a file\n"); return 1; }
      std::memcpy(out.data() + c0 * cs, cur.data(), m * cs);
      c0 = c1; m = m_next; ++turn;
    }
    if (out != src) { std::printf("FAILED: host-array turns do not reproduce the array\n"); return 1; }
    bytes += (long)src.size();
  }
  if (!with_missing || !with_units || !multi_file) { std::printf("FAILED: the walk missed a case (missing %ld, units %ld, multi-file %ld)\n", with_missing, with_units, multi_file); return 1; }
  std::printf("fill pipeline ok: %ld shapes, %ld turns, %ld bytes compared with the reference's reads; %ld shapes with a missing file, %ld cut into units, %ld with turns of several files; host-array source reproduced\n",
              shapes, turns, bytes, with_missing, with_units, multi_file);
  return 0;
}
