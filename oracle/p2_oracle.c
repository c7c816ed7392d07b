/* CPU ORACLE (test infrastructure, NOT product code).  See p2_oracle.h for the pinning status.
 *
 * Independent of the Python restatement (oracle/poseidon2_ref.py) and of the HIP kernels: this
 * one computes in 4 x 64-bit limbs, Montgomery radix 2^256, with unsigned __int128 products.
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#include "p2_oracle.h"
#include "p2_consts.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fr;

/* r = 21888242871839275222246405745257275088548364400416034343698204186575808495617 (README.md:76) */
static const fr FR_MOD = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
/* -r^-1 mod 2^64 */
static const uint64_t FR_INV = 0xc2e1f593efffffffULL;
/* 2^512 mod r */
static const fr FR_R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};

static int fr_geq(const fr* a, const fr* b) {
  for (int i = 3; i >= 0; --i) {
    if (a->l[i] > b->l[i]) return 1;
    if (a->l[i] < b->l[i]) return 0;
  }
  return 1;
}

static void fr_sub_nocheck(fr* r, const fr* a, const fr* b) {
  uint64_t borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a->l[i] - b->l[i] - borrow;
    r->l[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
}

static void fr_add(fr* r, const fr* a, const fr* b) {
  uint64_t carry = 0;
  fr t;
  for (int i = 0; i < 4; ++i) {
    u128 s = (u128)a->l[i] + b->l[i] + carry;
    t.l[i] = (uint64_t)s;
    carry = (uint64_t)(s >> 64);
  }
  /* a, b < r < 2^254 so no carry out of 256 bits */
  if (fr_geq(&t, &FR_MOD)) fr_sub_nocheck(&t, &t, &FR_MOD);
  *r = t;
}

/* Montgomery product a*b/2^256 mod r (CIOS, 4 limbs) */
static void fr_mul(fr* r, const fr* a, const fr* b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    uint64_t carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 p = (u128)a->l[j] * b->l[i] + t[j] + carry;
      t[j] = (uint64_t)p;
      carry = (uint64_t)(p >> 64);
    }
    u128 s = (u128)t[4] + carry;
    t[4] = (uint64_t)s;
    t[5] = (uint64_t)(s >> 64);
    uint64_t m = t[0] * FR_INV;
    u128 p = (u128)m * FR_MOD.l[0] + t[0];
    carry = (uint64_t)(p >> 64);
    for (int j = 1; j < 4; ++j) {
      p = (u128)m * FR_MOD.l[j] + t[j] + carry;
      t[j - 1] = (uint64_t)p;
      carry = (uint64_t)(p >> 64);
    }
    s = (u128)t[4] + carry;
    t[3] = (uint64_t)s;
    t[4] = t[5] + (uint64_t)(s >> 64);
  }
  fr out = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || fr_geq(&out, &FR_MOD)) fr_sub_nocheck(&out, &out, &FR_MOD);
  *r = out;
}

static void fr_from_bytes(fr* r, const uint8_t b[32]) { /* canonical LE -> Montgomery; reduces values >= r */
  fr t;
  memcpy(t.l, b, 32);
  while (fr_geq(&t, &FR_MOD)) fr_sub_nocheck(&t, &t, &FR_MOD);
  fr_mul(r, &t, &FR_R2);
}

static void fr_to_bytes(uint8_t b[32], const fr* a) {
  static const fr one = {{1, 0, 0, 0}};
  fr t;
  fr_mul(&t, a, &one);
  memcpy(b, t.l, 32);
}

static void fr_from_u64(fr* r, uint64_t v) {
  fr t = {{v, 0, 0, 0}};
  fr_mul(r, &t, &FR_R2);
}

/* ---- constants in Montgomery form, built once -------------------------------------------- */
static fr RC[80];
static fr CIV1, CIV2, KEYS[4], FR_ONE_M, FR_ZERO_M;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void init_consts(void) {
  for (int i = 0; i < 80; ++i) {
    fr t;
    memcpy(t.l, P2O_RC[i], 32);
    fr_mul(&RC[i], &t, &FR_R2);
  }
  /* Sponge.hs:19,34: capacity IV = 2^64 + 256*t + rate */
  fr c1 = {{0x0301, 1, 0, 0}}, c2 = {{0x0302, 1, 0, 0}};
  fr_mul(&CIV1, &c1, &FR_R2);
  fr_mul(&CIV2, &c2, &FR_R2);
  for (uint64_t k = 0; k < 4; ++k) fr_from_u64(&KEYS[k], k);
  fr_from_u64(&FR_ONE_M, 1);
  memset(&FR_ZERO_M, 0, sizeof FR_ZERO_M);
}
static void ensure_init(void) { pthread_once(&g_once, init_consts); }

/* ---- a1: permutation (Permutation.hs:14-45) ------------------------------------------------ */
static void sbox(fr* x) { /* Permutation.hs:14-17 */
  fr x2, x4;
  fr_mul(&x2, x, x);
  fr_mul(&x4, &x2, &x2);
  fr_mul(x, &x4, x);
}

static void external_round(fr st[3], const fr c[3]) { /* Permutation.hs:28-33 */
  fr s;
  for (int i = 0; i < 3; ++i) {
    fr_add(&st[i], &st[i], &c[i]);
    sbox(&st[i]);
  }
  fr_add(&s, &st[0], &st[1]);
  fr_add(&s, &s, &st[2]);
  for (int i = 0; i < 3; ++i) fr_add(&st[i], &st[i], &s);
}

static void internal_round(fr st[3], const fr* c) { /* Permutation.hs:19-26 */
  fr s;
  fr_add(&st[0], &st[0], c);
  sbox(&st[0]);
  fr_add(&s, &st[0], &st[1]);
  fr_add(&s, &s, &st[2]);
  fr_add(&st[0], &st[0], &s);           /* 2x' + y + z  */
  fr_add(&st[1], &st[1], &s);           /* x' + 2y + z  */
  fr_add(&st[2], &st[2], &st[2]);
  fr_add(&st[2], &st[2], &s);           /* x' + y + 3z  */
}

static void permute_m(fr st[3]) { /* Permutation.hs:40-45, state in Montgomery form */
  fr s;
  fr_add(&s, &st[0], &st[1]);           /* linearLayer, :35-36 */
  fr_add(&s, &s, &st[2]);
  for (int i = 0; i < 3; ++i) fr_add(&st[i], &st[i], &s);
  for (int r = 0; r < 4; ++r) external_round(st, &RC[3 * r]);
  for (int r = 0; r < 56; ++r) internal_round(st, &RC[12 + r]);
  for (int r = 0; r < 4; ++r) external_round(st, &RC[68 + 3 * r]);
}

void p2o_permute(const uint8_t in[96], uint8_t out[96]) {
  ensure_init();
  fr st[3];
  for (int i = 0; i < 3; ++i) fr_from_bytes(&st[i], in + 32 * i);
  permute_m(st);
  for (int i = 0; i < 3; ++i) fr_to_bytes(out + 32 * i, &st[i]);
}

void p2o_permute_batch(const uint8_t* in, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i) p2o_permute(in + 96 * i, out + 96 * i);
}

/* ---- generic range splitter over pthreads --------------------------------------------------- */
typedef void (*range_fn)(void* ctx, size_t lo, size_t hi);
typedef struct { range_fn fn; void* ctx; size_t lo, hi; } range_job;
static void* range_thread(void* p) {
  range_job* j = (range_job*)p;
  j->fn(j->ctx, j->lo, j->hi);
  return NULL;
}
static void run_ranges(range_fn fn, void* ctx, size_t n, int threads) {
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  if (threads == 1) { fn(ctx, 0, n); return; }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  range_job* jobs = (range_job*)malloc(sizeof(range_job) * threads);
  for (int t = 0; t < threads; ++t) {
    jobs[t].fn = fn; jobs[t].ctx = ctx;
    jobs[t].lo = n * (size_t)t / threads;
    jobs[t].hi = n * (size_t)(t + 1) / threads;
    pthread_create(&th[t], NULL, range_thread, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  free(th); free(jobs);
}

typedef struct { const uint8_t* in; uint8_t* out; } perm_ctx;
static void perm_range(void* c, size_t lo, size_t hi) {
  perm_ctx* p = (perm_ctx*)c;
  p2o_permute_batch(p->in + 96 * lo, p->out + 96 * lo, hi - lo);
}
void p2o_permute_batch_mt(const uint8_t* in, uint8_t* out, size_t n, int threads) {
  ensure_init();
  perm_ctx c = {in, out};
  run_ranges(perm_range, &c, n, threads);
}

/* ---- a6: keyed compression (Merkle.hs:202-203) --------------------------------------------- */
static void compress_m(fr* out, const fr* x, const fr* y, uint32_t key) {
  fr st[3] = {*x, *y, KEYS[key & 3]};
  permute_m(st);
  *out = st[0];
}

void p2o_compress(const uint8_t x[32], const uint8_t y[32], uint32_t key, uint8_t out[32]) {
  ensure_init();
  fr a, b, r;
  fr_from_bytes(&a, x);
  fr_from_bytes(&b, y);
  compress_m(&r, &a, &b, key);
  fr_to_bytes(out, &r);
}

/* ---- a3: sponges (Sponge.hs:14-43) ----------------------------------------------------------- */
typedef struct { fr st[3]; int have; fr pending; } sponge2_t;

static void sponge2_init(sponge2_t* s) {
  s->st[0] = FR_ZERO_M; s->st[1] = FR_ZERO_M; s->st[2] = CIV2; s->have = 0;
}
static void sponge2_absorb(sponge2_t* s, const fr* a) {   /* Sponge.hs:41-43 */
  if (!s->have) { s->pending = *a; s->have = 1; return; }
  fr_add(&s->st[0], &s->st[0], &s->pending);
  fr_add(&s->st[1], &s->st[1], a);
  permute_m(s->st);
  s->have = 0;
}
static void sponge2_finish(sponge2_t* s, fr* out) {       /* Sponge.hs:36-39: pad 1, then 0 to even */
  if (s->have) {
    sponge2_absorb(s, &FR_ONE_M);
  } else {
    sponge2_absorb(s, &FR_ONE_M);
    sponge2_absorb(s, &FR_ZERO_M);
  }
  *out = s->st[0];
}

void p2o_sponge2_felts(const uint8_t* felts, size_t n, uint8_t out[32]) {
  ensure_init();
  sponge2_t s; sponge2_init(&s);
  for (size_t i = 0; i < n; ++i) { fr a; fr_from_bytes(&a, felts + 32 * i); sponge2_absorb(&s, &a); }
  fr h; sponge2_finish(&s, &h);
  fr_to_bytes(out, &h);
}

void p2o_sponge1_felts(const uint8_t* felts, size_t n, uint8_t out[32]) { /* Sponge.hs:14-27 */
  ensure_init();
  fr st[3] = {FR_ZERO_M, FR_ZERO_M, CIV1};
  for (size_t i = 0; i <= n; ++i) {
    fr a;
    if (i < n) fr_from_bytes(&a, felts + 32 * i); else a = FR_ONE_M;
    fr_add(&st[0], &st[0], &a);
    permute_m(st);
  }
  fr_to_bytes(out, &st[0]);
}

/* ---- a4: bytes -> field elements (Slot.hs:243-270) ------------------------------------------ */
size_t p2o_felts_per_bytes(size_t len) { return (len + 1 + 30) / 31; }

static void chunk_at(const uint8_t* data, size_t len, size_t k, uint8_t out32[32]) {
  /* k-th 31-byte chunk of data || 0x01 || 0x00..., little-endian integer in 32 bytes */
  memset(out32, 0, 32);
  for (size_t i = 0; i < 31; ++i) {
    size_t pos = 31 * k + i;
    if (pos < len) out32[i] = data[pos];
    else if (pos == len) out32[i] = 0x01;
  }
}

void p2o_bytes_to_felts(const uint8_t* data, size_t len, uint8_t* out) {
  size_t n = p2o_felts_per_bytes(len);
  for (size_t k = 0; k < n; ++k) chunk_at(data, len, k, out + 32 * k);
}

static void hash_bytes_m(const uint8_t* data, size_t len, fr* out) {
  sponge2_t s; sponge2_init(&s);
  size_t n = p2o_felts_per_bytes(len);
  for (size_t k = 0; k < n; ++k) {
    uint8_t c[32]; fr a;
    chunk_at(data, len, k, c);
    fr_from_bytes(&a, c);
    sponge2_absorb(&s, &a);
  }
  sponge2_finish(&s, out);
}

void p2o_hash_bytes(const uint8_t* data, size_t len, uint8_t out[32]) {
  ensure_init();
  fr h; hash_bytes_m(data, len, &h);
  fr_to_bytes(out, &h);
}

/* ---- a5: hashCell over an array of cells (blocks/bn254.nim:23-29) --------------------------- */
typedef struct { const uint8_t* cells; size_t cell_size; uint8_t* out; } cells_ctx;
static void cells_range(void* c, size_t lo, size_t hi) {
  cells_ctx* p = (cells_ctx*)c;
  for (size_t i = lo; i < hi; ++i) p2o_hash_bytes(p->cells + i * p->cell_size, p->cell_size, p->out + 32 * i);
}
void p2o_hash_cells_mt(const uint8_t* cells, size_t cell_size, size_t n_cells, uint8_t* out, int threads) {
  ensure_init();
  cells_ctx c = {cells, cell_size, out};
  run_ranges(cells_range, &c, n_cells, threads);
}
void p2o_hash_cells(const uint8_t* cells, size_t cell_size, size_t n_cells, uint8_t* out) {
  p2o_hash_cells_mt(cells, cell_size, n_cells, out, 1);
}

/* ---- a7: Merkle tree (merkle/bn254.nim:24-63) ------------------------------------------------ */
size_t p2o_merkle_total(size_t n) {
  if (n == 0) return 0;
  size_t total = n, m = n;
  int bottom = 1;
  while (!(m == 1 && !bottom)) { m = (m + 1) / 2; total += m; bottom = 0; }
  return total;
}

/* layers in Montgomery form; returns number of layers */
static size_t merkle_tree_m(const fr* leaves, size_t n, fr* layers, size_t* layer_sizes) {
  size_t nl = 0, off = 0, m = n;
  int bottom = 1;
  memcpy(layers, leaves, n * sizeof(fr));
  for (;;) {
    if (layer_sizes) layer_sizes[nl] = m;
    nl++;
    if (m == 1 && !bottom) break;                      /* bn254.nim:34-36 */
    const fr* xs = layers + off;
    fr* ys = layers + off + m;
    size_t half = m / 2;
    for (size_t i = 0; i < half; ++i) compress_m(&ys[i], &xs[2 * i], &xs[2 * i + 1], bottom ? 1 : 0);   /* :47-50 */
    if (m & 1) compress_m(&ys[half], &xs[m - 1], &FR_ZERO_M, bottom ? 3 : 2);                           /* :51-53 */
    off += m;
    m = (m + 1) / 2;
    bottom = 0;
  }
  return nl;
}

size_t p2o_merkle_tree(const uint8_t* leaves, size_t n, uint8_t* layers_out, size_t* layer_sizes) {
  ensure_init();
  if (n == 0) return 0;
  size_t total = p2o_merkle_total(n);
  fr* lv = (fr*)malloc(sizeof(fr) * n);
  fr* ly = (fr*)malloc(sizeof(fr) * total);
  for (size_t i = 0; i < n; ++i) fr_from_bytes(&lv[i], leaves + 32 * i);
  size_t nl = merkle_tree_m(lv, n, ly, layer_sizes);
  for (size_t i = 0; i < total; ++i) fr_to_bytes(layers_out + 32 * i, &ly[i]);
  free(lv); free(ly);
  return nl;
}

void p2o_merkle_root(const uint8_t* leaves, size_t n, uint8_t out[32]) {
  size_t total = p2o_merkle_total(n);
  uint8_t* ly = (uint8_t*)malloc(32 * total);
  p2o_merkle_tree(leaves, n, ly, NULL);
  memcpy(out, ly + 32 * (total - 1), 32);
  free(ly);
}

/* ---- a10: fake data (slot.nim:22-32, dataset.nim:32) ---------------------------------------- */
void p2o_gen_fake_cell(uint64_t seed, uint64_t idx, size_t cell_size, uint8_t* out) {
  uint64_t seed1 = seed + 0xdeadcafeULL;
  uint64_t seed2 = idx + 0x98765432ULL;
  uint64_t state = 1;
  for (size_t i = 0; i < cell_size; ++i) {
    state = state * (state + seed1) * (state + seed2) + state * (state ^ 0x5a5a5a5aULL) + seed1 * state + (seed2 + 17);
    state = state % 1698428844001831ULL;
    out[i] = (uint8_t)state;
  }
}

uint64_t p2o_slot_seed(uint64_t seed, uint64_t slot_idx) { return seed + 72 + 1001 * slot_idx; }

/* ---- a9: slot root of a fake-data slot (gen_input/bn254.nim:21-30) -------------------------- */
typedef struct { uint64_t seed; size_t cell_size, cpb; fr* block_roots; } slot_ctx;
static void slot_block_range(void* c, size_t lo, size_t hi) {
  slot_ctx* p = (slot_ctx*)c;
  uint8_t* cell = (uint8_t*)malloc(p->cell_size);
  fr* leaves = (fr*)malloc(sizeof(fr) * p->cpb);
  fr* layers = (fr*)malloc(sizeof(fr) * p2o_merkle_total(p->cpb));
  for (size_t b = lo; b < hi; ++b) {
    for (size_t i = 0; i < p->cpb; ++i) {
      p2o_gen_fake_cell(p->seed, b * p->cpb + i, p->cell_size, cell);
      hash_bytes_m(cell, p->cell_size, &leaves[i]);
    }
    size_t total = p2o_merkle_total(p->cpb);
    merkle_tree_m(leaves, p->cpb, layers, NULL);         /* blocks/bn254.nim:60-67 */
    p->block_roots[b] = layers[total - 1];
  }
  free(cell); free(leaves); free(layers);
}

/* the block roots of a fake-data slot (blocks/bn254.nim:60-67 per block): n_cells / cellsPerBlock x 32 bytes.  What a
 * full-size fixture generator needs beside the root: the tree over them gives the top part of every Merkle path, and only
 * the sampled cells' own blocks have to be regenerated for the bottom part. */
void p2o_fake_slot_block_roots(uint64_t slot_seed, size_t cell_size, size_t block_size, size_t n_cells,
                               uint8_t* out, int threads) {
  ensure_init();
  size_t cpb = block_size / cell_size;
  size_t nblocks = n_cells / cpb;
  fr* roots = (fr*)malloc(sizeof(fr) * nblocks);
  slot_ctx c = {slot_seed, cell_size, cpb, roots};
  run_ranges(slot_block_range, &c, nblocks, threads);
  for (size_t b = 0; b < nblocks; ++b) fr_to_bytes(out + 32 * b, &roots[b]);
  free(roots);
}

void p2o_fake_slot_root(uint64_t slot_seed, size_t cell_size, size_t block_size, size_t n_cells,
                        uint8_t out[32], int threads) {
  ensure_init();
  size_t cpb = block_size / cell_size;
  size_t nblocks = n_cells / cpb;
  fr* roots = (fr*)malloc(sizeof(fr) * nblocks);
  slot_ctx c = {slot_seed, cell_size, cpb, roots};
  run_ranges(slot_block_range, &c, nblocks, threads);
  size_t total = p2o_merkle_total(nblocks);
  fr* layers = (fr*)malloc(sizeof(fr) * total);
  merkle_tree_m(roots, nblocks, layers, NULL);           /* gen_input/bn254.nim:28-29 */
  fr_to_bytes(out, &layers[total - 1]);
  free(roots); free(layers);
}

/* ---- a12: sampling (sample/bn254.nim:16-24, types/bn254.nim:47-59) -------------------------- */
uint64_t p2o_cell_index(const uint8_t entropy[32], const uint8_t slot_root[32], uint64_t n_cells, uint64_t counter) {
  ensure_init();
  uint8_t felts[96], h[32];
  memcpy(felts, entropy, 32);
  memcpy(felts + 32, slot_root, 32);
  memset(felts + 64, 0, 32);
  memcpy(felts + 64, &counter, 8);
  p2o_sponge2_felts(felts, 3, h);
  uint64_t lo;
  memcpy(&lo, h, 8);
  return lo & (n_cells - 1);                            /* n_cells is a power of two (:19-20) */
}
