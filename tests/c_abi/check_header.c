/* The ABI header must be plain C (C99): this file is compiled with gcc -std=c99 -pedantic -Wall -Werror, and
 * linked against libcodex_p2.so; it calls only host-side entry points (no GPU needed). */
#include <stdio.h>
#include <string.h>

#include "../../include/codex_p2.h"

int main(void) {
  uint8_t data[62], felts[3 * 32];
  cp2_config cfg;
  size_t i;
  for (i = 0; i < sizeof data; ++i) data[i] = (uint8_t)(i + 1);
  if (cp2_felts_per_bytes(sizeof data) != 3) return 1;
  if (cp2_bytes_to_felts(data, sizeof data, felts) != CP2_OK) return 2;
  if (felts[0] != 1 || felts[30] != 31 || felts[31] != 0 || felts[32] != 32 || felts[64] != 0x01) return 3;
  if (cp2_merkle_total(5) != 5 + 3 + 2 + 1 || cp2_merkle_num_layers(1) != 2) return 4;
  if (cp2_slot_seed(12345, 3) != 12345 + 72 + 3003) return 5;
  if (strcmp(cp2_strerror(CP2_ERR_NO_DEVICE), "no usable gfx950 HIP device") != 0) return 6;
  memset(&cfg, 0, sizeof cfg);
  if (cp2_write_circom_main(&cfg, "/nonexistent/x") == CP2_OK) return 7;
  if (cp2_permute_batch(NULL, data, data, 1) != CP2_ERR_INVALID) return 8;
  /* context-level knobs refuse a missing context instead of dereferencing it */
  if (cp2_trim(NULL) != CP2_ERR_INVALID || cp2_set_body_budget(NULL, 1, "/tmp") != CP2_ERR_INVALID ||
      cp2_set_ingest_direct(NULL, 1) != CP2_ERR_INVALID || cp2_set_ingest(NULL, 1, 2, 3) != CP2_ERR_INVALID) return 9;
  if (cp2_slot_trees_load(NULL, "/nonexistent", NULL) != CP2_ERR_INVALID || cp2_slot_trees_save(NULL, "/tmp/x") != CP2_ERR_INVALID) return 10;
  if (cp2_dataset_export_streamed(NULL, NULL, 1, NULL) != CP2_ERR_INVALID || cp2_dataset_streamed_json(NULL, 0, NULL, NULL) != CP2_ERR_INVALID) return 11;
  /* section e (several GPUs behind one handle): the split rule is plain arithmetic; handles refuse NULL */
  {
    uint64_t first = 99, count = 99, covered = 0;
    int r;
    for (r = 0; r < 3; ++r) {
      cp2_shard_range(32767, r, 3, &first, &count);
      if (first != covered || count != (r < 1 ? 10923u : 10922u)) return 12;
      covered += count;
    }
    if (covered != 32767) return 13;
    cp2_shard_range(5, 7, 3, &first, &count);
    if (first != 0 || count != 0) return 14;
  }
  if (cp2_multi_init(NULL, -1, NULL) != CP2_ERR_INVALID || cp2_multi_count(NULL) != 0 || cp2_multi_ctx(NULL, 0) != NULL ||
      cp2_multi_set_policy(NULL, CP2_GATHER_AUTO, 0) != CP2_ERR_INVALID || cp2_multi_device(NULL, 0) != -1) return 15;
  if (cp2_multi_dataset_build(NULL, &cfg, NULL) != CP2_ERR_INVALID || cp2_multi_dataset_shards(NULL) != 0 ||
      cp2_multi_proof_input_generate(NULL, 0, data, NULL) != CP2_ERR_INVALID ||
      cp2_multi_dataset_export_streamed(NULL, NULL, 1, NULL) != CP2_ERR_INVALID) return 16;
  if (cp2_dataset_set_roots_dev(NULL, NULL) != CP2_ERR_INVALID || cp2_dataset_copy_local_roots_dev(NULL, NULL) != CP2_ERR_INVALID ||
      cp2_dataset_local_roots_dev(NULL) != NULL || cp2_dataset_ctx(NULL) != NULL || cp2_dataset_range(NULL, NULL, NULL) != CP2_ERR_INVALID) return 17;
  if (cp2_set_keep_trees(NULL, 0) != CP2_ERR_INVALID || cp2_dataset_keeps_trees(NULL) != 0 || cp2_multi_set_split(NULL, 2) != CP2_ERR_INVALID ||
      cp2_multi_dataset_units_per_slot(NULL) != 0 || cp2_slot_trees_build_fake_units(NULL, 1, 2, 0, 1, 64, 128, 4, NULL) != CP2_ERR_INVALID) return 18;
  {
    int shards = 0;
    uint64_t units = 0;
    cfg.cell_size = 2048; cfg.block_size = 65536; cfg.n_slots = 11; cfg.n_cells = 1u << 22;
    if (cp2_multi_plan(&cfg, 8, 0, 0, &shards, &units) != CP2_OK || shards != 8 || units != 8) return 19;   /* 88 units, 11 per device */
    if (cp2_multi_plan(NULL, 8, 0, 0, &shards, &units) != CP2_ERR_INVALID) return 20;
  }
  printf("c abi ok\n");
  return 0;
}
