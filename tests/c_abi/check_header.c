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
  printf("c abi ok\n");
  return 0;
}
