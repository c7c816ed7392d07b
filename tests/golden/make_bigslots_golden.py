#!/usr/bin/env python3
"""Generates tests/golden/bigslots.json: ORACLE-computed results for SURVEY.md 8(d)'s OTHER stated scale-down of config 5,
"8 x k slots x 2^22 cells": several slots at the nominal 8 GiB slot size (cellSize 2048, blockSize 65536, nCells 2^22)
sharing one dataset tree, so that the `-m gpu` suite and bench.py can pin the multi-slot / sharded product path at the
full slot size (the only 2^22-cell run before this fixture was one slot).

  nSlots 8, nSamples 100, maxDepth 32, maxLog2NSlots 3, seed 12345, entropy 1234567
  -> the root of every slot, sha256 over all of them, the dataset root, sha256 + length of input.json of EVERY slot
     (every slot is a shard edge for some world size 1..8), and the dataset roots of the 5- and 4-slot prefixes
     (an odd dataset tree and the 2 + 2 split) with input.json of their first and last slots.

Everything is computed by the CPU oracle alone (oracle/p2_oracle.c for the hashing, oracle/poseidon2_ref.py for indexing /
merging / padding / JSON); no GPU and no product code is involved.  Self-derived, KAT-anchored.  Slot 0's
root must equal tests/golden/fullsize.json's config-3 root (same slot), which the script checks.
8 x 1.47e8 permutations: ~45 minutes on 7 threads.

  python tests/golden/make_bigslots_golden.py [threads] [n_slots]
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import c_oracle as C, poseidon2_ref as P  # noqa: E402


def to_int(layers):
    return [C.array_to_felts(l) for l in layers]


def slot_part(c, slot, entropy, threads):
    """Everything of one slot's proof input that depends on the slot alone (gen_input/bn254.nim:53-66): root, sampled
    indices, cells and merged + padded paths.  Only the block roots of the slot are kept (4 MiB); the blocks of the
    sampled cells are regenerated for the bottom five levels of their paths."""
    cs, bs, nc = c["cellSize"], c["blockSize"], c["nCells"]
    cpb = bs // cs
    seed = C.slot_seed(c["seed"], slot)
    broots = C.fake_slot_block_roots(seed, cs, bs, nc, threads)
    big = to_int(C.merkle_tree(broots))
    root = big[-1][0]
    e = C.felt_bytes(entropy)
    idx = [C.cell_index(e, C.felt_bytes(root), nc, k) for k in range(1, c["nSamples"] + 1)]
    inputs = []
    for ci in idx:
        b = ci // cpb
        cells = C.gen_fake_cells(seed, b * cpb, cpb, cs)
        mini = to_int(C.merkle_tree(C.hash_cells(cells, cs)))
        assert mini[-1][0] == big[0][b]
        prf = P.merge_merkle_proofs(P.merkle_proof(mini, ci % cpb), P.merkle_proof(big, b))
        inputs.append({"cellData": cells[ci % cpb].tobytes(), "merkleProof": P.pad_merkle_proof(prf, c["maxDepth"])})
    return {"root": root, "cellIndices": idx, "proofInputs": inputs}


def json_of(c, parts, roots_arr, n_slots, slot, entropy, max_log2):
    dset = to_int(C.merkle_tree(roots_arr[:n_slots]))
    prf = {"dataSetRoot": dset[-1][0], "entropy": entropy, "nCells": c["nCells"], "nSlots": n_slots, "slotIndex": slot,
           "slotRoot": parts[slot]["root"], "slotProof": P.pad_merkle_proof(P.merkle_proof(dset, slot), max_log2),
           "proofInputs": parts[slot]["proofInputs"]}
    text = P.export_json(prf)
    return {"json_sha256": hashlib.sha256(text.encode()).hexdigest(), "json_bytes": len(text)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    threads = int(args[0]) if args else max(1, min(16, len(os.sched_getaffinity(0))))
    n_slots = int(args[1]) if len(args) > 1 else 8
    C.build()
    entropy = 1234567
    c = dict(maxDepth=32, maxLog2NSlots=3, cellSize=2048, blockSize=65536, nSlots=n_slots, nCells=1 << 22, nSamples=100, seed=12345)
    t0 = time.perf_counter()
    parts = []
    for s in range(n_slots):
        parts.append(slot_part(c, s, entropy, threads))
        print("slot %d: root %064x  %.0f s" % (s, parts[-1]["root"], time.perf_counter() - t0), flush=True)
    roots = C.felts_to_array([p["root"] for p in parts])
    try:
        gold3 = json.load(open(os.path.join(HERE, "fullsize.json")))["config3"]["slot_root_hex"]
        assert "%064x" % parts[0]["root"] == gold3, "slot 0 differs from fullsize.json's config-3 root"
    except FileNotFoundError:
        pass
    out = {"note": "oracle-computed (C oracle + Python restatement), self-derived and KAT-anchored; see make_bigslots_golden.py",
           "threads": threads, "entropy": entropy, "config": c,
           "slot_roots_hex": ["%064x" % p["root"] for p in parts],
           "slot_roots_sha256": hashlib.sha256(roots.tobytes()).hexdigest(),
           "cell_indices_first8": [p["cellIndices"][:8] for p in parts],
           "dataset_root_hex": C.merkle_root(roots).tobytes()[::-1].hex(),
           "inputs": {str(s): json_of(c, parts, roots, n_slots, s, entropy, c["maxLog2NSlots"]) for s in range(n_slots)},
           "prefixes": {}}
    for m in (5, 4):       # an odd dataset tree, and the 2 + 2 split
        if m < n_slots:
            out["prefixes"][str(m)] = {"nSlots": m, "dataset_root_hex": C.merkle_root(roots[:m]).tobytes()[::-1].hex(),
                                       "inputs": {str(s): json_of(c, parts, roots, m, s, entropy, c["maxLog2NSlots"]) for s in (0, m - 1)}}
    out["oracle_seconds"] = round(time.perf_counter() - t0, 1)
    with open(os.path.join(HERE, "bigslots.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("dataset root %s  %.1f s" % (out["dataset_root_hex"], out["oracle_seconds"]), flush=True)


if __name__ == "__main__":
    main()
