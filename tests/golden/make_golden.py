#!/usr/bin/env python3
"""Generates the committed fixtures in tests/golden/ from the Python big-int restatement
(oracle/poseidon2_ref.py).  Run from the repo root:  python tests/golden/make_golden.py

Provenance of every file is recorded inside it:
  * kat_permutation.json   -- REFERENCE DATA: the known-answer test committed in
                              reference/haskell/src/Poseidon2/Example.hs:13-19 (input and expected output).
  * everything else        -- SELF-DERIVED, KAT-anchored: the reference defines these input sets
                              (reference/haskell/src/TestVectors.hs:28-75, reference/nim/testvectors/src/testvectors.nim:20-65,
                              workflow/params.sh, reference/haskell/cli/testMain.hs:12-22) but commits no expected values
                              and cannot be run in this image (no nim/ghc/circom).  The values are what the in-tree
                              specification yields; they pin this repo's three implementations to each other, not to a
                              run of the reference.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import poseidon2_ref as P  # noqa: E402

SELF = "self-derived from oracle/poseidon2_ref.py (KAT-anchored); the reference commits no expected values for this set"


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
        f.write("\n")
    print("wrote", name)


def main():
    dump("kat_permutation.json", {
        "provenance": "reference/haskell/src/Poseidon2/Example.hs:13-19 (exInput, exOutput) -- data copied from the reference's committed KAT",
        "input": ["0", "1", "2"],
        "output_hex": ["0x30610a447b7dec194697fb50786aa7421494bd64c221ba4d3b1af25fb07bd103",
                       "0x13f731d6ffbad391be22d2ac364151849e19fa38eced4e761bcd21dbdc600288",
                       "0x1433e2c8f68382c447c5c14b8b3df7cbfd9273dd655fe52f1357c27150da786f"],
    })
    assert [hex(v) for v in P.permutation((0, 1, 2))] == json.load(open(os.path.join(HERE, "kat_permutation.json")))["output_hex"]

    dump("sponge_felts.json", {
        "provenance": SELF, "input_set": "TestVectors.hs:28-44: sponge of [1..n] :: [Fr], n = 0..8",
        "rate1": [str(P.sponge1(list(range(1, n + 1)))) for n in range(9)],
        "rate2": [str(P.sponge2(list(range(1, n + 1)))) for n in range(9)],
    })
    dump("hash_bytes.json", {
        "provenance": SELF, "input_set": "TestVectors.hs:48-57: hash of bytes [1..n], n = 0..80 (10* byte padding, rate 2)",
        "hash": [str(P.hash_bytes(bytes(range(1, n + 1)))) for n in range(81)],
    })
    dump("merkle_roots.json", {
        "provenance": SELF,
        "input_set": "TestVectors.hs:61-75: Merkle root of [1..n] :: [Fr], n = 1..40; Merkle root of the field elements of bytes [1..n], n = 0..80",
        "felts": [str(P.merkle_root(list(range(1, n + 1)))) for n in range(1, 41)],
        "bytes": [str(P.merkle_root(P.bytes_to_felts(bytes(range(1, n + 1))))) for n in range(81)],
    })
    cells = {}
    for (seed, idx, size) in [(12345 + 72, 0, 2048), (12345 + 72 + 3003, 511, 2048), (12345 + 72, 7, 128), (666 + 72, 100, 256)]:
        c = P.gen_fake_cell(seed, idx, size)
        cells["%d/%d/%d" % (seed, idx, size)] = {"first32_hex": c[:32].hex(), "sha256": hashlib.sha256(c).hexdigest(),
                                                 "hashCell": str(P.hash_cell(c, size))}
    dump("fake_cells.json", {"provenance": SELF, "input_set": "genFakeCell(seed, idx, cellSize) keys 'seed/idx/size' (slot.nim:23-32)",
                             "cells": cells})

    configs = {
        # reference/haskell/cli/testMain.hs:12-22 smallDataSetCfg, slot 3, entropy 1234567
        "testmain_small": (dict(maxDepth=16, maxLog2NSlots=5, cellSize=128, blockSize=4096, nSlots=5, nCells=256, nSamples=10, seed=12345), 3, 1234567),
        # workflow/params.sh + workflow/cli_args.sh defaults (--maxslots=256 -> maxLog2NSlots 8)
        "params_default": (dict(maxDepth=32, maxLog2NSlots=8, cellSize=2048, blockSize=65536, nSlots=11, nCells=512, nSamples=5, seed=12345), 3, 1234567),
        # odd slot count, one block per slot (singleton big tree), tiny cells
        "odd_slots_one_block": (dict(maxDepth=8, maxLog2NSlots=4, cellSize=64, blockSize=256, nSlots=3, nCells=4, nSamples=3, seed=42), 2, 99),
    }
    meta = {}
    for name, (cfg, slot, entropy) in configs.items():
        p = P.generate_proof_input(cfg, slot, entropy)
        assert P.circuit_check(p, cfg)
        text = P.export_json(p)
        with open(os.path.join(HERE, "input_%s.json" % name), "w") as f:
            f.write(text)
        meta[name] = {"config": cfg, "slotIndex": slot, "entropy": entropy, "cellIndices": p["cellIndices"],
                      "slotRoot": str(p["slotRoot"]), "dataSetRoot": str(p["dataSetRoot"]),
                      "circom_main": P.circom_main(cfg), "json_sha256": hashlib.sha256(text.encode()).hexdigest()}
        print("wrote input_%s.json" % name)
    dump("proof_inputs.json", {"provenance": SELF + "; input_<name>.json are the exact texts exportProofInput would write (json/bn254.nim:57-74)",
                               "inputs": meta})


if __name__ == "__main__":
    main()
