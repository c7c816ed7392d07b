#!/usr/bin/env python3
"""Generates tests/golden/fullsize.json: ORACLE-computed results at BASELINE.json's full sizes, so that the `-m gpu`
suite can compare the HIP path with them in a fraction of a second instead of re-running minutes of CPU hashing.

  config 3  one fake slot, cellSize 2048, nCells 2^22 (8 GiB), seed 12345, slot 0  -> slot root
  config 4  4096 fake slots x 2^12 cells, nSamples 100, maxDepth 32, seed 12345, entropy 1234567
            -> sha256 over all 4096 slot roots, dataset root, sha256 + length of input.json of slots 1234 and 4095

Everything is computed by the CPU oracle alone (oracle/p2_oracle.c for the hashing, oracle/poseidon2_ref.py for
indexing / merging / padding / JSON); no GPU and no product code is involved.  Self-derived, KAT-anchored fixtures
(the reference commits no values above the permutation: SURVEY.md 8c).  About 5.9e8 + 1.5e8 permutations:
~15 minutes on 8 threads.

  python tests/golden/make_fullsize_golden.py [threads]
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from oracle import c_oracle as C, poseidon2_ref as P  # noqa: E402
from oracle_helpers import expected_proof_input_fast  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, min(16, len(os.sched_getaffinity(0))))
    C.build()
    out = {"note": "oracle-computed (C oracle + Python restatement), self-derived and KAT-anchored; see make_fullsize_golden.py",
           "threads": threads}

    t = time.perf_counter()
    root3 = C.fake_slot_root(C.slot_seed(12345, 0), 2048, 65536, 1 << 22, threads)
    out["config3"] = {"seed": 12345, "slot": 0, "cellSize": 2048, "blockSize": 65536, "nCells": 1 << 22,
                      "slot_root_hex": root3.tobytes()[::-1].hex(), "oracle_seconds": round(time.perf_counter() - t, 1)}
    print("config 3: %s  %.1f s" % (out["config3"]["slot_root_hex"], out["config3"]["oracle_seconds"]), flush=True)

    c = dict(maxDepth=32, maxLog2NSlots=12, cellSize=2048, blockSize=65536, nSlots=4096, nCells=1 << 12, nSamples=100, seed=12345)
    entropy = 1234567
    t = time.perf_counter()
    roots = np.stack([C.fake_slot_root(C.slot_seed(c["seed"], s), c["cellSize"], c["blockSize"], c["nCells"], threads)
                      for s in range(c["nSlots"])])
    droot = C.merkle_root(roots)
    inputs = {}
    for slot in (1234, 4095):
        text = P.export_json(expected_proof_input_fast(C, P, c, slot, entropy, threads=threads, slot_roots=roots))
        inputs[str(slot)] = {"json_sha256": hashlib.sha256(text.encode()).hexdigest(), "json_bytes": len(text)}
    out["config4"] = {"config": c, "entropy": entropy, "slot_roots_sha256": hashlib.sha256(roots.tobytes()).hexdigest(),
                      "dataset_root_hex": droot.tobytes()[::-1].hex(), "inputs": inputs,
                      "oracle_seconds": round(time.perf_counter() - t, 1)}
    print("config 4: dataset root %s  %.1f s" % (out["config4"]["dataset_root_hex"], out["config4"]["oracle_seconds"]), flush=True)
    with open(os.path.join(HERE, "fullsize.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
