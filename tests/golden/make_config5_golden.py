#!/usr/bin/env python3
"""Generates tests/golden/config5.json: ORACLE-computed results at BASELINE.json configs[4]'s dataset-tree scale
(32 768 slots, maxLog2NSlots = 15), so that the `-m gpu` suite can pin the sharded product path at that scale.

  cheap   32 768 fake slots x 32 cells of 2048 B (one block per slot), nSamples 100, maxDepth 32, seed 12345,
          entropy 1234567: sha256 over ALL slot roots, the dataset root (15 levels), and sha256 + length of input.json of
          the slots on every shard edge for world sizes 1, 2, 3 and 8 (slotProof of depth 15 with odd and even siblings)
  odd     the same with 32 767 slots: every layer of the dataset tree is odd (keys 2 / 3 at each level, merkle/bn254.nim:47-53)
  scaled  (optional, `--scaled`, hours of CPU) SURVEY.md 8(d)'s stated scale-down of config 5: 32 768 slots x 2^12 cells

Everything is computed by the CPU oracle alone (oracle/p2_oracle.c for the hashing, oracle/poseidon2_ref.py for indexing /
merging / padding / JSON); no GPU and no product code is involved.  Self-derived, KAT-anchored.

  python tests/golden/make_config5_golden.py [threads] [--scaled]
"""
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from oracle import c_oracle as C, poseidon2_ref as P  # noqa: E402
from oracle_helpers import expected_proof_input_fast  # noqa: E402


def shard_edges(n_slots, worlds=(1, 2, 3, 8)):
    """first and last slot of every shard of distributed.shard_range for the given world sizes (restated here: the
    fixture generator imports no product code)."""
    s = set()
    for w in worlds:
        base, rem = divmod(n_slots, w)
        for r in range(w):
            cnt = base + (1 if r < rem else 0)
            first = r * base + min(r, rem)
            if cnt:
                s |= {first, first + cnt - 1}
    return sorted(s)


def all_roots(c, threads, progress=None):
    def one(s):
        return C.fake_slot_root(C.slot_seed(c["seed"], s), c["cellSize"], c["blockSize"], c["nCells"], 1)
    out = np.zeros((c["nSlots"], 32), dtype=np.uint8)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:     # the C oracle releases the GIL (ctypes)
        for s, r in enumerate(ex.map(one, range(c["nSlots"]))):
            out[s] = r
            if progress and s % progress == 0:
                print("  slot %d / %d  %.0f s" % (s, c["nSlots"], time.perf_counter() - t0), flush=True)
    return out


def one_config(c, entropy, threads, slots, progress=None):
    t = time.perf_counter()
    roots = all_roots(c, threads, progress)
    droot = C.merkle_root(roots)
    inputs = {}
    for slot in slots:
        text = P.export_json(expected_proof_input_fast(C, P, c, slot, entropy, threads=threads, slot_roots=roots))
        inputs[str(slot)] = {"json_sha256": hashlib.sha256(text.encode()).hexdigest(), "json_bytes": len(text)}
    return {"config": c, "entropy": entropy, "slot_roots_sha256": hashlib.sha256(roots.tobytes()).hexdigest(),
            "dataset_root_hex": droot.tobytes()[::-1].hex(), "inputs": inputs, "oracle_seconds": round(time.perf_counter() - t, 1)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    threads = int(args[0]) if args else max(1, min(16, len(os.sched_getaffinity(0))))
    C.build()
    path = os.path.join(HERE, "config5.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out["note"] = ("oracle-computed (C oracle + Python restatement), self-derived and KAT-anchored; see make_config5_golden.py")
    entropy = 1234567
    base = dict(maxDepth=32, maxLog2NSlots=15, cellSize=2048, blockSize=65536, nSamples=100, seed=12345)
    if "--scaled" in sys.argv:
        c = dict(base, nSlots=32768, nCells=1 << 12)
        out["scaled"] = one_config(c, entropy, threads, shard_edges(c["nSlots"]), progress=512)
        print("scaled: dataset root %s  %.1f s" % (out["scaled"]["dataset_root_hex"], out["scaled"]["oracle_seconds"]), flush=True)
    else:
        for name, n_slots in (("cheap", 32768), ("odd", 32767)):
            c = dict(base, nSlots=n_slots, nCells=32)
            out[name] = one_config(c, entropy, threads, shard_edges(n_slots))
            print("%s: dataset root %s  %.1f s" % (name, out[name]["dataset_root_hex"], out[name]["oracle_seconds"]), flush=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
