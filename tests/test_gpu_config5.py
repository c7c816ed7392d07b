"""GPU suite, round 3 (-m gpu): BASELINE.json configs[4] at its own dataset-tree scale -- 32 768 slots, maxLog2NSlots = 15
(reference/nim/proof_input/src/gen_input/bn254.nim:41-51,72) -- through the sharded PRODUCT path
(distributed.dataset_root_sharded + HipBackend -> libcodex_p2.so), in fresh rank processes and in this process.

  * cheap geometry (32 cells per slot): everything against tests/golden/config5.json, computed by the oracle alone
    (make_config5_golden.py): sha256 over all 32 768 slot roots, the dataset root, input.json on every shard edge.
  * SURVEY.md 8(d)'s stated scale-down of config 5 (32 768 slots x 2^12 cells, 256 GiB of fake data generated and hashed on
    the device, ~6 s): random slot roots against the C oracle, dataset tree and slotProof (depth 15, odd and even siblings)
    against the oracle over the gathered roots, input.json byte for byte, a streamed shard with spilled bodies; and the
    oracle-only fixture of the same shape (config5.json "scaled": 3.4 hours of the C oracle): sha256 over all 32 768 slot roots,
    the dataset root and input.json on the shard edges."""
import hashlib
import importlib
import os

import numpy as np
import pytest

from oracle_helpers import expected_proof_input_fast
from rank_helpers import run_ranks

pytestmark = pytest.mark.gpu


def hexroot(a):
    return np.asarray(a, dtype=np.uint8).tobytes()[::-1].hex()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name,world,gather", [("cheap", 1, "dev"), ("cheap", 2, "dev"), ("cheap", 3, "dev"), ("odd", 2, "dev"), ("odd", 3, "host")])
def test_config5_dataset_scale_sharded_ranks_vs_oracle_fixture(golden, tmp_path, name, world, gather):
    """32 768 (and 32 767: every dataset-tree layer odd) slots over 1, 2 and 3 rank processes (3: 10 923 + 10 923 + 10 922).
    "dev": the slot roots never leave HBM on either side of the exchange (cp2_dataset_copy_local_roots_dev ->
    collective -> cp2_dataset_set_roots_dev); "host": host arrays on both sides (cp2_dataset_local_roots / _set_roots)."""
    g = golden("config5.json")[name]
    c, n = g["config"], g["config"]["nSlots"]
    res = run_ranks(world, c, g["entropy"], tmp_path, gather=gather)
    covered, checked = [], 0
    for r in res:
        assert r["native_so_loaded"] and not r["oracle_loaded"]            # the product path, not the checker
        assert r["dataset_root_hex"] == g["dataset_root_hex"]
        assert r["all_roots_sha256"] == g["slot_roots_sha256"]
        covered += list(range(r["first"], r["first"] + r["count"]))
        for slot, digest in r["inputs"].items():
            assert digest == g["inputs"][slot]["json_sha256"], (name, world, r["rank"], slot)
            checked += 1
    assert covered == list(range(n))
    assert checked == 2 * world
    counts = [r["count"] for r in sorted(res, key=lambda r: r["rank"])]
    assert counts == {1: [n], 2: [(n + 1) // 2, n // 2], 3: [n // 3 + (1 if n % 3 > 0 else 0), n // 3 + (1 if n % 3 > 1 else 0), n // 3]}[world]


SCALED = dict(maxDepth=32, maxLog2NSlots=15, cellSize=2048, blockSize=65536, nSlots=32768, nCells=1 << 12, nSamples=100, seed=12345)
ENTROPY = 1234567


def _threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def _check_roots_against_oracle(C, c, all_roots, slots):
    for s in slots:
        want = C.fake_slot_root(C.slot_seed(c["seed"], int(s)), c["cellSize"], c["blockSize"], c["nCells"], _threads())
        assert np.array_equal(all_roots[int(s)], want), s


def test_config5_scaled_world1_in_process(pkg, ctx, oracle, golden, entry, tmp_path):
    """World 1 through distributed.dataset_root_sharded(HipBackend) in this process: 4.7e9 permutations."""
    C, P = oracle
    d = importlib.import_module(entry.PKG_NAME + ".distributed")
    c = SCALED
    cfg = pkg.make_config(**c)
    backend = d.HipBackend(pkg, ctx)
    root, all_roots, (first, count) = d.dataset_root_sharded(backend, cfg, 0, 1, None, "cuda:0")
    assert (first, count) == (0, c["nSlots"]) and all_roots.shape == (c["nSlots"], 32)
    rng = np.random.default_rng(5)
    _check_roots_against_oracle(C, c, all_roots, [0, 16383, 16384, 32767] + list(rng.integers(0, c["nSlots"], size=60)))
    # the dataset tree (15 levels, 32 767 compressions) and cp2_merkle_tree at n = 32 768, against the oracle over the same roots
    assert np.array_equal(root, C.merkle_root(all_roots))
    layers, want_layers = ctx.merkle_tree(all_roots), C.merkle_tree(all_roots)
    assert len(layers) == 16 and all(np.array_equal(a, b) for a, b in zip(layers, want_layers))
    # input.json: slotProof of depth 15 with all-even (0), all-odd (32767), alternating (21845 = 0b101010101010101) and mixed siblings
    texts = {}
    for slot in (0, 16384, 21845, 32767):
        want = P.export_json(expected_proof_input_fast(C, P, c, slot, ENTROPY, threads=_threads(), slot_roots=all_roots))
        texts[slot] = backend.dataset.proof_input(slot, ENTROPY).json()
        assert texts[slot] == want, slot
    fix = golden("config5.json").get("scaled")
    if fix:     # the oracle-only fixture of this very shape (make_config5_golden.py --scaled)
        assert fix["config"] == c and fix["entropy"] == ENTROPY
        assert sha(all_roots) == fix["slot_roots_sha256"] and hexroot(root) == fix["dataset_root_hex"]
        for slot in (0, 16384, 32767):
            assert hashlib.sha256(texts[slot].encode()).hexdigest() == fix["inputs"][str(slot)]["json_sha256"]
    # a streamed shard of 1024 slots around the middle, bodies bounded to 64 MiB of host memory: the rest spills to files
    spill = tmp_path / "spill"
    out = tmp_path / "out"
    spill.mkdir()
    out.mkdir()
    s0, sn = 15872, 1024
    ctx.set_body_budget(64 << 20, str(spill))
    try:
        sd = ctx.dataset_streamed(cfg, ENTROPY, s0, sn, threads=_threads())
        (private,) = os.listdir(spill)                                 # one mkdtemp directory, mode 0700
        n_parts = len([f for f in os.listdir(spill / private) if f.endswith(".part")])
        assert 850 < n_parts < 1000                                    # 0.7 MB each: about 90 stay resident
        assert np.array_equal(sd.local_roots(), all_roots[s0:s0 + sn])
        sd.set_roots(all_roots)
        total = sd.export_streamed(str(out), threads=_threads())
        assert sd.streamed_json(s0) == backend.dataset.proof_input(s0, ENTROPY).json()           # resident body
        assert sd.streamed_json(16384) == texts[16384]                                           # spilled body
        assert open(out / "input_16384.json").read() == texts[16384]
        assert open(out / ("input_%d.json" % (s0 + sn - 1))).read() == backend.dataset.proof_input(s0 + sn - 1, ENTROPY).json()
        assert total == sum(os.path.getsize(out / f) for f in os.listdir(out)) and len(os.listdir(out)) == sn
        sd.free()
        assert not os.listdir(spill)                                   # spill files go with the dataset
    finally:
        ctx.set_body_budget(4 << 30, None)
        for f in os.listdir(out):
            os.remove(out / f)
    backend.dataset.free()
    ctx.trim()


def test_config5_scaled_two_rank_processes(oracle, golden, tmp_path):
    """The same shape as two gloo rank processes sharing GPU 0 (16 384 slots each), gathered roots saved by rank 0."""
    C, P = oracle
    c = SCALED
    roots_path = tmp_path / "roots.npy"
    res = run_ranks(2, c, ENTROPY, tmp_path, roots_path=roots_path, timeout=1100)
    all_roots = np.load(roots_path)
    assert all_roots.shape == (c["nSlots"], 32)
    rng = np.random.default_rng(6)
    _check_roots_against_oracle(C, c, all_roots, [16383, 16384] + list(rng.integers(0, c["nSlots"], size=14)))
    want_root = hexroot(C.merkle_root(all_roots))
    fix = golden("config5.json").get("scaled")
    for r in res:
        assert r["native_so_loaded"] and not r["oracle_loaded"]
        assert r["dataset_root_hex"] == want_root and r["all_roots_sha256"] == sha(all_roots)
        assert r["count"] == 16384
        for slot, digest in r["inputs"].items():
            text = P.export_json(expected_proof_input_fast(C, P, c, int(slot), ENTROPY, threads=_threads(), slot_roots=all_roots))
            assert hashlib.sha256(text.encode()).hexdigest() == digest, (r["rank"], slot)
            if fix:
                assert fix["inputs"][slot]["json_sha256"] == digest
    if fix:
        assert fix["slot_roots_sha256"] == sha(all_roots) and fix["dataset_root_hex"] == want_root
