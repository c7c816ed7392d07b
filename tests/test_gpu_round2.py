"""GPU suite, round 2 additions (-m gpu): full-size configurations against oracle-computed fixtures, the sharded
(multi-rank) product path on one GPU, the streamed build+export pipeline, slot files against the oracle directly,
cache validation and ingestion knobs.  Everything goes through the C ABI (ctypes) of libcodex_p2.so."""
import hashlib
import os

import numpy as np
import pytest

from oracle_helpers import expected_proof_input_fast
from rank_helpers import run_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hexroot(a):
    return np.asarray(a, dtype=np.uint8).tobytes()[::-1].hex()


# ---- BASELINE.json configs[2] and configs[3] at FULL size vs tests/golden/fullsize.json -----------------------
def test_fullsize_config3_slot_root(pkg, ctx, golden):
    """8 GiB slot (cellSize 2048, nCells 2^22, seed 12345, slot 0): 146 800 639 permutations on the GPU, root equal
    to the oracle's (computed once by tests/golden/make_fullsize_golden.py, minutes of CPU)."""
    g3 = golden("fullsize.json")["config3"]
    trees = ctx.slot_trees_fake(g3["seed"], g3["slot"], 1, g3["cellSize"], g3["blockSize"], g3["nCells"])
    assert hexroot(trees.roots()[0]) == g3["slot_root_hex"]
    trees.free()


def test_fullsize_config4_dataset_and_proof_inputs(pkg, ctx, golden, tmp_path):
    """4096 slots x 2^12 cells (32 GiB of fake data), nSamples 100, maxDepth 32, through the STREAMED pipeline: every
    slot root, the dataset root and the input.json of two slots equal the oracle's; the object path agrees."""
    g4 = golden("fullsize.json")["config4"]
    cfg = pkg.make_config(**g4["config"])
    ds = ctx.dataset_streamed(cfg, g4["entropy"], threads=12)
    roots = ds.local_roots()
    assert hashlib.sha256(roots.tobytes()).hexdigest() == g4["slot_roots_sha256"]
    ds.set_roots(None)
    assert hexroot(ds.root()) == g4["dataset_root_hex"]
    total = ds.export_streamed(None, threads=12)
    texts = {}
    for slot, want in g4["inputs"].items():
        text = ds.streamed_json(int(slot))
        texts[slot] = text
        assert len(text) == want["json_bytes"]
        assert hashlib.sha256(text.encode()).hexdigest() == want["json_sha256"], slot
    assert total > 4096 * 600000
    # the object path (any entropy after the build) on the same dataset gives the same text
    assert ds.proof_input(1234, g4["entropy"]).json() == texts["1234"]
    # and a strided set of files written by the pipeline equals the per-slot texts
    ds.free()


# ---- streamed pipeline == object path, small shapes ---------------------------------------------------------------
@pytest.mark.parametrize("group,threads", [(0, 1), (1, 3), (2, 2)])
def test_streamed_equals_object_path_fake(pkg, ctx, golden, tmp_path, group, threads):
    m = golden("proof_inputs.json")["inputs"]["testmain_small"]
    cfg = pkg.make_config(**m["config"])
    ref = ctx.dataset(cfg)
    want = {s: ref.proof_input(s, m["entropy"]).json() for s in range(m["config"]["nSlots"])}
    assert want[m["slotIndex"]] == golden("input_testmain_small.json")
    ds = ctx.dataset_streamed(cfg, m["entropy"], threads=threads, group_slots=group)      # group 1: 5 passes > ring depth
    d = tmp_path / ("out%d" % group)
    d.mkdir()
    total = ds.export_streamed(str(d), threads=threads)
    assert total == sum(len(t) for t in want.values())
    for s, t in want.items():
        assert open(d / ("input_%d.json" % s)).read() == t
        assert ds.streamed_json(s) == t
    assert ds.proof_input(2, 99).json() == ref.proof_input(2, 99).json()                 # still a normal dataset
    assert ds.export_streamed(None, threads=2) == total
    plain = ctx.dataset(cfg)
    with pytest.raises(Exception):
        plain.export_streamed(None)                                                      # nothing prepared


def test_streamed_many_passes_and_odd_slot_count(pkg, ctx, oracle):
    """11 slots in groups of 2 (6 passes, twice around the 3-deep ring), odd dataset tree, vs the oracle."""
    C, P = oracle
    c = dict(maxDepth=12, maxLog2NSlots=4, cellSize=256, blockSize=2048, nSlots=11, nCells=128, nSamples=7, seed=424242)
    ds = ctx.dataset_streamed(pkg.make_config(**c), 55555, threads=4, group_slots=2)
    ds.export_streamed(None, threads=3)
    for slot in (0, 5, 10):
        assert ds.streamed_json(slot) == P.export_json(expected_proof_input_fast(C, P, c, slot, 55555, threads=4))


# ---- f1: slot files against the oracle directly -------------------------------------------------------------------
def test_slot_files_vs_oracle_directly(pkg, ctx, oracle, tmp_path):
    """SlotFile source (slot.nim:57-68, dataset.nim:34): files written from ORACLE-generated cells; roots against the
    oracle's fake_slot_root, input.json against the Python restatement run with the `file` source (both paths, classic
    and streamed).  One file is short: the missing tail reads as zeros in both implementations."""
    C, P = oracle
    c = dict(maxDepth=10, maxLog2NSlots=3, cellSize=128, blockSize=1024, nSlots=3, nCells=64, nSamples=6, seed=777)
    base = str(tmp_path / "slotdata")
    for k in range(3):
        open("%s%d.dat" % (base, k), "wb").write(C.gen_fake_cells(C.slot_seed(777, k), 0, 64, 128).tobytes())
    cf = {k: v for k, v in c.items() if k != "seed"}
    cf["file"] = base
    ds = ctx.dataset(pkg.make_config(**cf))
    roots = ds.local_roots()
    for k in range(3):
        assert np.array_equal(roots[k], C.fake_slot_root(C.slot_seed(777, k), 128, 1024, 64, 2))
    for slot in (0, 2):
        want = P.export_json(P.generate_proof_input(dict(cf), slot, 31337))
        assert ds.proof_input(slot, 31337).json() == want
    st = ctx.dataset_streamed(pkg.make_config(**cf), 31337, threads=2, group_slots=1)
    st.export_streamed(None)
    assert st.streamed_json(2) == P.export_json(P.generate_proof_input(dict(cf), 2, 31337))
    # short file: truncate slot 1 to 40.5 cells
    data = open(base + "1.dat", "rb").read()
    open(base + "1.dat", "wb").write(data[:128 * 40 + 64])
    ds2 = ctx.dataset(pkg.make_config(**cf))
    for slot in (1,):
        assert ds2.proof_input(slot, 5).json() == P.export_json(P.generate_proof_input(dict(cf), slot, 5))


# ---- sharded datasets on ONE context (the multi-GPU product path without a second GPU) ---------------------------
@pytest.mark.parametrize("source", ["fake", "file"])
def test_sharded_datasets_equal_unsharded(pkg, ctx, oracle, golden, tmp_path, source):
    """cp2_dataset_build(first_slot > 0, n_local < n_slots) + set_roots(all): uneven split 2 + 3 of the 5-slot golden
    configuration; roots, dataset root and proof inputs equal the unsharded dataset and the committed golden."""
    C, P = oracle
    m = golden("proof_inputs.json")["inputs"]["testmain_small"]
    c = dict(m["config"])
    if source == "file":
        base = str(tmp_path / "s")
        for k in range(c["nSlots"]):
            C.gen_fake_cells(C.slot_seed(c["seed"], k), 0, c["nCells"], c["cellSize"]).tofile("%s%d.dat" % (base, k))
        del c["seed"]
        c["file"] = base
    cfg = pkg.make_config(**c)
    whole = ctx.dataset(cfg)
    a, b = ctx.dataset(cfg, 0, 2), ctx.dataset(cfg, 2, 3)
    all_roots = np.concatenate([a.local_roots(), b.local_roots()])
    assert np.array_equal(all_roots, whole.local_roots())
    with pytest.raises(Exception):
        b.set_roots(None)                       # a shard cannot make the dataset tree from its own roots
    a.set_roots(all_roots)
    b.set_roots(all_roots)
    assert np.array_equal(a.root(), whole.root()) and np.array_equal(b.root(), whole.root())
    e = m["entropy"]
    assert a.proof_input(1, e).json() == whole.proof_input(1, e).json()
    assert b.proof_input(3, e).json() == whole.proof_input(3, e).json() == golden("input_testmain_small.json")
    assert b.proof_input(4, e).json() == whole.proof_input(4, e).json()
    with pytest.raises(Exception):
        a.proof_input(3, e)                     # slot 3 is not local to shard a
    # streamed shards: bodies made during the build, heads after the gather
    sb = ctx.dataset_streamed(cfg, e, 2, 3, threads=2, group_slots=2)
    sb.set_roots(all_roots)
    sb.export_streamed(None)
    assert sb.streamed_json(3) == golden("input_testmain_small.json")


@pytest.mark.parametrize("world", [1, 2])
def test_hipbackend_sharded_ranks_vs_oracle(pkg, oracle, tmp_path, world):
    """BASELINE.json configs[4]'s code path: distributed.dataset_root_sharded with HipBackend, one fresh process per
    rank (world 1, then 2 ranks under gloo sharing GPU 0), 5 slots (uneven 3 + 2); gathered roots, dataset root and
    one proof input per shard edge against the C oracle + Python restatement."""
    C, P = oracle
    c = dict(maxDepth=16, maxLog2NSlots=3, cellSize=2048, blockSize=65536, nSlots=5, nCells=256, nSamples=20, seed=2024)
    entropy = 987654321
    res = run_ranks(world, c, entropy, tmp_path)
    roots = np.stack([C.fake_slot_root(C.slot_seed(c["seed"], s), c["cellSize"], c["blockSize"], c["nCells"], 4) for s in range(c["nSlots"])])
    want_root = hexroot(C.merkle_root(roots))
    want_sha = hashlib.sha256(roots.tobytes()).hexdigest()
    covered = []
    for r in res:
        assert r["native_so_loaded"] and not r["oracle_loaded"]            # the product path, not the checker
        assert r["dataset_root_hex"] == want_root and r["all_roots_sha256"] == want_sha
        covered += list(range(r["first"], r["first"] + r["count"]))
        for slot, sha in r["inputs"].items():
            text = P.export_json(expected_proof_input_fast(C, P, c, int(slot), entropy, threads=4, slot_roots=roots))
            assert hashlib.sha256(text.encode()).hexdigest() == sha, (r["rank"], slot)
    assert sorted(covered) == list(range(c["nSlots"]))
    if world == 2:
        assert [r["count"] for r in sorted(res, key=lambda r: r["rank"])] == [3, 2]


# ---- robustness ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cell_size", [128, 256, 384, 4096])
@pytest.mark.parametrize("n", [1, 63, 65, 191, 1000])
def test_gen_fake_cells_wide_path_ragged_counts(ctx, oracle, cell_size, n):
    """The LDS-transposed write-out of k_gen_fake_cells (cell sizes that are multiples of 128) with cell counts that
    are not multiples of the 64-lane wave, a non-zero first cell and a high seed."""
    C, _ = oracle
    seed, first = (1 << 64) - 12345, 7
    got = ctx.gen_fake_cells(seed, first, n, cell_size)
    assert np.array_equal(got, C.gen_fake_cells(seed, first, n, cell_size))


def test_cache_rejects_changed_or_corrupt_files(pkg, ctx, oracle, tmp_path):
    """A SlotFile cache is keyed on size + mtime of every slot file and carries a node checksum: changed data or a
    damaged cache means rebuild (and a correct answer), never stale nodes."""
    C, P = oracle
    c = dict(maxDepth=10, maxLog2NSlots=2, cellSize=128, blockSize=1024, nSlots=2, nCells=64, nSamples=4)
    base = str(tmp_path / "slot")
    for k in range(2):
        C.gen_fake_cells(C.slot_seed(1, k), 0, 64, 128).tofile("%s%d.dat" % (base, k))
    cfg = pkg.make_config(file=base, **c)
    cache = str(tmp_path / "trees.cp2")
    first = ctx.dataset(cfg, cache=cache).proof_input(1, 9).json()
    assert ctx.dataset(cfg, cache=cache).proof_input(1, 9).json() == first               # loaded
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]                         # written by rename, no leftovers
    # new contents for slot 1 (same size): must not reuse the cached nodes
    new = C.gen_fake_cells(C.slot_seed(2, 1), 0, 64, 128)
    new.tofile(base + "1.dat")
    os.utime(base + "1.dat", ns=(1, 1))                                                  # even with an OLDER mtime
    changed = ctx.dataset(cfg, cache=cache).proof_input(1, 9).json()
    assert changed == ctx.dataset(cfg).proof_input(1, 9).json() != first
    assert changed == P.export_json(P.generate_proof_input(dict(c, file=base), 1, 9))
    # flip one node byte in the cache: checksum mismatch -> load fails, cached build rebuilds
    raw = bytearray(open(cache, "rb").read())
    raw[-5] ^= 0x40
    open(cache, "wb").write(bytes(raw))
    with pytest.raises(Exception):
        ctx.slot_trees_load(cache)
    assert ctx.dataset(cfg, cache=cache).proof_input(1, 9).json() == changed
    ctx.slot_trees_load(cache).free()                                                    # rewritten intact
    # truncated / foreign files are refused
    open(cache, "wb").write(bytes(raw[:100]))
    with pytest.raises(Exception):
        ctx.slot_trees_load(cache)
    open(cache, "wb").write(b"CP2TREE1" + bytes(200))
    with pytest.raises(Exception):
        ctx.slot_trees_load(cache)


def test_ingest_knobs_give_identical_roots(pkg, ctx, oracle):
    """cp2_set_ingest: fill threads, ring depth and chunk size change the schedule, never the result."""
    C, _ = oracle
    rng = np.random.default_rng(11)
    cells = rng.integers(0, 256, size=(2 * 4096, 2048), dtype=np.uint8)                  # 2 slots x 8 MiB
    want = ctx.slot_trees_host(cells, 2, 2048, 65536, 4096).roots()
    assert np.array_equal(want[0], C.merkle_root(np.stack(
        [C.merkle_root(C.hash_cells(cells[b * 32:(b + 1) * 32], 2048, threads=4)) for b in range(128)])))
    for threads, ring, chunk in [(1, 2, 1 << 20), (16, 8, 3 << 20), (5, 3, 2048 * 100)]:
        ctx.set_ingest(threads, ring, chunk)
        assert np.array_equal(ctx.slot_trees_host(cells, 2, 2048, 65536, 4096).roots(), want)
    ctx.set_ingest(0, 0, 0)
    with pytest.raises(Exception):
        ctx.set_ingest(-1, 0, 0)


def test_repeated_host_calls_reuse_scratch(pkg, ctx, oracle):
    """The host-pointer seam called hash by hash (how the Nim shim's `compress` uses it): correct every time; the
    context's scratch pool means no hipMalloc / hipFree per call after the first."""
    C, _ = oracle
    rng = np.random.default_rng(5)
    xy = rng.integers(0, 256, size=(200, 64), dtype=np.uint8)
    xy[:, 31] &= 0x1F
    xy[:, 63] &= 0x1F
    for key in range(4):
        want = np.stack([C.compress(xy[i, :32], xy[i, 32:], key) for i in range(200)])
        got = np.stack([ctx.compress_batch(xy[i:i + 1], key)[0] for i in range(200)])
        assert np.array_equal(got, want)


def test_host_pointer_batches_are_streamed_in_chunks(ctx, oracle):
    """cp2_permute_batch / cp2_compress_batch on host arrays larger than the 2^20-item chunk: upload, kernel and
    download of neighbouring chunks overlap through the pinned ring; results against the oracle on a strided sample
    that covers every chunk edge."""
    C, _ = oracle
    rng = np.random.default_rng(2)
    n = (1 << 21) + (1 << 20) + 12345                                    # 3 full chunks + a ragged one
    x = rng.integers(0, 256, size=(n, 96), dtype=np.uint8)
    x[:, 31] &= 0x1F
    x[:, 63] &= 0x1F
    x[:, 95] &= 0x1F
    y = ctx.permute_batch(x)
    edges = [k * (1 << 20) + d for k in range(4) for d in (-1, 0, 1) if 0 <= k * (1 << 20) + d < n]
    idx = np.unique(np.concatenate([np.arange(0, n, 4099), np.array(edges), [n - 1]]))
    assert np.array_equal(y[idx], C.permute_batch(x[idx], threads=8))
    m = (1 << 20) + 777
    xy = np.ascontiguousarray(x[:m, :64])
    for key in (0, 3):
        got = ctx.compress_batch(xy, key)
        ids = np.unique(np.concatenate([np.arange(0, m, 9973), [(1 << 20) - 1, 1 << 20, m - 1]]))
        want = np.stack([C.compress(xy[i, :32], xy[i, 32:], key) for i in ids])
        assert np.array_equal(got[ids], want)


def test_degenerate_shapes_streamed_and_classic(pkg, ctx, oracle):
    """Zero samples, one cell per block, a single block per slot, a single slot, cell sizes that are not a multiple
    of 4: both pipelines against the Python restatement."""
    C, P = oracle
    shapes = [dict(maxDepth=6, maxLog2NSlots=2, cellSize=64, blockSize=256, nSlots=3, nCells=8, nSamples=0, seed=1),
              dict(maxDepth=9, maxLog2NSlots=1, cellSize=31, blockSize=31, nSlots=1, nCells=4, nSamples=3, seed=2),        # cpb = 1
              dict(maxDepth=5, maxLog2NSlots=3, cellSize=100, blockSize=800, nSlots=6, nCells=8, nSamples=5, seed=3),      # one block
              dict(maxDepth=32, maxLog2NSlots=1, cellSize=62, blockSize=124, nSlots=1, nCells=2, nSamples=2, seed=4)]     # 1-slot dataset tree: one compression, key 3
    for c in shapes:
        cfg = pkg.make_config(**c)
        ds = ctx.dataset(cfg)
        sd = ctx.dataset_streamed(cfg, 99, threads=2, group_slots=1)
        sd.export_streamed(None, threads=2)
        for slot in range(c["nSlots"]):
            want = P.export_json(P.generate_proof_input(dict(c), slot, 99))
            assert ds.proof_input(slot, 99).json() == want, (c, slot)
            assert sd.streamed_json(slot) == want, (c, slot)


def test_streamed_slots_larger_than_the_staging_chunk(pkg, ctx, oracle):
    """Three 4 GiB slots (2^21 cells): every slot spans two 2 GiB staging chunks, so slots complete in the middle of the
    chunk loop and the sampling of slot i overlaps the hashing of slot i+1.  Streamed == classic on roots and text, and
    the sampled paths of one slot re-derive its root through the oracle's reconstructRoot (the circuit's check)."""
    C, P = oracle
    c = dict(maxDepth=32, maxLog2NSlots=2, cellSize=2048, blockSize=65536, nSlots=3, nCells=1 << 21, nSamples=20, seed=99)
    cfg = pkg.make_config(**c)
    sd = ctx.dataset_streamed(cfg, 777, threads=4)
    ref = ctx.dataset(cfg)
    assert np.array_equal(sd.local_roots(), ref.local_roots())
    sd.export_streamed(None, threads=4)
    pi = ref.proof_input(2, 777)
    assert sd.streamed_json(2) == pi.json()
    root = pkg.array_to_felts(pi.roots()[1])[0]
    idx, paths, leaves, cells = pi.cell_indices(), pi.merkle_paths(), pi.leaf_hashes(), pi.cell_data()
    for k in range(4):
        ci = int(idx[k])
        leaf = pkg.array_to_felts(leaves[k:k + 1])[0]
        assert leaf == C.array_to_felts(C.hash_bytes(cells[k]))[0]
        assert cells[k].tobytes() == P.gen_fake_cell(P.slot_seed(99, 2), ci, 2048)
        path = pkg.array_to_felts(paths[k])
        bot = P.reconstruct_root({"numberOfLeaves": 32, "leafIndex": ci % 32, "leafValue": leaf, "merklePath": path[:5]})
        top = P.reconstruct_root({"numberOfLeaves": (1 << 21) // 32, "leafIndex": ci // 32, "leafValue": bot, "merklePath": path[5:21]})
        assert top == root and path[21:] == [0] * 11
    sd.free()
    ref.free()
