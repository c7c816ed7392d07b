"""GPU suite, round 3 additions (-m gpu): bounded host memory of the streamed path (spilled bodies), cp2_trim, error reporting of
the streamed workers, the reference's `k > 0` assert in sampling, launches sliced beyond one grid's worth of items."""
import ctypes
import os

import numpy as np
import pytest

from oracle_helpers import expected_proof_input_fast

pytestmark = pytest.mark.gpu


def test_streamed_bodies_spill_beyond_the_budget(pkg, ctx, oracle, tmp_path):
    """cp2_set_body_budget: the same texts whether a body stayed in memory or went through a spill file; files are removed
    with the dataset; an unwritable spill directory is an I/O error with the file name, not silent truncation."""
    C, P = oracle
    c = dict(maxDepth=12, maxLog2NSlots=4, cellSize=256, blockSize=2048, nSlots=11, nCells=128, nSamples=7, seed=424242)
    cfg = pkg.make_config(**c)
    want = {s: P.export_json(expected_proof_input_fast(C, P, c, s, 55555, threads=4)) for s in (0, 3, 10)}
    body = len(want[0])                                  # a little more than one body
    spill = tmp_path / "spill"
    spill.mkdir()
    def spilled():
        """the spill files, all inside ONE private directory (mkdtemp: mode 0700) and readable by the owner only"""
        dirs = os.listdir(spill)
        assert len(dirs) <= 1 and all(d.startswith("cp2_bodies_") for d in dirs)
        if not dirs:
            return []
        d = spill / dirs[0]
        assert os.stat(d).st_mode & 0o777 == 0o700
        files = os.listdir(d)
        assert all(os.stat(d / f).st_mode & 0o777 == 0o600 for f in files)
        return files

    try:
        for budget, lo, hi in ((1, 11, 11), (3 * body, 7, 10), (1 << 30, 0, 0)):
            ctx.set_body_budget(budget, str(spill))
            ds = ctx.dataset_streamed(cfg, 55555, threads=3, group_slots=2)
            assert lo <= len(spilled()) <= hi, (budget, os.listdir(spill))
            out = tmp_path / ("out%d" % budget)
            out.mkdir()
            total = ds.export_streamed(str(out), threads=2)
            assert total == sum(os.path.getsize(out / f) for f in os.listdir(out))
            for s, t in want.items():
                assert ds.streamed_json(s) == t and open(out / ("input_%d.json" % s)).read() == t
            ds.free()
            assert os.listdir(spill) == []               # files and the private directory go with the dataset
        # names the old layout used, planted as symlinks to a victim file: nothing follows or overwrites them
        victim = tmp_path / "victim"
        victim.write_text("untouched")
        for s in range(11):
            os.symlink(victim, spill / ("cp2_body_%d_0_%d.part" % (os.getpid(), s)))
        ctx.set_body_budget(1, str(spill))
        ds = ctx.dataset_streamed(cfg, 55555, threads=3, group_slots=2)
        assert ds.streamed_json(3) == want[3] and victim.read_text() == "untouched"
        ds.free()
        ctx.set_body_budget(1, str(tmp_path / "does" / "not" / "exist"))
        with pytest.raises(pkg.CodexP2Error) as e:
            ctx.dataset_streamed(cfg, 55555, threads=2, group_slots=2)
        assert e.value.status == -5 and "private spill directory" in str(e.value) and "not/exist" in str(e.value)
    finally:
        ctx.set_body_budget(4 << 30, None)


def test_streamed_missing_slot_file_names_the_file(pkg, ctx, oracle, tmp_path):
    """A slot file that disappears between the build and the sampling of its cells: CP2_ERR_IO and "cannot open <file>", like
    the classic path (the trees were built from the page-cache copy; here the file is simply absent from the start)."""
    C, _ = oracle
    base = str(tmp_path / "slot")
    for k in (0, 2):
        C.gen_fake_cells(C.slot_seed(1, k), 0, 64, 128).tofile("%s%d.dat" % (base, k))
    cfg = pkg.make_config(maxDepth=10, maxLog2NSlots=2, cellSize=128, blockSize=1024, nSlots=3, nCells=64, nSamples=4, file=base)
    for build in (lambda: ctx.dataset(cfg), lambda: ctx.dataset_streamed(cfg, 7, threads=2, group_slots=1)):
        with pytest.raises(pkg.CodexP2Error) as e:
            build()
        assert e.value.status == -5 and "cannot open" in str(e.value) and "slot1.dat" in str(e.value)


def test_trim_releases_cached_scratch_and_context_keeps_working(pkg, ctx, oracle):
    import torch
    C, _ = oracle
    cells = np.random.default_rng(3).integers(0, 256, size=(40000, 2048), dtype=np.uint8)     # 78 MiB: through the pinned ring
    want = ctx.hash_cells(cells, 2048)
    assert np.array_equal(want[:64], C.hash_cells(cells[:64], 2048, threads=4))
    free_cached, _ = torch.cuda.mem_get_info()
    ctx.trim()
    free_trimmed, _ = torch.cuda.mem_get_info()
    assert free_trimmed >= free_cached + (64 << 20)                # the device ring (3 chunks) went back to the system
    assert np.array_equal(ctx.hash_cells(cells, 2048), want)       # and the next call simply allocates again
    ctx.trim()


def test_sampling_rejects_a_single_cell_like_the_reference(pkg, ctx):
    """numberOfCells = 1 means log2 = 0 and `extractLowBits` asserts k > 0 (types/bn254.nim:48)."""
    e, r = pkg.felt_bytes(5), pkg.felt_bytes(6)
    with pytest.raises(pkg.CodexP2Error):
        ctx.cell_indices(e, r, 1, 3)
    assert list(ctx.cell_indices(e, r, 2, 3)) == [int(v) & 1 for v in ctx.cell_indices(e, r, 1 << 20, 3)]
    cfg = pkg.make_config(maxDepth=4, maxLog2NSlots=1, cellSize=64, blockSize=64, nSlots=2, nCells=1, nSamples=2, seed=3)
    ds = ctx.dataset(cfg)                                           # the trees exist (one cell, one block) ...
    assert ds.local_roots().shape == (2, 32)
    with pytest.raises(pkg.CodexP2Error):
        ds.proof_input(0, 1)                                        # ... but nothing can be sampled from them
    with pytest.raises(pkg.CodexP2Error):
        ctx.dataset_streamed(cfg, 1, threads=1)
    cfg0 = pkg.make_config(maxDepth=4, maxLog2NSlots=1, cellSize=64, blockSize=64, nSlots=2, nCells=1, nSamples=0, seed=3)
    assert '"cellData"' in ctx.dataset(cfg0).proof_input(1, 1).json()   # no samples, no cellIndex call: allowed


@pytest.mark.parametrize("cell_size,n_cells,cpb", [(2048, 4096, 32), (100, 2048, 8), (31, 4096, 1)])
def test_slot_files_through_o_direct_give_identical_trees(pkg, ctx, oracle, tmp_path, cell_size, n_cells, cpb):
    """cp2_set_ingest_direct: block-aligned O_DIRECT reads into the pinned ring (chunks of a whole number of 4 KiB blocks, the
    last request rounded up into the buffer's slack, a short / odd-length file finished through the buffered descriptor and
    zero-filled) against the buffered path and the oracle.  Where the file system refuses O_DIRECT the call falls back to
    buffered reads, so this passes on tmpfs as well."""
    C, P = oracle
    c = dict(maxDepth=16, maxLog2NSlots=2, cellSize=cell_size, blockSize=cell_size * cpb, nSlots=3, nCells=n_cells, nSamples=5)
    base = str(tmp_path / "d")
    for k in range(3):
        data = C.gen_fake_cells(C.slot_seed(99, k), 0, n_cells, cell_size).tobytes()
        if k == 1:
            data = data[:len(data) // 2 + 1234]                    # short file of odd length: the tail reads as zeros
        open("%s%d.dat" % (base, k), "wb").write(data)
    cfg = pkg.make_config(file=base, **c)
    got = {}
    try:
        for direct in (0, 1):
            ctx.set_ingest_direct(direct)
            for chunk in (0, 3 * 4096 * 5):                         # default chunking, and many small chunks per slot
                ctx.set_ingest(3, 3, chunk)
                ds = ctx.dataset(cfg)
                got[(direct, chunk)] = (ds.local_roots().copy(), ds.proof_input(1, 77).json())
                ds.free()
    finally:
        ctx.set_ingest_direct(-1)
        ctx.set_ingest(0, 0, 0)
    ref_roots, ref_json = got[(0, 0)]
    for k in (0, 2):
        assert np.array_equal(ref_roots[k], C.fake_slot_root(C.slot_seed(99, k), cell_size, cell_size * cpb, n_cells, 4))
    # the short slot, by the oracle: its cells with the missing tail as zeros -> cell hashes -> block trees -> slot tree
    raw = np.frombuffer(open(base + "1.dat", "rb").read(), dtype=np.uint8)
    cells = np.zeros(n_cells * cell_size, dtype=np.uint8)
    cells[:raw.size] = raw
    leaves = C.hash_cells(cells.reshape(n_cells, cell_size), cell_size, threads=8)
    block_roots = np.stack([C.merkle_root(leaves[b * cpb:(b + 1) * cpb]) for b in range(n_cells // cpb)])
    assert np.array_equal(ref_roots[1], C.merkle_root(block_roots))
    # its proof input: the sampled cells are the file's bytes (zeros past the end), every path re-derives the slot root
    pi = ctx.dataset(cfg).proof_input(1, 77)
    idx, cd, paths = pi.cell_indices(), pi.cell_data(), pi.merkle_paths()
    root = pkg.array_to_felts(ref_roots[1:2])[0]
    nb = n_cells // cpb
    db, dt = (cpb - 1).bit_length() if cpb > 1 else 1, (nb - 1).bit_length() if nb > 1 else 1
    for k in range(5):
        ci = int(idx[k])
        assert np.array_equal(cd[k], cells[ci * cell_size:(ci + 1) * cell_size])
        leaf = C.array_to_felts(C.hash_bytes(cd[k]))[0]
        path = pkg.array_to_felts(paths[k])
        bot = P.reconstruct_root({"numberOfLeaves": cpb, "leafIndex": ci % cpb, "leafValue": leaf, "merklePath": path[:db]})
        top = P.reconstruct_root({"numberOfLeaves": nb, "leafIndex": ci // cpb, "leafValue": bot, "merklePath": path[db:db + dt]})
        assert top == root
    for key, (roots, text) in got.items():
        assert np.array_equal(roots, ref_roots) and text == ref_json, key


def test_device_allocation_failure_is_an_error_not_a_crash(pkg, ctx, oracle):
    """A batch whose node buffer cannot fit the GPU (2^31 cells x 64 B of nodes = 128+ GiB twice over) comes back as
    CP2_ERR_ALLOC with the size in the message; the context keeps working afterwards."""
    C, _ = oracle
    with pytest.raises(pkg.CodexP2Error) as e:
        ctx.slot_trees_fake(1, 0, 1 << 13, 64, 64 * 32, 1 << 20)          # 2^33 cells: 512 GiB of nodes
    assert e.value.status == -4 and "hipMalloc" in str(e.value)
    cells = C.gen_fake_cells(5, 0, 64, 128)
    assert np.array_equal(ctx.hash_cells(cells, 128), C.hash_cells(cells, 128, threads=2))


def test_independent_contexts_on_concurrent_host_threads(pkg, ctx, oracle):
    """"One context per host thread; contexts are independent" (include/codex_p2.h): three threads, each with its own context,
    run the streamed pipeline, the object path, host-array hashing and slot-tree paths at the same time on one GPU; every
    result equals what the session's context computes alone."""
    import threading
    C, P = oracle
    jobs = []
    for k in range(3):
        c = dict(maxDepth=14, maxLog2NSlots=4, cellSize=[2048, 256, 100][k], blockSize=[2048, 256, 100][k] * 8, nSlots=7 + k, nCells=256,
                 nSamples=9, seed=1000 + k)
        cfg = pkg.make_config(**c)
        ref = ctx.dataset(cfg)
        want = [ref.proof_input(s, 4242 + k).json() for s in range(c["nSlots"])]
        cells = np.random.default_rng(k).integers(0, 256, size=(3000, c["cellSize"]), dtype=np.uint8)
        jobs.append((c, want, cells, ctx.hash_cells(cells, c["cellSize"])))
        ref.free()
    errors = []

    def work(k):
        try:
            c, want, cells, want_hashes = jobs[k]
            mine = pkg.Context(0)
            for rep in range(4):
                cfg = pkg.make_config(**c)
                sd = mine.dataset_streamed(cfg, 4242 + k, threads=2, group_slots=1 + rep % 3)
                sd.export_streamed(None, threads=2)
                got = [sd.streamed_json(s) for s in range(c["nSlots"])]
                assert got == want, ("streamed", k, rep)
                assert sd.proof_input(rep % c["nSlots"], 4242 + k).json() == want[rep % c["nSlots"]], ("object", k, rep)
                sd.free()
                assert np.array_equal(mine.hash_cells(cells, c["cellSize"]), want_hashes), ("hash", k, rep)
            mine.close()
        except BaseException as e:   # noqa: B902 -- reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)
    assert want and P.export_json(expected_proof_input_fast(C, P, jobs[2][0], 0, 4244, threads=4)) == jobs[2][1][0]


def test_tree_cache_streams_large_node_buffers(pkg, ctx, golden, tmp_path):
    """cp2_slot_trees_save / _load as pipelines over a 3-deep pinned ring of 64 MiB chunks: 1 GiB of nodes (4096 slots x 2^12
    cells: sixteen chunks, five times around the ring) round-trips to the same roots (the oracle fixture's) and paths; one
    flipped byte in a middle chunk, a missing tail, or bytes appended to the file are all refused."""
    import hashlib
    g4 = golden("fullsize.json")["config4"]
    c = g4["config"]
    trees = ctx.slot_trees_fake(c["seed"], 0, c["nSlots"], c["cellSize"], c["blockSize"], c["nCells"])
    roots = trees.roots()
    assert hashlib.sha256(roots.tobytes()).hexdigest() == g4["slot_roots_sha256"]
    cells = np.array([0, 1, 4095, 2048, 77], dtype=np.uint64)
    want_paths, want_leaves = trees.paths(4095, cells, 32)
    path = str(tmp_path / "big.cp2")
    trees.save(path)
    trees.free()
    size = os.path.getsize(path)
    assert size > (1 << 30) - (1 << 20)                # 2 x 4096 x 2^12 nodes less the halving tail, plus the header
    back = ctx.slot_trees_load(path)
    assert np.array_equal(back.roots(), roots)
    got_paths, got_leaves = back.paths(4095, cells, 32)
    assert np.array_equal(got_paths, want_paths) and np.array_equal(got_leaves, want_leaves)
    back.free()
    with open(path, "r+b") as f:                      # one byte in the ninth chunk
        f.seek(size // 2 + 12345)
        b = f.read(1)
        f.seek(size // 2 + 12345)
        f.write(bytes([b[0] ^ 1]))
    with pytest.raises(pkg.CodexP2Error) as e:
        ctx.slot_trees_load(path)
    assert "checksum" in str(e.value)
    with open(path, "r+b") as f:                      # repaired, then cut short / extended
        f.seek(size // 2 + 12345)
        f.write(b)
    ctx.slot_trees_load(path).free()
    os.truncate(path, size - 32)
    with pytest.raises(pkg.CodexP2Error):
        ctx.slot_trees_load(path)
    with open(path, "ab") as f:
        f.write(bytes(64))
    with pytest.raises(pkg.CodexP2Error):
        ctx.slot_trees_load(path)
    os.remove(path)


def test_batch_proof_inputs_from_slot_files_read_cells_in_parallel(pkg, ctx, oracle, tmp_path):
    """cp2_proof_inputs_generate_batch on the SlotFile source: 40 slots x 20 samples = 800 sampled cells read by several
    threads (each a contiguous, slot-ordered range) equal the one-slot calls (single thread), the streamed path (its own
    workers) and the oracle; a missing file is named."""
    C, P = oracle
    c = dict(maxDepth=10, maxLog2NSlots=6, cellSize=256, blockSize=2048, nSlots=40, nCells=64, nSamples=20, seed=31)
    base = str(tmp_path / "f")
    for k in range(40):
        C.gen_fake_cells(C.slot_seed(31, k), 0, 64, 256).tofile("%s%d.dat" % (base, k))
    cf = {k: v for k, v in c.items() if k != "seed"}
    cfg = pkg.make_config(file=base, **cf)
    ds = ctx.dataset(cfg)
    batch = [p.json() for p in ds.proof_inputs(list(range(40)), 2025)]
    assert batch == [ds.proof_input(s, 2025).json() for s in range(40)]
    for s in (0, 17, 39):
        assert batch[s] == P.export_json(expected_proof_input_fast(C, P, c, s, 2025, threads=4))
    sd = ctx.dataset_streamed(cfg, 2025, threads=3, group_slots=7)
    sd.export_streamed(None, threads=2)
    assert [sd.streamed_json(s) for s in range(40)] == batch
    os.remove(base + "23.dat")
    with pytest.raises(pkg.CodexP2Error) as e:
        ds.proof_inputs(list(range(40)), 2025)
    assert "cannot open" in str(e.value) and "f23.dat" in str(e.value)


@pytest.mark.parametrize("cpb,nblocks", [(3, 5), (5, 7), (2, 9), (7, 1), (1, 11), (6, 3)])
def test_every_leaf_proof_reconstructs_the_root_in_odd_trees(pkg, ctx, oracle, cpb, nblocks):
    """The reference's own Merkle property test (reference/haskell/src/Poseidon2/Merkle.hs:141-152: the proof of EVERY leaf
    re-derives the root) on slot trees whose block trees and big trees are odd at several levels (keys 2 / 3, a zero sibling
    where the reference reads out of range, merkle.nim:33-34): bottom proof inside the block, top proof over the block roots,
    both through the oracle's reconstructRoot (merkle.nim:51-74)."""
    C, P = oracle
    cs, n_cells = 64, cpb * nblocks
    trees = ctx.slot_trees_fake(77, 1, 2, cs, cs * cpb, n_cells)
    depth = trees.depth
    db = max(1, (cpb - 1).bit_length())
    dt = max(1, (nblocks - 1).bit_length())
    assert depth == db + dt
    for s in range(2):
        root = pkg.array_to_felts(trees.roots()[s:s + 1])[0]
        assert np.array_equal(trees.roots()[s], C.fake_slot_root(C.slot_seed(77, 1 + s), cs, cs * cpb, n_cells, 2))
        cells = C.gen_fake_cells(C.slot_seed(77, 1 + s), 0, n_cells, cs)
        paths, leaves = trees.paths(s, list(range(n_cells)), depth + 1)
        for ci in range(n_cells):
            leaf = pkg.array_to_felts(leaves[ci:ci + 1])[0]
            assert leaf == C.array_to_felts(C.hash_bytes(cells[ci]))[0]
            path = pkg.array_to_felts(paths[ci])
            bot = P.reconstruct_root({"numberOfLeaves": cpb, "leafIndex": ci % cpb, "leafValue": leaf, "merklePath": path[:db]})
            top = P.reconstruct_root({"numberOfLeaves": nblocks, "leafIndex": ci // cpb, "leafValue": bot, "merklePath": path[db:depth]})
            assert top == root and path[depth] == 0, (s, ci)


def test_entropy_is_stored_and_printed_as_its_canonical_representative(pkg, ctx, oracle):
    """`Entropy` is a field element in the reference (types/bn254.nim:21): 32 bytes that encode r + 5, 4r + 5 or 2^256 - 1 give the
    proof input (indices, paths, and the "entropy" line of input.json) of their residues, on both paths."""
    C, P = oracle
    c = dict(maxDepth=8, maxLog2NSlots=2, cellSize=64, blockSize=256, nSlots=3, nCells=16, nSamples=4, seed=9)
    cfg = pkg.make_config(**c)
    ds = ctx.dataset(cfg)
    as_bytes = lambda v: np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)   # noqa: E731
    for raw in (P.R_MOD + 5, 4 * P.R_MOD + 5, 2 ** 256 - 1, P.R_MOD):
        want = P.export_json(P.generate_proof_input(dict(c), 1, raw % P.R_MOD))
        assert ds.proof_input(1, as_bytes(raw)).json() == want
        sd = ctx.dataset_streamed(cfg, as_bytes(raw), threads=2, group_slots=1)
        sd.export_streamed(None)
        assert sd.streamed_json(1) == want
        sd.free()
        # and the third entry point: an object assembled from caller arrays (cp2_proof_input_create) with the RAW entropy
        pi = ds.proof_input(1, as_bytes(raw % P.R_MOD))
        d, s_root, _ = pi.roots()
        sp, idx, cells, paths = (np.ascontiguousarray(a) for a in (pi.slot_proof(), pi.cell_indices(), pi.cell_data(), pi.merkle_paths()))
        h = ctypes.c_void_p()
        e = np.ascontiguousarray(as_bytes(raw))
        st = ctx.L.cp2_proof_input_create(ctypes.byref(cfg), 1, *(ctypes.c_void_p(a.ctypes.data) for a in (d, e, s_root, sp)), idx.size,
                                          ctypes.c_void_p(idx.ctypes.data), ctypes.c_void_p(cells.ctypes.data), ctypes.c_void_p(paths.ctypes.data),
                                          None, ctypes.byref(h))
        assert st == 0
        made = pkg.ProofInput(ctx, h, cfg)
        assert made.json() == want and int.from_bytes(made.roots()[2].tobytes(), "little") == raw % P.R_MOD
