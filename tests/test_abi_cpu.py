"""CPU suite, part 2: the C-ABI library builds for gfx950, loads, exports every declared symbol, its
host-only entry points agree with the oracle, and it FAILS LOUDLY without a GPU (no fallback)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def test_library_builds_and_exports_every_header_symbol(pkg):
    L = pkg.load_library()
    names = pkg.exported_symbols()
    assert len(names) >= 50
    for n in names:
        assert hasattr(L, n), "include/codex_p2.h declares %s but the library does not export it" % n
    # and the binding covers the whole header
    assert set(names) == set(L._cp2_signatures.keys())


def test_code_object_is_gfx950_only(pkg):
    import re
    blob = open(pkg.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_no_gpu_means_loud_failure(pkg):
    if _have_gpu():
        pytest.skip("GPU present")
    with pytest.raises(pkg.CodexP2Error) as e:
        pkg.Context(0)
    assert e.value.status == -2   # CP2_ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkgdir = os.path.join(root, "codex-storage-proofs-circuits_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", ".inc")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "p2_oracle" not in text and "libp2oracle" not in text, f
                assert "from oracle" not in text and "import oracle" not in text, f


def test_host_only_entry_points_match_oracle(pkg, oracle):
    C, P = oracle
    L = pkg.load_library()
    for n in (0, 1, 30, 31, 32, 62, 2048):
        assert L.cp2_felts_per_bytes(n) == len(P.bytes_to_felts(bytes(n)))
        data = np.frombuffer(bytes((i * 7 + 3) & 0xFF for i in range(n)), dtype=np.uint8).copy()
        out = np.zeros((L.cp2_felts_per_bytes(n), 32), dtype=np.uint8)
        assert L.cp2_bytes_to_felts(ctypes.c_void_p(data.ctypes.data) if n else None, n, ctypes.c_void_p(out.ctypes.data)) == 0
        assert pkg.array_to_felts(out) == P.bytes_to_felts(data.tobytes())
    for n in (1, 2, 3, 5, 32, 33, 1 << 17):
        layers = P.merkle_tree(list(range(n))) if n <= 33 else None
        if layers:
            assert L.cp2_merkle_num_layers(n) == len(layers)
            assert L.cp2_merkle_total(n) == sum(len(l) for l in layers)
        assert L.cp2_merkle_total(n) == C.lib().p2o_merkle_total(n)
    assert L.cp2_slot_seed(12345, 3) == P.slot_seed(12345, 3)


def test_circom_main_text(pkg, golden, tmp_path):
    for name, m in golden("proof_inputs.json")["inputs"].items():
        c = m["config"]
        cfg = pkg.make_config(**c)
        path = str(tmp_path / (name + ".circom"))
        pkg.write_circom_main(cfg, path)
        assert open(path).read() == m["circom_main"]
    bad = pkg.make_config(cellSize=2048, blockSize=2048 * 3)
    with pytest.raises(pkg.CodexP2Error):
        pkg.write_circom_main(bad, str(tmp_path / "x.circom"))       # exactLog2 assert, misc.nim:25-28


def test_shard_ranges_cover_everything(entry):
    """distributed.shard_range (one process per GPU) and cp2_shard_range (one process, cp2_multi) are the same rule."""
    import importlib
    pkg = entry.load_package()
    d = importlib.import_module("codex_storage_proofs_circuits_amd.distributed")
    for n in (1, 7, 8, 11, 4096, 32767, 32768):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                f, c = d.shard_range(n, r, world)
                assert (f, c) == pkg.shard_range(n, r, world)
                got += list(range(f, f + c))
            assert got == list(range(n))


def test_multi_gpu_plan_is_what_the_design_says(pkg):
    """cp2_multi_plan (host-only): small datasets stay on one device; many slots are dealt out whole; FEW, LARGE slots are cut
    into units until the busiest device is within 6 % of its share."""
    big = dict(maxDepth=32, cellSize=2048, blockSize=65536, nSamples=100, seed=1)
    plan = lambda n_dev, **kw: pkg.multi_plan(pkg.make_config(**dict(big, **kw)), n_dev)   # noqa: E731
    assert plan(8, maxLog2NSlots=8, nSlots=11, nCells=512) == (1, 1)                      # workflow/params.sh: 5632 cells, one context
    assert plan(8, maxLog2NSlots=15, nSlots=32768, nCells=1 << 12) == (8, 1)              # config 5's scale-down: 4096 whole slots each
    assert plan(8, maxLog2NSlots=15, nSlots=32767, nCells=1 << 12) == (8, 1)              # 4096 + 7 x 4095 ... : within 6 %
    assert plan(8, maxLog2NSlots=4, nSlots=11, nCells=1 << 22) == (8, 8)                  # 11 slots of 8 GiB: 88 units, 11 each
    assert plan(8, maxLog2NSlots=1, nSlots=1, nCells=1 << 26) == (8, 8)                   # ONE 128 GiB slot: an eighth each
    assert plan(3, maxLog2NSlots=3, nSlots=8, nCells=1 << 22) == (3, 4)                   # 32 units: 11 / 11 / 10
    assert plan(2, maxLog2NSlots=3, nSlots=8, nCells=1 << 22) == (2, 1)                   # 4 + 4 whole slots
    assert plan(8, maxLog2NSlots=3, nSlots=3, nCells=1 << 17) == (2, 2)                   # 393 216 cells: two residencies, 6 units: 3 each
    cfg = pkg.make_config(**dict(big, maxLog2NSlots=4, nSlots=11, nCells=1 << 22))
    assert pkg.multi_plan(cfg, 8, units_per_slot=1) == (8, 1)                             # whole slots only, by request: 2 2 2 1 1 1 1 1
    assert pkg.multi_plan(cfg, 8, min_cells_per_device=1 << 40) == (1, 1)
    odd = pkg.make_config(**dict(big, maxLog2NSlots=4, nSlots=11, nCells=96, blockSize=2048 * 3))
    assert pkg.multi_plan(odd, 8, min_cells_per_device=1)[1] == 1                         # not a power-of-two geometry: never cut
    with pytest.raises(pkg.CodexP2Error):
        pkg.multi_plan(cfg, 0)
    with pytest.raises(pkg.CodexP2Error):
        pkg.multi_plan(cfg, 8, units_per_slot=3)


def test_multi_handle_fails_loudly_without_a_gpu(pkg):
    """cp2_multi_init: no device, no handle (no CPU fallback behind the multi-GPU entry points either)."""
    if _have_gpu():
        pytest.skip("GPU present")
    for devices in (None, [0], [0, 0]):
        with pytest.raises(pkg.CodexP2Error) as e:
            pkg.Multi(devices)
        assert e.value.status == -2


def test_header_is_plain_c99_and_links(pkg, tmp_path):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_header")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-o", exe,
                           os.path.join(root, "tests", "c_abi", "check_header.c"), "-L" + libdir, "-lcodex_p2",
                           "-Wl,-rpath," + libdir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c abi ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def _make_pi(pkg, cfg, slot, droot, ent, sroot, proof, cells, paths):
    L = pkg.load_library()
    p = lambda a: ctypes.c_void_p(np.ascontiguousarray(a).ctypes.data)   # noqa: E731
    keep = [np.ascontiguousarray(a) for a in (droot, ent, sroot, proof, cells, paths)]
    h = ctypes.c_void_p()
    st = L.cp2_proof_input_create(ctypes.byref(cfg), slot, *(ctypes.c_void_p(a.ctypes.data) for a in keep[:4]), cells.shape[0], None,
                                  ctypes.c_void_p(keep[4].ctypes.data), ctypes.c_void_p(keep[5].ctypes.data), None, ctypes.byref(h))
    assert st == 0
    text, ln = ctypes.c_void_p(), ctypes.c_size_t()
    assert L.cp2_proof_input_json(h, ctypes.byref(text), ctypes.byref(ln)) == 0
    s = ctypes.string_at(text, ln.value).decode()
    L.cp2_free_buffer(text)
    L.cp2_proof_input_free(h)
    return s


def test_json_writer_host_only_vs_oracle_edge_values(pkg, oracle):
    """The byte-exact writer (json/bn254.nim:57-74) needs no GPU: cp2_proof_input_create + cp2_proof_input_json against
    the Python restatement, on values that sit on every edge of the decimal conversion (base-10^19 chunks, reciprocal
    division, four numbers converted side by side): 0, 1, 9, 10, 10^19 - 1, 10^19, 10^19 + 1, 10^38, 10^57, 10^76,
    r - 1, 2^256 - 1, chunks that are all zeros or all nines, and random ones."""
    C, P = oracle
    rng = np.random.default_rng(123)
    edge = [0, 1, 9, 10, 99, 100, 10**19 - 1, 10**19, 10**19 + 1, 2 * 10**19 - 1, 10**38 - 1, 10**38, 10**38 + 10**19, 10**57, 10**57 - 1,
            10**76, 10**76 - 1, 10**76 + 10**57 + 10**38 + 10**19 + 1, P.R_MOD - 1, P.R_MOD, 2**256 - 1, 2**255, 2**64 - 1, 2**64, 2**128 - 1,
            2**128, 2**192, 12345678901234567890123456789012345678901234567890123456789012345678901234567,
            99999999999999999990000000000000000000, 10**19 * (10**19 - 1), 7 * 10**76]
    edge += [int.from_bytes(rng.bytes(32), "little") >> int(s) for s in rng.integers(0, 250, size=64)]
    felt = lambda v: np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)   # noqa: E731
    md, ml, cs, ns = 7, 3, 62, (len(edge) + 6) // 7
    vals = edge + [0] * (ns * md - len(edge))
    paths = np.stack([felt(v) for v in vals]).reshape(ns, md, 32)
    cells = rng.integers(0, 256, size=(ns, cs), dtype=np.uint8)
    cells[0, :] = 0                      # all-zero cell: elements 0, 0 and the lone 0x01 marker chunk
    cells[1, :] = 255
    droot, ent, sroot = felt(edge[5]), felt(edge[7]), felt(edge[18])
    proof = np.stack([felt(v) for v in (0, 10**38, 2**256 - 1)])
    cfg = pkg.make_config(maxDepth=md, maxLog2NSlots=ml, cellSize=cs, blockSize=cs * 2, nSlots=5, nCells=64, nSamples=ns)
    got = _make_pi(pkg, cfg, 4, droot, ent, sroot, proof, cells, paths)
    want = P.export_json({"dataSetRoot": edge[5], "entropy": edge[7], "nCells": 64, "nSlots": 5, "slotIndex": 4, "slotRoot": edge[18],
                          "slotProof": {"merklePath": [0, 10**38, 2**256 - 1]},
                          "proofInputs": [{"cellData": cells[i].tobytes(), "merkleProof": {"merklePath": vals[i * md:(i + 1) * md]}}
                                          for i in range(ns)]})
    assert got == want
    # entropy handed in as 32 arbitrary bytes is stored and printed as its residue mod r (a field element in the reference)
    got_r = _make_pi(pkg, cfg, 4, droot, felt(P.R_MOD + edge[7]), sroot, proof, cells, paths)
    assert got_r == want
    # zero samples, zero-length slot proof
    cfg0 = pkg.make_config(maxDepth=0, maxLog2NSlots=0, cellSize=31, blockSize=62, nSlots=1, nCells=2, nSamples=0)
    got0 = _make_pi(pkg, cfg0, 0, felt(1), felt(2), felt(3), np.zeros((1, 32), np.uint8), np.zeros((0, 31), np.uint8), np.zeros((0, 32), np.uint8))
    want0 = P.export_json({"dataSetRoot": 1, "entropy": 2, "nCells": 2, "nSlots": 1, "slotIndex": 0, "slotRoot": 3,
                           "slotProof": {"merklePath": []}, "proofInputs": []})
    assert got0 == want0


def test_bench_self_spawn_fails_fast_without_gpus(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts its own rank processes; when they die (here: no GPU) the parent
    must notice at once, stop the rest, clean its rendezvous directory and exit non-zero with nothing on stdout -- never
    sit in a rendezvous waiting for a rank that is gone."""
    import glob
    import sys
    import time
    if _have_gpu():
        pytest.skip("GPU present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t = time.time()
    # a private TMPDIR: tempfile.mkdtemp honours it, so the rendezvous directory can only appear (and must disappear) here
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=240, env=dict(os.environ, TMPDIR=str(tmp_path)))
    assert r.returncode == 1 and r.stdout == "" and time.time() - t < 120
    assert glob.glob(str(tmp_path / "cp2_bench_rdv_*")) == []


def test_multi_plan_and_shard_range_properties(pkg):
    """Property test (hypothesis) of the host-only sharding arithmetic, cp2_multi_plan + cp2_shard_range: whatever the dataset
    and the device count, the plan uses between one and all devices, never more shards than there are items to deal, cuts
    slots only into powers of two of at least two whole blocks, never makes the balance worse by cutting, honours "whole slots
    only", and the ranges tile the items contiguously with sizes that differ by at most one."""
    from hypothesis import given, settings, strategies as st

    def imbalance(items, world):
        return -(-items // world) * world / items

    @settings(max_examples=400, deadline=None)
    @given(n_slots=st.integers(1, 5000), log_cells=st.integers(1, 26), log_cpb=st.integers(0, 5), n_dev=st.integers(1, 16),
           min_cells=st.sampled_from([0, 1, 1 << 10, 1 << 20, 1 << 34]), split=st.sampled_from([0, 0, 1, 2, 8, 64]),
           pow2_cells=st.booleans())
    def check(n_slots, log_cells, log_cpb, n_dev, min_cells, split, pow2_cells):
        cpb = 1 << min(log_cpb, log_cells)
        n_cells = 1 << log_cells
        if not pow2_cells:
            n_cells = cpb * 3                                                 # a geometry that is never cut
        cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=16, cellSize=64, blockSize=64 * cpb, nSlots=n_slots, nCells=n_cells, nSamples=5, seed=1)
        world, units = pkg.multi_plan(cfg, n_dev, min_cells_per_device=min_cells, units_per_slot=split)
        assert 1 <= world <= n_dev and units >= 1 and units & (units - 1) == 0
        assert world <= n_slots * units
        if split == 1 or not pow2_cells:
            assert units == 1
        if units > 1:
            assert n_cells % units == 0 and (n_cells // units) // cpb >= 2    # a unit is at least two whole blocks
            if split == 0:                                                    # chosen, not forced: cutting never makes the balance worse
                assert imbalance(n_slots * units, world) <= imbalance(n_slots, world) + 1e-12
        if split > 1 and pow2_cells and n_cells // cpb >= 2 * split:
            assert units == split                                             # a forced split the geometry allows is taken
        need = min_cells or 768 * 256
        assert world <= max(1, -(-n_slots * n_cells // need))                 # a device gets a shard only with `need` cells of hashing for it
        items, covered, sizes = n_slots * units, 0, []
        for r in range(world):
            first, count = pkg.shard_range(items, r, world)
            assert first == covered
            covered += count
            sizes.append(count)
        assert covered == items and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)

    check()


def test_environment_is_parsed_strictly_and_the_offender_is_named(pkg, tmp_path):
    """Every CODEX_P2_* variable holds exactly what it takes or the library refuses to start -- BEFORE any device is touched, so this
    runs on the CPU: through cp2_check_environment, and through the cli twin's error message (what a user of workflow/prove.sh sees).
    VERDICT r04 item 5: CODEX_P2_GATHER=rcl silently meant "automatic", CODEX_P2_KEEP_TREES=3 too."""
    bad = [("CODEX_P2_GATHER", "rcl"), ("CODEX_P2_GATHER", "RCCL"), ("CODEX_P2_GATHER", "rccl "), ("CODEX_P2_KEEP_TREES", "3"), ("CODEX_P2_KEEP_TREES", "-1"),
           ("CODEX_P2_KEEP_TREES", "compact"), ("CODEX_P2_GPUS", "two"), ("CODEX_P2_GPUS", "0,x"), ("CODEX_P2_GPUS", "ALL"), ("CODEX_P2_GPUS", "0"),
           ("CODEX_P2_SPLIT", "3"), ("CODEX_P2_MIN_CELLS", "1e6"), ("CODEX_P2_MEM_LIMIT_MB", "1g"), ("CODEX_P2_MEM_LIMIT_MB", "-5"),
           ("CODEX_P2_EXCHANGE_TIMEOUT_S", "soon"), ("CODEX_P2_STAGE_MB", "0"), ("CODEX_P2_STAGE_MB", "big"), ("CODEX_P2_TEST_LDS_LIMIT", "64k")]
    good = [("CODEX_P2_GATHER", "auto"), ("CODEX_P2_GATHER", "rccl"), ("CODEX_P2_GATHER", "copy"), ("CODEX_P2_GATHER", "host"), ("CODEX_P2_KEEP_TREES", "auto"),
            ("CODEX_P2_KEEP_TREES", "0"), ("CODEX_P2_KEEP_TREES", "2"), ("CODEX_P2_GPUS", "all"), ("CODEX_P2_GPUS", "8"), ("CODEX_P2_GPUS", "0,0"), ("CODEX_P2_GPUS", "2,"),
            ("CODEX_P2_SPLIT", "4"), ("CODEX_P2_MEM_LIMIT_MB", "4096"), ("CODEX_P2_EXCHANGE_TIMEOUT_S", "0"), ("CODEX_P2_STAGE_MB", "64"), ("CODEX_P2_TEST_LDS_LIMIT", "65536")]
    clean = {k: v for k, v in os.environ.items() if not k.startswith("CODEX_P2_")}
    saved = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("CODEX_P2_")}
    try:
        assert pkg.check_environment() is None
        for var, val in bad + good:
            os.environ[var] = val                                   # (putenv: the library's getenv sees it)
            said = pkg.check_environment()
            del os.environ[var]
            if (var, val) in bad:
                assert said and var in said and ('"%s"' % val) in said and "takes" in said, (var, val, said)
            else:
                assert said is None, (var, val, said)
    finally:
        os.environ.update(saved)
    # the drop-in: the same words on stderr, a non-zero exit code, no output file -- with or without a GPU in the box
    args = ["--nslots=2", "--ncells=64", "--nsamples=2", "--field=bn254", "--hash=poseidon2"]
    for var, val in (("CODEX_P2_GATHER", "rcl"), ("CODEX_P2_KEEP_TREES", "3"), ("CODEX_P2_MEM_LIMIT_MB", "lots")):
        out = str(tmp_path / "x.json")
        r = subprocess.run([pkg.CLI_PATH] + args + ["--output=" + out], env=dict(clean, **{var: val}), capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "invalid argument" in r.stderr and var in r.stderr and val in r.stderr and not os.path.exists(out), (var, r.stderr)
    # a context refuses a malformed environment too (cp2_init), whatever the device situation
    r = subprocess.run([os.sys.executable, "-c", "import sys; sys.path.insert(0, %r); import __graft_entry__ as g; p = g.load_package()\n"
                        "try:\n    p.Context(0)\nexcept p.CodexP2Error as e:\n    print('status', e.status)" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))],
                       env=dict(clean, CODEX_P2_KEEP_TREES="7"), capture_output=True, text=True, timeout=120)
    assert "status -1" in r.stdout, (r.stdout, r.stderr[-500:])
