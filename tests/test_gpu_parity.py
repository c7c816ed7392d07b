"""GPU suite (-m gpu): the HIP path through the C ABI against the oracle, bit for bit.

Sizes the oracle finishes in seconds are compared element by element; BASELINE.json's full sizes are
covered by strided samples plus size-independent properties (permutation-of-inputs equivariance,
tree-from-leaves == tree-from-cells, proofs re-derive the root the way the circuit does)."""
import hashlib

import numpy as np
import pytest

from oracle_helpers import expected_proof_input_fast

pytestmark = pytest.mark.gpu


def rand_felts(rng, n, P):
    """n x 32 uint8 canonical field elements, uniform by rejection."""
    out = np.zeros((n, 32), dtype=np.uint8)
    for i in range(n):
        while True:
            v = int.from_bytes(rng.bytes(32), "little") & ((1 << 254) - 1)
            if v < P.R_MOD:
                break
        out[i] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
    return out


def test_native_library_is_what_runs(pkg, ctx):
    assert pkg.load_library().cp2_device_is_native(ctx.h) == 1
    maps = open("/proc/self/maps").read()
    assert "libcodex_p2.so" in maps


# ---- a1 permutation -----------------------------------------------------------------------------
def test_kat_through_abi(pkg, ctx, golden):
    kat = golden("kat_permutation.json")
    out = ctx.permute_batch(pkg.felts_to_array([int(v) for v in kat["input"]]).reshape(1, 96))
    assert [hex(v) for v in pkg.array_to_felts(out)] == kat["output_hex"]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 257, 4099])
def test_permute_batch_ragged_sizes(ctx, oracle, n):
    C, P = oracle
    rng = np.random.default_rng(n)
    x = rand_felts(rng, 3 * n, P).reshape(n, 96)
    assert np.array_equal(ctx.permute_batch(x), C.permute_batch(x, threads=8))


def test_permute_batch_empty_and_edge_values(pkg, ctx, oracle):
    C, P = oracle
    assert ctx.permute_batch(np.zeros((0, 96), dtype=np.uint8)).shape == (0, 96)
    vals = [0, 1, P.R_MOD - 1, P.R_MOD, P.R_MOD + 1, 2 ** 256 - 1, 2 ** 255, (1 << 248) - 1]   # includes non-canonical
    sts = np.concatenate([np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8) for v in
                          [a for i in range(len(vals)) for a in (vals[i], vals[-i - 1], vals[(3 * i) % len(vals)])]]).reshape(-1, 96)
    assert np.array_equal(ctx.permute_batch(sts), C.permute_batch(sts))


def test_permute_batch_2p24_every_state(ctx, oracle):
    """Config 2 as SURVEY.md 8(d) states it: 2^24 states, every element uniform in [0, r) by rejection from 254-bit
    candidates (the generator bench.py times, so the 24 % of the field in [2^253, r) is covered at scale), all of them
    compared element by element with the multi-threaded C oracle (about 10 s of host time), plus equivariance:
    permuting a reversed batch gives the reversed result."""
    import os
    import torch
    import bench
    C, P = oracle
    n = 1 << 24
    g = torch.Generator(device="cuda").manual_seed(0xC0DE)
    x = bench.uniform_felts_device(torch, torch.device("cuda", 0), 3 * n, g).reshape(n, 96)
    top = x[:, 31::32].to(torch.int32)                        # most significant byte of each element
    assert int(top.max()) <= 0x30 and int((top >= 0x20).sum()) > n // 2   # canonical, and the range above 2^253 is populated
    y = torch.empty_like(x)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.permute_batch_dev(x.data_ptr(), y.data_ptr(), n)
    torch.cuda.synchronize()
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    xs, ys = x.cpu().numpy(), y.cpu().numpy()
    chunk = 1 << 21                                           # bounded host memory for the oracle's output
    for c0 in range(0, n, chunk):
        assert np.array_equal(ys[c0:c0 + chunk], C.permute_batch(xs[c0:c0 + chunk], threads=threads)), c0
    xr = torch.flip(x, dims=[0]).contiguous()
    yr = torch.empty_like(xr)
    ctx.permute_batch_dev(xr.data_ptr(), yr.data_ptr(), n)
    torch.cuda.synchronize()
    assert torch.equal(torch.flip(yr, dims=[0]), y)
    ctx.reset_stream()


# ---- a6 / a3 ---------------------------------------------------------------------------------------
def test_compress_batch_all_keys(ctx, oracle):
    C, P = oracle
    rng = np.random.default_rng(5)
    xy = rand_felts(rng, 2 * 70, P).reshape(70, 64)
    for key in range(4):
        want = np.stack([C.compress(xy[i, :32], xy[i, 32:], key) for i in range(70)])
        assert np.array_equal(ctx.compress_batch(xy, key), want)
    with pytest.raises(Exception):
        ctx.compress_batch(xy, 4)


def test_sponge_felts_fixtures_and_random(pkg, ctx, oracle, golden):
    C, P = oracle
    g = golden("sponge_felts.json")["rate2"]
    for n in range(9):
        assert str(pkg.array_to_felts(ctx.sponge2_felts(pkg.felts_to_array(list(range(1, n + 1)))))[0]) == g[n]
    rng = np.random.default_rng(11)
    for nf in (1, 2, 3, 7, 67, 68):
        f = rand_felts(rng, nf * 5, P)
        got = ctx.sponge2_felts_batch(f, nf)
        want = np.stack([C.sponge2_felts(f[i * nf:(i + 1) * nf]) for i in range(5)])
        assert np.array_equal(got, want)


# ---- a5 hashCell -------------------------------------------------------------------------------------
def test_hash_bytes_fixtures(pkg, ctx, golden):
    g = golden("hash_bytes.json")["hash"]
    for n in range(81):
        assert str(pkg.array_to_felts(ctx.hash_bytes(bytes(range(1, n + 1))))[0]) == g[n]


@pytest.mark.parametrize("cell_size,n_cells", [(2048, 1), (2048, 300), (128, 257), (256, 64), (4096, 65), (64, 1000),
                                               (31, 70), (62, 70), (93, 3), (1, 5), (100, 129), (2047, 66), (16384, 3)])
def test_hash_cells_sizes(ctx, oracle, cell_size, n_cells):
    C, _ = oracle
    rng = np.random.default_rng(cell_size * 1000 + n_cells)
    cells = rng.integers(0, 256, size=(n_cells, cell_size), dtype=np.uint8)
    assert np.array_equal(ctx.hash_cells(cells, cell_size), C.hash_cells(cells, cell_size, threads=8))


def test_hash_cells_extreme_bytes(ctx, oracle):
    C, _ = oracle
    for fill in (0x00, 0xFF, 0x01):
        cells = np.full((65, 2048), fill, dtype=np.uint8)
        assert np.array_equal(ctx.hash_cells(cells, 2048), C.hash_cells(cells, 2048, threads=4))


def test_fake_cells_and_hash_fixtures(pkg, ctx, golden):
    for key, want in golden("fake_cells.json")["cells"].items():
        seed, idx, size = (int(v) for v in key.split("/"))
        cell = ctx.gen_fake_cells(seed, idx, 1, size)[0]
        assert cell[:32].tobytes().hex() == want["first32_hex"]
        assert hashlib.sha256(cell.tobytes()).hexdigest() == want["sha256"]
        assert str(pkg.array_to_felts(ctx.hash_cells(cell, size))[0]) == want["hashCell"]


def test_gen_fake_cells_vs_oracle(ctx, oracle):
    C, _ = oracle
    for (seed, first, n, size) in [(12417, 0, 130, 2048), (99, 1 << 21, 70, 128), (2 ** 64 - 5, 3, 5, 100), (7, 0, 3, 17)]:
        assert np.array_equal(ctx.gen_fake_cells(seed, first, n, size), C.gen_fake_cells(seed, first, n, size))


# ---- a7 Merkle ------------------------------------------------------------------------------------------
def test_merkle_fixtures(pkg, ctx, golden):
    g = golden("merkle_roots.json")
    for n in range(1, 41):
        assert str(pkg.array_to_felts(ctx.merkle_root(pkg.felts_to_array(list(range(1, n + 1)))))[0]) == g["felts"][n - 1]
    for n in (0, 1, 30, 31, 80):
        assert str(pkg.array_to_felts(ctx.merkle_root(ctx.bytes_to_felts(bytes(range(1, n + 1)))))[0]) == g["bytes"][n]


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 31, 32, 33, 255, 1000, 4097])
def test_merkle_tree_all_layers(ctx, oracle, n):
    C, P = oracle
    rng = np.random.default_rng(n)
    lv = rand_felts(rng, n, P)
    a, b = ctx.merkle_tree(lv), C.merkle_tree(lv)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_merkle_empty_is_an_error(ctx):
    with pytest.raises(Exception):
        ctx.merkle_tree(np.zeros((0, 32), dtype=np.uint8))       # Merkle.hs:72 "input is empty"


def test_merkle_trees_dev_layer_major(pkg, ctx, oracle):
    import torch
    C, P = oracle
    rng = np.random.default_rng(3)
    n, nseg = 5, 7
    lv = rand_felts(rng, n * nseg, P)
    total = pkg.load_library().cp2_merkle_total(n)
    d_in = torch.from_numpy(lv).cuda()
    d_out = torch.empty((total * nseg, 32), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.merkle_trees_dev(d_in.data_ptr(), n, nseg, d_out.data_ptr())
    torch.cuda.synchronize()
    ctx.reset_stream()
    got = d_out.cpu().numpy()
    off = 0
    sizes = [len(l) for l in C.merkle_tree(lv[:n])]
    want = [C.merkle_tree(lv[s * n:(s + 1) * n]) for s in range(nseg)]
    for k, m in enumerate(sizes):
        for s in range(nseg):
            assert np.array_equal(got[off + s * m: off + (s + 1) * m], want[s][k])
        off += m * nseg


# ---- a12 sampling ---------------------------------------------------------------------------------------
def test_cell_indices(pkg, ctx, oracle):
    C, P = oracle
    root = P.merkle_root([9, 8, 7])
    for n_cells in (2, 512, 1 << 22, 1 << 32):
        got = ctx.cell_indices(pkg.felt_bytes(1234567), pkg.felt_bytes(root), n_cells, 100)
        assert list(got) == P.cell_indices(1234567, root, n_cells, 100)
    with pytest.raises(Exception):
        ctx.cell_indices(pkg.felt_bytes(1), pkg.felt_bytes(2), 1000, 3)       # not a power of two


# ---- a8/a9/a13 slot trees ---------------------------------------------------------------------------------
@pytest.mark.parametrize("cell_size,block_size,n_cells,n_slots", [(128, 4096, 256, 3), (2048, 65536, 64, 2), (64, 256, 4, 3),
                                                                  (64, 64, 8, 2), (32, 96, 12, 2), (2048, 65536, 32, 1)])
def test_slot_trees_fake_roots_and_paths(pkg, ctx, oracle, cell_size, block_size, n_cells, n_slots):
    C, P = oracle
    seed, first = 12345, 2
    trees = ctx.slot_trees_fake(seed, first, n_slots, cell_size, block_size, n_cells)
    roots = trees.roots()
    cfg = dict(cellSize=cell_size, blockSize=block_size, nCells=n_cells, seed=seed)
    cpb = block_size // cell_size
    for s in range(n_slots):
        mini, big = P.build_slot_tree_full(cfg, first + s)
        assert pkg.array_to_felts(roots[s])[0] == big[-1][0]
        idx = sorted(set([0, n_cells - 1, n_cells // 2, min(n_cells - 1, cpb)]))
        depth = trees.depth
        paths, leaves = trees.paths(s, idx, depth + 2)
        for k, ci in enumerate(idx):
            bot = P.merkle_proof(mini[ci // cpb], ci % cpb)
            top = P.merkle_proof(big, ci // cpb)
            want = P.pad_merkle_proof(P.merge_merkle_proofs(bot, top), depth + 2)
            assert pkg.array_to_felts(paths[k]) == want["merklePath"]
            assert pkg.array_to_felts(leaves[k])[0] == want["leafValue"]
    with pytest.raises(Exception):
        trees.paths(0, [0], trees.depth - 1)          # padMerkleProof assert (types.nim:29)
    with pytest.raises(Exception):
        trees.paths(0, [n_cells], trees.depth)        # merkleProof index assert (merkle.nim:27)


def test_slot_trees_host_dev_fake_agree(pkg, ctx, oracle):
    import torch
    C, _ = oracle
    cs, bs, nc, ns = 256, 2048, 64, 3
    cells = np.concatenate([C.gen_fake_cells(C.slot_seed(777, s), 0, nc, cs) for s in range(ns)])
    fake = ctx.slot_trees_fake(777, 0, ns, cs, bs, nc).roots()
    host = ctx.slot_trees_host(cells, ns, cs, bs, nc).roots()
    d = torch.from_numpy(cells).cuda()
    dev = ctx.slot_trees_dev(d.data_ptr(), ns, cs, bs, nc).roots()
    assert np.array_equal(fake, host) and np.array_equal(fake, dev)
    for s in range(ns):
        assert np.array_equal(fake[s], C.fake_slot_root(C.slot_seed(777, s), cs, bs, nc, threads=4))


def test_slot_trees_geometry_errors(ctx):
    for (cs, bs, nc) in [(2048, 65536 + 1, 64), (2048, 65536, 48), (0, 65536, 64)]:
        with pytest.raises(Exception):
            ctx.slot_trees_fake(1, 0, 1, cs, bs, nc)


def test_slot_root_2p16_cells_vs_oracle_and_full_slot_property(pkg, ctx, oracle):
    """Config 3 shape.  2^16 cells of 2048 B against the multi-threaded oracle; at 2^20 cells (2 GiB) the
    property: tree over device-resident cells == tree over generated cells, and a proof re-derives the root."""
    import torch
    C, P = oracle
    seed = 12345
    t16 = ctx.slot_trees_fake(seed, 0, 1, 2048, 65536, 1 << 16)
    assert np.array_equal(t16.roots()[0], C.fake_slot_root(C.slot_seed(seed, 0), 2048, 65536, 1 << 16, threads=8))
    n = 1 << 20
    buf = torch.empty((n, 2048), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.gen_fake_cells_dev(C.slot_seed(seed, 0), 0, n, 2048, buf.data_ptr())
    tdev = ctx.slot_trees_dev(buf.data_ptr(), 1, 2048, 65536, n)
    root_dev = tdev.roots()[0]
    ctx.reset_stream()
    tfake = ctx.slot_trees_fake(seed, 0, 1, 2048, 65536, n)
    assert np.array_equal(root_dev, tfake.roots()[0])
    idx = [0, 12345, n - 1]
    paths, leaves = tfake.paths(0, idx, 32)
    for k, ci in enumerate(idx):
        cell = buf[ci].cpu().numpy()
        leaf = C.array_to_felts(C.hash_bytes(cell))[0]
        assert leaf == pkg.array_to_felts(leaves[k])[0]
        # reconstructRoot over the merged path: bottom tree (32 leaves) then top tree, merkle.nim:51-74
        path = pkg.array_to_felts(paths[k])
        bot = P.reconstruct_root({"numberOfLeaves": 32, "leafIndex": ci % 32, "leafValue": leaf, "merklePath": path[:5]})
        top = P.reconstruct_root({"numberOfLeaves": n // 32, "leafIndex": ci // 32, "leafValue": bot, "merklePath": path[5:20]})
        assert top == pkg.array_to_felts(root_dev)[0]
        assert path[20:] == [0] * 12


# ---- a14/a15 proof input ------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["testmain_small", "odd_slots_one_block", "params_default"])
def test_proof_input_json_byte_exact(pkg, ctx, oracle, golden, name, tmp_path):
    C, P = oracle
    m = golden("proof_inputs.json")["inputs"][name]
    cfg = pkg.make_config(**m["config"])
    ds = ctx.dataset(cfg)
    pi = ds.proof_input(m["slotIndex"], m["entropy"])
    text = pi.json()
    assert text == golden("input_%s.json" % name)
    assert hashlib.sha256(text.encode()).hexdigest() == m["json_sha256"]
    assert list(pi.cell_indices()) == m["cellIndices"]
    droot, sroot, ent = pi.roots()
    assert str(pkg.array_to_felts(droot)[0]) == m["dataSetRoot"] and str(pkg.array_to_felts(sroot)[0]) == m["slotRoot"]
    path = str(tmp_path / "input.json")
    pi.write_json(path)
    assert open(path).read() == text


def test_proof_input_satisfies_circuit_rules(pkg, ctx, oracle):
    """A configuration with no committed fixture: the emitted proof input must satisfy every constraint
    SampleAndProve imposes (oracle.circuit_check mirrors circuit/codex/*.circom)."""
    C, P = oracle
    c = dict(maxDepth=12, maxLog2NSlots=4, cellSize=256, blockSize=2048, nSlots=7, nCells=128, nSamples=9, seed=31337)
    cfg = pkg.make_config(**c)
    ds = ctx.dataset(cfg)
    for slot in (0, 6):
        pi = ds.proof_input(slot, 987654321)
        droot, sroot, ent = pi.roots()
        cells, paths, idx = pi.cell_data(), pi.merkle_paths(), pi.cell_indices()
        p = {"dataSetRoot": pkg.array_to_felts(droot)[0], "entropy": 987654321, "nCells": c["nCells"], "nSlots": c["nSlots"],
             "slotIndex": slot, "slotRoot": pkg.array_to_felts(sroot)[0],
             "slotProof": {"merklePath": pkg.array_to_felts(pi.slot_proof())},
             "proofInputs": [{"cellData": cells[i].tobytes(),
                              "merkleProof": {"merklePath": pkg.array_to_felts(paths[i]), "leafIndex": int(idx[i])}}
                             for i in range(c["nSamples"])]}
        assert P.circuit_check(p, c)
        for i in range(c["nSamples"]):
            assert cells[i].tobytes() == P.gen_fake_cell(P.slot_seed(c["seed"], slot), int(idx[i]), c["cellSize"])


def test_proof_input_from_slot_files(pkg, ctx, oracle, tmp_path):
    """SlotFile data source (slot.nim:57-68, dataset.nim:34): files '<base><k>.dat' give the same proof input
    as the fake source they were written from; a short file reads as zeros."""
    C, P = oracle
    c = dict(maxDepth=10, maxLog2NSlots=3, cellSize=128, blockSize=1024, nSlots=3, nCells=64, nSamples=4, seed=555)
    base = str(tmp_path / "slotdata")
    for k in range(3):
        open("%s%d.dat" % (base, k), "wb").write(C.gen_fake_cells(C.slot_seed(555, k), 0, 64, 128).tobytes())
    a = ctx.dataset(pkg.make_config(**c)).proof_input(1, 42).json()
    cf = dict(c)
    cf["file"] = base
    del cf["seed"]
    b = ctx.dataset(pkg.make_config(**cf)).proof_input(1, 42).json()
    assert a == b
    with pytest.raises(Exception):
        ctx.dataset(pkg.make_config(**dict(cf, file=str(tmp_path / "missing"))))


def test_dataset_errors(pkg, ctx):
    c = dict(maxDepth=4, maxLog2NSlots=1, cellSize=64, blockSize=256, nSlots=5, nCells=64, nSamples=2, seed=1)
    ds = ctx.dataset(pkg.make_config(**c))
    with pytest.raises(Exception):
        ds.proof_input(0, 1)          # 5 slots need 3 levels > maxLog2NSlots, and depth 6 > maxDepth
    with pytest.raises(Exception):
        ctx.dataset(pkg.make_config(**dict(c, maxLog2NSlots=3, maxDepth=8))).proof_input(5, 1)   # slot index out of range


def test_expected_fast_equals_python_oracle(oracle, golden):
    """The helper above against the committed fixture (so that it can stand in for the slow Python path)."""
    C, P = oracle
    m = golden("proof_inputs.json")["inputs"]["testmain_small"]
    assert P.export_json(expected_proof_input_fast(C, P, m["config"], m["slotIndex"], m["entropy"], 4)) == golden("input_testmain_small.json")


def test_config4_shape_many_slots_batched(pkg, ctx, oracle):
    """Config 4 shape scaled to the test budget: 256 slots x 2^10 cells batched in one build, 100 samples,
    maxDepth 32; three proof inputs are checked against the circuit rules, three slot roots against the oracle."""
    C, P = oracle
    c = dict(maxDepth=32, maxLog2NSlots=8, cellSize=2048, blockSize=65536, nSlots=256, nCells=1024, nSamples=100, seed=12345)
    ds = ctx.dataset(pkg.make_config(**c))
    roots = ds.local_roots()
    for s in (0, 100, 255):
        assert np.array_equal(roots[s], C.fake_slot_root(C.slot_seed(12345, s), 2048, 65536, 1024, threads=8))
    ds.set_roots(None)
    assert np.array_equal(ds.root(), C.merkle_root(roots))
    pi = ds.proof_input(100, 1234567)
    text = pi.json()
    assert text.count("\n") == 8 + (8 + 1) + 1 + 100 * (67 + 1) + 1 + 1 + 100 * (32 + 1) + 1 + 1
    droot, sroot, _ = pi.roots()
    idx, cells, paths = pi.cell_indices(), pi.cell_data(), pi.merkle_paths()
    p = {"dataSetRoot": pkg.array_to_felts(droot)[0], "entropy": 1234567, "nCells": 1024, "nSlots": 256, "slotIndex": 100,
         "slotRoot": pkg.array_to_felts(sroot)[0], "slotProof": {"merklePath": pkg.array_to_felts(pi.slot_proof())},
         "proofInputs": [{"cellData": cells[i].tobytes(), "merkleProof": {"merklePath": pkg.array_to_felts(paths[i]), "leafIndex": int(idx[i])}}
                         for i in range(3)]}
    c3 = dict(c, nSamples=3)
    assert P.circuit_check(p, c3)
    # byte-exact JSON (100 samples x (67 + 32) field elements) against the oracle for one slot of the batch
    want = expected_proof_input_fast(C, P, c, 100, 1234567)
    assert text == P.export_json(want)


# ---- host layer: the cli twin and the C++ mirror of the Nim interface ------------------------------------
def test_cli_twin_params_sh_defaults(pkg, golden, tmp_path):
    """workflow/prove.sh:26 / workflow/setup.sh:13 with workflow/cli_args.sh's flags: byte-exact input.json."""
    import subprocess
    m = golden("proof_inputs.json")["inputs"]["params_default"]
    args = ["--depth=32", "--maxslots=256", "--cellsize=2048", "--blocksize=65536", "--nsamples=5", "--entropy=1234567",
            "--seed=12345", "--nslots=11", "--ncells=512", "--index=3", "--field=bn254", "--hash=poseidon2"]
    out, circ = str(tmp_path / "input.json"), str(tmp_path / "proof_main.circom")
    r = subprocess.run([pkg.CLI_PATH] + args + ["-v", "--output=" + out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "nCells     = 512" in r.stdout and r.stdout.rstrip().endswith("done")
    assert open(out).read() == golden("input_params_default.json")
    r = subprocess.run([pkg.CLI_PATH] + args + ["--circom=" + circ], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and open(circ).read() == m["circom_main"]
    # short options and ':' separators (std/parseopt forms), testMain.hs small configuration
    m2 = golden("proof_inputs.json")["inputs"]["testmain_small"]
    out2 = str(tmp_path / "small.json")
    r = subprocess.run([pkg.CLI_PATH, "-d:16", "-N=32", "-c128", "-b:4096", "-n=10", "-e:1234567", "-S12345", "-s=5", "-K:256",
                        "-i3", "-F:bn254", "-H=poseidon2", "-o=" + out2], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(out2).read() == golden("input_testmain_small.json")
    # the reference's default field is Goldilocks (cli.nim:48): out of scope here, must fail loudly, not fall back
    r = subprocess.run([pkg.CLI_PATH, "--output=" + out2], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "bn254" in r.stderr


def test_cpp_mirror_of_nim_interface(pkg, oracle, tmp_path):
    import os
    import subprocess
    C, P = oracle
    exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "api_selftest")
    r = subprocess.run([exe, "12", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines[-1] == "ALL OK"
    assert "proof input leafValues: OK." in lines
    # generateProofInputBN254 / exportProofInputBN254 with the reference's signatures: a SlotProofInput VALUE exports to
    # the oracle's text; an edited copy of the value exports with the edit (nothing hidden behind an engine handle)
    c = dict(maxDepth=10, maxLog2NSlots=3, cellSize=128, blockSize=1024, nSlots=5, nCells=64, nSamples=6, seed=777)
    want = P.generate_proof_input(c, 3, 31337)
    assert open(tmp_path / "pi.json").read() == P.export_json(want)
    assert open(tmp_path / "pi_edited.json").read() == P.export_json(dict(want, entropy=42))
    vals = {l.split(" ")[0]: l.split(" ")[1] for l in lines if " 0x" in l}
    for n in range(1, 13):
        assert int(vals["root[%d]" % n], 16) == P.merkle_root([100 + i for i in range(n)])
    cfg = dict(cellSize=64, blockSize=512, nCells=16, seed=12345)
    mini, big = P.build_slot_tree_full(cfg, 2)
    assert int(vals["slotRoot"], 16) == big[-1][0]
    want = P.pad_merkle_proof(P.merge_merkle_proofs(P.merkle_proof(mini[1], 5), P.merkle_proof(big, 1)), 8)
    assert int(vals["cellHash[13]"], 16) == want["leafValue"]
    assert [int(vals["path[%d]" % i], 16) for i in range(8)] == want["merklePath"]
    idx_line = [l for l in lines if l.startswith("cellIndices")][0]
    assert [int(v) for v in idx_line.split()[1:]] == P.cell_indices(1234567, big[-1][0], 16, 6)


def test_leaf_hashes_and_proof_input_from_parts(pkg, ctx, oracle, golden):
    """cp2_proof_input_leaf_hashes = hash of each sampled cell (leafValue of the merged proof, merkle.nim:86-100);
    cp2_proof_input_create from the accessor arrays gives the same text (the Nim shim's exportProofInputBN254 route)."""
    C, _ = oracle
    m = golden("proof_inputs.json")["inputs"]["testmain_small"]
    cfg = pkg.make_config(**m["config"])
    pi = ctx.dataset(cfg).proof_input(m["slotIndex"], m["entropy"])
    assert np.array_equal(pi.leaf_hashes(), C.hash_cells(pi.cell_data(), m["config"]["cellSize"], threads=2))
    copy = pi.recreate()
    assert copy.json() == pi.json() == golden("input_testmain_small.json")
    assert np.array_equal(copy.leaf_hashes(), pi.leaf_hashes()) and np.array_equal(copy.cell_indices(), pi.cell_indices())


def test_batched_proof_inputs_equal_single(pkg, ctx, golden, tmp_path):
    """cp2_proof_inputs_generate_batch == n x cp2_proof_input_generate; threaded JSON writer == single writer."""
    m = golden("proof_inputs.json")["inputs"]["testmain_small"]
    cfg = pkg.make_config(**m["config"])
    ds = ctx.dataset(cfg)
    slots = [3, 0, 4, 3]
    batch = ds.proof_inputs(slots, m["entropy"])
    single = [ds.proof_input(s, m["entropy"]).json() for s in slots]
    assert [p.json() for p in batch] == single
    assert single[0] == golden("input_testmain_small.json")
    paths = [str(tmp_path / ("w%d.json" % i)) for i in range(len(slots))]
    total = pkg.write_json_batch(ctx, batch, paths, threads=3)
    assert total == sum(len(s) for s in single)
    assert [open(p).read() for p in paths] == single
    assert pkg.write_json_batch(ctx, batch, None, threads=2) == total
    assert ds.proof_inputs([], 1) == []
    import pytest as _pt
    with _pt.raises(Exception):
        ds.proof_inputs([5], 1)


def test_slot_tree_cache_round_trip(pkg, ctx, golden, tmp_path):
    """Persisted trees: a dataset restored from the cache file yields byte-identical proof inputs
    without hashing a single cell; a cache written for another configuration is ignored and rebuilt."""
    import os
    import subprocess
    m = golden("proof_inputs.json")["inputs"]["testmain_small"]
    cache = str(tmp_path / "trees.cp2")
    cfg = pkg.make_config(**m["config"])
    a = ctx.dataset(cfg, cache=cache)
    assert os.path.getsize(cache) > 32 * 5 * 256 * 3 // 2
    text_a = a.proof_input(m["slotIndex"], m["entropy"]).json()
    b = ctx.dataset(cfg, cache=cache)                      # second time: loaded
    assert b.proof_input(m["slotIndex"], m["entropy"]).json() == text_a == golden("input_testmain_small.json")
    assert b.proof_input(1, 777).json() == a.proof_input(1, 777).json()
    other = pkg.make_config(**dict(m["config"], seed=999))
    c = ctx.dataset(other, cache=cache)                    # mismatch -> rebuilt and overwritten
    assert c.proof_input(0, 1).json() == ctx.dataset(other).proof_input(0, 1).json()
    # standalone trees: save / load, roots and paths identical
    t = ctx.slot_trees_fake(5, 0, 2, 128, 1024, 32)
    p2 = str(tmp_path / "t.cp2")
    t.save(p2)
    u = ctx.slot_trees_load(p2)
    assert np.array_equal(t.roots(), u.roots())
    assert np.array_equal(t.paths(1, [0, 31], 8)[0], u.paths(1, [0, 31], 8)[0])
    with pytest.raises(Exception):
        ctx.slot_trees_load(str(tmp_path / "missing.cp2"))
    # the cli twin honours CODEX_P2_CACHE
    out = str(tmp_path / "cli.json")
    args = [pkg.CLI_PATH, "-d:16", "-N=32", "-c128", "-b:4096", "-n=10", "-e:1234567", "-S12345", "-s=5", "-K:256", "-i3",
            "-F:bn254", "-H=poseidon2", "-o=" + out]
    env = dict(os.environ, CODEX_P2_CACHE=str(tmp_path / "cli.cp2"))
    for _ in range(2):
        r = subprocess.run(args, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        assert open(out).read() == golden("input_testmain_small.json")


def test_streaming_ingestion_multi_chunk(pkg, ctx, oracle, tmp_path):
    """Host and file sources larger than one 64 MiB pipeline chunk (several ring turns) against the device path."""
    import torch
    C, _ = oracle
    cs, bs, nc, ns = 2048, 65536, 1 << 15, 3            # 64 MiB per slot, 192 MiB in all
    ctx.set_ingest(0, 0, 24 << 20)                       # 24 MiB chunks: 8 ring turns, chunk edges inside slots
    d = torch.empty((ns * nc, cs), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for s in range(ns):
        ctx.gen_fake_cells_dev(C.slot_seed(4711, s), 0, nc, cs, d[s * nc].data_ptr())
    torch.cuda.synchronize()
    ctx.reset_stream()
    want = ctx.slot_trees_fake(4711, 0, ns, cs, bs, nc).roots()
    cells = d.cpu().numpy()
    assert np.array_equal(ctx.slot_trees_host(cells, ns, cs, bs, nc).roots(), want)
    base = str(tmp_path / "slot")
    for s in range(ns):
        cells[s * nc:(s + 1) * nc].tofile("%s%d.dat" % (base, s))
    cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=2, cellSize=cs, blockSize=bs, nSlots=ns, nCells=nc, nSamples=3, file=base)
    ds = ctx.dataset(cfg)
    assert np.array_equal(ds.local_roots(), want)
    fake = ctx.dataset(pkg.make_config(maxDepth=32, maxLog2NSlots=2, cellSize=cs, blockSize=bs, nSlots=ns, nCells=nc, nSamples=3, seed=4711))
    assert ds.proof_input(2, 5).json() == fake.proof_input(2, 5).json()
    ctx.set_ingest(0, 0, 0)


def test_fuzz_sizes_against_oracle(pkg, ctx, oracle):
    """Randomised shapes (seeded): ragged cell sizes / counts, Merkle sizes, sponge lengths, all bit-exact."""
    C, P = oracle
    rng = np.random.default_rng(20261003)
    for _ in range(40):
        cs = int(rng.choice([1, 2, 3, 29, 30, 31, 32, 33, 61, 62, 63, 64, 124, 127, 128, 129, 192, 250, 256, 1000, 2048, 3000]))
        n = int(rng.integers(1, 200))
        cells = rng.integers(0, 256, size=(n, cs), dtype=np.uint8)
        assert np.array_equal(ctx.hash_cells(cells, cs), C.hash_cells(cells, cs, threads=8)), (cs, n)
    for _ in range(25):
        n = int(rng.integers(1, 600))
        lv = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)       # non-canonical leaves allowed: taken mod r
        a, b = ctx.merkle_tree(lv), C.merkle_tree(lv)
        assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:])), n
    for _ in range(15):
        nf, items = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        f = rng.integers(0, 256, size=(nf * items, 32), dtype=np.uint8)
        f[:, 31] &= 0x1F
        want = np.stack([C.sponge2_felts(f[i * nf:(i + 1) * nf]) for i in range(items)])
        assert np.array_equal(ctx.sponge2_felts_batch(f, nf), want), (nf, items)


def test_two_contexts_on_two_host_threads(pkg, oracle):
    """'One context per host thread; contexts are independent' (include/codex_p2.h): two threads, two contexts,
    interleaved calls, each result equal to the oracle's."""
    import threading
    C, _ = oracle
    rng = np.random.default_rng(77)
    data = [rng.integers(0, 256, size=(3000, 256), dtype=np.uint8) for _ in range(2)]
    want = [C.hash_cells(d, 256, threads=4) for d in data]
    errs = []

    def work(k):
        try:
            c = pkg.Context(0)
            for _ in range(5):
                if not np.array_equal(c.hash_cells(data[k], 256), want[k]):
                    errs.append("mismatch in thread %d" % k)
                lv = want[k][:257]
                if not np.array_equal(c.merkle_root(lv), C.merkle_root(lv)):
                    errs.append("merkle mismatch in thread %d" % k)
            c.close()
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_caller_stream_ordering(pkg, ctx, oracle):
    """_dev calls enqueue on the caller's stream: work queued by torch before and after is ordered with them."""
    import torch
    C, _ = oracle
    st = torch.cuda.Stream()
    n = 4096
    with torch.cuda.stream(st):
        x = torch.randint(0, 256, (n, 96), dtype=torch.uint8, device="cuda")
        x[:, 31] &= 0x1F
        x[:, 63] &= 0x1F
        x[:, 95] &= 0x1F
        y = torch.zeros_like(x)
        ctx.set_stream(st.cuda_stream)
        ctx.permute_batch_dev(x.data_ptr(), y.data_ptr(), n)
        z = y.clone()                       # queued on the same stream after the kernel
    st.synchronize()
    ctx.reset_stream()
    assert np.array_equal(z.cpu().numpy(), C.permute_batch(x.cpu().numpy(), threads=4))


def test_hash_cells_large_host_input_is_pipelined(ctx, oracle):
    """> 32 MiB of host cells goes through the pinned ingestion ring (several chunks); same digests."""
    C, _ = oracle
    rng = np.random.default_rng(9)
    cells = rng.integers(0, 256, size=(70000, 2048), dtype=np.uint8)     # 137 MiB: five ring turns of 32 MiB
    ctx.set_ingest(3, 2, 32 << 20)
    got = ctx.hash_cells(cells, 2048)
    ctx.set_ingest(0, 0, 0)
    idx = np.concatenate([np.arange(0, 70000, 997), [32767, 32768, 65535, 65536, 69999]])
    assert np.array_equal(got[idx], C.hash_cells(cells[idx], 2048, threads=8))


def test_pipelined_export_writes_identical_files(pkg, ctx, golden, tmp_path):
    m = golden("proof_inputs.json")["inputs"]["testmain_small"]
    ds = ctx.dataset(pkg.make_config(**m["config"]))
    slots = [0, 1, 2, 3, 4, 3, 1]
    single = {s: ds.proof_input(s, m["entropy"]).json() for s in set(slots)}
    total = ds.export_proof_inputs(slots, m["entropy"], str(tmp_path), threads=3, batch=2)      # 4 batches
    assert total == sum(len(single[s]) for s in slots)
    for s in set(slots):
        assert open(tmp_path / ("input_%d.json" % s)).read() == single[s]
    assert single[3] == golden("input_testmain_small.json")
    assert ds.export_proof_inputs(slots, m["entropy"], None, threads=2, batch=0) == total
    assert ds.export_proof_inputs([], m["entropy"]) == 0


def test_rccl_backend_available_world1():
    """The N>1 bench path uses backend "nccl" (= RCCL): create a communicator with world_size 1 in a child process,
    all-gather uint8 rows and MAX-reduce a float64, and check that RCCL's banner stays off stdout."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_probe.py")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() == "rccl world=1 ok True 1.5", r.stdout
