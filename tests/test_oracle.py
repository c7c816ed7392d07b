"""CPU suite, part 1: the oracle against the reference's KAT and the committed fixtures, and the two
independent restatements (Python big-int, C 4x64 Montgomery) against each other."""
import hashlib
import random

import numpy as np


def test_reference_kat_python(oracle, golden):
    _, P = oracle
    kat = golden("kat_permutation.json")
    out = P.permutation(tuple(int(v) for v in kat["input"]))
    assert [hex(v) for v in out] == kat["output_hex"]          # reference/haskell/src/Poseidon2/Example.hs:13-22


def test_reference_kat_c(oracle, golden):
    C, _ = oracle
    kat = golden("kat_permutation.json")
    st = C.felts_to_array([int(v) for v in kat["input"]]).reshape(1, 96)
    assert [hex(v) for v in C.array_to_felts(C.permute_batch(st))] == kat["output_hex"]


def test_round_constants_count_and_range(oracle):
    _, P = oracle
    from oracle.p2_consts import ROUND_CONSTS
    assert len(ROUND_CONSTS) == 80 and all(0 < c < P.R_MOD for c in ROUND_CONSTS)
    assert len(P.INITIAL_RC) == 4 and len(P.INTERNAL_RC) == 56 and len(P.FINAL_RC) == 4


def test_c_vs_python_permutation_random(oracle):
    C, P = oracle
    rnd = random.Random(7)
    sts = [[rnd.randrange(P.R_MOD) for _ in range(3)] for _ in range(64)]
    sts += [[0, 0, 0], [P.R_MOD - 1] * 3, [1, 0, P.R_MOD - 1]]
    arr = np.concatenate([C.felts_to_array(s).reshape(1, 96) for s in sts])
    out = C.permute_batch(arr, threads=2)
    for i, s in enumerate(sts):
        assert tuple(C.array_to_felts(out[i])) == P.permutation(tuple(s))


def test_non_canonical_inputs_are_taken_mod_r(oracle):
    C, P = oracle
    big = [P.R_MOD + 5, 2 ** 256 - 1, P.R_MOD]
    arr = np.concatenate([np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8) for v in big]).reshape(1, 96)
    assert tuple(C.array_to_felts(C.permute_batch(arr))) == P.permutation(tuple(v % P.R_MOD for v in big))


def test_sponge_fixtures(oracle, golden):
    C, P = oracle
    g = golden("sponge_felts.json")
    for n in range(9):
        xs = list(range(1, n + 1))
        assert str(P.sponge2(xs)) == g["rate2"][n]
        assert str(P.sponge1(xs)) == g["rate1"][n]
        assert str(C.array_to_felts(C.sponge2_felts(C.felts_to_array(xs)))[0]) == g["rate2"][n]
        assert str(C.array_to_felts(C.sponge1_felts(C.felts_to_array(xs)))[0]) == g["rate1"][n]


def test_hash_bytes_fixtures(oracle, golden):
    C, P = oracle
    g = golden("hash_bytes.json")["hash"]
    for n in range(81):
        b = bytes(range(1, n + 1))
        assert str(C.array_to_felts(C.hash_bytes(b))[0]) == g[n]
        assert C.array_to_felts(C.bytes_to_felts(b)) == P.bytes_to_felts(b)
    for n in (0, 1, 30, 31, 32, 61, 62, 63, 80):
        assert str(P.hash_bytes(bytes(range(1, n + 1)))) == g[n]


def test_bytes_to_felts_padding_rule(oracle):
    _, P = oracle
    # Slot.hs:243-250: always one 0x01, then zeros to a multiple of 31
    assert P.bytes_to_felts(b"") == [1]
    assert P.bytes_to_felts(b"\xff" * 30) == [int.from_bytes(b"\xff" * 30 + b"\x01", "little")]
    assert P.bytes_to_felts(b"\xff" * 31) == [int.from_bytes(b"\xff" * 31, "little"), 1]
    assert len(P.bytes_to_felts(bytes(2048))) == 67            # (cellSize+30) div 31, cli.nim:188


def test_merkle_fixtures(oracle, golden):
    C, P = oracle
    g = golden("merkle_roots.json")
    for n in range(1, 41):
        assert str(C.array_to_felts(C.merkle_root(C.felts_to_array(list(range(1, n + 1)))))[0]) == g["felts"][n - 1]
    for n in (1, 2, 3, 7, 40):
        assert str(P.merkle_root(list(range(1, n + 1)))) == g["felts"][n - 1]
    for n in range(81):
        felts = C.bytes_to_felts(bytes(range(1, n + 1)))
        assert str(C.array_to_felts(C.merkle_root(felts))[0]) == g["bytes"][n]


def test_merkle_keys_and_singleton(oracle):
    _, P = oracle
    assert P.merkle_tree([5]) == [[5], [P.compress(5, 0, 3)]]                       # singleton: key 3
    assert P.merkle_tree([5, 6])[1] == [P.compress(5, 6, 1)]                        # bottom even: key 1
    t = P.merkle_tree([1, 2, 3])
    assert t[1] == [P.compress(1, 2, 1), P.compress(3, 0, 3)]                       # bottom odd: key 3
    assert t[2] == [P.compress(t[1][0], t[1][1], 0)]                                # above: key 0
    t5 = P.merkle_tree([1, 2, 3, 4, 5])
    assert t5[2] == [P.compress(t5[1][0], t5[1][1], 0), P.compress(t5[1][2], 0, 2)]  # upper odd: key 2


def test_merkle_proofs_round_trip(oracle):
    """Merkle.hs:136-152 testAllMerkleProofs: every proof reconstructs the root, n = 1..24, leaves 1001.."""
    _, P = oracle
    for n in range(1, 25):
        layers = P.merkle_tree(list(range(1001, 1001 + n)))
        for j in range(n):
            assert P.reconstruct_root(P.merkle_proof(layers, j)) == layers[-1][0]


def test_c_merkle_layers_match_python(oracle):
    C, P = oracle
    for n in (1, 2, 3, 6, 17, 32):
        xs = [1000 + i * i for i in range(n)]
        assert [C.array_to_felts(l) for l in C.merkle_tree(C.felts_to_array(xs))] == P.merkle_tree(xs)


def test_fake_cell_fixtures(oracle, golden):
    C, P = oracle
    for key, want in golden("fake_cells.json")["cells"].items():
        seed, idx, size = (int(v) for v in key.split("/"))
        for cell in (P.gen_fake_cell(seed, idx, size), bytes(C.gen_fake_cell(seed, idx, size))):
            assert cell[:32].hex() == want["first32_hex"]
            assert hashlib.sha256(cell).hexdigest() == want["sha256"]
        assert str(C.array_to_felts(C.hash_bytes(P.gen_fake_cell(seed, idx, size)))[0]) == want["hashCell"]
    assert C.slot_seed(12345, 3) == 12345 + 72 + 3003 == P.slot_seed(12345, 3)


def test_sampling_matches(oracle):
    C, P = oracle
    root = P.merkle_root([1, 2, 3, 4])
    for n_cells in (2, 256, 1 << 22):
        for c in (1, 2, 100):
            assert C.cell_index(C.felt_bytes(1234567), C.felt_bytes(root), n_cells, c) == P.cell_index(1234567, root, n_cells, c)


def test_proof_input_json_fixtures(oracle, golden):
    """The small configurations are regenerated with the Python oracle and must equal the committed text;
    the large one is checked through its hash and structure."""
    _, P = oracle
    meta = golden("proof_inputs.json")["inputs"]
    for name in ("testmain_small", "odd_slots_one_block"):
        m = meta[name]
        p = P.generate_proof_input(m["config"], m["slotIndex"], m["entropy"])
        assert P.circuit_check(p, m["config"])
        assert P.export_json(p) == golden("input_%s.json" % name)
        assert p["cellIndices"] == m["cellIndices"]
        assert P.circom_main(m["config"]) == m["circom_main"]
    for name, m in meta.items():
        text = golden("input_%s.json" % name)
        assert hashlib.sha256(text.encode()).hexdigest() == m["json_sha256"]
        assert text.startswith('{\n  "dataSetRoot":      "') and text.endswith("    ]\n}\n")


def test_json_layout_exact(oracle):
    """json/bn254.nim:57-74 + json/shared.nim:17-25: prefixes, indentation, zero padding."""
    _, P = oracle
    cfg = dict(maxDepth=4, maxLog2NSlots=2, cellSize=32, blockSize=64, nSlots=2, nCells=4, nSamples=2, seed=1)
    lines = P.export_json(P.generate_proof_input(cfg, 1, 5)).split("\n")
    assert lines[0] == "{" and lines[1].startswith('  "dataSetRoot":      "') and lines[2] == ', "entropy":          "5"'
    assert lines[3] == ', "nCellsPerSlot":    4' and lines[4] == ', "nSlotsPerDataSet": 2' and lines[5] == ', "slotIndex":        1'
    assert lines[7] == ', "slotProof":' and lines[8].startswith('    [ "') and lines[9] == '    , "0"' and lines[10] == "    ]"
    assert lines[11] == ', "cellData":' and lines[12].startswith('    [ [ "') and lines[13].startswith('      , "')
    assert lines[14] == "      ]" and lines[15].startswith('    , [ "')
    assert lines[-2] == "}" and lines[-1] == ""


def test_c_slot_root_matches_python(oracle):
    C, P = oracle
    cfg = dict(cellSize=64, blockSize=512, nCells=32, seed=9)
    _, big = P.build_slot_tree_full(cfg, 2)
    assert C.array_to_felts(C.fake_slot_root(C.slot_seed(9, 2), 64, 512, 32, threads=3))[0] == big[-1][0]


def test_circom_side_specification_agrees_with_haskell_side(oracle, golden):
    """The consumer-side templates (circuit/poseidon2/*.circom, restated literally in oracle/circom_ref.py) and the
    producer-side restatement (Haskell twin) give the same permutation, sponges and keyed compression."""
    import random
    from oracle import circom_ref as Cm
    _, P = oracle
    kat = golden("kat_permutation.json")
    assert [hex(v) for v in Cm.Permutation([int(v) for v in kat["input"]])] == kat["output_hex"]
    rnd = random.Random(11)
    for _ in range(10):
        st = [rnd.randrange(P.R_MOD) for _ in range(3)]
        assert tuple(Cm.Permutation(st)) == P.permutation(tuple(st))
        assert Cm.KeyedCompression(st[2] % 4, st[:2]) == P.compress(st[0], st[1], st[2] % 4)
    for n in list(range(9)) + [67, 68]:
        xs = [rnd.randrange(P.R_MOD) for _ in range(n)]
        assert Cm.Poseidon2_hash_rate2(xs) == P.sponge2(xs)
        assert Cm.Poseidon2_hash_rate1(xs[:8]) == P.sponge1(xs[:8])
    g = golden("sponge_felts.json")
    for n in range(9):
        assert str(Cm.Poseidon2_hash_rate2(list(range(1, n + 1)))) == g["rate2"][n]
        assert str(Cm.Poseidon2_hash_rate1(list(range(1, n + 1)))) == g["rate1"][n]
    # the cell hash the circuit computes (single_cell.circom:63-65) on the felts the JSON carries
    cell = P.gen_fake_cell(P.slot_seed(12345, 3), 17, 2048)
    assert Cm.Poseidon2_hash_rate2(P.bytes_to_felts(cell)) == P.hash_cell(cell, 2048)


def test_reference_vector_printer_outputs_can_pin_the_fixtures(golden):
    """tools/pin_with_reference_vectors.py: the day `reference/nim/testvectors` (or TestVectors.hs) can be built, its printed
    output pins every fixture above the permutation.  Here: the parser reads both printers' line formats (rendered from the
    fixtures themselves), accepts a faithful output and names a wrong line."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pin", os.path.join(root, "tools", "pin_with_reference_vectors.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    sp, hb, mr = golden("sponge_felts.json"), golden("hash_bytes.json"), golden("merkle_roots.json")
    nim = ["", "NIM | test vectors for sponge of field elements with rate=1", "----"]
    nim += ["hash of [1..%d] : seq[F] =  %s" % (n, v) for n, v in enumerate(sp["rate1"])]
    nim += ["", "NIM | test vectors for sponge of field elements with rate=2", "----"]
    nim += ["hash of [1..%d] : seq[F] =  %s" % (n, v) for n, v in enumerate(sp["rate2"])]
    nim += ["", "NIM | test vectors for hash (padded sponge with rate=2) of bytes", "----"]
    nim += ["hash of [1..%d] : seq[byte] =  %s" % (n, v) for n, v in enumerate(hb["hash"])]
    nim += ["", "NIM | test vectors for Merkle roots of field elements", "----"]
    nim += ["Merkle root of [1..%d] : seq[F] =  %s" % (n + 1, v) for n, v in enumerate(mr["felts"])]
    nim += ["", "NIM | test vectors for Merkle roots of sequence of bytes", "----"]
    nim += ["Merkle root of [1..%d] : seq[byte] =  %s" % (n, v) for n, v in enumerate(mr["bytes"])]
    seen, bad = pin.compare(pin.parse("\n".join(nim)))
    assert (seen, bad) == (9 + 9 + 81 + 40 + 81, [])
    hs = "\n".join(nim).replace("NIM | ", "").replace(": seq[F]", ":: [Fr]").replace(": seq[byte] =", ":: [Byte]  =")     # TestVectors.hs:28-75
    assert pin.compare(pin.parse(hs)) == (220, [])
    wrong = "\n".join(nim).replace("hash of [1..3] : seq[byte] =  " + hb["hash"][3], "hash of [1..3] : seq[byte] =  12345")
    seen, bad = pin.compare(pin.parse(wrong))
    assert seen == 219 and len(bad) == 1 and "hash_bytes.json[hash] n=3" in bad[0]

