"""CPU ORACLE (test infrastructure, NOT product code) -- Python big-int restatement.

This module restates, with Python integers, the algorithm of the Codex storage-proof
"proof input" path of codex-storage/codex-storage-proofs-circuits.  It is the slow, readable
half of the oracle (the fast half is `p2_oracle.c`, an independent 4x64-bit Montgomery
implementation); the two are cross-checked against each other in tests/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under
oracle/.  The product path (the HIP library behind include/codex_p2.h) never does.

PINNING STATUS
  * permutation: pinned by the reference's one committed known-answer test,
    reference/haskell/src/Poseidon2/Example.hs:13-19 (tests/golden/kat_permutation.json).
  * everything above the permutation (sponge, byte padding, Merkle keys, sampling, JSON):
    the reference commits NO expected values (reference/haskell/src/TestVectors.hs and
    reference/nim/testvectors/src/testvectors.nim only print).  The arithmetic the Nim tool
    really calls lives in un-vendored third-party packages
    (codex-storage/nim-poseidon2 @ 4e2c6e619b2f2859aaa4b2aed2f346ea4d0c67a3,
     mratsim/constantine @ bc3845aa492b52f7fef047503b1592e830d1a774;
     reference/nim/proof_input/proof_input.nimble:11-12) and no Nim/Haskell/circom toolchain
    exists in this image, so the reference cannot be run.  For those layers: PARITY UNPINNED at
    the nim-poseidon2 boundary; they follow the in-tree Haskell + circom + README specification
    and are checked by the circuit's own consistency rules (tests re-derive every root the way
    circuit/codex/*.circom does).

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""

from .p2_consts import ROUND_CONSTS

# BN254 scalar field modulus: README.md:76, test/Params.hs:12
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617

INITIAL_RC = [tuple(ROUND_CONSTS[3 * i:3 * i + 3]) for i in range(4)]
INTERNAL_RC = ROUND_CONSTS[12:68]
FINAL_RC = [tuple(ROUND_CONSTS[68 + 3 * i:68 + 3 * i + 3]) for i in range(4)]

KEY_NONE, KEY_BOTTOM, KEY_ODD, KEY_ODD_BOTTOM = 0, 1, 2, 3


# --------------------------------------------------------------------------------------
# a1: permutation -- reference/haskell/src/Poseidon2/Permutation.hs:14-45
# --------------------------------------------------------------------------------------

def sbox(x):
    """Permutation.hs:14-17  x^5"""
    x2 = x * x % R_MOD
    x4 = x2 * x2 % R_MOD
    return x4 * x % R_MOD


def internal_round(c, st):
    """Permutation.hs:19-26"""
    x, y, z = st
    xp = sbox((x + c) % R_MOD)
    return ((2 * xp + y + z) % R_MOD, (xp + 2 * y + z) % R_MOD, (xp + y + 3 * z) % R_MOD)


def external_round(c, st):
    """Permutation.hs:28-33"""
    x, y, z = st
    xp = sbox((x + c[0]) % R_MOD)
    yp = sbox((y + c[1]) % R_MOD)
    zp = sbox((z + c[2]) % R_MOD)
    s = (xp + yp + zp) % R_MOD
    return ((xp + s) % R_MOD, (yp + s) % R_MOD, (zp + s) % R_MOD)


def linear_layer(st):
    """Permutation.hs:35-36"""
    x, y, z = st
    s = (x + y + z) % R_MOD
    return ((x + s) % R_MOD, (y + s) % R_MOD, (z + s) % R_MOD)


def permutation(st):
    """Permutation.hs:40-45: linearLayer, 4 external, 56 internal, 4 external rounds."""
    st = linear_layer(tuple(v % R_MOD for v in st))
    for c in INITIAL_RC:
        st = external_round(c, st)
    for c in INTERNAL_RC:
        st = internal_round(c, st)
    for c in FINAL_RC:
        st = external_round(c, st)
    return st


# --------------------------------------------------------------------------------------
# a3: sponge -- reference/haskell/src/Poseidon2/Sponge.hs:14-43
#     (circuit/poseidon2/poseidon2_sponge.circom:43-61 for the same padding and IV)
# --------------------------------------------------------------------------------------

def sponge1(felts):
    """Sponge.hs:14-27: rate 1, IV (0,0,2^64+0x0301), 10* padding over field elements."""
    civ = (1 << 64) + 0x0301
    st = (0, 0, civ)
    for a in list(felts) + [1]:
        st = permutation(((st[0] + a) % R_MOD, st[1], st[2]))
    return st[0]


def sponge2(felts):
    """Sponge.hs:30-43: rate 2, IV (0,0,2^64+0x0302), pad with 1 then 0 to even length."""
    civ = (1 << 64) + 0x0302
    xs = [v % R_MOD for v in felts]
    xs = xs + ([1] if len(xs) % 2 == 1 else [1, 0])
    st = (0, 0, civ)
    for i in range(0, len(xs), 2):
        st = permutation(((st[0] + xs[i]) % R_MOD, (st[1] + xs[i + 1]) % R_MOD, st[2]))
    return st[0]


# --------------------------------------------------------------------------------------
# a4: bytes -> field elements -- reference/haskell/src/Slot.hs:243-270, README.md:86-99
# --------------------------------------------------------------------------------------

def bytes_to_felts(data):
    """Slot.hs:243-270: append 0x01, zero-pad to a multiple of 31, 31-byte little-endian chunks."""
    bs = bytes(data) + b"\x01"
    if len(bs) % 31:
        bs += b"\x00" * (31 - len(bs) % 31)
    return [int.from_bytes(bs[i:i + 31], "little") for i in range(0, len(bs), 31)]


def hash_bytes(data):
    """Slot.hs:222-231 hashCell_ = sponge2 . cellDataToFieldElements;
    Nim call site reference/nim/proof_input/src/blocks/bn254.nim:27 Sponge.digest(bytes, rate=2)."""
    return sponge2(bytes_to_felts(data))


def hash_cell(cell, cell_size=None):
    """blocks/bn254.nim:23-29 hashCell (asserts the cell length)."""
    if cell_size is not None and len(cell) != cell_size:
        raise AssertionError("cells are expected to be exactly %d bytes" % cell_size)
    return hash_bytes(cell)


# --------------------------------------------------------------------------------------
# a6/a7: keyed compression and Merkle tree
#   reference/nim/proof_input/src/merkle/bn254.nim:18-63, reference/haskell/src/Poseidon2/Merkle.hs:69-83,156-203
# --------------------------------------------------------------------------------------

def compress(x, y, key=0):
    """Merkle.hs:202-203 keyedCompression: first component of perm(x, y, key)."""
    return permutation((x, y, key))[0]


def merkle_tree(leaves):
    """merkle/bn254.nim:24-63 merkleTreeWorker: all layers, bottom first.

    key = 1 on the bottom layer, 0 above; an odd tail node is compress(last, 0) with key+2;
    a singleton input still gets one compression (key 3)."""
    xs = [v % R_MOD for v in leaves]
    if not xs:
        raise AssertionError("merkle_tree: input is empty")  # Merkle.hs:72
    layers = []
    bottom = True
    while True:
        layers.append(xs)
        m = len(xs)
        if m == 1 and not bottom:
            return layers
        half = m // 2
        ys = [compress(xs[2 * i], xs[2 * i + 1], KEY_BOTTOM if bottom else KEY_NONE) for i in range(half)]
        if m % 2 == 1:
            ys.append(compress(xs[m - 1], 0, KEY_ODD_BOTTOM if bottom else KEY_ODD))
        xs = ys
        bottom = False


def merkle_root(leaves):
    """Merkle.hs:171-181 calcMerkleRoot / merkle.nim:14-17 treeRoot."""
    return merkle_tree(leaves)[-1][0]


def merkle_proof(layers, index):
    """merkle.nim:21-42: sibling per layer, ZERO when the sibling is out of range."""
    depth = len(layers) - 1
    nleaves = len(layers[0])
    assert 0 <= index < nleaves
    path = []
    k, m = index, nleaves
    for i in range(depth):
        j = k ^ 1
        path.append(layers[i][j] if j < m else 0)
        k >>= 1
        m = (m + 1) >> 1
    return {"leafIndex": index, "leafValue": layers[0][index], "merklePath": path, "numberOfLeaves": nleaves}


def reconstruct_root(proof):
    """merkle.nim:51-74 reconstructRoot."""
    m, j, h = proof["numberOfLeaves"], proof["leafIndex"], proof["leafValue"]
    bottom = 1
    for p in proof["merklePath"]:
        if j & 1:
            h = compress(p, h, bottom)
        elif j == m - 1:
            h = compress(h, p, bottom + 2)
        else:
            h = compress(h, p, bottom)
        bottom = 0
        j >>= 1
        m = (m + 1) >> 1
    return h


def merge_merkle_proofs(bottom_proof, top_proof):
    """merkle.nim:86-100 mergeMerkleProofs (asserts the bottom root equals the top leaf)."""
    assert reconstruct_root(bottom_proof) == top_proof["leafValue"]
    return {
        "leafIndex": top_proof["leafIndex"] * bottom_proof["numberOfLeaves"] + bottom_proof["leafIndex"],
        "leafValue": bottom_proof["leafValue"],
        "merklePath": bottom_proof["merklePath"] + top_proof["merklePath"],
        "numberOfLeaves": bottom_proof["numberOfLeaves"] * top_proof["numberOfLeaves"],
    }


def pad_merkle_proof(proof, newlen):
    """types.nim:27-37 padMerkleProof."""
    pad = newlen - len(proof["merklePath"])
    assert pad >= 0
    out = dict(proof)
    out["merklePath"] = proof["merklePath"] + [0] * pad
    return out


# --------------------------------------------------------------------------------------
# a10: fake slot data -- reference/nim/proof_input/src/slot.nim:22-32, dataset.nim:32
# --------------------------------------------------------------------------------------

M64 = (1 << 64) - 1


def gen_fake_cell(seed, idx, cell_size):
    """slot.nim:23-32 genFakeCell (wrapping uint64 arithmetic, then mod 1698428844001831)."""
    seed1 = (seed + 0xDEADCAFE) & M64
    seed2 = (idx + 0x98765432) & M64
    state = 1
    out = bytearray(cell_size)
    for i in range(cell_size):
        state = (state * ((state + seed1) & M64) * ((state + seed2) & M64)
                 + state * (state ^ 0x5A5A5A5A) + seed1 * state + ((seed2 + 17) & M64)) & M64
        state %= 1698428844001831
        out[i] = state & 0xFF
    return bytes(out)


def slot_seed(seed, slot_idx):
    """dataset.nim:32 parametricSlotSeed."""
    return (seed + 72 + 1001 * slot_idx) & M64


# --------------------------------------------------------------------------------------
# a12: sampling -- reference/nim/proof_input/src/sample/bn254.nim:16-27, types/bn254.nim:47-59
# --------------------------------------------------------------------------------------

def ceiling_log2(x):
    """misc.nim:10-23"""
    if x == 0:
        return -1
    return (x - 1).bit_length()


def cell_index(entropy, slot_root, n_cells, counter):
    """sample/bn254.nim:16-24: low log2(nCells) bits of sponge2[entropy, slotRoot, counter]."""
    log2 = ceiling_log2(n_cells)
    assert (1 << log2) == n_cells, "numberOfCells is assumed to be a power of two"
    h = sponge2([entropy, slot_root, counter])
    return h & ((1 << log2) - 1)


def cell_indices(entropy, slot_root, n_cells, n_samples):
    """sample/bn254.nim:26-27 (counters 1..nSamples)."""
    return [cell_index(entropy, slot_root, n_cells, c) for c in range(1, n_samples + 1)]


# --------------------------------------------------------------------------------------
# a8/a9/a14: slot tree and proof input -- blocks/bn254.nim:33-67, gen_input/bn254.nim:21-79
# --------------------------------------------------------------------------------------

def load_cell(cfg, slot_idx, cell_idx):
    """slot.nim:57-68 slotLoadCellData, dataset.nim:34,45-51 (fake data or '<base><k>.dat')."""
    if cfg.get("file"):
        with open("%s%d.dat" % (cfg["file"], slot_idx), "rb") as f:
            f.seek(cfg["cellSize"] * cell_idx)
            data = f.read(cfg["cellSize"])
        return data + b"\x00" * (cfg["cellSize"] - len(data))
    return gen_fake_cell(slot_seed(cfg["seed"], slot_idx), cell_idx, cfg["cellSize"])


def build_slot_tree_full(cfg, slot_idx):
    """gen_input/bn254.nim:21-30 buildSlotTreeFull -> (miniTrees, bigTree)."""
    cpb = cfg["blockSize"] // cfg["cellSize"]
    assert cpb * cfg["cellSize"] == cfg["blockSize"]
    nblocks = cfg["nCells"] // cpb
    assert nblocks * cpb == cfg["nCells"]
    mini = []
    for b in range(nblocks):
        leaves = [hash_cell(load_cell(cfg, slot_idx, b * cpb + i), cfg["cellSize"]) for i in range(cpb)]
        mini.append(merkle_tree(leaves))          # blocks/bn254.nim:60-67 networkBlockTree
    big = merkle_tree([t[-1][0] for t in mini])   # gen_input/bn254.nim:28-29
    return mini, big


def generate_proof_input(cfg, slot_idx, entropy):
    """gen_input/bn254.nim:35-74 generateProofInput.

    cfg keys: maxDepth, maxLog2NSlots, cellSize, blockSize, nSlots, nCells, nSamples, seed | file.
    (The reference rebuilds the slot tree once per sample, :57 -- same values, so built once here.)"""
    cpb = cfg["blockSize"] // cfg["cellSize"]
    trees = [build_slot_tree_full(cfg, i) for i in range(cfg["nSlots"])]
    slot_roots = [big[-1][0] for (_, big) in trees]
    dset_tree = merkle_tree(slot_roots)
    slot_proof = merkle_proof(dset_tree, slot_idx)
    mini, big = trees[slot_idx]
    our_root = slot_roots[slot_idx]
    indices = cell_indices(entropy, our_root, cfg["nCells"], cfg["nSamples"])
    inputs = []
    for ci in indices:
        bi = ci // cpb
        bot = merkle_proof(mini[bi], ci % cpb)
        top = merkle_proof(big, bi)
        prf = pad_merkle_proof(merge_merkle_proofs(bot, top), cfg["maxDepth"])
        inputs.append({"cellData": load_cell(cfg, slot_idx, ci), "merkleProof": prf})
    return {
        "dataSetRoot": dset_tree[-1][0],
        "entropy": entropy % R_MOD,
        "nCells": cfg["nCells"],
        "nSlots": cfg["nSlots"],
        "slotIndex": slot_idx,
        "slotRoot": our_root,
        "slotProof": pad_merkle_proof(slot_proof, cfg["maxLog2NSlots"]),
        "proofInputs": inputs,
        "cellIndices": indices,
    }


# --------------------------------------------------------------------------------------
# a15: JSON export -- json/bn254.nim:19-74, json/shared.nim:9-25, types/bn254.nim:29-43
# --------------------------------------------------------------------------------------

def _q(x):
    return '"%d"' % x   # toQuotedDecimalF: canonical decimal, no leading zeros, "0" for zero


def _write_felt_list(lines, prefix, xs):
    """json/shared.nim:17-25 writeList specialised to writeLnF."""
    indent = " " * len(prefix)
    for i, x in enumerate(xs):
        lines.append((prefix + "[ " if i == 0 else indent + ", ") + _q(x))
    lines.append(indent + "]")


def _write_list_of_lists(lines, xss):
    prefix = "    "
    indent = " " * len(prefix)
    for i, xs in enumerate(xss):
        _write_felt_list(lines, prefix + "[ " if i == 0 else indent + ", ", xs)
    lines.append(indent + "]")


def export_json(p):
    """json/bn254.nim:57-74 exportProofInput: exact text, every line newline-terminated."""
    lines = ["{"]
    lines.append('  "dataSetRoot":      ' + _q(p["dataSetRoot"]))
    lines.append(', "entropy":          ' + _q(p["entropy"]))
    lines.append(', "nCellsPerSlot":    %d' % p["nCells"])
    lines.append(', "nSlotsPerDataSet": %d' % p["nSlots"])
    lines.append(', "slotIndex":        %d' % p["slotIndex"])
    lines.append(', "slotRoot":         ' + _q(p["slotRoot"]))
    lines.append(', "slotProof":')
    _write_felt_list(lines, "    ", p["slotProof"]["merklePath"])
    lines.append(', "cellData":')
    _write_list_of_lists(lines, [bytes_to_felts(q["cellData"]) for q in p["proofInputs"]])
    lines.append(', "merklePaths":')
    _write_list_of_lists(lines, [q["merkleProof"]["merklePath"] for q in p["proofInputs"]])
    lines.append("}")
    return "\n".join(lines) + "\n"


def circom_main(cfg):
    """cli.nim:186-204 writeCircomMainComponent (Nim `$` of an int tuple prints "(a, b, c, d, e)")."""
    cpb = cfg["blockSize"] // cfg["cellSize"]
    depth = ceiling_log2(cpb)
    assert (1 << depth) == cpb, "exactLog2: not a power of two"
    params = (cfg["maxDepth"], cfg["maxLog2NSlots"], depth, (cfg["cellSize"] + 30) // 31, cfg["nSamples"])
    return ("pragma circom 2.0.0;\n"
            'include "sample_cells.circom";\n'
            "// SampleAndProven( maxDepth, maxLog2NSlots, blockTreeDepth, nFieldElemsPerCell, nSamples )\n"
            "component main {public [entropy,dataSetRoot,slotIndex]} = SampleAndProve(%d, %d, %d, %d, %d);\n" % params)


# --------------------------------------------------------------------------------------
# the circuit's own acceptance rules, used by tests as a consistency check of any proof input
#   circuit/codex/sample_cells.circom:58-148, single_cell.circom:30-73, merkle.circom:44-114
# --------------------------------------------------------------------------------------

def _circom():
    from . import circom_ref
    return circom_ref


def circuit_root_from_path(leaf, path_bits, last_bits, mask_bits, path):
    """circuit/codex/merkle.circom:44-114 RootFromMerklePath, signal for signal (hashing through the circom-side
    restatement oracle/circom_ref.py, not through this module's producer-side functions).

    path_bits / last_bits: bits of the leaf index / of the last index, LSB first (length depth);
    mask_bits: depth+1 bits [1,..,1,0,..,0]; path: depth siblings.  Returns recRoot."""
    depth = len(path)
    assert len(path_bits) == depth and len(last_bits) == depth and len(mask_bits) == depth + 1
    mc = [1] + list(mask_bits[1:])                       # maskBitsCorrected, merkle.circom:60-62
    aux = [0] * (depth + 1)
    aux[0] = leaf
    is_last = [0] * (depth + 1)                          # merkle.circom:73-80
    is_last[depth] = 1
    for i in range(depth - 1, -1, -1):
        is_last[i] = is_last[i + 1] * (1 if path_bits[i] == last_bits[i] else 0)
    for i in range(depth):                               # merkle.circom:85-103
        bottom = 1 if i == 0 else 0
        odd = is_last[i] * (1 - path_bits[i])
        L, Rr = aux[i], path[i]
        sw = ((Rr - L) * path_bits[i]) % R_MOD
        aux[i + 1] = _circom().KeyedCompression(bottom + 2 * odd, [(L + sw) % R_MOD, (Rr - sw) % R_MOD])
    return sum((mc[i] - mc[i + 1]) * aux[i + 1] for i in range(depth)) % R_MOD   # merkle.circom:106-112


def circuit_check(p, cfg):
    """What `SampleAndProve` constrains (circuit/codex/sample_cells.circom:58-148 with
    single_cell.circom:30-73): the dataset-root check, then per sample the index derivation, the cell hash,
    the bottom (block) tree and the middle tree up to slotRoot.  Returns True or raises AssertionError."""
    cpb = cfg["blockSize"] // cfg["cellSize"]
    bot_depth = ceiling_log2(cpb)
    max_depth, max_slots_log = cfg["maxDepth"], cfg["maxLog2NSlots"]
    n_cells, n_slots = p["nCells"], p["nSlots"]
    # --- top: sample_cells.circom:95-109 (ToBits(slotIndex), CeilingLog2(nSlots): bits of nSlots-1, mask)
    sidx = p["slotIndex"]
    sbits = [(sidx >> i) & 1 for i in range(max_slots_log)]
    lbits = [((n_slots - 1) >> i) & 1 for i in range(max_slots_log)]
    mask = [1 if ((n_slots - 1) >> i) != 0 else 0 for i in range(max_slots_log)] + [0]   # lib/log2.circom:108-130
    sp = p["slotProof"]["merklePath"]
    assert len(sp) == max_slots_log
    assert circuit_root_from_path(p["slotRoot"], sbits, lbits, mask, sp) == p["dataSetRoot"], "dataset root mismatch"
    # --- samples: sample_cells.circom:114-146
    lgmask = [1 if (1 << i) < n_cells else 0 for i in range(max_depth + 1)]               # lib/log2.circom:76-78
    assert lgmask[0] == 1 and lgmask[max_depth] == 0
    last_bits = lgmask[:max_depth]
    assert len(p["proofInputs"]) == cfg["nSamples"]
    for cnt, q in enumerate(p["proofInputs"]):
        h = _circom().Poseidon2_hash_rate2([p["entropy"], p["slotRoot"], cnt + 1])         # :23-48
        index_bits = [lgmask[i] * ((h >> i) & 1) for i in range(max_depth)]
        felts = bytes_to_felts(q["cellData"])
        assert len(felts) == (cfg["cellSize"] + 30) // 31
        leaf = _circom().Poseidon2_hash_rate2(felts)                                        # single_cell.circom:63-65
        path = q["merkleProof"]["merklePath"]
        assert len(path) == max_depth
        bot = circuit_root_from_path(leaf, index_bits[:bot_depth], last_bits[:bot_depth],
                                     lgmask[:bot_depth] + [0], path[:bot_depth])           # single_cell.circom:41-60
        mid = circuit_root_from_path(bot, index_bits[bot_depth:], last_bits[bot_depth:],
                                     lgmask[bot_depth:max_depth] + [0], path[bot_depth:])
        assert mid == p["slotRoot"], "slot root mismatch for sample %d" % (cnt + 1)        # single_cell.circom:71
    return True
