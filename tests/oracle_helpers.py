"""Oracle-side helpers shared by the GPU parity tests and the golden-fixture generators (TEST INFRASTRUCTURE)."""
import numpy as np


def expected_proof_input_fast(C, P, c, slot, entropy, threads=16, slot_roots=None):
    """The oracle's proof input for a fake-data configuration, with every hash done by the C oracle (fast) and the
    indexing / merging / padding / JSON by the Python restatement.  Same result as P.generate_proof_input."""
    cs, bs, nc, ns = c["cellSize"], c["blockSize"], c["nCells"], c["nSlots"]
    cpb = bs // cs
    to_int = lambda layers: [C.array_to_felts(l) for l in layers]   # noqa: E731
    roots = slot_roots if slot_roots is not None else np.stack(
        [C.fake_slot_root(C.slot_seed(c["seed"], s), cs, bs, nc, threads) for s in range(ns)])
    dset = to_int(C.merkle_tree(roots))
    cells = C.gen_fake_cells(C.slot_seed(c["seed"], slot), 0, nc, cs)
    leaves = C.hash_cells(cells, cs, threads=threads)
    mini = [to_int(C.merkle_tree(leaves[b * cpb:(b + 1) * cpb])) for b in range(nc // cpb)]
    big = to_int(C.merkle_tree(np.stack([C.felt_bytes(t[-1][0]) for t in mini])))
    assert big[-1][0] == dset[0][slot]
    e = C.felt_bytes(entropy)
    idx = [C.cell_index(e, C.felt_bytes(big[-1][0]), nc, k) for k in range(1, c["nSamples"] + 1)]
    inputs = []
    for ci in idx:
        prf = P.merge_merkle_proofs(P.merkle_proof(mini[ci // cpb], ci % cpb), P.merkle_proof(big, ci // cpb))
        inputs.append({"cellData": cells[ci].tobytes(), "merkleProof": P.pad_merkle_proof(prf, c["maxDepth"])})
    return {"dataSetRoot": dset[-1][0], "entropy": entropy, "nCells": nc, "nSlots": ns, "slotIndex": slot,
            "slotRoot": big[-1][0], "slotProof": P.pad_merkle_proof(P.merkle_proof(dset, slot), c["maxLog2NSlots"]),
            "proofInputs": inputs, "cellIndices": idx}
