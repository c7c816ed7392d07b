"""One-off full-size parity: the 8 GiB fake slot (cellSize 2048, nCells 2^22, seed 12345,
slot 0) hashed on the GPU against the multi-threaded C oracle on the host, plus 100 sampled proofs re-derived
with the oracle.  ~1.5e8 permutations on the CPU: about a minute on 16 threads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
C, P = g.load_oracle()
ctx = pkg.Context(0)
n_cells, cs, bs = 1 << 22, 2048, 65536
t = time.perf_counter()
trees = ctx.slot_trees_fake(12345, 0, 1, cs, bs, n_cells)
root = trees.roots()[0]
print("GPU slot root  %s  (%.2f s incl. on-device data generation)" % (root.tobytes()[::-1].hex(), time.perf_counter() - t), flush=True)
threads = max(1, min(16, len(os.sched_getaffinity(0))))
t = time.perf_counter()
want = C.fake_slot_root(C.slot_seed(12345, 0), cs, bs, n_cells, threads)
dt = time.perf_counter() - t
print("CPU slot root  %s  (%.1f s on %d threads, %.3e perm/s)" % (want.tobytes()[::-1].hex(), dt, threads, (35 * n_cells - 1) / dt), flush=True)
assert np.array_equal(root, want), "FULL-SIZE SLOT ROOT MISMATCH"
idx = ctx.cell_indices(pkg.felt_bytes(1234567), root, n_cells, 100)
paths, leaves = trees.paths(0, idx, 32)
r = pkg.array_to_felts(root)[0]
for k, ci in enumerate(idx):
    ci = int(ci)
    cell = C.gen_fake_cell(C.slot_seed(12345, 0), ci, cs)
    leaf = C.array_to_felts(C.hash_bytes(cell))[0]
    assert leaf == pkg.array_to_felts(leaves[k])[0]
    path = pkg.array_to_felts(paths[k])
    bot = P.reconstruct_root({"numberOfLeaves": 32, "leafIndex": ci % 32, "leafValue": leaf, "merklePath": path[:5]})
    top = P.reconstruct_root({"numberOfLeaves": n_cells // 32, "leafIndex": ci // 32, "leafValue": bot, "merklePath": path[5:22]})
    assert top == r and path[22:] == [0] * 10
print("full-size slot root bit-exact; 100 sampled cells + merged paths re-derive it (oracle reconstructRoot)")
