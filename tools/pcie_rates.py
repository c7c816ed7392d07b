"""PCIe-inclusive rates of the host-pointer ABI (never bench.py's `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
rng = np.random.default_rng(1)
n = 1 << 22
x = rng.integers(0, 256, size=(n, 96), dtype=np.uint8); x[:, 31] &= 0x1f; x[:, 63] &= 0x1f; x[:, 95] &= 0x1f
ctx.permute_batch(x[:1024])
t = time.perf_counter(); ctx.permute_batch(x); dt = time.perf_counter() - t
print("cp2_permute_batch host pointers, 2^22 states: %.3f s -> %.3e perm/s (PCIe + pageable copies included)" % (dt, n / dt))
cs, bs, nc = 2048, 65536, 1 << 20
cells = rng.integers(0, 256, size=(nc, cs), dtype=np.uint8)
ctx.slot_trees_host(cells[:1 << 15], 1, cs, bs, 1 << 15).roots()
t = time.perf_counter(); tr = ctx.slot_trees_host(cells, 1, cs, bs, nc); r = tr.roots(); dt = time.perf_counter() - t
print("cp2_slot_trees_build_host, 2 GiB slot from host memory: %.3f s -> %.2f GB/s, %.3e perm/s" % (dt, nc * cs / dt / 1e9, (35 * nc - 1) / dt))
path = "/tmp/cp2_slot0.dat"
cells.tofile(path)
cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=1, cellSize=cs, blockSize=bs, nSlots=1, nCells=nc, nSamples=5, file="/tmp/cp2_slot")
t = time.perf_counter(); ds = ctx.dataset(cfg); r2 = ds.local_roots(); dt = time.perf_counter() - t
print("slot file (page cache warm), 2 GiB: %.3f s -> %.2f GB/s; root matches host build: %s" % (dt, nc * cs / dt / 1e9, bool((r2[0] == r[0]).all())))
os.remove(path)
