import sys, os, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import __graft_entry__ as g
pkg = g.load_package(); ctx = pkg.Context(0)
cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=12, cellSize=2048, blockSize=65536, nSlots=4096, nCells=1 << 12, nSamples=100, seed=12345)
t=time.perf_counter(); ds = ctx.dataset(cfg); ds.set_roots(None); print("build", time.perf_counter()-t)
for _ in range(2):
    t=time.perf_counter(); pis = ds.proof_inputs(list(range(4096)), 1234567); print("generate", time.perf_counter()-t)
    for p in pis: p.free()
