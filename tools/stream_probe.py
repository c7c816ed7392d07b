import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=12, cellSize=2048, blockSize=65536, nSlots=4096, nCells=4096, nSamples=100, seed=12345)
t = time.perf_counter(); ds = ctx.dataset(cfg); ds.set_roots(None); print("classic trees %.4f" % (time.perf_counter() - t)); ds.free()
for group in (256, 256, 128, 64, 512, 256):
    for threads in (16, 32):
        t0 = time.perf_counter()
        sd = ctx.dataset_streamed(cfg, 1234567, threads=threads, group_slots=group)
        t1 = time.perf_counter()
        sd.set_roots(None); n = sd.export_streamed(None, threads=threads)
        t2 = time.perf_counter()
        print("group %4d threads %2d: build+bodies %.4f  heads %.4f  total %.4f -> %.0f witnesses/s" % (group, threads, t1 - t0, t2 - t1, t2 - t0, 4096 / (t2 - t0)), flush=True)
        sd.free()
