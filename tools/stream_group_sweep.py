"""configs[3] (4096 slots x 2^12 cells, 100 samples, every input.json) through the streamed build at several group sizes: how much
of the streamed path's distance from the plain tree build is the tail of each group's hash launch?  Usage: stream_group_sweep.py [groups ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=12, cellSize=2048, blockSize=65536, nSlots=4096, nCells=1 << 12, nSamples=100, seed=12345)
groups = [int(a) for a in sys.argv[1:]] or [0, 240, 192, 144, 96, 288, 0]
t = time.perf_counter(); ds = ctx.dataset(cfg); dt = time.perf_counter() - t
root = ds.root().tobytes(); ds.free()
print("plain tree build (no proof inputs): %.4f s" % dt, flush=True)
t = time.perf_counter(); ds = ctx.dataset(cfg); dt = time.perf_counter() - t; ds.free()
print("plain tree build (no proof inputs): %.4f s" % dt, flush=True)
for grp in groups:
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sd = ctx.dataset_streamed(cfg, 1234567, threads=16, group_slots=grp)
        t1 = time.perf_counter()
        sd.set_roots(None)
        nb = sd.export_streamed(None, threads=16)
        t2 = time.perf_counter()
        ok = sd.root().tobytes() == root
        sd.free()
        print("group_slots %4d: build with bodies %.4f s, total %.4f s -> %.0f witnesses/s, json %d bytes, root ok %s" % (grp, t1 - t0, t2 - t0, 4096 / (t2 - t0), nb, ok), flush=True)
