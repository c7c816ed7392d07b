"""Where does the time go when slots are small?  Dataset build (fake data) for geometries from 32 cells to 2^16 cells per slot,
same total data (2 GiB): seconds, perms/s, and the ratio to the large-slot rate.  Usage: small_slots_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
total_cells = 1 << 20                      # 2 GiB of 2 KiB cells
print("%10s %10s %10s %12s %10s" % ("cells/slot", "slots", "seconds", "perms/s", "launches"))
for log2c in (5, 6, 8, 10, 12, 14, 16, 20):
    nc = 1 << log2c
    ns = total_cells // nc
    cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=max(1, (ns - 1).bit_length()), cellSize=2048, blockSize=65536, nSlots=ns, nCells=nc, nSamples=10, seed=3)
    ctx.dataset(cfg).free()                # warm-up (allocations)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ds = ctx.dataset(cfg)
        ds.set_roots(None)
        dt = time.perf_counter() - t
        ds.free()
        best = min(best, dt)
    perms = ns * (35 * nc - 1) + ns - 1
    layers = 5 + max(1, (nc // 32 - 1).bit_length() if nc > 32 else 1) + max(1, (ns - 1).bit_length())
    print("%10d %10d %10.4f %12.3e %10d" % (nc, ns, best, perms / best, layers), flush=True)
