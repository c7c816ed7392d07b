"""Host -> HBM ingestion rate against slot size (is the gap to the kernel's own rate a start-up cost or a per-chunk one?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
cs, bs = 2048, 65536
big = np.empty((1 << 22, cs), dtype=np.uint8)
big[:] = np.arange(cs, dtype=np.uint8)[None, :]
big[:, 0] = (np.arange(1 << 22) & 0xff).astype(np.uint8)
ctx.slot_trees_host(big[:1 << 18], 1, cs, bs, 1 << 18).free()
for log2 in (19, 20, 21, 22):
    nc = 1 << log2
    best = 1e9
    for _ in range(2):
        t = time.perf_counter(); tr = ctx.slot_trees_host(big[:nc], 1, cs, bs, nc); dt = time.perf_counter() - t; tr.free()
        best = min(best, dt)
    d = torch.from_numpy(big[:nc]).cuda()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); tr = ctx.slot_trees_dev(d.data_ptr(), 1, cs, bs, nc); b.record(); torch.cuda.synchronize(); tr.free()
    ctx.reset_stream()
    kms = a.elapsed_time(b)
    del d
    print("slot of %5.1f GiB: host path %.1f ms = %.1f GB/s;  same slot resident in HBM %.1f ms = %.1f GB/s;  difference %.1f ms" %
          (nc * cs / 2**30, best * 1e3, nc * cs / best / 1e9, kms, nc * cs / (kms * 1e-3) / 1e9, best * 1e3 - kms), flush=True)
