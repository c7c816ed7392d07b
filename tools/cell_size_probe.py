"""k_hash_cells across cell sizes: 2 GiB of resident fake data per size, permutations/s and GB/s (is any size off the 7.5e8 perm/s
the 2048-byte cell reaches?).  Usage: cell_size_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
total = 2 << 30
buf = torch.empty(total, dtype=torch.uint8, device=dev)
print("cell size   cells      perms/cell   ms      perms/s     GB/s")
for cs in (31, 62, 64, 100, 128, 256, 512, 1024, 2048, 4096, 8192, 16384):
    n = total // cs
    ctx.gen_fake_cells_dev(12345, 0, n, cs, buf.data_ptr())
    out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    ctx.hash_cells_dev(buf.data_ptr(), cs, n, out.data_ptr())
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ctx.hash_cells_dev(buf.data_ptr(), cs, n, out.data_ptr())
        b.record(stream)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    nf = (cs + 1 + 30) // 31                 # field elements of a cell (10* padding), + the sponge's own padding to an even count
    perms = (nf + 1 + 1) // 2 if nf % 2 else (nf + 2) // 2
    print("%9d %9d %9d %9.2f %11.3e %8.2f" % (cs, n, perms, best, perms * n / (best * 1e-3), n * cs / (best * 1e-3) / 1e9), flush=True)
    del out
