"""Statistical soak: many random permutations and cells on the GPU against the C oracle (rare-carry bugs hide at
~2^-29 per limb event; 2^28 states exercise ~10^13 limb operations).  Usage: soak.py [log2_states] [seed_base]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
C, P = g.load_oracle()
ctx = pkg.Context(0)
threads = max(1, min(16, len(os.sched_getaffinity(0))))
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
seed_base = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
chunk = 1 << 22
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
t0 = time.time()
bad = 0
for it in range((1 << log2n) // chunk):
    gen = torch.Generator(device="cuda").manual_seed(seed_base + it)
    x = torch.randint(0, 256, (chunk, 96), dtype=torch.uint8, device="cuda", generator=gen)
    if it % 2 == 0:                      # half canonical (< 2^253), half arbitrary 256-bit values
        x[:, 31] &= 0x1F; x[:, 63] &= 0x1F; x[:, 95] &= 0x1F
    if it % 5 == 0:                      # long runs of zero / one bits inside the limbs
        x[:, 4:28] = 0xFF if it % 10 == 0 else 0x00
    y = torch.empty_like(x)
    ctx.permute_batch_dev(x.data_ptr(), y.data_ptr(), chunk)
    torch.cuda.synchronize()
    if not np.array_equal(y.cpu().numpy(), C.permute_batch(x.cpu().numpy(), threads=threads)):
        bad += 1
        print("MISMATCH in chunk", it, flush=True)
    if it % 8 == 7:
        print("permutations: %d x 2^22 states ok=%s (%.0f s)" % (it + 1, bad == 0, time.time() - t0), flush=True)
rng = np.random.default_rng(5 + seed_base)
ncell = 0
for it in range(12):
    cs = int(rng.choice([2048, 2048, 1024, 4096, 512, 31, 62, 100, 2047]))
    n = (1 << 26) // max(cs, 64) // 8
    cells = rng.integers(0, 256, size=(n, cs), dtype=np.uint8)
    if not np.array_equal(ctx.hash_cells(cells, cs), C.hash_cells(cells, cs, threads=threads)):
        bad += 1
        print("CELL MISMATCH cs", cs, flush=True)
    ncell += n
print("soak done (seed base %d): 2^%d permutation states, %d cells, mismatching chunks: %d, %.0f s" % (seed_base, log2n, ncell, bad, time.time() - t0))
sys.exit(1 if bad else 0)
