"""Quick GPU parity + timing probe (development aid; the real tests are in tests/)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
C, P = g.load_oracle()
ctx = pkg.Context(0)
out = ctx.permute_batch(pkg.felts_to_array([0, 1, 2]).reshape(1, 96))
print("KAT", tuple(pkg.array_to_felts(out)) == P.permutation((0, 1, 2)), [hex(v) for v in pkg.array_to_felts(out)])
rng = np.random.default_rng(1)
x = rng.integers(0, 256, size=(1 << 12, 96), dtype=np.uint8)
got = ctx.permute_batch(x); want = C.permute_batch(x, 8)
print("perm random 4096 (non-canonical inputs)", np.array_equal(got, want))
for cs in (128, 256, 2048, 100, 31, 62, 1):
    cells = rng.integers(0, 256, size=(130, cs), dtype=np.uint8)
    print("hash_cells", cs, np.array_equal(ctx.hash_cells(cells, cs), C.hash_cells(cells, cs, 8)))
for n in (1, 2, 3, 5, 8, 33, 100):
    lv = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); lv[:, 31] &= 0x1f
    a = ctx.merkle_tree(lv); b = C.merkle_tree(lv)
    print("merkle", n, len(a) == len(b) and all(np.array_equal(p, q) for p, q in zip(a, b)))
g.smoke()
import torch
n = 1 << 22
t_in = torch.randint(0, 256, (n, 96), dtype=torch.uint8, device="cuda")
t_out = torch.empty_like(t_in)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(2):
    ctx.permute_batch_dev(t_in.data_ptr(), t_out.data_ptr(), n)
torch.cuda.synchronize()
t = time.time()
for _ in range(3):
    ctx.permute_batch_dev(t_in.data_ptr(), t_out.data_ptr(), n)
torch.cuda.synchronize()
dt = (time.time() - t) / 3
print("permute_batch 2^22: %.3f ms  %.3e perm/s" % (dt * 1e3, n / dt))
