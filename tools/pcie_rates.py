"""PCIe-inclusive rates of the host-pointer ABI (never bench.py's `value`).  CP2_TRACE=1 shows where
the time of the chunked host-array path goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
rng = np.random.default_rng(1)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 24)     # configs[1]'s 2^24 states: 1.6 GB in, 1.6 GB out
lg = n.bit_length() - 1
x = rng.integers(0, 256, size=(n, 96), dtype=np.uint8); x[:, 31] &= 0x1f; x[:, 63] &= 0x1f; x[:, 95] &= 0x1f
ctx.permute_batch(x[:1 << 21])
y = None
for rep in range(3):
    t = time.perf_counter(); y = ctx.permute_batch(x); dt = time.perf_counter() - t
    print("cp2_permute_batch host arrays, 2^%d states (fresh output array): %.3f s -> %.3e perm/s, %.1f GB/s each way" % (lg, dt, n / dt, n * 96 / dt / 1e9), flush=True)
import ctypes
out = np.empty_like(x); out[:] = 0
for rep in range(3):
    t = time.perf_counter()
    ctx._ck(ctx.L.cp2_permute_batch(ctx.h, ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(out.ctypes.data), n), "permute")
    dt = time.perf_counter() - t
    print("cp2_permute_batch host arrays, 2^%d states (output array already touched): %.3f s -> %.3e perm/s, %.1f GB/s each way" % (lg, dt, n / dt, n * 96 / dt / 1e9), flush=True)
assert np.array_equal(out, y)

# the caller's arrays pinned (torch pinned tensors = hipHostMalloc): no ring, no host copies
import torch
xin = torch.from_numpy(x).pin_memory()
xout = torch.empty_like(xin).pin_memory()
for rep in range(4):
    t = time.perf_counter()
    ctx._ck(ctx.L.cp2_permute_batch(ctx.h, ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(xout.data_ptr()), n), "permute")
    dt = time.perf_counter() - t
    print("cp2_permute_batch PINNED host arrays, 2^%d states: %.3f s -> %.3e perm/s, %.1f GB/s each way" % (lg, dt, n / dt, n * 96 / dt / 1e9), flush=True)
assert np.array_equal(xout.numpy(), y)
print("pinned output equals the pageable path's")
