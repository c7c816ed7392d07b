"""One timing sample of the two hot kernels with the library named by CODEX_P2_LIB (child of tools/ab_kernels.py)."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
ctx.set_stream(st.cuda_stream)
n = 1 << 24
gen = torch.Generator(device=dev).manual_seed(7)
x = torch.randint(0, 256, (n, 96), dtype=torch.uint8, device=dev, generator=gen)
for off in (31, 63, 95):
    x[:, off] &= 0x1F
y = torch.empty_like(x)
for _ in range(2):
    ctx.permute_batch_dev(x.data_ptr(), y.data_ptr(), n)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
ev[0].record(st)
for i in range(6):
    ctx.permute_batch_dev(x.data_ptr(), y.data_ptr(), n)
    ev[i + 1].record(st)
torch.cuda.synchronize()
perm_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(6))
digest = hashlib.sha256(y[::65537].cpu().numpy().tobytes()).hexdigest()[:16]
nc = 1 << 21
cells = x.view(-1)[: nc * 2048 if nc * 2048 <= x.numel() else x.numel()]
nc = cells.numel() // 2048
out = torch.empty((nc, 32), dtype=torch.uint8, device=dev)
ctx.hash_cells_dev(cells.data_ptr(), 2048, nc, out.data_ptr())
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(st)
ctx.hash_cells_dev(cells.data_ptr(), 2048, nc, out.data_ptr())
b.record(st)
torch.cuda.synchronize()
hdig = hashlib.sha256(out[::4099].cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps({"perm_ms_median": perm_ms[len(perm_ms) // 2], "perm_ms_min": perm_ms[0], "hash_ms": a.elapsed_time(b), "hash_cells": nc,
                  "perm_digest": digest, "hash_digest": hdig}))
