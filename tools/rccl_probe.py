"""RCCL on this box with world_size 1: communicator creation, uint8 all_gather, float64 MAX all_reduce, and that the
banner RCCL printf()s on stdout (NCCL_DEBUG=VERSION is exported here) can be kept off stdout the way bench.py does."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from bench import _stdout_to_stderr
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
with _stdout_to_stderr():
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    w = torch.zeros(1, device=dev); dist.all_reduce(w); torch.cuda.synchronize()
t = torch.arange(64, dtype=torch.uint8, device=dev).reshape(2, 32)
out = [torch.empty_like(t)]
dist.all_gather(out, t); dist.barrier()
x = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(x, op=dist.ReduceOp.MAX)
torch.cuda.synchronize(); print("rccl world=1 ok", bool(torch.equal(out[0], t)), x.item())
dist.destroy_process_group()
