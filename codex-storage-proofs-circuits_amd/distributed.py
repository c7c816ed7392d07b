"""Multi-GPU sharding of a dataset: one process per GPU, slots are independent until the
dataset-level tree (reference/nim/proof_input/src/gen_input/bn254.nim:41-49), so each rank builds the trees
of a contiguous range of slots with no communication, then ONE all-gather of 32-byte slot roots
(RCCL over xGMI with backend "nccl", gloo on CPU) and every rank builds the identical dataset tree.

The compute backend is injected: the product uses HipBackend (libcodex_p2.so); the CPU tests inject an
oracle-backed stand-in to exercise exactly this sharding / gather / ordering logic under gloo."""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous split; the first (n_items % world) ranks hold one extra item.  Returns (first, count)."""
    base, rem = divmod(n_items, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


class HipBackend:
    """Slot roots and dataset root through the C ABI on this rank's GPU."""

    def __init__(self, pkg, ctx):
        self.pkg, self.ctx = pkg, ctx
        self.dataset = None

    def local_slot_roots(self, cfg, first, count):
        if count == 0:
            return np.zeros((0, 32), dtype=np.uint8)
        self.dataset = self.ctx.dataset(cfg, first, count)
        return self.dataset.local_roots()

    def dataset_root(self, cfg, all_roots):
        if self.dataset is not None:
            self.dataset.set_roots(all_roots)
            return self.dataset.root()
        return self.ctx.merkle_root(all_roots)


def gather_slot_roots(local_roots, n_slots, rank, world, dist=None, device="cpu"):
    """All-gather the per-rank (count, 32) uint8 root arrays into the (n_slots, 32) array in slot order.
    Shards may differ by one row, so rows are padded to the largest shard for the collective."""
    if world == 1 or dist is None:
        assert local_roots.shape[0] == n_slots
        return np.ascontiguousarray(local_roots)
    import torch
    max_rows = (n_slots + world - 1) // world
    buf = torch.zeros((max_rows, 32), dtype=torch.uint8)
    if local_roots.shape[0]:
        buf[:local_roots.shape[0]] = torch.from_numpy(np.ascontiguousarray(local_roots))
    buf = buf.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    rows = []
    for r in range(world):
        _, cnt = shard_range(n_slots, r, world)
        rows.append(out[r][:cnt].cpu().numpy())
    return np.ascontiguousarray(np.concatenate(rows, axis=0))


def dataset_root_sharded(backend, cfg, rank, world, dist=None, device="cpu"):
    """Returns (dataset_root (32,) uint8, all_roots (n_slots, 32) uint8, (first, count))."""
    n_slots = int(cfg.n_slots)
    first, count = shard_range(n_slots, rank, world)
    local = backend.local_slot_roots(cfg, first, count)
    all_roots = gather_slot_roots(local, n_slots, rank, world, dist, device)
    return backend.dataset_root(cfg, all_roots), all_roots, (first, count)
