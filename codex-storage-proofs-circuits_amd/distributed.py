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

    def build_local(self, cfg, first, count):
        """The trees of this rank's slots, roots left on the device (the device-to-device gather reads them there)."""
        self.dataset = self.ctx.dataset(cfg, first, count) if count else None
        return self.dataset

    def dataset_root(self, cfg, all_roots):
        if self.dataset is not None:
            self.dataset.set_roots(all_roots)
            return self.dataset.root()
        return self.ctx.merkle_root(all_roots)


def gather_slot_roots(local_roots, n_slots, rank, world, dist=None, device="cpu"):
    """All-gather the per-rank (count, 32) uint8 root arrays into the (n_slots, 32) array in slot order.
    Shards may differ by one row, so rows are padded to the largest shard for the collective.  HOST arrays in, host array
    out: the path of the gloo / CPU tests; on GPUs use gather_slot_roots_dev (no host hop on either side)."""
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


def gather_slot_roots_dev(dataset, ctx, n_slots, rank, world, dist, device):
    """THE exchange step, device to device: the local roots are copied inside HBM (cp2_dataset_copy_local_roots_dev) into this
    rank's row block of one device tensor, ONE all_gather_into_tensor (RCCL over xGMI) fills the other blocks in place, and the
    (n_slots, 32) device tensor that comes back is what cp2_dataset_set_roots_dev takes -- no host copy of a root anywhere
    (round 3 bounced device -> host -> device -> RCCL -> host -> device).  Shards may differ by one row: blocks are padded to
    the largest shard for the collective and the gaps closed on the device."""
    import torch
    max_rows = (n_slots + world - 1) // world
    gath = torch.zeros((world, max_rows, 32), dtype=torch.uint8, device=device)
    torch.cuda.current_stream(device).synchronize()       # the zero fill ran on torch's stream; the copy below runs on the context's
    if dataset is not None:
        dataset.copy_local_roots_dev(gath[rank].data_ptr())
        ctx.sync()                                        # the copy ran on the context's stream; the collective runs on torch's
    if world > 1:
        if dist.get_backend() == "nccl":                  # RCCL: device buffers straight over xGMI
            dist.all_gather_into_tensor(gath.view(world * max_rows, 32), gath[rank].clone())
        else:                                             # rehearsal on one GPU (gloo has no device all-gather): staged by hand
            parts = [torch.empty((max_rows, 32), dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, gath[rank].cpu())
            gath = torch.stack(parts).to(device)
    if n_slots % world == 0:
        all_roots = gath.view(n_slots, 32)
    else:
        all_roots = torch.cat([gath[r, :shard_range(n_slots, r, world)[1]] for r in range(world)], dim=0).contiguous()
    torch.cuda.synchronize(device)                        # the library reads the tensor on its own stream next
    return all_roots


def dataset_root_sharded(backend, cfg, rank, world, dist=None, device="cpu", on_built=None):
    """Returns (dataset_root (32,) uint8, all_roots (n_slots, 32) uint8, (first, count)).
    With a HipBackend on a CUDA/HIP device the gather is device to device; otherwise (the CPU tests' oracle-backed backend,
    gloo rehearsals) it goes through host arrays.  Every rank issues the SAME collective, also a rank that holds no slot
    (world > n_slots): it contributes an empty block and computes the root from the gathered roots.
    on_built(err): called on every rank after its local build and before the collective, with the exception the build raised or
    None -- the caller's chance to agree across ranks that everyone will enter the collective (bench.py: a bounded exchange
    through the rendezvous store) and to raise on all ranks alike otherwise.  Without it a local failure is raised at once."""
    n_slots = int(cfg.n_slots)
    first, count = shard_range(n_slots, rank, world)
    on_device = isinstance(backend, HipBackend) and str(device).startswith("cuda")
    err, local = None, None
    try:
        if on_device:
            backend.build_local(cfg, first, count)
        else:
            local = backend.local_slot_roots(cfg, first, count)
    except Exception as e:
        if on_built is None:
            raise
        err = e
    if on_built is not None:
        on_built(err)
        if err is not None:
            raise err
    if on_device:
        all_dev = gather_slot_roots_dev(backend.dataset, backend.ctx, n_slots, rank, world, dist, device)
        all_roots = all_dev.cpu().numpy()
        if backend.dataset is not None:
            backend.dataset.set_roots_dev(all_dev.data_ptr())
            return backend.dataset.root(), all_roots, (first, count)
        return backend.ctx.merkle_root(all_roots), all_roots, (first, count)      # a rank without slots: the tree over the gathered roots
    all_roots = gather_slot_roots(local, n_slots, rank, world, dist, device)
    return backend.dataset_root(cfg, all_roots), all_roots, (first, count)
