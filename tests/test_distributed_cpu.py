"""CPU suite, part 3: the multi-GPU path's host logic under gloo, world_size 2 and 3.
The HIP backend is replaced by an oracle-backed stand-in (tests may use the oracle) so that what is
exercised here is the sharding, the padded all-gather and the slot ordering of distributed.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    def __init__(self, C):
        self.C = C

    def local_slot_roots(self, cfg, first, count):
        C = self.C
        return np.stack([C.fake_slot_root(C.slot_seed(cfg.seed, first + s), cfg.cell_size, cfg.block_size, cfg.n_cells, 1)
                         for s in range(count)]) if count else np.zeros((0, 32), dtype=np.uint8)

    def dataset_root(self, cfg, all_roots):
        return self.C.merkle_root(all_roots)


def _worker(rank, world, port, n_slots, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.load_package()
    C, _ = g.load_oracle()
    d = importlib.import_module("codex_storage_proofs_circuits_amd.distributed")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    cfg = pkg.make_config(cellSize=64, blockSize=256, nCells=16, nSlots=n_slots, nSamples=2, seed=4242)
    root, all_roots, (first, count) = d.dataset_root_sharded(OracleBackend(C), cfg, rank, world, dist, "cpu")
    np.save(os.path.join(out_dir, "root_%d.npy" % rank), root)
    np.save(os.path.join(out_dir, "all_%d.npy" % rank), all_roots)
    np.save(os.path.join(out_dir, "range_%d.npy" % rank), np.array([first, count]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_slots,world", [(5, 2), (8, 2), (11, 3), (2, 3)])
def test_sharded_dataset_root_gloo(oracle, tmp_path, n_slots, world):
    """Uneven shards (3 + 2; 4 + 4 + 3), and a world larger than the dataset (2 slots on 3 ranks: one rank holds nothing
    and still takes part in the gather)."""
    import torch.multiprocessing as mp
    C, _ = oracle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, n_slots, str(tmp_path)), nprocs=world, join=True)
    want_roots = np.stack([C.fake_slot_root(C.slot_seed(4242, k), 64, 256, 16, 1) for k in range(n_slots)])
    want = C.merkle_root(want_roots)
    covered = []
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("root_%d.npy" % r)), want)
        assert np.array_equal(np.load(tmp_path / ("all_%d.npy" % r)), want_roots)
        f, c = np.load(tmp_path / ("range_%d.npy" % r))
        covered += list(range(f, f + c))
    assert covered == list(range(n_slots))


def test_gather_pads_uneven_shards_at_32768_slots_over_8_ranks():
    """gather_slot_roots' padding arithmetic at configs[4]'s scale without processes: the per-rank buffers an 8-rank all-gather
    would exchange (rows padded to the largest shard) reassemble into slot order, for an even and for uneven slot counts."""
    import importlib
    import __graft_entry__ as g
    g.load_package()
    d = importlib.import_module("codex_storage_proofs_circuits_amd.distributed")

    class FakeDist:
        """all_gather over buffers prepared for every rank up front"""
        def __init__(self, bufs):
            self.bufs = bufs

        def all_gather(self, out, buf):
            for o, b in zip(out, self.bufs):
                o.copy_(b)

    import torch
    for n_slots in (32768, 32767, 32771):
        world = 8
        roots = np.random.default_rng(n_slots).integers(0, 256, size=(n_slots, 32), dtype=np.uint8)
        max_rows = (n_slots + world - 1) // world
        bufs = []
        for r in range(world):
            f, c = d.shard_range(n_slots, r, world)
            b = torch.zeros((max_rows, 32), dtype=torch.uint8)
            b[:c] = torch.from_numpy(roots[f:f + c])
            bufs.append(b)
        for r in (0, 3, 7):
            f, c = d.shard_range(n_slots, r, world)
            got = d.gather_slot_roots(roots[f:f + c], n_slots, r, world, FakeDist(bufs), "cpu")
            assert np.array_equal(got, roots)
