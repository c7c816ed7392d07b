"""CPU suite, part 3: the multi-GPU path's host logic under gloo, world_size 2.
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


@pytest.mark.parametrize("n_slots", [5, 8])
def test_sharded_dataset_root_world2(oracle, tmp_path, n_slots):
    import torch.multiprocessing as mp
    C, _ = oracle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, n_slots, str(tmp_path)), nprocs=2, join=True)
    want_roots = np.stack([C.fake_slot_root(C.slot_seed(4242, k), 64, 256, 16, 1) for k in range(n_slots)])
    want = C.merkle_root(want_roots)
    covered = []
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / ("root_%d.npy" % r)), want)
        assert np.array_equal(np.load(tmp_path / ("all_%d.npy" % r)), want_roots)
        f, c = np.load(tmp_path / ("range_%d.npy" % r))
        covered += list(range(f, f + c))
    assert covered == list(range(n_slots))
