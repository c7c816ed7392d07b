"""One rank of the sharded dataset build on a real GPU (child process of tests/rank_helpers.py).

  RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment; argv: <out.json> <gpu index> <config json> <entropy> [roots.npy]
  (roots.npy: rank 0 saves the gathered (n_slots, 32) root array there, for the parent's oracle comparisons)

Uses the PRODUCT path only (distributed.HipBackend -> libcodex_p2.so); collectives over gloo so that several ranks can
share one GPU (RCCL refuses two ranks per device).  The parent compares what is written here with the oracle."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, gpu, cfg_json, entropy = sys.argv[1], int(sys.argv[2]), json.loads(sys.argv[3]), int(sys.argv[4])
    roots_path = sys.argv[5] if len(sys.argv) > 5 else None
    import importlib
    import torch.distributed as dist
    import __graft_entry__ as g
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = g.load_package()
    d = importlib.import_module(g.PKG_NAME + ".distributed")
    ctx = pkg.Context(gpu)
    cfg = pkg.make_config(**cfg_json)
    backend = d.HipBackend(pkg, ctx)
    # CP2_TEST_GATHER=host: host arrays through the collective (round 3's path); default: the roots stay in HBM on both sides
    # of the exchange (cp2_dataset_copy_local_roots_dev -> collective -> cp2_dataset_set_roots_dev)
    where = "cpu" if os.environ.get("CP2_TEST_GATHER") == "host" else "cuda:%d" % gpu
    root, all_roots, (first, count) = d.dataset_root_sharded(backend, cfg, rank, world, dist if world > 1 else None, where)
    if roots_path and rank == 0:
        import numpy as np
        np.save(roots_path, all_roots)
    res = {"rank": rank, "world": world, "first": first, "count": count, "dataset_root_hex": root.tobytes()[::-1].hex(),
           "all_roots_sha256": hashlib.sha256(all_roots.tobytes()).hexdigest(), "inputs": {}}
    for slot in sorted({first, first + count - 1}) if count else []:
        text = backend.dataset.proof_input(slot, entropy).json()
        res["inputs"][str(slot)] = hashlib.sha256(text.encode()).hexdigest()
    maps = open("/proc/self/maps").read()
    res["native_so_loaded"] = "libcodex_p2.so" in maps
    res["oracle_loaded"] = "libp2oracle" in maps
    with open(out_path, "w") as f:
        json.dump(res, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
