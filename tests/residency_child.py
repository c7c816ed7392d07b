"""A fresh process for the residency tests (tests/test_gpu_round5.py): its own environment (the memory cap, the staging size and the
test knobs are read by the library at run time) and an empty allocation ledger.

  residency_child.py <json>     {"what": "single" | "multi" | "streamed" | "cached", "config": {...}, "devices": [0, 0, ...],
                                 "entropy": int, "slots": [..], "cache": path}
prints one JSON object: what every dataset kept (cp2_dataset_keeps_trees), sha256 over the slot roots, the dataset root,
sha256 + length of the input.json of the slots asked for, and the [cp2 trace] lines that talk about stepping down."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    job = json.loads(sys.argv[1])
    pkg = g.load_package()
    cfg = pkg.make_config(**job["config"])
    entropy, slots = job.get("entropy", 1234567), job.get("slots", [])
    res = {"what": job["what"]}
    sha = lambda b: hashlib.sha256(b).hexdigest()
    hexroot = lambda r: r.tobytes()[::-1].hex()
    if job["what"] == "multi":
        m = pkg.Multi(job["devices"])
        ds = m.dataset_streamed(cfg, entropy, threads=8) if job.get("streamed") else m.dataset(cfg, cache=job.get("cache"))
        L = m.L
        modes = []
        for i in range(len(ds.shards())):
            h = L.cp2_multi_dataset_shard(ds.h, i, None, None, None)
            modes.append(L.cp2_dataset_keeps_trees(h) if h else None)
        res.update(modes=modes, shards=ds.shards(), units_per_slot=ds.units_per_slot, gather=m.gather_mode(),
                   slot_roots_sha256=sha(ds.slot_roots().tobytes()), dataset_root_hex=hexroot(ds.root()))
        if job.get("streamed"):
            ds.export_streamed(None, threads=8)
            texts = {s: ds.streamed_json(s) for s in slots}
        else:
            texts = {s: ds.proof_input(s, entropy).json() for s in slots}
        res["inputs"] = {str(s): {"json_sha256": sha(t.encode()), "json_bytes": len(t)} for s, t in texts.items()}
        ds.free()
        m.close()
    else:
        ctx = pkg.Context(0)
        if job["what"] == "streamed":
            ds = ctx.dataset_streamed(cfg, entropy, threads=8)
        else:
            ds = ctx.dataset(cfg, cache=job.get("cache"))
        res.update(modes=[ds.tree_mode], slot_roots_sha256=sha(ds.local_roots().tobytes()))
        ds.set_roots(None)
        res["dataset_root_hex"] = hexroot(ds.root())
        if job["what"] == "streamed":
            ds.export_streamed(None, threads=8)
            texts = {s: ds.streamed_json(s) for s in slots}
        else:
            texts = {s: ds.proof_input(s, entropy).json() for s in slots}
        res["inputs"] = {str(s): {"json_sha256": sha(t.encode()), "json_bytes": len(t)} for s, t in texts.items()}
        ds.free()
        ctx.close()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
