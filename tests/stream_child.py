"""A fresh process for the streamed-build tests of round 6 (tests/test_gpu_round6.py): its own environment -- the A/B knobs
(CP2_STREAM_SERIAL, CP2_STREAM_RAMP), the staging size and the test hooks are read by the library once per process or per context.

  stream_child.py <json>    {"config": {...}, "entropy": int, "group": int, "keep": -1|0|1|2, "threads": int,
                             "ingest": {"threads": n, "ring": n, "chunk_bytes": n, "direct": 0|1, "mapped": 0|1},
                             "later": {"slot": s, "entropy": e}}
prints one JSON object: what the dataset kept, the dataset root, sha256 of the slot roots and the sha256 of EVERY slot's input.json."""
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
    ctx = pkg.Context(0)
    ing = job.get("ingest") or {}
    if ing:
        ctx.set_ingest(ing.get("threads", 0), ing.get("ring", 0), ing.get("chunk_bytes", 0))
        ctx.set_ingest_direct(ing.get("direct", -1))
        ctx.set_ingest_mapped(ing.get("mapped", -1))
    ctx.set_keep_trees(job.get("keep", -1))
    cfg = pkg.make_config(**job["config"])
    ds = ctx.dataset_streamed(cfg, job.get("entropy", 1234567), threads=job.get("threads", 4), group_slots=job.get("group", 0))
    ds.export_streamed(None, threads=job.get("threads", 4))
    sha = lambda b: hashlib.sha256(b).hexdigest()   # noqa: E731
    res = {"mode": ds.tree_mode, "dataset_root_hex": ds.root().tobytes()[::-1].hex(), "slot_roots_sha256": sha(ds.local_roots().tobytes()),
           "json_sha256": [sha(ds.streamed_json(s).encode()) for s in range(job["config"]["nSlots"])]}
    if job.get("later"):         # a proof input for ANOTHER entropy from what the dataset kept (every node / compact layers / roots only)
        res["later_json_sha256"] = sha(ds.proof_input(job["later"]["slot"], job["later"]["entropy"]).json().encode())
    ds.free()
    ctx.close()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
