"""Real slot files at the nominal slot size: 8 slot files of 8 GiB ("<base><k>.dat", dataset.nim:34) -- written here with the
reference's fake data so that the oracle-only fixture tests/golden/bigslots.json pins the result -- ingested through the pinned
ring (host threads -> pinned buffers -> copy stream -> hash), kept compact, then proved.  Reports the ingestion rate from the
page cache (the files were just written) and, after evicting them, from the box's storage.
Usage: slot_files_nominal.py [directory]"""
import hashlib, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bigslots.json")))
c = gold["config"]
n_slots, n_cells, cs = c["nSlots"], c["nCells"], c["cellSize"]
ctx = pkg.Context(0)
dev = torch.device("cuda", 0)
work = tempfile.mkdtemp(prefix="cp2_slots_", dir=sys.argv[1] if len(sys.argv) > 1 else None)
base = os.path.join(work, "slot")
try:
    t0 = time.time()
    chunk = 1 << 19                                               # cells per chunk: 1 GiB
    buf = torch.empty((chunk, cs), dtype=torch.uint8, device=dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for k in range(n_slots):
        with open("%s%d.dat" % (base, k), "wb") as f:
            for first in range(0, n_cells, chunk):
                ctx.gen_fake_cells_dev(ctx.slot_seed(c["seed"], k), first, chunk, cs, buf.data_ptr())
                torch.cuda.synchronize()
                buf.cpu().numpy().tofile(f)
    ctx.reset_stream()
    del buf
    print("wrote %d slot files of %.0f GiB in %.0f s" % (n_slots, n_cells * cs / 2**30, time.time() - t0), flush=True)
    cfg = pkg.make_config(**dict({k: v for k, v in c.items() if k != "seed"}, file=base))
    ctx.set_keep_trees(2)
    hexroot = lambda a: np.asarray(a, dtype=np.uint8).tobytes()[::-1].hex()      # noqa: E731
    res = {}
    os.environ["CP2_TRACE"] = "1"                                   # the library says how many chunks went by mapping / through the ring
    for label, evict, direct in (("page_cache", False, 0), ("page_cache_mapped", False, 0), ("page_cache_again", False, 0), ("page_cache_mapped_again", False, 0),
                                 ("evicted_buffered", True, 0), ("evicted_o_direct", True, 1)):
        if evict:
            for k in range(n_slots):
                fd = os.open("%s%d.dat" % (base, k), os.O_RDONLY)
                os.fsync(fd)
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                os.close(fd)
        ctx.set_ingest_direct(direct)
        ctx.set_ingest_mapped(1 if "mapped" in label else -1)
        t = time.time()
        ds = ctx.dataset(cfg)
        dt = time.time() - t
        ok = hexroot(ds.root()) == gold["dataset_root_hex"] and [hexroot(r) for r in ds.local_roots()] == gold["slot_roots_hex"]
        res[label] = {"seconds": round(dt, 2), "GB_per_s": round(n_slots * n_cells * cs / dt / 1e9, 2), "roots_equal_oracle_fixture": ok}
        print("%-18s %6.2f s  %6.2f GB/s  slot roots and dataset root equal tests/golden/bigslots.json: %s" % (label, dt, res[label]["GB_per_s"], ok), flush=True)
        if label == "page_cache":
            t = time.time()
            texts = {s: ds.proof_input(s, gold["entropy"]).json() for s in (0, 3, 7)}
            res["proof_inputs"] = {"seconds_for_three": round(time.time() - t, 3),
                                   "equal_oracle_fixture": all(hashlib.sha256(texts[s].encode()).hexdigest() == gold["inputs"][str(s)]["json_sha256"] for s in texts)}
            print("three proof inputs from the compact layers + the touched blocks READ FROM THE FILES: %.3f s, byte-exact vs the fixture: %s" %
                  (res["proof_inputs"]["seconds_for_three"], res["proof_inputs"]["equal_oracle_fixture"]), flush=True)
        ds.free()
    ctx.set_ingest_direct(-1)
    print(json.dumps(res))
    ok_all = all(v.get("roots_equal_oracle_fixture", True) for v in res.values()) and res["proof_inputs"]["equal_oracle_fixture"]
finally:
    shutil.rmtree(work, ignore_errors=True)
sys.exit(0 if ok_all else 1)
