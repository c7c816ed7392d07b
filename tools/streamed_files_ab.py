"""Streamed proof inputs from REAL slot files beside the same shape from the fake source, one box, one call (SURVEY.md 8 f1 x a14;
slot.nim:57-68, gen_input/bn254.nim:56-64).

Two shapes, both written once as "<base><k>.dat" (dataset.nim:34) with the reference's fake data -- so the file build and the
fake build must give the SAME roots and the SAME input.json texts -- and read back from the page cache:
  small   configs[3]'s scale-down: 4096 slots x 2^12 cells of 2 KiB (32 GiB), 100 samples, maxDepth 32
  big     N slots x 2^22 cells (8 GiB each; N = 16 by default: 128 GiB is what a 270 GiB host allowance holds in the page cache
          beside the pinned rings)
For each: cp2_dataset_build_streamed + cp2_dataset_export_streamed (no directory: texts are formed, not written), alternating
file / fake, best of the repeats; the dataset root, every slot root and a strided set of full texts compared between the two.
Usage: streamed_files_ab.py [directory] [small|big|both] [big_slots] [repeats]"""
import hashlib, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
where = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
which = sys.argv[2] if len(sys.argv) > 2 else "both"
big_slots = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else 16
repeats = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "-" else 3
ENTROPY, SEED, CS = 1234567, 12345, 2048
thr = int(os.environ.get("SFAB_THREADS", "0")) or max(1, min(16, len(os.sched_getaffinity(0))))   # formatting threads (SFAB_THREADS: sweeps)
ctx = pkg.Context(0)
if os.environ.get("SFAB_KEEP"):                          # what the datasets keep of their trees: 1 every node, 2 compact, 0 roots only (default: the library's choice)
    ctx.set_keep_trees(int(os.environ["SFAB_KEEP"]))
dev = torch.device("cuda", 0)


def write_slots(base, n_slots, n_cells):
    t0 = time.time()
    chunk = min(n_cells, 1 << 19)                                  # cells per write: at most 1 GiB
    buf = torch.empty((chunk, CS), dtype=torch.uint8, device=dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for k in range(n_slots):
        with open("%s%d.dat" % (base, k), "wb") as f:
            for first in range(0, n_cells, chunk):
                ctx.gen_fake_cells_dev(ctx.slot_seed(SEED, k), first, chunk, CS, buf.data_ptr())
                torch.cuda.synchronize()
                buf.cpu().numpy().tofile(f)
    ctx.reset_stream()
    del buf
    print("wrote %d slot files of %.3f GiB in %.1f s" % (n_slots, n_cells * CS / 2**30, time.time() - t0), flush=True)


def cpu_seconds():
    import resource
    r = resource.getrusage(resource.RUSAGE_SELF)          # every thread of the process: fill threads, formatting threads, the runtime's own
    return r.ru_utime + r.ru_stime


def one(cfg, n_slots, check_slots):
    c0 = cpu_seconds()
    t = time.perf_counter()
    sd = ctx.dataset_streamed(cfg, ENTROPY, threads=thr)
    tb = time.perf_counter() - t
    sd.set_roots(None)
    nbytes = sd.export_streamed(None, threads=thr)
    dt = time.perf_counter() - t
    cpu = cpu_seconds() - c0
    root = np.asarray(sd.root(), dtype=np.uint8).tobytes()
    roots = hashlib.sha256(np.asarray(sd.local_roots(), dtype=np.uint8).tobytes()).hexdigest()
    texts = {s: hashlib.sha256(sd.streamed_json(s).encode()).hexdigest() for s in check_slots}
    sd.free()
    return {"build_s": tb, "total_s": dt, "cpu_s": cpu, "json_bytes": int(nbytes), "root": root.hex(), "roots_sha": roots, "texts": texts}


def shape(label, n_slots, n_cells, max_log2):
    work = tempfile.mkdtemp(prefix="cp2_sfab_", dir=where)
    base = os.path.join(work, "slot")
    res = {"shape": "%d slots x 2^%d cells of %d B" % (n_slots, n_cells.bit_length() - 1, CS), "threads": thr}
    try:
        write_slots(base, n_slots, n_cells)
        c = dict(maxDepth=32, maxLog2NSlots=max_log2, cellSize=CS, blockSize=65536, nSlots=n_slots, nCells=n_cells, nSamples=100, seed=SEED)
        cfg_fake = pkg.make_config(**c)
        cfg_file = pkg.make_config(**dict({k: v for k, v in c.items() if k != "seed"}, file=base))
        check = sorted(set([0, 1, n_slots // 2, n_slots - 1] + list(range(0, n_slots, max(1, n_slots // 61)))))
        runs = {"file": [], "fake": []}
        for r in range(repeats + 1):                               # the first pair warms pools, rings and the code object
            for src, cfg in (("file", cfg_file), ("fake", cfg_fake)):
                x = one(cfg, n_slots, check)
                runs[src].append(x)
                print("%-5s %-4s run %d: build with bodies %.4f s, total %.4f s -> %.1f witnesses/s, %.2f GB/s of cells, json %d bytes; %.2f CPU-seconds = %.1f cores busy" %
                      (label, src, r, x["build_s"], x["total_s"], n_slots / x["total_s"], n_slots * n_cells * CS / x["total_s"] / 1e9, x["json_bytes"], x["cpu_s"], x["cpu_s"] / x["total_s"]), flush=True)
        # the plain tree build of the same slots (no proof inputs: launches at full occupancy, one layer pass at the end), file and fake
        plain = {"file": [], "fake": []}
        for r in range(3):
            for src, cfg in (("file", cfg_file), ("fake", cfg_fake)):
                t = time.perf_counter()
                ds = ctx.dataset(cfg)
                plain[src].append(time.perf_counter() - t)
                ds.free()
        print("%-5s plain tree build (no proof inputs): file %s s, fake %s s -> file/fake rate %.4f" %
              (label, [round(x, 4) for x in plain["file"]], [round(x, 4) for x in plain["fake"]], min(plain["fake"][1:]) / min(plain["file"][1:])), flush=True)
        res["plain_tree_build_s"] = {k: round(min(v[1:]), 4) for k, v in plain.items()}
        # COLD files (SFAB_COLD=1; a disk-backed directory): every slot file evicted from the page cache (fsync + POSIX_FADV_DONTNEED) before
        # each build; buffered reads, then O_DIRECT straight into the pinned ring -- what the box's storage delivers, not the pipe
        if os.environ.get("SFAB_COLD"):
            def evict():
                for k in range(n_slots):
                    fd = os.open("%s%d.dat" % (base, k), os.O_RDONLY)
                    os.fsync(fd)
                    os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                    os.close(fd)
            cold = {}
            for how, direct in (("buffered", 0), ("o_direct", 1), ("buffered_again", 0), ("o_direct_again", 1)):
                evict()
                ctx.set_ingest_direct(direct)
                x = one(cfg_file, n_slots, check)
                cold[how] = round(x["total_s"], 4)
                ok_cold = x["root"] == runs["fake"][-1]["root"] and x["texts"] == runs["fake"][-1]["texts"]
                print("%-5s COLD files, %s: total %.4f s -> %.1f witnesses/s, %.2f GB/s of cells; texts equal the fake source's: %s" %
                      (label, how, x["total_s"], n_slots / x["total_s"], n_slots * n_cells * CS / x["total_s"] / 1e9, ok_cold), flush=True)
            ctx.set_ingest_direct(-1)
            res["cold_files_total_s"] = cold
        same = all(a["root"] == b["root"] and a["roots_sha"] == b["roots_sha"] and a["texts"] == b["texts"] and a["json_bytes"] == b["json_bytes"]
                   for a in runs["file"] for b in runs["fake"])
        best = {s: min(x["total_s"] for x in runs[s][1:]) for s in runs}
        res.update({"best_total_s": {s: round(v, 4) for s, v in best.items()},
                    "witnesses_per_s": {s: round(n_slots / v, 1) for s, v in best.items()},
                    "cells_GB_per_s": {s: round(n_slots * n_cells * CS / v / 1e9, 2) for s, v in best.items()},
                    "file_over_fake": round(best["fake"] / best["file"], 4),
                    "cpu_seconds_per_run": {s: round(min(x["cpu_s"] for x in runs[s][1:]), 2) for s in runs},
                    "roots_and_%d_texts_identical_file_vs_fake" % len(check): same})
        print("%-5s file/fake witnesses/s = %.4f; dataset root, slot roots and %d full texts identical: %s" % (label, res["file_over_fake"], len(check), same), flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return res, same


out, ok = {}, True
if which in ("small", "both"):
    out["small"], s = shape("small", 4096, 1 << 12, 12)
    ok = ok and s
if which in ("big", "both"):
    out["big"], s = shape("big", big_slots, 1 << 22, max(1, (big_slots - 1).bit_length()))
    ok = ok and s
print(json.dumps(out))
sys.exit(0 if ok else 1)
