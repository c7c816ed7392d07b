"""bench.py's extra legs (everything after the timed region): configs[2]-[4] of BASELINE.json at their stated sizes or SURVEY.md
8(d)'s scale-downs, ingestion, the drop-in's own workload, the CPU baseline.  Each returns a dict that bench.py's leg runner merges
into the JSON line's `extra`; none of them is the headline.  bench.py's docstring describes what each reports."""
import glob
import json
import os
import shutil
import subprocess
import sys
import time

from bench_orchestration import BENCH_PY, LEG_WORST_S, _stdout_to_stderr, inject  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBPS = 8000.0             # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def uniform_felts_device(torch, dev, m, gen):
    """(m, 32) uint8 on `dev`: canonical little-endian field elements, uniform in [0, r) by rejection from 254-bit
    candidates.  Acceptance probability r / 2^254 = 0.756."""
    r_bytes = torch.tensor(list(R_MOD.to_bytes(32, "little")), dtype=torch.uint8, device=dev)

    def candidates(k):
        c = torch.randint(0, 256, (k, 32), dtype=torch.uint8, device=dev, generator=gen)
        c[:, 31] &= 0x3F
        return c

    def below_r(c):
        lt = torch.zeros(c.shape[0], dtype=torch.bool, device=dev)
        eq = torch.ones(c.shape[0], dtype=torch.bool, device=dev)
        for b in range(31, -1, -1):
            col = c[:, b]
            lt |= eq & (col < r_bytes[b])
            eq &= col == r_bytes[b]
        return lt

    out = candidates(m)
    bad = (~below_r(out)).nonzero().flatten()
    while bad.numel():
        c = candidates(bad.numel())
        out[bad] = c
        bad = bad[~below_r(c)]
    return out


def newest_profile(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return fs[-1] if fs else None


def host_threads():
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))   # a one-GPU box's CPU share is 16 cores
    except AttributeError:
        return max(1, min(16, os.cpu_count() or 1))


def slot_root_leg(torch, ctx, pkg, dev, stream):
    """Config 3: one 8 GiB fake slot resident in HBM -> cell hashes (34 perms/cell) -> block + slot trees.
    Also times k_hash_cells alone over the same slot (HIP events on the launch stream) for its own roofline block."""
    n_cells, cs, bs = 1 << 22, 2048, 65536
    buf = torch.empty((n_cells, cs), dtype=torch.uint8, device=dev)
    leaves = torch.empty((n_cells, 32), dtype=torch.uint8, device=dev)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record(stream)
    ctx.gen_fake_cells_dev(ctx.slot_seed(12345, 0), 0, n_cells, cs, buf.data_ptr())
    e[1].record(stream)
    trees = ctx.slot_trees_dev(buf.data_ptr(), 1, cs, bs, n_cells)     # warm-up pass
    torch.cuda.synchronize()
    trees.free()
    e[2].record(stream)
    trees = ctx.slot_trees_dev(buf.data_ptr(), 1, cs, bs, n_cells)
    e[3].record(stream)
    torch.cuda.synchronize()
    root = trees.roots()[0]
    trees.free()
    hash_ms = []
    for _ in range(3):                                                 # the hash kernel alone: one launch over the whole slot
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ctx.hash_cells_dev(buf.data_ptr(), cs, n_cells, leaves.data_ptr())
        b.record(stream)
        torch.cuda.synchronize()
        hash_ms.append(a.elapsed_time(b))
    gen_ms, build_ms = e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])
    perms = 35 * n_cells - 1
    alg_bytes = n_cells * cs + 2 * 32 * n_cells      # cells read once, leaf layer written and read back
    del buf, leaves
    gold = None
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["config3"]["slot_root_hex"]
    except Exception:
        pass
    root_hex = root.tobytes()[::-1].hex()
    leg = {"slot_root": {"workload": "configs[2]: cellSize=2048, nCells=2^22 (8 GiB) sponge+tree, 1 GPU",
                         "build_ms": round(build_ms, 2), "perms": perms, "perms_per_s": perms / (build_ms * 1e-3),
                         "algorithmic_GBps": round(alg_bytes / (build_ms * 1e-3) / 1e9, 2),
                         "fake_data_gen_ms": round(gen_ms, 2), "slot_root_hex": root_hex,
                         "equals_oracle_fixture": (root_hex == gold) if gold else None}}
    # k_hash_cells against the HBM roof: algorithmic bytes of one launch = the cells read once + the digests written
    h_alg = n_cells * cs + 32 * n_cells
    h_avg = sum(hash_ms) / len(hash_ms)
    h_ach = h_alg / (h_avg * 1e-3) / 1e9
    traffic, source = None, None
    tpath = newest_profile("r*_hash_cells_traffic.json")
    if tpath:
        try:
            prof = json.load(open(tpath))
            traffic = prof["hbm_read_bytes_per_launch"] + prof["hbm_write_bytes_per_launch"]
            source = "%s (rocprofv3 --pmc, separate passes; not measured in this run)" % os.path.relpath(tpath, ROOT)
        except Exception:
            traffic = None
    h_valu = None
    if tpath:
        try:
            h_valu = json.load(open(tpath)).get("valu_issue")      # the issue-side record of THIS kernel: one PMC pass + its own ISA (tools/profile_summarize.py)
            if h_valu:
                h_valu = dict(h_valu, source=os.path.relpath(tpath, ROOT))
        except Exception:
            h_valu = None
    roof = {"bound": "hbm", "achieved": round(h_ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(h_ach / HBM_PEAK_GBPS, 5),
            "traffic": traffic, "traffic_source": source, "kernel": "k_hash_cells", "avg_launch_ms": round(h_avg, 3),
            "launch_ms_min_max": [round(min(hash_ms), 3), round(max(hash_ms), 3)], "algorithmic_bytes_per_launch": h_alg,
            "perms_per_launch": 34 * n_cells, "perms_per_s": 34 * n_cells / (h_avg * 1e-3),
            "note": "one launch over the 8 GiB slot of configs[2] (2^22 cells x 2048 B read, 2^22 x 32 B written); VALU-issue bound like "
                    "k_permute_batch: 34 permutations per 2080 B"}
    if h_valu:
        roof["valu_issue"] = h_valu
    return leg, roof


def witness_leg(torch, ctx, pkg):
    """Config 4 (the metric's second half): nSamples=100, maxDepth=32, 4096 slots batched on one GPU.
    4096 x 8 GiB does not fit HBM, so nCells = 2^12 per slot (8 MiB), 32 GiB of fake data
    generated and hashed on the device; one witness = one SlotProofInput serialised as input.json.
    Classic: build every tree, then generate + serialise (pipelined in batches).  Streamed: the same work as one
    pipeline in which the proof inputs of finished slots are produced while later slots are still hashing."""
    n_slots, n_cells = 4096, 1 << 12
    cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=12, cellSize=2048, blockSize=65536, nSlots=n_slots, nCells=n_cells,
                          nSamples=100, seed=12345)
    threads = host_threads()
    ctx.reset_stream()
    torch.cuda.synchronize()
    res = {}
    # ---- classic, twice: the first pass of a context also allocates its device staging (two 2 GiB buffers), the node
    # buffer and the pinned landing zones, which the context keeps
    classic = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ds = ctx.dataset(cfg)                 # every slot tree, built once (sync on return)
        ds.set_roots(None)                    # dataset tree over the 4096 slot roots
        t1 = time.perf_counter()
        nbytes = ds.export_proof_inputs(list(range(n_slots)), 1234567, None, threads=threads, batch=512)
        t2 = time.perf_counter()
        root_hex = ds.root().tobytes()[::-1].hex()
        ds.free()
        classic.append((t1 - t0, t2 - t1))
    # witnesses/s WITHOUT the JSON serialisation as well: all trees, then every SlotProofInput as an object
    # (sampling, paths, cells downloaded into pinned memory; accessors only, no text)
    torch.cuda.synchronize()
    o0 = time.perf_counter()
    ds = ctx.dataset(cfg)
    ds.set_roots(None)
    pis = ds.proof_inputs(list(range(n_slots)), 1234567)
    o1 = time.perf_counter()
    for p_ in pis:
        p_.free()
    del pis
    ds.free()
    # a node proves every slot it holds each period: all 4096 proof inputs (with JSON) for a NEW entropy from a dataset that keeps its
    # trees compact (block roots and up, 1/32 of the nodes): batched passes over the touched blocks
    compact = None
    ctx.set_keep_trees(2)
    try:
        c0 = time.perf_counter()
        cds = ctx.dataset(cfg)
        cds.set_roots(None)
        c1 = time.perf_counter()
        cbytes = cds.export_proof_inputs(list(range(n_slots)), 7654321, None, threads=threads, batch=1024)
        c2 = time.perf_counter()
        compact = {"build_s": round(c1 - c0, 4), "all_proof_inputs_new_entropy_s": round(c2 - c1, 4), "witnesses_per_s_from_compact_layers": n_slots / (c2 - c1),
                   "json_bytes": cbytes, "device_bytes_kept_per_slot": 2 * (n_cells // 32) * 32}
        cds.free()
    finally:
        ctx.set_keep_trees(-1)
    best_classic = min(classic, key=lambda c: c[0] + c[1])     # the components of ONE run: the one with the smallest total
    t0, t1, t2 = 0.0, best_classic[0], best_classic[0] + best_classic[1]
    # ---- streamed, twice: the first pass also pays for the pinned staging (hipHostMalloc), which the context keeps
    runs = []
    for _ in range(2):
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        sd = ctx.dataset_streamed(cfg, 1234567, threads=threads)
        s1 = time.perf_counter()
        sd.set_roots(None)
        nb2 = sd.export_streamed(None, threads=threads)
        s2 = time.perf_counter()
        assert nb2 == nbytes and sd.root().tobytes()[::-1].hex() == root_hex
        sd.free()
        runs.append({"build_with_bodies_s": round(s1 - s0, 4), "dataset_tree_and_heads_s": round(s2 - s1, 4), "total_s": round(s2 - s0, 4)})
    perms = n_slots * (35 * n_cells - 1) + (n_slots - 1) + 200 * n_slots
    gold = None
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["config4"]["dataset_root_hex"]
    except Exception:
        pass
    best = min(r["total_s"] for r in runs)
    res["witnesses"] = {"workload": "configs[3]: nSamples=100, maxDepth=32, 4096 slots x 2^12 cells (32 GiB fake data) batched, 1 GPU",
                        "json_threads": threads, "json_bytes": nbytes,
                        "classic": {"build_trees_s": round(t1 - t0, 4), "pipelined_generate_and_json_s": round(t2 - t1, 4),
                                    "witnesses_per_s_with_json": n_slots / (t2 - t0),
                                    "witnesses_per_s_with_json_first_run": n_slots / (classic[0][0] + classic[0][1]),
                                    "runs_trees_then_export_s": [[round(a, 4), round(b, 4)] for a, b in classic]},
                        "streamed_runs": runs,
                        "witnesses_per_s_with_json": n_slots / best,
                        "witnesses_per_s_without_json": n_slots / (o1 - o0), "trees_and_objects_s": round(o1 - o0, 4),
                        "new_entropy_on_built_trees": {"every_node_resident_s": round(t2 - t1, 4), "witnesses_per_s_every_node_resident": n_slots / (t2 - t1),
                                                       "compact": compact},
                        "witnesses_per_s_with_json_first_run": n_slots / runs[0]["total_s"],
                        "perms": perms, "perms_per_s_build": (perms - 200 * n_slots) / (t1 - t0),
                        "dataset_root_hex": root_hex, "equals_oracle_fixture": (root_hex == gold) if gold else None}
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    return res


def ingest_leg(torch, ctx, pkg, dev):
    """Real (non-fake) slots: host memory and page-cache files -> pinned ring -> HBM -> hash.  Rates against the pinned
    hipMemcpyAsync H2D peak measured here and against the rate the hash kernel sustains from HBM."""
    import numpy as np
    ctx.reset_stream()
    cs, bs, nc = 2048, 65536, 1 << 21                     # one 4 GiB slot
    nbytes = nc * cs
    # pinned H2D peak (torch pinned tensor -> device, 1 GiB, best of 4)
    src = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    best = 0.0
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dst.copy_(src, non_blocking=True)
        b.record()
        torch.cuda.synchronize()
        best = max(best, (1 << 30) / (a.elapsed_time(b) * 1e-3) / 1e9)
    del src, dst
    # kernel rate from HBM for the same slot
    d = torch.empty((nc, cs), dtype=torch.uint8, device=dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.gen_fake_cells_dev(ctx.slot_seed(1, 0), 0, nc, cs, d.data_ptr())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    tr = ctx.slot_trees_dev(d.data_ptr(), 1, cs, bs, nc)
    b.record()
    torch.cuda.synchronize()
    kernel_gbps = nbytes / (a.elapsed_time(b) * 1e-3) / 1e9
    want = tr.roots()[0].copy()
    tr.free()
    cells = d.cpu().numpy()
    del d
    ctx.reset_stream()
    import shutil
    import tempfile
    tmpdir = tempfile.mkdtemp(prefix="cp2_bench_")
    path_base = os.path.join(tmpdir, "slot")
    table = []
    try:
        cells.tofile(path_base + "0.dat")
        cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=1, cellSize=cs, blockSize=bs, nSlots=1, nCells=nc, nSamples=5, file=path_base)
        ingest_table(ctx, pkg, np, cells, cfg, cs, bs, nc, nbytes, want, table)
        cold = cold_file_rates(ctx, np, path_base + "0.dat", cfg, nbytes, want)
    finally:                      # whatever happened: the 4 GiB file goes, the context's ingestion knobs go back to their defaults
        shutil.rmtree(tmpdir, ignore_errors=True)
        ctx.set_ingest(0, 0, 0)
    bh = max(r["host_pointer_GBps"] for r in table)
    bf = max(r["page_cache_file_GBps"] for r in table)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    return {"ingest": {"workload": "one 4 GiB slot (cellSize 2048) from host memory / from a page-cache-warm slot file through the pinned ring (mapped: the same file with cp2_set_ingest_mapped(1), resident chunks uploaded from registered windows of its mapping, no host thread copying)",
                       "pinned_h2d_peak_GBps": round(best, 2), "hash_from_hbm_GBps": round(kernel_gbps, 2), "by_fill_threads": table,
                       "best_host_pointer_GBps": bh, "best_page_cache_file_GBps": bf,
                       "best_page_cache_file_mapped_GBps": max(r["page_cache_file_mapped_GBps"] for r in table),
                       "host_pointer_frac_of_h2d_peak": round(bh / best, 3), "file_frac_of_h2d_peak": round(bf / best, 3),
                       "host_pointer_frac_of_kernel_rate": round(bh / kernel_gbps, 3), "cold_file": cold}}


def witnesses_from_files_leg(torch, ctx, pkg, dev):
    """Config 4 from REAL slot files (SURVEY.md 8 f1 x a14; slot.nim:57-68, gen_input/bn254.nim:56-64): the 4096 slots of the
    `witnesses` leg written once as "<base><k>.dat" (8 MiB each, the reference's fake data, so roots and texts must equal the fake
    build's) and read back from the page cache through the multi-file ingestion pipe, streamed build + every input.json formed.
    tools/streamed_files_ab.py is the same measurement with its fake-source twin alternating on one box."""
    import numpy as np
    import shutil
    import tempfile
    n_slots, n_cells, cs = 4096, 1 << 12, 2048
    threads = host_threads()
    where = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (48 << 30) else None   # tmpfs pages ARE page cache; else $TMPDIR
    if where is None and shutil.disk_usage(tempfile.gettempdir()).free < (40 << 30):
        return {"witnesses_from_files": {"skipped": "no 40 GiB of file space for 4096 slot files of 8 MiB"}}
    work = tempfile.mkdtemp(prefix="cp2_bench_wff_", dir=where)
    base = os.path.join(work, "slot")
    try:
        t0 = time.perf_counter()
        per = 256                                                     # slots generated per launch: 2 GiB
        buf = torch.empty((per * n_cells, cs), dtype=torch.uint8, device=dev)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for k0 in range(0, n_slots, per):
            for j in range(per):
                ctx.gen_fake_cells_dev(ctx.slot_seed(12345, k0 + j), 0, n_cells, cs, buf[j * n_cells:].data_ptr())
            torch.cuda.synchronize()
            host = buf.cpu().numpy()
            for j in range(per):
                host[j * n_cells:(j + 1) * n_cells].tofile("%s%d.dat" % (base, k0 + j))
        del buf, host
        ctx.reset_stream()
        torch.cuda.synchronize()
        write_s = time.perf_counter() - t0
        cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=12, cellSize=cs, blockSize=65536, nSlots=n_slots, nCells=n_cells, nSamples=100, file=base)
        runs = []
        root_hex = None
        for _ in range(3):
            s0 = time.perf_counter()
            sd = ctx.dataset_streamed(cfg, 1234567, threads=threads)
            s1 = time.perf_counter()
            sd.set_roots(None)
            nb = sd.export_streamed(None, threads=threads)
            s2 = time.perf_counter()
            root_hex = sd.root().tobytes()[::-1].hex()
            sd.free()
            runs.append({"build_with_bodies_s": round(s1 - s0, 4), "total_s": round(s2 - s0, 4), "json_bytes": int(nb)})
        gold = None
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["config4"]["dataset_root_hex"]
        except Exception:
            pass
        best = min(r["total_s"] for r in runs[1:])
        return {"witnesses_from_files": {"workload": "configs[3] from slot files: 4096 files of 8 MiB (2^12 cells of 2048 B) in the page cache (%s), nSamples=100, maxDepth=32, streamed build, every input.json formed" % (where or tempfile.gettempdir()),
                                         "json_threads": threads, "files_written_s": round(write_s, 1), "streamed_runs": runs,
                                         "witnesses_per_s_with_json": n_slots / best, "cells_GBps": round(n_slots * n_cells * cs / best / 1e9, 2),
                                         "dataset_root_hex": root_hex, "equals_oracle_fixture": (root_hex == gold) if gold else None}}
    finally:
        shutil.rmtree(work, ignore_errors=True)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)


def cold_file_rates(ctx, np, path, cfg, nbytes, want):
    """The same slot file when it is NOT in the page cache (fsync + POSIX_FADV_DONTNEED evicts it; no root needed): what the
    box's storage delivers through buffered reads and through O_DIRECT reads straight into the pinned ring."""
    out = {"note": "file evicted from the page cache before each build (fsync + posix_fadvise DONTNEED); storage-bound, box dependent"}
    try:
        for direct, name in ((0, "buffered_GBps"), (1, "o_direct_GBps")):
            fd = os.open(path, os.O_RDONLY)
            os.fsync(fd)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            os.close(fd)
            ctx.set_ingest_direct(direct)
            t = time.perf_counter()
            ds = ctx.dataset(cfg)
            dt = time.perf_counter() - t
            ok = bool(np.array_equal(ds.local_roots()[0], want))
            ds.free()
            out[name] = round(nbytes / dt / 1e9, 2)
            out[name.replace("_GBps", "_root_ok")] = ok
    finally:
        ctx.set_ingest_direct(-1)
    return out


def ingest_table(ctx, pkg, np, cells, cfg, cs, bs, nc, nbytes, want, table):
    for threads, chunk_mb in ((8, 64), (8, 192), (4, 384), (8, 384), (16, 384)):
        ctx.set_ingest(threads, 3, chunk_mb << 20)
        warm = min(nc, (chunk_mb << 20) // cs)
        ctx.slot_trees_host(cells[:warm], 1, cs, bs, warm).free()               # pinned ring of this size allocated outside the timing
        t = time.perf_counter()
        trh = ctx.slot_trees_host(cells, 1, cs, bs, nc)
        dt_h = time.perf_counter() - t
        ok_h = bool(np.array_equal(trh.roots()[0], want))
        trh.free()
        t = time.perf_counter()
        ds = ctx.dataset(cfg)
        dt_f = time.perf_counter() - t
        ok_f = bool(np.array_equal(ds.local_roots()[0], want))
        ds.free()
        # the same file with mapped ingestion switched on: resident chunks uploaded from registered windows of the file's mapping
        ctx.set_ingest_mapped(1)
        try:
            t = time.perf_counter()
            ds = ctx.dataset(cfg)
            dt_r = time.perf_counter() - t
            ok_f = ok_f and bool(np.array_equal(ds.local_roots()[0], want))
            ds.free()
        finally:
            ctx.set_ingest_mapped(-1)
        table.append({"fill_threads": threads, "chunk_MiB": chunk_mb, "host_pointer_GBps": round(nbytes / dt_h / 1e9, 2), "page_cache_file_GBps": round(nbytes / dt_f / 1e9, 2),
                      "page_cache_file_mapped_GBps": round(nbytes / dt_r / 1e9, 2), "roots_match_device_build": ok_h and ok_f})


def dataset_leg(torch, bdist, coord, ctx, pkg, dev, rank, world):
    """Config 5's shape at SURVEY.md 8(d)'s stated scale-down: 32 768 slots (maxLog2NSlots = 15) of 2^12 cells x 2048 B (256 GiB
    of fake data, generated and hashed on the devices) sharded over the ranks in contiguous ranges; each rank builds its slot
    trees with no communication, ONE all-gather of the 32-byte slot roots (RCCL over xGMI; none at N = 1), the 15-level dataset
    tree on every rank, one proof input (slotProof of depth 15) for the first slot of every rank.  Strong scaling: the 32 768
    slots are fixed, every rank must end with the same dataset root (and with the oracle's, when the committed fixture
    tests/golden/config5.json carries this shape).
    Ranks start together (the leg's go decision, a bounded store wait); `seconds` is the MAX over ranks of each rank's own
    start-to-finish time.  The only collective is the gather itself (bdist: bounded); everything else the ranks tell each other
    goes through the store (coord), so a rank that fails or dies is named within seconds."""
    import importlib
    d = importlib.import_module("codex_storage_proofs_circuits_amd.distributed")
    n_slots, n_cells = 32768, 1 << 12
    cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=15, cellSize=2048, blockSize=65536, nSlots=n_slots, nCells=n_cells,
                          nSamples=100, seed=12345)
    ctx.reset_stream()
    torch.cuda.synchronize()
    if inject("rank_exit"):
        os._exit(3)                                                  # (rehearsal) this rank dies at the start of the leg
    if inject("rank_hang"):
        time.sleep(1e6)                                              # (rehearsal) ... or hangs there
    t0 = time.perf_counter()
    backend = d.HipBackend(pkg, ctx)
    first, count = d.shard_range(n_slots, rank, world)
    on_device = dev.type == "cuda"                                   # RCCL: the roots never leave HBM; gloo rehearsal: host arrays
    err, local = None, None
    try:
        try:
            if on_device:
                backend.build_local(cfg, first, count)               # this rank's slot trees: no communication
            else:
                local = backend.local_slot_roots(cfg, first, count)
        except Exception as e:
            err = e
        # every rank reaches the collective or none does: a rank whose build failed says so in the store first
        coord.all_ok("dataset/built", err)
        # THE exchange step: device to device (copy inside HBM -> all_gather_into_tensor over RCCL/xGMI -> cp2_dataset_set_roots_dev)
        if on_device:
            all_dev = d.gather_slot_roots_dev(backend.dataset, ctx, n_slots, rank, world, bdist if world > 1 else None, dev)
            backend.dataset.set_roots_dev(all_dev.data_ptr())        # 15-level dataset tree on every rank
            root = backend.dataset.root()
        else:
            all_roots = d.gather_slot_roots(local, n_slots, rank, world, bdist if world > 1 else None, dev)
            root = backend.dataset_root(cfg, all_roots)
        t1 = time.perf_counter()
        text = backend.dataset.proof_input(first, 1234567).json()    # a proof input for one of this rank's own slots
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        if backend.dataset is not None:
            backend.dataset.free()
        ctx.trim()
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    root_hex = root.tobytes()[::-1].hex()
    # what every rank ended with, through the store: its root, its time
    said = coord.exchange("dataset/result", json.dumps({"root": root_hex, "s": dt, "tree_s": t1 - t0}))
    results = {r: json.loads(v) for r, v in said.items()}
    same = len(results) == world and all(v["root"] == root_hex for v in results.values())
    dt_max = max(v["s"] for v in results.values())
    tree_max = max(v["tree_s"] for v in results.values())
    perms = n_slots * (35 * n_cells - 1) + (n_slots - 1) * world + 200 * world
    gold = None
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "config5.json")))["scaled"]["dataset_root_hex"]
    except Exception:
        pass
    return {"dataset": {"workload": "configs[4] shape, SURVEY.md 8(d) scale-down: 32768 slots x 2^12 cells x 2048 B sharded over %d GPU(s) "
                                    "(%d slots on rank 0), one gather of slot roots -> 15-level dataset tree on every rank, one proof input per rank"
                                    % (world, count),
                        "exchange": ("device to device (all_gather_into_tensor, %s)" % bdist.get_backend() if world > 1 and on_device else
                                     "none (one rank)" if world == 1 else "host arrays (%s rehearsal)" % bdist.get_backend()),
                        "scaling": "strong", "seconds": round(dt_max, 4), "seconds_is": "max over ranks of each rank's own start-to-finish time",
                        "roots_and_dataset_tree_s": round(tree_max, 4), "perms_per_s": perms / dt_max,
                        "slots_per_s": n_slots / dt_max, "all_ranks_agree": bool(same), "ranks": world, "ranks_reporting": len(results),
                        "per_rank_seconds": {str(r): round(v["s"], 4) for r, v in sorted(results.items())},
                        "proof_input_json_bytes": len(text), "dataset_root_hex": root_hex,
                        "equals_oracle_fixture": (root_hex == gold) if gold else None}}


def cpu_baseline(C, np, torch, dev):
    """The oracle timed on this box's host cores on a bounded sample of the same workload (config 2 shape): states from the
    SAME generator as the GPU leg (uniform_felts_device: every element uniform in [0, r) by rejection), copied to the host."""
    cores = host_threads()
    n1 = 1 << 17
    gen = torch.Generator(device=dev).manual_seed(0xC0DE)
    x = uniform_felts_device(torch, dev, 3 * n1, gen).reshape(n1, 96).cpu().numpy()
    t = time.perf_counter()
    C.permute_batch(x, threads=1)
    single = n1 / (time.perf_counter() - t)
    nm = min(1 << 21, n1 * 2 * cores)
    xm = np.tile(x, (nm // n1, 1))
    t = time.perf_counter()
    C.permute_batch(xm, threads=cores)
    multi = nm / (time.perf_counter() - t)
    # config 3 scaled to 2^16 cells (a 128 MiB fake slot -> slot root: generation, sponge, block and slot trees)
    nc3 = 1 << 16
    t = time.perf_counter()
    C.fake_slot_root(C.slot_seed(12345, 0), 2048, 65536, nc3, cores)
    slot_s = time.perf_counter() - t
    return {"value": multi, "unit": "permutations/s", "cores": cores, "kind": "port",
            "sample": "C oracle (oracle/p2_oracle.c, 4x64-bit Montgomery; NOT the Nim reference binary, which cannot be built here): "
                      "%d states on %d threads; single-thread rate on %d states reported beside it; states from the GPU leg's own generator "
                      "(uniform in [0, r))" % (nm, cores, n1),
            "single_thread_value": single,
            # "probe `nim --version` there" -- the reference is Nim over un-vendored packages; with no compiler on
            # the box there is nothing of it to time, and the port stands in
            "reference_toolchain_on_this_box": {"nim": shutil.which("nim"), "nimble": shutil.which("nimble")},
            "slot_root_2p16_cells": {"seconds": round(slot_s, 3), "perms_per_s": (35 * nc3 - 1) / slot_s,
                                     "note": "config 3 scaled to 2^16 cells of 2048 B (fake data generated, hashed and treed on %d threads)" % cores}}


def big_slots_leg(torch, bdist, coord, ctx, pkg, dev, rank, world):
    """Config 5's OTHER stated scale-down ("8 x k slots x 2^22 cells"): 8 slots at the nominal 8 GiB slot size
    (64 GiB generated and hashed on the devices; every slot crosses four 2 GiB staging chunks), sharded over the ranks like the
    32 768-slot leg (strong scaling), one exchange of slot roots, dataset tree, one proof input per rank that holds a slot;
    against the oracle-only fixture tests/golden/bigslots.json.  Coordination as in dataset_leg: the gather is the only
    collective and it is bounded; `seconds` = max over ranks."""
    import importlib
    d = importlib.import_module("codex_storage_proofs_circuits_amd.distributed")
    n_slots, n_cells = 8, 1 << 22
    cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=3, cellSize=2048, blockSize=65536, nSlots=n_slots, nCells=n_cells, nSamples=100, seed=12345)
    ctx.reset_stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    backend = d.HipBackend(pkg, ctx)
    text, first, count = "", 0, 0
    try:
        root, all_roots, (first, count) = d.dataset_root_sharded(backend, cfg, rank, world, bdist if world > 1 else None,
                                                                 dev if dev.type == "cuda" else "cpu",
                                                                 on_built=lambda err: coord.all_ok("big_slots/built", err))
        t1 = time.perf_counter()
        if count:
            text = backend.dataset.proof_input(first, 1234567).json()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    except Exception:
        if backend.dataset is not None:
            backend.dataset.free()
        ctx.trim()
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        raise
    said = coord.exchange("big_slots/result", json.dumps({"s": dt, "tree_s": t1 - t0, "root": root.tobytes()[::-1].hex()}))
    results = {r: json.loads(v) for r, v in said.items()}
    dt, t_tree = max(v["s"] for v in results.values()), max(v["tree_s"] for v in results.values())
    agree = len(results) == world and len({v["root"] for v in results.values()}) == 1
    import hashlib
    gold = None
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bigslots.json")))
    except Exception:
        pass
    # what a proof input costs once the trees exist, by what is kept of them (one rank is enough): every node resident, or compact
    # (block roots and up, 1/32 of the nodes: the bottom of each path from the touched blocks)
    latency = None
    if world == 1:
        def median_ms(ds_, slots):
            ts = []
            for s_ in slots:
                a = time.perf_counter()
                ds_.proof_input(s_, 7654321)
                ts.append((time.perf_counter() - a) * 1e3)
            return round(sorted(ts)[len(ts) // 2], 3)
        full_ms = median_ms(backend.dataset, range(n_slots))
        backend.dataset.free()
        ctx.set_keep_trees(2)
        try:
            a = time.perf_counter()
            cds = ctx.dataset(cfg)
            build_s = time.perf_counter() - a
            ok = all(hashlib.sha256(cds.proof_input(s_, 1234567).json().encode()).hexdigest() == gold["inputs"][str(s_)]["json_sha256"]
                     for s_ in (0, 5)) if gold else None
            latency = {"every_node_resident_ms": full_ms, "compact_ms": median_ms(cds, range(n_slots)), "compact_build_s": round(build_s, 3),
                       "device_bytes_per_slot": {"every_node_resident": 2 * n_cells * 32, "compact": 2 * (n_cells // 32) * 32},
                       "compact_input_json_equals_oracle_fixture": ok,
                       "note": "median wall time of cp2_proof_input_generate over the 8 slots (100 samples each), new entropy, trees already built"}
            cds.free()
        finally:
            ctx.set_keep_trees(-1)
    elif backend.dataset is not None:
        backend.dataset.free()
    ctx.trim()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    perms = n_slots * (35 * n_cells - 1) + (n_slots - 1) * world + 200 * min(world, n_slots)
    root_hex = root.tobytes()[::-1].hex()
    return {"dataset_big_slots": {"workload": "configs[4] shape, SURVEY.md 8(d)'s other scale-down: 8 slots x 2^22 cells x 2048 B (nominal 8 GiB slots) sharded "
                                              "over %d GPU(s) (%d slots on rank 0), one exchange of slot roots, dataset tree, one proof input per rank" % (world, count),
                                  "scaling": "strong", "seconds": round(dt, 4), "seconds_is": "max over ranks of each rank's own start-to-finish time",
                                  "roots_and_dataset_tree_s": round(t_tree, 4), "all_ranks_agree": bool(agree), "ranks_reporting": len(results),
                                  "perms_per_s": perms / dt, "GB_per_s_hashed": round(n_slots * n_cells * 2048 / dt / 1e9, 2),
                                  "dataset_root_hex": root_hex, "proof_input_latency": latency,
                                  "equals_oracle_fixture": (root_hex == gold["dataset_root_hex"] and
                                                            hashlib.sha256(text.encode()).hexdigest() == gold["inputs"][str(first)]["json_sha256"]) if gold else None}}


def inprocess_child(n_dev, what="main"):
    """Child mode (`bench.py --inprocess-leg N --inprocess-what W`): ONE process, N devices, through the C ABI's own multi-GPU
    entry points (cp2_multi_init / cp2_multi_dataset_build / cp2_multi_proof_input_generate: exactly what the cli twin and a
    Nim caller get).  Config 5's shape at SURVEY.md 8(d)'s scale-down; prints one JSON object.
      main              first build of the process (context creation, code-object load, communicator creation) + a proof input +
                        a second build on the warm handle; the exchange chosen automatically
      rccl, copy, host  ONE build with that exchange asked for BY NAME (nothing can fall back silently): what it did, how long,
                        whether every device ended with the same root as the fixture
      few               a dataset of few, large slots (11 x 2^18 cells), which is cut by units, against the same built whole on one device
    One process per question: a way that hangs (RCCL's first contact with two real devices) costs its own timeout, not the others' answers."""
    if os.environ.get("BENCH_INJECT") == "child_hang" and what == ("rccl" if n_dev > 1 else "main"):
        time.sleep(1e6)                             # (rehearsal) the leg's first child hangs: the parent's timeout has to end it
    import hashlib
    import __graft_entry__ as g
    pkg = g.load_package()
    n_slots, n_cells = 32768, 1 << 12
    cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=15, cellSize=2048, blockSize=65536, nSlots=n_slots, nCells=n_cells, nSamples=100, seed=12345)
    gold = None
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "config5.json")))["scaled"]
    except Exception:
        pass
    # rehearsal on a one-GPU box (BENCH_SHARE_GPU, as for the ranks): n_dev contexts on device 0, the host-gather branch
    devices = [0] * n_dev if os.environ.get("BENCH_SHARE_GPU") else list(range(n_dev))
    with _stdout_to_stderr():                      # RCCL prints its version banner on stdout when NCCL_DEBUG is set
        t0 = time.perf_counter()
        m = pkg.Multi(devices)
        t1 = time.perf_counter()
        if what == "main":
            ds = m.dataset(cfg)                        # includes context creation, code-object load and (N > 1) communicator creation
            t2 = time.perf_counter()
            text = ds.proof_input(n_slots - 1, 1234567).json()
            t3 = time.perf_counter()
            root = ds.root()
            n_shards = len(ds.shards())
            agree = all((ds.shard_root(i) == root).all() for i in range(n_shards)) if ds.units_per_slot == 1 else True
            mode_first = m.gather_mode()
            ds.free()
            ds = m.dataset(cfg)                        # a second build on the warm handle: contexts, code objects and communicators exist
            t4 = time.perf_counter()
            ds.free()
            perms = n_slots * (35 * n_cells - 1) + (n_slots - 1) * n_dev + 200
            root_hex = root.tobytes()[::-1].hex()
            res = {"devices": n_dev, "shards": n_shards, "gather": mode_first, "handle_init_s": round(t1 - t0, 4),
                   "first_build_s": round(t2 - t1, 4), "warm_build_s": round(t4 - t3, 4), "one_proof_input_json_s": round(t3 - t2, 4),
                   "perms_per_s_first": perms / (t2 - t0), "perms_per_s_warm": perms / (t4 - t3), "slots_per_s_warm": n_slots / (t4 - t3),
                   "all_devices_agree": bool(agree), "dataset_root_hex": root_hex,
                   "equals_oracle_fixture": (root_hex == gold["dataset_root_hex"] and
                                             hashlib.sha256(text.encode()).hexdigest() == gold["inputs"][str(n_slots - 1)]["json_sha256"]) if gold else None}
        elif what in ("rccl", "copy", "host"):
            m.set_policy({"rccl": pkg.GATHER_RCCL, "copy": pkg.GATHER_COPY, "host": pkg.GATHER_HOST}[what], 0)
            try:
                ta = time.perf_counter()
                d2 = m.dataset(cfg)
                tb = time.perf_counter()
                r2 = d2.root()
                same = all(bool((d2.shard_root(i) == r2).all()) for i in range(len(d2.shards())))
                res = {"mode": m.gather_mode(), "first_build_s": round(tb - ta, 4), "handle_init_s": round(t1 - t0, 4), "shards": len(d2.shards()),
                       "every_device_has_the_same_root": same, "dataset_root_hex": r2.tobytes()[::-1].hex(),
                       "equals_oracle_fixture": (r2.tobytes()[::-1].hex() == gold["dataset_root_hex"]) if gold else None}
                d2.free()
            except Exception as e:            # e.g. RCCL by name on a rehearsal box whose contexts share one device: refused, with the reason
                res = {"error": str(e)[:400]}
        else:
            few = pkg.make_config(maxDepth=32, maxLog2NSlots=8, cellSize=2048, blockSize=65536, nSlots=11, nCells=1 << 18, nSamples=100, seed=777)
            ta = time.perf_counter()
            d3 = m.dataset(few)
            tb = time.perf_counter()
            whole = m.ctx(0).dataset(few)
            tc = time.perf_counter()
            text3 = d3.proof_input(10, 424242).json()
            res = {"workload": "11 slots x 2^18 cells x 2048 B (5.9 GB)", "units_per_slot": d3.units_per_slot, "shards": len(d3.shards()),
                   "mode": m.gather_mode(), "build_s": round(tb - ta, 4), "one_device_build_s": round(tc - tb, 4),
                   "root_and_input_json_equal_one_device": bool((d3.root() == whole.root()).all()) and text3 == whole.proof_input(10, 424242).json()}
            whole.free()
            d3.free()
        m.close()
    print(json.dumps(res), flush=True)
    return 0


def run_child(argv, timeout_s, env=None):
    """A fresh subprocess (never a re-exec of this process, which has touched the GPU) with a hard timeout: terminated, then killed.
    Returns the last JSON object it printed, or {"error": ...} naming how it ended."""
    if timeout_s < 1.0:
        return {"skipped": "budget (%.0f s left for a child process)" % timeout_s}
    t0 = time.perf_counter()
    p = subprocess.Popen(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    try:
        so, se = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        p.terminate()
        try:
            so, se = p.communicate(timeout=5)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
        return {"error": "timed out after %.0f s (child terminated)" % timeout_s, "stderr_tail": (se or "")[-400:]}
    line = [l for l in so.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not line:
        return {"error": "rc %d" % p.returncode, "stderr_tail": (se or "")[-600:]}
    return dict(json.loads(line[-1]), child_wall_s=round(time.perf_counter() - t0, 2))


def inprocess_leg(torch, coord, budget, ctx, rank, world):
    """The same 32 768 x 2^12 dataset as `dataset`, but through cp2_multi_* in ONE fresh process over all `world` devices (the
    drop-in's path: no launcher, no torch.distributed).  Rank 0 starts the children once every rank has released its device
    memory; the other ranks wait on the rendezvous store (a bounded host-side wait: nothing spins on their GPUs meanwhile).
    N > 1: RCCL is asked for FIRST and BY NAME, in a process of its own -- if the in-process communicator (ncclCommInitAll) is
    what fails on first contact with real devices, that is what the record says, and the other ways still answer."""
    ctx.trim()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    coord.exchange("inprocess/released", "ok")          # every rank's device memory is back (bounded; a silent rank is named)
    res = None
    if rank == 0:
        me = [sys.executable, BENCH_PY, "--inprocess-leg", str(world)]
        cap = float(os.environ.get("BENCH_CHILD_CAP_S", "120"))
        worst = LEG_WORST_S["dataset_inprocess"](world)
        # the library bounds its exchange itself (communicator creation and the collective: CODEX_P2_EXCHANGE_TIMEOUT_S); half the child's
        # cap, so that an exchange that never completes is reported by the LIBRARY, in its own words, before the child is terminated
        cenv = dict(os.environ, CODEX_P2_EXCHANGE_TIMEOUT_S=os.environ.get("CODEX_P2_EXCHANGE_TIMEOUT_S", str(int(cap // 2))))
        child = lambda what: run_child(me + ["--inprocess-what", what], budget.child_timeout(cap), env=cenv) if budget.fits(worst) else {"skipped": "budget"}   # noqa: E731
        try:
            ways = {}
            if world > 1:
                for name in ("rccl", "copy", "host"):
                    ways[name] = child(name)
            res = child("main")
            if world > 1:
                ways["few_large_slots"] = child("few")
                res["exchange_every_way"] = ways
        finally:
            coord.post("inprocess/done", "ok", rank_key=False)
    elif coord.store is not None:
        deadline = time.monotonic() + budget.remaining()      # rank 0 is bounded by the same budget
        key = coord._key("inprocess/done")
        while not coord.store.check([key]) and time.monotonic() < deadline and 0 not in coord.failures:
            coord._poll_dead()
            time.sleep(0.05)
    if res is None:
        return {}
    res["workload"] = ("configs[4] shape (32768 slots x 2^12 cells x 2048 B) through cp2_multi_* in ONE process over %d device(s): contiguous slot ranges, "
                       "one host thread + context per device, one device-to-device exchange of slot roots, dataset tree on every device; one fresh child "
                       "process per question (main / rccl / copy / host / few), each under its own timeout" % world)
    return {"dataset_inprocess": res}


def cli_default_leg(pkg, g):
    """What workflow/prove.sh:26 actually runs: the cli twin on workflow/params.sh's defaults (11 slots x 512 cells x 2048 B,
    5 samples: about 2e5 permutations), as a FRESH PROCESS each time.  Wall time, and with CP2_TRACE the split into HIP
    runtime init / context / code-object load / buffers / hashing / dataset tree / sampling / JSON.  Beside it the C oracle,
    single thread, on the same configuration counted the reference's way ((nSlots + nSamples) slot-tree builds,
    gen_input/bn254.nim:42,57) and the build-once way (nSlots)."""
    import re
    import shutil
    import tempfile
    args = ["--depth=32", "--maxslots=256", "--cellsize=2048", "--blocksize=65536", "--nsamples=5", "--entropy=1234567",
            "--seed=12345", "--nslots=11", "--ncells=512", "--index=3", "--field=bn254", "--hash=poseidon2"]
    tmp = tempfile.mkdtemp(prefix="cp2_cli_")
    try:
        out = os.path.join(tmp, "input.json")
        walls, split = [], {}
        for i in range(4):
            env = dict(os.environ, CP2_TRACE="1") if i == 3 else dict(os.environ)
            env.pop("CODEX_P2_GPUS", None)
            t = time.perf_counter()
            r = subprocess.run([pkg.CLI_PATH] + args + ["--output=" + out], capture_output=True, text=True, timeout=120, env=env)
            dt = time.perf_counter() - t
            if r.returncode != 0:
                raise RuntimeError("cli twin failed: " + r.stderr[-400:])
            if i < 3:
                walls.append(dt)
            else:
                for m in re.finditer(r"\[cp2 trace\] (.*?)\s+([0-9.]+) ms", r.stderr):
                    split[m.group(1).strip()] = split.get(m.group(1).strip(), 0.0) + float(m.group(2))
                split["(whole process, traced run)"] = round(dt * 1e3, 1)
        golden = open(os.path.join(ROOT, "tests", "golden", "input_params_default.json")).read()
        ok = open(out).read() == golden
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    C, _ = g.load_oracle()
    t = time.perf_counter()
    for s in range(11):
        C.fake_slot_root(C.slot_seed(12345, s), 2048, 65536, 512, 1)
    once = time.perf_counter() - t
    t = time.perf_counter()
    for _ in range(5):
        C.fake_slot_root(C.slot_seed(12345, 3), 2048, 65536, 512, 1)
    again = time.perf_counter() - t
    perms_once = 11 * (35 * 512 - 1)
    return {"cli_default": {"workload": "workflow/params.sh defaults through the cli twin as a fresh process (what workflow/prove.sh:26 runs): 11 slots x 512 cells x 2048 B, "
                                        "5 samples, index 3; %d permutations of hashing" % perms_once,
                            "wall_s_runs": [round(w, 4) for w in walls], "wall_s_best": round(min(walls), 4), "input_json_equals_oracle_fixture": ok,
                            "trace_ms": {k: round(v, 3) for k, v in split.items()},
                            "cpu_oracle_single_thread": {"build_once_s": round(once, 3), "reference_way_s": round(once + again, 3),
                                                         "note": "C oracle (a port, not the Nim binary), one thread: slot trees of the 11 slots (build once) and, the reference's "
                                                                 "way, the proving slot's tree again per sample (gen_input/bn254.nim:42,57): 16 builds"}}}
