"""Where do the per-group layer passes / k_sample_paths / gathers of a streamed build land relative to the hash launches -- from slot
files and from the fake source?  

  run        (ON THE GPU BOX, under `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 ... run [dir]`)
             writes configs[3]'s scale-down as slot files, then, separated by marker launches (k_permute_batch of ONE state):
             a warm-up build from files, a measured build from files, a measured build from the fake source
  summarize  (anywhere) <trace dir> [out.txt]: per measured build -- span, how much of it some k_hash_cells launch was running,
             the gaps between hash launches, the small kernels of the passes (count, total, how much of their time ran BESIDE a hash
             launch), the host-to-device copies (busy fraction of the span: does the upload bind?), per-pass latency
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(where):
    import shutil, tempfile, time
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    ctx = pkg.Context(0)
    dev = torch.device("cuda", 0)
    n_slots, n_cells, cs = 4096, 1 << 12, 2048
    thr = max(1, min(16, len(os.sched_getaffinity(0))))
    work = tempfile.mkdtemp(prefix="cp2_sftr_", dir=where)
    base = os.path.join(work, "slot")
    marker = pkg.felts_to_array([0, 1, 2]).reshape(1, 96)
    try:
        per = 256
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
        c = dict(maxDepth=32, maxLog2NSlots=12, cellSize=cs, blockSize=65536, nSlots=n_slots, nCells=n_cells, nSamples=100, seed=12345)
        cfg_fake = pkg.make_config(**c)
        cfg_file = pkg.make_config(**dict({k: v for k, v in c.items() if k != "seed"}, file=base))
        for label, cfg in (("warm_file", cfg_file), ("warm_fake", cfg_fake), ("file", cfg_file), ("fake", cfg_fake)):
            ctx.permute_batch(marker)                                  # marker launch: a k_permute_batch grid of one workgroup
            t = time.perf_counter()
            sd = ctx.dataset_streamed(cfg, 1234567, threads=thr)
            sd.set_roots(None)
            sd.export_streamed(None, threads=thr)
            print("%-9s %.4f s" % (label, time.perf_counter() - t), flush=True)
            sd.free()
        ctx.permute_batch(marker)
    finally:
        shutil.rmtree(work, ignore_errors=True)


def union_len(iv):
    iv = sorted(iv)
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def overlap_with(iv, cover):
    """total length of the intervals `iv` that lies inside the union of `cover`"""
    cover = sorted(cover)
    merged = []
    for a, b in cover:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    tot = 0
    import bisect
    starts = [m[0] for m in merged]
    for a, b in iv:
        i = max(0, bisect.bisect_right(starts, a) - 1)
        while i < len(merged) and merged[i][0] < b:
            tot += max(0, min(b, merged[i][1]) - max(a, merged[i][0]))
            i += 1
    return tot


def summarize(d, out_path=None):
    kt = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getsize)[-1]
    mc = sorted(glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True), key=os.path.getsize)
    K = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0) or 0))
         for r in csv.DictReader(open(kt))]
    K.sort(key=lambda x: x[1])
    M = []
    if mc:
        for r in csv.DictReader(open(mc[-1])):
            M.append((r.get("Direction", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    markers = [k for k in K if "k_permute_batch" in k[0] and k[3] <= 256]
    lines = ["# kernel + memory-copy trace of the streamed build of configs[3]'s scale-down (4096 slots x 2^12 cells, 100 samples), one process:",
             "# warm-ups, then one build from slot files in the page cache and one from the fake source (tools/streamed_files_trace.py run; summarised by the same tool)",
             "# trace: %s" % os.path.relpath(kt, ROOT)]
    labels = ["warm_file", "warm_fake", "file", "fake"]
    res = {}
    for i, label in enumerate(labels):
        if i + 1 >= len(markers):
            break
        lo, hi = markers[i][2], markers[i + 1][1]
        seg = [k for k in K if lo <= k[1] < hi]
        hashk = [(a, b) for n, a, b, _ in seg if "k_hash_cells" in n]
        gen = [(a, b) for n, a, b, _ in seg if "k_gen_fake_cells" in n]
        small = {nm: [(a, b) for n, a, b, _ in seg if nm in n] for nm in ("k_compress_layer", "k_sample_paths", "k_gather_rows")}
        if not hashk:
            continue
        # the build: from the marker to the last gather of the last pass (what follows in the segment -- the dataset tree, the texts
        # read back, the free -- is not the streamed build)
        t0, t1 = lo, max(b for n, a, b, _ in seg if "k_gather_rows" in n)
        span = t1 - t0
        h2d = [(a, b) for dname, a, b in M if t0 <= a < t1 and ("HOST_TO_DEVICE" in dname.upper() or "H2D" in dname.upper())]
        d2h = [(a, b) for dname, a, b in M if t0 <= a < t1 and ("DEVICE_TO_HOST" in dname.upper() or "D2H" in dname.upper())]
        hs = sorted(hashk)
        gaps = []
        cur_end = hs[0][1]
        for a, b in hs[1:]:
            if a > cur_end:
                gaps.append(a - cur_end)
            cur_end = max(cur_end, b)
        small_all = [iv for v in small.values() for iv in v]
        r = {"span_ms": span / 1e6,
             "hash_launches": len(hashk), "hash_busy_frac": union_len(hashk) / span, "hash_sum_ms": sum(b - a for a, b in hashk) / 1e6,
             "two_hash_launches_at_once_frac": (sum(b - a for a, b in hashk) - union_len(hashk)) / span,
             "hash_gaps": len(gaps), "hash_gap_total_ms": sum(gaps) / 1e6, "hash_gap_max_ms": (max(gaps) if gaps else 0) / 1e6,
             "tail_after_last_hash_ms": (t1 - max(b for _, b in hashk)) / 1e6,
             "small_kernels": {nm: {"launches": len(v), "total_ms": sum(b - a for a, b in v) / 1e6, "avg_us": (sum(b - a for a, b in v) / max(1, len(v))) / 1e3} for nm, v in small.items()},
             "small_kernel_time_beside_a_hash_launch_frac": overlap_with(small_all, hashk) / max(1, sum(b - a for a, b in small_all)),
             "h2d_copies": len(h2d), "h2d_busy_frac": union_len(h2d) / span if h2d else 0.0, "h2d_sum_ms": sum(b - a for a, b in h2d) / 1e6,
             "d2h_copies": len(d2h), "d2h_sum_ms": sum(b - a for a, b in d2h) / 1e6}
        # passes: a k_sample_paths launch closes the layer passes before it; latency = first layer kernel after the previous pass ... last gather of this pass
        sp = sorted(small["k_sample_paths"])
        ga = sorted(small["k_gather_rows"])
        cl = sorted(small["k_compress_layer"])
        lat = []
        prev = t0
        for a, b in sp:
            first_layer = [x for x in cl if prev <= x[0] < a]
            last_g = [x for x in ga if x[0] >= b]
            nxt = [x[0] for x in sp if x[0] > a]
            last_g = [x for x in last_g if not nxt or x[0] < nxt[0]]
            if first_layer and last_g:
                lat.append((max(x[1] for x in last_g) - first_layer[0][0]) / 1e6)
            prev = b
        if lat:
            lat.sort()
            r["passes"] = len(lat)
            r["pass_latency_ms"] = {"median": lat[len(lat) // 2], "max": lat[-1], "min": lat[0]}
        res[label] = r
    for label in ("file", "fake"):
        if label in res:
            lines.append("== %s" % label)
            lines.append(json.dumps(res[label], indent=1))
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    print(text)


def timeline(d, first=4, last=16):
    """per turn of the measured build from files: when its upload and its hash launch started and ended (ms from the build's start)"""
    kt = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getsize)[-1]
    mc = sorted(glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True), key=os.path.getsize)[-1]
    K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]), r.get("Stream_Id", "?")) for r in csv.DictReader(open(kt)))
    M = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"], r.get("Stream_Id", "?")) for r in csv.DictReader(open(mc)))
    mk = [k for k in K if "k_permute_batch" in k[2] and k[3] <= 256]
    lo, hi = mk[2][1], mk[3][0]
    H = [k for k in K if lo <= k[0] < hi and "k_hash_cells" in k[2]]
    U = [m for m in M if lo <= m[0] < hi and "HOST_TO_DEVICE" in m[2] and m[1] - m[0] > 500000]
    prev_end = lo
    for i, (h, u) in enumerate(zip(H, U)):
        if first <= i < last:
            print("turn %2d: upload %7.2f .. %7.2f on stream %s | hash %7.2f .. %7.2f on stream %s (%d cells) | the upload started %+.2f ms from the end of hash %d, the hash %+.2f ms from the end of the hash before it" %
                  (i, (u[0] - lo) / 1e6, (u[1] - lo) / 1e6, u[3], (h[0] - lo) / 1e6, (h[1] - lo) / 1e6, h[4], h[3], (u[0] - H[i - 2][1]) / 1e6 if i >= 2 else 0.0, i - 2, (h[0] - prev_end) / 1e6))
        prev_end = max(prev_end, h[1])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else None)
    elif len(sys.argv) > 2 and sys.argv[1] == "timeline":
        timeline(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "summarize":
        summarize(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        print(__doc__)
