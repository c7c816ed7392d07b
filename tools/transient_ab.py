"""A/B on one box, one process: the same dataset of nominal 8 GiB slots built keeping every node (the builder runs once over
everything: the reference rate), compact and roots-only (batch by batch: what config 5's nominal share has to use), plain and
streamed.  VERDICT r04 item 4: the transient builds drained the device between batches (1.3-2 %); since round 5 the batches
pipeline, and this tool says how close to the resident build's rate they run.  Same roots required from every variant.
Usage: transient_ab.py [n_slots = 128] [repeats = 1]      (128 slots = 1 TiB: about 25 s per variant)
       transient_ab.py <n_slots> <repeats> vs <dir of another build: <dir>/codex-storage-proofs-circuits_amd/{__init__.py, libcodex_p2.so}>"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g

n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 128
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if len(sys.argv) > 3 and sys.argv[3] == "vs":
    # A/B against ANOTHER build of the package on the same box (e.g. build/r04: round 4's library and binding, which synchronised and
    # freed per batch): alternating fresh processes, the variants that differ between the two (compact, compact_streamed)
    import subprocess
    other = os.path.abspath(sys.argv[4])
    runs = {"this": [], "other": []}
    for rep in range(repeats):
        for who in ("other", "this"):
            env = dict(os.environ, TRANSIENT_AB_PKG=other) if who == "other" else dict(os.environ)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n_slots), "1", "only", "compact,compact_streamed"], env=env, capture_output=True, text=True, timeout=3000)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            assert r.returncode == 0 and line, r.stderr[-2000:]
            runs[who].append(json.loads(line[-1])["seconds_best"])
            print(who, runs[who][-1], flush=True)
    best = {w: {k: min(x[k] for x in v) for k in v[0]} for w, v in runs.items()}
    print(json.dumps({"n_slots": n_slots, "TiB": n_slots * 8 / 1024, "other_package": sys.argv[4], "seconds_best": best, "runs": runs,
                      "this_over_other_rate": {k: round(best["other"][k] / best["this"][k], 5) for k in best["this"]}}))
    sys.exit(0)
only = set(sys.argv[4].split(",")) if len(sys.argv) > 4 and sys.argv[3] == "only" else None
if os.environ.get("TRANSIENT_AB_PKG"):      # another build of the package (its own binding + library)
    import importlib.util
    d = os.path.join(os.environ["TRANSIENT_AB_PKG"], "codex-storage-proofs-circuits_amd")
    spec = importlib.util.spec_from_file_location("cp2_other_pkg", os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["cp2_other_pkg"] = pkg
    spec.loader.exec_module(pkg)
else:
    pkg = g.load_package()
n_cells, cs, bs = 1 << 22, 2048, 65536
c = dict(maxDepth=32, maxLog2NSlots=max(1, (n_slots - 1).bit_length()), cellSize=cs, blockSize=bs, nSlots=n_slots, nCells=n_cells, nSamples=100, seed=12345)
cfg = pkg.make_config(**c)
ctx = pkg.Context(0)
perms = n_slots * (35 * n_cells - 1)
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bigslots.json")))
hexroot = lambda a: np.asarray(a, dtype=np.uint8).tobytes()[::-1].hex()      # noqa: E731
ctx.dataset(pkg.make_config(**dict(c, nSlots=2, maxLog2NSlots=1))).free()     # warm-up: code object, scratch
res, ref_roots = {}, None
for rep in range(repeats):
    for name, mode, streamed in (("resident", 1, False), ("compact", 2, False), ("roots_only", 0, False), ("resident_streamed", 1, True),
                                 ("compact_streamed", 2, True)):
        if only and name not in only:
            continue
        ctx.set_keep_trees(mode)
        ctx.trim()
        # the resident build allocates every node up front (hipMalloc of 0.25 GiB per slot: about a second per 32 GiB), which is not
        # hashing: its CP2_TRACE lap "generate + hash + layers" is the time to compare a transient build with
        import tempfile
        os.environ["CP2_TRACE"] = "1"
        sys.stderr.flush()
        saved, tf = os.dup(2), tempfile.TemporaryFile()
        os.dup2(tf.fileno(), 2)
        try:
            t0 = time.perf_counter()
            ds = ctx.dataset_streamed(cfg, 1234567, threads=12) if streamed else ctx.dataset(cfg)
            dt = time.perf_counter() - t0
        finally:
            os.dup2(saved, 2)
            os.close(saved)
            del os.environ["CP2_TRACE"]
        tf.seek(0)
        laps = {m.group(1).strip(): float(m.group(2)) for m in __import__("re").finditer(r"\[cp2 trace\] (.*?)\s+([0-9.]+) ms", tf.read().decode())}
        hashing_s = laps.get("fake slots: generate + hash + layers", laps.get("trees (sampling overlapped)"))
        if mode == 1 and hashing_s:
            res.setdefault(name + "_without_allocation", []).append(hashing_s / 1e3)
        roots = ds.local_roots()
        assert ds.tree_mode == mode
        if ref_roots is None:
            ref_roots = roots.copy()
            assert [hexroot(r) for r in roots[:8]] == gold["slot_roots_hex"][:min(8, n_slots)], "slots 0..7 != tests/golden/bigslots.json"
        assert np.array_equal(roots, ref_roots), name
        ds.free()
        res.setdefault(name, []).append(dt)
        print("%-18s %8.3f s  %.4e perm/s  %.2f GB/s" % (name, dt, perms / dt, n_slots * n_cells * cs / dt / 1e9), flush=True)
best = {k: min(v) for k, v in res.items()}
if only:
    print(json.dumps({"n_slots": n_slots, "seconds_best": {k: round(v, 3) for k, v in best.items()}}))
    sys.exit(0)
out = {"n_slots": n_slots, "TiB": n_slots * n_cells * cs / 2**40, "seconds_best": {k: round(v, 3) for k, v in best.items()}, "seconds_all": {k: [round(x, 3) for x in v] for k, v in res.items()},
       "perms_per_s": {k: perms / v for k, v in best.items()},
       "rate_vs_resident_without_its_allocation": {k: round(best.get("resident_without_allocation", best["resident"]) / v, 5) for k, v in best.items() if "streamed" not in k},
       "rate_vs_resident_streamed": {k: round(best["resident_streamed"] / v, 5) for k, v in best.items() if "streamed" in k},
       "roots_0_7_equal_fixture": True, "every_variant_same_roots": True}
print(json.dumps(out))
