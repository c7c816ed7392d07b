"""A/B on one box, one process: the same dataset of nominal 8 GiB slots built keeping every node (the builder runs once over
everything: the reference rate), compact and roots-only (batch by batch: what config 5's nominal share has to use), plain and
streamed.  VERDICT r04 item 4: the transient builds drained the device between batches (1.3-2 %); since round 5 the batches
pipeline, and this tool says how close to the resident build's rate they run.  Same roots required from every variant.
Usage: transient_ab.py [n_slots = 128] [repeats = 1]      (128 slots = 1 TiB: about 25 s per variant)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 128
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 1
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
        ctx.set_keep_trees(mode)
        ctx.trim()
        t0 = time.perf_counter()
        ds = ctx.dataset_streamed(cfg, 1234567, threads=12) if streamed else ctx.dataset(cfg)
        dt = time.perf_counter() - t0
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
out = {"n_slots": n_slots, "TiB": n_slots * n_cells * cs / 2**40, "seconds_best": {k: round(v, 3) for k, v in best.items()}, "seconds_all": {k: [round(x, 3) for x in v] for k, v in res.items()},
       "perms_per_s": {k: perms / v for k, v in best.items()},
       "vs_resident": {k: round(best["resident"] / v, 5) for k, v in best.items() if not k.endswith("streamed")},
       "vs_resident_streamed": {k: round(best["resident_streamed"] / v, 5) for k, v in best.items() if k.endswith("streamed")},
       "roots_0_7_equal_fixture": True, "every_variant_same_roots": True}
print(json.dumps(out))
