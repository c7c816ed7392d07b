"""A/B of library variants on the config-4 tree build (4096 slots x 2^12 cells, fake data) and the streamed pipeline:
interleaved rounds, one child process per sample.  python tools/ab_trees.py default build/variants/libX.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json, os
sys.path.insert(0, %r)
import torch
import __graft_entry__ as g
pkg = g.load_package(); ctx = pkg.Context(0)
cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=12, cellSize=2048, blockSize=65536, nSlots=4096, nCells=4096, nSamples=100, seed=12345)
ds = ctx.dataset(pkg.make_config(maxDepth=32, maxLog2NSlots=6, cellSize=2048, blockSize=65536, nSlots=64, nCells=4096, nSamples=100, seed=1)); ds.free()
best = 1e9
for _ in range(3):
    t = time.perf_counter(); ds = ctx.dataset(cfg); dt = time.perf_counter() - t; root = ds.root().tobytes().hex(); ds.free(); best = min(best, dt)
sbest = 1e9
for _ in range(3):
    t = time.perf_counter(); sd = ctx.dataset_streamed(cfg, 1234567, threads=16); sd.set_roots(None); n = sd.export_streamed(None, threads=16); dt = time.perf_counter() - t; sd.free(); sbest = min(sbest, dt)
print(json.dumps({"classic_trees_s": best, "streamed_total_s": sbest, "root": root[:16], "bytes": n}))
''' % ROOT
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(2):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["CODEX_P2_LIB"] = os.path.abspath(l)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        try:
            res[l].append(json.loads(out.stdout.strip().split("\n")[-1]))
        except Exception:
            res[l].append({"error": out.stderr[-300:]})
        print(l, res[l][-1], flush=True)
