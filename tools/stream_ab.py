"""A/B of the streamed proof-input pipeline on ONE box: configs[3] (4096 slots x 2^12 cells, 100 samples) with and without the
ramp-down of the last groups (CP2_STREAM_RAMP=0), alternating, CP2_TRACE laps on stderr.  Usage: stream_ab.py [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package(); ctx = pkg.Context(0)
cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=12, cellSize=2048, blockSize=65536, nSlots=4096, nCells=1 << 12, nSamples=100, seed=12345)
thr = max(1, min(16, len(os.sched_getaffinity(0))))
ts = []
for i in range(4):
    t = time.perf_counter(); sd = ctx.dataset_streamed(cfg, 1234567, threads=thr); sd.set_roots(None); n = sd.export_streamed(None, threads=thr); ts.append(time.perf_counter() - t); sd.free()
t = time.perf_counter(); ds = ctx.dataset(cfg); tb = time.perf_counter() - t; ds.free()
print("streamed runs", [round(x, 4) for x in ts], "best", round(min(ts[1:]), 4), "| trees alone", round(tb, 4), flush=True)
''' % ROOT
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for ramp in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, CP2_STREAM_RAMP=ramp), capture_output=True, text=True)
        print("ramp=%s:" % ramp, out.stdout.strip() or out.stderr[-500:], flush=True)
