"""How fast is the persisted-tree cache against a rebuild?  cp2_slot_trees_save / cp2_slot_trees_load for node buffers of
1 GiB (4096 slots x 2^12 cells) and 8 GiB (32768 x 2^12).  Usage: cache_probe.py [dir]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
d = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
for n_slots in (4096, 32768):
    t = time.perf_counter()
    trees = ctx.slot_trees_fake(12345, 0, n_slots, 2048, 65536, 1 << 12)
    roots = trees.roots()
    t_build = time.perf_counter() - t
    path = os.path.join(d, "cp2_cache_probe_%d.bin" % n_slots)
    t = time.perf_counter()
    trees.save(path)
    t_save = time.perf_counter() - t
    size = os.path.getsize(path)
    trees.free()
    t = time.perf_counter()
    back = ctx.slot_trees_load(path)
    t_load = time.perf_counter() - t
    ok = np.array_equal(back.roots(), roots)
    back.free()
    os.remove(path)
    print("%6d slots: nodes %.2f GiB; build %.2f s; save %.2f s (%.2f GB/s); load %.2f s (%.2f GB/s); roots equal: %s" %
          (n_slots, size / 2**30, t_build, t_save, size / t_save / 1e9, t_load, size / t_load / 1e9, ok), flush=True)
