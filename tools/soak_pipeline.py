"""Soak of the host pipelines (streamed build + export with spilled bodies, sharded datasets, ingestion knobs, host-array streaming, cp2_trim):
random shapes, the streamed text against the object path on every slot and against the oracle on some, with resident
host memory and free device memory watched for leaks.  Usage: soak_pipeline.py [seconds] [seed]"""
import hashlib, os, resource, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import __graft_entry__ as g
from oracle_helpers import expected_proof_input_fast
pkg = g.load_package()
C, P = g.load_oracle()
ctx = pkg.Context(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
t0 = time.time()
it = bad = 0
free0 = None
tmp = tempfile.mkdtemp(prefix="cp2soak")
while time.time() - t0 < budget:
    it += 1
    cs = int(rng.choice([64, 128, 256, 2048, 2048, 100, 31]))
    cpb = int(rng.choice([1, 2, 4, 32]))
    nblocks = int(rng.choice([1, 2, 8, 64, 256]))
    nc = cpb * nblocks
    if nc & (nc - 1) or nc < 2:      # one cell: nothing can be sampled (extractLowBits asserts k > 0, types/bn254.nim:48)
        continue
    ns_slots = int(rng.integers(1, 24))
    c = dict(maxDepth=16, maxLog2NSlots=5, cellSize=cs, blockSize=cs * cpb, nSlots=ns_slots, nCells=nc, nSamples=int(rng.integers(1, 40)),
             seed=int(rng.integers(0, 1 << 40)))
    entropy = int(rng.integers(1, 1 << 62))
    group = int(rng.choice([0, 1, 2, 5]))
    threads = int(rng.choice([1, 2, 7]))
    use_file = (it % 4 == 0) and (cs & 3) == 0
    cc = dict(c)
    if use_file:
        base = os.path.join(tmp, "s%d_" % it)
        for k in range(ns_slots):
            C.gen_fake_cells(C.slot_seed(c["seed"], k), 0, nc, cs).tofile("%s%d.dat" % (base, k))
        del cc["seed"]
        cc["file"] = base
    cfg = pkg.make_config(**cc)
    ctx.set_ingest(int(rng.choice([0, 1, 3])), int(rng.choice([0, 2, 4])), int(rng.choice([0, 1 << 16, 1 << 20])))
    ctx.set_ingest_mapped(int(rng.choice([-1, 0, 1, 1])))      # slot files (just written: cached) by mapping or through the ring
    # bodies kept in memory up to a random budget (1 byte: every body spills; 50 KB: some do; 4 GiB: none does)
    ctx.set_body_budget(int(rng.choice([1, 50000, 4 << 30])), tmp)
    ref = ctx.dataset(cfg)
    sd = ctx.dataset_streamed(cfg, entropy, threads=threads, group_slots=group)
    sd.export_streamed(None, threads=threads)
    for s in range(ns_slots):
        if sd.streamed_json(s) != ref.proof_input(s, entropy).json():
            bad += 1
            print("MISMATCH streamed vs object path", c, "slot", s, flush=True)
    if it % 7 == 0:
        s = int(rng.integers(0, ns_slots))
        want = P.export_json(expected_proof_input_fast(C, P, c, s, entropy, threads=8))
        if sd.streamed_json(s) != want:
            bad += 1
            print("MISMATCH vs oracle", c, "slot", s, flush=True)
    if ns_slots >= 2:       # two shards
        k = int(rng.integers(1, ns_slots))
        a, b = ctx.dataset(cfg, 0, k), ctx.dataset_streamed(cfg, entropy, k, ns_slots - k, threads=threads, group_slots=group)
        roots = np.concatenate([a.local_roots(), b.local_roots()])
        b.set_roots(roots)
        b.export_streamed(None, threads=threads)
        if b.streamed_json(ns_slots - 1) != ref.proof_input(ns_slots - 1, entropy).json():
            bad += 1
            print("MISMATCH sharded", c, flush=True)
        a.free(); b.free()
    ref.free(); sd.free()
    if any(f.endswith(".part") or f.startswith("cp2_bodies_") for _, ds_, fs in os.walk(tmp) for f in fs + ds_):
        bad += 1
        print("spill files left behind", os.listdir(tmp)[:4], flush=True)
    if it % 50 == 0:
        ctx.trim()              # cached scratch back to the system; the next iteration allocates again
    if use_file:
        for k in range(ns_slots):
            os.remove("%s%d.dat" % (base, k))
    if it % 10 == 0:
        n = int(rng.integers(1 << 20, 3 << 20))
        x = rng.integers(0, 256, size=(n, 96), dtype=np.uint8)
        y = ctx.permute_batch(x)
        idx = rng.integers(0, n, size=256)
        if not np.array_equal(y[idx], C.permute_batch(x[idx], threads=4)):
            bad += 1
            print("MISMATCH host-array permute", n, flush=True)
        if it % 20 == 0:         # the same through arrays the caller pinned (no ring, copy engines in place)
            import ctypes
            xin = torch.from_numpy(x).pin_memory()
            xout = torch.zeros_like(xin).pin_memory()
            ctx._ck(ctx.L.cp2_permute_batch(ctx.h, ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(xout.data_ptr()), n), "cp2_permute_batch")
            if not np.array_equal(xout.numpy(), y):
                bad += 1
                print("MISMATCH pinned host-array permute", n, flush=True)
            del xin, xout
    if it % 20 == 0:
        free, total = torch.cuda.mem_get_info()
        if free0 is None:
            free0 = free
        print("iteration %d  bad=%d  maxrss %.0f MB  device free %.2f GiB (first reading %.2f)  %.0f s" %
              (it, bad, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, free / 2**30, free0 / 2**30, time.time() - t0), flush=True)
ctx.set_ingest(0, 0, 0)
ctx.set_ingest_mapped(-1)
ctx.set_body_budget(4 << 30, None)
print("pipeline soak done: %d iterations, mismatches: %d, %.0f s" % (it, bad, time.time() - t0))
sys.exit(1 if bad else 0)
