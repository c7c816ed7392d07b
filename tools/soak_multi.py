"""Soak of the in-process multi-context path (cp2_multi_*, include/codex_p2.h section e): 2-4 contexts on device 0 (the host-gather
branch; a one-GPU box has no second device), random dataset shapes and shard splits, plain / streamed / cached builds, fake and
slot-file sources -- EVERY dataset root and every slot root against the C oracle, proof-input JSON against the oracle on some
slots and against the single-context object path on others, with resident host memory and free device memory watched.
Usage: soak_multi.py [seconds] [seed] [units]"""
import os, resource, shutil, sys, tempfile, time
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import __graft_entry__ as g
from oracle_helpers import expected_proof_input_fast
pkg = g.load_package()
C, P = g.load_oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
UNITS = len(sys.argv) > 3 and sys.argv[3] == "units"          # third argument "units": every slot cut into 2 / 4 / 8 units (plain, streamed and cached builds)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
single = pkg.Context(0)
handles = {n: pkg.Multi([0] * n) for n in (2, 3, 4)}          # long-lived handles: contexts and their scratch pools are reused
t0 = time.time()
it = bad = checked_roots = checked_json = by_units = roots_only = 0
free0 = None
tmp = tempfile.mkdtemp(prefix="cp2soakm")
modes = Counter()
while time.time() - t0 < budget:
    it += 1
    big = it % 9 == 0                                          # now and then a shape that keeps several contexts' kernels in flight together
    cs = 2048 if big else int(rng.choice([64, 128, 256, 2048, 100, 31]))
    cpb = 32 if big else int(rng.choice([1, 2, 4, 32]))
    nblocks = int(rng.choice([32, 128])) if big else int(rng.choice([8, 16, 64, 256] if UNITS else [1, 2, 8, 64]))
    nc = cpb * nblocks
    if nc & (nc - 1) or nc < 2:
        continue
    n_slots = int(rng.integers(24, 96)) if big else int(rng.integers(1, 40))
    c = dict(maxDepth=16, maxLog2NSlots=7, cellSize=cs, blockSize=cs * cpb, nSlots=n_slots, nCells=nc, nSamples=int(rng.integers(1, 30)),
             seed=int(rng.integers(0, 1 << 40)))
    entropy = int(rng.integers(1, 1 << 62))
    n_ctx = int(rng.choice([2, 3, 4]))
    m = handles[n_ctx]
    # min cells per device 1: every context gets a shard (as many as there are slots); a random larger value: fewer shards
    min_cells = int(rng.choice([1, 1, 1, nc * max(1, n_slots // 2), 1 << 30]))
    m.set_policy(int(rng.choice([pkg.GATHER_AUTO, pkg.GATHER_HOST, pkg.GATHER_COPY])), min_cells)
    split = int(rng.choice([2, 4, 8] if UNITS else [0, 0, 1, 2, 4]))   # choose / whole slots only / every slot cut into 2, 4 (8) units
    m.set_split(split)
    keep = int(rng.choice([-1, -1, 0, 2]))                     # now and then roots-only / compact datasets
    for i in range(m.count):
        m.ctx(i).set_keep_trees(keep)
    single.set_keep_trees(int(rng.choice([-1, 0, 2])))
    roots_only += keep in (0, 2)
    use_file = (it % 5 == 0) and (cs & 3) == 0 and not big
    cc = dict(c)
    if use_file:
        base = os.path.join(tmp, "s%d_" % it)
        for k in range(n_slots):
            C.gen_fake_cells(C.slot_seed(c["seed"], k), 0, nc, cs).tofile("%s%d.dat" % (base, k))
        del cc["seed"]
        cc["file"] = base
        mapped = int(rng.choice([-1, 0, 1, 1]))                # mapped ingestion of the (just written, so cached) files on the multi side, the ring on the other or the other way round
        for i in range(m.count):
            m.ctx(i).set_ingest_mapped(mapped)
        single.set_ingest_mapped(int(rng.choice([0, 1])))
    cfg = pkg.make_config(**cc)
    kind = int(rng.integers(0, 3))                             # plain / streamed / cached (round 5: a streamed build cut by units is two-phase, same plan as the others)
    threads = int(rng.choice([1, 3, 8]))
    if kind == 0:
        ds = m.dataset(cfg)
    elif kind == 1:
        ds = m.dataset_streamed(cfg, entropy, threads=threads, group_slots=int(rng.choice([0, 1, 3])))
    else:
        cache = os.path.join(tmp, "cache%d" % it)
        m.dataset(cfg, cache=cache).free()                     # written ...
        ds = m.dataset(cfg, cache=cache)                       # ... and loaded
    shards = ds.shards()
    modes[m.gather_mode().split(" ")[0]] += 1
    S = ds.units_per_slot
    by_units += S > 1
    want_world = min(n_ctx, n_slots * S, max(1, -(-n_slots * nc // max(1, min_cells))))
    if len(shards) != want_world or [(f, k) for _, f, k in shards] != [pkg.shard_range(n_slots * S, r, len(shards)) for r in range(len(shards))]:
        bad += 1
        print("WRONG SPLIT", c, n_ctx, min_cells, split, S, shards, flush=True)
    # every slot root and the dataset root against the C oracle (as EVERY shard's device computed the latter)
    want_roots = np.stack([C.fake_slot_root(C.slot_seed(c["seed"], s), cs, cs * cpb, nc, 8) for s in range(n_slots)])
    want_root = C.merkle_root(want_roots)
    got_roots = ds.slot_roots()
    checked_roots += n_slots + len(shards)
    if not np.array_equal(got_roots, want_roots):
        bad += 1
        print("MISMATCH slot roots", c, shards, flush=True)
    if not np.array_equal(ds.root(), want_root):
        bad += 1
        print("MISMATCH dataset root", c, shards, flush=True)
    for i in range(len(shards) if S == 1 else 0):              # by whole slots every device builds the dataset tree itself
        if not np.array_equal(ds.shard_root(i), want_root):
            bad += 1
            print("MISMATCH dataset root on shard", i, c, shards, flush=True)
    # proof inputs: an edge slot against the oracle, a random slot against the single-context object path
    edge = int(rng.choice([f // S for _, f, k in shards] + [(f + k - 1) // S for _, f, k in shards]))   # a slot on a shard edge
    s_rand = int(rng.integers(0, n_slots))
    if kind == 1:
        ds.export_streamed(None, threads=threads)
        t_edge, t_rand = ds.streamed_json(edge), ds.streamed_json(s_rand)
    else:
        t_edge, t_rand = ds.proof_input(edge, entropy).json(), ds.proof_input(s_rand, entropy).json()
    if it % 3 == 0:
        checked_json += 1
        if t_edge != P.export_json(expected_proof_input_fast(C, P, c, edge, entropy, threads=8, slot_roots=want_roots)):
            bad += 1
            print("MISMATCH input.json vs oracle", c, "slot", edge, flush=True)
    if it % 4 == 0:                                            # the batched export (by units: one gather per device, devices in parallel) against the per-slot path
        exp = os.path.join(tmp, "exp%d" % it)
        os.mkdir(exp)
        pick = sorted(set(int(x) for x in rng.integers(0, n_slots, size=min(n_slots, 5))))
        nbytes = ds.export_proof_inputs(pick, entropy, exp, threads=threads, batch=int(rng.choice([0, 1, 2])))
        texts = [open(os.path.join(exp, "input_%d.json" % s_)).read() for s_ in pick]
        if nbytes != sum(len(t) for t in texts) or any(t != ds.proof_input(s_, entropy).json() for s_, t in zip(pick, texts)):
            bad += 1
            print("MISMATCH batched export vs per-slot proof inputs", c, pick, flush=True)
    ref = single.dataset(cfg)
    if t_rand != ref.proof_input(s_rand, entropy).json():
        bad += 1
        print("MISMATCH multi vs single context", c, "slot", s_rand, flush=True)
    ref.free()
    ds.free()
    for f in os.listdir(tmp):
        p = os.path.join(tmp, f)
        shutil.rmtree(p) if os.path.isdir(p) else os.remove(p)
    if it % 40 == 0:
        for h in handles.values():
            for i in range(h.count):
                h.ctx(i).trim()
        single.trim()
    if it % 20 == 0:
        free, total = torch.cuda.mem_get_info()
        if free0 is None:
            free0 = free
        print("iteration %d  bad=%d  roots checked %d  json vs oracle %d  by units %d  gather %s  maxrss %.0f MB  device free %.2f GiB (first reading %.2f)  %.0f s" %
              (it, bad, checked_roots, checked_json, by_units, dict(modes), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, free / 2**30, free0 / 2**30,
               time.time() - t0), flush=True)
shutil.rmtree(tmp, ignore_errors=True)
print("multi-context soak done: %d iterations (%d of them cut by units, %d roots-only), %d roots and %d input.json texts checked against the oracle, gather modes %s, mismatches: %d, %.0f s" %
      (it, by_units, roots_only, checked_roots, checked_json, dict(modes), bad, time.time() - t0))
sys.exit(1 if bad else 0)
