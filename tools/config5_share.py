"""Config 5 at its NOMINAL slot size, one GPU's share: BASELINE.json configs[4] is 32 768 slots of 8 GiB over 8 GPUs, i.e. 4096
slots x 2^22 cells x 2048 B = 32 TiB of slot data per GPU.  This runs exactly that share on one MI355X as a dataset of its own
(4096 slots, 12-level dataset tree): fake data generated and hashed on the device, ROOTS ONLY (the resident trees would need
1 TiB; the library decides that by itself from what the device has free), then a proof input for one slot from its tree
rebuilt on demand.  No oracle can follow at this size (14 days of CPU), so the run is pinned by properties:
  * the roots of slots 0..7 equal the oracle-only fixture tests/golden/bigslots.json (same seed, same slots);
  * two more slot roots, chosen far apart, are recomputed by the C oracle on the host WHILE the GPU hashes;
  * the dataset root equals the oracle's Merkle root over the 4096 GPU-computed slot roots;
  * the emitted proof input passes the circuit-side checker (oracle.circuit_check: every `===` of the circom templates).
Usage (needs CP2_TRACE=1 in the environment for progress lines):  config5_share.py [n_slots] [auto | 0 | 2]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
C, P = g.load_oracle()
n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_cells, cs, bs = 1 << 22, 2048, 65536
c = dict(maxDepth=32, maxLog2NSlots=max(1, (n_slots - 1).bit_length()), cellSize=cs, blockSize=bs, nSlots=n_slots, nCells=n_cells, nSamples=100, seed=12345)
cfg = pkg.make_config(**c)
ctx = pkg.Context(0)
free0, total = torch.cuda.mem_get_info()
print("device memory free %.1f GiB of %.1f; resident trees would need %.1f GiB" % (free0 / 2**30, total / 2**30, n_slots * 0.25 + n_slots * 2**-13), flush=True)
checks = [s for s in (1234, n_slots - 1) if s < n_slots and s > 7]
oracle_roots = {}


def oracle_side():
    for s in checks:
        t = time.time()
        oracle_roots[s] = C.fake_slot_root(C.slot_seed(c["seed"], s), cs, bs, n_cells, 14)
        print("  [host] C oracle root of slot %d in %.0f s" % (s, time.time() - t), flush=True)


th = threading.Thread(target=oracle_side)
th.start()
t0 = time.time()
mode_arg = sys.argv[2] if len(sys.argv) > 2 else "auto"      # auto: what the library picks from the free memory; 0 / 2: roots only / compact by request
if mode_arg != "auto":
    ctx.set_keep_trees(int(mode_arg))
elif n_slots * 0.2501 + 8 < free0 / 2**30 * 0.9:
    ctx.set_keep_trees(2)                  # a rehearsal at a size whose full trees WOULD fit: compact by request
ds = ctx.dataset(cfg)
dt = time.time() - t0
mode = ds.tree_mode
print("the dataset keeps: %s" % {1: "every node", 2: "block roots and up (compact)", 0: "roots only"}[mode], flush=True)
assert mode != 1
free1, _ = torch.cuda.mem_get_info()
perms = n_slots * (35 * n_cells - 1)
print("built %d slots x 2^22 cells (%.1f TiB) in %.1f s: %.3e perm/s, %.2f GB/s hashed; device memory in use after the build %.2f GiB" %
      (n_slots, n_slots * n_cells * cs / 2**40, dt, perms / dt, n_slots * n_cells * cs / dt / 1e9, (free0 - free1) / 2**30), flush=True)
roots = ds.local_roots()
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bigslots.json")))
hexroot = lambda a: np.asarray(a, dtype=np.uint8).tobytes()[::-1].hex()      # noqa: E731
ok_fixture = [hexroot(r) for r in roots[:8]] == gold["slot_roots_hex"][:min(8, n_slots)]
print("roots of slots 0..7 equal tests/golden/bigslots.json:", ok_fixture, flush=True)
t1 = time.time()
root = ds.root()
ok_tree = bool(np.array_equal(root, C.merkle_root(roots)))
print("dataset root %s (%d-level tree over the %d roots) equals the oracle's tree over the same roots: %s  (%.3f s)" % (hexroot(root), len(ds.ctx.merkle_tree(roots[:n_slots])) - 1 if False else c["maxLog2NSlots"], n_slots, ok_tree, time.time() - t1), flush=True)
slot = min(7, n_slots - 1)
lat = []
for s_ in (0, n_slots // 2, n_slots - 1, slot):      # the first call also sizes the context's scratch
    t2 = time.time()
    pi = ds.proof_input(s_, 1234567)
    lat.append(round(time.time() - t2, 4))
dt_pi = lat[-1]
print("proof-input latency (slots 0, %d, %d, %d): %s s" % (n_slots // 2, n_slots - 1, slot, lat), flush=True)
text = pi.json()
d, sroot, e = pi.roots()
to_int = lambda a: int.from_bytes(np.asarray(a, dtype=np.uint8).tobytes(), "little")   # noqa: E731
prf = {"dataSetRoot": to_int(d), "entropy": to_int(e), "nCells": n_cells, "nSlots": n_slots, "slotIndex": slot, "slotRoot": to_int(sroot),
       "slotProof": {"merklePath": [to_int(x) for x in pi.slot_proof()]},
       "proofInputs": [{"cellData": pi.cell_data()[i].tobytes(), "merkleProof": {"merklePath": [to_int(x) for x in pi.merkle_paths()[i]]}}
                       for i in range(c["nSamples"])]}
ok_circuit = bool(P.circuit_check(prf, c))
print("proof input of slot %d: %.4f s, %d bytes of input.json; passes the circuit-side checker: %s" %
      (slot, dt_pi, len(text), ok_circuit), flush=True)
th.join()
ok_oracle = all(np.array_equal(roots[s], oracle_roots[s]) for s in checks)
print("slot roots %s equal the C oracle's: %s" % (checks, ok_oracle), flush=True)
print(json.dumps({"n_slots": n_slots, "TiB_hashed": n_slots * n_cells * cs / 2**40, "seconds": round(dt, 1), "perms_per_s": perms / dt,
                  "GB_per_s": n_slots * n_cells * cs / dt / 1e9, "device_GiB_in_use_after_build": round((free0 - free1) / 2**30, 2),
                  "tree_mode": mode, "proof_input_s": round(dt_pi, 4), "proof_input_latencies_s": lat, "dataset_root_hex": hexroot(root),
                  "checks": {"slots_0_7_vs_fixture": ok_fixture, "dataset_tree_vs_oracle": ok_tree, "slots_vs_c_oracle": {str(s): bool(np.array_equal(roots[s], oracle_roots[s])) for s in checks},
                             "circuit_check": ok_circuit}}))
sys.exit(0 if (ok_fixture and ok_tree and ok_circuit and ok_oracle) else 1)
