"""BASELINE.json configs[3] ("end-to-end proof input: nSamples=100, maxDepth=32, 4096 slots batched, 1 GPU") with config 3's slot
geometry -- 8 GiB slots (cellSize 2048, nCells 2^22) -- instead of SURVEY.md 8(d)'s 8 MiB scale-down: 4096 slots x 8 GiB do not
fit HBM (32 TiB of data, 1 TiB of tree nodes), but the streamed roots-only build does not need them to: one pass over the
(device-generated) data, the proof-input body of every slot made while the trees of its batch exist, 32-byte roots kept.
Emits all 4096 input.json texts (serialise only).  Pinned by:
  * slot roots 0..7 = tests/golden/bigslots.json; the dataset root = the oracle's tree over the 4096 roots;
  * the COMPLETE input.json of two slots, byte for byte, against the oracle: their slot-dependent part (root, sampled indices,
    cells, merged paths) recomputed by the C oracle + Python restatement on the host while the GPU runs, dataSetRoot / slotProof
    from the oracle's dataset tree;
  * oracle.circuit_check on one of them.
Usage (CP2_TRACE=1 for progress):  config4_nominal.py [n_slots]"""
import hashlib, importlib.util, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
C, P = g.load_oracle()
spec = importlib.util.spec_from_file_location("mb", os.path.join(ROOT, "tests", "golden", "make_bigslots_golden.py"))
mb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mb)
n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
entropy = 1234567
c = dict(maxDepth=32, maxLog2NSlots=max(1, (n_slots - 1).bit_length()), cellSize=2048, blockSize=65536, nSlots=n_slots, nCells=1 << 22, nSamples=100, seed=12345)
cfg = pkg.make_config(**c)
ctx = pkg.Context(0)
threads = max(1, min(16, len(os.sched_getaffinity(0))))
check_slots = sorted({min(7, n_slots - 1), (n_slots * 5) // 8})
parts = {}


def oracle_side():
    for s in check_slots:
        t = time.time()
        parts[s] = mb.slot_part(c, s, entropy, max(1, threads - 4))
        print("  [host] oracle: slot %d (root, 100 sampled cells, merged paths) in %.0f s" % (s, time.time() - t), flush=True)


th = threading.Thread(target=oracle_side)
th.start()
free0, _ = torch.cuda.mem_get_info()
if n_slots * 0.2501 + 8 < free0 / 2**30 * 0.9:
    ctx.set_keep_trees(2)                      # a rehearsal whose trees WOULD fit: compact by request (the full run gets it by itself)
t0 = time.time()
ds = ctx.dataset_streamed(cfg, entropy, threads=threads, group_slots=1)
t1 = time.time()
assert not ds.keeps_trees
nbytes = ds.export_streamed(None, threads=threads)
t2 = time.time()
free1, _ = torch.cuda.mem_get_info()
perms = n_slots * (35 * (1 << 22) - 1) + 200 * n_slots
print("streamed, trees dropped batch by batch: %d slots x 8 GiB (%.1f TiB), bodies of all slots in %.1f s + heads %.2f s: %.2f witnesses/s with JSON (%.2f GB of text), %.3e perm/s; "
      "device memory in use after the build %.2f GiB" % (n_slots, n_slots / 128, t1 - t0, t2 - t1, n_slots / (t2 - t0), nbytes / 1e9, perms / (t2 - t0), (free0 - free1) / 2**30), flush=True)
roots = ds.local_roots()
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bigslots.json")))
hexroot = lambda a: np.asarray(a, dtype=np.uint8).tobytes()[::-1].hex()      # noqa: E731
ok_fixture = [hexroot(r) for r in roots[:8]] == gold["slot_roots_hex"][:min(8, n_slots)]
ok_tree = bool(np.array_equal(ds.root(), C.merkle_root(roots)))
print("roots of slots 0..7 equal bigslots.json: %s; dataset root %s equals the oracle's tree over the %d roots: %s" % (ok_fixture, hexroot(ds.root()), n_slots, ok_tree), flush=True)
th.join()
ok_json, ok_circuit = {}, None
dset = mb.to_int(C.merkle_tree(roots))
for s in check_slots:
    prf = {"dataSetRoot": dset[-1][0], "entropy": entropy, "nCells": c["nCells"], "nSlots": n_slots, "slotIndex": s, "slotRoot": parts[s]["root"],
           "slotProof": P.pad_merkle_proof(P.merkle_proof(dset, s), c["maxLog2NSlots"]), "proofInputs": parts[s]["proofInputs"]}
    want = P.export_json(prf)
    got = ds.streamed_json(s)
    ok_json[str(s)] = got == want
    print("input.json of slot %d: %d bytes, sha256 %s, equals the oracle's text byte for byte: %s" % (s, len(got), hashlib.sha256(got.encode()).hexdigest()[:16], got == want), flush=True)
    if ok_circuit is None:
        ok_circuit = bool(P.circuit_check(prf, c))
print("circuit-side checker on slot %d: %s" % (check_slots[0], ok_circuit), flush=True)
# the dataset stays usable: a proof input for ANOTHER entropy, from what it kept of the trees
lat = []
for s_ in (1, n_slots // 3, n_slots - 1):
    t = time.time()
    ds.proof_input(s_, 7654321)
    lat.append(round(time.time() - t, 4))
print("the dataset keeps: %s; proof inputs for a new entropy afterwards: %s s" % ({1: "every node", 2: "block roots and up (compact)", 0: "roots only"}[ds.tree_mode], lat), flush=True)
print(json.dumps({"n_slots": n_slots, "TiB_hashed": n_slots / 128, "build_with_bodies_s": round(t1 - t0, 1), "heads_s": round(t2 - t1, 2), "json_GB": nbytes / 1e9,
                  "witnesses_per_s_with_json": n_slots / (t2 - t0), "perms_per_s": perms / (t2 - t0), "device_GiB_in_use_after_build": round((free0 - free1) / 2**30, 2),
                  "tree_mode": ds.tree_mode, "new_entropy_proof_input_s": lat, "checks": {"slots_0_7_vs_fixture": ok_fixture, "dataset_tree_vs_oracle": ok_tree, "input_json_vs_oracle": ok_json, "circuit_check": ok_circuit}}))
sys.exit(0 if (ok_fixture and ok_tree and all(ok_json.values()) and ok_circuit) else 1)
