"""One-off full-size parity for config 4: 4096 slots x 2^12 cells x 2048 B, nSamples=100, maxDepth=32.
GPU: every slot tree, the dataset root, the proof input of one slot.  CPU: the same with the C oracle (all 4096 slot
roots: ~5.9e8 permutations, a few minutes on 16 threads) + the Python restatement for indexing and JSON.
The two input.json texts must be identical."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as g
from oracle_helpers import expected_proof_input_fast
pkg = g.load_package()
C, P = g.load_oracle()
ctx = pkg.Context(0)
c = dict(maxDepth=32, maxLog2NSlots=12, cellSize=2048, blockSize=65536, nSlots=4096, nCells=1 << 12, nSamples=100, seed=12345)
slot, entropy = 1234, 1234567
t = time.perf_counter()
ds = ctx.dataset(pkg.make_config(**c))
text = ds.proof_input(slot, entropy).json()
print("GPU: trees of 4096 slots + proof input of slot %d in %.2f s; dataset root %s" % (slot, time.perf_counter() - t, ds.root().tobytes()[::-1].hex()), flush=True)
t = time.perf_counter()
want = P.export_json(expected_proof_input_fast(C, P, c, slot, entropy, threads=max(1, min(16, len(os.sched_getaffinity(0))))))
print("CPU oracle: %.1f s" % (time.perf_counter() - t), flush=True)
print("input.json sha256 GPU %s" % hashlib.sha256(text.encode()).hexdigest())
print("input.json sha256 CPU %s" % hashlib.sha256(want.encode()).hexdigest())
assert text == want, "CONFIG 4 JSON MISMATCH"
print("config 4 (4096 slots) input.json byte-identical, %d bytes" % len(text))
