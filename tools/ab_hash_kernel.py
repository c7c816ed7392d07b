"""A/B of library variants on k_hash_cells alone: one launch over the 8 GiB slot of configs[2] (HIP events), alternating fresh
processes, the slot root checked against the oracle-only fixture for every variant.
Usage: ab_hash_kernel.py <rounds> default|<lib.so> ...        (child mode: ab_hash_kernel.py --child)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--child":
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    ctx = pkg.Context(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    n_cells, cs, bs = 1 << 22, 2048, 65536
    buf = torch.empty((n_cells, cs), dtype=torch.uint8, device=dev)
    leaves = torch.empty((n_cells, 32), dtype=torch.uint8, device=dev)
    ctx.gen_fake_cells_dev(ctx.slot_seed(12345, 0), 0, n_cells, cs, buf.data_ptr())
    t = ctx.slot_trees_dev(buf.data_ptr(), 1, cs, bs, n_cells)
    torch.cuda.synchronize()
    root = t.roots()[0].tobytes()[::-1].hex()
    t.free()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["config3"]["slot_root_hex"]
    ms = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ctx.hash_cells_dev(buf.data_ptr(), cs, n_cells, leaves.data_ptr())
        b.record(stream)
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    print(json.dumps({"ms": [round(x, 3) for x in ms], "root_ok": root == gold}))
    sys.exit(0)
rounds, libs = int(sys.argv[1]), sys.argv[2:]
res = {l: [] for l in libs}
for rnd in range(rounds):
    for l in libs:
        env = dict(os.environ, CODEX_P2_LIB=os.path.abspath(l)) if l != "default" else dict(os.environ)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
        line = [x for x in r.stdout.splitlines() if x.startswith("{")]
        d = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
        res[l].append(d)
        print(l, d, flush=True)
summary = {l: {"min_ms": min(min(d["ms"][1:]) for d in v if "ms" in d), "median_of_medians_ms": sorted(sorted(d["ms"][1:])[len(d["ms"][1:]) // 2] for d in v if "ms" in d)[len(v) // 2],
               "root_ok": all(d.get("root_ok") for d in v)} for l, v in res.items()}
base = summary[libs[0]]["median_of_medians_ms"]
for l in libs:
    summary[l]["vs_first"] = round(summary[l]["median_of_medians_ms"] / base, 5)
print(json.dumps(summary))
