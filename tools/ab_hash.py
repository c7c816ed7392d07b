"""A/B of library variants on the 8 GiB slot-root leg (k_hash_cells dominated)."""
import os, subprocess, sys, json
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(2):
    for l in libs:
        env = dict(os.environ, CODEX_P2_LIB=os.path.abspath(l)) if l != "default" else dict(os.environ)
        out = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().split("\n")[-1])
            res[l].append(d["extra"]["slot_root"]["build_ms"])
        except Exception as e:
            res[l].append("ERR " + out.stderr[-300:])
for l in libs:
    print(l, res[l])
