"""A/B of library variants in ONE process per variant but interleaved rounds on the same device (dev aid)."""
import os, subprocess, sys, json
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, CODEX_P2_LIB=os.path.abspath(l)) if l != "default" else dict(os.environ)
        out = subprocess.run([sys.executable, "bench.py", "--steps", "8", "--warmup", "2", "--no-extra", "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().split("\n")[-1])
            res[l].append(round(d["roofline"]["avg_launch_ms"], 3))
        except Exception as e:
            res[l].append("ERR " + out.stderr[-200:])
for l in libs:
    print(l, res[l])
