"""A/B of kernel variants on ONE box in ONE gpurun call: interleaved rounds, one child process per sample.
  python tools/ab_kernels.py default build/variants/libcodex_p2_x.so ...   (default = the in-tree library)
Output digests must agree between variants (same inputs): a faster wrong kernel is not a variant."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
rounds = int(os.environ.get("AB_ROUNDS", "3"))
res = {l: [] for l in libs}
for rnd in range(rounds):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["CODEX_P2_LIB"] = os.path.abspath(l)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_one.py")], env=env, capture_output=True, text=True)
        try:
            res[l].append(json.loads(out.stdout.strip().split("\n")[-1]))
        except Exception:
            res[l].append({"error": out.stderr[-300:]})
        print(l, res[l][-1], flush=True)
base = None
for l in libs:
    ok = [r for r in res[l] if "error" not in r]
    if not ok:
        print("%-50s FAILED" % l)
        continue
    pm = sorted(r["perm_ms_median"] for r in ok)[len(ok) // 2]
    hm = sorted(r["hash_ms"] for r in ok)[len(ok) // 2]
    if base is None:
        base = (pm, hm, ok[0]["perm_digest"], ok[0]["hash_digest"])
    same = ok[0]["perm_digest"] == base[2] and ok[0]["hash_digest"] == base[3]
    print("%-50s permute %.3f ms (%+.2f%%)   hash_cells %.3f ms (%+.2f%%)   outputs %s" %
          (l, pm, 100 * (pm / base[0] - 1), hm, 100 * (hm / base[1] - 1), "identical" if same else "DIFFER"))
