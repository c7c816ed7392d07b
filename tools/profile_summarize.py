#!/usr/bin/env python3
"""Turns gpurun_out/prof_<round>/ (tools/profile_round.sh) into the committed summaries under profiles/."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
O = os.path.join(ROOT, "gpurun_out", "prof_" + R)
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
newest = lambda pattern: sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]   # gpurun_out/ accumulates runs
stats = newest(os.path.join(O, "kt", "*", "*_kernel_stats.csv"))[0]
rows = [r for r in csv.DictReader(open(stats)) if r["Name"].startswith("cp2k::")]
with open(os.path.join(P, "%s_bench_kernel_stats.csv" % R), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in csv.DictReader(open(stats)):
        w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
counters = {}
for d in ("fetch", "write", "sq", "sq2"):
    fs = newest(os.path.join(O, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Kernel_Name"].startswith("cp2k::k_permute_batch"):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        counters[k] = {"per_launch_avg": sum(v) / len(v), "launches": len(v)}
n = 1 << 24
perm = [r for r in rows if "k_permute_batch" in r["Name"]][0]
avg_ms = float(perm["AverageNs"]) * 1e-6
fetch_kb, write_kb = counters["FETCH_SIZE"]["per_launch_avg"], counters["WRITE_SIZE"]["per_launch_avg"]
res = {
    "round": R, "kernel": "cp2k::k_permute_batch", "workload": "2^24 states (configs[1])",
    "commands": "tools/profile_round.sh: rocprofv3 --kernel-trace --stats / --pmc <group> (separate passes) -- python3 bench.py ...",
    "kernel_trace_avg_launch_ms": avg_ms,
    "counters": counters,
    "correction": "gfx950: FETCH_SIZE tallies 64 B per 128-B request for 16 B/lane streaming reads -> doubled "
                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; both reported in KiB",
    "hbm_read_bytes_per_launch": int(fetch_kb * 1024 * 2), "hbm_write_bytes_per_launch": int(write_kb * 1024),
    "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024),
    "algorithmic_bytes_per_launch": 192 * n,
    "valu_insts_per_wave": counters["SQ_INSTS_VALU"]["per_launch_avg"] / (n / 64),
    "shader_clock_GHz_from_GRBM_GUI_ACTIVE": counters["GRBM_GUI_ACTIVE"]["per_launch_avg"] / 8 / (avg_ms * 1e-3) / 1e9,
    # SQ_ACTIVE_INST_VALU counts quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
    # (a 2-cycle VOP2 instruction still counts one quad-cycle, so the raw ratio can read slightly above 1)
    "valu_busy_frac_raw": counters["SQ_ACTIVE_INST_VALU"]["per_launch_avg"] * 4 / (counters["GRBM_GUI_ACTIVE"]["per_launch_avg"] / 8 * 1024),
    "valu_busy_frac": min(1.0, counters["SQ_ACTIVE_INST_VALU"]["per_launch_avg"] * 4 / (counters["GRBM_GUI_ACTIVE"]["per_launch_avg"] / 8 * 1024)),
}
json.dump(res, open(os.path.join(P, "%s_permute_batch_traffic.json" % R), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("kernel_trace_avg_launch_ms", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch",
                                      "valu_insts_per_wave", "shader_clock_GHz_from_GRBM_GUI_ACTIVE")}))
for r in rows:
    print(r["Name"][:40], r["Calls"], "avg_ms=%.3f" % (float(r["AverageNs"]) * 1e-6))
