#!/usr/bin/env python3
"""Turns gpurun_out/prof_<round>/ (tools/profile_round.sh) into the committed summaries under profiles/."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
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
        name = r["Name"] if r["Name"].startswith("cp2k::") or len(r["Name"]) < 100 else r["Name"][:96] + "..."   # torch's input generators
        w.writerow([name] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
# per-launch durations of the headline kernel from the trace itself (the --stats average includes the first launch)
trace = newest(os.path.join(O, "kt", "*", "*_kernel_trace.csv"))
launches = []
if trace:
    for r in csv.DictReader(open(trace[0])):
        # only the 2^24-state launches of the timed loop: the ingest leg also runs this kernel on 2^20-state chunks
        if r["Kernel_Name"].startswith("cp2k::k_permute_batch") and int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) == (1 << 24):
            launches.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)


def counters_of(dirs, kernel):
    out = {}
    for d in dirs:
        fs = newest(os.path.join(O, d, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if r["Kernel_Name"].startswith(kernel):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = {"per_launch_avg": sum(v) / len(v), "per_launch_max": max(v), "launches": len(v)}
    return out


counters = counters_of(("fetch", "write", "sq", "sq2"), "cp2k::k_permute_batch")
n = 1 << 24
perm = [r for r in rows if "k_permute_batch" in r["Name"]][0]
avg_ms = sum(launches) / len(launches) if launches else float(perm["AverageNs"]) * 1e-6
fetch_kb, write_kb = counters["FETCH_SIZE"]["per_launch_avg"], counters["WRITE_SIZE"]["per_launch_avg"]
res = {
    "round": R, "kernel": "cp2k::k_permute_batch", "workload": "2^24 states (configs[1]), every element uniform in [0, r)",
    "commands": "tools/profile_round.sh: rocprofv3 --kernel-trace --stats / --pmc <group> (separate passes) -- python3 bench.py ...",
    "kernel_trace_avg_launch_ms": avg_ms,
    "kernel_trace_launches_ms": [round(x, 4) for x in launches],
    "kernel_trace_note": "average over the 2^24-state launches of the kernel trace (3 warm-up + 10 timed), listed one by one; the --stats CSV "
                         "row of this kernel also averages in the 2^20-state chunk launches of the ingest leg's host-array call "
                         "(%s calls, %.3f ms on average), so it is not comparable; bench.py's own average (HIP events) covers the 10 timed launches"
                         % (perm["Calls"], float(perm["AverageNs"]) * 1e-6),
    "kernel_trace_avg_excluding_first_ms": (sum(launches[1:]) / len(launches[1:])) if len(launches) > 1 else None,
    "counters": counters,
    "correction": "gfx950: FETCH_SIZE tallies 64 B per 128-B request for 16 B/lane streaming reads -> doubled "
                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; both reported in KiB",
    "hbm_read_bytes_per_launch": int(fetch_kb * 1024 * 2), "hbm_write_bytes_per_launch": int(write_kb * 1024),
    "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024),
    "algorithmic_bytes_per_launch": 192 * n,
    "valu_insts_per_wave": counters["SQ_INSTS_VALU"]["per_launch_avg"] / (n / 64),
    "mad_u64_u32_per_permutation": 33120,
    "shader_clock_GHz_from_GRBM_GUI_ACTIVE": counters["GRBM_GUI_ACTIVE"]["per_launch_avg"] / 8 / (avg_ms * 1e-3) / 1e9,
    # SQ_ACTIVE_INST_VALU counts quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs.  Unclamped: in this
    # kernel every VALU instruction costs about one 4-cycle issue slot (profiles/r02_marginal_cost_probe.txt), so the
    # ratio is instructions x 4 / cycles and can read slightly above 1 when the PMC pass and the timing pass clock differently
    "valu_busy_frac_raw": counters["SQ_ACTIVE_INST_VALU"]["per_launch_avg"] * 4 / (counters["GRBM_GUI_ACTIVE"]["per_launch_avg"] / 8 * 1024),
    "issue_cycles_per_wave_model": None,
}
json.dump(res, open(os.path.join(P, "%s_permute_batch_traffic.json" % R), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("kernel_trace_avg_launch_ms", "kernel_trace_avg_excluding_first_ms", "hbm_bytes_per_launch",
                                      "algorithmic_bytes_per_launch", "valu_insts_per_wave", "shader_clock_GHz_from_GRBM_GUI_ACTIVE",
                                      "valu_busy_frac_raw")}))
# config 3's kernel: the launch over the whole 8 GiB slot is the one with the largest counter value
hc = counters_of(("hfetch", "hwrite"), "cp2k::k_hash_cells")
if "FETCH_SIZE" in hc and "WRITE_SIZE" in hc:
    rd, wr = hc["FETCH_SIZE"]["per_launch_max"] * 1024 * 2, hc["WRITE_SIZE"]["per_launch_max"] * 1024
    h = {"round": R, "kernel": "cp2k::k_hash_cells", "workload": "configs[2]: 2^22 cells x 2048 B (8 GiB slot), the largest launch of the run",
         "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr),
         "algorithmic_read_bytes": 1 << 33, "algorithmic_write_bytes": (1 << 22) * 32,
         "read_over_algorithmic": rd / (1 << 33), "correction": res["correction"]}
    json.dump(h, open(os.path.join(P, "%s_hash_cells_traffic.json" % R), "w"), indent=1)
    print(json.dumps(h))
for r in rows:
    print(r["Name"][:40], r["Calls"], "avg_ms=%.3f" % (float(r["AverageNs"]) * 1e-6))
