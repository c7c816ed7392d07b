#!/usr/bin/env python3
"""Turns gpurun_out/prof_<round>/ (tools/profile_round.sh) into the committed summaries under profiles/."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r03"
O = os.path.join(ROOT, "gpurun_out", "prof_" + R)
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
newest = lambda pattern: sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]   # gpurun_out/ accumulates runs
def main_process_file(pattern, key_col):
    """The kernel-trace directory holds one set of files per PROCESS (bench.py's child legs are traced too): the bench process
    itself is the one with the most k_permute_batch launches."""
    best, best_n = None, -1
    for f in sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True):   # newest first: gpurun_out/ accumulates runs
        n = sum(1 for r in csv.DictReader(open(f)) if "cp2k::k_permute_batch" in r.get(key_col, "") and
                (key_col != "Name" or int(r["Calls"]) >= 1)) if key_col != "Name" else \
            sum(int(r["Calls"]) for r in csv.DictReader(open(f)) if "cp2k::k_permute_batch" in r["Name"])
        if n > best_n:
            best, best_n = f, n
    return best


stats = main_process_file(os.path.join(O, "kt", "*", "*_kernel_stats.csv"), "Name")
rows = [r for r in csv.DictReader(open(stats)) if "cp2k::" in r["Name"]]   # templates print as "void cp2k::k<..>(..)"
with open(os.path.join(P, "%s_bench_kernel_stats.csv" % R), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in csv.DictReader(open(stats)):
        name = r["Name"] if "cp2k::" in r["Name"] or len(r["Name"]) < 100 else r["Name"][:96] + "..."   # torch's input generators
        w.writerow([name] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
# per-launch durations of the headline kernel from the trace itself (the --stats average includes the first launch)
trace = [stats.replace("_kernel_stats.csv", "_kernel_trace.csv")] if os.path.exists(stats.replace("_kernel_stats.csv", "_kernel_trace.csv")) else []
launches = []
if trace:
    for r in csv.DictReader(open(trace[0])):
        # only the 2^24-state launches of the timed loop: the ingest leg also runs this kernel on 2^20-state chunks
        if "cp2k::k_permute_batch" in r["Kernel_Name"] and int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) == (1 << 24):
            launches.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)


def counters_of(dirs, kernel, grid=None):
    """Per-launch averages of every counter collected for `kernel` in the given passes; `_pass_kernel_ms[d]` = that kernel's own
    average duration INSIDE pass d (its dispatch timestamps in the counter CSV), so that a counter is only ever set against
    the time and the clock of the pass it was collected in."""
    out, pass_ms = {}, {}
    for d in dirs:
        fs = newest(os.path.join(O, d, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        agg = collections.defaultdict(list)
        dur = {}
        for r in csv.DictReader(open(fs[0])):
            if kernel in r["Kernel_Name"] and (grid is None or int(r["Grid_Size"]) == grid):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        for k, v in agg.items():
            out[k] = {"per_launch_avg": sum(v) / len(v), "per_launch_max": max(v), "launches": len(v), "pass": d}
        if dur:
            pass_ms[d] = sum(dur.values()) / len(dur)
    out["_pass_kernel_ms"] = pass_ms
    return out


counters = counters_of(("fetch", "write", "sq", "sq2"), "cp2k::k_permute_batch", grid=1 << 24)
pass_ms = counters.pop("_pass_kernel_ms")
box = open(os.path.join(O, "box.txt")).read().strip().splitlines() if os.path.exists(os.path.join(O, "box.txt")) else []
n = 1 << 24
perm = [r for r in rows if "k_permute_batch" in r["Name"]][0]
avg_ms = sum(launches) / len(launches) if launches else float(perm["AverageNs"]) * 1e-6
fetch_kb, write_kb = counters["FETCH_SIZE"]["per_launch_avg"], counters["WRITE_SIZE"]["per_launch_avg"]
res = {
    "round": R, "kernel": "cp2k::k_permute_batch", "workload": "2^24 states (configs[1]), every element uniform in [0, r)",
    "commands": "tools/profile_round.sh: rocprofv3 --kernel-trace --stats / --pmc <group> (separate passes) -- python3 bench.py ...",
    "kernel_trace_avg_launch_ms": avg_ms,
    "kernel_trace_launches_ms": [round(x, 4) for x in launches],
    "kernel_trace_note": "average over the 2^24-state launches of the kernel trace (5 warm-up + 20 timed: the driver's command), listed one by one; the --stats CSV "
                         "row of this kernel holds %s calls averaging %.3f ms (the same launches when no other leg runs this kernel); bench.py's own "
                         "average (HIP events) covers the 20 timed launches only" % (perm["Calls"], float(perm["AverageNs"]) * 1e-6),
    "kernel_trace_avg_excluding_first_ms": (sum(launches[1:]) / len(launches[1:])) if len(launches) > 1 else None,
    "counters": counters,
    "correction": "gfx950: FETCH_SIZE tallies 64 B per 128-B request for 16 B/lane streaming reads -> doubled "
                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; both reported in KiB",
    "hbm_read_bytes_per_launch": int(fetch_kb * 1024 * 2), "hbm_write_bytes_per_launch": int(write_kb * 1024),
    "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024),
    "algorithmic_bytes_per_launch": 192 * n,
    "valu_insts_per_wave": counters["SQ_INSTS_VALU"]["per_launch_avg"] / (n / 64),
    "box": box,
    "kernel_ms_inside_each_pmc_pass": {k: round(v, 4) for k, v in pass_ms.items()},
}
# The instruction-issue view, from ONE pass (`sq`): GRBM_GUI_ACTIVE is summed over the 8 XCDs, so /8 = shader cycles of the
# launch; 1024 SIMDs issue SQ_INSTS_VALU wave-instructions in them.  No time and no clock enters any ratio below; the clock of
# the pass (cycles / the kernel's duration inside the same pass) is reported beside them.
# The roof is the class-weighted floor of the kernel's own instruction stream (tools/valu_roof.py): CDNA4's SIMD-32 issues a
# wave64 VALU instruction over 2 cycles, the multiplies and the other half-rate integer forms over 4 -- NOT "4 = full rate".
cyc = counters["GRBM_GUI_ACTIVE"]["per_launch_avg"] / 8
insts = counters["SQ_INSTS_VALU"]["per_launch_avg"]
sq_ms = pass_ms.get("sq")
waves = n / 64
cyc_wave_perm = cyc * 1024 / waves                      # shader cycles one SIMD spends per wave-permutation (64 permutations)
res["valu_issue"] = {
    "bound": "valu-issue", "pass": "sq (rocprofv3 --pmc SQ_INSTS_VALU ... GRBM_GUI_ACTIVE, one pass)",
    "cycles_per_valu_instruction": round(cyc * 1024 / insts, 4),
    "cycles_per_wave_permutation": round(cyc_wave_perm, 1),
    "valu_insts_per_permutation": insts / waves,
    "shader_cycles_per_launch": cyc, "valu_wave_insts_per_launch": insts,
    "kernel_ms_in_this_pass": sq_ms, "shader_clock_GHz_in_this_pass": (cyc / (sq_ms * 1e-3) / 1e9) if sq_ms else None,
}
cls_path = os.path.join(P, "%s_valu_classes.json" % R)
if os.path.exists(cls_path):
    cl = json.load(open(cls_path))
    mad = next(r for r in cl["opcodes"] if r["opcode"] == "v_mad_u64_u32")
    non_mad = cl["valu_insts_per_permutation_from_isa"] - mad["per_permutation"]
    floors = {"nominal_per_opcode": cl["class_floor_cycles_nominal"],
              "nominal_if_every_non_multiply_issued_in_2": 4.0 * mad["per_permutation"] + 2.0 * non_mad,
              "homogeneous_stream_prices_per_opcode_NOT_a_floor": cl["class_floor_cycles_measured"]}
    v = res["valu_issue"]
    v["instruction_classes_per_permutation"] = cl["per_class"]
    v["class_floor_cycles"] = {k: (round(x, 1) if x else None) for k, x in floors.items()}
    v["frac_of_class_floor"] = {k: (round(x / cyc_wave_perm, 4) if x else None) for k, x in floors.items()}
    v["classes_source"] = "profiles/%s_valu_classes.json (tools/valu_roof.py: dynamic opcode counts from the ISA; ISA total %d vs SQ_INSTS_VALU %.0f per permutation)" % (
        R, cl["valu_insts_per_permutation_from_isa"], insts / waves)
    v["rates_source"] = {"nominal": "MI355X_MICROARCH.md: SIMD-32, `v_fma_f32 (wave64) 2 cyc`; plain 32-bit add/logic/shift-right/move 2 cycles, every other form 4",
                         "measured": cl["measured_rates_source"]}
    # int_mul_ops/s against the measured peak of the multiplier, inside this pass (counts / this pass's cycles)
    mad_rate = mad["measured_saturated_cycles"] or 4.0
    lane_mul_per_cycle = mad["per_permutation"] * n / cyc                   # 64-bit multiply-accumulates per shader cycle, whole chip
    peak_per_cycle = 1024 * 64 / mad_rate
    if "SQ_WAVE_CYCLES" in counters:
        v["waves_per_simd"] = {"by_vgprs": "102 VGPRs -> 4 (hipcc -Rpass-analysis=kernel-resource-usage)",
                               "measured_SQ_WAVE_CYCLES_x4_over_cycles_x_1024_simds": round(counters["SQ_WAVE_CYCLES"]["per_launch_avg"] * 4 / (cyc * 1024), 2)}
    v["int_mul"] = {"v_mad_u64_u32_per_permutation": mad["per_permutation"], "of_which_in_the_80_sboxes": 33120,
                    "lane_multiplies_per_cycle_chip": round(lane_mul_per_cycle, 1),
                    "ubench_peak_per_cycle_chip": round(peak_per_cycle, 1), "saturated_cycles_per_v_mad_u64_u32": mad_rate,
                    "frac_of_ubench_peak": round(lane_mul_per_cycle / peak_per_cycle, 4),
                    "int_mul_ops_per_s_at_this_pass_clock": (lane_mul_per_cycle * cyc / (sq_ms * 1e-3)) if sq_ms else None,
                    "share_of_issue_cycles": round(mad["per_permutation"] * mad_rate / cyc_wave_perm, 4)}
    v["explanation"] = ("CDNA4 SIMD-32: a wave64 VALU instruction issues over 2 cycles, the multiplies and the other half-rate integer forms over 4. "
                        "The kernel's stream is %d %% v_mad_u64_u32, %d %% other half-rate forms and %d %% plain 2-cycle forms; priced opcode by opcode at those "
                        "nominal rates its floor is class_floor_cycles.nominal_per_opcode, and the measured cycles per wave-permutation are that floor / "
                        "frac_of_class_floor.nominal_per_opcode (the looser bound `..._issued_in_2` pretends every non-multiply were a 2-cycle form). "
                        "Priced instead at what each opcode costs in a saturated stream of ITSELF (tools/ubench_classes.hip, same counters) the sum lies ABOVE "
                        "the kernel's measured cycles: the mixed stream issues faster than its opcodes do alone (a v_mad_u64_u32 costs at most %.2f cycles "
                        "inside the kernel against %.2f alone), so that sum is no floor; it prices SURVEY.md 8(d)'s int-mul peak. "
                        "The issue port is nearly saturated by THIS stream; what is left is the instruction count, above all the multiplies"
                        % (round(100 * mad["share"]), round(100 * cl["per_class"].get("half-rate", 0) / cl["valu_insts_per_permutation_from_isa"]),
                           round(100 * cl["per_class"].get("simple", 0) / cl["valu_insts_per_permutation_from_isa"]),
                           (cyc_wave_perm - sum(r["per_permutation"] * r["priced_at"] for r in cl["opcodes"] if r["opcode"] != "v_mad_u64_u32")) / mad["per_permutation"],
                           mad_rate))
json.dump(res, open(os.path.join(P, "%s_permute_batch_traffic.json" % R), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("kernel_trace_avg_launch_ms", "kernel_trace_avg_excluding_first_ms", "hbm_bytes_per_launch",
                                      "algorithmic_bytes_per_launch", "valu_insts_per_wave", "valu_issue")}))
# config 3's kernel: the launches over the whole 8 GiB slot (grid = 2^22 cells; five per run: warm-up, build, three timed); their MEDIAN
# (the first such launch of a process reads about 1 % more: first touch of the freshly generated slot)
def median_of_grid(d, kernel, grid, counter):
    fs = newest(os.path.join(O, d, "*", "*_counter_collection.csv"))
    v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0]))
               if kernel in r["Kernel_Name"] and int(r["Grid_Size"]) == grid and r["Counter_Name"] == counter) if fs else []
    return (v[len(v) // 2], len(v), v[-1]) if v else (None, 0, None)


f_med, f_n, f_max = median_of_grid("hfetch", "cp2k::k_hash_cells", 1 << 22, "FETCH_SIZE")
w_med, w_n, w_max = median_of_grid("hwrite", "cp2k::k_hash_cells", 1 << 22, "WRITE_SIZE")
if f_med and w_med:
    rd, wr = f_med * 1024 * 2, w_med * 1024
    h = {"round": R, "kernel": "cp2k::k_hash_cells", "workload": "configs[2]: 2^22 cells x 2048 B (8 GiB slot): median of the %d launches over the whole slot" % f_n,
         "read_over_algorithmic_max_launch": f_max * 1024 * 2 / (1 << 33),
         "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr),
         "algorithmic_read_bytes": 1 << 33, "algorithmic_write_bytes": (1 << 22) * 32,
         "read_over_algorithmic": rd / (1 << 33), "correction": res["correction"]}
    # ---- the issue side of THIS kernel, from its own `hsq` pass and its own ISA (tools/valu_roof.py -> <round>_valu_classes_hash_cells.json)
    hc = counters_of(("hsq", "hsq2"), "cp2k::k_hash_cells", grid=1 << 22)
    h_pass_ms = hc.pop("_pass_kernel_ms")
    if "SQ_INSTS_VALU" in hc and "GRBM_GUI_ACTIVE" in hc:
        n_cells, steps = 1 << 22, 34
        waves_h = n_cells / 64
        cyc_h = hc["GRBM_GUI_ACTIVE"]["per_launch_avg"] / 8
        insts_h = hc["SQ_INSTS_VALU"]["per_launch_avg"]
        cyc_wave_cell = cyc_h * 1024 / waves_h
        hv = {"bound": "valu-issue", "pass": "hsq (rocprofv3 --pmc SQ_INSTS_VALU ... GRBM_GUI_ACTIVE, one pass, the %d launches over the whole 8 GiB slot)" % hc["SQ_INSTS_VALU"]["launches"],
              "valu_insts_per_cell": insts_h / waves_h, "valu_insts_per_permutation": insts_h / waves_h / steps,
              "cycles_per_wave_cell": round(cyc_wave_cell, 1), "cycles_per_wave_permutation": round(cyc_wave_cell / steps, 1),
              "cycles_per_valu_instruction": round(cyc_h * 1024 / insts_h, 4),
              "shader_cycles_per_launch": cyc_h, "valu_wave_insts_per_launch": insts_h,
              "kernel_ms_in_this_pass": h_pass_ms.get("hsq"),
              "shader_clock_GHz_in_this_pass": (cyc_h / (h_pass_ms["hsq"] * 1e-3) / 1e9) if h_pass_ms.get("hsq") else None,
              "waves_per_simd": {"by_registers": 3, "measured_SQ_WAVE_CYCLES_x4_over_cycles_x_1024_simds":
                                 round(hc["SQ_WAVE_CYCLES"]["per_launch_avg"] * 4 / (cyc_h * 1024), 2) if "SQ_WAVE_CYCLES" in hc else None},
              "other_counters_per_launch": {k: v["per_launch_avg"] for k, v in hc.items() if k not in ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")}}
        hcls_path = os.path.join(P, "%s_valu_classes_hash_cells.json" % R)
        if os.path.exists(hcls_path):
            cl = json.load(open(hcls_path))
            mad = next(r for r in cl["opcodes"] if r["opcode"] == "v_mad_u64_u32")
            n_isa = cl["valu_insts_per_permutation_from_isa"]
            floors = {"nominal_per_opcode": cl["class_floor_cycles_nominal"],
                      "nominal_if_every_non_multiply_issued_in_2": 4.0 * mad["per_permutation"] + 2.0 * (n_isa - mad["per_permutation"]),
                      "homogeneous_stream_prices_per_opcode_NOT_a_floor": cl["class_floor_cycles_measured"]}
            per_perm = cyc_wave_cell / steps
            hv["instruction_classes_per_permutation"] = {k: round(v, 1) for k, v in cl["per_class"].items()}
            hv["isa_count_vs_counter"] = {"valu_insts_per_cell_from_isa": cl["valu_insts_per_cell_from_isa"], "SQ_INSTS_VALU_per_wave_cell": insts_h / waves_h,
                                          "isa_over_counter": round(cl["valu_insts_per_cell_from_isa"] / (insts_h / waves_h), 4),
                                          "note": "the ISA count walks the kernel's loop nest at cellSize 2048 (17 lines, 34 absorb steps); the staging blocks outside the absorb loop "
                                                  "(0.5 % of the stream) are priced approximately"}
            if "SQ_INSTS_VALU_INT64" in hc:      # an independent check of the class mix: the hardware's own count of 64-bit integer instructions
                isa64 = sum(r["per_permutation"] for r in cl["opcodes"] if r["opcode"] in ("v_mad_u64_u32", "v_lshl_add_u64", "v_lshlrev_b64", "v_mad_i64_i32"))
                hv["int64_share_check"] = {"SQ_INSTS_VALU_INT64_over_SQ_INSTS_VALU": round(hc["SQ_INSTS_VALU_INT64"]["per_launch_avg"] / insts_h, 4),
                                           "isa_share_of_v_mad_u64_u32_plus_64bit_adds": round(isa64 / n_isa, 4),
                                           "note": "SQ_INSTS_VALU_INT64 (pass hsq2) counts the 64-bit multiply-accumulates and adds; the ISA-derived share of the same opcodes agrees"}
            hv["class_floor_cycles"] = {k: (round(x, 1) if x else None) for k, x in floors.items()}
            hv["frac_of_class_floor"] = {k: (round(x / per_perm, 4) if x else None) for k, x in floors.items()}
            hv["classes_source"] = "profiles/%s_valu_classes_hash_cells.json (tools/valu_roof.py)" % R
            ub3 = None
            ub_path = os.path.join(P, "%s_ubench_classes_pmc.json" % R)
            if os.path.exists(ub_path):
                ub3 = json.load(open(ub_path)).get("by_occupancy", {}).get("v_mad_u64_u32", {})
            mad_rate = mad["measured_saturated_cycles"] or 4.0
            lane_mul_per_cycle = mad["per_permutation"] * steps * n_cells / cyc_h
            peak_per_cycle = 1024 * 64 / mad_rate
            hv["int_mul"] = {"v_mad_u64_u32_per_permutation": round(mad["per_permutation"], 1),
                             "lane_multiplies_per_cycle_chip": round(lane_mul_per_cycle, 1), "ubench_peak_per_cycle_chip": round(peak_per_cycle, 1),
                             "saturated_cycles_per_v_mad_u64_u32": mad_rate, "frac_of_ubench_peak": round(lane_mul_per_cycle / peak_per_cycle, 4),
                             "v_mad_u64_u32_cycles_at_this_kernels_occupancy": (ub3 or {}).get("waves_per_simd_3"),
                             "frac_of_ubench_rate_at_3_waves": round(lane_mul_per_cycle / (1024 * 64 / ub3["waves_per_simd_3"]), 4) if ub3 and ub3.get("waves_per_simd_3") else None,
                             "share_of_issue_cycles": round(mad["per_permutation"] * mad_rate / per_perm, 4)}
        h["valu_issue"] = hv
    json.dump(h, open(os.path.join(P, "%s_hash_cells_traffic.json" % R), "w"), indent=1)
    print(json.dumps(h))
for r in rows:
    print(r["Name"][:48], r["Calls"], "avg_ms=%.3f" % (float(r["AverageNs"]) * 1e-6))
