#!/usr/bin/env python3
"""gpurun_out/<dir>/ (rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -- tools/ubench_classes) ->
profiles/<round>_ubench_classes_pmc.json: saturated issue cycles per wave-instruction per SIMD of every opcode, in the very units
the kernel's own figure is in (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs / SQ_INSTS_VALU).  Usage: ... <round> <dir under gpurun_out>"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, D = sys.argv[1], sys.argv[2]
fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", D, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
disp = collections.defaultdict(dict)
for r in rows:
    if "ub_v_" in r["Kernel_Name"]:
        d = disp[r["Dispatch_Id"]]
        d["op"] = r["Kernel_Name"].split("ub_")[1].split("(")[0]
        d["grid"] = int(r["Grid_Size"])
        d[r["Counter_Name"]] = float(r["Counter_Value"])
per = collections.defaultdict(list)
for d in disp.values():
    waves = d["SQ_WAVES"]
    w_per_simd = round(waves / 1024)
    cyc = d["GRBM_GUI_ACTIVE"] / 8 * 1024 / d["SQ_INSTS_VALU"]
    per[(d["op"], w_per_simd)].append(cyc)
table = collections.defaultdict(dict)
for (op, w), v in per.items():
    table[op]["waves_per_simd_%d" % w] = round(min(v), 4)          # the second launch of each pair (the first also loads code)
best = {op: min(t.values()) for op, t in table.items()}
out = {"round": R, "tool": "tools/ubench_classes.hip under rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES (one pass)",
       "unit": "shader cycles per wave-instruction per SIMD = GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs / SQ_INSTS_VALU; loops of 32 independent instructions of one kind",
       "by_occupancy": table, "cycles_per_instruction": best,
       "note": "cycles_per_instruction = the lowest (saturated) figure over the occupancies run (3, 4, 5, 8 waves per SIMD); k_permute_batch runs at 4 "
               "(102 VGPRs), k_hash_cells at 3 (LDS ring): by_occupancy holds the price list at each"}
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_ubench_classes_pmc.json" % R), "w"), indent=1, sort_keys=True)
for op, t in sorted(table.items(), key=lambda kv: -best[kv[0]]):
    print("%-16s %s" % (op, t))
