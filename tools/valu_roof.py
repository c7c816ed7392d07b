#!/usr/bin/env python3
"""The VALU-issue roof of k_permute_batch, class by class (round 4: replaces the "4.0 cycles = full rate" label).

CDNA4 has SIMD-32 vector units: a wave64 VALU instruction issues over 2 cycles (/opt/skills/guides/MI355X_MICROARCH.md,
"Wave scheduling" and the constants row `v_fma_f32 (wave64) 2 cyc (SIMD-32)`); multiplies and the other "half-rate" integer
forms take 4.  So the floor of an instruction stream is  sum over opcodes of  count x saturated cycles of that opcode,  not
"4 x instructions".  This tool
  1. rebuilds the DYNAMIC instruction count per permutation, opcode by opcode, from the generated ISA and the loop structure of
     the kernel (the loop over the two halves x2 holds the loop over three unmasked external rounds x3, one masked external
     round and -- first half only -- the 28 pairs of internal rounds; same decomposition as tools/cycle_model.py, whose total
     agrees with SQ_INSTS_VALU to 0.04 %),
  2. prices every opcode twice: at the guide's NOMINAL rate (2 cycles for the plain 32-bit add / logic / shift-right / move
     forms, 4 for everything else) -- the floor -- and at the cost the opcode shows in a saturated HOMOGENEOUS stream of itself,
     measured on this chip with the same counters the kernel is measured with (tools/ubench_classes.hip under rocprofv3 --pmc:
     GRBM_GUI_ACTIVE / 8 x 1024 SIMDs / SQ_INSTS_VALU, committed as profiles/<round>_ubench_classes_pmc.json).  The second sum is
     NOT a floor: a mixed stream issues faster than its opcodes do alone (the kernel beats it), it is there as the measured
     per-opcode price list and for SURVEY.md 8(d)'s "int-mul rate against the measured peak",
  3. writes profiles/<round>_valu_classes.json: counts, classes, both floors in cycles per wave-permutation per SIMD.
tools/profile_summarize.py sets them against the kernel's own PMC pass.   Usage: valu_roof.py [round]"""
import collections, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "codex-storage-proofs-circuits_amd", "csrc", "kernels.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", src, "-o", "/tmp/k_roof.s"],
                      stderr=subprocess.DEVNULL)
lines = open("/tmp/k_roof.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4cp2k\d+k_permute_batch", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))


def tally(a, b):
    c = collections.Counter()
    for l in body[a:b]:
        m = re.match(r"^\s+(v_[a-z0-9_]+)", l)
        if m:
            c[re.sub(r"_(e32|e64|dpp|sdwa)$", "", m.group(1))] += 1
    return c


big = sorted([t for t in loops if t[1] - t[0] > 500], key=lambda t: t[1] - t[0])
assert len(big) == 3, "expected the internal-pair loop, the external-round loop and the loop over the halves"
(i0, i1), (e0, e1), (o0, o1) = big
C_int, C_ext, C_out, C_all = tally(i0, i1), tally(e0, e1), tally(o0, o1), tally(0, len(body))
int_inside = o0 <= i0 and i1 <= o1
rest_outer = C_out - C_ext - (C_int if int_inside else collections.Counter())
prologue = C_all - C_out - (collections.Counter() if int_inside else C_int)
total = collections.Counter()
for c, k in ((prologue, 1), (rest_outer, 2), (C_ext, 6), (C_int, 28)):
    for key, v in c.items():
        total[key] += k * v

# the guide's nominal rates: plain 32-bit VOP2 add / sub / logic / shift-right / move = 2 cycles on a SIMD-32; every other form 4
SIMPLE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32",
          "v_add_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32"}
ub_path = os.path.join(ROOT, "profiles", "%s_ubench_classes_pmc.json" % R)
ub_all = json.load(open(ub_path)) if os.path.exists(ub_path) else {}
ub = ub_all.get("cycles_per_instruction", {})


def price(total, kernel, method, unit, out_name, per="per_permutation", extra=None):
    n_total = sum(total.values())
    rows, nominal, measured, missing = [], 0.0, 0.0, []
    for op, cnt in total.most_common():
        nom = 2.0 if op in SIMPLE else 4.0
        meas = ub.get(op)
        if meas is None:
            missing.append(op)
            meas_used = ub.get("v_add_u32" if op in SIMPLE else "v_mul_lo_u32", nom)    # class representative
        else:
            meas_used = meas
        nominal += cnt * nom
        measured += cnt * meas_used
        rows.append({"opcode": op, per: cnt, "share": round(cnt / n_total, 4), "class": "simple" if op in SIMPLE else ("multiply-accumulate 64" if op == "v_mad_u64_u32" else "half-rate"),
                     "nominal_cycles": nom, "measured_saturated_cycles": meas, "priced_at": meas_used})
    cls = collections.Counter()
    for r in rows:
        cls[r["class"]] += r[per]
    out = {"round": R, "kernel": kernel, "method": method,
           "valu_insts_%s_from_isa" % per: n_total, "per_class": dict(cls), "opcodes": rows,
           "nominal_rates": "MI355X_MICROARCH.md: SIMD-32, `v_fma_f32 (wave64) 2 cyc`; plain 32-bit add/logic/shift-right/move forms 2 cycles, every other form 4",
           "measured_rates_source": os.path.relpath(ub_path, ROOT) if ub else None,
           "opcodes_priced_by_class_representative": missing,
           "class_floor_cycles_nominal": nominal, "class_floor_cycles_measured": measured if ub else None,
           "measured_sum_is_not_a_floor": "homogeneous-stream prices; a mixed stream issues faster than the sum of them",
           "unit": unit}
    out.update(extra or {})
    json.dump(out, open(os.path.join(ROOT, "profiles", out_name), "w"), indent=1)
    print("%s: VALU %s (ISA): %s   classes: %s" % (kernel, per.replace("_", " "), n_total, dict(cls)))
    for r in rows[:14]:
        print("  %-18s %9.1f  %5.1f %%  %-22s nominal %.0f  measured %s" % (r["opcode"], r[per], 100 * r["share"], r["class"], r["nominal_cycles"], r["measured_saturated_cycles"]))
    print("class floor: nominal %.0f cycles, measured-rate %s" % (nominal, ("%.0f" % measured) if ub else "n/a (no ubench file yet)"))
    return out


price(total, "cp2k::k_permute_batch",
      "dynamic count per permutation = prologue x1 + rest of the half loop x2 + unmasked external round x6 + internal round pair x28, from the gfx950 ISA of csrc/kernels.hip",
      "shader cycles per wave-permutation (64 permutations) per SIMD", "%s_valu_classes.json" % R)

# ---- k_hash_cells<256> at cellSize 2048 (configs[2]): 17 lines staged, 34 absorb steps per cell --------------------------------
# The kernel's loop nest in the ISA (same compiler output as the library): per line a staging block, then the absorb loop; one
# absorb step = ring read + chunk_pair (one of two alignments, alternating) + 2 x to_mont + the permutation, whose two halves
# share one copy of the code (three unmasked external rounds in a loop + the masked one), the first half followed by the loop
# over the 28 pairs of internal rounds.  Regions are found from the labels and backward branches, not from line numbers.
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4cp2k\d+k_hash_cellsILi256E", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
back = []          # (target line, branch line) of every backward branch
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        back.append((labels[m.group(1)], i))
nv = lambda a, b: sum(1 for l in body[a:b] if re.match(r"^\s+v_", l))
big = sorted([t for t in back if nv(*t) > 500], key=lambda t: nv(*t))
# the two smallest: the loop over the pairs of internal rounds and the loop over the unmasked external rounds (disjoint; the
# external one comes first in the code); the larger ones go back to the header of a half / of the absorb loop
inner = [t for t in big if nv(*t) > 1000 and not any(u != t and t[0] <= u[0] and u[1] <= t[1] for u in big)]
assert len(inner) == 2, inner
(e0, e1), (i0, i1) = sorted(inner)
assert e1 < i0, "expected the external-round loop before the internal-pair loop"
half0 = max(t[0] for t in back if t[0] < e0 and t[1] > i1)      # header of a half: target of the branch that follows the internal rounds
absorb0 = max(t[0] for t in back if t[0] < half0 and half0 - 8 <= t[1] <= e0)   # absorb-loop header: target of the branch at the half header
half_end = max(t[1] for t in back if t[0] == half0 and e1 < t[1] < i0)          # "second half done?": the branch between the masked round and the internal rounds
int_end = max(t[1] for t in back if t[0] == half0 and t[1] > i1)
# the two chunk_pair alignments inside the absorb prelude: consecutive forward branches over ~20 VALU each; one runs per step
pre = (absorb0, half0)
variants = []
for i in range(pre[0], pre[1]):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", body[i])
    if m and labels[m.group(1)] > i and labels[m.group(1)] < pre[1] and 10 <= nv(i, labels[m.group(1)]) <= 40:
        variants.append((i, labels[m.group(1)]))
assert len(variants) == 2, variants
C = lambda a, b: tally(a, b)
absorb_pre = C(*pre)
for a, b in variants:                                   # one of the two runs per absorb step: each at half weight
    half_w = C(a, b)
    for k, v in half_w.items():
        absorb_pre[k] -= v / 2.0
half = collections.Counter()
for c, k in ((C(half0, e0), 1), (C(e0, e1 + 1), 3), (C(e1 + 1, half_end + 1), 1)):
    for key, v in c.items():
        half[key] += k * v
ints = collections.Counter()
for c, k in ((C(half_end + 1, i0), 1), (C(i0, i1 + 1), 28), (C(i1 + 1, int_end + 1), 1)):
    for key, v in c.items():
        ints[key] += k * v
perm = collections.Counter()
for c, k in ((absorb_pre, 1), (half, 2), (ints, 1)):
    for key, v in c.items():
        perm[key] += k * v
# everything outside the absorb loop: prologue (reduction-table fill), the per-line staging block (x17 lines; its copy loops are
# unrolled by 8: 4 trips of 32 loads), epilogue (conversion + store).  Well under 1 % of the stream; priced approximately.
outside_pre, outside_post = C(0, absorb0), C(int_end + 1, len(body))
N_LINES, N_STEPS = 17, 34
cell = collections.Counter()
for key, v in perm.items():
    cell[key] += N_STEPS * v
for key, v in outside_pre.items():
    cell[key] += N_LINES * v * 2.0          # staging: the common path's unrolled copy loop runs 4 trips; other paths are skipped: ~2 x static
for key, v in outside_post.items():
    cell[key] += v
per_perm_equiv = collections.Counter({k: v / N_STEPS for k, v in cell.items()})
price(per_perm_equiv, "cp2k::k_hash_cells<256>",
      "cellSize 2048: dynamic count per CELL = 34 absorb steps x (ring read + chunk_pair + 2 to_mont + permutation: 2 x (3 x external-round loop body + masked round) "
      "+ 28 x internal-pair loop body) + 17 lines x staging + prologue / epilogue, from the gfx950 ISA of csrc/kernels.hip; listed per permutation (= per cell / 34)",
      "shader cycles per wave-permutation (64 cells x 1 absorb step) per SIMD", "%s_valu_classes_hash_cells.json" % R,
      extra={"valu_insts_per_cell_from_isa": sum(cell.values()), "permutation_part_per_step": sum(perm.values()),
             "outside_the_absorb_loop_per_cell_approx": sum(cell.values()) - N_STEPS * sum(perm.values()),
             "regions_isa_lines": {"absorb_prelude": pre, "half": (half0, half_end), "external_loop": (e0, e1), "internal_loop": (i0, i1), "internal_end": int_end,
                                   "chunk_pair_variants": variants}})
