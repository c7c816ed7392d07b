#!/usr/bin/env python3
"""Issue-cycle model of k_permute_batch from its ISA (round 2).

Inside this kernel every VALU instruction costs one issue slot of about 4.1 cycles whatever its kind, v_mov_b32 about 0.9
(tools/pad_probe.py, profiles/r02_marginal_cost_probe.txt), so time = slots x instructions.  The instruction count per
permutation is rebuilt from the loop structure of the generated code: the loop over the two halves (x2) holds the loop over
three unmasked external rounds (x3), one masked external round, and -- first half only -- the 28 pairs of internal rounds.
Read next to SQ_INSTS_VALU (profiles/r02_permute_batch_traffic.json)."""
import collections, re, subprocess, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "codex-storage-proofs-circuits_amd", "csrc", "kernels.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", src, "-o", "/tmp/k_model.s"],
                      stderr=subprocess.DEVNULL)
lines = open("/tmp/k_model.s").read().split("\n")
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
            c["mov" if m.group(1).startswith("v_mov_b32") else ("mad" if m.group(1).startswith("v_mad_u64_u32") else "other")] += 1
    return c


big = sorted([t for t in loops if t[1] - t[0] > 500], key=lambda t: t[1] - t[0])
assert len(big) == 3, "expected the internal-pair loop, the external-round loop and the loop over the halves"
(i0, i1), (e0, e1), (o0, o1) = big                      # internal pair < external round < halves (encloses both)
C_int, C_ext, C_out, C_all = tally(i0, i1), tally(e0, e1), tally(o0, o1), tally(0, len(body))
int_inside = o0 <= i0 and i1 <= o1                       # the compiler may place the internal rounds outside the loop over the halves
rest_outer = C_out - C_ext - (C_int if int_inside else collections.Counter())   # the masked external round (+ entry / exit of the internal rounds)
prologue = C_all - C_out - (collections.Counter() if int_inside else C_int)
total = collections.Counter()
for c, k in ((prologue, 1), (rest_outer, 2), (C_ext, 6), (C_int, 28)):
    for key, v in c.items():
        total[key] += k * v
n = sum(total.values())
cycles = 4.4 * total["mad"] + 4.05 * total["other"] + 0.9 * total["mov"]
print("external round (unmasked): %5d VALU (%d v_mad_u64_u32, %d moves)   x6" % (sum(C_ext.values()), C_ext["mad"], C_ext["mov"]))
print("internal round pair      : %5d VALU (%d v_mad_u64_u32, %d moves)   x28" % (sum(C_int.values()), C_int["mad"], C_int["mov"]))
print("rest of the half loop    : %5d VALU   x2 (the masked external round%s)" % (sum(rest_outer.values()), " + entering / leaving the internal rounds" if int_inside else ""))
print("prologue / epilogue      : %5d VALU (conversions from / to canonical form, first linear layer%s)" % (sum(prologue.values()), "" if int_inside else ", entering / leaving the internal rounds"))
print("per permutation          : about %d VALU instructions (%d v_mad_u64_u32, %d moves): %.0f issue cycles per 64 permutations per SIMD" %
      (n, total["mad"], total["mov"], cycles))
print("(the clock figures of profiles/ come from GRBM_GUI_ACTIVE in PMC passes, which run a few % below un-profiled launches;")
print(" measured launches are 5-8 % faster than this model at the same nominal clock: it prices every instruction at its MARGINAL cost)")
for ghz in (2.18, 2.30, 2.35):
    print("at %.2f GHz, 1024 SIMDs: %.3e permutations/s  (2^24 states in %.2f ms)" % (ghz, ghz * 1e9 / cycles * 64 * 1024, (1 << 24) / (ghz * 1e9 / cycles * 64 * 1024) * 1e3))
