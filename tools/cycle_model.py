#!/usr/bin/env python3
"""Issue-cycle model of k_permute_batch from its ISA and the measured per-class instruction costs
(tools/ubench_valu.hip, profiles/r01_ubench_valu.txt): predicts cycles per permutation-wave and perm/s, to be read
next to the measured launch time.  The kernel is VALU-issue bound, so the prediction should land within a few %."""
import collections, re, subprocess, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "codex-storage-proofs-circuits_amd", "csrc", "kernels.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", src, "-o", "/tmp/k_model.s"],
                      stderr=subprocess.DEVNULL)
lines = open("/tmp/k_model.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4cp2k\d+k_permute_batch", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32",
        "v_fma_f32", "v_fmac_f32", "v_add_f32", "v_mul_f32"}
def cost(op):
    base = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
    if base in ("v_mad_u64_u32", "v_mad_i64_i32"): return 4.56
    if base in ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32"): return 4.54
    if base.startswith("v_cndmask"): return 22.9
    if base in FAST: return 2.3
    return 4.3
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def tally(a, b):
    n, cyc, c = 0, 0.0, collections.Counter()
    for l in body[a:b]:
        m = re.match(r"^\s+(v_[a-z0-9_]+)", l)
        if m:
            n += 1; cyc += cost(m.group(1)); c[re.sub(r"_e32$|_e64$", "", m.group(1))] += 1
    return n, cyc, c
loops.sort(key=lambda t: t[1] - t[0])
# innermost big loops: the external-round body (iterated 8x) and the internal-round pair (28x)
big = [t for t in loops if t[1] - t[0] > 500]
inner = sorted(big, key=lambda t: t[1] - t[0])[:2]
ext = max(inner, key=lambda t: t[1] - t[0]); intl = min(inner, key=lambda t: t[1] - t[0])
n_ext, c_ext, k_ext = tally(*ext); n_int, c_int, k_int = tally(*intl)
n_all, c_all, _ = tally(0, len(body))
outer = [t for t in big if t not in inner]
n_rest = n_all - n_ext - n_int; c_rest = c_all - c_ext - c_int
insts = 8 * n_ext + 28 * n_int + n_rest
cycles = 8 * c_ext + 28 * c_int + c_rest
print("external-round body : %5d VALU, %7.0f cycles  (x8)" % (n_ext, c_ext))
print("internal-round pair : %5d VALU, %7.0f cycles  (x28)" % (n_int, c_int))
print("prologue/epilogue   : %5d VALU, %7.0f cycles" % (n_rest, c_rest))
print("per permutation-wave: %d VALU instructions, %.0f issue cycles (mads: %.0f%%)" %
      (insts, cycles, 100 * 4.56 * (8 * k_ext["v_mad_u64_u32"] + 28 * k_int["v_mad_u64_u32"]) / cycles))
for ghz in (2.33, 2.40):
    print("at %.2f GHz, 1024 SIMDs: %.3e permutations/s  (2^24 states in %.2f ms)" %
          (ghz, ghz * 1e9 / cycles * 64 * 1024, (1 << 24) / (ghz * 1e9 / cycles * 64 * 1024) * 1e3))
