#!/usr/bin/env python3
"""Counts VALU/LDS/SALU instructions per loop of a kernel in the gfx950 ISA of csrc/kernels.hip.
The kernels are VALU-issue bound, so instructions per permutation is the figure of merit offline."""
import collections, re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "codex-storage-proofs-circuits_amd", "csrc", "kernels.hip")
kernel = sys.argv[1] if len(sys.argv) > 1 else "k_permute_batch"
extra = sys.argv[2:]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", src, "-o", "/tmp/k.s"] + extra,
                      stderr=subprocess.DEVNULL)
lines = open("/tmp/k.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4cp2k\d+%s" % kernel, l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
# loops: label .. backward branch to the label
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i, m.group(1)))
def count(a, b):
    c = collections.Counter()
    for l in body[a:b]:
        m = re.match(r"^\s+([vsd][a-z0-9_]+|global_\w+|buffer_\w+|flat_\w+)", l)
        if m:
            op = m.group(1)
            c["VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM"] += 1
            c[op] += 1
    return c
tot = count(0, len(body))
print("kernel %s: total VALU %d (mad %d) SALU %d LDS %d VMEM %d" % (kernel, tot["VALU"], tot["v_mad_u64_u32"], tot["SALU"], tot["LDS"], tot["VMEM"]))
for a, b, name in loops:
    c = count(a, b)
    top = ", ".join("%s %d" % (k, v) for k, v in c.most_common(14) if k not in ("VALU", "SALU", "LDS", "VMEM"))
    print("  loop %s [%d lines]: VALU %d (mad %d) SALU %d LDS %d VMEM %d | %s" % (name, b - a, c["VALU"], c["v_mad_u64_u32"], c["SALU"], c["LDS"], c["VMEM"], top))
meta = [l for l in lines[end:end + 400] if "vgpr_count" in l or "sgpr_count" in l or "spill" in l]
print("  " + " ".join(m.strip() for m in meta[:4]))
