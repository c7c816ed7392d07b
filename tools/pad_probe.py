"""Marginal cost of one more instruction of each kind INSIDE k_permute_batch.

  python tools/pad_probe.py build     # here: one library per instruction kind under build/variants/pad_*.so (hipcc, parallel)
  python tools/pad_probe.py run       # on the GPU box: times each against the unpadded library, prints cycles per added instruction

CP2_PAD_N copies of the instruction are issued after each of the 17 columns of each of the 240 multiplications of a
permutation (8160 extra wave-instructions per permutation at N = 2)."""
import json, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "codex-storage-proofs-circuits_amd")
OUT = os.path.join(ROOT, "build", "variants")
N = 2
KINDS = {
    "and_b32": "v_and_b32 %0, %1, %0",
    "add_u32": "v_add_u32 %0, %1, %0",
    "mov_b32": "v_mov_b32 %0, %1",
    "lshrrev_b32": "v_lshrrev_b32 %0, 1, %0",
    "lshlrev_b32": "v_lshlrev_b32 %0, 1, %0",
    "lshrrev_b64": "v_lshrrev_b64 %2, 1, %2",
    "alignbit_b32": "v_alignbit_b32 %0, %1, %0, 29",
    "mul_lo_u32": "v_mul_lo_u32 %0, %1, %0",
    "mad_u64_u32": "v_mad_u64_u32 %2, vcc, %0, %1, %2",
    "add3_u32": "v_add3_u32 %0, %1, %0, %0",
    "lshl_add_u32": "v_lshl_add_u32 %0, %0, 3, %1",
    "bfe_u32": "v_bfe_u32 %0, %0, 3, 29",
    "lshl_or_b32": "v_lshl_or_b32 %0, %1, 3, %0",
    "and_or_b32": "v_and_or_b32 %0, %1, %0, %0",
    "mad_u32_u24": "v_mad_u32_u24 %0, %1, %0, %0",
    "lshl_add_u64": "v_lshl_add_u64 %2, %2, 0, %2",
}
# second set: pairs (CP2_PAD_N = 1, two instructions per site): dependent on each other vs independent
PAIRS = {
    "and_dep": "v_and_b32 %0, %1, %0\\n\\tv_and_b32 %0, %1, %0",
    "and_indep": "v_and_b32 %0, %1, %1\\n\\tv_and_b32 %3, %1, %1",
    "add_indep": "v_add_u32 %0, %1, %1\\n\\tv_add_u32 %3, %1, %1",
    "mov_dep": "v_mov_b32 %0, %1\\n\\tv_mov_b32 %1, %0",
    "mov_indep": "v_mov_b32 %0, %1\\n\\tv_mov_b32 %3, %1",
    "mul_lo_indep": "v_mul_lo_u32 %0, %1, %1\\n\\tv_mul_lo_u32 %3, %1, %1",
    "mad64_dep": "v_mad_u64_u32 %2, vcc, %0, %1, %2\\n\\tv_mad_u64_u32 %2, vcc, %0, %1, %2",
    "mad64_indep": "v_mad_u64_u32 %2, vcc, %0, %1, %2\\n\\tv_mad_u64_u32 %4, vcc, %0, %1, %4",
    "lshr64_indep": "v_lshrrev_b64 %2, 1, %2\\n\\tv_lshrrev_b64 %4, 1, %4",
    "and_then_mad64": "v_and_b32 %0, %1, %1\\n\\tv_mad_u64_u32 %2, vcc, %3, %1, %2",
    "mad24_indep": "v_mad_u32_u24 %0, %1, %1, %1\\n\\tv_mad_u32_u24 %3, %1, %1, %1",
    "nop_pair": "s_nop 0\\n\\ts_nop 0",
}


def build_one(name):
    so = os.path.join(OUT, "pad_%s.so" % name)
    text, n = (KINDS[name], N) if name in KINDS else (PAIRS[name], 1)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-pthread", "-shared",
           '-DCP2_PAD_ASM="%s"' % text, "-DCP2_PAD_N=%d" % n, "-o", so] + \
          [os.path.join(PKG, "csrc", f) for f in ("kernels.hip", "codex_p2_abi.cpp", "slot_trees.cpp", "proof_input.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, r.stderr[-300:]


if sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    which = PAIRS if (len(sys.argv) > 2 and sys.argv[2] == "pairs") else KINDS
    with ThreadPoolExecutor(4) as ex:
        for name, rc, err in ex.map(build_one, which):
            print(name, "ok" if rc == 0 else "FAILED " + err, flush=True)
else:
    def sample(lib):
        env = dict(os.environ)
        if lib:
            env["CODEX_P2_LIB"] = lib
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_one.py")], env=env, capture_output=True, text=True)
        return json.loads(out.stdout.strip().split("\n")[-1])
    base = min(sample(None)["perm_ms_min"] for _ in range(2))
    ghz = float(os.environ.get("PAD_GHZ", "2.31"))
    extra = 17 * 240 * N                       # added wave-instructions per permutation-wave
    print("unpadded: %.3f ms per 2^24 states" % base)
    print("%-14s %9s %9s %s" % ("instruction", "ms", "slowdown", "cycles per added wave-instruction per SIMD (at %.2f GHz)" % ghz))
    for name in list(KINDS) + list(PAIRS):
        if len(sys.argv) > 2 and sys.argv[2] == "pairs" and name not in PAIRS:
            continue
        so = os.path.join(OUT, "pad_%s.so" % name)
        if not os.path.exists(so):
            continue
        try:
            ms = sample(so)["perm_ms_min"]
        except Exception as e:
            print("%-14s failed: %r" % (name, e))
            continue
        # one SIMD runs 2^24 / 64 / 1024 = 256 permutation-waves per launch
        cyc = (ms - base) * 1e-3 * ghz * 1e9 / 256 / extra
        print("%-14s %9.3f %8.2f%% %6.2f" % (name, ms, 100 * (ms / base - 1), cyc), flush=True)
