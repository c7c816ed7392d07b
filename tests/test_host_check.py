"""CPU suite, part 4: the DEVICE arithmetic source (csrc/fr_gfx950.hpp, poseidon2_dev.hpp) compiled for the
host with CP2_HOST_CHECK: 128-bit shadow accumulator (traps on 64-bit column overflow), asserted limb/value
bounds of the lazy reduction, AddressSanitizer + UBSan, every result compared with the C oracle."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_arithmetic_on_host_with_sanitizers(tmp_path):
    exe = str(tmp_path / "host_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-o", exe, os.path.join(ROOT, "tests", "host_check", "host_check.cpp"),
                           os.path.join(ROOT, "oracle", "p2_oracle.c"), "-lpthread"])
    r = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ", 0 mismatches, no bound violations" in r.stdout


def test_fake_builder_turn_plan_with_sanitizers(tmp_path):
    """How the fake-data builder cuts a batch into staging turns (csrc/fake_turns.hpp, the header trees_build_fake itself uses):
    walked over ~2e5 shapes under AddressSanitizer + UBSan -- no empty turn, none beyond the staging chunk or the batch, ramp turns
    whole slots, and never a turn on the second staging buffer unless the plan allocates it (round 5: a single chunk cut into
    several turns by the ramp-down found that buffer missing, a GPU memory fault).  No GPU."""
    exe = str(tmp_path / "turn_plan_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + os.path.join(ROOT, "codex-storage-proofs-circuits_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "host_check", "turn_plan_check.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "turn plan ok" in r.stdout and " 0 shapes where" not in r.stdout, r.stdout


def test_slot_file_builder_plan_and_layer_scheduler_with_sanitizers(tmp_path):
    """The slot-file builder's arithmetic (csrc/ingest_turns.hpp: turns over a batch of slot files, the fill threads' byte
    ranges, which file bytes a turn's buffer holds) and the layer scheduler's decision (layer_take), the very header
    trees_build_files / LayerScheduler use: walked over 1e5 shapes under AddressSanitizer + UBSan -- every turn has a buffer of
    sufficient size, every buffer byte is written once from the file byte slot.nim:57-68 reads, every slot's layers are built
    exactly once after its last cell; the same layer walk over the fake-data builder's plans.  No GPU."""
    exe = str(tmp_path / "ingest_plan_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + os.path.join(ROOT, "codex-storage-proofs-circuits_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "host_check", "ingest_plan_check.cpp")])
    r = subprocess.run([exe, "100000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ingest plan ok: 100000 shapes" in r.stdout, r.stdout


def test_fill_pipeline_against_real_files_asan_ubsan_and_tsan(tmp_path):
    """The host side of the ingestion pipe (csrc/fill_pipeline.hpp, the class IngestPipe itself uses: grains from a shared counter,
    fills posted two turns deep, workers running on into the next turn while the building thread joins this one) against REAL slot
    files in a scratch directory -- short files, a missing file (named, lowest slot first), slots cut into units, O_DIRECT requested,
    the host-array source -- every turn compared with the reference's own read of a cell (slot.nim:57-68).  Built with
    AddressSanitizer + UBSan (ring buffers of exactly a turn's size) and again with ThreadSanitizer.  No GPU, no HIP."""
    src = os.path.join(ROOT, "tests", "host_check", "fill_pipeline_check.cpp")
    inc = "-I" + os.path.join(ROOT, "codex-storage-proofs-circuits_amd", "csrc")
    for name, flags, shapes in (("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "150"), ("tsan", ["-fsanitize=thread"], "40")):
        exe = str(tmp_path / ("fill_pipeline_" + name))
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-Wall", *flags, inc, "-o", exe, src])
        scratch = tmp_path / ("files_" + name)
        scratch.mkdir()
        r = subprocess.run([exe, str(scratch), shapes], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (name, r.stdout[-2000:], r.stderr[-4000:])
        assert "fill pipeline ok: %s shapes" % shapes in r.stdout and "ThreadSanitizer" not in r.stderr, (name, r.stdout, r.stderr[-2000:])


def _build_text_check(pkg, tmp_path, sanitize=("-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"), name="host_text_check"):
    libdir = os.path.dirname(pkg.LIB_PATH)
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", *sanitize,
                           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
                           os.path.join(ROOT, "tests", "host_check", "host_text_check.cpp"),
                           "-L" + libdir, "-lcodex_p2", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib",
                           "-lpthread"])
    return exe


def _run(exe, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, *args], capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


def test_host_text_and_body_store_with_sanitizers(pkg, tmp_path):
    """The product's host-side hot loops -- the byte-exact JSON formatter (csrc/json_text.hpp; json/bn254.nim:57-74,
    json/shared.nim:17-25, types/bn254.nim:29-43) and the streamed build's body store with its private spill files
    (csrc/body_store.hpp) -- compiled from the product's own headers under AddressSanitizer + UBSan.  No GPU."""
    import random
    import sys
    sys.path.insert(0, ROOT)
    from oracle import poseidon2_ref as ref
    exe = _build_text_check(pkg, tmp_path)

    # 1. 256-bit integers to decimal: every edge of the base-10^19 chunking and of the reciprocal division, then random
    rnd = random.Random(20261004)
    vals = [0, 1, 9, 10, 99, 100, 2**32 - 1, 2**32, 2**64 - 1, 2**64, 2**128 - 1, 2**128, 2**192 - 1, 2**192, 2**256 - 1, ref.R_MOD - 1, ref.R_MOD]
    for k in (19, 38, 57, 76):
        vals += [10**k - 1, 10**k, 10**k + 1, 9 * 10**k, 10**k + 10**(k - 1)]
    vals += [11 * 10**76, 2**256 - 1 - 10**19, (2**64 - 1) * 10**19, (10**19 - 1) * (2**64) + (2**64 - 1)]
    vals += [rnd.getrandbits(rnd.choice((1, 8, 63, 64, 65, 127, 128, 191, 192, 250, 254, 256))) for _ in range(20000)]
    vals = [v for v in vals if v < 2**256]
    blob = tmp_path / "dec.bin"
    blob.write_bytes(b"".join(v.to_bytes(32, "little") for v in vals))
    got = _run(exe, "dec", str(blob)).decode().splitlines()
    assert len(got) == len(vals)
    for v, line in zip(vals, got):
        assert line == '"%d"' % v, (v, line)

    # 2. whole texts against the oracle's writer, on shapes that move every bound: no samples, one sample, depth 0 / 1 / 32,
    #    no slotProof entries, cells of 1, 31, 32, 62, 2048 bytes, all-ones data (the longest numbers)
    def text_case(md, ml, cs, ns, fill=None):
        felt = (lambda: ref.R_MOD - 1) if fill == "max" else (lambda: rnd.randrange(ref.R_MOD) if rnd.random() < 0.9 else rnd.randrange(1000))
        cell = (lambda: b"\xff" * cs) if fill == "max" else (lambda: bytes(rnd.getrandbits(8) for _ in range(cs)))
        p = {"dataSetRoot": felt(), "entropy": felt(), "nCells": rnd.choice((2, 512, 2**22, 2**40)), "nSlots": rnd.choice((1, 11, 32768, 2**33)),
             "slotIndex": rnd.choice((0, 3, 2**33 - 1)), "slotRoot": felt(), "slotProof": {"merklePath": [felt() for _ in range(ml)]},
             "proofInputs": [{"cellData": cell(), "merkleProof": {"merklePath": [felt() for _ in range(md)]}} for _ in range(ns)]}
        b = b"".join(x.to_bytes(8, "little") for x in (ml, md, cs, p["nCells"], p["nSlots"], p["slotIndex"], ns))
        b += b"".join(p[k].to_bytes(32, "little") for k in ("dataSetRoot", "entropy", "slotRoot"))
        b += b"".join(x.to_bytes(32, "little") for x in p["slotProof"]["merklePath"])
        b += b"".join(q["cellData"] for q in p["proofInputs"])
        b += b"".join(x.to_bytes(32, "little") for q in p["proofInputs"] for x in q["merkleProof"]["merklePath"])
        f = tmp_path / "json.bin"
        f.write_bytes(b)
        assert _run(exe, "json", str(f)).decode() == ref.export_json(p), (md, ml, cs, ns, fill)

    for (md, ml, cs, ns) in [(32, 8, 2048, 3), (0, 0, 1, 0), (1, 0, 31, 1), (5, 1, 32, 2), (16, 15, 62, 7), (32, 8, 93, 5), (12, 3, 64, 100)]:
        text_case(md, ml, cs, ns)
    text_case(32, 8, 2048, 4, fill="max")
    text_case(3, 2, 31, 2, fill="max")

    # 3. the body store: budget, concurrent spills, 0700 directory / 0600 files, a planted symlink, O_EXCL, clean-up
    spill = tmp_path / "spill"
    spill.mkdir()
    assert b"body store ok" in _run(exe, "store", str(spill))
    assert list(spill.iterdir()) == []


def test_body_store_workers_with_thread_sanitizer(pkg, tmp_path):
    """put() runs on the formatting workers: the same harness under ThreadSanitizer (four workers spilling at once)."""
    exe = _build_text_check(pkg, tmp_path, sanitize=("-fsanitize=thread",), name="host_text_check_tsan")
    spill = tmp_path / "spill"
    spill.mkdir()
    r = subprocess.run([exe, "store", str(spill)], capture_output=True, timeout=600)
    assert r.returncode == 0 and b"body store ok" in r.stdout and b"ThreadSanitizer" not in r.stderr, (r.returncode, r.stderr[-3000:])
