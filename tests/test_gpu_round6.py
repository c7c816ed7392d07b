"""GPU suite, round 6: streamed proof inputs from REAL slot files through the multi-file ingestion pipe (ring, O_DIRECT, mapped;
every node / compact / roots only), every input.json against the oracle; the A/B knobs that ship; the launch shape decided once per
context; the versioned boundary.  Reference: slot.nim:57-68, dataset.nim:34, gen_input/bn254.nim:56-64."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle_helpers import expected_proof_input_fast

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = lambda s: hashlib.sha256(s.encode()).hexdigest()   # noqa: E731


def oracle_texts(C, P, c, entropy, threads=8):
    """input.json of EVERY slot of a fake-data configuration from the oracle (C oracle for the hashes, Python restatement for the rest)."""
    roots = np.stack([C.fake_slot_root(C.slot_seed(c["seed"], s), c["cellSize"], c["blockSize"], c["nCells"], threads) for s in range(c["nSlots"])])
    return [P.export_json(expected_proof_input_fast(C, P, c, s, entropy, threads=threads, slot_roots=roots)) for s in range(c["nSlots"])]


def write_slot_files(C, c, base):
    for k in range(c["nSlots"]):
        C.gen_fake_cells(C.slot_seed(c["seed"], k), 0, c["nCells"], c["cellSize"]).tofile("%s%d.dat" % (base, k))


def file_config(c, base):
    return dict({k: v for k, v in c.items() if k != "seed"}, file=base)


# geometry A: slots far smaller than a ring buffer -- every turn of the pipe holds 5 whole slot files; 37 slots = 8 turns, the sampling
#             passes follow the turns and are cut into groups of 4 (10+ passes: more than once around the landing ring)
# geometry B: slots larger than a ring buffer -- 2.67 turns per slot file, turns end on the slot boundaries; 7 slots, one pass per slot
GEOM_A = dict(maxDepth=14, maxLog2NSlots=6, cellSize=256, blockSize=2048, nSlots=37, nCells=512, nSamples=9, seed=60601)
GEOM_B = dict(maxDepth=14, maxLog2NSlots=3, cellSize=256, blockSize=2048, nSlots=7, nCells=512, nSamples=9, seed=60602)
CHUNK_A, CHUNK_B = 5 * 512 * 256, 192 * 256


@pytest.fixture(scope="module")
def file_cases(oracle, tmp_path_factory):
    C, P = oracle
    d = tmp_path_factory.mktemp("r6slots")
    cases = {}
    for name, c, chunk in (("A", GEOM_A, CHUNK_A), ("B", GEOM_B, CHUNK_B)):
        base = str(d / ("slot%s_" % name))
        write_slot_files(C, c, base)
        cases[name] = (c, base, chunk, oracle_texts(C, P, c, 424243))
    return cases


@pytest.mark.parametrize("how", ["ring", "direct", "mapped"])
@pytest.mark.parametrize("name", ["A", "A12", "B"])
def test_streamed_from_slot_files_every_input_json_vs_oracle(pkg, file_cases, name, how):
    """cp2_dataset_build_streamed + export from slot files, EVERY slot's input.json byte for byte against the oracle: turns of many
    files (A: passes of 4 slots, less than a ring turn's 5; A12: passes of 12 slots = 2.4 ring turns each, three of them and a
    ramp-down of 7 + 6 at the end) and slots of several turns (B: a pass per slot of 2.67 turns), through the pinned ring, with
    O_DIRECT, and uploaded from mappings of the files."""
    group = {"A": 4, "A12": 12, "B": 1}[name]
    c, base, chunk, want = file_cases[name[0]]
    ctx = pkg.Context(0)
    try:
        ctx.set_ingest(3, 2, chunk)                    # 3 fill threads, the smallest ring (2 pinned + 3 device buffers): every buffer is reused many times
        ctx.set_ingest_direct(1 if how == "direct" else 0)
        ctx.set_ingest_mapped(1 if how == "mapped" else 0)
        ds = ctx.dataset_streamed(pkg.make_config(**file_config(c, base)), 424243, threads=3, group_slots=group)
        assert ds.tree_mode == 1
        ds.export_streamed(None, threads=3)
        got = [ds.streamed_json(s) for s in range(c["nSlots"])]
        bad = [s for s in range(c["nSlots"]) if got[s] != want[s]]
        assert not bad, "input.json differs from the oracle for slots %s" % bad
        ds.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("keep", [2, 0])
def test_streamed_from_slot_files_transient_batches_pipeline(pkg, oracle, file_cases, keep):
    """Compact / roots-only streamed builds from slot files: several batches through ONE ingestion pipe and two node buffers (round 6;
    a pipe per batch, drained, before).  A fresh process with a 1 MiB staging size: 37 slots go in three batches.  Every input.json
    equals the oracle's, and the kept layers answer for a later entropy."""
    c, base, chunk, want = file_cases["A"]
    C, P = oracle
    job = {"config": file_config(c, base), "entropy": 424243, "group": 4, "keep": keep, "threads": 3, "ingest": {"threads": 3, "ring": 2, "chunk_bytes": chunk, "direct": 0, "mapped": 0},
           "later": {"slot": 29, "entropy": 31337}}
    clean = {k: v for k, v in os.environ.items() if not k.startswith("CODEX_P2_") and not k.startswith("CP2_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stream_child.py"), json.dumps(job)], capture_output=True, text=True, timeout=900,
                       env=dict(clean, CODEX_P2_STAGE_MB="1", CP2_TRACE="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["mode"] == keep
    assert res["json_sha256"] == [sha(t) for t in want]
    assert "37 of 37 slots enqueued" in r.stderr, r.stderr[-2000:]
    assert res["later_json_sha256"] == sha(P.export_json(expected_proof_input_fast(C, P, c, 29, 31337, threads=8)))   # (sampled cells read from the files)


def test_short_and_missing_slot_files(pkg, oracle, tmp_path):
    """A slot file that ends early reads as zeros past its end (slot.nim:61-66) -- also when it is one of many files of a turn; a
    missing file is CP2_ERR_IO naming the file, from the fill threads of the multi-file pipe too."""
    C, P = oracle
    c = dict(maxDepth=10, maxLog2NSlots=3, cellSize=128, blockSize=1024, nSlots=6, nCells=64, nSamples=5, seed=778)
    base = str(tmp_path / "s")
    write_slot_files(C, c, base)
    data = open(base + "4.dat", "rb").read()
    open(base + "4.dat", "wb").write(data[:128 * 40 + 64])          # 40.5 cells
    cf = file_config(c, base)
    ctx = pkg.Context(0)
    try:
        ds = ctx.dataset_streamed(pkg.make_config(**cf), 99, threads=2, group_slots=2)   # default chunk: all six files in one turn
        ds.export_streamed(None, threads=2)
        for slot in (3, 4, 5):
            assert ds.streamed_json(slot) == P.export_json(P.generate_proof_input(dict(cf), slot, 99))
        ds.free()
        os.remove(base + "2.dat")
        with pytest.raises(pkg.CodexP2Error) as e:
            ctx.dataset_streamed(pkg.make_config(**cf), 99, threads=2)
        assert "cannot open" in str(e.value) and "s2.dat" in str(e.value)
        with pytest.raises(pkg.CodexP2Error):
            ctx.dataset(pkg.make_config(**cf))
    finally:
        ctx.close()


# ---- the knobs that ship must at least be right -----------------------------------------------------------
KNOB_CFG = dict(maxDepth=16, maxLog2NSlots=9, cellSize=2048, blockSize=65536, nSlots=300, nCells=1024, nSamples=5, seed=60603)


@pytest.fixture(scope="module")
def knob_case(oracle, tmp_path_factory):
    C, P = oracle
    d = tmp_path_factory.mktemp("r6knobs")
    base = str(d / "k")
    write_slot_files(C, KNOB_CFG, base)                       # 300 files of 2 MiB
    return base, [sha(t) for t in oracle_texts(C, P, KNOB_CFG, 777001, threads=16)]


def run_child(job, **env):
    clean = {k: v for k, v in os.environ.items() if not k.startswith("CODEX_P2_") and not k.startswith("CP2_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stream_child.py"), json.dumps(job)], capture_output=True, text=True, timeout=900,
                       env=dict(clean, CP2_TRACE="1", **{k: str(v) for k, v in env.items()}))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), r.stderr


@pytest.mark.parametrize("env", [{}, {"CP2_STREAM_SERIAL": 1}, {"CP2_STREAM_RAMP": 0}, {"CP2_STREAM_SERIAL": 1, "CP2_STREAM_RAMP": 0},
                                 {"CODEX_P2_TEST_LDS_LIMIT": 65536}], ids=["default", "serial", "no_ramp", "serial_no_ramp", "no_room"])
@pytest.mark.parametrize("source", ["fake", "file"])
def test_streamed_build_under_the_ab_knobs_and_without_room(knob_case, env, source):
    """The streamed build of 300 slots of 2 MiB (the fake builder's ramp-down cuts its single chunk into several turns) under every
    A/B knob the library reads, and with the hash launches refused their room (the launch shape is decided ONCE per context, by
    cp2_init, from the device's LDS per workgroup -- capped here by the test hook; no launch is ever retried): every input.json
    equals the oracle's, from generated data and from slot files; the trace says which launch shape the context got."""
    base, want = knob_case
    cfg = dict(KNOB_CFG) if source == "fake" else file_config(KNOB_CFG, base)
    res, err = run_child({"config": cfg, "entropy": 777001, "group": 0, "threads": 4}, **env)
    assert res["json_sha256"] == want
    if "CODEX_P2_TEST_LDS_LIMIT" in env:
        assert "hold every workgroup slot: no room" in err, err[-2000:]
    else:
        assert "leave room: two workgroups per CU" in err, err[-2000:]


def test_slot_file_rings_shrink_when_the_device_cannot_hold_them(knob_case):
    """The ingestion pipe's rings (3 pinned + 4 device buffers of one turn each: 2.4 GiB of device memory for this 600 MiB dataset)
    under a 1 GiB cap on what the process may hold on the device: the turn is halved until the rings fit -- smaller launches, the same
    input.json for every slot -- instead of the build failing with CP2_ERR_ALLOC in every residency mode."""
    base, want = knob_case
    res, err = run_child({"config": file_config(KNOB_CFG, base), "entropy": 777001, "group": 0, "threads": 4}, CODEX_P2_MEM_LIMIT_MB=1024)
    assert res["json_sha256"] == want
    assert "half the turn" in err, err[-3000:]


# ---- a collective that does not complete ---------------------------------------------------------------------
def test_an_exchange_that_times_out_is_final_and_nothing_waits_on_its_streams():
    """Test-only fault "hang_collective": the exchange of slot roots (device-to-device copies between two contexts) is followed, on
    every participating stream, by work that outlasts CODEX_P2_EXCHANGE_TIMEOUT_S by five seconds.  The build fails after the
    time-out with an error that says so -- and that is final: no second attempt through host memory is enqueued behind the stuck
    work (it would wait without a bound), the contexts take no further work, and neither the failed build's clean-up nor closing the
    handle waits for the streams (their buffers are dropped from the books).  The automatic mode decides with the same rule
    (csrc/exchange_policy.hpp, walked on the CPU)."""
    code = r"""
import json, os, sys, time
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package()
cfg = pkg.make_config(maxDepth=8, maxLog2NSlots=3, cellSize=128, blockSize=512, nSlots=6, nCells=16, nSamples=3, seed=7)
m = pkg.Multi([0, 0])
m.set_policy(pkg.GATHER_COPY, 1)                       # one cell per device is enough for a shard: two shards, a real exchange
out = {}
t0 = time.time()
try:
    m.dataset(cfg)
    out["first"] = "built"
except pkg.CodexP2Error as e:
    out["first"] = str(e)
out["build_s"] = time.time() - t0
t0 = time.time()
try:
    m.dataset(cfg)
    out["second"] = "built"
except pkg.CodexP2Error as e:
    out["second"] = str(e)
out["second_s"] = time.time() - t0
t0 = time.time()
m.close()
out["close_s"] = time.time() - t0
print(json.dumps(out), flush=True)
os._exit(0)
""" % ROOT
    clean = {k: v for k, v in os.environ.items() if not k.startswith("CODEX_P2_")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(clean, CODEX_P2_TEST_EXCHANGE_FAULT="hang_collective", CODEX_P2_EXCHANGE_TIMEOUT_S="2"))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "did not complete within 2 s" in out["first"] and "take no further work" in out["first"], out
    assert 1.5 < out["build_s"] < 5.5, out                # the time-out, not the seven seconds the streams stay busy
    assert "takes no further work" in out["second"] and out["second_s"] < 1.0, out
    assert out["close_s"] < 1.0, out
