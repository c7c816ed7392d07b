"""GPU suite, round 5: residency that survives a shared device, the verified + bounded exchange, the cache that accepts what it
finds, pipelined transient builds, the opt-in multi-GPU default.  Everything against the oracle-only fixtures."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hexroot(r):
    return r.tobytes()[::-1].hex()


def child(job, **env):
    """tests/residency_child.py in a fresh process: its own environment, an empty allocation ledger."""
    clean = {k: v for k, v in os.environ.items() if not k.startswith("CODEX_P2_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "residency_child.py"), json.dumps(job)], capture_output=True, text=True, timeout=900,
                       env=dict(clean, CP2_TRACE="1", **{k: str(v) for k, v in env.items()}))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), r.stderr


# ---- what dataset_tree_mode plans with (csrc/proof_input.cpp), restated so that the caps below sit inside its windows -----------------
def layer_sum(n):
    s, m, bottom = 0, n, True
    while True:
        s += m
        if m == 1 and not bottom:
            return s
        m, bottom = (m + 1) // 2, False


def plan_bytes(c, n_local, stage_mb=2048):
    cpb = c["blockSize"] // c["cellSize"]
    nblocks = c["nCells"] // cpb
    per_slot = (nblocks * (layer_sum(cpb) - 1) + layer_sum(nblocks)) * 32
    compact = layer_sum(nblocks) * 32
    data = n_local * c["nCells"] * c["cellSize"]
    staging, headroom = 2 * min(data, stage_mb << 20), min(data, 1 << 30)
    batch = max(1, min(n_local, ((stage_mb << 20) // 2) // per_slot))
    need1 = per_slot * n_local + staging + headroom
    need2 = compact * n_local + 2 * batch * per_slot + staging + headroom
    return need1 / 0.9, need2 / 0.9           # free bytes from which "every node" / "compact" is chosen


def test_memory_cap_reaches_all_three_residency_modes(golden):
    """CODEX_P2_MEM_LIMIT_MB (a cap on what the process may hold on the device, honoured by the allocations themselves) walks the automatic
    choice through its three outcomes on config 5's scale-down (32 768 slots x 2^12 cells: 8 GiB of nodes, 0.25 GiB compact) without
    288 GB of anything: every node under 16 GiB, compact under 11 GiB, roots only under 7 GiB -- same roots, same input.json each time."""
    g5 = golden("config5.json")["scaled"]
    free1, free2 = plan_bytes(g5["config"], 32768)
    assert free2 < 11 * 2**30 < free1 < 16 * 2**30 and 7 * 2**30 < free2
    slots = [0, 16384]
    for cap_mb, want_mode in ((16384, 1), (11264, 2), (7168, 0)):
        res, err = child({"what": "single", "config": g5["config"], "entropy": g5["entropy"], "slots": slots}, CODEX_P2_MEM_LIMIT_MB=cap_mb)
        assert res["modes"] == [want_mode], (cap_mb, res["modes"], err[-1500:])
        assert res["slot_roots_sha256"] == g5["slot_roots_sha256"] and res["dataset_root_hex"] == g5["dataset_root_hex"]
        for s in slots:
            assert res["inputs"][str(s)] == {k: g5["inputs"][str(s)][k] for k in ("json_sha256", "json_bytes")}, (cap_mb, s)
        assert "did not fit after all" not in err                     # the estimate held: no step-down was needed


def test_an_automatic_choice_that_does_not_fit_after_all_steps_down(golden):
    """The figure the choice is made from is a snapshot -- another tenant can take the memory before the allocation.  Simulated with the
    test-only CODEX_P2_TEST_OPTIMISTIC=1 (the choice starts at "every node" without looking) under a cap that holds the compact build
    only: the first attempt fails with an allocation error, everything is freed, the scratch pool trimmed, and the build is retried
    one mode down -- said so in the trace, same roots.  A mode the caller NAMED is never changed: the same cap with
    CODEX_P2_KEEP_TREES=1 is an allocation error."""
    g5 = golden("config5.json")["scaled"]
    job = {"what": "single", "config": g5["config"], "entropy": g5["entropy"], "slots": [32767]}
    res, err = child(job, CODEX_P2_MEM_LIMIT_MB=10240, CODEX_P2_TEST_OPTIMISTIC=1)
    assert res["modes"] == [2], (res, err[-1500:])
    assert "keeping every node did not fit after all" in err and "retrying with the compact layers" in err
    assert res["slot_roots_sha256"] == g5["slot_roots_sha256"] and res["dataset_root_hex"] == g5["dataset_root_hex"]
    assert res["inputs"]["32767"]["json_sha256"] == g5["inputs"]["32767"]["json_sha256"]
    res, err = child(dict(job, what="streamed"), CODEX_P2_MEM_LIMIT_MB=10240, CODEX_P2_TEST_OPTIMISTIC=1)      # the streamed build has the same chain
    assert res["modes"] == [2] and "streamed build: keeping every node did not fit after all" in err, err[-1500:]
    assert res["dataset_root_hex"] == g5["dataset_root_hex"] and res["inputs"]["32767"]["json_sha256"] == g5["inputs"]["32767"]["json_sha256"]
    clean = {k: v for k, v in os.environ.items() if not k.startswith("CODEX_P2_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "residency_child.py"), json.dumps(job)], capture_output=True, text=True, timeout=600,
                       env=dict(clean, CODEX_P2_MEM_LIMIT_MB="10240", CODEX_P2_KEEP_TREES="1"))
    assert r.returncode != 0 and "status -4" in r.stderr, r.stderr[-1500:]                  # CP2_ERR_ALLOC, no silent change of a named mode


def test_four_contexts_on_one_device_share_what_it_has_free(golden):
    """cp2_multi over [0, 0, 0, 0]: every shard's automatic choice plans with a QUARTER of what the device had free before the shards
    started (round 4: each saw all of it, all chose "every node", the build ended in CP2_ERR_ALLOC).  Config 4's shape (4096 slots x
    2^12 cells) with 64 MiB staging chunks so that the window between compact and resident is reachable with 32 GiB of data: under a
    6 GiB cap a quarter (1536 MiB) holds the compact build (1360) but not the resident one (1564) -> compact on every shard; under
    8 GiB every shard keeps every node.  Roots, dataset root and input.json = the oracle-only fixture either way."""
    g4 = golden("fullsize.json")["config4"]
    free1, free2 = plan_bytes(g4["config"], 1024, stage_mb=64)
    assert free2 < (6144 << 20) / 4 < free1 < (8192 << 20) / 4
    slots = [1234, 4095]
    for cap_mb, want in ((6144, [2, 2, 2, 2]), (8192, [1, 1, 1, 1])):
        res, err = child({"what": "multi", "devices": [0, 0, 0, 0], "config": g4["config"], "entropy": g4["entropy"], "slots": slots},
                         CODEX_P2_MEM_LIMIT_MB=cap_mb, CODEX_P2_STAGE_MB=64)
        assert res["modes"] == want and res["units_per_slot"] == 1 and len(res["shards"]) == 4, (cap_mb, res, err[-1500:])
        assert res["slot_roots_sha256"] == g4["slot_roots_sha256"] and res["dataset_root_hex"] == g4["dataset_root_hex"]
        for s in slots:
            assert res["inputs"][str(s)] == {k: g4["inputs"][str(s)][k] for k in ("json_sha256", "json_bytes")}
        assert "did not fit after all" not in err


@pytest.mark.parametrize("mode", [2, 0])
def test_pipelined_transient_builds_over_many_batches(golden, mode):
    """The compact / roots-only builds pipeline their batches (two node buffers used alternately, the copy-out of batch k under the
    hashing of batch k + 1, nothing synchronised in between).  16 MiB staging chunks cut config 4's 4096 slots into 128 batches of 32:
    plain and streamed, every slot root, the dataset root and two complete input.json texts against the oracle-only fixture."""
    g4 = golden("fullsize.json")["config4"]
    for what in ("single", "streamed"):
        res, err = child({"what": what, "config": g4["config"], "entropy": g4["entropy"], "slots": [1234, 4095]}, CODEX_P2_KEEP_TREES=mode, CODEX_P2_STAGE_MB=16)
        assert res["modes"] == [mode], (what, res)
        assert res["slot_roots_sha256"] == g4["slot_roots_sha256"] and res["dataset_root_hex"] == g4["dataset_root_hex"], what
        for s in ("1234", "4095"):
            assert res["inputs"][s] == {k: g4["inputs"][s][k] for k in ("json_sha256", "json_bytes")}, (what, s)
        assert "4096 of 4096 slots enqueued" in err


# ---- the exchange: verified, bounded, and still usable after a failure ------------------------------------------------------------------
def test_exchange_is_verified_whatever_carried_it(pkg, golden):
    """After the exchange every device must find its own slot roots at its own rows of the list it received, and all devices must
    compute the same dataset root.  The test-only CODEX_P2_TEST_EXCHANGE_FAULT=corrupt flips one byte of the last context's gathered
    copy on the device paths -- what a wrong rank-to-device mapping or a misplaced block looks like: the build fails with
    "exchange verification failed", by slots and by units, for peer copies and for the RCCL path (a one-rank communicator asked for
    by name); it never returns a dataset with a wrong root.  And the handle, its contexts and its cached communicator still work."""
    c = golden("proof_inputs.json")["inputs"]["params_default"]["config"]
    want = golden("proof_inputs.json")["inputs"]["params_default"]
    cfg = pkg.make_config(**c)
    for devices, policy, split in (([0, 0, 0], pkg.GATHER_COPY, 1), ([0, 0], pkg.GATHER_COPY, 2), ([0], pkg.GATHER_RCCL, 1)):
        m = pkg.Multi(devices)
        m.set_policy(policy, 1)
        m.set_split(split)
        ok = m.dataset(cfg)
        root = ok.root().copy()
        assert ok.proof_input(want["slotIndex"], want["entropy"]).json() == golden("input_params_default.json")
        ok.free()
        os.environ["CODEX_P2_TEST_EXCHANGE_FAULT"] = "corrupt"
        try:
            with pytest.raises(pkg.CodexP2Error) as e:
                m.dataset(cfg)
            assert "exchange verification failed" in str(e.value) and ("copy" in str(e.value) or "rccl" in str(e.value)), str(e.value)
        finally:
            del os.environ["CODEX_P2_TEST_EXCHANGE_FAULT"]
        again = m.dataset(cfg)                                        # the handle (and, for RCCL, its cached communicator) after the failure
        assert (again.root() == root).all() and ("rccl" in m.gather_mode() if policy == pkg.GATHER_RCCL else "copy" in m.gather_mode())
        assert again.proof_input(want["slotIndex"], want["entropy"]).json() == golden("input_params_default.json")
        again.free()
        m.close()


def test_rccl_by_name_on_one_device_after_a_failed_build(pkg, golden, tmp_path):
    """CP2_GATHER_RCCL on [0] (a one-rank communicator: the RCCL code path itself) before and after a build that FAILS (a missing slot
    file): the error names the file, and the next build on the same handle uses the cached communicator and gives the fixture's text."""
    m0 = golden("proof_inputs.json")["inputs"]["params_default"]
    cfg = pkg.make_config(**m0["config"])
    m = pkg.Multi([0])
    m.set_policy(pkg.GATHER_RCCL, 0)
    a = m.dataset(cfg)
    assert "rccl" in m.gather_mode()
    a.free()
    with pytest.raises(pkg.CodexP2Error) as e:
        m.dataset(pkg.make_config(**dict(m0["config"], file=str(tmp_path / "nothing_here"))))
    assert "nothing_here0.dat" in str(e.value)
    b = m.dataset(cfg)
    assert "rccl" in m.gather_mode() and b.proof_input(m0["slotIndex"], m0["entropy"]).json() == golden("input_params_default.json")
    b.free()
    m.close()


def test_several_devices_are_opt_in(pkg):
    """cp2_multi_init with no device named takes ONE device (the first visible gfx950) until the exchange between two real devices has a
    committed record; CODEX_P2_GPUS=all / a count / a list opts in."""
    import torch
    saved = os.environ.pop("CODEX_P2_GPUS", None)
    try:
        m = pkg.Multi()
        assert m.count == 1 and m.devices() == [0]
        m.close()
        os.environ["CODEX_P2_GPUS"] = "all"
        m = pkg.Multi()
        assert m.count == torch.cuda.device_count()
        m.close()
        os.environ["CODEX_P2_GPUS"] = "0,0"
        m = pkg.Multi()
        assert m.devices() == [0, 0]
        m.close()
    finally:
        os.environ.pop("CODEX_P2_GPUS", None)
        if saved is not None:
            os.environ["CODEX_P2_GPUS"] = saved


# ---- the cache accepts what it finds; roots-only datasets notice changed data -----------------------------------------------------------
def test_cached_build_accepts_any_valid_representation_and_never_evicts_a_richer_one(pkg, golden, tmp_path, capfd):
    """Which representation a cached build WANTS depends on what the device has free that day.  (1) A run that would keep every node
    finds the compact layers in the cache: it takes them (the dataset is compact, nothing is hashed), instead of rebuilding every
    slot and overwriting the file.  (2) A run that keeps its trees compact finds a full tree cache at the path: the tree cache stays,
    the compact layers go to "<path>.kept" beside it, and the next compact run loads from there.  (3) A named mode is literal: a
    roots-only run does not take a compact cache."""
    m0 = golden("proof_inputs.json")["inputs"]["testmain_small"]
    cfg, want = pkg.make_config(**m0["config"]), golden("input_testmain_small.json")
    os.environ["CP2_TRACE"] = "1"
    try:
        ctx = pkg.Context(0)
        cache = str(tmp_path / "a.cp2")
        ctx.set_keep_trees(2)
        ctx.dataset(cfg, cache=cache).free()                         # the compact layers are what is cached
        assert open(cache, "rb").read(8) == b"CP2KEPT1"
        ctx.set_keep_trees(-1)                                        # automatic: an empty 288 GB device -> "every node" is what it wants
        capfd.readouterr()
        ds = ctx.dataset(cfg, cache=cache)
        err = capfd.readouterr().err
        assert ds.tree_mode == 2 and "taken as it is" in err and "generate + hash" not in err, err
        assert ds.proof_input(m0["slotIndex"], m0["entropy"]).json() == want and open(cache, "rb").read(8) == b"CP2KEPT1"
        ds.free()
        # (2)
        cache2 = str(tmp_path / "b.cp2")
        ctx.set_keep_trees(1)
        ctx.dataset(cfg, cache=cache2).free()
        tree_bytes = open(cache2, "rb").read()
        assert tree_bytes[:8] == b"CP2TREE3"
        ctx.set_keep_trees(2)
        ctx.dataset(cfg, cache=cache2).free()                        # built compact; the tree cache is not overwritten
        assert open(cache2, "rb").read() == tree_bytes and open(cache2 + ".kept", "rb").read(8) == b"CP2KEPT1"
        capfd.readouterr()
        ds = ctx.dataset(cfg, cache=cache2)
        err = capfd.readouterr().err
        assert ds.tree_mode == 2 and "compact layers loaded from the cache" in err and "generate + hash" not in err, err
        assert ds.proof_input(m0["slotIndex"], m0["entropy"]).json() == want
        ds.free()
        ctx.set_keep_trees(1)
        ds = ctx.dataset(cfg, cache=cache2)                          # and the tree cache still loads
        assert ds.tree_mode == 1 and ds.proof_input(m0["slotIndex"], m0["entropy"]).json() == want
        ds.free()
        # (3)
        ctx.set_keep_trees(0)
        capfd.readouterr()
        ds = ctx.dataset(cfg, cache=cache)                           # holds compact layers; roots only was NAMED: built, and written in place of them
        err = capfd.readouterr().err
        assert ds.tree_mode == 0 and "built and written to the cache" in err, err
        ds.free()
        ctx.close()
    finally:
        del os.environ["CP2_TRACE"]


def test_roots_only_dataset_notices_changed_slot_data(pkg, oracle, golden, tmp_path):
    """Roots only: slotRoot, slotProof and dataSetRoot come from the stored roots, indices / paths / cells from the slot's tree rebuilt
    on demand.  When the slot file changed in between, the rebuilt root no longer equals the stored one: CP2_ERR_IO ("... does not hash
    to its stored root"), never an input.json over mixed data (compact mode has had the same check per block)."""
    C, _ = oracle
    c = golden("proof_inputs.json")["inputs"]["testmain_small"]["config"]
    base = str(tmp_path / "slotdata")
    for k in range(c["nSlots"]):
        C.gen_fake_cells(C.slot_seed(c["seed"], k), 0, c["nCells"], c["cellSize"]).tofile("%s%d.dat" % (base, k))
    ctx = pkg.Context(0)
    ctx.set_keep_trees(0)
    ds = ctx.dataset(pkg.make_config(**dict(c, file=base)))
    assert ds.tree_mode == 0 and ds.proof_input(3, 1234567).json() == golden("input_testmain_small.json")
    raw = bytearray(open(base + "3.dat", "rb").read())
    raw[777] ^= 0x40
    open(base + "3.dat", "wb").write(bytes(raw))
    with pytest.raises(pkg.CodexP2Error) as e:
        ds.proof_input(3, 1234567)
    assert e.value.status == -5 and "slot 3 does not hash to its stored root" in str(e.value)
    assert ds.proof_input(2, 1234567).json()                          # the other slots still prove
    ds.free()
    ctx.close()


def test_a_rank_without_slots_issues_the_same_collective(pkg, oracle, tmp_path):
    """World 3, two slots: rank 2 holds nothing.  Round 4 sent it down the host path while its peers used the device gather (two
    different collectives for one step); now every rank takes the device path, the empty one contributes an empty block
    and computes the root from the gathered roots.  Fresh rank processes sharing GPU 0, gloo."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from rank_helpers import run_ranks
    C, _ = oracle
    c = dict(maxDepth=16, maxLog2NSlots=2, cellSize=2048, blockSize=65536, nSlots=2, nCells=256, nSamples=7, seed=31337)
    res = run_ranks(3, c, 55555, tmp_path)
    roots = np.stack([C.fake_slot_root(C.slot_seed(c["seed"], s_), c["cellSize"], c["blockSize"], c["nCells"], 4) for s_ in range(2)])
    want = hexroot(C.merkle_root(roots))
    assert sorted((r["rank"], r["count"]) for r in res) == [(0, 1), (1, 1), (2, 0)]
    for r in res:
        assert r["dataset_root_hex"] == want and r["all_roots_sha256"] == hashlib.sha256(roots.tobytes()).hexdigest(), r["rank"]


def test_a_communicator_creation_that_never_returns_costs_a_timeout_not_the_process(pkg, golden):
    """ncclCommInitAll runs on a helper thread under CODEX_P2_EXCHANGE_TIMEOUT_S.  With the test-only fault "hang_init" (the thread never
    returns) a build that asked for RCCL by name fails after the timeout with an error that says so; RCCL is then not used again in
    the process (its thread is still somewhere inside the library), the automatic mode goes on working, and RCCL by name says why not.
    A fresh process: the abandonment is process-wide."""
    code = r"""
import json, os, sys, time
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package()
cfg = pkg.make_config(maxDepth=8, maxLog2NSlots=2, cellSize=128, blockSize=512, nSlots=3, nCells=16, nSamples=3, seed=7)
m = pkg.Multi([0])
m.set_policy(pkg.GATHER_RCCL, 0)
out = {}
t0 = time.time()
try:
    m.dataset(cfg)
    out["first"] = "built"
except pkg.CodexP2Error as e:
    out["first"] = str(e)
out["seconds"] = time.time() - t0
del os.environ["CODEX_P2_TEST_EXCHANGE_FAULT"]
try:
    m.dataset(cfg)
    out["second"] = "built"
except pkg.CodexP2Error as e:
    out["second"] = str(e)
m.set_policy(pkg.GATHER_AUTO, 0)
d = m.dataset(cfg)
out["auto"] = m.gather_mode()
out["root"] = d.root().tobytes().hex()
print(json.dumps(out))
os._exit(0)          # the abandoned thread sleeps for ever: leave without joining it
""" % ROOT
    clean = {k: v for k, v in os.environ.items() if not k.startswith("CODEX_P2_")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(clean, CODEX_P2_TEST_EXCHANGE_FAULT="hang_init", CODEX_P2_EXCHANGE_TIMEOUT_S="2"))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "ncclCommInitAll did not return within 2 s" in out["first"] and 1.5 < out["seconds"] < 30, out
    assert "RCCL gather requested but unavailable" in out["second"] and "abandoned" in out["second"], out
    assert out["auto"].startswith("none") and len(out["root"]) == 64


def test_slot_files_in_the_page_cache_are_uploaded_straight_from_a_mapping(pkg, oracle, golden, tmp_path, capfd):
    """Mapped ingestion (cp2_set_ingest_mapped, opt-in): chunks of a slot file whose pages are all in the page cache are registered
    with the runtime and uploaded from the mapping itself -- no pread into the pinned ring -- and everything else goes through the
    ring; same trees either way.  Files just written are in the cache: every chunk takes the mapped path (the trace says how many did),
    chunks of 100-byte cells whose boundaries fall inside pages do not (a window is whole pages of its own); switched off, the ring carries everything; a file evicted from
    the cache (fsync + POSIX_FADV_DONTNEED) is read through the ring; a short file is zero-filled (slot.nim:61-66) by the ring for the
    chunk that reaches past its end."""
    C, _ = oracle
    os.environ["CP2_TRACE"] = "1"
    forced = os.environ.pop("CP2_INGEST_MAPPED", None)      # (the suite is sometimes run with mapping forced on: this test sets the mode itself)
    try:
        ctx = pkg.Context(0)
        for cs, nc, n_slots in ((2048, 1 << 16, 3), (100, 1 << 12, 2)):
            base = str(tmp_path / ("f%d_" % cs))
            for k in range(n_slots):
                C.gen_fake_cells(C.slot_seed(777, k), 0, nc, cs).tofile("%s%d.dat" % (base, k))
            bs = cs * 32
            want = np.stack([C.fake_slot_root(C.slot_seed(777, k), cs, bs, nc, 8) for k in range(n_slots)])
            cfg = pkg.make_config(maxDepth=24, maxLog2NSlots=2, cellSize=cs, blockSize=bs, nSlots=n_slots, nCells=nc, nSamples=4, file=base)
            if cs == 100:
                ctx.set_ingest(4, 3, 65536)               # 64 KiB chunks of 100-byte cells: no chunk boundary on a page boundary
            for mapped in (1, 0):
                ctx.set_ingest_mapped(mapped)
                capfd.readouterr()
                ds = ctx.dataset(cfg)
                err = capfd.readouterr().err
                assert np.array_equal(ds.local_roots(), want), (cs, mapped)
                line = [l for l in err.splitlines() if "from the page cache by mapping" in l][-1]
                n_mapped = int(line.split("slot files:")[1].split("chunk")[0])
                if cs == 100:
                    assert n_mapped == 0, line            # windows are whole pages of their own: these chunks all go through the ring
                else:
                    assert (n_mapped > 0) == (mapped == 1) and ("%d chunk(s) from the page cache by mapping, 0 through" % n_mapped in line) == (mapped == 1), (cs, mapped, line)
                pi = ds.proof_input(1, 4242).json()
                ds.free()
            ctx.set_ingest_mapped(-1)
            ctx.set_ingest(0, 0, 0)
        # evicted from the page cache: nothing is resident, the ring reads it; same root
        ctx.set_ingest_mapped(1)
        base = str(tmp_path / "f2048_")
        for k in range(3):
            fd = os.open("%s%d.dat" % (base, k), os.O_RDONLY)
            os.fsync(fd)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            os.close(fd)
        cfg = pkg.make_config(maxDepth=24, maxLog2NSlots=2, cellSize=2048, blockSize=65536, nSlots=3, nCells=1 << 16, nSamples=4, file=base)
        want = np.stack([C.fake_slot_root(C.slot_seed(777, k), 2048, 65536, 1 << 16, 8) for k in range(3)])
        capfd.readouterr()
        ds = ctx.dataset(cfg)
        err = capfd.readouterr().err
        assert np.array_equal(ds.local_roots(), want)
        import subprocess
        fs = subprocess.run(["stat", "-f", "-c", "%T", str(tmp_path)], capture_output=True, text=True).stdout.strip()
        if fs not in ("tmpfs", "ramfs"):                                                      # (a memory file system has nothing to evict to)
            assert "slot files: 0 chunk(s) from the page cache by mapping" in err, err        # nothing was resident: the ring read it all
        ds.free()
        # a short file: cells past its end read as zeros, whichever path carried the chunks before
        cells = C.gen_fake_cells(C.slot_seed(777, 0), 0, 1 << 16, 2048)
        cells[50000:] = 0
        cells[:50000].tofile(base + "0.dat")
        ds = ctx.dataset(cfg)
        assert np.array_equal(ds.local_roots()[0], ctx.slot_trees_host(cells, 1, 2048, 65536, 1 << 16).roots()[0])
        assert np.array_equal(ds.local_roots()[1:], want[1:])
        ds.free()
        # left alone (-1, no CP2_INGEST_MAPPED in the environment) the ring carries everything: mapping is opt-in
        ctx.set_ingest_mapped(-1)
        capfd.readouterr()
        ds = ctx.dataset(cfg)
        assert "slot files: 0 chunk(s) from the page cache by mapping" in capfd.readouterr().err
        ds.free()
        ctx.close()
    finally:
        del os.environ["CP2_TRACE"]
        if forced is not None:
            os.environ["CP2_INGEST_MAPPED"] = forced


def test_host_arrays_the_caller_pinned_go_through_without_the_ring(pkg, oracle, capfd):
    """cp2_permute_batch / cp2_compress_pairs on host arrays (the reference's calling convention: Nim seqs): pageable arrays are copied
    through a pinned ring by host threads; arrays the caller pinned (hipHostMalloc -- here torch pinned tensors) are read and written
    in place by the copy engines.  Same results, checked against the C oracle on a sample; the trace says which way it went.  Sizes
    above the 2^20 items of the single-launch path, ragged."""
    import ctypes
    import torch
    C, _ = oracle
    os.environ["CP2_TRACE"] = "1"
    try:
        ctx = pkg.Context(0)
        n = (5 << 20) + 12345                         # six chunks on the pinned path: its four device buffers are used again
        rng = np.random.default_rng(55)
        x = rng.integers(0, 256, size=(n, 96), dtype=np.uint8)
        x[:, 31] &= 0x1f; x[:, 63] &= 0x1f; x[:, 95] &= 0x1f
        capfd.readouterr()
        y = ctx.permute_batch(x)
        err = capfd.readouterr().err
        assert "stream_map" in err and "straight from" not in err
        xin = torch.from_numpy(x).pin_memory()
        xout = torch.zeros_like(xin).pin_memory()
        ctx._ck(ctx.L.cp2_permute_batch(ctx.h, ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(xout.data_ptr()), n), "cp2_permute_batch")
        err = capfd.readouterr().err
        assert "straight from / to the caller's pinned arrays" in err, err
        assert np.array_equal(xout.numpy(), y)
        idx = np.concatenate([np.arange(64), rng.integers(0, n, size=512), np.arange(n - 64, n)])
        assert np.array_equal(y[idx], C.permute_batch(x[idx], threads=4))
        # one pinned, one not: the ring
        out2 = np.zeros_like(x)
        ctx._ck(ctx.L.cp2_permute_batch(ctx.h, ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(out2.ctypes.data), n), "cp2_permute_batch")
        err = capfd.readouterr().err
        assert "straight from" not in err and np.array_equal(out2, y)
        ctx.close()
    finally:
        os.environ.pop("CP2_TRACE", None)


@pytest.mark.parametrize("n_slots,n_cells,stage_mb", [(100, 1 << 12, 0), (61, 1 << 12, 0), (37, 1 << 13, 0), (64, 1 << 12, 96), (3, 1 << 17, 0)])
def test_streamed_build_whose_single_chunk_the_ramp_cuts_into_several_turns(pkg, oracle, n_slots, n_cells, stage_mb):
    """The streamed build hashes its groups alternately on two streams, each with its own staging buffer, and the ramp-down at the end
    of the fake builder cuts even a dataset that fits ONE staging chunk into several turns once it holds more than one residency of the
    hash kernel (768 x 256 cells): the second turn needs the second buffer.  (Round 5's soak found the case: a null staging pointer
    is a GPU memory fault.)  Sizes between one residency and one chunk, an odd slot count, a small staging chunk (many turns), slots
    larger than a residency: streamed text = object path on every slot, roots = C oracle on three."""
    C, P = oracle
    if stage_mb:
        os.environ["CODEX_P2_STAGE_MB"] = str(stage_mb)
    try:
        ctx = pkg.Context(0)
        c = dict(maxDepth=20, maxLog2NSlots=7, cellSize=2048, blockSize=65536, nSlots=n_slots, nCells=n_cells, nSamples=7, seed=4242 + n_slots)
        cfg = pkg.make_config(**c)
        ref = ctx.dataset(cfg)
        for k in (0, n_slots // 2, n_slots - 1):
            assert np.array_equal(ref.local_roots()[k], C.fake_slot_root(C.slot_seed(c["seed"], k), 2048, 65536, n_cells, 8)), k
        for keep in (-1, 2, 0):
            ctx.set_keep_trees(keep)
            sd = ctx.dataset_streamed(cfg, 987654321, threads=4)
            sd.set_roots(None)
            sd.export_streamed(None, threads=4)
            assert np.array_equal(sd.local_roots(), ref.local_roots()), keep
            for s_ in range(n_slots):
                assert sd.streamed_json(s_) == ref.proof_input(s_, 987654321).json(), (keep, s_)
            sd.free()
        ctx.set_keep_trees(-1)
        ref.free()
        ctx.close()
    finally:
        os.environ.pop("CODEX_P2_STAGE_MB", None)
