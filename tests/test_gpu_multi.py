"""GPU suite, round 4 (-m gpu): several GPUs of one node behind ONE handle of the C ABI (cp2_multi_*, include/codex_p2.h section e):
one process, one host thread + one context per device, contiguous slot ranges, ONE exchange of slot roots, the dataset tree on
every device, proof inputs routed to the owning device (reference/nim/proof_input/src/gen_input/bn254.nim:41-51,72;
workflow/prove.sh:26).

A one-GPU box cannot hold two distinct devices, so the two branches of the exchange are exercised like this:
  devices [0]        RCCL asked for by name: a communicator of one rank, in-place ncclAllGather, cp2_dataset_set_roots_dev
  devices [0, 0]     two (three) contexts on device 0: the host-gather branch ("a device holds more than one shard")
Everything is compared with oracle-only fixtures (tests/golden/config5.json, proof_inputs.json, bigslots.json) or with the
C oracle + Python restatement directly."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from oracle_helpers import expected_proof_input_fast
from rank_helpers import run_ranks

pytestmark = pytest.mark.gpu


def hexroot(a):
    return np.asarray(a, dtype=np.uint8).tobytes()[::-1].hex()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def tsha(text):
    return hashlib.sha256(text.encode()).hexdigest()


def _threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


@pytest.mark.parametrize("name,devices,gather", [("cheap", [0], "rccl"), ("cheap", [0, 0], "auto"), ("odd", [0, 0], "auto"),
                                                  ("odd", [0, 0, 0], "host"), ("cheap", [0], "auto"),
                                                  ("odd", [0, 0, 0], "copy"), ("cheap", [0, 0], "copy"), ("odd", [0] * 5, "copy")])
def test_multi_dataset_at_config5_scale_vs_oracle_fixture(pkg, golden, name, devices, gather):
    """32 768 / 32 767 slots (maxLog2NSlots = 15) through cp2_multi_dataset_build: sha256 over all slot roots, the dataset
    root as EVERY shard's device computed it, input.json byte-exact (sha256 of the oracle's text) on every shard edge."""
    g = golden("config5.json")[name]
    c, n = g["config"], g["config"]["nSlots"]
    cfg = pkg.make_config(**c)
    m = pkg.Multi(devices)
    assert m.count == len(devices) and m.devices() == devices
    m.set_policy({"auto": pkg.GATHER_AUTO, "rccl": pkg.GATHER_RCCL, "host": pkg.GATHER_HOST, "copy": pkg.GATHER_COPY}[gather], 0)
    ds = m.dataset(cfg)
    shards = ds.shards()
    world = len(devices)
    assert [(f, k) for _, f, k in shards] == [pkg.shard_range(n, r, world) for r in range(world)]
    mode = m.gather_mode()
    if gather == "rccl":
        assert mode.startswith("rccl"), mode                     # a one-rank communicator: the RCCL code path itself ran
    elif world == 1:
        assert mode.startswith("none"), mode
    elif gather == "copy":
        # device-to-device copies through the RCCL path's own buffers: with 32 767 slots the shards differ by one row, so the
        # padded layout AND its compaction run here (the only parts of the RCCL path a one-GPU box cannot reach through RCCL)
        assert mode.startswith("copy (%d device-to-device copies" % (world * (world - 1))), mode
    else:
        assert mode.startswith("host") and ("more than one shard" in mode or "requested" in mode), mode
    assert sha(ds.slot_roots()) == g["slot_roots_sha256"]
    assert hexroot(ds.root()) == g["dataset_root_hex"]
    for i in range(world):
        assert hexroot(ds.shard_root(i)) == g["dataset_root_hex"], i
    edges = sorted({e for _, f, k in shards for e in (f, f + k - 1)})
    for slot in edges:
        if str(slot) in g["inputs"]:
            text = ds.proof_input(slot, g["entropy"]).json()
            assert tsha(text) == g["inputs"][str(slot)]["json_sha256"] and len(text) == g["inputs"][str(slot)]["json_bytes"], slot
    assert {0, n - 1} <= set(edges) and all(str(e) in g["inputs"] for e in (0, n - 1))
    with pytest.raises(pkg.CodexP2Error):
        ds.proof_input(n, g["entropy"])                          # no shard holds it: slot index out of range
    ds.free()
    m.close()


def _n_devices():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_n_devices() < 2, reason="needs at least two GPUs: RCCL between REAL devices (a one-GPU box covers the one-rank communicator and the host gather)")
@pytest.mark.parametrize("name", ["cheap", "odd"])
def test_multi_over_every_real_device_rccl_vs_oracle_fixture(pkg, golden, name):
    """On a multi-GPU node: every visible device, RCCL asked for by name (no silent host fallback), 32 768 / 32 767 slots against the
    oracle-only fixture -- the device-to-device ncclAllGather itself, uneven shards included -- then the same dataset cut by units."""
    g = golden("config5.json")[name]
    c, n = g["config"], g["config"]["nSlots"]
    m = pkg.Multi(list(range(_n_devices())))                      # explicitly: with no device named a cp2_multi takes ONE (several are opt-in)
    world = m.count
    assert world == _n_devices()
    m.set_policy(pkg.GATHER_RCCL, 1)
    m.set_split(1)
    ds = m.dataset(pkg.make_config(**c))
    assert m.gather_mode().startswith("rccl") and len(ds.shards()) == world
    assert sha(ds.slot_roots()) == g["slot_roots_sha256"]
    for i in range(world):
        assert hexroot(ds.shard_root(i)) == g["dataset_root_hex"], i
    for slot in (0, n - 1):
        assert tsha(ds.proof_input(slot, g["entropy"]).json()) == g["inputs"][str(slot)]["json_sha256"]
    ds.free()
    m.set_policy(pkg.GATHER_COPY, 1)                              # the same exchange as peer copies between the real devices
    ds = m.dataset(pkg.make_config(**c))
    assert m.gather_mode().startswith("copy") and sha(ds.slot_roots()) == g["slot_roots_sha256"]
    for i in range(world):
        assert hexroot(ds.shard_root(i)) == g["dataset_root_hex"], i
    ds.free()
    m.set_policy(pkg.GATHER_RCCL, 1)
    # few, large slots over all devices: the reference's default run cut by units, RCCL carrying the unit roots
    m0 = golden("proof_inputs.json")["inputs"]["params_default"]
    m.set_split(0)
    ds = m.dataset(pkg.make_config(**m0["config"]))
    assert len(ds.shards()) == min(world, 11 * ds.units_per_slot) and m.gather_mode().startswith("rccl")
    assert ds.proof_input(m0["slotIndex"], m0["entropy"]).json() == golden("input_params_default.json")
    ds.free()
    m.close()


def test_multi_argument_checks(pkg, ctx):
    """bad knobs and pointers are refused, not dereferenced"""
    import ctypes
    m = pkg.Multi([0])
    for bad in (3, 6, -1):
        with pytest.raises(pkg.CodexP2Error):
            m.set_split(bad)                                     # not a power of two / negative
    with pytest.raises(pkg.CodexP2Error):
        m.set_policy(7, 0)
    with pytest.raises(pkg.CodexP2Error):
        pkg.Multi([5])                                           # no such device on a one-GPU box: loud, no fallback
    cfg = pkg.make_config(maxDepth=8, maxLog2NSlots=2, cellSize=128, blockSize=512, nSlots=3, nCells=16, nSamples=3, seed=7)
    ds = ctx.dataset(cfg)
    assert ctx.L.cp2_dataset_set_roots_dev(ds.h, ctypes.c_void_p(ds.local_roots_dev() + 8)) == -6    # CP2_ERR_ALIGN
    assert ctx.L.cp2_dataset_set_roots_dev(ds.h, None) == -1
    first, count = ctypes.c_uint64(), ctypes.c_uint64()
    assert ctx.L.cp2_dataset_range(ds.h, ctypes.byref(first), ctypes.byref(count)) == 0 and (first.value, count.value) == (0, 3)
    ds.set_roots_dev(ds.local_roots_dev())                       # all three roots are local: the tree over the device buffer itself
    want = ds.root().copy()
    ds.set_roots(ds.local_roots())
    assert np.array_equal(ds.root(), want)
    ds.free()
    m.close()


def test_eight_shards_at_config5_scale_down_vs_oracle_fixture(pkg, golden):
    """The 8-GPU split itself -- 32 768 slots x 2^12 cells, eight shards of 4096 slots, eight host threads and contexts -- on the
    one device a test box has: dataset root as every shard computed it, input.json on the edges of the 8-way split."""
    g = golden("config5.json").get("scaled")
    if not g:
        pytest.skip("config5.json has no `scaled` fixture")
    m = pkg.Multi([0] * 8)
    ds = m.dataset(pkg.make_config(**g["config"]))
    assert [k for _, _, k in ds.shards()] == [4096] * 8 and ds.units_per_slot == 1
    assert sha(ds.slot_roots()) == g["slot_roots_sha256"]
    assert all(hexroot(ds.shard_root(i)) == g["dataset_root_hex"] for i in range(8))
    for slot in (0, 4095, 4096, 16383, 28672, 32767):
        assert tsha(ds.proof_input(slot, g["entropy"]).json()) == g["inputs"][str(slot)]["json_sha256"], slot
    ds.free()
    m.close()


def test_forced_rccl_on_a_repeated_device_is_refused_with_a_reason(pkg, golden):
    c = golden("config5.json")["cheap"]["config"]
    m = pkg.Multi([0, 0])
    m.set_policy(pkg.GATHER_RCCL, 0)
    with pytest.raises(pkg.CodexP2Error) as e:
        m.dataset(pkg.make_config(**c))
    assert e.value.status == -1 and "more than one shard" in str(e.value)
    m.close()


def test_small_datasets_stay_on_one_device_unless_told_otherwise(pkg, golden):
    """workflow/params.sh's default run (11 slots x 512 cells = 5632 cells, far below one hash-kernel residency): one shard,
    one context, no exchange -- and the same input.json when it is spread over three contexts anyway (uneven 4 + 4 + 3)."""
    m0 = golden("proof_inputs.json")["inputs"]["params_default"]
    cfg = pkg.make_config(**m0["config"])
    want = golden("input_params_default.json")
    m = pkg.Multi([0, 0, 0])
    ds = m.dataset(cfg)
    assert len(ds.shards()) == 1 and m.gather_mode().startswith("none")
    assert ds.proof_input(m0["slotIndex"], m0["entropy"]).json() == want
    ds.free()
    m.set_policy(pkg.GATHER_AUTO, 1)
    m.set_split(1)                                               # whole slots only (left to itself it would cut these 11 slots by units)
    ds = m.dataset(cfg)
    assert [(f, k) for _, f, k in ds.shards()] == [(0, 4), (4, 4), (8, 3)] and m.gather_mode().startswith("host")
    assert ds.proof_input(m0["slotIndex"], m0["entropy"]).json() == want
    ds.free()
    m.close()


def test_multi_streamed_and_batched_exports_equal_the_object_path(pkg, oracle, tmp_path):
    """cp2_multi_dataset_build_streamed + _export_streamed and cp2_multi_dataset_export_proof_inputs over three shards
    (7 slots: 3 + 2 + 2): every file byte-identical to the single-context object path and, for the edges, to the oracle."""
    C, P = oracle
    c = dict(maxDepth=14, maxLog2NSlots=3, cellSize=256, blockSize=2048, nSlots=7, nCells=128, nSamples=9, seed=777)
    entropy = 424242
    cfg = pkg.make_config(**c)
    single = pkg.Context(0)
    ref = single.dataset(cfg)
    want = {s: ref.proof_input(s, entropy).json() for s in range(7)}
    for s in (0, 2, 3, 6):
        assert want[s] == P.export_json(expected_proof_input_fast(C, P, c, s, entropy, threads=4))
    m = pkg.Multi([0, 0, 0])
    m.set_policy(pkg.GATHER_AUTO, 1)
    # round 5: a streamed build follows the same plan as a plain one -- 7 slots over 3 contexts are cut by units (8 per slot), built
    # balanced, exchanged, and every input.json made from the devices that hold the units (two-phase); the texts are the object path's
    su = m.dataset_streamed(cfg, entropy, threads=4, group_slots=1)
    assert su.units_per_slot == 8 and [(f, k) for _, f, k in su.shards()] == [(0, 19), (19, 19), (38, 18)]
    out0 = tmp_path / "streamed_units"
    out0.mkdir()
    assert su.export_streamed(str(out0), threads=3) == sum(len(t) for t in want.values())
    for s in range(7):
        assert su.streamed_json(s) == want[s] and open(out0 / ("input_%d.json" % s)).read() == want[s]
    assert su.proof_input(5, 99).json() == ref.proof_input(5, 99).json()          # still a normal dataset for other entropies
    su.free()
    m.set_split(1)                                                                 # whole slots by request: the overlapped per-slot pipeline on every shard
    sd = m.dataset_streamed(cfg, entropy, threads=4, group_slots=1)
    assert [(f, k) for _, f, k in sd.shards()] == [(0, 3), (3, 2), (5, 2)]
    m.set_split(0)
    out1, out2 = tmp_path / "streamed", tmp_path / "batched"
    out1.mkdir()
    out2.mkdir()
    total = sd.export_streamed(str(out1), threads=3)
    assert total == sum(len(t) for t in want.values())
    for s in range(7):
        assert sd.streamed_json(s) == want[s] and open(out1 / ("input_%d.json" % s)).read() == want[s]
    ds = m.dataset(cfg)
    total2 = ds.export_proof_inputs([6, 0, 3, 4, 1], entropy, str(out2), threads=3, batch=2)
    assert sorted(os.listdir(out2)) == ["input_%d.json" % s for s in (0, 1, 3, 4, 6)]
    assert total2 == sum(len(want[s]) for s in (0, 1, 3, 4, 6))
    for s in (0, 1, 3, 4, 6):
        assert open(out2 / ("input_%d.json" % s)).read() == want[s]
    sd.free()
    ds.free()
    ref.free()
    m.close()
    single.close()


def test_multi_cached_build_writes_one_file_per_shard(pkg, tmp_path):
    c = dict(maxDepth=12, maxLog2NSlots=3, cellSize=128, blockSize=1024, nSlots=5, nCells=64, nSamples=4, seed=31)
    cfg = pkg.make_config(**c)
    m = pkg.Multi([0, 0])
    m.set_policy(pkg.GATHER_AUTO, 1)
    m.set_split(1)                                               # whole slots (5 slots over 2 contexts would be cut by units: see below)
    cache = str(tmp_path / "trees.cp2")
    a = m.dataset(cfg, cache=cache)
    text = a.proof_input(4, 5).json()
    assert sorted(os.listdir(tmp_path)) == ["trees.cp2.shard0of2", "trees.cp2.shard1of2"]
    b = m.dataset(cfg, cache=cache)                              # loaded: no cell is hashed again
    assert b.proof_input(4, 5).json() == text and hexroot(b.root()) == hexroot(a.root())
    single = pkg.Context(0)
    assert single.dataset(cfg).proof_input(4, 5).json() == text
    single.close()
    a.free()
    b.free()
    m.close()


def test_multi_slot_files_missing_file_names_device_and_file(pkg, oracle, tmp_path):
    C, _ = oracle
    base = str(tmp_path / "slot")
    for k in (0, 1, 3):
        C.gen_fake_cells(C.slot_seed(1, k), 0, 64, 128).tofile("%s%d.dat" % (base, k))
    cfg = pkg.make_config(maxDepth=10, maxLog2NSlots=2, cellSize=128, blockSize=1024, nSlots=4, nCells=64, nSamples=4, file=base)
    m = pkg.Multi([0, 0])
    m.set_policy(pkg.GATHER_AUTO, 1)
    with pytest.raises(pkg.CodexP2Error) as e:
        m.dataset(cfg)
    assert e.value.status == -5 and "slot2.dat" in str(e.value) and "slots 2..4" in str(e.value)
    C.gen_fake_cells(C.slot_seed(1, 2), 0, 64, 128).tofile(base + "2.dat")
    ds = m.dataset(cfg)                                          # the handle keeps working after a failed build
    single = pkg.Context(0)
    assert ds.proof_input(2, 9).json() == single.dataset(cfg).proof_input(2, 9).json()
    single.close()
    ds.free()
    m.close()


def test_cli_twin_spreads_over_contexts_without_a_new_flag(pkg, golden, tmp_path):
    """The drop-in itself: CODEX_P2_GPUS picks the devices (here two contexts on device 0), CODEX_P2_MIN_CELLS=1 makes even
    params.sh's small default spread; the flag set and the output stay the reference's."""
    args = ["--depth=32", "--maxslots=256", "--cellsize=2048", "--blocksize=65536", "--nsamples=5", "--entropy=1234567",
            "--seed=12345", "--nslots=11", "--ncells=512", "--index=3", "--field=bn254", "--hash=poseidon2"]
    want = golden("input_params_default.json")
    for env_extra, marker in (({}, "none"), ({"CODEX_P2_GPUS": "0,0", "CODEX_P2_MIN_CELLS": "1"}, "host"), ({"CODEX_P2_GPUS": "1"}, "none"),
                              ({"CODEX_P2_GPUS": "0,0,0", "CODEX_P2_MIN_CELLS": "1", "CODEX_P2_GATHER": "copy"}, "copy")):
        out = str(tmp_path / ("input_%s.json" % marker))
        r = subprocess.run([pkg.CLI_PATH] + args + ["-v", "--output=" + out], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, CP2_TRACE="1", **env_extra))
        assert r.returncode == 0, r.stderr
        assert open(out).read() == want
        assert "[cp2 trace] slot roots exchanged: %s" % marker in r.stderr, r.stderr
    # RCCL in a process that never loaded torch: librccl is found by dlopen and a one-rank communicator carries the exchange
    out = str(tmp_path / "input_rccl.json")
    r = subprocess.run([pkg.CLI_PATH] + args + ["--output=" + out], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, CP2_TRACE="1", CODEX_P2_GPUS="0,", CODEX_P2_GATHER="rccl"))
    assert r.returncode == 0 and open(out).read() == want and "slot roots exchanged: rccl" in r.stderr, r.stderr
    r = subprocess.run([pkg.CLI_PATH] + args + ["--output=" + str(tmp_path / "x.json")], capture_output=True, text=True, timeout=60,
                       env=dict(os.environ, CODEX_P2_GPUS="7,"))
    assert r.returncode != 0 and "no usable gfx950 HIP device" in r.stderr      # a device that is not there: loud, no fallback
    for bad in ("two", "0,x", "-1", "0;1"):
        r = subprocess.run([pkg.CLI_PATH] + args + ["--output=" + str(tmp_path / "x.json")], capture_output=True, text=True, timeout=60,
                           env=dict(os.environ, CODEX_P2_GPUS=bad))
        assert r.returncode != 0 and "invalid argument" in r.stderr, (bad, r.stderr)     # a malformed list is refused, not guessed at
    for var, bad in (("CODEX_P2_SPLIT", "3"), ("CODEX_P2_SPLIT", "-2"), ("CODEX_P2_SPLIT", "two"), ("CODEX_P2_MIN_CELLS", "1e6"), ("CODEX_P2_MIN_CELLS", "-1")):
        r = subprocess.run([pkg.CLI_PATH] + args + ["--output=" + str(tmp_path / "x.json")], capture_output=True, text=True, timeout=60,
                           env=dict(os.environ, **{var: bad}))
        assert r.returncode != 0 and "invalid argument" in r.stderr and var in r.stderr, (var, bad, r.stderr)


def test_cli_twin_reads_slot_files_whole_and_by_units(pkg, oracle, golden, tmp_path):
    """`--file=<base>` (slot k = <base>k.dat, dataset.nim:34; "untested" in the reference, proof_input/README.md:40): files that
    hold testMain.hs's fake data must give the committed input.json -- on one context, and with the five slots cut by units over
    two contexts (every context reads ITS byte range of the files)."""
    C, _ = oracle
    c = golden("proof_inputs.json")["inputs"]["testmain_small"]["config"]
    base = str(tmp_path / "slotdata")
    for k in range(c["nSlots"]):
        C.gen_fake_cells(C.slot_seed(c["seed"], k), 0, c["nCells"], c["cellSize"]).tofile("%s%d.dat" % (base, k))
    args = [pkg.CLI_PATH, "-d:16", "-N=32", "-c128", "-b:4096", "-n=10", "-e:1234567", "-f=" + base, "-s=5", "-K:256", "-i3", "-F:bn254", "-H=poseidon2"]
    for env_extra, marker in (({}, "slot trees on 1"), ({"CODEX_P2_GPUS": "0,0", "CODEX_P2_MIN_CELLS": "1"}, "unit trees on 2")):
        out = str(tmp_path / "file.json")
        r = subprocess.run(args + ["-v", "-o=" + out], capture_output=True, text=True, timeout=300, env=dict(os.environ, CP2_TRACE="1", **env_extra))
        assert r.returncode == 0, r.stderr
        assert 'dataSource = (kind: SlotFile, filename: "%s")' % base in r.stdout
        assert marker in r.stderr, r.stderr
        assert open(out).read() == golden("input_testmain_small.json")
    os.remove(base + "4.dat")                                        # a missing slot file: an error that names it, no partial output
    r = subprocess.run(args + ["-o=" + str(tmp_path / "x.json")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "slotdata4.dat" in r.stderr and not os.path.exists(tmp_path / "x.json")


def test_cli_twin_fuzz_against_the_oracle(pkg, oracle, tmp_path):
    """The drop-in end to end on random flag sets (cli.nim:118-154): geometry, slot count, samples, index, entropy, seed --
    one context, two or three contexts, whole slots or cut by units -- input.json byte for byte against the Python restatement
    (gen_input/bn254.nim:35-79 + json/bn254.nim:57-78) and accepted by the circuit-side checker."""
    C, P = oracle
    rng = np.random.default_rng(20261004)
    runs = 0
    for it in range(24):
        cs = int(rng.choice([64, 128, 256, 2048]))
        cpb = int(rng.choice([1, 2, 8, 32]))
        nblocks = int(rng.choice([1, 2, 4, 16, 64]))
        nc = cpb * nblocks
        if nc < 2:
            continue
        n_slots = int(rng.integers(1, 14))
        c = dict(maxDepth=int(rng.integers(12, 33)), maxLog2NSlots=int(rng.integers(4, 9)), cellSize=cs, blockSize=cs * cpb, nSlots=n_slots, nCells=nc,
                 nSamples=int(rng.integers(1, 25)), seed=int(rng.integers(0, 1 << 31)))
        index, entropy = int(rng.integers(0, n_slots)), int(rng.integers(0, 1 << 62))
        args = ["--depth=%d" % c["maxDepth"], "--maxslots=%d" % (1 << c["maxLog2NSlots"]), "--cellsize=%d" % cs, "--blocksize=%d" % (cs * cpb),
                "--nsamples=%d" % c["nSamples"], "--entropy=%d" % entropy, "--seed=%d" % c["seed"], "--nslots=%d" % n_slots,
                "-K:%d" % nc if it % 2 else "--log2ncells=%d" % (nc.bit_length() - 1), "--index=%d" % index, "--field=bn254", "--hash=poseidon2"]
        env = dict(os.environ)
        mode = it % 3
        if mode:
            env.update(CODEX_P2_GPUS=",".join(["0"] * (mode + 1)), CODEX_P2_MIN_CELLS="1", CODEX_P2_SPLIT=str(int(rng.choice([0, 1, 2, 4]))))
        out = str(tmp_path / ("fuzz%d.json" % it))
        r = subprocess.run([pkg.CLI_PATH] + args + ["--output=" + out], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, (args, r.stderr)
        prf = expected_proof_input_fast(C, P, c, index, entropy, threads=_threads())
        assert open(out).read() == P.export_json(prf), (args, mode, env.get("CODEX_P2_SPLIT"))
        if it % 4 == 0 and cpb > 1:
            assert P.circuit_check(prf, c)
        runs += 1
    assert runs >= 20


# ---- roots-only datasets: the trees do not have to fit the device ------------------------------------------------------
@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("name", ["params_default", "testmain_small", "odd_slots_one_block"])
def test_roots_only_dataset_gives_the_same_proof_inputs(pkg, golden, oracle, tmp_path, name, mode):
    """cp2_set_keep_trees(0): the slot trees are built batch by batch and dropped, the roots stay, the tree of the proved slot is
    rebuilt on demand; cp2_set_keep_trees(2): the part of every slot tree from the block roots up stays, the bottom of each path
    is recomputed from the touched blocks -- same roots, same dataset tree, same input.json (the committed oracle text), fake and
    file sources, single, batched and streamed generation, also through cp2_multi and the cli twin (CODEX_P2_KEEP_TREES)."""
    C, _ = oracle
    m0 = golden("proof_inputs.json")["inputs"][name]
    c = m0["config"]
    want = golden("input_%s.json" % name)
    single = pkg.Context(0)
    keep = single.dataset(pkg.make_config(**c))
    assert keep.keeps_trees                                       # small: left to itself the library keeps the trees
    single.set_keep_trees(mode)
    base = str(tmp_path / "slot")
    for k in range(c["nSlots"]):
        C.gen_fake_cells(C.slot_seed(c["seed"], k), 0, c["nCells"], c["cellSize"]).tofile("%s%d.dat" % (base, k))
    for src in (dict(seed=c["seed"]), dict(file=base)):
        cfg = pkg.make_config(**dict({k: v for k, v in c.items() if k != "seed"}, **src))
        ds = single.dataset(cfg)
        assert ds.tree_mode == mode
        assert np.array_equal(ds.local_roots(), keep.local_roots()) and np.array_equal(ds.root(), keep.root())
        assert ds.proof_input(m0["slotIndex"], m0["entropy"]).json() == want
        sd = single.dataset_streamed(cfg, m0["entropy"], threads=3, group_slots=2)   # streamed: bodies made while a batch's trees exist
        assert sd.tree_mode == mode and np.array_equal(sd.local_roots(), keep.local_roots())
        sd.export_streamed(None, threads=2)
        assert sd.streamed_json(m0["slotIndex"]) == want
        assert sd.streamed_json(c["nSlots"] - 1) == keep.proof_input(c["nSlots"] - 1, m0["entropy"]).json()
        assert sd.proof_input(0, 987654321).json() == keep.proof_input(0, 987654321).json()   # another entropy: the slot's tree rebuilt
        sd.free()
        every = ds.proof_inputs(list(range(c["nSlots"])), m0["entropy"])          # a batch: one rebuilt tree after the other
        assert every[m0["slotIndex"]].json() == want
        assert every[0].json() == keep.proof_input(0, m0["entropy"]).json()
        ds.free()
    single.set_keep_trees(-1)
    # two contexts, whole slots, both roots-only
    m = pkg.Multi([0, 0])
    m.set_policy(pkg.GATHER_AUTO, 1)
    m.set_split(1)
    for i in range(m.count):
        m.ctx(i).set_keep_trees(mode)
    md = m.dataset(pkg.make_config(**c))
    assert md.proof_input(m0["slotIndex"], m0["entropy"]).json() == want and hexroot(md.root()) == hexroot(keep.root())
    md.free()
    m.close()
    keep.free()
    single.close()


@pytest.mark.parametrize("mode,marker", [("0", "roots-only build"), ("2", "compact build")])
def test_cli_twin_roots_only(pkg, golden, tmp_path, mode, marker):
    args = ["--depth=32", "--maxslots=256", "--cellsize=2048", "--blocksize=65536", "--nsamples=5", "--entropy=1234567",
            "--seed=12345", "--nslots=11", "--ncells=512", "--index=3", "--field=bn254", "--hash=poseidon2"]
    out = str(tmp_path / "input.json")
    r = subprocess.run([pkg.CLI_PATH] + args + ["--output=" + out], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, CP2_TRACE="1", CODEX_P2_KEEP_TREES=mode))
    assert r.returncode == 0 and marker in r.stderr, r.stderr
    assert open(out).read() == golden("input_params_default.json")


@pytest.mark.parametrize("mode", [2, 0])
def test_compact_and_roots_only_datasets_are_cached_as_what_they_keep(pkg, oracle, golden, tmp_path, mode):
    """cp2_dataset_build_cached on a compact (roots-only) dataset writes the compact layers (the roots) -- 1/32 (1/2^21) of the
    tree cache -- and a later build loads them instead of hashing any slot: same proof inputs; a damaged cache or changed slot
    files mean rebuild, never stale layers.  Through the cli twin as well (CODEX_P2_CACHE + CODEX_P2_KEEP_TREES)."""
    C, P = oracle
    m0 = golden("proof_inputs.json")["inputs"]["testmain_small"]
    c = m0["config"]
    want = golden("input_testmain_small.json")
    ctx = pkg.Context(0)
    ctx.set_keep_trees(mode)
    cache = str(tmp_path / "kept.cp2")
    a = ctx.dataset(pkg.make_config(**c), cache=cache)
    assert a.tree_mode == mode and a.proof_input(m0["slotIndex"], m0["entropy"]).json() == want
    size = os.path.getsize(cache)
    full_nodes = c["nSlots"] * (2 * c["nCells"] - 1) * 32
    assert size < (full_nodes / 20 if mode == 2 else 1024)         # compact: 1/32 of the nodes (+ header); roots: 32 bytes per slot
    b = ctx.dataset(pkg.make_config(**c), cache=cache)             # loaded
    assert b.tree_mode == mode and b.proof_input(m0["slotIndex"], m0["entropy"]).json() == want
    assert np.array_equal(a.local_roots(), b.local_roots())
    a.free()
    b.free()
    raw = bytearray(open(cache, "rb").read())                      # one flipped payload byte: checksum mismatch -> rebuilt and rewritten
    raw[-7] ^= 0x20
    open(cache, "wb").write(bytes(raw))
    d = ctx.dataset(pkg.make_config(**c), cache=cache)
    assert d.proof_input(m0["slotIndex"], m0["entropy"]).json() == want and open(cache, "rb").read() != bytes(raw)
    d.free()
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]
    # SlotFile source: a cache is keyed on size + mtime of every slot file
    base = str(tmp_path / "slot")
    for k in range(c["nSlots"]):
        C.gen_fake_cells(C.slot_seed(c["seed"], k), 0, c["nCells"], c["cellSize"]).tofile("%s%d.dat" % (base, k))
    fcfg = pkg.make_config(file=base, **{k: v for k, v in c.items() if k != "seed"})
    fcache = str(tmp_path / "kept_files.cp2")
    assert ctx.dataset(fcfg, cache=fcache).proof_input(m0["slotIndex"], m0["entropy"]).json() == want
    assert ctx.dataset(fcfg, cache=fcache).proof_input(m0["slotIndex"], m0["entropy"]).json() == want          # loaded
    C.gen_fake_cells(C.slot_seed(999, 3), 0, c["nCells"], c["cellSize"]).tofile(base + "3.dat")                # new contents for slot 3
    changed = ctx.dataset(fcfg, cache=fcache).proof_input(m0["slotIndex"], m0["entropy"]).json()
    ctx.set_keep_trees(1)
    assert changed == ctx.dataset(fcfg).proof_input(m0["slotIndex"], m0["entropy"]).json() != want
    ctx.close()
    # the drop-in: first run builds and writes, second run loads (no slot is hashed again)
    args = [pkg.CLI_PATH, "-d:16", "-N=32", "-c128", "-b:4096", "-n=10", "-e:1234567", "-S12345", "-s=5", "-K:256", "-i3", "-F:bn254", "-H=poseidon2"]
    env = dict(os.environ, CP2_TRACE="1", CODEX_P2_KEEP_TREES=str(mode), CODEX_P2_CACHE=str(tmp_path / "cli.cp2"))
    for marker in ("built and written to the cache", "loaded from the cache"):
        out = str(tmp_path / "cli.json")
        r = subprocess.run(args + ["-o=" + out], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and marker in r.stderr, r.stderr
        assert open(out).read() == want


def test_compact_dataset_proves_every_slot_in_batched_passes(pkg, golden, tmp_path):
    """A node proves EVERY slot it holds each period.  Config 4's shape (4096 slots x 2^12 cells, 100 samples) kept compact: all
    4096 proof inputs for a new entropy come from batched passes over the touched blocks (327 slots per pass), written as
    input.json files; slots 1234 and 4095 against the oracle-only fixture, every file against the dataset that keeps every node."""
    g = golden("fullsize.json")["config4"]
    cfg = pkg.make_config(**g["config"])
    ctx = pkg.Context(0)
    ctx.set_keep_trees(2)
    ds = ctx.dataset(cfg)
    assert ds.tree_mode == 2 and hexroot(ds.root()) == g["dataset_root_hex"]
    out = tmp_path / "compact"
    out.mkdir()
    total = ds.export_proof_inputs(list(range(4096)), g["entropy"], str(out), threads=_threads(), batch=1024)
    assert len(os.listdir(out)) == 4096 and total == sum(os.path.getsize(out / f) for f in os.listdir(out))
    for slot, want in g["inputs"].items():
        text = open(out / ("input_%s.json" % slot)).read()
        assert tsha(text) == want["json_sha256"] and len(text) == want["json_bytes"], slot
    ds.free()
    ctx.set_keep_trees(1)
    full = ctx.dataset(cfg)
    for slot in (0, 326, 327, 2048, 4094):                          # the edges of the 327-slot passes among them
        assert open(out / ("input_%d.json" % slot)).read() == full.proof_input(slot, g["entropy"]).json(), slot
    full.free()
    ctx.close()


def test_compact_dataset_notices_changed_slot_data(pkg, oracle, tmp_path):
    """Compact datasets re-hash the touched blocks of the slot FILE at proof time: data that changed since the build no longer
    hashes to the stored block root -- an I/O error naming block and slot, never a proof over mixed data."""
    C, _ = oracle
    c = dict(maxDepth=12, maxLog2NSlots=2, cellSize=128, blockSize=1024, nSlots=2, nCells=256, nSamples=40)
    base = str(tmp_path / "slot")
    for k in range(2):
        C.gen_fake_cells(C.slot_seed(5, k), 0, 256, 128).tofile("%s%d.dat" % (base, k))
    ctx = pkg.Context(0)
    ctx.set_keep_trees(2)
    ds = ctx.dataset(pkg.make_config(file=base, **c))
    assert ds.tree_mode == 2
    good = ds.proof_input(1, 77).json()
    raw = bytearray(open(base + "1.dat", "rb").read())
    for off in range(0, len(raw), 1024):                            # one byte in every block
        raw[off] ^= 1
    open(base + "1.dat", "wb").write(bytes(raw))
    with pytest.raises(pkg.CodexP2Error) as e:
        ds.proof_input(1, 77)
    assert e.value.status == -5 and "does not hash to its stored root" in str(e.value) and "slot 1" in str(e.value)
    assert ds.proof_input(0, 77).json() and good                    # the untouched slot still proves
    ds.free()
    ctx.close()


def test_bigslots_roots_only_vs_oracle_fixture(pkg, golden):
    """8 slots of 8 GiB with only the roots kept (2 GiB of nodes in flight instead of 2 GiB per 8 slots resident ... at 4096 slots
    per GPU, config 5's nominal share, the resident trees would need 1 TiB): roots, dataset root and input.json of slots 0 and 7
    -- each from its tree rebuilt on demand -- against the oracle-only fixture."""
    g = _big(golden)
    ctx = pkg.Context(0)
    for mode in (0, 2):
        ctx.set_keep_trees(mode)
        ds = ctx.dataset(pkg.make_config(**g["config"]))
        assert ds.tree_mode == mode
        assert [hexroot(r) for r in ds.local_roots()] == g["slot_roots_hex"] and hexroot(ds.root()) == g["dataset_root_hex"]
        for slot in ((0, 7) if mode == 0 else range(8)):
            text = ds.proof_input(slot, g["entropy"]).json()
            assert tsha(text) == g["inputs"][str(slot)]["json_sha256"], (mode, slot)
        ds.free()
    # the compact layers of the 8 slots cached (64 MiB against 2 GiB of nodes): the second build hashes nothing
    import tempfile
    import time
    with tempfile.TemporaryDirectory() as td:
        ctx.set_keep_trees(2)
        cache = os.path.join(td, "big.cp2")
        ctx.dataset(pkg.make_config(**g["config"]), cache=cache).free()
        assert os.path.getsize(cache) < 80 << 20
        t0 = time.perf_counter()
        ds = ctx.dataset(pkg.make_config(**g["config"]), cache=cache)
        t_load = time.perf_counter() - t0
        assert t_load < 0.6 and hexroot(ds.root()) == g["dataset_root_hex"]          # a rebuild takes 1.6 s
        assert tsha(ds.proof_input(5, g["entropy"]).json()) == g["inputs"]["5"]["json_sha256"]
        ds.free()
    ctx.set_keep_trees(0)
    # streamed and roots-only: every input.json of the 8 slots in one pass, nodes of at most 8 slots alive
    sd = ctx.dataset_streamed(pkg.make_config(**g["config"]), g["entropy"], threads=_threads(), group_slots=1)
    assert not sd.keeps_trees and hexroot(sd.root()) == g["dataset_root_hex"]
    sd.export_streamed(None, threads=2)
    for slot in range(8):
        assert tsha(sd.streamed_json(slot)) == g["inputs"][str(slot)]["json_sha256"], slot
    sd.free()
    ctx.close()


# ---- by units: several devices sharing ONE slot ("within one very large slot the same scheme one level down") ----
def test_unit_roots_and_paths_are_pieces_of_the_slot_tree(pkg, ctx, oracle, tmp_path):
    """cp2_slot_trees_build_fake_units / _file_units: the root of a unit is the node of its slot's tree above the unit's cells,
    and a path inside a unit is the bottom of the merged path of the same cell (merkle.nim:21-42,86-100) -- against the oracle's
    full slot trees; a batch that starts and ends in the middle of slots."""
    C, P = oracle
    cs, bs, nc, S, seed = 256, 2048, 256, 4, 4711                    # 8 cells per block, 32 blocks per slot, units of 8 blocks
    cpb, P_cells = bs // cs, nc // S
    first, n_units = 3, 6                                            # units 3 .. 8: the tail of slot 0, all of slot 1, the head of slot 2
    t = ctx.slot_trees_fake_units(seed, S, first, n_units, cs, bs, P_cells)
    assert t.count == n_units and t.depth == 3 + 3
    roots = t.roots()
    big = {}
    for slot in (0, 1, 2):
        big[slot] = C.merkle_tree(C.fake_slot_block_roots(C.slot_seed(seed, slot), cs, bs, nc, 4))
    for i in range(n_units):
        u = first + i
        assert np.array_equal(roots[i], big[u // S][3][u % S]), u     # layer 3 of the slot's tree: one node per 8 blocks
    # paths of a few cells of unit 5 (= second unit of slot 1), against merkleProof on the oracle's block tree + slot tree
    u, local = 5, [0, 7, 8, 63]
    got, leaves = t.paths(u - first, local, 6)
    cells1 = C.gen_fake_cells(C.slot_seed(seed, 1), 0, nc, cs)
    hashes = C.hash_cells(cells1, cs, threads=4)
    to_int = lambda layers: [C.array_to_felts(l) for l in layers]    # noqa: E731
    for j, c_local in enumerate(local):
        c_slot = (u % S) * P_cells + c_local
        b = c_slot // cpb
        mini = to_int(C.merkle_tree(hashes[b * cpb:(b + 1) * cpb]))
        full = P.merge_merkle_proofs(P.merkle_proof(mini, c_slot % cpb), P.merkle_proof(to_int(big[1]), b))
        assert pkg.array_to_felts(got[j]) == full["merklePath"][:6], c_local
        assert np.array_equal(leaves[j], hashes[c_slot])
    t.free()
    # the same units read from slot files at their byte offsets
    base = str(tmp_path / "slot")
    for slot in (0, 1, 2):
        C.gen_fake_cells(C.slot_seed(seed, slot), 0, nc, cs).tofile("%s%d.dat" % (base, slot))
    tf = ctx.slot_trees_file_units(base, S, first, n_units, cs, bs, P_cells)
    assert np.array_equal(tf.roots(), roots)
    tf.free()
    with pytest.raises(pkg.CodexP2Error):
        ctx.slot_trees_fake_units(seed, 32, 0, 4, cs, bs, cpb)       # a unit must hold at least two whole blocks


@pytest.mark.parametrize("name,devices,want_units,want_counts", [("params_default", [0, 0, 0], 4, [15, 15, 14]),
                                                                  ("testmain_small", [0, 0], 2, [5, 5]),
                                                                  ("odd_slots_one_block", [0, 0], 1, [2, 1])])
def test_few_large_slots_are_cut_by_units(pkg, golden, tmp_path, name, devices, want_units, want_counts):
    """The reference's own configurations over 2-3 contexts: 11 slots over 3 would leave 4 / 4 / 3 (9 % over the busiest
    device's share), so every slot is cut into 4 units (44 units: 15 / 15 / 14); one block per slot cannot be cut (whole
    slots).  input.json equals the committed oracle text either way, and cp2_multi_dataset_export_proof_inputs writes it."""
    m0 = golden("proof_inputs.json")["inputs"][name]
    cfg = pkg.make_config(**m0["config"])
    want = golden("input_%s.json" % name)
    m = pkg.Multi(devices)
    m.set_policy(pkg.GATHER_AUTO, 1)
    ds = m.dataset(cfg)
    assert ds.units_per_slot == want_units and [k for _, _, k in ds.shards()] == want_counts
    assert ds.proof_input(m0["slotIndex"], m0["entropy"]).json() == want
    out = tmp_path / "out"
    out.mkdir()
    total = ds.export_proof_inputs([m0["slotIndex"], 0], m0["entropy"], str(out), threads=2)
    assert open(out / ("input_%d.json" % m0["slotIndex"])).read() == want
    assert total == len(want) + os.path.getsize(out / "input_0.json")
    single = pkg.Context(0)
    ref = single.dataset(cfg)
    assert np.array_equal(ds.slot_roots(), ref.local_roots()) and np.array_equal(ds.root(), ref.root())
    assert open(out / "input_0.json").read() == ref.proof_input(0, m0["entropy"]).json()
    ref.free()
    single.close()
    ds.free()
    m.set_split(1)                                                   # whole slots only, by request
    ds = m.dataset(cfg)
    assert ds.units_per_slot == 1 and ds.proof_input(m0["slotIndex"], m0["entropy"]).json() == want
    ds.free()
    m.close()


def test_units_from_slot_files_and_one_slot_over_several_contexts(pkg, oracle, tmp_path):
    """SlotFile source by units (every context reads ITS byte range of the slot files), and the extreme of the scheme: a dataset
    of ONE slot over four contexts -- against the oracle directly."""
    C, P = oracle
    c = dict(maxDepth=12, maxLog2NSlots=2, cellSize=128, blockSize=1024, nSlots=1, nCells=512, nSamples=11)
    base = str(tmp_path / "slot")
    C.gen_fake_cells(C.slot_seed(99, 0), 0, 512, 128).tofile(base + "0.dat")
    m = pkg.Multi([0, 0, 0, 0])
    m.set_policy(pkg.GATHER_AUTO, 1)
    for src in (dict(seed=99), dict(file=base)):
        ds = m.dataset(pkg.make_config(**c, **src))
        assert ds.units_per_slot == 4 and [(f, k) for _, f, k in ds.shards()] == [(0, 1), (1, 1), (2, 1), (3, 1)]
        want = P.export_json(expected_proof_input_fast(C, P, dict(c, seed=99), 0, 31337, threads=4))
        assert ds.proof_input(0, 31337).json() == want
        ds.free()
    m.close()


def test_cached_builds_are_cut_by_units_too(pkg, oracle, tmp_path, capfd):
    """cp2_multi_dataset_build_cached on a dataset of few, large slots: every shard's UNIT trees go to
    "<cache>.units<S>.shard<i>of<n>" (the cache format records units_per_slot); a second build loads them and hashes nothing;
    damage, another seed and changed slot files are noticed; a batch of units saved by hand loads back as the same units."""
    C, P = oracle
    c = dict(maxDepth=12, maxLog2NSlots=2, cellSize=128, blockSize=1024, nSlots=3, nCells=512, nSamples=9)
    base = str(tmp_path / "slot")
    for k in range(3):
        C.gen_fake_cells(C.slot_seed(99, k), 0, 512, 128).tofile(base + "%d.dat" % k)
    want = P.export_json(expected_proof_input_fast(C, P, dict(c, seed=99), 2, 4242, threads=4))
    m = pkg.Multi([0, 0])
    m.set_policy(pkg.GATHER_AUTO, 1)
    for tag, src in (("fake", dict(seed=99)), ("file", dict(file=base))):
        cache = str(tmp_path / ("units_%s.cp2" % tag))
        cfg = pkg.make_config(**c, **src)
        a = m.dataset(cfg, cache=cache)
        assert a.units_per_slot == 2 and [(f, k) for _, f, k in a.shards()] == [(0, 3), (3, 3)]      # 6 units, 3 per context
        files = sorted(f for f in os.listdir(tmp_path) if f.startswith("units_%s" % tag))
        assert files == ["units_%s.cp2.units2.shard0of2" % tag, "units_%s.cp2.units2.shard1of2" % tag]
        assert a.proof_input(2, 4242).json() == want
        a.free()
        stamp = [os.path.getmtime(str(tmp_path / f)) for f in files]
        os.environ["CP2_TRACE"] = "1"
        capfd.readouterr()
        try:
            b = m.dataset(cfg, cache=cache)                            # loaded: no generator / ingestion / hashing stage in the trace
        finally:
            del os.environ["CP2_TRACE"]
        err = capfd.readouterr().err
        assert "unit trees on 2" in err and "fake slots" not in err, err
        assert b.units_per_slot == 2 and b.proof_input(2, 4242).json() == want
        b.free()
        assert [os.path.getmtime(str(tmp_path / f)) for f in files] == stamp                      # not rewritten
        raw = bytearray(open(str(tmp_path / files[1]), "rb").read())
        raw[-40] ^= 1                                                  # one node of shard 1 flipped: checksum -> rebuilt and rewritten
        open(str(tmp_path / files[1]), "wb").write(bytes(raw))
        d = m.dataset(cfg, cache=cache)
        assert d.proof_input(2, 4242).json() == want and open(str(tmp_path / files[1]), "rb").read() != bytes(raw)
        d.free()
    # another seed under the same cache name: the unit trees do not describe it -> rebuilt
    other = m.dataset(pkg.make_config(**c, seed=100), cache=str(tmp_path / "units_fake.cp2"))
    assert other.proof_input(2, 4242).json() == P.export_json(expected_proof_input_fast(C, P, dict(c, seed=100), 2, 4242, threads=4))
    other.free()
    # a slot file rewritten with other data: size + mtime of the file behind every unit are part of the cache
    cells = C.gen_fake_cells(C.slot_seed(99, 2), 0, 512, 128)
    cells[300] ^= 1
    cells.tofile(base + "2.dat")
    os.utime(base + "2.dat", (1, 1))
    changed = m.dataset(pkg.make_config(**c, file=base), cache=str(tmp_path / "units_file.cp2"))
    text = changed.proof_input(2, 4242).json()
    assert text != want
    changed.free()
    whole = pkg.Context(0)
    assert whole.dataset(pkg.make_config(**c, file=base)).proof_input(2, 4242).json() == text      # = the slots built whole
    # the seam: a batch of units saved and loaded by hand is the same batch
    t = whole.slot_trees_fake_units(99, 4, 5, 3, 128, 1024, 128)       # units 5, 6, 7 of 4 per slot
    t.save(str(tmp_path / "seam.cp2"))
    u = whole.slot_trees_load(str(tmp_path / "seam.cp2"))
    assert np.array_equal(t.roots(), u.roots()) and u.count == 3
    whole.close()
    m.close()


def test_one_128gib_slot_over_eight_contexts(pkg, oracle):
    """"within one very large slot (2^26+ cells) the same scheme applies one level down".  ONE slot of 2^26 cells
    (128 GiB, generated and hashed on the device) over eight contexts -- an eighth of the slot each, unit roots exchanged, the
    three upper layers and the (singleton) dataset tree built once -- against the same slot built whole by one context, and the
    proof input through the circuit-side checker."""
    _, P = oracle
    c = dict(maxDepth=32, maxLog2NSlots=1, cellSize=2048, blockSize=65536, nSlots=1, nCells=1 << 26, nSamples=20, seed=2026)
    cfg = pkg.make_config(**c)
    m = pkg.Multi([0] * 8)
    ds = m.dataset(cfg)
    assert ds.units_per_slot == 8 and [(f, k) for _, f, k in ds.shards()] == [(i, 1) for i in range(8)]
    text = ds.proof_input(0, 31415926).json()
    root, droot = ds.slot_roots()[0].copy(), ds.root().copy()
    ds.free()
    m.close()
    ctx = pkg.Context(0)
    ctx.set_keep_trees(2)                                          # compact: 128 MiB kept of the 4 GiB of nodes
    whole = ctx.dataset(cfg)
    assert np.array_equal(whole.local_roots()[0], root) and np.array_equal(whole.root(), droot)
    pi = whole.proof_input(0, 31415926)
    assert pi.json() == text
    to_int = lambda a: int.from_bytes(np.asarray(a, dtype=np.uint8).tobytes(), "little")   # noqa: E731
    d, sroot, e = pi.roots()
    prf = {"dataSetRoot": to_int(d), "entropy": to_int(e), "nCells": c["nCells"], "nSlots": 1, "slotIndex": 0, "slotRoot": to_int(sroot),
           "slotProof": {"merklePath": [to_int(x) for x in pi.slot_proof()]},
           "proofInputs": [{"cellData": pi.cell_data()[i].tobytes(), "merkleProof": {"merklePath": [to_int(x) for x in pi.merkle_paths()[i]]}}
                           for i in range(c["nSamples"])]}
    assert P.circuit_check(prf, c)
    whole.free()
    ctx.close()


# ---- SURVEY.md 8(d), config 5's other stated scale-down: several slots at the nominal 8 GiB slot size ------------------
def _big(golden):
    try:
        return golden("bigslots.json")
    except FileNotFoundError:
        pytest.skip("tests/golden/bigslots.json not generated (tests/golden/make_bigslots_golden.py)")


def test_bigslots_by_units_over_three_contexts_and_one_8gib_slot_over_two(pkg, golden, tmp_path):
    """8 slots of 8 GiB over three contexts: whole slots would be 3 / 3 / 2, so every slot is cut into 4 units of 2^20 cells
    (32 units: 11 / 11 / 10) and slots 2 and 5 are shared by two contexts each; then ONE 8 GiB slot over two contexts (config 3's
    slot, two units of 4 GiB).  Slot roots, dataset root and input.json against the oracle-only fixtures."""
    g = _big(golden)
    c = g["config"]
    m = pkg.Multi([0, 0, 0])
    ds = m.dataset(pkg.make_config(**c))
    assert ds.units_per_slot == 4 and [k for _, _, k in ds.shards()] == [11, 11, 10]
    assert [hexroot(r) for r in ds.slot_roots()] == g["slot_roots_hex"] and hexroot(ds.root()) == g["dataset_root_hex"]
    for slot in (0, 2, 5, 7):
        text = ds.proof_input(slot, g["entropy"]).json()
        assert tsha(text) == g["inputs"][str(slot)]["json_sha256"] and len(text) == g["inputs"][str(slot)]["json_bytes"], slot
    # round 5: ALL eight at once -- the touched units grouped per device, one batched gather per device, the devices in parallel,
    # the texts formatted on host threads: every file byte-identical to the oracle-only fixture (and so to the per-slot path above)
    out = tmp_path / "batched"
    out.mkdir()
    total = ds.export_proof_inputs(list(range(8)), g["entropy"], str(out), threads=_threads(), batch=3)
    assert total == sum(g["inputs"][str(s_)]["json_bytes"] for s_ in range(8))
    for s_ in range(8):
        assert tsha(open(out / ("input_%d.json" % s_)).read()) == g["inputs"][str(s_)]["json_sha256"], s_
    ds.free()
    # ... and the STREAMED kind follows the same plan (cp2_multi_plan is the plan of every kind): cut by units, built balanced
    # (11 / 11 / 10 units instead of 3 / 3 / 2 slots), exchanged, then every input.json from the devices that hold the units
    assert pkg.multi_plan(pkg.make_config(**c), 3) == (3, 4)
    su = m.dataset_streamed(pkg.make_config(**c), g["entropy"], threads=_threads())
    assert su.units_per_slot == 4 and [k for _, _, k in su.shards()] == [11, 11, 10] and hexroot(su.root()) == g["dataset_root_hex"]
    out2 = tmp_path / "streamed"
    out2.mkdir()
    assert su.export_streamed(str(out2), threads=_threads()) == total
    for s_ in range(8):
        assert tsha(open(out2 / ("input_%d.json" % s_)).read()) == g["inputs"][str(s_)]["json_sha256"] == tsha(su.streamed_json(s_)), s_
    su.free()
    m.close()
    m = pkg.Multi([0, 0])
    ds = m.dataset(pkg.make_config(**dict(c, nSlots=1, maxLog2NSlots=1)))
    assert ds.units_per_slot == 2 and [k for _, _, k in ds.shards()] == [1, 1]
    assert hexroot(ds.slot_roots()[0]) == golden("fullsize.json")["config3"]["slot_root_hex"] == g["slot_roots_hex"][0]
    pi = ds.proof_input(0, g["entropy"])
    assert [int(v) for v in pi.cell_indices()[:8]] == g["cell_indices_first8"][0]
    ds.free()
    m.close()


@pytest.mark.parametrize("devices", [[0], [0, 0]])
def test_bigslots_multi_vs_oracle_fixture(pkg, golden, devices):
    """8 slots x 2^22 cells x 2048 B (64 GiB generated and hashed on the device, 1.17e9 permutations) through cp2_multi: the
    2 GiB staging chunk is a quarter of a slot, so every slot crosses four chunk boundaries (slot_trees.cpp, trees_build_fake)."""
    g = _big(golden)
    c = g["config"]
    cfg = pkg.make_config(**c)
    m = pkg.Multi(devices)
    ds = m.dataset(cfg)
    assert len(ds.shards()) == len(devices)
    roots = ds.slot_roots()
    assert [hexroot(r) for r in roots] == g["slot_roots_hex"] and sha(roots) == g["slot_roots_sha256"]
    assert hexroot(ds.root()) == g["dataset_root_hex"]
    for slot in (0, 3, 4, 7):
        pi = ds.proof_input(slot, g["entropy"])
        assert [int(v) for v in pi.cell_indices()[:8]] == g["cell_indices_first8"][slot]
        text = pi.json()
        assert tsha(text) == g["inputs"][str(slot)]["json_sha256"] and len(text) == g["inputs"][str(slot)]["json_bytes"], slot
    ds.free()
    m.close()


@pytest.mark.parametrize("world", [1, 2])
def test_bigslots_sharded_rank_processes_vs_oracle_fixture(golden, tmp_path, world):
    """The same through distributed.dataset_root_sharded(HipBackend) as 1 and 2 rank processes (4 + 4 slots of 8 GiB)."""
    g = _big(golden)
    res = run_ranks(world, g["config"], g["entropy"], tmp_path, timeout=1100)
    for r in res:
        assert r["native_so_loaded"] and not r["oracle_loaded"]
        assert r["dataset_root_hex"] == g["dataset_root_hex"] and r["all_roots_sha256"] == g["slot_roots_sha256"]
        assert r["count"] == 8 // world
        for slot, digest in r["inputs"].items():
            assert digest == g["inputs"][slot]["json_sha256"], (r["rank"], slot)


def test_bigslots_streamed_one_slot_per_group_and_prefix_datasets(pkg, ctx, golden, tmp_path):
    """A streamed build with group_slots = 1 (sampling / gathers / bodies of slot k overlap the hashing of slot k + 1) at the
    full slot size, and the 5- and 4-slot prefixes of the same slots (an odd dataset tree; the 2 + 2 split's tree)."""
    g = _big(golden)
    c = g["config"]
    for n_slots in (5, 4):
        p = g["prefixes"][str(n_slots)]
        cfg = pkg.make_config(**dict(c, nSlots=n_slots))
        sd = ctx.dataset_streamed(cfg, g["entropy"], threads=_threads(), group_slots=1)
        sd.export_streamed(None, threads=2)
        assert hexroot(sd.root()) == p["dataset_root_hex"]
        for slot, want in p["inputs"].items():
            text = sd.streamed_json(int(slot))
            assert tsha(text) == want["json_sha256"] and len(text) == want["json_bytes"], (n_slots, slot)
        sd.free()
    ctx.trim()


@pytest.mark.gpu
def test_binding_frees_what_lives_in_a_handle_before_the_handle(pkg):
    """The C ABI's rule -- every dataset made through a context / multi handle is freed before the handle -- is kept by the
    binding itself: closing a handle frees what is still alive inside it, and a late free() or destructor touches nothing."""
    c = dict(maxDepth=12, maxLog2NSlots=3, cellSize=128, blockSize=1024, nSlots=6, nCells=64, nSamples=4, seed=5)
    cfg = pkg.make_config(**c)
    ctx = pkg.Context(0)
    ds, trees = ctx.dataset(cfg), ctx.slot_trees_fake(5, 0, 2, 128, 1024, 64)
    text = ds.proof_input(3, 9).json()
    pi = ds.proof_input(3, 9)                                   # a proof input owns its buffers: it outlives everything
    ctx.close()
    assert ds.h is None and trees.h is None
    ds.free(); trees.free()                                     # nothing left to do, nothing touched
    assert pi.json() == text
    m = pkg.Multi([0, 0])
    m.set_policy(pkg.GATHER_AUTO, 1)
    mds = m.dataset(cfg)
    inner = m.ctx(1).dataset(cfg)                               # made through a context the handle owns
    assert m.ctx(1) is m.ctx(1)
    assert mds.proof_input(3, 9).json() == text and inner.proof_input(3, 9).json() == text
    m.close()
    assert mds.h is None and inner.h is None
    del mds, inner, ds, trees
