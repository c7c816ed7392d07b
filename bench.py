#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: Poseidon2-BN254 permutations/s per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: the batched permutation kernel over 2^24 states
(BASELINE.json configs[1], 1.5 GiB in + 1.5 GiB out, already resident in HBM when the clock starts).
With N ranks every rank runs the same batch on its own GPU (weak scaling, no data-path collective: the
permutation has no exchange step); value = all states of all ranks / max-over-ranks time.

The JSON line also carries
  roofline      the dominant kernel (k_permute_batch) against the HBM roof BASELINE.json prescribes:
                achieved = 192 algorithmic B/permutation x 2^24 / average launch time (HIP events on the
                launch stream).  The kernel is VALU-integer bound, so `frac` is small by nature;
                `valu` reports the instruction-issue view next to it.
  cpu_baseline  the C oracle (a port of the same algorithm, NOT the Nim binary: no Nim toolchain exists)
                timed on this box's host cores on a bounded sample, rank 0 at N=1 only.
  extra         config 3 (8 GiB slot -> slot root: sponge + trees) and, for N>1, the config-5 exchange
                (RCCL all-gather of slot roots -> dataset root) on a small dataset.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0             # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_PERM = 192               # 96 B in + 96 B out per permutation (config 2)
N_STATES = 1 << 24                 # BASELINE.json configs[1]


import contextlib


@contextlib.contextmanager
def _stdout_to_stderr():
    """Point the process-level stdout (fd 1, what C libraries printf to) at stderr for the duration."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--states", type=int, default=N_STATES, help="states per step per GPU (default 2^24)")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-3 / config-5 extra legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as g

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher (see the docstring)" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d but --gpus=%d" % (world, args.gpus))
    # Rehearsal knobs (not used by the driver): BENCH_BACKEND=gloo + BENCH_SHARE_GPU=1 run N ranks on ONE GPU with
    # CPU-side collectives, to exercise the multi-rank logic on a one-GPU box (RCCL refuses two ranks per GPU).
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        # RCCL printf()s a version banner on STDOUT when NCCL_DEBUG is set (the GPU boxes export NCCL_DEBUG=VERSION,
        # and NCCL_DEBUG_FILE does not catch it); stdout must carry exactly one JSON line, so file descriptor 1 points
        # at stderr while the communicator is created.
        with _stdout_to_stderr():
            if backend == "nccl":
                try:
                    dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
                except TypeError:                                    # older torch: no device_id argument
                    dist.init_process_group("nccl")
            else:
                dist.init_process_group(backend)
            warm = torch.zeros(1, device=coll_dev)
            dist.all_reduce(warm)                                    # forces communicator creation inside the redirect
            if coll_dev.type == "cuda":
                torch.cuda.synchronize()

    pkg = g.load_package()
    ctx = pkg.Context(local_rank)                        # raises without a gfx950 GPU: no fallback
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    n = args.states
    gen = torch.Generator(device=dev).manual_seed(0xC0DE + rank)
    x = torch.randint(0, 256, (n, 96), dtype=torch.uint8, device=dev, generator=gen)
    for off in (31, 63, 95):
        x[:, off] &= 0x1F                                # < 2^253 < r: canonical, effectively uniform
    y = torch.empty_like(x)

    def step():
        ctx.permute_batch_dev(x.data_ptr(), y.data_ptr(), n)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for i in range(args.steps):
        step()
        ev[i + 1].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]

    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    total_perms = n * args.steps * world
    value = total_perms / elapsed

    # ---- sanity: the result of the timed kernel is the reference permutation (a few states, vs the oracle).
    # Rank 0 only: the oracle is a checker built on demand, N ranks must not race its build.
    import numpy as np
    C = None
    if rank == 0:
        C, P = g.load_oracle()
        idx = torch.tensor([0, 1, n // 2, n - 1], device=dev)
        assert np.array_equal(y[idx].cpu().numpy(), C.permute_batch(x[idx].cpu().numpy())), "bench output != oracle"
    if world > 1:
        dist.barrier()

    avg_ms = sum(kernel_ms) / len(kernel_ms)
    achieved = BYTES_PER_PERM * n / (avg_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 correction + WRITE_SIZE, separate rocprofv3
    # passes; summary committed under profiles/).  It is a property of the 2^24-state launch only.
    traffic, valu = None, None
    tpath = os.path.join(ROOT, "profiles", "r01_permute_batch_traffic.json")
    if os.path.exists(tpath) and n == N_STATES:
        try:
            prof = json.load(open(tpath))
            traffic = prof.get("hbm_bytes_per_launch")
            ipw = prof.get("valu_insts_per_wave")
            if ipw:
                # the real ceiling: VALU issue.  Measured (tools/ubench_valu.hip): the VOP3 integer ops this kernel
                # is made of (v_mad_u64_u32 and friends) issue once per 4 cycles per SIMD; 1024 SIMDs at 2.4 GHz.
                wave_insts = ipw * n / 64
                peak = 1024 * 2.4e9 / 4
                valu = {"bound": "valu-issue", "achieved": wave_insts / (avg_ms * 1e-3), "peak": peak,
                        "unit": "wave-instructions/s", "frac": round(min(1.0, wave_insts / (avg_ms * 1e-3) / peak), 4),
                        "valu_insts_per_permutation": ipw, "pmc_valu_busy_frac": prof.get("valu_busy_frac"),
                        # integer multiplies/s over the microbenchmarked peak.  33 120 v_mad_u64_u32 per
                        # permutation (80 S-boxes x 414); tools/ubench_valu.hip: 4.56 cycles per wave-instruction per SIMD
                        "mad_u64_u32_per_s": 33120 * n / (avg_ms * 1e-3),
                        "mad_u64_u32_frac_of_ubench_peak": round(33120 * (n / 64) / (avg_ms * 1e-3) / (1024 * 2.4e9 / 4.56), 4),
                        "note": "peak = 1 instruction / 4 cycles / SIMD (the VOP3 integer class, 3/4 of the mix; the rest are "
                                "2-cycle VOP2 ops, so the raw ratio can exceed 1): the issue port is saturated",
                        "source": "SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE, profiles/r01_permute_batch_traffic.json"}
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "kernel": "k_permute_batch", "avg_launch_ms": round(avg_ms, 4),
                "algorithmic_bytes_per_launch": BYTES_PER_PERM * n,
                "note": "VALU-integer bound by construction (about 5.5e4 VALU instructions per permutation against 192 B); see DESIGN.md"}

    out = {
        "metric": "Poseidon2-BN254 permutations/sec per GPU; full proof-input witnesses/sec",
        "value": value, "unit": "permutations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32x9 (29-bit limbs, Montgomery mod BN254 r)", "data": "synthetic",
        "config": {"workload": "configs[1]: batched Poseidon2 t=3 permutation, 2^%d random canonical Fr states per GPU per step, bit-exact vs oracle"
                               % (n.bit_length() - 1), "states_per_gpu": n, "parallelism": "independent shards, %d rank(s)" % world},
        "per_gpu_value": value / world, "collective_backend": (backend if world > 1 else None),
        "roofline": roofline,
    }
    if valu:
        out["valu_roofline"] = valu

    # ---- extra legs (outside the timed region) -------------------------------------------------------
    extra = {}
    if not args.no_extra:
        try:
            extra.update(slot_root_leg(torch, ctx, pkg, dev, stream))
        except Exception as e:   # never lose the headline line to an extra leg
            extra["slot_root_error"] = repr(e)
        if world == 1:
            try:
                extra.update(witness_leg(torch, ctx, pkg))
            except Exception as e:
                extra["witness_error"] = repr(e)
        if world > 1:
            try:
                extra.update(dataset_leg(torch, dist, ctx, pkg, coll_dev, rank, world))
            except Exception as e:
                extra["dataset_error"] = repr(e)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(C, np)
    if extra:
        out["extra"] = extra
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def slot_root_leg(torch, ctx, pkg, dev, stream):
    """Config 3: one 8 GiB fake slot resident in HBM -> cell hashes (34 perms/cell) -> block + slot trees."""
    n_cells, cs, bs = 1 << 22, 2048, 65536
    buf = torch.empty((n_cells, cs), dtype=torch.uint8, device=dev)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record(stream)
    ctx.gen_fake_cells_dev(ctx.slot_seed(12345, 0), 0, n_cells, cs, buf.data_ptr())
    e[1].record(stream)
    trees = ctx.slot_trees_dev(buf.data_ptr(), 1, cs, bs, n_cells)     # warm-up pass
    torch.cuda.synchronize()
    trees.free()
    e[2].record(stream)
    trees = ctx.slot_trees_dev(buf.data_ptr(), 1, cs, bs, n_cells)
    e[3].record(stream)
    torch.cuda.synchronize()
    root = trees.roots()[0]
    gen_ms, build_ms = e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])
    perms = 35 * n_cells - 1
    alg_bytes = n_cells * cs + 2 * 32 * n_cells      # cells read once, leaf layer written and read back
    del buf
    return {"slot_root": {"workload": "configs[2]: cellSize=2048, nCells=2^22 (8 GiB) sponge+tree, 1 GPU",
                          "build_ms": round(build_ms, 2), "perms": perms, "perms_per_s": perms / (build_ms * 1e-3),
                          "algorithmic_GBps": round(alg_bytes / (build_ms * 1e-3) / 1e9, 2),
                          "fake_data_gen_ms": round(gen_ms, 2), "slot_root_hex": root.tobytes()[::-1].hex()}}


def witness_leg(torch, ctx, pkg):
    """Config 4 (the metric's second half): nSamples=100, maxDepth=32, 4096 slots batched on one GPU.
    4096 x 8 GiB does not fit HBM, so nCells = 2^12 per slot (8 MiB), 32 GiB of fake data
    generated and hashed on the device; one witness = one SlotProofInput serialised as input.json."""
    n_slots, n_cells = 4096, 1 << 12
    cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=12, cellSize=2048, blockSize=65536, nSlots=n_slots, nCells=n_cells,
                          nSamples=100, seed=12345)
    ctx.reset_stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ds = ctx.dataset(cfg)                 # every slot tree, built once (sync on return)
    ds.set_roots(None)                    # dataset tree over the 4096 slot roots
    t1 = time.perf_counter()
    pis = ds.proof_inputs(list(range(n_slots)), 1234567)
    t2 = time.perf_counter()
    threads = 16
    try:
        threads = max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        pass
    nbytes = pkg.write_json_batch(ctx, pis, None, threads=threads)
    t3 = time.perf_counter()
    for p in pis:
        p.free()
    # the same two stages overlapped (GPU batch k+1 while the host serialises batch k)
    t4 = time.perf_counter()
    nbytes2 = ds.export_proof_inputs(list(range(n_slots)), 1234567, None, threads=threads, batch=512)
    t5 = time.perf_counter()
    assert nbytes2 == nbytes
    perms = n_slots * (35 * n_cells - 1) + (n_slots - 1) + 200 * n_slots
    root_hex = ds.root().tobytes()[::-1].hex()
    ds.free()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    return {"witnesses": {"workload": "configs[3]: nSamples=100, maxDepth=32, 4096 slots x 2^12 cells (32 GiB fake data) batched, 1 GPU",
                          "build_trees_s": round(t1 - t0, 3), "generate_4096_proof_inputs_s": round(t2 - t1, 3),
                          "json_serialise_s": round(t3 - t2, 3), "json_threads": threads, "json_bytes": nbytes,
                          "pipelined_generate_and_json_s": round(t5 - t4, 3),
                          "witnesses_per_s_without_json": n_slots / (t2 - t0),
                          "witnesses_per_s_with_json": n_slots / ((t1 - t0) + (t5 - t4)),
                          "witnesses_per_s_with_json_unpipelined": n_slots / (t3 - t0),
                          "perms": perms, "perms_per_s_build": (perms - 200 * n_slots) / (t1 - t0),
                          "dataset_root_hex": root_hex}}


def dataset_leg(torch, dist, ctx, pkg, dev, rank, world):
    """Config 5's shape at a scale that finishes in a second: 8 GiB slots (cellSize 2048, nCells 2^22) sharded
    over the ranks, two per GPU; each rank builds its slot trees with no communication, ONE all-gather of the
    32-byte slot roots (RCCL over xGMI), the dataset tree on every rank; every rank must get the same root."""
    import importlib
    d = importlib.import_module("codex_storage_proofs_circuits_amd.distributed")
    per_rank, n_cells = 2, 1 << 22
    n_slots = per_rank * world
    cfg = pkg.make_config(maxDepth=32, maxLog2NSlots=max(1, (n_slots - 1).bit_length()), cellSize=2048, blockSize=65536,
                          nSlots=n_slots, nCells=n_cells, nSamples=100, seed=12345)
    ctx.reset_stream()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    backend = d.HipBackend(pkg, ctx)
    root, all_roots, (first, count) = d.dataset_root_sharded(backend, cfg, rank, world, dist, dev)
    pi = backend.dataset.proof_input(first, 1234567)          # a proof input for one of this rank's own slots
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    n_json = len(pi.json())
    r = torch.from_numpy(root.copy()).to(dev)
    rs = [torch.empty_like(r) for _ in range(world)]
    dist.all_gather(rs, r)
    same = all(torch.equal(rs[0], q) for q in rs)
    perms = n_slots * (35 * n_cells - 1) + n_slots - 1
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    return {"dataset": {"workload": "configs[4] shape: %d slots of 8 GiB (2 per GPU) sharded over %d GPUs, one all-gather of slot roots -> "
                                    "dataset root, one proof input per rank" % (n_slots, world),
                        "seconds": round(dt, 4), "perms_per_s": perms / dt, "all_ranks_agree": bool(same),
                        "proof_input_json_bytes": n_json, "dataset_root_hex": root.tobytes()[::-1].hex()}}


def cpu_baseline(C, np):
    """The oracle timed on this box's host cores on a bounded sample of the same workload (config 2 shape)."""
    # a one-GPU box's CPU share is 16 cores even though more are visible; never oversubscribe it
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    rng = np.random.default_rng(0xC0DE)
    n1 = 1 << 17
    x = rng.integers(0, 256, size=(n1, 96), dtype=np.uint8)
    x[:, 31] &= 0x1F
    x[:, 63] &= 0x1F
    x[:, 95] &= 0x1F
    t = time.perf_counter()
    C.permute_batch(x, threads=1)
    single = n1 / (time.perf_counter() - t)
    nm = min(1 << 21, n1 * 2 * cores)
    xm = np.tile(x, (nm // n1, 1))
    t = time.perf_counter()
    C.permute_batch(xm, threads=cores)
    multi = nm / (time.perf_counter() - t)
    return {"value": multi, "unit": "permutations/s", "cores": cores, "kind": "port",
            "sample": "C oracle (oracle/p2_oracle.c, 4x64-bit Montgomery; NOT the Nim reference binary, which cannot be built here): "
                      "%d states on %d threads; single-thread rate on %d states reported beside it" % (nm, cores, n1),
            "single_thread_value": single}


if __name__ == "__main__":
    main()
