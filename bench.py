#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: Poseidon2-BN254 permutations/s per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher spawns the N rank processes itself (before anything touches the
GPU) and relays rank 0's JSON line.

Nothing after the timed region can lose the line (bench_orchestration.py; the legs themselves are in bench_legs.py): everything after the headline runs under ONE
deadline (--extra-budget-s, default 240 s: a leg starts only if its worst case fits what is left, else it is recorded as
"skipped: budget"); at N > 1 every wait after the headline is a bounded wait on the rendezvous store and the one data-path
collective of the legs (the gather of slot roots) is bounded too, so a dead or hung rank costs seconds and is named in
extra.rank_failures; a watchdog thread prints the line with whatever has finished when the deadline passes or when the launcher
sends SIGTERM (what torchrun does to the surviving ranks when one dies); the line is printed BEFORE the process group is torn
down.  BENCH_INJECT=rank_exit|rank_hang|gather_error[@rank]|child_hang injects those faults (rehearsals only:
profiles/r05_bench_2rank_inject_*.json).

A "step" is one pass of the hot path over one batch: the batched permutation kernel over 2^24 states
(BASELINE.json configs[1], 1.5 GiB in + 1.5 GiB out, already resident in HBM when the clock starts; every element
uniform in [0, r) by rejection).  With N ranks every rank runs the same-sized batch on its own GPU (weak scaling, no
data-path collective: the permutation has no exchange step); value = all states of all ranks / max-over-ranks time.

The JSON line also carries
  roofline      the dominant kernel (k_permute_batch) against the HBM roof BASELINE.json prescribes:
                achieved = 192 algorithmic B/permutation x 2^24 / average launch time (HIP events on the launch
                stream, measured in this run).  `traffic` is NOT measured in this run: it is the per-launch HBM byte
                count of the committed rocprofv3 --pmc summary named in `traffic_source`.
  roofline_hash_cells  the same block for k_hash_cells over the 8 GiB slot of configs[2] (2^33 B read + 2^27 B of digests written
                per launch), the kernel that holds most of the GPU time of a slot build.
  valu_issue    the instruction-issue view (the binding resource): cycles one SIMD spends per
                wave-permutation from ONE committed rocprofv3 --pmc pass, against the class-weighted issue floor of the kernel's own
                instruction stream (CDNA4 SIMD-32: 2 cycles for the plain VALU forms, 4 for the multiplies and the other half-rate
                forms; tools/valu_roof.py), and the 64-bit multiply rate against the multiplier's measured peak.
                Counters and ISA counts only: no timing of this run mixed in.
  cpu_baseline  the C oracle (a port of the same algorithm, NOT the Nim binary: no Nim toolchain exists)
                timed on this box's host cores on a bounded sample, rank 0 at N=1 only.
  extra         config 3 (8 GiB slot -> slot root), config 4 (4096 slots -> 4096 input.json texts: witnesses/s, classic
                and streamed; `witnesses_from_files`: the same 4096 slots written as real slot files and read back from the
                page cache), ingestion rates against the measured pinned H2D peak, and config 5's shape at SURVEY.md 8(d)'s
                stated scale-down (32 768 slots x 2^12 cells sharded over the ranks, one gather of slot roots, dataset tree
                of 15 levels, one proof input per rank) at every N including 1; the same shape through the C ABI's own
                multi-GPU entry points in ONE process (cp2_multi_*: what the cli twin / a Nim caller gets, `dataset_inprocess`;
                at N > 1 one fresh child process per question, each under its own timeout: the roots exchanged every way the library
                has, each BY NAME and RCCL first -- RCCL all-gather, peer copies, host memory --, the automatic choice, and a dataset
                of few, large slots cut by units: `exchange_every_way`);
                config 5's other stated scale-down, 8 slots at the nominal 8 GiB slot size (`dataset_big_slots`); and the
                drop-in's own workload, the cli twin on workflow/params.sh's defaults as a fresh process, split into HIP init /
                code-object load / hashing / JSON, beside the C oracle on the same configuration (`cli_default`).
"""
import argparse
import contextlib
import glob
import json
import os
import socket
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0             # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_PERM = 192               # 96 B in + 96 B out per permutation (config 2)
N_STATES = 1 << 24                 # BASELINE.json configs[1]
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617



# The orchestration after the headline (Budget, Coord, BoundedDist, Lifeline, LegRunner, spawn_ranks) and the extra legs live beside
# this file; what stays here is the contract above, the timed region and the order of the legs.
from bench_orchestration import (LEG_WORST_S, Budget, BoundedDist, CollectiveTimeout, Coord, LegRunner, Lifeline, _stdout_to_stderr,  # noqa: E402,F401
                                 inject, leg_decision, spawn_ranks)
from bench_legs import (big_slots_leg, cli_default_leg, cpu_baseline, dataset_leg, host_threads, ingest_leg, inprocess_child,  # noqa: E402,F401
                        inprocess_leg, newest_profile, slot_root_leg, uniform_felts_device, witness_leg, witnesses_from_files_leg)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--states", type=int, default=N_STATES, help="states per step per GPU (default 2^24)")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-3 / config-4 / ingest / config-5 extra legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-child-legs", action="store_true",
                    help="skip the legs that start child processes (dataset_inprocess, cli_default): for runs under a counter-collecting profiler")
    ap.add_argument("--legs", default="", help="comma-separated names of the extra legs to run (default: all): " + ", ".join(sorted(LEG_WORST_S)))
    ap.add_argument("--extra-budget-s", type=float, default=240.0,
                    help="ONE deadline for everything after the headline: a leg starts only if its worst case fits what is left; when it "
                         "passes, the line is printed with what is there (default 240)")
    ap.add_argument("--hard-limit-s", type=float, default=560.0,
                    help="no headline after this many seconds (stuck in rendezvous / communicator creation): say where and exit 1")
    ap.add_argument("--inprocess-leg", type=int, default=0, metavar="N",
                    help="(child mode of the dataset_inprocess leg) build config 5's scale-down on N devices in THIS process through cp2_multi_* and print one JSON object")
    ap.add_argument("--inprocess-what", default="main", choices=["main", "rccl", "copy", "host", "few"],
                    help="(child mode) main: first + warm build, automatic exchange; rccl / copy / host: one build with that exchange asked for "
                         "BY NAME; few: a dataset of few, large slots cut by units")
    args = ap.parse_args()

    if args.inprocess_leg:
        return inprocess_child(args.inprocess_leg, args.inprocess_what)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus=%d" % (world, args.gpus))
    life = Lifeline(rank)
    life.arm(args.hard_limit_s)
    life.phase("import torch")

    import torch
    import torch.distributed as dist
    import __graft_entry__ as g

    # Rehearsal knobs (not used by the driver): BENCH_BACKEND=gloo + BENCH_SHARE_GPU=1 run N ranks on ONE GPU with
    # CPU-side collectives, to exercise the multi-rank logic on a one-GPU box (RCCL refuses two ranks per GPU).
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    store = None
    if world > 1:
        # RCCL printf()s a version banner on STDOUT when NCCL_DEBUG is set (the GPU boxes export NCCL_DEBUG=VERSION,
        # and NCCL_DEBUG_FILE does not catch it); stdout must carry exactly one JSON line, so file descriptor 1 points
        # at stderr while the communicator is created.
        import datetime
        kw = {}
        life.phase("rendezvous")
        if os.environ.get("BENCH_INIT_FILE"):                        # self-spawned ranks: file store instead of a TCP port
            # Only the RENDEZVOUS is bounded (a rank that never comes up must not leave the others waiting): every rank
            # announces itself in the store and waits for all the others under BENCH_INIT_TIMEOUT_S.  The process group keeps
            # torch's default collective timeout, so a long leg or a loaded box cannot trip a watchdog later.
            store = dist.FileStore(os.environ["BENCH_INIT_FILE"], world)
            store.set("cp2_bench_up_%d" % rank, "1")
            store.wait(["cp2_bench_up_%d" % r for r in range(world)],
                       datetime.timedelta(seconds=int(os.environ.get("BENCH_INIT_TIMEOUT_S", "120"))))
            kw.update(store=store, rank=rank, world_size=world)
        life.phase("communicator creation (%s)" % backend)
        with _stdout_to_stderr():
            if backend == "nccl":
                try:
                    dist.init_process_group("nccl", device_id=dev, **kw)   # "nccl" is RCCL on ROCm
                except TypeError:                                    # older torch: no device_id argument
                    dist.init_process_group("nccl", **kw)
            else:
                dist.init_process_group(backend, **kw)
            warm = torch.zeros(1, device=coll_dev)
            dist.all_reduce(warm)                                    # forces communicator creation inside the redirect
            if coll_dev.type == "cuda":
                torch.cuda.synchronize()
        if store is None:
            store = dist.distributed_c10d._get_default_store()       # torchrun: the agent's TCPStore (it outlives any rank)
    coord = Coord(store if world > 1 else None, rank, world, sync_timeout_s=float(os.environ.get("BENCH_SYNC_TIMEOUT_S", "30")),
                  dead_dir=os.environ.get("BENCH_DEAD_DIR"))
    life.coord = coord

    life.phase("library + context")
    pkg = g.load_package()
    ctx = pkg.Context(local_rank)                        # raises without a gfx950 GPU: no fallback
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    life.phase("input generation")
    n = args.states
    gen = torch.Generator(device=dev).manual_seed(0xC0DE + rank)
    x = uniform_felts_device(torch, dev, 3 * n, gen).reshape(n, 96)
    y = torch.empty_like(x)

    def step():
        ctx.permute_batch_dev(x.data_ptr(), y.data_ptr(), n)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    life.phase("timed region")
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

    # ---- sanity: the result of the timed kernel is the reference permutation (a strided sample, vs the oracle).
    # Rank 0 only: the oracle is a checker built on demand, N ranks must not race its build.
    life.phase("oracle check of the timed kernel's output")
    import numpy as np
    C = None
    if rank == 0:
        C, P = g.load_oracle()
        idx = torch.cat([torch.tensor([0, 1, n // 2, n - 1], device=dev), torch.arange(0, n, max(1, n // 2048), device=dev)])
        assert np.array_equal(y[idx].cpu().numpy(), C.permute_batch(x[idx].cpu().numpy(), threads=4)), "bench output != oracle"

    avg_ms = sum(kernel_ms) / len(kernel_ms)
    achieved = BYTES_PER_PERM * n / (avg_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 correction + WRITE_SIZE, separate rocprofv3
    # passes; summary committed under profiles/).  A property of the 2^24-state launch only; not measured in this run.
    traffic, traffic_source, valu = None, None, None
    tpath = newest_profile("r*_permute_batch_traffic.json")
    if tpath and n == N_STATES:
        try:
            prof = json.load(open(tpath))
            traffic = prof.get("hbm_bytes_per_launch")
            traffic_source = "%s (rocprofv3 --pmc, separate passes of this command; not measured in this run)" % os.path.relpath(tpath, ROOT)
            valu = prof.get("valu_issue")          # one PMC pass, counters only (tools/profile_summarize.py)
            if valu:
                valu = dict(valu, source=os.path.relpath(tpath, ROOT),
                            note="every figure is from the one --pmc pass named in `pass`, a property of the kernel and the chip: "
                                 "nothing of this run's timing or clock is mixed in")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_source,
                "kernel": "k_permute_batch", "avg_launch_ms": round(avg_ms, 4),
                "launch_ms_min_max": [round(min(kernel_ms), 4), round(max(kernel_ms), 4)],
                "algorithmic_bytes_per_launch": BYTES_PER_PERM * n,
                "note": "VALU-issue bound by construction (about 5.1e4 VALU instructions per permutation against 192 B): see valu_issue and DESIGN.md section 5"}

    out = {
        "metric": "Poseidon2-BN254 permutations/sec per GPU; full proof-input witnesses/sec",
        "value": value, "unit": "permutations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32x9 (29-bit limbs, Montgomery mod BN254 r)", "data": "synthetic",
        "config": {"workload": "configs[1]: batched Poseidon2 t=3 permutation, 2^%d Fr states per GPU per step, every element uniform in "
                               "[0, r) by rejection, bit-exact vs oracle" % (n.bit_length() - 1),
                   "states_per_gpu": n, "parallelism": "independent shards, %d rank(s)" % world},
        "per_gpu_value": value / world, "collective_backend": (backend if world > 1 else None),
        "roofline": roofline,
    }
    if valu:
        out["valu_issue"] = valu

    # ---- the headline exists: from here on the line cannot be lost -----------------------------------------
    budget = Budget(args.extra_budget_s)
    life.headline(out, args.extra_budget_s + 10.0)           # the watchdog's deadline: the budget plus time to print
    if rank == 0 and os.environ.get("BENCH_HEADLINE_FLAG"):
        open(os.environ["BENCH_HEADLINE_FLAG"], "w").close()  # (self-spawned ranks) tells the parent that rank 0 can finish alone
    del x, y
    torch.cuda.empty_cache()
    def before_collective(what):
        if inject("gather_error"):          # (rehearsal) this rank fails right where the others enter the collective
            raise RuntimeError("injected: gather_error before %s" % what)

    bdist = BoundedDist(dist, coord, timeout_s=float(os.environ.get("BENCH_COLLECTIVE_TIMEOUT_S", "20")), before=before_collective) if world > 1 else None
    legs = LegRunner(coord, budget, life, rank, world, after_leg=torch.cuda.empty_cache)

    only = set(x for x in args.legs.split(",") if x)
    if only - set(LEG_WORST_S):
        raise SystemExit("--legs: unknown leg(s) %s" % sorted(only - set(LEG_WORST_S)))

    def run_leg(name, fn, **kw):
        if only and name not in only:
            return False
        return legs.run(name, fn, LEG_WORST_S[name](world), **kw)

    if not args.no_cpu_baseline and world == 1:
        run_leg("cpu_baseline", lambda: {"cpu_baseline": cpu_baseline(C, np, torch, dev)})
    if not args.no_extra:
        def slot_root():
            leg, hash_roof = slot_root_leg(torch, ctx, pkg, dev, stream)
            return dict(leg, roofline_hash_cells=hash_roof)
        run_leg("slot_root", slot_root, all_ranks=False)
        if world == 1:
            run_leg("witnesses", lambda: witness_leg(torch, ctx, pkg))
            run_leg("ingest", lambda: ingest_leg(torch, ctx, pkg, dev))
            run_leg("witnesses_from_files", lambda: witnesses_from_files_leg(torch, ctx, pkg, dev))
        run_leg("dataset", lambda: dataset_leg(torch, bdist, coord, ctx, pkg, coll_dev, rank, world), collective=world > 1)
        run_leg("dataset_big_slots", lambda: big_slots_leg(torch, bdist, coord, ctx, pkg, coll_dev, rank, world), collective=world > 1)
        if not args.no_child_legs:
            run_leg("dataset_inprocess", lambda: inprocess_leg(torch, coord, budget, ctx, rank, world))
            if world == 1:
                run_leg("cli_default", lambda: cli_default_leg(pkg, g))
    life.phase("committed records")
    # config 5 at its nominal slot size (one GPU's share, 32 TiB: 13 minutes, so not run here): the committed record, named as such
    rec = newest_profile("r*_config5_nominal_share.txt")
    if rec and rank == 0:
        try:
            last = [l for l in open(rec).read().splitlines() if l.startswith("{")][-1]
            life.record({"config5_nominal_share_record": dict(json.loads(last), source="%s (tools/config5_share.py; NOT measured in this run)" % os.path.relpath(rec, ROOT),
                                                              workload="configs[4] at its nominal slot size, one GPU's share: 4096 slots x 2^22 cells x 2048 B = 32 TiB; what the build kept of the trees is the record's tree_mode (2 = compact, 0 = roots only)")})
        except Exception:
            pass
    rec = newest_profile("r*_config4_nominal.txt")
    if rec and rank == 0:
        try:
            last = [l for l in open(rec).read().splitlines() if l.startswith("{")][-1]
            life.record({"config4_nominal_record": dict(json.loads(last), source="%s (tools/config4_nominal.py; NOT measured in this run)" % os.path.relpath(rec, ROOT),
                                                        workload="configs[3] at configs[2]'s slot size: 4096 slots x 8 GiB, nSamples=100, maxDepth=32, every input.json, streamed build; what it kept of the trees is the record's tree_mode (2 = compact, 0 = roots only)")})
        except Exception:
            pass
    if legs.decisions:
        life.record({"legs": legs.decisions, "extra_budget_s": args.extra_budget_s, "extra_seconds_used": round(budget.elapsed(), 1)})
    # ---- the line first, the teardown after it (destroy_process_group must not be able to block the print) -------------
    life.phase("print")
    text = life.finish()
    if text:
        print(text, flush=True)
    if world > 1:
        life.phase("teardown")
        clean = not coord.collectives_broken and not coord.failures
        if clean:
            import threading
            th = threading.Thread(target=lambda: dist.destroy_process_group(), daemon=True)
            th.start()
            th.join(15.0)
            clean = not th.is_alive()
        if not clean:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)              # an abandoned communicator (or a dead peer) can block interpreter shutdown: the line is out, leave


if __name__ == "__main__":
    main()
