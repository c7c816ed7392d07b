"""A rank process of bench.py's orchestration with the GPU work replaced by sleeps (tests/test_bench_orchestration.py).

Started by bench.spawn_ranks(script=this file) -- the same environment a real rank gets (RANK, WORLD_SIZE, BENCH_INIT_FILE,
BENCH_DEAD_DIR, BENCH_HEADLINE_FLAG) -- or by torchrun, the way the driver starts bench.py at N > 1.  It takes the real code path after the headline -- Lifeline, Coord over the file store,
Budget, LegRunner, BoundedDist over a gloo process group -- with legs that only sleep, and the faults of FAKE_INJECT:
  rank_exit    rank 1 dies at the start of leg "b"
  rank_hang    rank 1 hangs at the start of leg "b"
  gather_skip  rank 1 raises instead of entering leg "b"'s collective
  main_hang    rank 0's main thread blocks in C (libc sleep) inside leg "c": only the watchdog can print the line
  slow         leg "b" takes longer than the whole budget allows for leg "c"
"""
import ctypes
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    fault = os.environ.get("FAKE_INJECT", "")
    budget_s = float(os.environ.get("FAKE_BUDGET_S", "20"))
    life = bench.Lifeline(rank)
    life.arm(60.0)
    import torch
    import torch.distributed as dist
    if os.environ.get("BENCH_INIT_FILE"):            # started by bench.spawn_ranks
        store = dist.FileStore(os.environ["BENCH_INIT_FILE"], world)
        store.set("cp2_bench_up_%d" % rank, "1")
        store.wait(["cp2_bench_up_%d" % r for r in range(world)], datetime.timedelta(seconds=60))
        dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
    else:                                            # started by torchrun, as the driver starts bench.py: the agent's TCPStore
        dist.init_process_group("gloo")
        store = dist.distributed_c10d._get_default_store()
    coord = bench.Coord(store, rank, world, sync_timeout_s=float(os.environ.get("FAKE_SYNC_S", "3")), dead_dir=os.environ.get("BENCH_DEAD_DIR"))
    life.coord = coord
    dist.barrier()                                   # the timed region's barrier
    out = {"metric": "fake", "value": 1.0, "n_gpus": world}
    budget = bench.Budget(budget_s)
    life.headline(out, budget_s + 2.0)
    if rank == 0 and os.environ.get("BENCH_HEADLINE_FLAG"):
        open(os.environ["BENCH_HEADLINE_FLAG"], "w").close()

    def before(what):
        if fault == "gather_skip" and rank == 1:
            raise RuntimeError("injected before %s" % what)

    bdist = bench.BoundedDist(dist, coord, timeout_s=2.0, before=before)
    legs = bench.LegRunner(coord, budget, life, rank, world)

    def leg_a():
        time.sleep(0.1)
        return {"a": {"rank": rank}}

    def leg_b():
        if rank == 1 and fault == "rank_exit":
            os._exit(3)
        if rank == 1 and fault == "rank_hang":
            time.sleep(1e6)
        if fault == "slow":
            time.sleep(3.0)
        coord.all_ok("b/built", None)
        t = torch.full((4,), float(rank))
        outs = [torch.empty(4) for _ in range(world)]
        bdist.all_gather(outs, t)
        return {"b": {"gathered": [float(o[0]) for o in outs]}}

    def leg_c():
        if rank == 0 and fault == "main_hang":
            while True:
                ctypes.CDLL(None).sleep(1000)        # blocked in C: no Python signal handler, no exception can get us out
        time.sleep(0.1)
        return {"c": {"ok": True}}

    legs.run("a", leg_a, worst_s=1.0, all_ranks=False)
    legs.run("b", leg_b, worst_s=5.0, collective=True)
    legs.run("c", leg_c, worst_s=float(os.environ.get("FAKE_C_WORST_S", "1.0")), all_ranks=False)
    legs.run("d", lambda: {"d": {"ok": True}}, worst_s=1.0, collective=True)
    life.record({"legs": legs.decisions})
    text = life.finish()
    if text:
        print(text, flush=True)
    if coord.collectives_broken or coord.failures:
        sys.stdout.flush()
        os._exit(0)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
