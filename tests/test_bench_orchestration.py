"""CPU suite: bench.py's orchestration after the headline.

The first N-GPU run of bench.py is also the first run of its N > 1 legs on real devices; these tests show -- without a GPU --
that nothing those legs do can lose the headline line: the budget arithmetic, the bounded store coordination (a silent rank, a
dead rank), the bounded collective (a rank that never enters it), the watchdog (main thread stuck in C, SIGTERM from the
launcher), and all of it together through bench.spawn_ranks with the fake rank of tests/bench_fake_rank.py under injected faults.
"""
import json
import os
import signal
import subprocess
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class FakeClock:
    def __init__(self):
        self.t = 100.0

    def __call__(self):
        return self.t


# ---------------------------------------------------------------------------------------------------------------------
# budget arithmetic
# ---------------------------------------------------------------------------------------------------------------------
def test_budget_arithmetic():
    clk = FakeClock()
    b = bench.Budget(240.0, clock=clk)
    assert b.remaining() == 240.0 and b.fits(240.0) and not b.fits(240.1)
    clk.t += 100.0
    assert b.elapsed() == 100.0 and b.remaining() == 140.0
    assert b.child_timeout(120.0) == 120.0                      # capped: the leg needs seconds, 120 is the hang allowance
    clk.t += 100.0
    assert b.child_timeout(120.0) == 35.0                       # what is left minus the reserve for printing the line
    assert b.child_timeout(120.0, reserve_s=0.0) == 40.0
    clk.t += 39.5
    assert b.child_timeout(120.0) == 0.0 and b.remaining() == 0.5
    clk.t += 10.0
    assert b.remaining() == 0.0 and not b.fits(0.1) and b.fits(0.0)


def test_leg_decision_rules():
    clk = FakeClock()
    b = bench.Budget(100.0, clock=clk)
    assert bench.leg_decision(b, 30.0, True, set()) == "go"
    assert bench.leg_decision(b, 30.0, True, {3, 1}).startswith("skipped: rank(s) [1, 3] failed earlier")
    assert bench.leg_decision(b, 30.0, False, {1}) == "go"       # a rank-local leg does not need the failed rank
    assert bench.leg_decision(b, 30.0, False, set(), needs_collective=True, collectives_broken=True).startswith("skipped: a collective")
    clk.t += 80.0
    d = bench.leg_decision(b, 30.0, True, set())
    assert d.startswith("skipped: budget") and "30 s" in d and "20 s left" in d
    assert bench.leg_decision(b, 20.0, True, set()) == "go"


def test_every_leg_has_a_worst_case_and_the_n2_run_fits_300_s():
    """python bench.py --gpus 2: the legs that run at N > 1, each at its worst case, fit the default budget; budget + the
    watchdog's margin + a generous headline phase stay under 300 s."""
    import argparse  # noqa: F401
    w = 2
    n2 = ["slot_root", "dataset", "dataset_big_slots"] + ["dataset_inprocess"] * 5
    assert sum(bench.LEG_WORST_S[k](w) for k in n2) <= 240.0
    assert 240.0 + 10.0 + 45.0 < 300.0
    for k, f in bench.LEG_WORST_S.items():
        assert f(1) >= f(8) > 0, k


# ---------------------------------------------------------------------------------------------------------------------
# Coord over a real FileStore (threads stand in for ranks)
# ---------------------------------------------------------------------------------------------------------------------
def _coords(tmp_path, world, **kw):
    import torch.distributed as dist
    path = str(tmp_path / "store")
    return [bench.Coord(dist.FileStore(path, world), r, world, **kw) for r in range(world)]


def _in_threads(fns):
    res, th = [None] * len(fns), []
    for i, f in enumerate(fns):
        def run(i=i, f=f):
            try:
                res[i] = f()
            except Exception as e:      # noqa: BLE001
                res[i] = e
        th.append(threading.Thread(target=run))
        th[-1].start()
    for t in th:
        t.join(30)
    return res


def test_coord_world_one_is_immediate():
    c = bench.Coord(None, 0, 1)
    assert c.decide("x", lambda: "go") == "go"
    c.all_ok("y")
    with pytest.raises(ValueError):
        c.all_ok("z", ValueError("boom"))
    assert c.exchange("w", "v") == {0: "v"} and c.collect("q") == ({}, [])


def test_coord_decide_and_exchange(tmp_path):
    cs = _coords(tmp_path, 3, sync_timeout_s=5.0)
    res = _in_threads([lambda c=c: (c.decide("leg/a", lambda: "go" if c.rank == 0 else "never evaluated"), c.exchange("r", "v%d" % c.rank)) for c in cs])
    assert all(r[0] == "go" for r in res)
    assert all(r[1] == {0: "v0", 1: "v1", 2: "v2"} for r in res)
    # the same name again (a leg in a loop): a fresh round, not the previous round's keys
    res = _in_threads([lambda c=c: c.decide("leg/a", lambda: "second") for c in cs])
    assert res == ["second"] * 3
    assert not any(c.failures for c in cs)


def test_coord_names_a_silent_rank_within_the_timeout_and_never_waits_for_it_again(tmp_path):
    cs = _coords(tmp_path, 3, sync_timeout_s=0.5)
    t0 = time.monotonic()
    res = _in_threads([lambda c=c: c.decide("leg/b", lambda: "go") for c in cs[:2]])       # rank 2 never arrives
    assert res[0] == "go" and res[1] == "go"
    assert 0.4 < time.monotonic() - t0 < 5.0
    assert 2 in cs[0].failures and "silent" in cs[0].failures[2]
    t0 = time.monotonic()
    got, missing = cs[0].collect("anything", 10.0, ranks=[2])
    assert missing == [2] and time.monotonic() - t0 < 0.5                                   # not waited for again


def test_coord_all_ok_raises_on_every_rank_alike(tmp_path):
    cs = _coords(tmp_path, 2, sync_timeout_s=5.0)
    res = _in_threads([lambda: cs[0].all_ok("built", None), lambda: cs[1].all_ok("built", RuntimeError("hipMalloc failed"))])
    assert all(isinstance(r, RuntimeError) and "rank 1" in str(r) and "hipMalloc" in str(r) for r in res)


def test_coord_reads_the_parents_dead_markers_at_once(tmp_path):
    cs = _coords(tmp_path, 2, sync_timeout_s=30.0, dead_dir=str(tmp_path))
    (tmp_path / "dead_1").write_text("exited with code 3")
    t0 = time.monotonic()
    with pytest.raises(RuntimeError) as e:
        cs[0].all_ok("built", None)
    assert time.monotonic() - t0 < 2.0 and "exited with code 3" in str(e.value)
    assert cs[0].decide("leg/x", lambda: bench.leg_decision(bench.Budget(10), 1.0, True, set(cs[0].failures))).startswith("skipped: rank(s) [1]")


def test_a_rank_that_left_the_leg_with_an_error_is_not_waited_for_and_not_declared_dead(tmp_path):
    cs = _coords(tmp_path, 2, sync_timeout_s=30.0)
    cs[0].bail_name = cs[1].bail_name = "leg/x/done"
    cs[1].post("leg/x/done", "error: RuntimeError('hipMalloc')")         # rank 1 raised inside the leg and went straight to its end
    t0 = time.monotonic()
    got = cs[0].exchange("x/result", "mine")
    assert got == {0: "mine"} and time.monotonic() - t0 < 2.0 and not cs[0].failures
    with pytest.raises(RuntimeError) as e:
        cs[0].all_ok("x/built", None)
    assert "rank 1 left the leg early" in str(e.value) and not cs[0].failures
    assert cs[0].exchange("leg/x/done", "ok") == {0: "ok", 1: "error: RuntimeError('hipMalloc')"}


def test_nonzero_rank_gives_up_on_a_silent_rank_zero(tmp_path):
    cs = _coords(tmp_path, 2, sync_timeout_s=0.2)
    d = cs[1].decide("leg/z", lambda: "go")
    assert d == "skipped: no decision from rank 0" and 0 in cs[1].failures


# ---------------------------------------------------------------------------------------------------------------------
# the watchdog
# ---------------------------------------------------------------------------------------------------------------------
LIFELINE_CHILD = r"""
import ctypes, os, sys, time
sys.path.insert(0, %r)
import bench
life = bench.Lifeline(int(os.environ.get("RANK", "0")))
life.arm(float(os.environ.get("HARD", "30")))
life.phase("rendezvous")
def stuck_in_c():
    while True:                         # (a signal interrupts libc's sleep; a collective that hangs does not return on EINTR)
        ctypes.CDLL(None).sleep(1000)
if os.environ.get("NO_HEADLINE"):
    stuck_in_c()
life.headline({"metric": "m", "value": 7.5}, float(os.environ.get("DEADLINE", "1.0")))
life.record({"slot_root": {"ok": True}, "roofline_hash_cells": {"frac": 0.1}})
life.phase("extra leg: dataset")
print("READY", file=sys.stderr, flush=True)
if os.environ.get("REDIRECT"):
    os.dup2(2, 1)                       # fd 1 points elsewhere when the watchdog fires
if os.environ.get("FINISH"):
    print(life.finish(), flush=True)
    time.sleep(float(os.environ.get("DEADLINE", "1.0")) + 1.0)      # the watchdog must stay quiet now
    sys.exit(0)
stuck_in_c()
""" % ROOT


def _lifeline_child(env, sigterm_after=None):
    p = subprocess.Popen([sys.executable, "-c", LIFELINE_CHILD], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if sigterm_after is not None:
        assert "READY" in p.stderr.readline()
        time.sleep(sigterm_after)
        p.send_signal(signal.SIGTERM)
    so, se = p.communicate(timeout=60)
    return p.returncode, so, se


def test_watchdog_prints_the_line_when_the_deadline_passes_with_the_main_thread_stuck_in_c():
    t0 = time.monotonic()
    rc, so, se = _lifeline_child({"DEADLINE": "1.0"})
    assert rc == 0 and time.monotonic() - t0 < 20
    lines = [l for l in so.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 7.5 and d["roofline_hash_cells"] == {"frac": 0.1} and d["extra"]["slot_root"] == {"ok": True}
    assert d["extra"]["bench_aborted"]["reason"] == "deadline" and d["extra"]["bench_aborted"]["phase"] == "extra leg: dataset"


def test_watchdog_prints_the_line_on_sigterm_even_with_stdout_redirected():
    """torchrun ends the surviving ranks with SIGTERM when one rank dies: rank 0 must still get its line out."""
    rc, so, se = _lifeline_child({"DEADLINE": "60", "REDIRECT": "1"}, sigterm_after=0.3)
    d = json.loads([l for l in so.splitlines() if l.startswith("{")][0])
    assert rc == 0 and d["value"] == 7.5 and d["extra"]["bench_aborted"]["reason"] == "signal %d" % signal.SIGTERM


def test_watchdog_on_other_ranks_only_ends_the_process():
    rc, so, se = _lifeline_child({"DEADLINE": "0.5", "RANK": "1"})
    assert rc == 0 and so.strip() == ""


def test_watchdog_stands_down_after_a_normal_finish():
    rc, so, se = _lifeline_child({"DEADLINE": "0.5", "FINISH": "1"})
    lines = [l for l in so.splitlines() if l.startswith("{")]
    assert rc == 0 and len(lines) == 1 and "bench_aborted" not in lines[0]


def test_no_headline_within_the_hard_limit_says_where_and_fails():
    rc, so, se = _lifeline_child({"HARD": "0.5", "NO_HEADLINE": "1"})
    assert rc == 1 and so.strip() == ""
    d = json.loads([l for l in se.splitlines() if l.startswith("{")][-1])
    assert d["bench_error"] == "no headline" and d["phase"] == "rendezvous"


# ---------------------------------------------------------------------------------------------------------------------
# everything together: spawn_ranks + Lifeline + Coord + LegRunner + BoundedDist (gloo), faults injected
# ---------------------------------------------------------------------------------------------------------------------
SPAWN = r"""
import sys
sys.path.insert(0, %r)
import bench
bench.spawn_ranks(2, argv=[], script=%r)
""" % (ROOT, os.path.join(ROOT, "tests", "bench_fake_rank.py"))


def _spawn(fault, world=2, **env):
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, "-c", SPAWN.replace("spawn_ranks(2,", "spawn_ranks(%d," % world)],
                       env=dict(os.environ, FAKE_INJECT=fault, OMP_NUM_THREADS="1", **{k: str(v) for k, v in env.items()}),
                       capture_output=True, text=True, timeout=120)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout, p.stderr[-2000:])
    return p.returncode, json.loads(lines[0]), time.monotonic() - t0


def test_two_ranks_no_fault():
    rc, d, dt = _spawn("")
    e = d["extra"]
    assert rc == 0 and d["value"] == 1.0 and e["legs"] == {"a": "go", "b": "go", "c": "go", "d": "go"}
    assert e["b"]["gathered"] == [0.0, 1.0] and "rank_failures" not in e and "bench_aborted" not in e


def test_eight_ranks_the_drivers_largest_world():
    """The store coordination at the world size of the driver's last run: eight ranks, clean, and with rank 1 dying inside a leg."""
    rc, d, dt = _spawn("", world=8)
    e = d["extra"]
    assert rc == 0 and d["n_gpus"] == 8 and e["legs"] == {"a": "go", "b": "go", "c": "go", "d": "go"}
    assert e["b"]["gathered"] == [float(r) for r in range(8)] and "rank_failures" not in e
    rc, d, dt = _spawn("rank_exit", world=8, FAKE_SYNC_S=30)
    e = d["extra"]
    assert rc == 0 and dt < 30 and "exited with code 3" in e["rank_failures"]["1"] and sorted(e["b_rank_errors"]) == ["0", "2", "3", "4", "5", "6", "7"]
    assert e["c"] == {"ok": True} and e["legs"]["d"].startswith("skipped: rank(s) [1]")


def test_a_rank_that_dies_is_named_and_costs_seconds():
    rc, d, dt = _spawn("rank_exit", FAKE_SYNC_S=30)
    e = d["extra"]
    assert rc == 0 and dt < 25, dt                                          # far below the 30 s sync timeout: the marker names it at once
    assert "exited with code 3" in e["rank_failures"]["1"]
    assert "b_error" in e and "rank 1" in e["b_error"]
    assert e["legs"]["c"] == "go" and e["c"] == {"ok": True}                # a rank-local leg still runs
    assert e["legs"]["d"].startswith("skipped: rank(s) [1] failed earlier")


def test_a_rank_that_hangs_is_named_after_one_sync_timeout():
    rc, d, dt = _spawn("rank_hang", FAKE_SYNC_S=2, FAKE_BUDGET_S=15)
    e = d["extra"]
    assert rc == 0 and dt < 40
    assert "silent" in e["rank_failures"]["1"] and e["legs"]["d"].startswith("skipped: rank(s) [1]")
    assert e["c"] == {"ok": True} and "bench_aborted" not in e


def test_a_rank_that_never_enters_the_collective_costs_the_collective_timeout_only():
    rc, d, dt = _spawn("gather_skip")
    e = d["extra"]
    assert rc == 0 and dt < 40
    assert "did not complete" in e["b_error"]                                # rank 0's bounded wait
    assert "injected" in e["b_rank_errors"]["1"]                             # ... and what rank 1 said happened
    assert e["legs"]["d"].startswith("skipped: a collective did not complete")
    assert e["c"] == {"ok": True}


def test_rank_zero_stuck_in_c_still_prints_through_the_watchdog():
    rc, d, dt = _spawn("main_hang", FAKE_BUDGET_S=8)
    e = d["extra"]
    assert rc == 0 and dt < 40
    assert e["bench_aborted"]["reason"] == "deadline" and e["bench_aborted"]["phase"] == "extra leg: c"
    assert e["a"] == {"rank": 0} and e["b"]["gathered"] == [0.0, 1.0]        # what had finished is in the line


def test_a_leg_that_does_not_fit_what_is_left_is_skipped_by_every_rank():
    rc, d, dt = _spawn("slow", FAKE_BUDGET_S=6, FAKE_C_WORST_S=5)
    e = d["extra"]
    assert rc == 0 and e["legs"]["c"].startswith("skipped: budget") and e["legs"]["d"] == "go" and e["d"] == {"ok": True}


# ---------------------------------------------------------------------------------------------------------------------
# the driver's launcher: python -m torch.distributed.run.  It ends every surviving rank with SIGTERM when one rank dies.
# ---------------------------------------------------------------------------------------------------------------------
def _torchrun(fault, **env):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "bench_fake_rank.py")],
                       env=dict(os.environ, FAKE_INJECT=fault, OMP_NUM_THREADS="1", **{k: str(v) for k, v in env.items()}),
                       capture_output=True, text=True, timeout=180)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout, p.stderr[-3000:])
    return p.returncode, json.loads(lines[0]), time.monotonic() - t0


def test_under_torchrun_no_fault():
    rc, d, dt = _torchrun("")
    assert rc == 0 and d["extra"]["legs"] == {"a": "go", "b": "go", "c": "go", "d": "go"} and d["extra"]["b"]["gathered"] == [0.0, 1.0]


def test_under_torchrun_a_dead_rank_does_not_cost_the_line():
    """Rank 1 dies inside a leg; torchrun SIGTERMs rank 0 (and would SIGKILL it 30 s later): the line is out before that, it names the
    signal and the phase, and it carries the headline and the leg that had finished."""
    rc, d, dt = _torchrun("rank_exit", FAKE_SYNC_S=30, FAKE_BUDGET_S=60)
    e = d["extra"]
    assert d["value"] == 1.0 and e["a"] == {"rank": 0} and dt < 60
    assert ("bench_aborted" in e and e["bench_aborted"]["reason"] == "signal %d" % signal.SIGTERM) or "1" in e.get("rank_failures", {})


def test_under_torchrun_a_hung_rank_is_named():
    rc, d, dt = _torchrun("rank_hang", FAKE_SYNC_S=2, FAKE_BUDGET_S=15)
    e = d["extra"]
    assert rc == 0 and "silent" in e["rank_failures"]["1"] and e["c"] == {"ok": True} and "bench_aborted" not in e
