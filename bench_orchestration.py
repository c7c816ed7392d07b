"""bench.py's orchestration after the headline: ONE deadline for the extra legs (Budget, leg_decision), bounded waits on the
rendezvous store (Coord), a bounded wrapper around the one data-path collective (BoundedDist), the watchdog that guarantees the JSON
line (Lifeline), the leg runner every rank follows (LegRunner) and the launcher-less spawn of N rank processes (spawn_ranks).
No GPU code here: tests/test_bench_orchestration.py drives all of it on the CPU (through `import bench`, which re-exports it)."""
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH_PY = os.path.join(ROOT, "bench.py")           # what spawn_ranks starts as a rank process


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


# ======================================================================================================================
# Orchestration: nothing after the headline may lose the headline.
#
# At N > 1 the legs after the timed region have never run on two real devices (this pipeline's GPU boxes hold one), so the
# first N-GPU run is also their first run.  Everything below exists so that such a run cannot lose its one JSON line and
# names what went wrong instead:
#   Budget     one deadline for everything after the headline (--extra-budget-s): a leg starts only if its worst case fits
#              what is left, else it is recorded as "skipped: budget".
#   Coord      every N > 1 wait after the headline is a BOUNDED wait on the rendezvous store (torchrun's TCPStore, or the file
#              store of self-spawned ranks), never dist.barrier(): a dead or hung rank costs seconds and is named in
#              extra.rank_failures; rank 0 decides which legs run and publishes the decision, so all ranks take the same path.
#   BoundedDist  the data-path collectives of the legs (the gather of slot roots) run async with a bounded wait; one that does
#              not complete is named, and no later collective is attempted on the abandoned communicator.
#   Lifeline   a watchdog thread: when the deadline passes, or SIGTERM arrives (torchrun ends the surviving ranks that way when
#              one rank dies), rank 0 prints the line with what it has -- the headline and every finished leg -- and exits.
#              The main thread may be stuck inside a collective or a HIP call at that moment; the thread is not.
# tests/test_bench_orchestration.py exercises all four on the CPU (gloo, world 2, injected faults).
# ======================================================================================================================
class Budget:
    """One deadline for everything after the headline."""

    def __init__(self, total_s, clock=time.monotonic):
        self.total_s, self.clock, self.t0 = float(total_s), clock, clock()

    def elapsed(self):
        return self.clock() - self.t0

    def remaining(self):
        return max(0.0, self.total_s - self.elapsed())

    def fits(self, worst_s):
        return float(worst_s) <= self.remaining()

    def child_timeout(self, cap_s=120.0, reserve_s=5.0):
        """Timeout for a child process: what is left (minus a reserve for printing the line), never more than `cap_s`."""
        return max(0.0, min(float(cap_s), self.remaining() - reserve_s))


def leg_decision(budget, worst_s, needs_all_ranks, failed_ranks, needs_collective=False, collectives_broken=False):
    """"go", or the reason a leg is skipped -- pure arithmetic (rank 0 evaluates it, every rank follows it)."""
    if needs_all_ranks and failed_ranks:
        return "skipped: rank(s) %s failed earlier" % sorted(failed_ranks)
    if needs_collective and collectives_broken:
        return "skipped: a collective did not complete earlier (the communicator is abandoned)"
    if not budget.fits(worst_s):
        return "skipped: budget (worst case %.0f s, %.0f s left of %.0f)" % (worst_s, budget.remaining(), budget.total_s)
    return "go"


class CollectiveTimeout(RuntimeError):
    pass


class Coord:
    """Bounded rank coordination through the rendezvous store.  world == 1 (store None): everything is immediate."""

    def __init__(self, store, rank, world, sync_timeout_s=30.0, poll_s=0.002, dead_dir=None, clock=time.monotonic, prefix="cp2b"):
        self.store, self.rank, self.world = store, rank, world
        self.sync_timeout_s, self.poll_s, self.dead_dir, self.clock, self.prefix = sync_timeout_s, poll_s, dead_dir, clock, prefix
        self.failures = {}             # rank -> why (first reason wins); a failed rank is never waited for again
        self.collectives_broken = False
        self.bail_name = None          # set by LegRunner while a leg runs: a rank that has posted this (its "leg done", i.e. it left
                                       # the leg early with an error) is not waited for inside the leg -- and is NOT a failed rank
        self._seq = {}

    def _key(self, name, rank=None):
        return "%s/%s" % (self.prefix, name) if rank is None else "%s/%s/%d" % (self.prefix, name, rank)

    def _uniq(self, name):
        """The same name used twice (a leg run in a loop) must not see the previous round's keys."""
        n = self._seq.get(name, 0)
        self._seq[name] = n + 1
        return name if n == 0 else "%s#%d" % (name, n)

    def _poll_dead(self):
        """Self-spawned ranks: the parent drops a marker file when a rank process exits non-zero."""
        if not self.dead_dir:
            return
        for r in range(self.world):
            if r not in self.failures:
                p = os.path.join(self.dead_dir, "dead_%d" % r)
                if os.path.exists(p):
                    try:
                        why = open(p).read().strip() or "exited"
                    except OSError:
                        why = "exited"
                    self.failures[r] = "rank process %s" % why

    def post(self, name, value="ok", rank_key=True):
        if self.store is not None:
            self.store.set(self._key(name, self.rank if rank_key else None), str(value))

    def collect(self, name, timeout_s=None, ranks=None):
        """Wait (bounded) until every rank in `ranks` (default: all) has posted `name`.  Returns ({rank: value}, [missing]).
        Ranks that already failed are not waited for; ranks that do not show up are recorded in self.failures."""
        if self.store is None:
            return {}, []
        timeout_s = self.sync_timeout_s if timeout_s is None else timeout_s
        ranks = list(range(self.world)) if ranks is None else list(ranks)
        got, deadline = {}, self.clock() + timeout_s
        while True:
            self._poll_dead()
            for r in ranks:
                if r not in got and r not in self.failures and self.store.check([self._key(name, r)]):
                    got[r] = self.store.get(self._key(name, r)).decode()
            pending = [r for r in ranks if r not in got and r not in self.failures]
            if self.bail_name and name != self.bail_name:
                pending = [r for r in pending if not self.store.check([self._key(self.bail_name, r)])]
            if not pending:
                break
            if self.clock() >= deadline:
                for r in pending:
                    self.failures[r] = "silent: nothing posted for '%s' within %.0f s" % (name, timeout_s)
                break
            time.sleep(self.poll_s)
        return got, [r for r in ranks if r not in got]

    def decide(self, name, fn, timeout_s=None):
        """Rank 0 waits (bounded) for every live rank to arrive at `name`, evaluates fn() and publishes the result; the other
        ranks wait (bounded) for it.  No word from rank 0 in time: "skipped: no decision from rank 0"."""
        if self.store is None:
            return fn()
        name = self._uniq(name)
        timeout_s = self.sync_timeout_s if timeout_s is None else timeout_s
        self.post(name + "/at")
        if self.rank == 0:
            self.collect(name + "/at", timeout_s)
            d = str(fn())
            self.post(name + "/go", d, rank_key=False)
            return d
        deadline = self.clock() + 2 * timeout_s + 5.0          # rank 0 may itself be waiting `timeout_s` for a silent rank
        key = self._key(name + "/go")
        while not self.store.check([key]):
            self._poll_dead()
            if 0 in self.failures or self.clock() >= deadline:
                self.failures.setdefault(0, "silent: no decision for '%s'" % name)
                return "skipped: no decision from rank 0"
            time.sleep(self.poll_s)
        return self.store.get(key).decode()

    def all_ok(self, name, err=None, timeout_s=None):
        """Every rank says whether its local step worked; raises (on every rank alike) when one did not or stayed silent --
        BEFORE anyone enters the collective that would otherwise wait for it."""
        if self.store is None:
            if err:
                raise err
            return
        self.post(name, "ok" if err is None else "error: %r" % (err,))
        got, missing = self.collect(name, timeout_s)
        bad = {r: v for r, v in got.items() if v != "ok"}
        if bad or missing:
            raise RuntimeError("step '%s': %s" % (name, "; ".join(["rank %d %s" % (r, v) for r, v in sorted(bad.items())] +
                                                                  ["rank %d %s" % (r, self.failures.get(r, "left the leg early")) for r in missing])))

    def exchange(self, name, value, timeout_s=None):
        """Every rank posts a small value; returns {rank: value} of the ranks that did (bounded)."""
        if self.store is None:
            return {0: str(value)}
        self.post(name, value)
        got, _ = self.collect(name, timeout_s)
        return got


class BoundedDist:
    """torch.distributed's collectives with a bounded wait (async_op + polling is_completed): what distributed.py is handed
    instead of the module.  A collective that does not complete raises CollectiveTimeout and marks the communicator abandoned."""

    def __init__(self, dist, coord, timeout_s=20.0, poll_s=0.001, before=None):
        self.dist, self.coord, self.timeout_s, self.poll_s, self.before = dist, coord, timeout_s, poll_s, before

    def get_backend(self):
        return self.dist.get_backend()

    def _wait(self, work, what):
        deadline = time.monotonic() + self.timeout_s
        while not work.is_completed():
            if time.monotonic() >= deadline:
                self.coord.collectives_broken = True
                raise CollectiveTimeout("%s did not complete within %.0f s (a rank never entered it?)" % (what, self.timeout_s))
            time.sleep(self.poll_s)
        work.wait()

    def _run(self, what, fn):
        if self.coord.collectives_broken:
            raise CollectiveTimeout("%s not attempted: an earlier collective did not complete" % what)
        if self.before:
            self.before(what)                                   # fault injection (tests, rehearsals)
        self._wait(fn(), what)

    def all_gather_into_tensor(self, out, inp):
        self._run("all_gather_into_tensor", lambda: self.dist.all_gather_into_tensor(out, inp, async_op=True))

    def all_gather(self, outs, inp):
        self._run("all_gather", lambda: self.dist.all_gather(outs, inp, async_op=True))


class Lifeline:
    """Guarantees the one JSON line.  arm() starts a watchdog thread; it fires when `deadline_s` passes or SIGTERM / SIGINT
    arrives, and then -- on rank 0 -- prints the line built from the headline and whatever legs have finished, names the
    phase the main thread was in, and ends the process.  Other ranks just end.  finish() is the normal way out: the main
    thread takes the line itself and the watchdog stands down."""

    def __init__(self, rank=0, emit=None, exit_fn=os._exit, clock=time.monotonic):
        import threading
        self.rank, self.exit_fn, self.clock = rank, exit_fn, clock
        # the watchdog writes to a duplicate of the ORIGINAL stdout descriptor: immune to a redirect of fd 1 in force at that moment
        # (_stdout_to_stderr) and to whatever holds Python's buffered stdout
        self._out_fd = os.dup(1)
        self.emit = emit or (lambda s: os.write(self._out_fd, (s + "\n").encode()))
        self.lock = threading.Lock()
        self.out, self.extra, self.phase_name, self.t0 = None, {}, "start", clock()
        self.deadline, self.printed, self._thread, self._rfd, self._wfd = None, False, None, None, None
        self.coord = None

    def phase(self, name):
        self.phase_name = name

    def headline(self, out, deadline_s):
        """The headline is computed: from now on a line can always be printed.  deadline_s counts from now."""
        with self.lock:
            self.out = out
            self.deadline = self.clock() + deadline_s

    def record(self, part):
        with self.lock:
            self.extra.update(part)

    def leg_seconds(self, name, seconds):
        with self.lock:
            self.extra.setdefault("leg_seconds", {})[name] = round(seconds, 2)

    def arm(self, hard_limit_s, signals=True):
        """Start the watchdog.  Before headline() the only deadline is `hard_limit_s` (no line exists yet: a diagnostic goes to
        stderr and the exit code is 1)."""
        import signal
        import threading
        self.deadline = self.clock() + hard_limit_s
        self._rfd, self._wfd = os.pipe()
        os.set_blocking(self._wfd, False)
        if signals and threading.current_thread() is threading.main_thread():
            for sig in (signal.SIGTERM, signal.SIGINT):
                signal.signal(sig, lambda *_: None)        # the C-level handler writes the signal number to the wakeup fd
            signal.set_wakeup_fd(self._wfd, warn_on_full_buffer=False)
        self._thread = threading.Thread(target=self._watch, name="bench-lifeline", daemon=True)
        self._thread.start()

    def _watch(self):
        import select
        while True:
            with self.lock:
                left = self.deadline - self.clock()
                done = self.printed
            if done:
                return
            if left <= 0:
                return self._fire("deadline")
            r, _, _ = select.select([self._rfd], [], [], min(left, 1.0))
            if r:
                b = os.read(self._rfd, 64)
                if b == b"q":
                    return
                return self._fire("signal %d" % b[0] if b else "signal")

    def line(self, aborted=None):
        out = dict(self.out)
        extra = dict(self.extra)
        if self.coord is not None and self.coord.failures:
            extra["rank_failures"] = {str(r): w for r, w in sorted(self.coord.failures.items())}
        if aborted:
            extra["bench_aborted"] = aborted
        for k in ("roofline_hash_cells", "cpu_baseline"):        # top-level blocks some legs produce
            if k in extra:
                out[k] = extra.pop(k)
        if extra:
            out["extra"] = extra
        return json.dumps(out, default=str)

    def _fire(self, reason):
        with self.lock:
            if self.printed:
                return
            self.printed = True
            at = round(self.clock() - self.t0, 1)
            if self.rank == 0 and self.out is not None:
                self.emit(self.line({"reason": reason, "phase": self.phase_name, "at_s": at,
                                     "note": "the watchdog printed this line: the main thread was still in `phase`"}))
                code = 0
            else:
                if self.out is None:
                    sys.stderr.write(json.dumps({"bench_error": "no headline", "reason": reason, "phase": self.phase_name, "rank": self.rank,
                                                 "at_s": at}) + "\n")
                    sys.stderr.flush()
                code = 0 if self.out is not None else 1
        self.exit_fn(code)

    def finish(self):
        """Normal completion: returns the line (rank 0) or None; the watchdog stands down.  None when the watchdog got there first."""
        with self.lock:
            if self.printed:
                return None
            self.printed = True
            text = self.line() if (self.rank == 0 and self.out is not None) else ""
        if self._wfd is not None:
            try:
                os.write(self._wfd, b"q")
            except OSError:
                pass
        return text


class LegRunner:
    """Runs the legs after the headline: rank 0 decides (budget, failed ranks, abandoned communicator) and every rank follows the
    same decision; a leg's result or error lands in the line at once (Lifeline.record: the watchdog can print it whenever it has
    to); every rank reports how the leg went through the store (bounded)."""

    def __init__(self, coord, budget, life, rank, world, after_leg=None):
        self.coord, self.budget, self.life, self.rank, self.world, self.after_leg = coord, budget, life, rank, world, after_leg
        self.decisions = {}

    def run(self, name, fn, worst_s, all_ranks=True, collective=False, only_rank0=False):
        coord, life = self.coord, self.life
        life.phase("extra leg: " + name)
        d = coord.decide("leg/" + name, lambda: leg_decision(self.budget, worst_s, all_ranks and self.world > 1, set(coord.failures),
                                                             collective, coord.collectives_broken))
        self.decisions[name] = d
        if d != "go":
            life.record({name + "_skipped": d})
            return False
        t_leg = time.perf_counter()
        err = None
        coord.bail_name = "leg/" + name + "/done"
        try:
            if not only_rank0 or self.rank == 0:
                part = fn()
                if part:
                    life.record(part)
        except Exception as e:   # never lose the headline line to an extra leg
            err = e
            life.record({name + "_error": repr(e)})
        if self.after_leg:
            self.after_leg()
        if self.world > 1:
            got = coord.exchange("leg/" + name + "/done", "ok" if err is None else "error: %r" % (err,))
            coord.bail_name = None
            bad = {r: v for r, v in got.items() if v != "ok"}
            if collective and (bad or len(got) < self.world):
                coord.collectives_broken = True          # a rank may have left a collective half-entered: no further collective
            if self.rank == 0 and bad:
                life.record({name + "_rank_errors": {str(r): v[:300] for r, v in sorted(bad.items())}})
        life.leg_seconds(name, time.perf_counter() - t_leg)
        return err is None


def spawn_ranks(n, argv=None, script=None, extra_env=None):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (nothing in THIS process has touched
    the GPU), relay rank 0's stdout.  No exec: children are ordinary subprocesses.  All children are polled.
    A rank that exits non-zero is named in a marker file the surviving ranks' bounded waits read ("dead_<rank>"); if that
    happens BEFORE rank 0 has its headline (flag file) nothing can be printed and every rank is stopped; after it, rank 0 is
    left to finish on its own deadline (its extra legs skip what needs the dead rank) and the others are stopped once it is out."""
    import shutil
    import tempfile
    # rendezvous through a file store in a private directory: no port is picked here that another process could take
    # before the children bind it (MASTER_ADDR / MASTER_PORT stay set for anything that reads them)
    rdv_dir = tempfile.mkdtemp(prefix="cp2_bench_rdv_")
    rdv, flag = os.path.join(rdv_dir, "store"), os.path.join(rdv_dir, "headline")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    out0 = tempfile.TemporaryFile()
    argv = sys.argv[1:] if argv is None else argv
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_INIT_TIMEOUT_S="120", BENCH_INIT_FILE=rdv,
                   BENCH_DEAD_DIR=rdv_dir, BENCH_HEADLINE_FLAG=flag)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script or BENCH_PY] + list(argv), env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    live = {r: p for r, p in enumerate(procs)}
    stop_all = False
    while 0 in live and not stop_all:
        time.sleep(0.1)
        for r, p in list(live.items()):
            code = p.poll()
            if code is None:
                continue
            del live[r]
            if code != 0:
                with open(os.path.join(rdv_dir, "dead_%d.tmp" % r), "w") as f:
                    f.write("exited with code %d" % code)
                os.replace(os.path.join(rdv_dir, "dead_%d.tmp" % r), os.path.join(rdv_dir, "dead_%d" % r))
                if r == 0 or not os.path.exists(flag):
                    stop_all = True          # no headline yet (or rank 0 itself is gone): nothing left to wait for
    rc0 = procs[0].poll()
    for p in live.values():      # rank 0 is out (or nothing can be printed): stop exactly the processes started above
        p.terminate()
    for p in live.values():
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    if rc0 is None:
        rc0 = procs[0].wait()
    shutil.rmtree(rdv_dir, ignore_errors=True)
    out0.seek(0)
    text = out0.read().decode()
    sys.stdout.write(text)
    sys.stdout.flush()
    has_line = any(l.startswith("{") for l in text.splitlines())
    sys.exit(0 if (rc0 == 0 or has_line) and has_line else 1)


# worst-case seconds of each extra leg (what the budget check uses; measured times are a third of these or less:
# DESIGN.md section 6): a function of the world size where the work is sharded
LEG_WORST_S = {
    "cpu_baseline": lambda w: 40.0,
    "slot_root": lambda w: 15.0,
    "witnesses": lambda w: 40.0,
    "ingest": lambda w: 70.0,
    "witnesses_from_files": lambda w: 60.0,
    "dataset": lambda w: 15.0 + 15.0 / w,
    "dataset_big_slots": lambda w: 15.0 + 10.0 / w,
    "dataset_inprocess": lambda w: 15.0 + 20.0 / w,          # per child process (main / rccl / copy / host / few)
    "cli_default": lambda w: 25.0,
}


def inject(what):
    """BENCH_INJECT=<fault>[@rank] (test / rehearsal only): is fault `what` to be injected in THIS process?"""
    v = os.environ.get("BENCH_INJECT", "")
    if not v:
        return False
    name, _, r = v.partition("@")
    return name == what and int(r or "1") == int(os.environ.get("RANK", "0"))
