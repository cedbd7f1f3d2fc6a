"""What keeps a run from being lost: the line in hand (Fallback), the per-phase watchdog, the global time budget and the
per-phase clock that ends up in the line as `phases_s`."""
import os
import sys
import time

from . import common


class Budget:
    """--budget-s: seconds the whole run may take, counted from process start.  Optional work asks fits(need) first."""

    def __init__(self, seconds):
        self.total = float(seconds)

    def used(self):
        return time.monotonic() - common.T_START

    def left(self):
        return self.total - self.used()

    def fits(self, need, reserve=0.0):
        return self.left() - reserve > need


class Phases:
    """Wall seconds per phase of the run (start-up, transport trials, bring-up, first measurement, tuning, final measurement, ...): where
    the time of a multi-GPU run went is in the line itself."""

    def __init__(self):
        self.seconds, self.current, self.t0 = {}, None, time.monotonic()
        self.seconds["start_up"] = round(self.t0 - common.T_START, 3)  # interpreter, imports, argument parsing

    def start(self, name):
        self.stop()
        self.current, self.t0 = name, time.monotonic()

    def stop(self):
        if self.current is not None:
            self.seconds[self.current] = round(self.seconds.get(self.current, 0.0) + time.monotonic() - self.t0, 3)
            self.current = None

    def snapshot(self):
        out = dict(self.seconds)
        if self.current is not None:
            out[self.current] = round(out.get(self.current, 0.0) + time.monotonic() - self.t0, 3)
        out["total"] = round(time.monotonic() - common.T_START, 3)
        return out


class Fallback:
    """What rank 0 prints if the run cannot finish: the best COMPLETE measurement so far (timed steps + result check of one
    route), marked, or nothing.  A hang inside a transport call cannot be undone from Python, but it need not cost the number
    that is already in hand."""

    def __init__(self, rank, phases=None, no_line=None):
        # no_line(why): what to print when the run is ended before ANY complete measurement exists (an error line: stdout still carries one JSON line)
        self.rank, self.best, self.printed, self.phases, self.no_line = rank, None, False, phases, no_line

    def keep(self, line):
        if self.best is None and self.rank == 0 and os.environ.get("HNH_BENCH_ANNOUNCE"):
            sys.stderr.write("[bench.py] a complete measurement is in hand\n")  # (tests wait for this before they send a signal)
            sys.stderr.flush()
        self.best = line

    def emit_best(self, why):
        """True when a line went out."""
        if self.rank != 0 or self.printed or self.best is None:
            return False
        out = dict(self.best)
        out["incomplete"] = why
        if self.phases is not None:
            out["phases_s"] = self.phases.snapshot()
        common.emit(out)
        self.printed = True
        return True

    def watch_sigterm(self):
        """torch.distributed.run ends the surviving workers with SIGTERM when one of them exits, bench.py's own launcher forwards the
        SIGTERM / SIGINT it is sent to rank 0 first: a thread that sigwait()s for them prints the line in hand even while the main
        thread sits in a C call."""
        import signal
        import threading
        if self.rank != 0 or not hasattr(signal, "pthread_sigmask"):
            return
        try:
            signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM, signal.SIGINT})
        except (ValueError, OSError):
            return

        def wait():
            signal.sigwait({signal.SIGTERM, signal.SIGINT})
            ok = self.emit_best("the run was ended from outside (SIGTERM / SIGINT: the launcher's or the driver's time limit, or another "
                                "rank that failed or hung) before it was over")
            if not ok and not self.printed and self.no_line is not None and "HNH_BENCH_STATUS_DIR" not in os.environ:
                try:  # (a worker of bench.py's own launcher leaves the error line to the launcher, which names the phase of every rank)
                    self.no_line("the run was ended from outside (SIGTERM / SIGINT) before its first complete measurement")
                except Exception:  # noqa: BLE001
                    pass
            os._exit(0 if ok else 143)

        threading.Thread(target=wait, daemon=True).start()


class Watchdog:
    """Per-phase watchdog of a multi-GPU run: a phase that does not finish in time prints rank + phase and ends the
    process (transport problems show up as hangs inside C calls; ctypes releases the GIL there).  If rank 0 already holds a
    complete measurement it prints that line, marked, and exits 0."""

    def __init__(self, rank, seconds, enabled, fallback=None, budget=None):
        self.rank, self.seconds, self.enabled, self.fallback, self.budget = rank, seconds, enabled, fallback, budget
        self.timer = None
        self.name = "start-up"
        # a self-launched run (launch() below) reads these files to say which phase a failed or stuck rank was in
        d = os.environ.get("HNH_BENCH_STATUS_DIR")
        self.status = os.path.join(d, "rank%d.phase" % rank) if d else None
        self.note("start-up")

    def note(self, name):
        self.name = name
        if self.status:
            try:
                with open(self.status, "w") as f:
                    f.write(name)
            except OSError:
                pass

    def phase(self, name, seconds=None):
        import threading
        self.done()
        self.note(name)
        if not self.enabled:
            return
        limit = seconds or self.seconds
        if self.budget is not None:  # no phase's allowance reaches more than five minutes past --budget-s (the driver's limit is 1800 s)
            limit = min(limit, max(60.0, self.budget.left() + 300.0))

        def fire():
            sys.stderr.write("[bench.py watchdog] rank %d stuck in phase '%s' for more than %.0f s - giving up\n" % (self.rank, self.name, limit))
            sys.stderr.flush()
            if self.fallback is not None and self.fallback.emit_best("rank %d was stuck in phase '%s' for more than %.0f s" % (self.rank, self.name, limit)):
                os._exit(0)
            os._exit(3)

        self.timer = threading.Timer(limit, fire)
        self.timer.daemon = True
        self.timer.start()

    def done(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None
