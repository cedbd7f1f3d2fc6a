"""`python bench.py --gpus N` typed as is (no WORLD_SIZE in the environment): one worker process per GPU, started here."""
import json
import os
import sys
import time

from .common import emit, error_line


def launch(args, argv):
    """`python bench.py --gpus N` typed as is (no WORLD_SIZE in the environment): start the N workers ourselves — one process
    per GPU, rendezvous on 127.0.0.1 at a free port, the same environment torch.distributed.run would give them — forward
    rank 0's JSON line, and return the worst exit code.  Whatever happens ONE JSON line is printed: a rank that fails or
    hangs is named together with the phase it was in (the workers keep that in a status file), the others are ended.  A SIGTERM or
    SIGINT sent to this launcher — a driver's time limit, ctrl-C — and the launcher's own time limit are passed to rank 0 FIRST: it
    prints the complete measurement it holds, marked "incomplete" (guards.Fallback), and that line is this run's result."""
    import shutil
    import signal
    import socket
    import subprocess
    import tempfile
    import threading
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    status_dir = tempfile.mkdtemp(prefix="hnh_bench_")
    worker = os.environ.get("HNH_BENCH_WORKER") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")  # tests substitute a worker with a CPU transport
    procs, lines, pumps = [], [], []

    def pump(stream, rank):
        for ln in stream:
            if rank == 0 and ln.lstrip().startswith("{") and '"metric"' in ln:
                lines.append(ln.strip())
            else:
                sys.stderr.write(ln if rank == 0 else "[rank %d] %s" % (rank, ln))
                sys.stderr.flush()

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HNH_BENCH_STATUS_DIR=status_dir)
        p = subprocess.Popen([sys.executable, worker] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        procs.append(p)
        t = threading.Thread(target=pump, args=(p.stdout, r), daemon=True)
        t.start()
        pumps.append(t)

    def phase_of(r):
        try:
            with open(os.path.join(status_dir, "rank%d.phase" % r)) as f:
                return f.read().strip() or "start-up"
        except OSError:
            return "start-up (before the benchmark body)"

    ended_by = []  # the signal this launcher was sent, if any

    def on_signal(signum, frame):
        ended_by.append(signum)
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sg, on_signal)
        except (ValueError, OSError):  # (not the main thread: tests that call launch() in-process)
            pass

    def ask_rank0_for_its_line():
        """rank 0 prints the line in hand on SIGTERM and exits 0 (143 when it holds none); the others are ended afterwards"""
        if procs[0].poll() is None:
            procs[0].terminate()
            try:
                procs[0].wait(timeout=20)
            except subprocess.TimeoutExpired:
                pass

    deadline = time.monotonic() + args.launch_timeout
    first_bad, grace, timed_out, asked = None, None, False, None
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        if (ended_by or now > deadline) and first_bad is None and asked is None:
            asked = ("signal %d sent to the launcher" % ended_by[0]) if ended_by else ("the launcher's time limit of %.0f s" % args.launch_timeout)
            ask_rank0_for_its_line()
            grace = time.monotonic()  # ... and now end the others
            now = grace + 1.0
        for r, p in enumerate(procs):
            if first_bad is None and asked is None and p.poll() not in (None, 0):
                first_bad, grace = (r, p.returncode, phase_of(r)), now + 20.0  # the others get a moment to report, then are ended
        if (grace is not None and now > grace) or now > deadline:
            timed_out = now > deadline and first_bad is None and asked is None
            for p in procs:  # exactly the processes started above
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    for t in pumps:
        t.join(timeout=5)
    codes = [p.returncode for p in procs]
    phases = {str(r): phase_of(r) for r in range(n)}
    if asked is not None:  # ended from outside (or by the time limit): rank 0's line in hand is the result
        for ln in reversed(lines):
            try:
                got = json.loads(ln)
            except ValueError:
                continue
            if got.get("value") is not None:
                got["incomplete"] = (got.get("incomplete", "") + "; " if got.get("incomplete") else "") + "run ended by " + asked
                got["exit_codes"], got["phases"] = codes, phases
                shutil.rmtree(status_dir, ignore_errors=True)
                emit(got)
                return 0
        shutil.rmtree(status_dir, ignore_errors=True)
        msg = ("no result after %.0f s: the launcher ended its workers" % args.launch_timeout) if not ended_by else \
            ("run ended by %s before a complete measurement was in hand" % asked)
        emit(error_line(args, msg, failed_rank=None, phases=phases, exit_codes=codes))
        return max((abs(c) if c is not None else 1) for c in codes) or 1
    if first_bad is None and not timed_out:  # everybody had ended between two polls
        bad = [r for r, c in enumerate(codes) if c != 0]
        if bad:
            first_bad = (bad[0], codes[bad[0]], phases[str(bad[0])])
    shutil.rmtree(status_dir, ignore_errors=True)
    worst = max((abs(c) if c is not None else 1) for c in codes)
    if timed_out:
        emit(error_line(args, "no result after %.0f s: the launcher ended its workers" % args.launch_timeout,
                        failed_rank=None, phases=phases, exit_codes=codes))
        return worst or 1
    if first_bad is None and len(lines) == 1 and worst == 0:
        emit(lines[0])
        return 0
    if first_bad is not None:
        r, code, ph = first_bad
        # a rank that gave up AFTER rank 0 held a complete measurement: rank 0 has printed that line, marked "incomplete" — it is the result
        if lines:
            try:
                got = json.loads(lines[-1])
                if got.get("value") is not None and "incomplete" in got:
                    got["incomplete"] += "; rank %d exited with code %s in phase '%s'" % (r, code, ph)
                    got["exit_codes"], got["phases"] = codes, phases
                    emit(got)
                    return 0
            except ValueError:
                pass
        msg = "rank %d exited with code %s in phase '%s'" % (r, code, ph)
        # a failed result check still carries a measured line: keep it, marked
        extra = {"failed_rank": r, "phase": ph, "phases": phases, "exit_codes": codes}
        if lines:
            try:
                extra["line_of_rank0"] = json.loads(lines[-1])
            except ValueError:
                pass
        emit(error_line(args, msg, **extra))
        return worst or 1
    emit(error_line(args, "the workers ended without a result line (%d lines seen)" % len(lines), failed_rank=None,
                    phases=phases, exit_codes=codes))
    return worst or 1
