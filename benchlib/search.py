"""Several GPUs, 1.5D dense shift: transport, replication factor and route of the moving operand.  The reference takes c on the
command line (bench_erdos_renyi.cpp:23-28) and relays the moving operand round a neighbour ring (one xGMI link per direction); the
default here fetches every block straight from its owner (all links at once) in chunks, with one windowed kernel pass per landed
chunk — how many chunks trades kernel efficiency against fetch/compute overlap, c trades ring traffic against replication traffic,
and the transports differ in who moves the bytes (RCCL channels, copy engines, a pull kernel); all of it depends on the xGMI
bandwidth actually delivered.  Unless flags fix them, the candidates are MEASURED (1 warm-up + 5 calls each, the median, max over
ranks): first the default route on every transport, then replication factors and chunk shapes on the fastest one.  A candidate that
fails is recorded as null with its reason and the search goes on without its transport; a candidate is only started while the
run's time budget (--budget-s) has room for the slowest candidate seen so far."""
import os
import time

from .common import DEFAULT_CHUNKS, STATIC_WINDOWS, route_name


def quick_time(b, calls=5):
    """one warm-up call, then `calls` calls timed one by one (barrier + device synchronise around each, max over ranks): the
    MEDIAN — candidates a per cent apart are within the noise of a mean of three"""
    b.step()
    b.barrier()
    times = []
    for _ in range(calls):
        t0 = time.perf_counter()
        b.step()
        b.barrier()
        times.append(b.max_over_ranks(time.perf_counter() - t0))
    times.sort()
    return times[len(times) // 2] * 1e3


def try_route(b, route, calls=5):
    """quick_time(route) with failure isolation: (ms, None) or (None, reason); a transport on which a candidate failed is
    not used again (its streams may hold half a call)."""
    tr = route[0]
    if b.transports[tr]["dead"] is not None:
        return None, "skipped: " + b.transports[tr]["dead"]
    err = None
    try:
        b.build(route)
        ms = quick_time(b, calls)
    except Exception as e:  # noqa: BLE001 — every failure of a candidate is a recorded null, not the end of the run
        err, ms = "%s: %s" % (type(e).__name__, str(e)[:200]), None
    if not b.all_ok(err is None):
        err = err or "failed on another rank"
        b.transports[tr]["dead"] = "transport %s gave up on %s" % (tr, route_name(route))
        try:
            b.free_current()
        except Exception:  # noqa: BLE001
            b.route = b.op = b.A = b.B = b.S = b.buf = b.als = b.gat = b.gat_x = None
        return None, err
    return ms, None


def candidates(args, n, tr, fixed_mode, default_q):
    """The routes tried on transport `tr`: replication factors that divide n x {chunk shapes of the mesh fetch, relay ring, 15d_fusion1}."""
    cs = [args.c] if args.c else [c for c in (1, 2, 4) if n % c == 0]
    cand = []
    for c in cs:
        if n // c == 1:  # the whole ring is one rank: nothing shifts, the layers only replicate and reduce
            cand.append((tr, c, "none", None))
            continue
        if fixed_mode != "relay":
            # chunk shapes: the library's default, symmetric Q = 2 / 4 (/ 3), and for c = 1 a longer falling shape.  (Q = 8 was a candidate
            # until round 6: it never won a measurement — every window re-streams three dense rows per sparse row — and it was the one
            # candidate in which four processes sharing a GPU hung)
            qs = [str(args.chunks)] if args.chunks else sorted(
                {default_q, DEFAULT_CHUNKS, "2", "4"} | ({"3", "3,4,4,3,2,1,1"} if c == 1 else set()), key=lambda q: (q != default_q, len(q), q))
            cand += [(tr, c, "mesh", q) for q in qs]
            if c == 1 and not args.chunks and os.environ.get("HNH_WINDOW_MERGE") != "0":
                # the default shape with one pass per chunk whatever has landed: adaptive windows (the default) against rounds 2-4's behaviour,
                # measured on the node's own links
                cand.append((tr, c, "mesh", default_q + STATIC_WINDOWS))
        if fixed_mode != "mesh":
            cand.append((tr, c, "relay", None))
        if fixed_mode is None and args.app == "vanilla":
            # the other fusion strategy of the same schedule (replication reuse): twice the gathers, but its stationary operand is
            # replicated once for both kernels; SDDMM over the mesh fetch in row-range passes, SpMM as a mesh reduce-scatter (its own
            # default chunk shape unless --chunks fixes one)
            cand.append((tr, c, "fusion1", str(args.chunks) if args.chunks else None))
    return cand


def tune(b, args, dog, budget, route0, fixed_mode, default_q, reserve_s):
    """Runs the search.  Returns (tuning {route: ms | None}, failures {route: reason}, winner | None, stopped_early reason | None).
    `reserve_s`: what has to stay in the budget after the search (the winner's full measurement)."""
    tuning, failures, slowest, stopped = {}, {}, 0.0, None

    def trial(route):
        nonlocal slowest, stopped
        if stopped is not None:
            return
        # every rank takes the same decision: the budget is compared on rank 0's clock and agreed through all_ok
        if not b.all_ok(budget.fits(max(slowest, 5.0), reserve_s)):
            stopped = "the time budget (--budget-s %.0f) has no room for another candidate (%.0f s left, slowest so far %.0f s, %.0f s kept for the final measurement)" % (
                budget.total, budget.left(), slowest, reserve_s)
            return
        dog.phase("route tuning: " + route_name(route))  # (every candidate has the watchdog's whole allowance)
        t0 = time.monotonic()
        ms, why = try_route(b, route)
        slowest = max(slowest, b.max_over_ranks(time.monotonic() - t0))
        tuning[route] = ms
        if ms is None:
            failures[route] = why

    stage1 = [(tr,) + tuple(route0[1:]) for tr in b.usable()]
    for route in stage1:
        trial(route)
    alive = {k: v for k, v in tuning.items() if v is not None}
    if alive:
        best_tr = min(alive, key=alive.get)[0]
        for route in candidates(args, b.n, best_tr, fixed_mode, default_q):
            if route not in tuning:
                trial(route)
    alive = {k: v for k, v in tuning.items() if v is not None}
    winner = min(alive, key=alive.get) if alive else None  # the same choice on every rank: the times are the all-reduced maxima
    return tuning, failures, winner, stopped
