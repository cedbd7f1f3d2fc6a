"""The order of a benchmark run: environment, rendezvous, transports (several GPUs), the default route measured in full — from there
on a complete line is in hand — then whatever the time budget still has room for: other transports, the route search and the
winner's measurement, counter traffic, the secondary workloads, the CPU baseline."""
import json
import os
import sys
import time

from . import baseline, cli, common, guards, launcher, search, secondary, timed, transports
from .common import emit, error_line, route_name


def run(args, make_world=None):
    """`make_world` is replaceable so that tests can drive this exact function over gloo on CPU (None = the product's transports)."""
    product = make_world is None
    if args.gpus > 1 and "HNH_KEEP_OMP" not in os.environ:
        # torch.distributed.run pins OMP_NUM_THREADS=1 per worker; the host-side setup (generator, sorts, CSR build)
        # is OpenMP code, so give every rank its share of the host cores instead (must happen before libgomp starts)
        os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL / mapped peer memory across processes)
    # HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share one serialise: with the
    # framework's own streams (compute, its unmasked twin, communication, the pull's forked streams) and RCCL's in the process, make
    # sure streams that wait for OTHER PROCESSES never share a queue with the streams those processes wait for
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    os.environ.setdefault("HNH_IPC_WAIT_S", "120")
    if args.gpus > 1:
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # one node: RCCL's bootstrap must not depend on an external interface
    if args.ring_mode:
        os.environ["HNH_RING_MODE"] = args.ring_mode
    if args.chunks:
        common.set_chunk_spec(str(args.chunks))
    if args.gpus > 1 and getattr(args, "nchannels", None):
        os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(args.nchannels)
    if getattr(args, "probe_transport", None):
        return transports.probe_main(args)
    for name, default in (("workload", "er"), ("app", "vanilla"), ("transport", "auto"), ("no_secondary", True), ("probe_timeout", 300.0),
                          ("budget_s", 1200.0)):
        if not hasattr(args, name):  # (tests build their own argument namespaces)
            setattr(args, name, default)

    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    phases = guards.Phases()
    budget = guards.Budget(args.budget_s)
    fallback = guards.Fallback(rank, phases, (lambda why: emit(error_line(args, why, failed_rank=None, phases_s=phases.snapshot()))) if rank == 0 else None)
    # SIGTERM / SIGINT are blocked HERE, before torch, OpenMP or HIP start a thread: every thread of the process inherits the mask,
    # so the signal stays pending for the sigwait() thread, which prints the line in hand — whenever it arrives after the first
    # complete measurement — instead of landing on some library thread with the default action
    fallback.watch_sigterm()
    import torch  # first: one HIP runtime per process (see distributed_sddmm_amd/_kernels.py)
    from distributed_sddmm_amd import api as H

    if world_size != n:  # main() self-launches when WORLD_SIZE is absent; this is a launcher that disagrees with --gpus
        raise SystemExit("bench.py --gpus %d was started as rank %d of WORLD_SIZE=%d: the launcher's process count and --gpus disagree" % (n, rank, world_size))
    dist = None
    if n > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: never depend on the container hostname resolving
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo", rank=rank, world_size=n)  # bootstrap + barriers only; data moves over the device transports
    dog = guards.Watchdog(rank, args.watchdog, n > 1, fallback, budget)
    wl = common.Workload(args.workload, args.logm, args.edge_factor)
    b = timed.Bench(args, H, torch, dist, rank, n, dog, wl)
    extra, preflight, probe = {}, None, None

    # ---- transports.  One GPU: none.  Several GPUs through the product path: every wanted transport is tried in a child process
    # first, the usable ones are created here and run their preflight.  Tests substitute their own single transport.
    if product and n > 1:
        phases.start("transport_trials")
        dog.phase("transport creation (device selection)")
        device, ndev = transports.visible_device(rank, n, local_rank)
        assert H.load_backend(None) == transports.PRODUCT_BACKEND
        wanted = {"auto": ["rccl", "ipc", "ipc-kernel"], "rccl": ["rccl"], "ipc": ["ipc", "ipc-kernel"]}[args.transport]
        trials = [w for w in wanted if w != "ipc-kernel"]  # (the two ipc variants share every primitive but the copy)
        args.probe_timeout = min(args.probe_timeout, max(30.0, budget.left() / (2.0 * len(trials) + 1.0)))  # the trials may take a third of the budget at most
        dog.phase("transport trials in child processes (%s)" % ", ".join(wanted), args.probe_timeout * len(trials) + 120.0)
        probe = transports.probe_transports(args, dist, rank, n, trials)
        dog.done()
        if not any(v.startswith("ok") for v in probe.values()):
            # nothing passed its trial: the trial machinery itself (child start-up, rendezvous) may be what failed — try the transports
            # here after all, under the watchdog, rather than give up without a number
            sys.stderr.write("[bench.py] rank %d: no transport passed its child-process trial (%r); trying them in this process\n" % (rank, probe))
            probe = {k: "ok (trial failed: %s; created in the benchmark process)" % v[:120] for k, v in probe.items()}
        later = [name for name in wanted if probe[name if name != "ipc-kernel" else "ipc"].startswith("ok")]
    else:
        phases.start("bring_up")
        dog.phase("transport creation")
        world, device_sync = (transports.gpu_world if product else make_world)(H, dist, rank, n, local_rank)
        b.add_transport("single" if n == 1 else "default", world, device_sync)
        later = []
    dog.done()

    # ---- bringing a transport up in this process: creation, then the preflight — every transport primitive the schedules use, on small
    # buffers with known contents, each under the watchdog — then the order in which the ranks created their communicators is compared.
    # A transport that fails either step ON ANY RANK is left alone (the ranks agree); the others are not affected.
    if n > 1 and not args.no_preflight:
        preflight = {}

    def bring_up(name, create):
        if create:
            dog.phase("transport creation (%s)" % name)
            err = None
            try:
                world = transports.make_gpu_transport(H, dist, rank, n, device, name)
            except Exception as e:  # noqa: BLE001
                err = str(e)[:200]
            if not b.all_ok(err is None):
                probe[name] = "creation failed in the benchmark process: %s" % (err or "on another rank")
                return False
            b.add_transport(name, world, torch.cuda.synchronize)
            dog.done()
        if preflight is None:
            return True
        dog.note("preflight [%s]" % name)
        err = None
        try:
            res = transports.run_preflight(H, b.world(name), 1 << 16, dog)
        except Exception as e:  # noqa: BLE001
            err = str(e)[:200]
        if not b.all_ok(err is None):
            sys.stderr.write("[bench.py preflight] rank %d, transport %s: %s\n" % (rank, name, err or "failed on another rank"))
            b.transports[name]["dead"] = "preflight failed: %s" % (err or "on another rank")
            if probe is not None:
                probe[name] = "preflight failed in the benchmark process: %s" % (err or "on another rank")
            return False
        preflight[name] = res
        sig = [None] * n
        dist.all_gather_object(sig, b.world(name).split_signature())
        if len(set(sig)) != 1:
            sys.stderr.write("[bench.py preflight] ranks created their communicators in different orders: %r\n" % (sig,))
            sys.stderr.flush()
            os._exit(4)
        return True

    # Only ONE transport is brought up before the first measurement: whatever the others do when they are created or run their preflight
    # — fail, or hang until the watchdog ends the run — happens with a complete line in hand.
    phases.start("bring_up")
    if later:
        while later and not bring_up(later.pop(0), True):
            pass
        if not b.usable():
            raise SystemExit("bench.py --gpus %d: no usable device-to-device transport on this node: %r" % (n, probe))
    elif n > 1 and not bring_up(b.usable()[0], False):
        raise SystemExit("bench.py --gpus %d: the transport failed its preflight" % n)

    # ---- the default route, measured in full first: from here on there is a number in hand whatever the search runs into
    phases.start("first_measurement")
    dog.phase("set-up (generator, redistribution, CSR blocks)", max(args.watchdog, 600.0))
    first = b.usable()[0]
    # (what the flags / environment fixed, read before build() starts writing HNH_RING_MODE itself)
    fixed_mode = os.environ.get("HNH_RING_MODE") if (args.ring_mode or "HNH_RING_MODE" in os.environ) else None
    c0 = args.c or 1
    mode0 = "none" if n // c0 == 1 else os.environ.get("HNH_RING_MODE", "mesh")
    default_q = common.current_chunk_spec(args.alg)
    route0 = (first, c0, mode0, default_q if mode0 == "mesh" else None)
    t_first = time.monotonic()
    b.build(route0)
    res = b.measure()
    first_s = b.max_over_ranks(time.monotonic() - t_first)  # what a full measurement costs on this node (set-up included)
    tuning_failures, stopped_early = {}, []

    def finish_line(res, tuning):
        ex = dict(extra)
        if preflight is not None:
            ex["preflight"] = {"primitives_ok": sorted(next(iter(preflight.values()))) if preflight else [], "transports": sorted(preflight),
                               "communicator_split_order": "identical on all ranks"}
        line = timed.compose_line(args, b, res, ex)
        if probe is not None:
            line["config"]["transport_trials"] = probe
        if tuning is not None:
            line["config"]["route_tuning_ms_per_step"] = {route_name(k): (round(v, 4) if v is not None else None) for k, v in tuning.items()}
            if tuning_failures:
                line["config"]["route_tuning_failures"] = {route_name(k): v for k, v in tuning_failures.items()}
        if stopped_early:
            line["config"]["budget_stops"] = list(stopped_early)
        line["config"]["budget_s"] = args.budget_s
        line["phases_s"] = phases.snapshot()
        return line

    if rank == 0:
        fallback.keep(finish_line(res, None))
    phases.start("bring_up_other_transports")
    for name in later:  # the remaining transports, with that line in hand — each only while the budget has room for a hang's worth of it
        if not b.all_ok(budget.fits(args.watchdog + 150.0, reserve=first_s)):
            stopped_early.append("transport %s was not brought up: %.0f s left of --budget-s" % (name, budget.left()))
            continue
        bring_up(name, True)
        if rank == 0:
            fallback.keep(finish_line(res, None))

    # ---- several GPUs, 1.5D dense shift: the search over transports, replication factors and routes (search.py), then the winner in full
    tuning = None
    if n > 1 and args.alg == "15d_fusion2" and not args.no_tune:
        total = len(b.usable()) + len(search.candidates(args, n, first, fixed_mode, default_q)) - 1
        if total > 1:
            phases.start("tuning")
            dog.phase("route tuning (transports, replication factor, mesh chunk shapes, relay ring)", max(args.watchdog, 900.0))
            reserve = 1.5 * first_s + 30.0  # the winner's full measurement has to fit behind the search
            tuning, tuning_failures, winner, stopped = search.tune(b, args, dog, budget, route0, fixed_mode, default_q, reserve)
            if stopped:
                stopped_early.append(stopped)
            if rank == 0:
                fallback.keep(finish_line(res, tuning))
            if winner is not None and winner != res["route"] and b.all_ok(budget.fits(1.2 * first_s)):
                phases.start("final_measurement")
                dog.phase("final measurement of the fastest route", max(args.watchdog, 600.0))
                err, res2 = None, None
                try:
                    b.build(winner)
                    res2 = b.measure()
                except Exception as e:  # noqa: BLE001
                    err = "%s: %s" % (type(e).__name__, str(e)[:200])
                if b.all_ok(err is None):
                    if res2["elapsed"] <= res["elapsed"] or (res["check"] and not res["check"].get("ok", True)):
                        res = res2
                else:
                    tuning_failures[winner] = "final measurement: " + (err or "failed on another rank")
                    b.route = None  # (whatever is left of it is not used again)
            elif winner is not None and winner != res["route"]:
                stopped_early.append("the fastest candidate (%s) was not measured in full: %.0f s left of --budget-s" % (route_name(winner), budget.left()))

    out = None
    if rank == 0:
        phases.start("counter_traffic")
        out = finish_line(res, tuning)
        dur = out["roofline"]["avg_launch_ms"] * 1e-3
        traffic, traffic_source, live = None, None, None
        if n == 1 and not args.no_live_traffic and H.backend_name() == "hip-gfx950":
            dog.note("live counter passes")
            live = baseline.live_traffic(args)
        if live is not None:
            traffic = live["bytes_per_launch"]
            # two rocprofv3 --pmc passes of this command run by this process, outside the timed region; 2 x FETCH_SIZE + WRITE_SIZE
            # (the micro-architecture guide's gfx950 correction)
            traffic_source = "live rocprofv3 --pmc passes: 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction)"
            out["roofline"]["traffic_detail"] = {"launches_sampled": live["launches_sampled"], "seconds": live["seconds"],
                                                 "fetch_size_kb_raw": live["fetch_size_kb_raw"], "write_size_kb_raw": live["write_size_kb_raw"]}
        else:
            tf = os.path.join(common.ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tf):
                try:
                    with open(tf) as f:
                        rec = json.load(f)
                    if args.workload == "er" and args.app == "vanilla" and rec.get("workload_key") == "er%d_ef%d_r%d_n%d" % (args.logm, args.edge_factor, args.r, n):
                        traffic = rec.get("bytes_per_launch")
                        traffic_source = "profiles/hbm_traffic.json (static: counter passes of an earlier run, not live)"
                except Exception:
                    traffic = None
        out["roofline"].update({"traffic": traffic, "traffic_source": traffic_source,
                                # the counter-side rate (L2 <-> fabric bytes per launch / launch time; Infinity-Cache hits included)
                                "traffic_rate": (traffic / dur / 1e9) if (traffic is not None and dur > 0) else None})
        out["phases_s"] = phases.snapshot()
        fallback.keep(out)

    # ---- one GPU: the other workloads of the reference's harness, bounded, outside the timed region
    if n == 1 and not args.no_secondary:
        phases.start("secondary")
        dog.note("secondary workloads")
        sec = secondary.secondary(args, b)
        if out is not None:
            out["secondary"] = sec
            out["phases_s"] = phases.snapshot()
            fallback.keep(out)
    # ---- the reference on this box's host cores, beside every line: measured at N = 1 (and left for the runs that follow on this
    # host), quoted at N > 1 — or its bounded sample leg, with the GPU line already in hand
    if rank == 0 and not args.no_cpu_baseline and out["backend"] == transports.PRODUCT_BACKEND:
        phases.start("cpu_baseline")
        dog.note("CPU baseline (the compiled reference on the host cores)")
        out["cpu_baseline"] = baseline.cpu_baseline_for_line(args, n, budget)
    phases.stop()
    if rank == 0:
        out["phases_s"] = phases.snapshot()
        emit(out)
        fallback.printed = True

    dog.phase("teardown", max(args.watchdog, 600.0))  # (the other ranks wait here while rank 0 runs the CPU baseline's sample leg)
    try:
        b.free_current()
    except Exception:  # noqa: BLE001
        pass
    if dist is not None:
        dist.barrier()
    for t in b.transports.values():  # (a transport that gave up mid-call is left to the process exit)
        if t["dead"] is not None:
            t["world"], t["sp"] = None, None
    b.close_transports()
    dog.done()
    check = res["check"]
    if check is not None and not check["ok"]:
        raise SystemExit("bench.py: the result check FAILED: %r" % (check,))
    if int(os.environ.get("HNH_COMM_CUS", "0") or 0) > 0:
        # a process that created a CU-masked stream can hang in the HIP runtime's exit handler (profiles/r04_masked_stream_exit_hang.log):
        # the contexts and their streams were destroyed explicitly above and the line is out, so leave without running exit handlers
        if dist is not None:
            dist.destroy_process_group()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    return out if rank == 0 else None


def main():
    argv = sys.argv[1:]
    args = cli.parse(argv)
    common.claim_stdout()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launcher.launch(args, argv))
    try:
        run(args)
    except BaseException as e:  # one GPU, or a worker: a failure is still reported as one JSON line by whoever owns stdout
        if (int(os.environ.get("RANK", "0")) == 0 and "HNH_BENCH_STATUS_DIR" not in os.environ and not args.probe_transport
                and not (isinstance(e, SystemExit) and e.code in (0, None))):
            emit(error_line(args, "%s: %s" % (type(e).__name__, str(e)[:500]), failed_rank=0))
        raise
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
