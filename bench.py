#!/usr/bin/env python3
"""Headline benchmark: fused SDDMM -> SpMM (Distributed_Sparse::fusedSpMM, the reference's
benchmark_dist.cpp:117-149 loop) on an Erdős–Rényi matrix, nnz*R per second.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1 : BASELINE config 2 — ER 2^20 x 2^20 (edge factor 96, ~1.0066e8 nnz), R = 128, one MI355X, the local
        fused kernel behind `15d_fusion2` (no shift).  A "step" = one fusedSpMM(A, B, S, buf, Amat) call.
N > 1 : BASELINE config 3 — the SAME global matrix strong-scaled over N GPUs with the 1.5D dense-shifting
        schedule (RCCL send/recv ring over xGMI, overlapped with the local kernel), one process per GPU.
Inputs are resident in HBM before the timed region (A = B = 0.001, S = 1 as benchmark_dist.cpp:102-106).

One JSON line is printed by rank 0.  Extra objects:
  roofline     — the dominant kernel (fused row pass): algorithmic bytes per launch / average launch
                 duration measured live with HIP events on the compute stream, against 8.0 TB/s HBM.
  cpu_baseline — the reference itself (oracle/_ref/ref_driver = unmodified reference sources + MKL/MPICH)
                 timed on this box's host cores on a bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)


_JSON_FD = None


def claim_stdout():
    """stdout carries exactly ONE line, the JSON result: everything else this process writes to file descriptor 1 (the host
    library mirrors the reference's console messages, e.g. "R-mat generator created ... nonzeros") goes to stderr instead."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (obj if isinstance(obj, str) else json.dumps(obj)) + "\n"
    if _JSON_FD is None:
        sys.stdout.write(line)
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line.encode())


DEFAULT_CHUNKS = "1,2,2,2,1,1"  # the library's default shape of the mesh fetch (dense_shift_15d.hpp)


def set_chunk_spec(spec):
    """A chunk spec is a number (Q symmetric chunks, HNH_MESH_CHUNKS) or a comma list of heights (HNH_MESH_TAPER)."""
    if "," in spec:
        os.environ["HNH_MESH_TAPER"] = spec
        os.environ.pop("HNH_MESH_CHUNKS", None)
    else:
        os.environ["HNH_MESH_CHUNKS"] = spec
        os.environ.pop("HNH_MESH_TAPER", None)


def current_chunk_spec():
    return os.environ.get("HNH_MESH_TAPER") or os.environ.get("HNH_MESH_CHUNKS") or DEFAULT_CHUNKS


def route_name(route):
    c, mode, q = route
    mesh = ("mesh/heights %s" % q) if (q and "," in str(q)) else ("mesh/%s chunks" % q)
    return "c=%d %s" % (c, {"mesh": mesh, "relay": "relay ring", "none": "replication only"}[mode])


def keyed(idx, salt):
    """Deterministic value in [0.5, 1.5) per global index (multiplicative hash): the operands of the result check."""
    import numpy as np
    h = (idx.astype(np.uint64) * np.uint64(2654435761) + np.uint64(salt) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    return 0.5 + h.astype(np.float64) / 4294967296.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--logm", type=int, default=20)
    ap.add_argument("--edge-factor", type=int, default=96)
    ap.add_argument("--r", type=int, default=128)
    ap.add_argument("--alg", default="15d_fusion2")
    ap.add_argument("--c", type=int, default=None, help="replication factor of the 1.5D/2.5D schedule (the reference's command-line "
                    "argument, bench_erdos_renyi.cpp:23-28).  Not given: 1 on one GPU; on several GPUs the candidates 1 / 2 / 4 that "
                    "divide N are MEASURED together with the route (below) and the fastest is timed")
    ap.add_argument("--ring-mode", choices=["mesh", "relay"], default=None,
                    help="route of the 1.5D dense shift's moving operand: mesh = every block straight from its owner (default), "
                         "relay = the reference's neighbour ring (sets HNH_RING_MODE)")
    ap.add_argument("--chunks", default=None, help="chunks of the pipelined mesh fetch: a number Q = symmetric chunks of heights "
                    "(1, 2, .., 2, 1) (HNH_MESH_CHUNKS), or a comma list of heights, e.g. 1,2,2,2,1,1 (HNH_MESH_TAPER)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-logm", type=int, default=18, help="size of the CPU baseline's thread-sweep sample")
    ap.add_argument("--cpu-trials", type=int, default=2)
    ap.add_argument("--no-cpu-full", action="store_true", help="skip the CPU baseline's run at the GPU line's full size")
    ap.add_argument("--no-check", action="store_true", help="skip the closed-form result check (outside the timed region)")
    ap.add_argument("--no-preflight", action="store_true", help="skip the transport self-tests of a multi-GPU run")
    ap.add_argument("--no-tune", action="store_true", help="several GPUs: keep the default route (mesh fetch, 4 chunks) instead of measuring "
                    "mesh with 2 / 4 / 8 chunks and the relay ring")
    ap.add_argument("--watchdog", type=float, default=240.0, help="seconds a multi-GPU phase may take before the rank reports "
                    "where it is stuck and exits non-zero")
    ap.add_argument("--no-live-traffic", action="store_true", help="one GPU: do not run the two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) "
                    "behind roofline.traffic; the tracked profiles/hbm_traffic.json is quoted instead")
    ap.add_argument("--nchannels", type=int, default=None, help="several GPUs: pin RCCL's channel count (NCCL_MIN/MAX_NCHANNELS); every "
                    "channel is a workgroup that competes with the row kernel for CUs and HBM")
    ap.add_argument("--comm-cus", type=int, default=None, help="compute units masked off the compute stream (HNH_COMM_CUS).  Several GPUs: they "
                    "are set aside for the communication stream, RCCL's kernels run only there (default 0).  One GPU: not given = 0 and 16 "
                    "are both measured and the faster is timed")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="self-launched run (--gpus N without WORLD_SIZE): seconds "
                    "before the launcher ends its workers and reports the phase each one was in")
    return ap.parse_args(argv)


class Watchdog:
    """Per-phase watchdog of a multi-GPU run: a phase that does not finish in time prints rank + phase and ends the
    process with a non-zero code (RCCL problems show up as hangs inside C calls; ctypes releases the GIL there)."""

    def __init__(self, rank, seconds, enabled):
        self.rank, self.seconds, self.enabled = rank, seconds, enabled
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

        def fire():
            sys.stderr.write("[bench.py watchdog] rank %d stuck in phase '%s' for more than %.0f s - giving up\n" % (self.rank, name, limit))
            sys.stderr.flush()
            os._exit(3)

        self.timer = threading.Timer(limit, fire)
        self.timer.daemon = True
        self.timer.start()

    def done(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def cpu_baseline(args):
    """The reference timed on the host cores: a thread sweep on a bounded sample (ER 2^cpu_logm, same edge factor and R), then
    the best thread count ONCE at the GPU line's own size (1 warm-up + cpu_trials timed calls, benchmark_dist.cpp:117-149);
    `value` is the full-size figure when that leg ran."""
    import numpy as np
    from distributed_sddmm_amd import api as H
    from oracle import refrun as RR
    ncpu = os.cpu_count() or 1
    m = 1 << args.cpu_logm
    if RR.available():
        rows, cols = H.generate_er(m, m, m * args.edge_factor, 12345)
        # The reference does not scale with the thread count on big hosts (measured on 2 x EPYC 9575F: 32 threads
        # beat 64/128/256, and 1 MPI rank beats 4..32, profiles/r01_cpu_baseline_sweep.log), so a few counts are
        # tried on the sample and the best one is used; `cores` is the thread count of the reported run.
        tried, best = [], None
        # (MPI ranks, OpenMP/MKL threads per rank): the thread counts on one rank, then the same cores split over several ranks
        # (the reference is an MPI + OpenMP code; on the driver box one rank beat 4 .. 32, but that is the box's call)
        configs = [(1, t) for t in sorted({min(ncpu, 16), min(ncpu, 32), min(ncpu, 64)})]
        configs += [(pr, max(1, min(ncpu, 64) // pr)) for pr in (4, 8) if ncpu >= 2 * pr]
        for ranks, threads in configs:
            try:
                res = RR.bench(m, m, rows, cols, args.r, "15d_fusion2", ranks, 1, True, args.cpu_trials, threads=threads, timeout=300.0)
            except Exception as e:  # one configuration failing (e.g. no MPI launcher for several ranks) does not lose the others
                tried.append((ranks, threads, None, str(e)[:80]))
                continue
            tried.append((ranks, threads, res["nnz_R_per_s"], None))
            if best is None or res["nnz_R_per_s"] > best[2]["nnz_R_per_s"]:
                best = (ranks, threads, res)
        if best is None:
            raise RuntimeError("the compiled reference ran in none of the configurations: %r" % (tried,))
        ranks, threads, res = best
        comp = res["perf_stats"].get("Computation Time", 0.0)
        sweep = "ER 2^%d, edge factor %d (%d nnz), ranks x threads -> nnz*R/s: %s (host has %d hardware threads)" % (
            args.cpu_logm, args.edge_factor, len(rows),
            ", ".join("%dx%d: %s" % (pr, t, ("%.2e" % v) if v is not None else "failed") for pr, t, v, _ in tried), ncpu)
        out = {"value": res["nnz_R_per_s"], "unit": "nnz*R/s", "cores": ranks * threads, "kind": "reference",
               "sample": "ER 2^%d, edge factor %d (%d nnz), R=%d, 15d_fusion2 fused, %d timed fusedSpMM calls after 1 warm-up, "
                         "%d MPI rank(s) x %d OpenMP/MKL threads (best of the sweep)" % (args.cpu_logm, args.edge_factor, len(rows), args.r,
                                                                                       args.cpu_trials, ranks, threads),
               "thread_sweep": sweep, "ranks": ranks, "threads_per_rank": threads, "elapsed_s": res["elapsed"],
               "kernel_only_value": (len(rows) * args.r * args.cpu_trials / comp) if comp > 0 else None}
        if not args.no_cpu_full and args.logm != args.cpu_logm:
            try:
                t0 = time.perf_counter()
                mf = 1 << args.logm
                rows, cols = H.generate_er(mf, mf, mf * args.edge_factor, 12345)
                full = RR.bench(mf, mf, rows, cols, args.r, "15d_fusion2", ranks, 1, True, args.cpu_trials, threads=threads, timeout=900.0)
                compf = full["perf_stats"].get("Computation Time", 0.0)
                out.update({"sample_value": out["value"], "sample_workload": out["sample"],
                            "value": full["nnz_R_per_s"], "elapsed_s": full["elapsed"],
                            "kernel_only_value": (len(rows) * args.r * args.cpu_trials / compf) if compf > 0 else None,
                            "sample": "the GPU line's own workload: ER 2^%d, edge factor %d (%d nnz), R=%d, 15d_fusion2 fused, %d timed "
                                      "fusedSpMM calls after 1 warm-up, %d MPI rank(s) x %d OpenMP/MKL threads (chosen by the sweep); "
                                      "whole leg incl. the reference's set-up %.0f s" % (args.logm, args.edge_factor, len(rows), args.r,
                                                                                       args.cpu_trials, ranks, threads, time.perf_counter() - t0)})
            except Exception as e:  # keep the sample figure
                out["full_size_error"] = str(e)[:300]
        return out
    # no compiled reference on this box: time the numpy port on a smaller sample
    from oracle import oracle as O
    m = 1 << 14
    rows, cols = O.erdos_renyi(14, args.edge_factor)
    a, b = np.full((m, args.r), 0.001), np.full((m, args.r), 0.001)
    t0 = time.perf_counter()
    O.fused_a(rows, cols, np.ones(len(rows)), a, b)
    dt = time.perf_counter() - t0
    return {"value": len(rows) * args.r / dt, "unit": "nnz*R/s", "cores": 1, "kind": "port",
            "sample": "numpy restatement, ER 2^14, edge factor %d (%d nnz), R=%d, one fused call" % (args.edge_factor, len(rows), args.r)}


def live_traffic(args):
    """roofline.traffic collected in THIS run: two short rocprofv3 passes of this very command in a child process — FETCH_SIZE
    and WRITE_SIZE each in its own pass (they do not fit one; a counter pass is never combined with a trace domain) — read from
    rocprofv3's rocpd database, per launch of the fused row kernel, with the micro-architecture guide's gfx950 correction
    (FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes: x 2; WRITE_SIZE as reported).  Runs outside
    the timed region while this process is idle.  None when rocprofv3 is absent, this run is itself being profiled, or a pass fails."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None or "rocprofiler" in os.environ.get("LD_PRELOAD", "") or "ROCPROFILER_SDK_TOOL_LIBRARIES" in os.environ or "ROCP_TOOL_LIBRARIES" in os.environ:
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    means, t0 = {}, time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hnh_pmc_", dir="/tmp")
        try:
            cmd = [prof, "--pmc", counter, "-d", d, "-o", "pass", "--", sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "2",
                   "--warmup", "1", "--no-cpu-baseline", "--no-check", "--no-live-traffic", "--logm", str(args.logm), "--edge-factor",
                   str(args.edge_factor), "--r", str(args.r), "--alg", args.alg]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=420, check=True)
            vals = []
            for db in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
                cur = sqlite3.connect(db).cursor()
                vals += [r[0] for r in cur.execute("select value from counters_collection where kernel_name like ? and counter_name = ?",
                                                   ("%::row_kernel<%", counter))]
            if not vals:
                return None
            means[counter] = (sum(vals) / len(vals), len(vals))
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch_kb, write_kb = means["FETCH_SIZE"][0], means["WRITE_SIZE"][0]
    return {"bytes_per_launch": 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0, "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb,
            "launches_sampled": means["FETCH_SIZE"][1], "seconds": round(time.perf_counter() - t0, 1)}


def gpu_world(H, dist, rank, n, local_rank):
    """The product transport: one process per GPU, RCCL over xGMI (unique id bootstrapped through torch.distributed)."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    ndev = torch.cuda.device_count()
    if n > 1 and 1 < ndev < n:
        raise SystemExit("bench.py --gpus %d: this process sees %d GPUs; one process per GPU needs either all %d visible to every rank "
                         "(LOCAL_RANK picks one) or exactly one per rank (launcher-side isolation)" % (n, ndev, n))
    # ndev == 1 with several ranks: the launcher gave every rank its own device; if they are in fact the same physical GPU,
    # RCCL's communicator creation reports it (duplicate GPU) and the run ends with that error
    device = local_rank % ndev
    torch.cuda.set_device(device)
    assert H.load_backend(None) == "hip-gfx950"
    if n == 1:
        return H.World.single(device), torch.cuda.synchronize
    ident = [H.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ident, src=0)
    try:
        return H.World.rccl(rank, n, device, ident[0]), torch.cuda.synchronize
    except Exception as e:
        raise SystemExit("bench.py --gpus %d, rank %d on device %d of %d visible: the RCCL communicator could not be created: %s\n"
                         "(\"invalid usage\" here usually means two ranks share one physical GPU, which RCCL refuses)" % (n, rank, device, ndev, e))


def run(args, make_world=gpu_world):
    """`make_world` is replaceable so that tests can drive this exact function over gloo on CPU."""
    if args.gpus > 1 and "HNH_KEEP_OMP" not in os.environ:
        # torch.distributed.run pins OMP_NUM_THREADS=1 per worker; the host-side setup (generator, sorts, CSR build)
        # is OpenMP code, so give every rank its share of the host cores instead (must happen before libgomp starts)
        os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL across processes)
    # HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share one serialise: with the
    # framework's own streams and RCCL's in the process, make sure the compute and the communication stream never have to share
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if args.gpus > 1:
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # one node: RCCL's bootstrap must not depend on an external interface
    if args.ring_mode:
        os.environ["HNH_RING_MODE"] = args.ring_mode
    if args.chunks:
        set_chunk_spec(str(args.chunks))
    if args.gpus > 1 and getattr(args, "comm_cus", None):
        os.environ["HNH_COMM_CUS"] = str(args.comm_cus)
    if args.gpus > 1 and getattr(args, "nchannels", None):
        os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(args.nchannels)
    import torch  # first: one HIP runtime per process (see distributed_sddmm_amd/_kernels.py)
    from distributed_sddmm_amd import api as H

    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    if world_size != n:  # main() self-launches when WORLD_SIZE is absent; this is a launcher that disagrees with --gpus
        raise SystemExit("bench.py --gpus %d was started as rank %d of WORLD_SIZE=%d: the launcher's process count and --gpus disagree" % (n, rank, world_size))
    dist = None
    if n > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: never depend on the container hostname resolving
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo", rank=rank, world_size=n)  # bootstrap + barriers only; data moves over RCCL
    dog = Watchdog(rank, args.watchdog, n > 1)
    dog.phase("transport creation (RCCL communicator)")
    world, device_sync = make_world(H, dist, rank, n, local_rank)
    dog.done()

    # ---- multi-GPU preflight: every transport primitive the schedules use, on small buffers with known contents, each under
    # the watchdog, so that a transport problem is reported as "rank r, primitive X" instead of a hang; then the order in which
    # the ranks created their communicators is compared
    preflight = None
    if n > 1 and not args.no_preflight:
        preflight = {}
        for what, name in enumerate(H.World.PREFLIGHT):
            dog.phase("preflight: " + name, 90.0)
            err = world.preflight(what, 1 << 16)
            dog.done()
            if not err <= 1e-9:
                sys.stderr.write("[bench.py preflight] rank %d: %s delivered wrong data (max deviation %.3e)\n" % (rank, name, err))
                sys.stderr.flush()
                os._exit(4)
            preflight[name] = err
        sig = [None] * n
        dist.all_gather_object(sig, world.split_signature())
        if len(set(sig)) != 1:
            sys.stderr.write("[bench.py preflight] ranks created their communicators in different orders: %r\n" % (sig,))
            sys.stderr.flush()
            os._exit(4)

    def barrier():
        if dist is not None:
            dist.barrier()
        world.sync()
        device_sync()

    # ---- one GPU: how many compute units the compute stream uses.  The fused pass is bound by the memory side, not by CUs, and runs
    # ~1 % FASTER with 8 .. 24 of the 256 CUs masked off its stream (fewer requesters queueing at the fabric;
    # profiles/r03_kbench_cus_off_x_waves_cap.log).  Measured here, not assumed: unless --comm-cus fixes it, the candidates 0 and 16
    # are timed (1 warm-up + 3 calls each, outside the timed region) and the faster one is what gets timed; both are recorded.
    cu_tuning = None
    if n == 1 and make_world is gpu_world and args.comm_cus is None and not args.no_tune and "HNH_COMM_CUS" not in os.environ:
        cu_tuning = {}
        for off in (0, 16):
            os.environ["HNH_COMM_CUS"] = str(off)
            if off:
                world.close()
                world, device_sync = make_world(H, dist, rank, n, local_rank)
            sp_t = H.SpmatLocal.load_tuples(world, False, args.logm, args.edge_factor)
            op_t = H.DistributedSparse(world, args.alg, sp_t, args.r, args.c or 1)
            sp_t.free()
            xs = (op_t.like_A_matrix(0.001), op_t.like_B_matrix(0.001), op_t.like_S_values(1.0), op_t.like_S_values(0.0))
            op_t.fusedSpMM(*xs, H.AMAT)
            world.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                op_t.fusedSpMM(*xs, H.AMAT)
            world.sync()
            cu_tuning[off] = (time.perf_counter() - t0) / 3 * 1e3
            for x in xs:
                x.free()
            op_t.free()
        best_off = min(cu_tuning, key=cu_tuning.get)
        if best_off != 16:  # the world in hand was made with 16 CUs off
            os.environ["HNH_COMM_CUS"] = str(best_off)
            world.close()
            world, device_sync = make_world(H, dist, rank, n, local_rank)
    elif n == 1 and args.comm_cus is not None and "HNH_COMM_CUS" not in os.environ and make_world is gpu_world:
        os.environ["HNH_COMM_CUS"] = str(args.comm_cus)
        world.close()
        world, device_sync = make_world(H, dist, rank, n, local_rank)

    # ---- build: same global matrix on every rank count (strong scaling)
    dog.phase("set-up (generator, redistribution, CSR blocks)", max(args.watchdog, 600.0))
    t_setup = time.perf_counter()
    sp = H.SpmatLocal.load_tuples(world, False, args.logm, args.edge_factor)
    info = sp.info()
    nnz, m = info["dist_nnz"], info["M"]
    c_now = args.c or 1
    op = H.DistributedSparse(world, args.alg, sp, args.r, c_now)
    A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
    S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
    barrier()
    t_setup = time.perf_counter() - t_setup

    # ---- several GPUs, 1.5D dense shift: replication factor and route of the moving operand.  The reference takes c on the
    # command line (bench_erdos_renyi.cpp:23-28) and relays the moving operand round a neighbour ring (one xGMI link per
    # direction); the default here fetches every block straight from its owner (all links at once) in chunks, with one windowed
    # kernel pass per landed chunk — how many chunks trades kernel efficiency against fetch/compute overlap, and c trades ring
    # traffic against replication traffic; both depend on the xGMI bandwidth actually delivered.  Unless --c / --ring-mode /
    # --chunks fix them, the candidates are MEASURED here (1 warm-up + 3 calls each, max over ranks), outside the timed region;
    # the fastest one is what gets timed and the JSON line records all of them.
    tuning = None
    if n > 1 and args.alg == "15d_fusion2" and not args.no_tune:
        default_q = current_chunk_spec()
        fixed_mode = os.environ.get("HNH_RING_MODE") if (args.ring_mode or "HNH_RING_MODE" in os.environ) else None
        cs = [args.c] if args.c else [c for c in (1, 2, 4) if n % c == 0]
        candidates = []
        for c in cs:
            if n // c == 1:  # the whole ring is one rank: nothing shifts, the layers only replicate and reduce
                candidates.append((c, "none", None))
                continue
            if fixed_mode != "relay":
                # chunk shapes: the library's default, symmetric Q = 2 / 4 (/ 3 / 8), and for c = 1 a longer falling shape
                qs = [str(args.chunks)] if args.chunks else sorted(
                    {default_q, DEFAULT_CHUNKS, "2", "4"} | ({"3", "8", "3,4,4,3,2,1,1"} if c == 1 else set()), key=lambda q: (q != default_q, len(q), q))
                candidates += [(c, "mesh", q) for q in qs]
            if fixed_mode != "mesh":
                candidates.append((c, "relay", None))
        mode0 = os.environ.get("HNH_RING_MODE", "mesh")
        built = (c_now, "none", None) if n // c_now == 1 else (c_now, mode0, default_q if mode0 == "mesh" else None)  # the operator above
        candidates.sort(key=lambda k: k != built)  # the built one first
        if len(candidates) > 1:
            dog.phase("route tuning (replication factor, mesh chunk counts, relay ring)", max(args.watchdog, 900.0))

            def rebuild(route):
                nonlocal op, A, B, S, buf, built, c_now
                if route == built:
                    return
                for x in (A, B, S, buf):
                    x.free()
                op.free()
                c_now = route[0]
                if route[1] != "none":
                    os.environ["HNH_RING_MODE"] = route[1]
                if route[2] is not None:
                    set_chunk_spec(route[2])
                op = H.DistributedSparse(world, args.alg, sp, args.r, c_now)
                A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
                S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
                built = route

            tuning = {}
            for route in candidates:
                dog.note("route tuning: " + route_name(route))
                rebuild(route)
                op.fusedSpMM(A, B, S, buf, H.AMAT)
                barrier()
                t0 = time.perf_counter()
                for _ in range(3):
                    op.fusedSpMM(A, B, S, buf, H.AMAT)
                barrier()
                t = torch.tensor([(time.perf_counter() - t0) / 3], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tuning[route] = float(t.item()) * 1e3
            rebuild(min(tuning, key=tuning.get))  # the same choice on every rank: the times are the all-reduced maxima
            A.fill(0.001)
            barrier()
    sp.free()
    ring_mode_now = None if (n == 1 or n // c_now == 1) else os.environ.get("HNH_RING_MODE", "mesh")
    transport_kind = op.json_algorithm_info().get("transport", "?")  # "rccl" in production; tests substitute other transports

    def step():
        op.fusedSpMM(A, B, S, buf, H.AMAT)

    dog.phase("warm-up steps")
    for _ in range(args.warmup):
        step()
    barrier()
    dog.phase("timed steps")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    dog.phase("roofline leg and result check", max(args.watchdog, 600.0))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline leg (outside the timed region): HIP events around every local kernel launch
    prof_calls = max(2, min(5, args.steps))
    op.kernel_profile(1)
    for _ in range(prof_calls):
        step()
    world.sync()
    kern_ms, launches = op.kernel_profile(0)
    local_nnz = op.info()["nS"]
    launches_per_call_local = max(1, launches // prof_calls)
    # SURVEY 8(d), per fused call of this rank: per nonzero 8R + 24 bytes, per output row 16R (row operand read + output
    # row written ONCE) — however many launches the implementation uses (it re-reads rows per launch; that is its cost)
    alg_bytes_per_call = local_nnz * (8 * args.r + 24) + 16 * args.r * op.info()["localArows"] * c_now
    if dist is not None:
        t = torch.tensor([kern_ms, float(launches), float(alg_bytes_per_call)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        kern_ms, launches, alg_bytes_per_call = float(t[0]) / n, int(t[1]) // n, float(t[2]) / n
    barrier()

    # ---- result check at the reported size (outside the timed region), with operands keyed by GLOBAL row and column so that a
    # block that lands in the wrong rows of a landing buffer, comes from the wrong peer or is a stale chunk changes the answer:
    # A[i,k] = a_i u_k, B[j,k] = b_j v_k (hashes of the global indices), S = 1  =>  sddmm(i,j) = a_i b_j W with W = sum_k u_k v_k
    # and one fused call leaves  A[i,k] = W a_i v_k sum_{j in row i} b_j^2.  The sum comes from the HOST generator's draws
    # (bit-identical to the device generator, independent of every device code path) in O(nnz).
    check = None
    if not args.no_check:
        import numpy as np
        grows, gcols = H.generate_er(m, m, m * args.edge_factor, 12345)
        nnz_host = int(len(grows))
        a_key, b_key = keyed(np.arange(m), 1), keyed(np.arange(m), 2)
        u_key, v_key = keyed(np.arange(args.r), 3), keyed(np.arange(args.r), 4)
        rowsum = np.bincount(grows, weights=b_key[gcols] ** 2, minlength=m)
        del grows, gcols
        want_row = float(np.dot(u_key, v_key)) * a_key * rowsum  # times v_k per column

        def keyed_local(mat_mode, row_key, col_key):
            parts = []
            for top, left, rc, cc in op.submatrices(mat_mode):
                blk = np.zeros((rc, cc))
                keep = int(max(0, min(rc, m - top)))
                blk[:keep] = row_key[top:top + keep, None] * col_key[None, left:left + cc]
                parts.append(blk.reshape(-1))
            return np.concatenate(parts)

        A.upload(keyed_local(H.AMAT, a_key, u_key).reshape(A.shape))
        B.upload(keyed_local(H.BMAT, b_key, v_key).reshape(B.shape))
        step()
        world.sync()
        got = A.download().reshape(-1)
        worst, elems_checked, off = 0.0, 0, 0
        for top, left, rc, cc in op.submatrices(H.AMAT):
            keep = int(max(0, min(rc, m - top)))
            blk = got[off:off + rc * cc].reshape(rc, cc)[:keep]
            off += rc * cc
            if keep:
                worst = max(worst, float(np.max(np.abs(blk - want_row[top:top + keep, None] * v_key[None, left:left + cc]))))
                elems_checked += keep * cc
        ref = float(want_row.max() * v_key.max())
        local_n = float(op.info()["nS"])
        if dist is not None:
            t = torch.tensor([worst], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst = float(t[0])
            t = torch.tensor([float(elems_checked), local_n], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            elems_checked, local_n = int(t[0]), float(t[1])
        rows_checked = elems_checked // args.r  # every rank checks the rows (and, under an R split, the columns) it owns
        check = {"what": "one fresh fusedSpMM from operands keyed by global row and column (A[i,k] = a_i u_k, B[j,k] = b_j v_k, S = 1) against "
                         "the closed form A[i,k] = (u.v) a_i v_k sum_{j in row i} b_j^2, the sum taken over the host generator's nonzeros",
                 "rel_err": worst / ref, "tolerance": 1e-11, "rows_checked": int(rows_checked),
                 "nnz_operator": int(nnz), "nnz_host_generator": nnz_host, "nnz_in_blocks_all_ranks": int(local_n),
                 "ok": bool(worst / ref <= 1e-11 and nnz_host == nnz and rows_checked == m)}
        barrier()

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = nnz * args.r * args.steps / elapsed
        launches_per_call = max(1, launches // prof_calls)
        dur = kern_ms / max(1, launches) * 1e-3  # average launch duration, seconds
        bytes_per_launch = alg_bytes_per_call / launches_per_call
        achieved = bytes_per_launch / dur if dur > 0 else 0.0
        traffic, traffic_source, live = None, None, None
        if n == 1 and not args.no_live_traffic and H.backend_name() == "hip-gfx950":
            live = live_traffic(args)
        if live is not None:
            traffic = live["bytes_per_launch"]
            traffic_source = ("live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this command run by this process (%d launches "
                              "sampled, %.0f s, outside the timed region); 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of the "
                              "micro-architecture guide); raw KB: fetch %.0f, write %.0f" % (live["launches_sampled"], live["seconds"],
                                                                                              live["fetch_size_kb_raw"], live["write_size_kb_raw"]))
        else:
            tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tf):
                try:
                    with open(tf) as f:
                        rec = json.load(f)
                    if rec.get("workload_key") == "er%d_ef%d_r%d_n%d" % (args.logm, args.edge_factor, args.r, n):
                        traffic = rec.get("bytes_per_launch")
                        traffic_source = ("profiles/hbm_traffic.json (static: rocprofv3 FETCH_SIZE/WRITE_SIZE passes of an earlier run of "
                                          "this command, not collected live)")
                except Exception:
                    traffic = None
        out = {
            "backend": H.backend_name(),
            "metric": "fused SDDMM+SpMM nnz*R/s", "value": value, "unit": "nnz*R/s", "n_gpus": n, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Erdos-Renyi 2^%d x 2^%d, edge factor %d (%d unique nnz), R=%d, fused SDDMM->SpMM (fusedSpMM, Amat), "
                                   "%s c=%d on %d x MI355X%s" % (args.logm, args.logm, args.edge_factor, nnz, args.r, args.alg, c_now, n,
                                                                "" if n == 1 else ", %s (%s)" % ("RCCL over xGMI" if transport_kind == "rccl" else "transport: " + transport_kind,
                                                                    {"relay": "neighbour relay ring", "mesh": "chunked fetch from the owners",
                                                                     None: "replication only, nothing shifts"}[ring_mode_now])),
                       "nnz": nnz, "M": m, "R": args.r, "algorithm": args.alg, "c": c_now, "transport": "none" if n == 1 else transport_kind,
                       "ring_mode": ring_mode_now,
                       # Q symmetric chunks (a number) or the chunk heights (a comma list)
                       "mesh_chunks": (current_chunk_spec() if ring_mode_now == "mesh" else None),
                       "rccl_channels": (os.environ.get("NCCL_MAX_NCHANNELS", "default") if n > 1 else None),
                       # compute units masked off the compute stream (several GPUs: they run the communication stream)
                       "comm_cus": int(os.environ.get("HNH_COMM_CUS", "0")),
                       "setup_s": round(t_setup, 2)},
            # `achieved` is an ALGORITHMIC rate (SURVEY 8d byte model / measured launch time), not DRAM utilisation: part of every
            # launch's gathers is served by the 256 MiB Infinity Cache, which sits behind the counters `traffic` comes from
            "roofline": {"bound": "hbm", "bound_detail": "hbm gather model (Infinity-Cache assisted); the saturated resource is the memory side "
                                                          "serving scattered dense rows, see DESIGN.md section 3 and profiles/r02_gather_probe*.log",
                         "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "kernel": ("row_kernel<fused> (hnh_fused_sddmm_spmm_csr), one launch per Infinity-Cache panel of B" if n == 1 else
                                    "row_kernel<fused> (hnh_fused_sddmm_spmm_csr): one launch per visiting block of the relay ring"
                                    if ring_mode_now == "relay" else
                                    "row_kernel<fused> (hnh_fused_sddmm_spmm_csr): the rank's one block (replication only)" if ring_mode_now is None else
                                    "row_kernel<fused> (hnh_fused_sddmm_spmm_csr / _w): own block, then one windowed pass over the fetched blocks per landed chunk"),
                         "avg_launch_ms": dur * 1e3,
                         # the counter-side rate (L2 <-> fabric bytes per launch / launch time; Infinity-Cache hits included)
                         # and the compulsory floor of a call (every dense row and every nonzero touched once)
                         "traffic_rate": (traffic / dur / 1e9) if (traffic is not None and dur > 0) else None,
                         "compulsory_bytes_per_call": 8 * args.r * (2 * m + m) + 24 * nnz,
                         "launches_per_step": launches_per_call, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "model": "per fused call nnz*(8R+24) + 16*R*rows (SURVEY 8d), divided evenly over its launches"},
        }
        if check is not None:
            out["check"] = check
        if preflight is not None:
            out["preflight"] = {"primitives_ok": sorted(preflight), "communicator_split_order": "identical on all ranks"}
        if cu_tuning is not None:
            out["config"]["cu_tuning_ms_per_step"] = {"%d CUs masked off the compute stream" % k: round(v, 4) for k, v in cu_tuning.items()}
        if tuning is not None:
            out["config"]["route_tuning_ms_per_step"] = {route_name(k): round(v, 4) for k, v in tuning.items()}
        if n == 1 and not args.no_cpu_baseline and out["backend"] == "hip-gfx950":
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "nnz*R/s", "cores": os.cpu_count(), "kind": "reference",
                                       "sample": "FAILED: %s" % str(e)[:300]}
        emit(out)

    dog.phase("teardown")
    for x in (A, B, S, buf):
        x.free()
    op.free()
    if dist is not None:
        dist.barrier()
    world.close()
    dog.done()
    if check is not None and not check["ok"]:
        raise SystemExit("bench.py: the result check FAILED: %r" % (check,))
    return out if rank == 0 else None


def error_line(args, message, **extra):
    """The one JSON line of a run that failed: the contract's keys with value null, plus what went wrong and where."""
    out = {"metric": "fused SDDMM+SpMM nnz*R/s", "value": None, "unit": "nnz*R/s", "n_gpus": args.gpus, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic", "config": {"workload": "Erdos-Renyi 2^%d, edge factor %d, R=%d, %s on %d x MI355X" % (
               args.logm, args.edge_factor, args.r, args.alg, args.gpus)}, "error": message}
    out.update(extra)
    return out


def launch(args, argv):
    """`python bench.py --gpus N` typed as is (no WORLD_SIZE in the environment): start the N workers ourselves — one process
    per GPU, rendezvous on 127.0.0.1 at a free port, the same environment torch.distributed.run would give them — forward
    rank 0's JSON line, and return the worst exit code.  Whatever happens ONE JSON line is printed: a rank that fails or
    hangs is named together with the phase it was in (the workers keep that in a status file), the others are ended."""
    import shutil
    import socket
    import subprocess
    import tempfile
    import threading
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    status_dir = tempfile.mkdtemp(prefix="hnh_bench_")
    worker = os.environ.get("HNH_BENCH_WORKER") or os.path.abspath(__file__)  # tests substitute a worker with a CPU transport
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

    deadline = time.monotonic() + args.launch_timeout
    first_bad, grace, timed_out = None, None, False
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        for r, p in enumerate(procs):
            if first_bad is None and p.poll() not in (None, 0):
                first_bad, grace = (r, p.returncode, phase_of(r)), now + 20.0  # the others get a moment to report, then are ended
        if (grace is not None and now > grace) or now > deadline:
            timed_out = now > deadline and first_bad is None
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


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    claim_stdout()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args, argv))
    try:
        run(args)
    except BaseException as e:  # one GPU, or a worker: a failure is still reported as one JSON line by whoever owns stdout
        if int(os.environ.get("RANK", "0")) == 0 and "HNH_BENCH_STATUS_DIR" not in os.environ and not (isinstance(e, SystemExit) and e.code in (0, None)):
            emit(error_line(args, "%s: %s" % (type(e).__name__, str(e)[:500]), failed_rank=0))
        raise
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
