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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--logm", type=int, default=20)
    ap.add_argument("--edge-factor", type=int, default=96)
    ap.add_argument("--r", type=int, default=128)
    ap.add_argument("--alg", default="15d_fusion2")
    ap.add_argument("--c", type=int, default=1, help="replication factor of the 1.5D/2.5D schedule")
    ap.add_argument("--ring-mode", choices=["mesh", "relay"], default=None,
                    help="route of the 1.5D dense shift's moving operand: mesh = every block straight from its owner (default), "
                         "relay = the reference's neighbour ring (sets HNH_RING_MODE)")
    ap.add_argument("--chunks", type=int, default=None, help="column chunks of the pipelined mesh fetch (sets HNH_MESH_CHUNKS)")
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
    return ap.parse_args()


class Watchdog:
    """Per-phase watchdog of a multi-GPU run: a phase that does not finish in time prints rank + phase and ends the
    process with a non-zero code (RCCL problems show up as hangs inside C calls; ctypes releases the GIL there)."""

    def __init__(self, rank, seconds, enabled):
        self.rank, self.seconds, self.enabled = rank, seconds, enabled
        self.timer = None

    def phase(self, name, seconds=None):
        import threading
        self.done()
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
        for threads in sorted({min(ncpu, 16), min(ncpu, 32), min(ncpu, 64)}):
            res = RR.bench(m, m, rows, cols, args.r, "15d_fusion2", 1, 1, True, args.cpu_trials, threads=threads)
            tried.append((threads, res["nnz_R_per_s"]))
            if best is None or res["nnz_R_per_s"] > best[1]["nnz_R_per_s"]:
                best = (threads, res)
        threads, res = best
        comp = res["perf_stats"].get("Computation Time", 0.0)
        sweep = "ER 2^%d, edge factor %d (%d nnz): %s nnz*R/s at 16/32/64 threads of %d" % (
            args.cpu_logm, args.edge_factor, len(rows), ", ".join("%d: %.2e" % t for t in tried), ncpu)
        out = {"value": res["nnz_R_per_s"], "unit": "nnz*R/s", "cores": threads, "kind": "reference",
               "sample": "ER 2^%d, edge factor %d (%d nnz), R=%d, 15d_fusion2 fused, %d timed fusedSpMM calls after 1 warm-up, "
                         "1 MPI rank x %d OpenMP/MKL threads (best of the sweep)" % (args.cpu_logm, args.edge_factor, len(rows), args.r,
                                                                                  args.cpu_trials, threads),
               "thread_sweep": sweep, "elapsed_s": res["elapsed"],
               "kernel_only_value": (len(rows) * args.r * args.cpu_trials / comp) if comp > 0 else None}
        if not args.no_cpu_full and args.logm != args.cpu_logm:
            try:
                t0 = time.perf_counter()
                mf = 1 << args.logm
                rows, cols = H.generate_er(mf, mf, mf * args.edge_factor, 12345)
                full = RR.bench(mf, mf, rows, cols, args.r, "15d_fusion2", 1, 1, True, args.cpu_trials, threads=threads, timeout=900.0)
                compf = full["perf_stats"].get("Computation Time", 0.0)
                out.update({"sample_value": out["value"], "sample_workload": out["sample"],
                            "value": full["nnz_R_per_s"], "elapsed_s": full["elapsed"],
                            "kernel_only_value": (len(rows) * args.r * args.cpu_trials / compf) if compf > 0 else None,
                            "sample": "the GPU line's own workload: ER 2^%d, edge factor %d (%d nnz), R=%d, 15d_fusion2 fused, %d timed "
                                      "fusedSpMM calls after 1 warm-up, 1 MPI rank x %d OpenMP/MKL threads (thread count chosen by the sweep); "
                                      "whole leg incl. the reference's set-up %.0f s" % (args.logm, args.edge_factor, len(rows), args.r,
                                                                                       args.cpu_trials, threads, time.perf_counter() - t0)})
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
    if args.gpus > 1:
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # one node: RCCL's bootstrap must not depend on an external interface
    if args.ring_mode:
        os.environ["HNH_RING_MODE"] = args.ring_mode
    if args.chunks:
        os.environ["HNH_MESH_CHUNKS"] = str(args.chunks)
    import torch  # first: one HIP runtime per process (see distributed_sddmm_amd/_kernels.py)
    from distributed_sddmm_amd import api as H

    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    if world_size != n:
        raise SystemExit("bench.py --gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run); got %d" % (n, n, world_size))
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

    # ---- build: same global matrix on every rank count (strong scaling)
    dog.phase("set-up (generator, redistribution, CSR blocks)", max(args.watchdog, 600.0))
    t_setup = time.perf_counter()
    sp = H.SpmatLocal.load_tuples(world, False, args.logm, args.edge_factor)
    info = sp.info()
    nnz, m = info["dist_nnz"], info["M"]
    op = H.DistributedSparse(world, args.alg, sp, args.r, args.c)
    A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
    S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
    barrier()
    t_setup = time.perf_counter() - t_setup

    # ---- several GPUs, 1.5D dense shift: how the moving operand travels.  The reference relays it round a neighbour ring
    # (one xGMI link per direction); the default here fetches every block straight from its owner (all links at once) in chunks,
    # with one windowed kernel pass per landed chunk — how many chunks trades kernel efficiency against fetch/compute overlap
    # and depends on the xGMI bandwidth actually delivered.  Unless --ring-mode / --chunks fix the route, the candidates are
    # MEASURED here (1 warm-up + 3 calls each, max over ranks), outside the timed region; the fastest one is what gets timed
    # and the JSON line records all of them.
    tuning = None
    if n > 1 and args.alg == "15d_fusion2" and args.chunks is None and not args.no_tune and n // args.c > 1 and args.ring_mode != "relay":
        dog.phase("route tuning (mesh chunk counts, relay ring)", max(args.watchdog, 900.0))
        default_q = int(os.environ.get("HNH_MESH_CHUNKS", "4"))
        candidates = [("mesh", q) for q in sorted({default_q, 2, 4, 8}, key=lambda q: (q != default_q, q))]  # the built one first
        if args.ring_mode is None:
            candidates.append(("relay", None))
        built = ("mesh", default_q) if os.environ.get("HNH_RING_MODE", "mesh") == "mesh" else ("relay", None)  # what the operator above is

        def rebuild(route):
            nonlocal op, A, B, S, buf, built
            if route == built:
                return
            for x in (A, B, S, buf):
                x.free()
            op.free()
            os.environ["HNH_RING_MODE"] = route[0]
            if route[1] is not None:
                os.environ["HNH_MESH_CHUNKS"] = str(route[1])
            op = H.DistributedSparse(world, args.alg, sp, args.r, args.c)
            A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
            S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
            built = route

        tuning = {}
        for route in candidates:
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
    alg_bytes_per_call = local_nnz * (8 * args.r + 24) + 16 * args.r * op.info()["localArows"] * args.c
    if dist is not None:
        t = torch.tensor([kern_ms, float(launches), float(alg_bytes_per_call)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        kern_ms, launches, alg_bytes_per_call = float(t[0]) / n, int(t[1]) // n, float(t[2]) / n
    barrier()

    # ---- result check at the reported size (outside the timed region): with A = B = 0.001 and S = 1 one fused call has the
    # closed form A[i, :] = deg(i) * R * 1e-9 (every SDDMM value is R * 1e-6; row i of the SpMM adds deg(i) of them times
    # 0.001).  deg comes from the host generator (bit-identical draws, independent of every device code path).
    check = None
    if not args.no_check:
        import numpy as np
        grows, _ = H.generate_er(m, m, m * args.edge_factor, 12345)
        deg = np.bincount(grows, minlength=m).astype(np.float64)
        nnz_host = int(len(grows))
        del grows
        A.fill(0.001)
        B.fill(0.001)
        step()
        world.sync()
        got = A.download().reshape(-1)
        want_scale = args.r * 1e-9
        worst, elems_checked, off = 0.0, 0, 0
        for top, left, rc, cc in op.submatrices(H.AMAT):
            keep = int(max(0, min(rc, m - top)))
            blk = got[off:off + rc * cc].reshape(rc, cc)[:keep]
            off += rc * cc
            if keep:
                worst = max(worst, float(np.max(np.abs(blk - deg[top:top + keep, None] * want_scale))))
                elems_checked += keep * cc
        ref = float(deg.max()) * want_scale
        local_n = float(op.info()["nS"])
        if dist is not None:
            t = torch.tensor([worst], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst = float(t[0])
            t = torch.tensor([float(elems_checked), local_n], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            elems_checked, local_n = int(t[0]), float(t[1])
        rows_checked = elems_checked // args.r  # every rank checks the rows (and, under an R split, the columns) it owns
        check = {"what": "one fresh fusedSpMM from A = B = 0.001, S = 1 against the closed form A[i,:] = deg(i)*R*1e-9, deg from the host generator",
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
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                with open(tf) as f:
                    rec = json.load(f)
                if rec.get("workload_key") == "er%d_ef%d_r%d_n%d" % (args.logm, args.edge_factor, args.r, n):
                    traffic = rec.get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "backend": H.backend_name(),
            "metric": "fused SDDMM+SpMM nnz*R/s", "value": value, "unit": "nnz*R/s", "n_gpus": n, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Erdos-Renyi 2^%d x 2^%d, edge factor %d (%d unique nnz), R=%d, fused SDDMM->SpMM (fusedSpMM, Amat), "
                                   "%s c=%d on %d x MI355X%s" % (args.logm, args.logm, args.edge_factor, nnz, args.r, args.alg, args.c, n,
                                                                "" if n == 1 else ", RCCL over xGMI (%s)" % (
                                                                    "neighbour relay ring" if os.environ.get("HNH_RING_MODE", "mesh") == "relay" else "chunked fetch from the owners")),
                       "nnz": nnz, "M": m, "R": args.r, "algorithm": args.alg, "c": args.c, "transport": "none" if n == 1 else "rccl",
                       "ring_mode": os.environ.get("HNH_RING_MODE", "mesh") if n > 1 else None,
                       "mesh_chunks": ((int(os.environ["HNH_MESH_CHUNKS"]) if "HNH_MESH_CHUNKS" in os.environ else "default")
                                       if n > 1 and os.environ.get("HNH_RING_MODE", "mesh") == "mesh" else None),
                       "setup_s": round(t_setup, 2)},
            # `achieved` is an ALGORITHMIC rate (SURVEY 8d byte model / measured launch time), not DRAM utilisation: part of every
            # launch's gathers is served by the 256 MiB Infinity Cache, which sits behind the counters `traffic` comes from
            "roofline": {"bound": "hbm", "bound_detail": "hbm gather model (Infinity-Cache assisted); the saturated resource is the memory side "
                                                          "serving scattered dense rows, see DESIGN.md section 3 and profiles/r02_gather_probe*.log",
                         "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "traffic_source": "profiles/hbm_traffic.json (static: rocprofv3 FETCH_SIZE/WRITE_SIZE passes of an earlier run of "
                                           "this command, not collected live)" if traffic is not None else None,
                         "kernel": ("row_kernel<fused> (hnh_fused_sddmm_spmm_csr), one launch per Infinity-Cache panel of B" if n == 1 else
                                    "row_kernel<fused> (hnh_fused_sddmm_spmm_csr): one launch per visiting block of the relay ring"
                                    if os.environ.get("HNH_RING_MODE", "mesh") == "relay" else
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
        if tuning is not None:
            out["config"]["route_tuning_ms_per_step"] = {("mesh/%d chunks" % q if m == "mesh" else "relay ring"): round(v, 4)
                                                         for (m, q), v in tuning.items()}
        if n == 1 and not args.no_cpu_baseline and out["backend"] == "hip-gfx950":
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "nnz*R/s", "cores": os.cpu_count(), "kind": "reference",
                                       "sample": "FAILED: %s" % str(e)[:300]}
        print(json.dumps(out), flush=True)

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


def main():
    args = parse()
    run(args)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
