"""The timed path of bench.py: one route's operator and operands (build), one step of the reference's timed loop
(benchmark_dist.cpp:117-141), the measurement (warm-up, K timed steps bracketed by barrier + device synchronise, max over ranks,
the event-bracketed roofline leg, the result check) and the JSON line made from it."""
import os
import time

from . import checks, common
from .common import GAT_LAYERS, HBM_PEAK, route_name


class Bench:
    """Operator + operands of ONE route at a time, on one of several transports; builds, times and checks it."""

    def __init__(self, args, H, torch, dist, rank, n, dog, workload):
        self.args, self.H, self.torch, self.dist, self.rank, self.n, self.dog, self.wl = args, H, torch, dist, rank, n, dog, workload
        self.transports = {}  # name -> {"world", "sync", "sp", "dead"}
        self.route, self.op, self.A, self.B, self.S, self.buf, self.als, self.gat, self.gat_x = None, None, None, None, None, None, None, None, None
        self.nnz, self.m = None, None
        self.setup_s, self.parse_s = None, None

    # -- transports
    def add_transport(self, name, world, device_sync):
        self.transports[name] = {"world": world, "sync": device_sync, "sp": None, "dead": None}

    def usable(self):
        return [k for k, t in self.transports.items() if t["dead"] is None]

    def world(self, name=None):
        return self.transports[name or self.route[0]]["world"]

    def close_transports(self):
        for t in self.transports.values():
            if t["sp"] is not None:
                t["sp"].free()
                t["sp"] = None
            if t["world"] is not None:
                t["world"].close()
                t["world"] = None

    def barrier(self, name=None):
        t = self.transports[name or self.route[0]]
        if self.dist is not None:
            self.dist.barrier()
        t["world"].sync()
        t["sync"]()

    def max_over_ranks(self, v):
        if self.dist is None:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(self, ok):
        """True iff every rank says so (a candidate that failed on one rank failed)."""
        if self.dist is None:
            return ok
        t = self.torch.tensor([1.0 if ok else 0.0], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    # -- one route
    def free_current(self):
        for x in (self.A, self.B, self.S, self.buf, self.gat_x):
            if x is not None:
                x.free()
        for x in (self.als, self.gat):
            if x is not None:
                x.free()
        if self.op is not None:
            self.op.free()
        self.route, self.op, self.A, self.B, self.S, self.buf, self.als, self.gat, self.gat_x = None, None, None, None, None, None, None, None, None

    def build(self, route):
        if route == self.route:
            return
        self.free_current()
        H, args = self.H, self.args
        tr, c, mode, q = route
        t = self.transports[tr]
        if mode in ("mesh", "relay"):
            os.environ["HNH_RING_MODE"] = mode
        if q is not None:
            common.set_chunk_spec(q)
        elif mode == "fusion1":  # (its own default chunk shape, not the previous candidate's: the library reads the same two variables)
            os.environ.pop("HNH_MESH_TAPER", None)
            os.environ.pop("HNH_MESH_CHUNKS", None)
        t0 = time.perf_counter()
        if t["sp"] is None:
            t["sp"] = self.wl.load(H, t["world"])
            t["world"].sync()
            self.parse_s = time.perf_counter() - t0  # generator / file parser + duplicate merge (the tuples are on the device)
            info = t["sp"].info()
            self.nnz, self.m = info["dist_nnz"], info["M"]
        r0 = GAT_LAYERS[0][0] if args.app == "gat" else args.r
        self.op = H.DistributedSparse(t["world"], "15d_fusion1" if mode == "fusion1" else args.alg, t["sp"], r0, c)
        self.route = route
        if args.app == "als":
            self.als = H.DistributedALS(self.op, True)
        elif args.app == "gat":
            self.gat = H.GAT(self.op, GAT_LAYERS, 0.2)
            self.op.setRValue(GAT_LAYERS[0][0])
            self.gat_x = H.Dense.create(t["world"], *self.gat.buffer_shape(0))
            self.gat_x.fill(0.001)
            self.gat.set_input(self.gat_x)
        else:
            self.A, self.B = self.op.like_A_matrix(0.001), self.op.like_B_matrix(0.001)
            self.S, self.buf = self.op.like_S_values(1.0), self.op.like_S_values(0.0)
        self.barrier()
        if self.setup_s is None:
            self.setup_s = time.perf_counter() - t0

    def step(self):
        if self.als is not None:
            self.als.run_cg(1)  # benchmark_dist.cpp:134-137
        elif self.gat is not None:
            self.gat.forwardPass()  # benchmark_dist.cpp:131-133
        else:
            self.op.fusedSpMM(self.A, self.B, self.S, self.buf, self.H.AMAT)

    # -- the full measurement of the current route
    def measure(self):
        """warm-up, K timed steps (barrier + device synchronise on both sides, max over ranks), the roofline leg and the check."""
        args, torch, dist, H = self.args, self.torch, self.dist, self.H
        self.dog.phase("warm-up steps [%s]" % route_name(self.route))
        for _ in range(args.warmup):
            self.step()
        self.barrier()
        self.dog.phase("timed steps [%s]" % route_name(self.route))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            self.step()
        self.barrier()
        elapsed = self.max_over_ranks(time.perf_counter() - t0)
        self.dog.phase("roofline leg and result check [%s]" % route_name(self.route), max(args.watchdog, 600.0))
        # roofline leg (outside the timed region): HIP events around every local kernel launch
        prof_calls = max(2, min(5, args.steps))
        self.op.kernel_profile(1)
        for _ in range(prof_calls):
            self.step()
        self.world().sync()
        kern_ms, launches = self.op.kernel_profile(0)
        info = self.op.info()
        c_now = self.route[1]
        # SURVEY 8(d), per fused call of this rank: per nonzero 8R + 24 bytes, per output row 16R (row operand read + output row written
        # ONCE) — however many launches the implementation uses (it re-reads rows per launch; that is its cost)
        if args.app == "gat":  # one fused head per (layer, head) at R = features per head
            alg_bytes_per_step = sum(h * (info["nS"] * (8 * f + 24) + 16 * f * info["localArows"] * c_now) for _, f, h in GAT_LAYERS)
        elif args.app == "als":  # run_cg(1): two half-steps of (1 + 1 + 10) fused calls (als_conjugate_gradients.cpp:38-141)
            alg_bytes_per_step = 2 * 12 * (info["nS"] * (8 * args.r + 24) + 16 * args.r * info["localArows"] * c_now)
        else:
            alg_bytes_per_step = info["nS"] * (8 * args.r + 24) + 16 * args.r * info["localArows"] * c_now
        kern_ms_slowest = kern_ms
        if dist is not None:  # per-rank means for the kernel-level figures; the slowest rank's kernel time for what the step exposes
            t = torch.tensor([kern_ms, float(launches), float(alg_bytes_per_step)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            kern_ms, launches, alg_bytes_per_step = float(t[0]) / self.n, int(t[1]) // self.n, float(t[2]) / self.n
            kern_ms_slowest = self.max_over_ranks(kern_ms_slowest)
        self.barrier()
        check = None
        if not args.no_check:
            check = checks.check(self) if args.app == "vanilla" else checks.check_app(self)
            self.barrier()
        transport_kind = self.op.json_algorithm_info().get("transport", "?")  # (collective: every rank asks)
        ranks = self.world().identities()  # (collective) where every rank runs: pid, device ordinal, PCI bus id, the RCCL communicator's view
        return {"route": self.route, "elapsed": elapsed, "kern_ms": kern_ms, "kern_ms_slowest": kern_ms_slowest, "launches": launches, "prof_calls": prof_calls,
                "alg_bytes_per_step": alg_bytes_per_step, "check": check, "transport_kind": transport_kind, "ranks": ranks}


def compose_line(args, b, res, extra):
    """The JSON line of one complete measurement (rank 0)."""
    H, n = b.H, b.n
    tr, c_now, mode, q = res["route"]
    ms_per_step = res["elapsed"] / args.steps * 1e3
    value = b.nnz * args.r * args.steps / res["elapsed"]
    launches_per_step = max(1, res["launches"] // res["prof_calls"])
    dur = res["kern_ms"] / max(1, res["launches"]) * 1e-3  # average launch duration, seconds
    bytes_per_launch = res["alg_bytes_per_step"] / launches_per_step
    achieved_kernel = bytes_per_launch / dur if dur > 0 else 0.0
    # SURVEY 8(d), the whole step: the GLOBAL problem's algorithmic bytes over the driver-timed step (kernels + every exposed shift,
    # replication and wait — what the reference's elapsed time contains, benchmark_dist.cpp:117-149) against n x the HBM peak
    per_call = lambda r_: common.fused_bytes(b.nnz, r_, b.m)  # noqa: E731
    total_bytes = {"vanilla": per_call(args.r), "als": 2 * 12 * per_call(args.r),
                   "gat": sum(h * per_call(f) for _, f, h in GAT_LAYERS)}[args.app]
    achieved_step = total_bytes / (ms_per_step * 1e-3) / n  # B/s per GPU
    kernel_ms_per_step = res["kern_ms_slowest"] / res["prof_calls"]
    # one GPU: the dominant kernel's own rate (its launches are the step); several GPUs: the step's, exposed communication included
    achieved = achieved_kernel if n == 1 else achieved_step
    ring_mode_now = None if (n == 1 or mode == "none") else ("mesh fetch + mesh reduce-scatter" if mode == "fusion1" else mode)
    alg_now = "15d_fusion1" if mode == "fusion1" else args.alg
    step_is = {"vanilla": "fused SDDMM->SpMM (fusedSpMM, Amat)", "als": "one ALS step by batched CG (run_cg(1): 24 fused calls)",
               "gat": "one GAT forward pass (3 layers, 14 heads)"}[args.app]
    # where the ranks ran, from the ranks themselves: a line of N processes that shared fewer than N GPUs says so in its first sentence
    ranks = res.get("ranks") or []
    distinct = len({r["pci_bus_id"] for r in ranks}) or n
    where = ("%d x MI355X" % n) if distinct == n else ("%d PROCESSES SHARING %d x MI355X (time-sliced: not a scaling number)" % (n, distinct))
    how = "" if n == 1 else ", %s (%s)" % (
        {"rccl": "RCCL over xGMI", "ipc": "ipc-pull, copy engines", "ipc-kernel": "ipc-pull, pull kernel"}.get(tr, "transport: " + tr),
        {"relay": "relay ring", "mesh": "chunked mesh fetch", None: "replication only"}.get(ring_mode_now, ring_mode_now))
    out = {
        "backend": H.backend_name(),
        "metric": "fused SDDMM+SpMM nnz*R/s", "value": value, "unit": "nnz*R/s", "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic" if b.wl.kind != "mtx" else "file",
        "config": {"workload": "%s, R=%d, %s, %s c=%d on %s%s" % (b.wl.describe(b.nnz), args.r, step_is, alg_now, c_now, where, how),  # (<= 200 characters in the printed line)
                   "nnz": b.nnz, "M": b.m, "R": args.r, "algorithm": alg_now, "app": args.app, "c": c_now,
                   "transport": "none" if n == 1 else res["transport_kind"],
                   "transport_variant": None if n == 1 else tr, "ring_mode": ring_mode_now,
                   # Q symmetric chunks (a number) or the chunk heights (a comma list)
                   "mesh_chunks": (q if ring_mode_now == "mesh" else None),
                   "rccl_channels": (os.environ.get("NCCL_MAX_NCHANNELS", "default") if n > 1 else None),
                   # compute units masked off the compute stream (the library's default, 0, unless HNH_COMM_CUS says otherwise)
                   "comm_cus": int(os.environ.get("HNH_COMM_CUS", "0") or 0),
                   # set-up of the first route: the tuples (generated on the device, or parsed from the file and merged), then redistribution,
                   # CSR blocks and operands; parse_s is the first part alone
                   "setup_s": round(b.setup_s or 0.0, 2), "parse_s": round(b.parse_s or 0.0, 3),
                   # one record per rank: [rank, pid, device ordinal, PCI bus id] + RCCL's own [ncclCommCount, ncclCommUserRank, ncclCommCuDevice]
                   "distinct_devices": distinct,
                   "ranks": [[r["rank"], r["pid"], r["device_ordinal"], r["pci_bus_id"]] + ([r["comm_count"], r["comm_rank"], r["comm_device"]] if "comm_count" in r else [])
                             for r in ranks]},
        # `achieved` is an ALGORITHMIC rate (SURVEY 8d byte model / measured launch time), not DRAM utilisation: part of every
        # launch's gathers is served by the 256 MiB Infinity Cache, which sits behind the counters `traffic` comes from
        "roofline": {"bound": "hbm", "bound_detail": "HBM gather model, Infinity-Cache assisted (DESIGN.md 3.2)",
                     "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": None, "traffic_source": None,
                     # one GPU: the step IS the dominant kernel's launches; several: SURVEY 8d's whole-step figure, exposed communication counts
                     "frac_is": "frac_kernel (one GPU)" if n == 1 else "frac_step = total B_fused / ms_per_step / (n_gpus x 8 TB/s)",
                     "frac_kernel": achieved_kernel / HBM_PEAK, "achieved_kernel": achieved_kernel / 1e9,
                     "frac_step": achieved_step / HBM_PEAK, "achieved_step_per_gpu": achieved_step / 1e9,
                     "algorithmic_bytes_per_step_all_gpus": total_bytes,
                     # the slowest rank's event-bracketed kernel time per step, and what the timed step spends beyond it
                     "kernel_ms_per_step": kernel_ms_per_step, "exposed_comm_ms": ms_per_step - kernel_ms_per_step,
                     "kernel": ("row_kernel<fused>, 1 launch per Infinity-Cache panel of B" if n == 1 else
                                "row_kernel<fused>, 1 launch per visiting block (relay ring)" if ring_mode_now == "relay" else
                                "row_kernel<fused>, the rank's one block (replication only)" if ring_mode_now is None else
                                "row_kernel<sddmm> on row ranges + row_kernel<spmm> staging passes (15d_fusion1, mesh)" if mode == "fusion1" else
                                "row_kernel<fused>: own block + adaptive windowed passes over fetched blocks"),
                     # device time of a kernel CALL (HIP events around it on the compute stream) divided by the row-kernel launches it made
                     # (structure plans are cached: a steady-state call launches row kernels only)
                     "avg_launch_ms": dur * 1e3, "avg_launch_ms_is": "HIP-event time of the call / its row-kernel launches",
                     "traffic_rate": None,
                     "compulsory_bytes_per_call": 8 * args.r * (2 * b.m + b.m) + 24 * b.nnz,
                     "launches_per_step": launches_per_step, "algorithmic_bytes_per_launch": bytes_per_launch,
                     "model": "nnz*(8R+24) + 16*R*rows per fused call (SURVEY 8d), split evenly over launches"},
    }
    if res["check"] is not None:
        out["check"] = res["check"]
    out.update(extra)
    return out
