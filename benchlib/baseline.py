"""The two legs of the line that are not the GPU measurement: `cpu_baseline` — the reference itself (oracle/_ref/ref_driver = its
unmodified sources + MKL/MPICH) timed on this box's host cores — and `roofline.traffic` — rocprofv3 counter passes of this very
command.  Both run outside the timed region, after the GPU line is in hand.  This module is the ONLY place bench.py touches oracle/."""
import json
import os
import socket
import sys
import time

from . import common


def _cache_path(args):
    key = "er%d_ef%d_r%d_s%d_t%d" % (args.logm, args.edge_factor, args.r, args.cpu_logm, args.cpu_trials)
    return os.path.join(os.environ.get("HNH_BENCH_CACHE_DIR", "/tmp"), "hnh_cpu_baseline_%s_%s.json" % (socket.gethostname(), key))


def cached_cpu_baseline(args):
    """The record an earlier run ON THIS HOST left for the same workload (the driver runs N = 1, 2, 4, 8 back to back), or None."""
    try:
        with open(_cache_path(args)) as f:
            rec = json.load(f)
        if rec.get("value"):
            rec["cached"] = "measured by an earlier run of bench.py on this host %.0f s ago (%s)" % (time.time() - rec.pop("_stamp", time.time()), _cache_path(args))
            return rec
    except (OSError, ValueError):
        pass
    return None


def store_cpu_baseline(args, rec):
    if not rec or not rec.get("value"):
        return
    try:
        tmp = _cache_path(args) + ".%d" % os.getpid()
        with open(tmp, "w") as f:
            json.dump(dict(rec, _stamp=time.time()), f)
        os.replace(tmp, _cache_path(args))
    except OSError:
        pass


def cpu_baseline_for_line(args, n, budget):
    """`cpu_baseline` of a line, whatever N: one GPU runs the legs (and leaves the record for the runs that follow on this host);
    several GPUs quote that record, or — none on this host — run the bounded SAMPLE leg only, if the time budget has room for it."""
    try:
        if n == 1:
            rec = cpu_baseline(args)
            store_cpu_baseline(args, rec)
            return rec
        rec = cached_cpu_baseline(args)
        if rec is not None:
            return rec
        if not budget.fits(120.0):
            return {"value": None, "unit": "nnz*R/s", "cores": os.cpu_count(), "kind": "reference",
                    "sample": "NOT RUN: no record of an N = 1 run on this host and %.0f s left of --budget-s" % budget.left()}
        import argparse
        return cpu_baseline(argparse.Namespace(**dict(vars(args), cpu_full=False)), quick=True)
    except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
        return {"value": None, "unit": "nnz*R/s", "cores": os.cpu_count(), "kind": "reference", "sample": "FAILED: %s" % str(e)[:300]}


def cpu_baseline(args, quick=False):
    """The reference timed on the host cores, bounded (about 30-40 s): ONE run of the compiled reference on a sample of the GPU line's
    workload (ER 2^cpu_logm, same edge factor and R; one MPI rank), which behind one set-up times the loop of benchmark_dist.cpp:117-149
    (1 warm-up + cpu_trials fused calls) at a few OpenMP/MKL thread counts, two batches per count, the better batch counted; `value` is
    the best point, `cores` the threads it used.  --cpu-full adds the winner once at the GPU line's own size (about a minute more, nearly
    all of it the reference's set-up) and reports that as `value` instead."""
    import numpy as np
    from distributed_sddmm_amd import api as H
    from oracle import refrun as RR
    RR.PREEXEC = common.unblock_signals
    ncpu = os.cpu_count() or 1
    m = 1 << args.cpu_logm
    if RR.available():
        t0 = time.perf_counter()
        rows, cols = H.generate_er(m, m, m * args.edge_factor, 12345)
        # The reference does not scale with the thread count on big hosts (2 x EPYC 9575F: 32 threads beat 64/128/256 and one MPI rank
        # beats 4 .. 32 on every box so far, profiles/archive/r01_cpu_baseline_sweep.log, BENCH_r04/r05), so a few counts are tried
        counts = sorted({min(ncpu, 16), min(ncpu, 32), min(ncpu, 64)}) if not quick else [min(ncpu, 32)]
        res = RR.sweep(m, m, rows, cols, args.r, "15d_fusion2", 1, 1, True, args.cpu_trials, 2, counts, timeout=300.0)
        best = max(res["points"], key=lambda pt: pt["nnz_R_per_s"])
        threads, comp = best["threads"], best.get("computation_time", 0.0)
        out = {"value": best["nnz_R_per_s"], "unit": "nnz*R/s", "cores": threads, "kind": "reference",
               "sample": "ER 2^%d ef %d (%.3g nnz) R=%d fused, 1 rank x %d thr, best of 2 x %d calls; leg %.0f s" % (
                   args.cpu_logm, args.edge_factor, len(rows), args.r, threads, args.cpu_trials, time.perf_counter() - t0),
               "threads_used": threads, "host_hw_threads": ncpu, "ranks": 1, "elapsed_s": best["elapsed"],
               "thread_sweep": {str(pt["threads"]): float("%.4g" % pt["nnz_R_per_s"]) for pt in res["points"]},
               "batches_s": {str(pt["threads"]): [round(x, 3) for x in pt["batches"]] for pt in res["points"]},
               "kernel_only_value": (len(rows) * args.r * args.cpu_trials / comp) if comp > 0 else None}
        if getattr(args, "cpu_full", False) and args.logm != args.cpu_logm:
            try:
                t1 = time.perf_counter()
                mf = 1 << args.logm
                rows, cols = H.generate_er(mf, mf, mf * args.edge_factor, 12345)
                full = RR.bench(mf, mf, rows, cols, args.r, "15d_fusion2", 1, 1, True, args.cpu_trials, threads=threads, timeout=900.0)
                compf = full["perf_stats"].get("Computation Time", 0.0)
                out.update({"sample_value": out["value"], "sample_workload": out["sample"],
                            "value": full["nnz_R_per_s"], "elapsed_s": full["elapsed"],
                            "kernel_only_value": (len(rows) * args.r * args.cpu_trials / compf) if compf > 0 else None,
                            "sample": "the GPU line's size: ER 2^%d ef %d (%.3g nnz) R=%d fused, 1 rank x %d thr, %d calls; leg %.0f s" % (
                                args.logm, args.edge_factor, len(rows), args.r, threads, args.cpu_trials, time.perf_counter() - t1)})
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
            cmd = [prof, "--pmc", counter, "-d", d, "-o", "pass", "--", sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
                   "--warmup", "1", "--no-cpu-baseline", "--no-check", "--no-live-traffic", "--no-secondary", "--logm", str(args.logm), "--edge-factor",
                   str(args.edge_factor), "--r", str(args.r), "--alg", args.alg, "--workload", args.workload, "--app", args.app]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=420, check=True,
                           preexec_fn=common.unblock_signals)
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
