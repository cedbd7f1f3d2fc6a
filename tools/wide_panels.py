"""Development tool: wide operands (R = 256 / 384 / 512) — Infinity-Cache panels x first-panel overwrite, measured.

The fused pass on config 2's matrix through the host layer's entry point (block descriptor + structure plan) for
    panels in {1, 2, 4, 8}   (HNH_PANEL_BYTES picks the count: a fresh kernel context per cell)
  x Out handling in {overwrite: the first panel's launch stores Out (HNH_FUSED_OUT_OVERWRITE), accumulate: Out is zeroed first and
    every launch read-modify-writes it}
with HIP-event times against the SURVEY 8(d) byte model and, with --traffic, the L2<->fabric bytes per call from two rocprofv3
counter passes of this very script (FETCH_SIZE, WRITE_SIZE, each in its own pass; 2 x FETCH_SIZE + WRITE_SIZE, the micro-architecture
guide's gfx950 correction) — what the review of round 4 asked for instead of the paper argument of DESIGN 3.2
(bench_heatmap.cpp:33 sweeps R to 448; the one time the reference prints is at R = 256).

    python tools/wide_panels.py [--r 256,384,512] [--panels 1,2,4,8] [--iters 3] [--traffic]
"""
import argparse
import ctypes as C
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cells(a):
    return [(r, p, ow) for r in a.r for p in a.panels for ow in ((1, 0) if not a.counting else (1,))]


def measure(a):
    from distributed_sddmm_amd import _kernels as K
    from distributed_sddmm_amd import api as H
    m = 1 << a.logm
    rows_i, cols_i = H.generate_er(m, m, m * a.ef)
    nnz = len(rows_i)
    rowptr = np.concatenate(([0], np.cumsum(np.bincount(rows_i, minlength=m)))).astype(np.int32)
    base = K.Ctx(0)
    d_rowptr, d_c = base.upload(rowptr), base.upload(cols_i.astype(np.int32))
    del rows_i, cols_i
    mx = C.c_int()
    base.check(base.lib.hnh_csr_max_row_nnz(base.h, m, d_rowptr.ptr, C.byref(mx), 0), "max_row")
    base.sync()
    out = []
    for R in a.r:
        dv = K.DevArray(base, (nnz,), np.float64)
        dA, dB, dOut = (K.DevArray(base, (m, R), np.float64) for _ in range(3))
        base.lib.hnh_fill_f64(base.h, dA.ptr, m * R, 0.001, 0)
        base.lib.hnh_fill_f64(base.h, dB.ptr, m * R, 0.001, 0)
        base.sync()
        model = nnz * (8 * R + 24) + 16 * R * m
        for panels in a.panels:
            os.environ["HNH_PANEL_BYTES"] = str(int(m * R * 8 / panels) + 1)
            os.environ["HNH_MAX_PANELS"] = "8"
            if panels == 1:
                os.environ["HNH_NO_PANELS"] = "1"
            else:
                os.environ.pop("HNH_NO_PANELS", None)
            ctx = K.Ctx(0)
            lib = ctx.lib
            plan = C.c_void_p()
            ctx.check(lib.hnh_csr_plan_create(ctx.h, C.byref(plan)), "plan")
            blk = K.CsrBlock(m, nnz, m, mx.value, 0, d_rowptr.ptr, d_c.ptr, plan)
            ev0, ev1 = C.c_void_p(), C.c_void_p()
            lib.hnh_event_create(ctx.h, C.byref(ev0))
            lib.hnh_event_create(ctx.h, C.byref(ev1))
            for overwrite in ((1, 0) if not a.counting else (1,)):
                def call():
                    if not overwrite:
                        ctx.check(lib.hnh_memset(ctx.h, dOut.ptr, 0, 8 * m * R, 0), "memset")
                    ctx.check(lib.hnh_fused_sddmm_spmm_csr_p(ctx.h, C.byref(blk), dv.ptr, None, dA.ptr, dB.ptr, dOut.ptr, R, 1 | (2 if overwrite else 0),
                                                             None, None, 0), "fused_p")
                call()
                ctx.sync()
                ts = []
                for _ in range(1 if a.counting else a.iters):
                    lib.hnh_event_record(ctx.h, ev0, 0)
                    call()
                    lib.hnh_event_record(ctx.h, ev1, 0)
                    lib.hnh_event_sync(ctx.h, ev1)
                    ms = C.c_float()
                    lib.hnh_event_elapsed_ms(ctx.h, ev0, ev1, C.byref(ms))
                    ts.append(ms.value)
                t = float(np.median(ts))
                out.append({"R": R, "panels": panels, "out": "overwrite" if overwrite else "zero + accumulate", "ms": t,
                            "model_bytes": model, "frac": model / (t * 1e-3) / 8e12})
                if not a.counting:
                    print("R=%d panels=%d Out %-17s %.3f ms  %.1f%% of 8 TB/s by the 8(d) model" % (R, panels, out[-1]["out"], t, 100 * out[-1]["frac"]), flush=True)
            ctx.check(lib.hnh_csr_plan_destroy(ctx.h, plan), "plan destroy")
            ctx.close()
        for d in (dv, dA, dB, dOut):
            d.free()
    return out


def counter_pass(a, counter):
    """per fused CALL of every cell (overwrite only): the counter summed over the call's row-kernel launches, in dispatch order"""
    prof = shutil.which("rocprofv3")
    if prof is None:
        return None
    d = tempfile.mkdtemp(prefix="hnh_wide_", dir="/tmp")
    try:
        cmd = [prof, "--pmc", counter, "-d", d, "-o", "pass", "--", sys.executable, os.path.abspath(__file__), "--counting", "--r", ",".join(map(str, a.r)),
               "--panels", ",".join(map(str, a.panels)), "--logm", str(a.logm), "--ef", str(a.ef)]
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, check=True)
        vals = []
        for db in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
            cur = sqlite3.connect(db).cursor()
            vals += list(cur.execute("select dispatch_id, value from counters_collection where kernel_name like ? and counter_name = ? order by dispatch_id",
                                     ("%::row_kernel<%", counter)))
        vals = [v for _, v in sorted(vals)]
        per_call, i = [], 0
        for (r, p, ow) in cells(argparse.Namespace(**dict(vars(a), counting=True))):
            # per cell: one warm-up call + one counted call, `p` launches each
            if i + 2 * p > len(vals):
                return None
            per_call.append(sum(vals[i + p:i + 2 * p]))
            i += 2 * p
        return per_call if i == len(vals) else None
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("counter pass %s failed: %s\n" % (counter, e))
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logm", type=int, default=20)
    ap.add_argument("--ef", type=int, default=96)
    ap.add_argument("--r", default="256,384,512")
    ap.add_argument("--panels", default="1,2,4,8")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--traffic", action="store_true")
    ap.add_argument("--counting", action="store_true", help=argparse.SUPPRESS)  # internal: the profiled child (one warm-up + one call per cell)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    a.r = [int(x) for x in str(a.r).split(",")]
    a.panels = [int(x) for x in str(a.panels).split(",")]
    table = measure(a)
    if a.counting:
        return
    if a.traffic:
        fetch, write = counter_pass(a, "FETCH_SIZE"), counter_pass(a, "WRITE_SIZE")
        if fetch and write:
            k = 0
            for row in table:
                if row["out"] != "overwrite":
                    continue
                row["traffic_bytes"] = 2.0 * fetch[k] * 1024.0 + write[k] * 1024.0
                row["traffic_over_model"] = row["traffic_bytes"] / row["model_bytes"]
                print("R=%d panels=%d: counter traffic %.2f GB per call = %.3f x the 8(d) model (raw KB: fetch %.0f, write %.0f)"
                      % (row["R"], row["panels"], row["traffic_bytes"] / 1e9, row["traffic_over_model"], fetch[k], write[k]), flush=True)
                k += 1
        else:
            print("counter passes unavailable", flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()
