"""Experiment: does splitting the local block into column panels (so that a panel of B stays in the 256 MiB
Infinity Cache) speed up the fused kernel?  P panels = P launches over per-panel CSR structures."""
import argparse, ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_sddmm_amd import _kernels as K
from distributed_sddmm_amd import api as H

ap = argparse.ArgumentParser()
ap.add_argument("--logm", type=int, default=20); ap.add_argument("--ef", type=int, default=96); ap.add_argument("--r", type=int, default=128)
ap.add_argument("--panels", default="1,2,4,8,16,32"); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
m = 1 << a.logm; R = a.r
rows_i, cols_i = H.generate_er(m, m, m * a.ef, 12345)
nnz = len(rows_i)
ctx = K.Ctx(0); lib = ctx.lib
dA, dB, dOut = (K.DevArray(ctx, (m, R), np.float64) for _ in range(3))
lib.hnh_fill_f64(ctx.h, dA.ptr, m * R, 0.001, 0); lib.hnh_fill_f64(ctx.h, dB.ptr, m * R, 0.001, 0)
ev0, ev1 = C.c_void_p(), C.c_void_p(); lib.hnh_event_create(ctx.h, C.byref(ev0)); lib.hnh_event_create(ctx.h, C.byref(ev1))
alg = nnz * (8 * R + 24) + 16 * R * m
for P in [int(x) for x in a.panels.split(",")]:
    w = (m + P - 1) // P
    pan = cols_i // w
    blocks = []
    for p in range(P):
        sel = pan == p
        r, c = rows_i[sel], cols_i[sel]
        rp = np.zeros(m + 1, np.int64); np.add.at(rp, r + 1, 1); rp = np.cumsum(rp).astype(np.int32)
        blocks.append((ctx.upload(rp), ctx.upload(c.astype(np.int32)), K.DevArray(ctx, (max(len(c), 1),), np.float64)))
    arr = (K.CsrBlock * P)()
    for p, (drp, dc, dv) in enumerate(blocks):
        arr[p] = K.CsrBlock(drp.ptr, dc.ptr, dv.ptr, dB.ptr, dv.shape[0], 200)

    def run_blocks():
        for p, (drp, dc, dv) in enumerate(blocks):
            ctx.check(lib.hnh_fused_sddmm_spmm_csr_ex(ctx.h, m, drp.ptr, dc.ptr, dv.ptr, None, dA.ptr, dB.ptr, dOut.ptr, R, 1 | (2 if p == 0 else 0), dv.shape[0], 200, -1, 0), "fused")

    def run_multi():
        ctx.check(lib.hnh_fused_sddmm_spmm_csr_multi(ctx.h, m, P, C.byref(arr), dA.ptr, dOut.ptr, R, 3, 0), "multi")

    run = run_multi if os.environ.get("PROBE_MULTI") == "1" else run_blocks
    run(); ctx.sync(); ts = []
    for _ in range(a.iters):
        lib.hnh_event_record(ctx.h, ev0, 0); run(); lib.hnh_event_record(ctx.h, ev1, 0); lib.hnh_event_sync(ctx.h, ev1)
        ms = C.c_float(); lib.hnh_event_elapsed_ms(ctx.h, ev0, ev1, C.byref(ms)); ts.append(ms.value)
    t = float(np.median(ts)) * 1e-3
    print(("multi  " if os.environ.get("PROBE_MULTI") == "1" else "blocks ") + "panels %3d (%.0f MiB of B each): %.3f ms  %.3e nnz*R/s  %.1f%% of 8TB/s (algorithmic)" % (P, w * R * 8 / 2**20, t * 1e3, nnz * R / t, 100 * alg / t / 8e12), flush=True)
    for b in blocks:
        for d in b: d.free()
