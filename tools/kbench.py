"""Kernel-level micro benchmark (development tool): fused / sddmm / spmm on an ER block, HIP-event timed."""
import argparse
import ctypes as C
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_sddmm_amd import _kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logm", type=int, default=20)
    ap.add_argument("--ef", type=int, default=96)
    ap.add_argument("--r", default="128", help="embedding width, or a comma list to sweep")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--ops", default="fused,sddmm,spmm")
    ap.add_argument("--rmat", action="store_true", help="skewed R-MAT graph instead of Erdos-Renyi")
    ap.add_argument("--panels", action="store_true", help="pass the hints (nnz, longest row, cols) so that the library may "
                    "run the pass as Infinity-Cache panels")
    ap.add_argument("--sort-rows", action="store_true", help="experiment: reorder the sparse rows by length (rows of one wave then have "
                    "equal trip counts) — isolates the intra-wave imbalance of the narrow-row kernels")
    a = ap.parse_args()
    t0 = time.time()
    if a.rmat:
        from distributed_sddmm_amd import api as H
        rows_i, cols_i = H.generate_rmat(a.logm, (1 << a.logm) * a.ef)
        deg = np.bincount(rows_i, minlength=1 << a.logm)
        print("R-MAT: max row degree %d, mean %.1f, rows with > 4096 nnz: %d" % (deg.max(), deg.mean(), int((deg > 4096).sum())), flush=True)
    else:
        from distributed_sddmm_amd import api as H
        rows_i, cols_i = H.generate_er(1 << a.logm, 1 << a.logm, (1 << a.logm) * a.ef)
    m = 1 << a.logm
    nnz = len(rows_i)
    deg = np.bincount(rows_i, minlength=m)
    if a.sort_rows:
        order = np.argsort(deg, kind="stable")          # new row k = old row order[k]
        old_ptr = np.concatenate(([0], np.cumsum(deg)))
        deg = deg[order]
        src = np.concatenate([np.arange(old_ptr[r], old_ptr[r + 1]) for r in order]) if m <= (1 << 12) else None
        if src is None:  # vectorised gather of the row pieces
            new_ptr = np.concatenate(([0], np.cumsum(deg)))
            src = np.repeat(old_ptr[order] - new_ptr[:-1], deg) + np.arange(nnz)
        cols_i = cols_i[src]
    rowptr = np.concatenate(([0], np.cumsum(deg))).astype(np.int32)
    print("generated nnz=%d in %.1fs" % (nnz, time.time() - t0), flush=True)
    ctx = K.Ctx(0)
    lib = ctx.lib
    d_rowptr, d_c = ctx.upload(rowptr), ctx.upload(cols_i.astype(np.int32))
    del rows_i, cols_i
    for R in [int(x) for x in str(a.r).split(",")]:
        dv = K.DevArray(ctx, (nnz,), np.float64)
        dA, dB, dOut = (K.DevArray(ctx, (m, R), np.float64) for _ in range(3))
        lib.hnh_fill_f64(ctx.h, dA.ptr, m * R, 0.001, 0)
        lib.hnh_fill_f64(ctx.h, dB.ptr, m * R, 0.001, 0)
        lib.hnh_fill_f64(ctx.h, dv.ptr, nnz, 0.0, 0)
        ev0, ev1 = C.c_void_p(), C.c_void_p()
        lib.hnh_event_create(ctx.h, C.byref(ev0)); lib.hnh_event_create(ctx.h, C.byref(ev1))

        def timed(fn, name, bytes_alg):
            fn(); ctx.sync()
            ts = []
            for _ in range(a.iters):
                lib.hnh_event_record(ctx.h, ev0, 0)
                fn()
                lib.hnh_event_record(ctx.h, ev1, 0)
                lib.hnh_event_sync(ctx.h, ev1)
                ms = C.c_float()
                lib.hnh_event_elapsed_ms(ctx.h, ev0, ev1, C.byref(ms))
                ts.append(ms.value)
            t = float(np.median(ts)) * 1e-3
            frac = bytes_alg / t / 8e12
            # a model that charges memory for bytes a cache serves (the COO kernel's row-operand gathers; any operand that fits the
            # 256 MiB Infinity Cache) can exceed the HBM peak: such a line is a rate of the MODEL, not of HBM
            note = "  [cache-assisted: the byte model counts gathers that L1/L2/Infinity Cache serve]" if frac > 1.0 else ""
            print("%-16s R=%d  %.3f ms  %.3e nnz*R/s  alg %.2f GB -> %.2f TB/s (%.1f%% of 8 TB/s)%s" % (
                name, R, t * 1e3, nnz * R / t, bytes_alg / 1e9, bytes_alg / t / 1e12, 100 * frac, note), flush=True)

        ops = a.ops.split(",")
        if a.panels:
            mx = C.c_int()
            ctx.check(lib.hnh_csr_max_row_nnz(ctx.h, m, d_rowptr.ptr, C.byref(mx), 0), "max_row")
            timed(lambda: ctx.check(lib.hnh_fused_sddmm_spmm_csr_ex(ctx.h, m, d_rowptr.ptr, d_c.ptr, dv.ptr, None, dA.ptr, dB.ptr, dOut.ptr, R, 3,
                                                                    nnz, mx.value, m, 0), "fused"), "fusedP", nnz * (8 * R + 24) + 16 * R * m)
            timed(lambda: ctx.check(lib.hnh_sddmm_csr_ex(ctx.h, m, d_rowptr.ptr, d_c.ptr, dv.ptr, dA.ptr, dB.ptr, R, nnz, mx.value, m, 0), "sddmm"),
                  "sddmmP", nnz * (8 * R + 20) + 8 * R * m)
            timed(lambda: ctx.check(lib.hnh_spmm_csr_ex(ctx.h, m, d_rowptr.ptr, d_c.ptr, dv.ptr, dB.ptr, dOut.ptr, R, nnz, mx.value, m, 0), "spmm"),
                  "spmmP", nnz * (8 * R + 12) + 16 * R * m)
        if "plan" in ops:  # the host layer's way: block descriptor + structure plan (steady state: row kernels only), SDDMM storing
            mx = C.c_int()
            ctx.check(lib.hnh_csr_max_row_nnz(ctx.h, m, d_rowptr.ptr, C.byref(mx), 0), "max_row")
            plan = C.c_void_p()
            ctx.check(lib.hnh_csr_plan_create(ctx.h, C.byref(plan)), "plan")
            blk = K.CsrBlock(m, nnz, m, mx.value, 0, d_rowptr.ptr, d_c.ptr, plan)
            timed(lambda: ctx.check(lib.hnh_fused_sddmm_spmm_csr_p(ctx.h, C.byref(blk), dv.ptr, None, dA.ptr, dB.ptr, dOut.ptr, R, 3, None, None, 0), "fused_p"),
                  "fused/plan", nnz * (8 * R + 24) + 16 * R * m)
            timed(lambda: ctx.check(lib.hnh_sddmm_csr_p(ctx.h, C.byref(blk), dv.ptr, dA.ptr, dB.ptr, R, 0, None, 0), "sddmm_p"),
                  "sddmm/plan", nnz * (8 * R + 20) + 8 * R * m)
            timed(lambda: ctx.check(lib.hnh_sddmm_csr_p(ctx.h, C.byref(blk), dv.ptr, dA.ptr, dB.ptr, R, 1, None, 0), "sddmm_p"),
                  "sddmm/plan,store", nnz * (8 * R + 20) + 8 * R * m)
            timed(lambda: ctx.check(lib.hnh_spmm_csr_p(ctx.h, C.byref(blk), dv.ptr, dB.ptr, dOut.ptr, R, None, 0), "spmm_p"),
                  "spmm/plan", nnz * (8 * R + 12) + 16 * R * m)
            ctx.check(lib.hnh_csr_plan_destroy(ctx.h, plan), "plan destroy")
        if "fold" in ops:  # a whole sddmmA: storing SDDMM + the closing Hadamard pass, against the SDDMM with the Hadamard folded in
            mx = C.c_int()
            ctx.check(lib.hnh_csr_max_row_nnz(ctx.h, m, d_rowptr.ptr, C.byref(mx), 0), "max_row")
            plan = C.c_void_p()
            ctx.check(lib.hnh_csr_plan_create(ctx.h, C.byref(plan)), "plan")
            blk = K.CsrBlock(m, nnz, m, mx.value, 0, d_rowptr.ptr, d_c.ptr, plan)
            d_sv, d_res = K.DevArray(ctx, (nnz,), np.float64), K.DevArray(ctx, (nnz,), np.float64)
            ctx.check(lib.hnh_fill_f64(ctx.h, d_sv.ptr, nnz, 0.5, 0), "fill")
            model = nnz * (8 * R + 20) + 8 * R * m  # (the same SDDMM model for both: what a caller gets per sddmmA)

            def two_pass():
                ctx.check(lib.hnh_sddmm_csr_p(ctx.h, C.byref(blk), dv.ptr, dA.ptr, dB.ptr, R, 1, None, 0), "sddmm_p")
                ctx.check(lib.hnh_hadamard_f64(ctx.h, d_res.ptr, d_sv.ptr, dv.ptr, nnz, 0), "hadamard")

            timed(two_pass, "sddmm+hadamard", model)
            timed(lambda: ctx.check(lib.hnh_sddmm_csr_ps(ctx.h, C.byref(blk), d_res.ptr, d_sv.ptr, dA.ptr, dB.ptr, R, 1, None, 0), "sddmm_ps"),
                  "sddmm folded", model)
            ctx.check(lib.hnh_csr_plan_destroy(ctx.h, plan), "plan destroy")
            d_sv.free(); d_res.free()
        if "fused" in ops:
            timed(lambda: ctx.check(lib.hnh_fused_sddmm_spmm_csr(ctx.h, m, d_rowptr.ptr, d_c.ptr, dv.ptr, None, dA.ptr, dB.ptr,
                                                                 dOut.ptr, R, 3, 0), "fused"), "fused", nnz * (8 * R + 24) + 16 * R * m)
        if "sddmm" in ops:
            timed(lambda: ctx.check(lib.hnh_sddmm_csr(ctx.h, m, d_rowptr.ptr, d_c.ptr, dv.ptr, dA.ptr, dB.ptr, R, 0), "sddmm"),
                  "sddmm", nnz * (8 * R + 20) + 8 * R * m)
        if "coo" in ops:
            d_r = K.DevArray(ctx, (nnz,), np.int32)
            ctx.check(lib.hnh_expand_rowptr(ctx.h, m, d_rowptr.ptr, d_r.ptr, 0), "expand")
            timed(lambda: ctx.check(lib.hnh_sddmm_coo(ctx.h, nnz, d_r.ptr, d_c.ptr, dv.ptr, dA.ptr, dB.ptr, R, 0), "coo"),
                  "coo", nnz * (16 * R + 24))
        if "ew" in ops:
            n = m * R
            timed(lambda: ctx.check(lib.hnh_fill_f64(ctx.h, dOut.ptr, n, 1.5, 0), "fill"), "fill", 8 * n)
            timed(lambda: ctx.check(lib.hnh_hadamard_f64(ctx.h, dOut.ptr, dA.ptr, dB.ptr, n, 0), "hadamard"), "hadam", 24 * n)
            timed(lambda: ctx.check(lib.hnh_axpy_f64(ctx.h, dOut.ptr, dA.ptr, 0.5, n, 0), "axpy"), "axpy", 24 * n)
            dvec = K.DevArray(ctx, (m,), np.float64)
            timed(lambda: ctx.check(lib.hnh_rowdot_f64(ctx.h, dA.ptr, dB.ptr, dvec.ptr, m, R, 0), "rowdot"), "rowdot", 16 * n)
            timed(lambda: ctx.check(lib.hnh_row_scale_add_f64(ctx.h, dOut.ptr, None, 1.0, dA.ptr, dvec.ptr, -1.0, m, R, 0), "rsa"), "rsadd", 24 * n)
            timed(lambda: ctx.check(lib.hnh_memcpy(ctx.h, dOut.ptr, dA.ptr, 8 * n, K.D2D, 0), "copy"), "d2dcp", 16 * n)
            timed(lambda: ctx.check(lib.hnh_memset(ctx.h, dOut.ptr, 0, 8 * n, 0), "memset"), "mset", 8 * n)
        if "spmm" in ops:
            timed(lambda: ctx.check(lib.hnh_spmm_csr(ctx.h, m, d_rowptr.ptr, d_c.ptr, dv.ptr, dB.ptr, dOut.ptr, R, 0), "spmm"),
                  "spmm", nnz * (8 * R + 12) + 16 * R * m)

        for d in (dv, dA, dB, dOut):
            d.free()


if __name__ == "__main__":
    main()
