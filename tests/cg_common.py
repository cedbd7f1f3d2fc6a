"""Shared body of the row-epilogue tests of the fused SDDMM -> SpMM call (include/hnh_kernels.h): hnh_cg_update — the call also
performs the rest of one batched-CG iteration on every finished row — and relu_dst — the finished row leaves through a ReLU into
a column block of a wider matrix (a GAT head's output, gat.hpp:96-101).  The same checks run against the oracle's C test
double on the CPU and against the HIP library on the GPU; the expectation is the reference's sequence of whole-matrix
statements (als_conjugate_gradients.cpp:82-139) in numpy."""
import ctypes as C

import numpy as np

from distributed_sddmm_amd import _kernels as K

TOL = 1e-11


def rel(x, y):
    return float(np.max(np.abs(x - y)) / max(float(np.max(np.abs(y))), 1e-300)) if x.size else 0.0


def addr(d):
    """device pointer of an uploaded array as an int (the GPU handle holds an int, the CPU one a c_void_p)"""
    return d.ptr.value if hasattr(d.ptr, "value") else d.ptr


def block(rows, cols, seed, hubs):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 24, rows)
    lens[rng.integers(0, rows, rows // 10)] = 0  # empty rows: Mp = lambda p there
    if hubs:  # hub rows are completed by several groups (atomically combined segments): the epilogue becomes its own launch
        lens[1], lens[rows // 2] = 2500, 1100
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cidx = np.concatenate([np.sort(rng.choice(cols, n, replace=False)) for n in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    return rowptr, np.repeat(np.arange(rows), lens), cidx


def reference_iteration(rowptr, ridx, cidx, p, Y, x, r, rsold, lam, eps):
    """computeQueries (S == 1) followed by als_conjugate_gradients.cpp:91-139, statement by statement."""
    vals = np.einsum("ij,ij->i", p[ridx], Y[cidx])                   # SDDMM
    Mp = np.zeros_like(p)
    np.add.at(Mp, ridx, vals[:, None] * Y[cidx])                      # SpMM
    Mp += lam * p                                                     # :282
    bdot = np.einsum("ij,ij->i", p, Mp)                               # :91
    bdot = bdot + eps                                                 # :99
    rsold = rsold + eps                                               # :100
    alpha = rsold / bdot                                              # :102
    x = x + alpha[:, None] * p                                        # :112-117
    r = r - alpha[:, None] * Mp                                       # :118
    rsnew = np.einsum("ij,ij->i", r, r)                               # :120
    coeffs = rsnew / rsold                                            # :136
    p = r + coeffs[:, None] * p                                       # :137
    return vals, Mp, x, r, p, rsnew


def run(api, R, hubs=False, windows=0, standalone=False):
    """windows > 0: the block is processed as that many column windows (hnh_fused_sddmm_spmm_csr_w), CG update with the last.
    standalone: plain fused call, then hnh_row_epilogue_x."""
    lib, h = api.lib, api.h
    rows, cols = (60, 6000) if hubs else (173, 400)
    rowptr, ridx, cidx = block(rows, cols, R * 7 + windows + hubs, hubs)
    nnz = len(cidx)
    rng = np.random.default_rng(R + 100)
    p, Y, x, r = (rng.uniform(-1, 1, (n, R)) for n in (rows, cols, rows, rows))
    rsold = np.einsum("ij,ij->i", r, r) * rng.uniform(0.5, 1.5, rows)
    lam, eps = 1e-3, 1e-8
    want = reference_iteration(rowptr, ridx, cidx, p, Y, x, r, rsold, lam, eps)

    d_rp, d_c, dv, dp, dY, dx, dr, drs, dMp = (api.upload(a) for a in (rowptr, cidx, np.zeros(nnz), p, Y, x, r, rsold, np.full((rows, R), 7.0)))
    cg = K.CgUpdate(addr(dx), addr(dr), addr(dp), addr(drs), eps)
    ex = K.FusedExtras(0.0, lam, None, C.pointer(cg))
    OW = K.FUSED_VALUES_OVERWRITE | K.FUSED_OUT_OVERWRITE
    maxrow = int(np.diff(rowptr).max()) if rows else 0
    if standalone:
        api.check(lib.hnh_fused_sddmm_spmm_csr_x(h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dp.ptr, dY.ptr, dMp.ptr, R, OW, nnz, maxrow, cols, None, 0), "fused")
        api.check(lib.hnh_row_epilogue_x(h, dMp.ptr, dp.ptr, C.byref(ex), rows, R, 0), "row_epilogue_x")
    elif windows:
        bounds = np.linspace(0, cols, windows + 1).astype(np.int32)[1:-1]
        dsplit = api.upload(np.zeros(max(1, len(bounds)) * rows, np.int32))
        api.check(lib.hnh_csr_window_bounds(h, rows, d_rp.ptr, d_c.ptr, len(bounds), bounds.ctypes.data_as(C.c_void_p), dsplit.ptr, 0), "bounds")
        base = addr(dsplit)
        for q in range(windows):
            last = q == windows - 1
            win = K.CsrWindow(None if q == 0 else base + (q - 1) * rows * 4, None if last else base + q * rows * 4, 1 if last else 0)
            api.check(lib.hnh_fused_sddmm_spmm_csr_w(h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dp.ptr, dY.ptr, dMp.ptr, R,
                                                     K.FUSED_VALUES_OVERWRITE | (K.FUSED_OUT_OVERWRITE if q == 0 else 0), nnz, maxrow,
                                                     C.byref(ex) if last else None, C.byref(win), 0), "fused_w")
        dsplit.free()
    else:
        api.check(lib.hnh_fused_sddmm_spmm_csr_x(h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dp.ptr, dY.ptr, dMp.ptr, R, OW, nnz, maxrow, cols,
                                                 C.byref(ex), 0), "fused_x with cg")
    api.check(lib.hnh_stream_sync(h, 0), "sync")
    got = (dv.get(), dMp.get().reshape(rows, R), dx.get().reshape(rows, R), dr.get().reshape(rows, R), dp.get().reshape(rows, R), drs.get())
    for name, g, w in zip(("values", "Mp", "x", "r", "p", "rsold"), got, want):
        assert rel(np.asarray(g), w) <= TOL, (name, R, hubs, windows, standalone)

    # caller errors: p must be the row operand; operands must not alias
    bad = K.CgUpdate(cg.x, cg.r, cg.x, cg.rsold, eps)
    exb = K.FusedExtras(0.0, lam, None, C.pointer(bad))
    assert lib.hnh_fused_sddmm_spmm_csr_x(h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dp.ptr, dY.ptr, dMp.ptr, R, OW, nnz, maxrow, cols, C.byref(exb), 0) != 0
    bad = K.CgUpdate(cg.x, cg.x, cg.p, cg.rsold, eps)
    exb = K.FusedExtras(0.0, lam, None, C.pointer(bad))
    assert lib.hnh_row_epilogue_x(h, dMp.ptr, dp.ptr, C.byref(exb), rows, R, 0) != 0
    for d in (d_rp, d_c, dv, dp, dY, dx, dr, drs, dMp):
        d.free()


def run_relu(api, R, hubs=False, windows=0, heads=3, head=1):
    """LeakyReLU between the halves, ReLU on the way out, into column block `head` of a rows x (heads * R) matrix."""
    lib, h = api.lib, api.h
    rows, cols = (60, 6000) if hubs else (173, 173)
    rowptr, ridx, cidx = block(rows, cols, R * 5 + windows + hubs, hubs)
    nnz = len(cidx)
    rng = np.random.default_rng(R + 200)
    A = rng.uniform(-1, 1, (cols, R)) / np.sqrt(R)
    X = A[:rows] if not hubs else rng.uniform(-1, 1, (rows, R)) / np.sqrt(R)
    alpha = 0.2
    vals = np.einsum("ij,ij->i", X[ridx], A[cidx])
    vals = np.where(vals > 0, vals, alpha * vals)
    Hm = np.zeros((rows, R))
    np.add.at(Hm, ridx, vals[:, None] * A[cidx])
    wide0 = rng.uniform(-1, 1, (rows, heads * R))
    want = wide0.copy()
    want[:, head * R:(head + 1) * R] = np.maximum(Hm, 0.0)

    d_rp, d_c, dv, dX, dA, dH, dwide = (api.upload(a) for a in (rowptr, cidx, np.zeros(nnz), X, A, np.full((rows, R), 7.0), wide0))
    ex = K.FusedExtras(alpha, 0.0, None, None, addr(dwide) + head * R * 8, heads * R)
    flags = K.FUSED_VALUES_OVERWRITE | K.FUSED_LEAKY_RELU
    maxrow = int(np.diff(rowptr).max())
    if windows:
        bounds = np.linspace(0, cols, windows + 1).astype(np.int32)[1:-1]
        dsplit = api.upload(np.zeros(max(1, len(bounds)) * rows, np.int32))
        api.check(lib.hnh_csr_window_bounds(h, rows, d_rp.ptr, d_c.ptr, len(bounds), bounds.ctypes.data_as(C.c_void_p), dsplit.ptr, 0), "bounds")
        base = addr(dsplit)
        act = K.FusedExtras(alpha, 0.0, None, None, None, 0)
        for q in range(windows):
            last = q == windows - 1
            win = K.CsrWindow(None if q == 0 else base + (q - 1) * rows * 4, None if last else base + q * rows * 4, 1 if last else 0)
            api.check(lib.hnh_fused_sddmm_spmm_csr_w(h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dA.ptr, dH.ptr, R,
                                                     flags | (K.FUSED_OUT_OVERWRITE if q == 0 else 0), nnz, maxrow,
                                                     C.byref(ex if last else act), C.byref(win), 0), "fused_w relu")
        dsplit.free()
    else:
        api.check(lib.hnh_fused_sddmm_spmm_csr_x(h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dA.ptr, dH.ptr, R, flags | K.FUSED_OUT_OVERWRITE,
                                                 nnz, maxrow, cols, C.byref(ex), 0), "fused_x relu")
    api.check(lib.hnh_stream_sync(h, 0), "sync")
    assert rel(np.asarray(dv.get()), vals) <= TOL
    got = np.asarray(dwide.get()).reshape(rows, heads * R)
    assert rel(got, want) <= TOL, (R, hubs, windows)
    assert np.array_equal(got[:, :head * R], wide0[:, :head * R]) and np.array_equal(got[:, (head + 1) * R:], wide0[:, (head + 1) * R:])
    # caller errors: together with cg; without a pitch
    cg = K.CgUpdate(addr(dH), addr(dwide), addr(dX), addr(dv), 0.0)
    bad = K.FusedExtras(alpha, 0.0, None, C.pointer(cg), addr(dwide), heads * R)
    assert lib.hnh_fused_sddmm_spmm_csr_x(h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dA.ptr, dH.ptr, R, flags, nnz, maxrow, cols, C.byref(bad), 0) != 0
    bad = K.FusedExtras(alpha, 0.0, None, None, addr(dwide), 0)
    assert lib.hnh_fused_sddmm_spmm_csr_x(h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dA.ptr, dH.ptr, R, flags, nnz, maxrow, cols, C.byref(bad), 0) != 0
    for d in (d_rp, d_c, dv, dX, dA, dH, dwide):
        d.free()
