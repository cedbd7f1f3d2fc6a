"""GPU parity of the hand-written HIP kernels (through the C ABI of include/hnh_kernels.h) against the
CPU oracle (oracle/oracle.py: sddmm_local = sparse_kernels.cpp:44-55, spmm_local = :95-107).

Tolerance (BASELINE.md / SURVEY §8d): fp64, only the summation order differs ->
    max|x - x_ref| <= 1e-11 * max|x_ref|.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-11


@pytest.fixture(scope="module")
def ctx():
    from distributed_sddmm_amd import _kernels as K
    c = K.Ctx(0)
    assert K.load().hnh_backend_name() == b"hip-gfx950"
    yield c
    c.close()


def rel(x, y):
    return float(np.max(np.abs(x - y)) / max(float(np.max(np.abs(y))), 1e-300)) if x.size else 0.0


def random_block(rows, cols, nnz_target, seed, empty_rows=True):
    rng = np.random.default_rng(seed)
    r = rng.integers(0, rows, nnz_target)
    if empty_rows and rows > 4:
        r = r[(r % 5) != 3]  # leave some rows empty
    c = rng.integers(0, cols, len(r))
    keys = np.unique(r.astype(np.int64) * cols + c)
    r, c = (keys // cols).astype(np.int32), (keys % cols).astype(np.int32)
    rowptr = np.zeros(rows + 1, dtype=np.int32)
    np.add.at(rowptr, r + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    return rowptr, r, c


RS = [1, 2, 3, 4, 8, 16, 17, 32, 64, 100, 128, 192, 256, 320, 384, 448, 512, 130, 200, 257, 301, 600]


@pytest.mark.parametrize("R", RS)
def test_sddmm_spmm_fused_vs_oracle(ctx, R):
    from distributed_sddmm_amd import _kernels as K
    lib = ctx.lib
    rows, cols = 301, 257
    rowptr, ridx, cidx = random_block(rows, cols, 4000, seed=R)
    nnz = len(cidx)
    rng = np.random.default_rng(R + 1000)
    X = rng.uniform(-1, 1, (rows, R)); Y = rng.uniform(-1, 1, (cols, R))
    v0 = rng.uniform(-1, 1, nnz); out0 = rng.uniform(-1, 1, (rows, R)); sv = rng.uniform(-1, 1, nnz)
    d_rowptr, d_ridx, d_cidx = ctx.upload(rowptr), ctx.upload(ridx), ctx.upload(cidx)
    dX, dY = ctx.upload(X), ctx.upload(Y)

    # --- SDDMM (CSR and COO views): values += <X[r], Y[c]>
    want = O.sddmm_local(ridx, cidx, v0, X, Y)
    dv = ctx.upload(v0)
    ctx.check(lib.hnh_sddmm_csr(ctx.h, rows, d_rowptr.ptr, d_cidx.ptr, dv.ptr, dX.ptr, dY.ptr, R, 0), "sddmm_csr")
    assert rel(dv.get(), want) <= TOL
    dv.set(v0)
    ctx.check(lib.hnh_sddmm_coo(ctx.h, nnz, d_ridx.ptr, d_cidx.ptr, dv.ptr, dX.ptr, dY.ptr, R, 0), "sddmm_coo")
    assert rel(dv.get(), want) <= TOL

    # --- SpMM: Out += S * Y   (alpha = beta = 1)
    want_out = O.spmm_local(rowptr, cidx, v0, Y, out0)
    dv.set(v0)
    dOut = ctx.upload(out0)
    ctx.check(lib.hnh_spmm_csr(ctx.h, rows, d_rowptr.ptr, d_cidx.ptr, dv.ptr, dY.ptr, dOut.ptr, R, 0), "spmm_csr")
    assert rel(dOut.get(), want_out) <= TOL

    # --- fused, accumulate semantics (reference call pair, 15D_dense_shift.hpp:203-217)
    vals_after = O.sddmm_local(ridx, cidx, v0, X, Y)
    want_out = O.spmm_local(rowptr, cidx, vals_after, Y, out0)
    dv.set(v0); dOut.set(out0)
    ctx.check(lib.hnh_fused_sddmm_spmm_csr(ctx.h, rows, d_rowptr.ptr, d_cidx.ptr, dv.ptr, None, dX.ptr, dY.ptr,
                                           dOut.ptr, R, 0, 0), "fused")
    assert rel(dv.get(), vals_after) <= TOL
    assert rel(dOut.get(), want_out) <= TOL

    # --- fused, overwrite flags (values and Out known to be zero) + svalues extension
    dots = O.sddmm_local(ridx, cidx, np.zeros(nnz), X, Y)
    want_out = O.spmm_local(rowptr, cidx, dots * sv, Y, np.zeros((rows, R)))
    dv.set(v0); dOut.set(out0)  # garbage on purpose: overwrite must ignore it
    dsv = ctx.upload(sv)
    ctx.check(lib.hnh_fused_sddmm_spmm_csr(ctx.h, rows, d_rowptr.ptr, d_cidx.ptr, dv.ptr, dsv.ptr, dX.ptr, dY.ptr,
                                           dOut.ptr, R, K.FUSED_VALUES_OVERWRITE | K.FUSED_OUT_OVERWRITE, 0), "fused ow")
    assert rel(dv.get(), dots) <= TOL
    assert rel(dOut.get(), want_out) <= TOL
    for d in (d_rowptr, d_ridx, d_cidx, dX, dY, dv, dOut, dsv):
        d.free()


def test_empty_and_degenerate(ctx):
    lib = ctx.lib
    R = 16
    # rows = 0 and nnz = 0 are no-ops (sparse_kernels.cpp:25-27,85-87)
    assert lib.hnh_sddmm_csr(ctx.h, 0, None, None, None, None, None, R, 0) == 0
    assert lib.hnh_sddmm_coo(ctx.h, 0, None, None, None, None, None, R, 0) == 0
    assert lib.hnh_spmm_csr(ctx.h, 0, None, None, None, None, None, R, 0) == 0
    # a block whose rows are all empty: SpMM leaves Out unchanged, overwrite-fused zeroes it
    rows = 70
    rowptr = np.zeros(rows + 1, dtype=np.int32)
    out0 = np.arange(rows * R, dtype=np.float64).reshape(rows, R)
    d_rowptr, d_c, d_v = ctx.upload(rowptr), ctx.upload(np.zeros(1, np.int32)), ctx.upload(np.zeros(1))
    dX, dOut = ctx.upload(np.ones((rows, R))), ctx.upload(out0)
    ctx.check(lib.hnh_spmm_csr(ctx.h, rows, d_rowptr.ptr, d_c.ptr, d_v.ptr, dX.ptr, dOut.ptr, R, 0), "spmm empty")
    assert np.array_equal(dOut.get(), out0)
    dY = ctx.upload(np.ones((rows, R)))
    ctx.check(lib.hnh_fused_sddmm_spmm_csr(ctx.h, rows, d_rowptr.ptr, d_c.ptr, d_v.ptr, None, dX.ptr, dY.ptr, dOut.ptr, R, 3, 0), "fused empty")
    assert np.all(dOut.get() == 0.0)
    # argument errors are reported, not crashed on
    assert lib.hnh_sddmm_csr(ctx.h, 4, d_rowptr.ptr, d_c.ptr, d_v.ptr, dX.ptr, dY.ptr, 0, 0) != 0
    assert lib.hnh_sddmm_csr(ctx.h, -1, d_rowptr.ptr, d_c.ptr, d_v.ptr, dX.ptr, dY.ptr, R, 0) != 0
    assert lib.hnh_spmm_csr(ctx.h, 4, d_rowptr.ptr, d_c.ptr, d_v.ptr, dX.ptr, dX.ptr, R, 0) != 0  # aliasing


def test_long_row_and_unaligned(ctx):
    """One hub row with many nonzeros (skewed graphs) and an operand pointer that is only 8-byte aligned."""
    lib = ctx.lib
    rows, cols, R = 9, 5000, 128
    rng = np.random.default_rng(5)
    c_hub = np.sort(rng.choice(cols, 3001, replace=False)).astype(np.int32)
    rowptr = np.array([0, 0, 3001, 3001, 3003, 3003, 3003, 3003, 3003, 3004], dtype=np.int32)
    cidx = np.concatenate([c_hub, np.array([7, 4999, 0], dtype=np.int32)])
    ridx = np.repeat(np.arange(rows, dtype=np.int32), np.diff(rowptr))
    X = rng.uniform(-1, 1, (rows, R)); Y = rng.uniform(-1, 1, (cols, R)); v0 = rng.uniform(-1, 1, len(cidx))
    d_rowptr, d_c, dv = ctx.upload(rowptr), ctx.upload(cidx), ctx.upload(v0)
    dX, dY = ctx.upload(X), ctx.upload(Y)
    ctx.check(lib.hnh_sddmm_csr(ctx.h, rows, d_rowptr.ptr, d_c.ptr, dv.ptr, dX.ptr, dY.ptr, R, 0), "sddmm hub")
    assert rel(dv.get(), O.sddmm_local(ridx, cidx, v0, X, Y)) <= TOL
    # R = 3 rows starting 8 bytes into an allocation -> scalar (W = 1) path
    R3 = 3
    Xp = np.zeros(rows * R3 + 1); Xp[1:] = rng.uniform(-1, 1, rows * R3)
    Yp = np.zeros(cols * R3 + 1); Yp[1:] = rng.uniform(-1, 1, cols * R3)
    dXp, dYp = ctx.upload(Xp), ctx.upload(Yp)
    dv.set(v0)
    ctx.check(lib.hnh_sddmm_csr(ctx.h, rows, d_rowptr.ptr, d_c.ptr, dv.ptr, dXp.ptr + 8, dYp.ptr + 8, R3, 0), "sddmm unaligned")
    assert rel(dv.get(), O.sddmm_local(ridx, cidx, v0, Xp[1:].reshape(rows, R3), Yp[1:].reshape(cols, R3))) <= TOL


def test_elementwise(ctx):
    lib = ctx.lib
    n = 100003
    rng = np.random.default_rng(3)
    a, b = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    da, db, do = ctx.upload(a), ctx.upload(b), ctx.upload(np.zeros(n))
    ctx.check(lib.hnh_hadamard_f64(ctx.h, do.ptr, da.ptr, db.ptr, n, 0), "hadamard")
    assert np.array_equal(do.get(), a * b)
    ctx.check(lib.hnh_fill_f64(ctx.h, do.ptr, n, 2.5, 0), "fill")
    assert np.all(do.get() == 2.5)
    ctx.check(lib.hnh_axpy_f64(ctx.h, do.ptr, da.ptr, -3.0, n, 0), "axpy")
    assert rel(do.get(), 2.5 - 3.0 * a) <= 1e-15
    rows = 50
    rowptr, ridx, _ = random_block(rows, 40, 300, seed=9)
    dr, di = ctx.upload(rowptr), ctx.upload(np.full(len(ridx), -1, np.int32))
    ctx.check(lib.hnh_expand_rowptr(ctx.h, rows, dr.ptr, di.ptr, 0), "expand")
    assert np.array_equal(di.get(), ridx)


@pytest.mark.parametrize("R", [1, 3, 8, 16, 64, 128, 130, 256])
def test_rowwise_als_helpers(ctx, R):
    """hnh_rowdot_f64 / hnh_row_scale_add_f64 / hnh_vec_* against numpy (als_conjugate_gradients.cpp:9-29,99-137)."""
    lib = ctx.lib
    rows = 1237
    rng = np.random.default_rng(R)
    A, B = rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, (rows, R))
    yv, xv = rng.uniform(-1, 1, rows), rng.uniform(0.5, 1.5, rows)
    dA, dB, dyv, dxv, dout = ctx.upload(A), ctx.upload(B), ctx.upload(yv), ctx.upload(xv), ctx.upload(np.zeros(rows))
    ctx.check(lib.hnh_rowdot_f64(ctx.h, dA.ptr, dB.ptr, dout.ptr, rows, R, 0), "rowdot")
    assert rel(dout.get(), np.einsum("ij,ij->i", A, B)) <= 1e-14
    ctx.check(lib.hnh_row_scale_add_f64(ctx.h, dA.ptr, dyv.ptr, 0.5, dB.ptr, dxv.ptr, -2.0, rows, R, 0), "row_scale_add")
    assert rel(dA.get(), 0.5 * yv[:, None] * A - 2.0 * xv[:, None] * B) <= 1e-15
    dA.set(A)
    ctx.check(lib.hnh_row_scale_add_f64(ctx.h, dA.ptr, None, 1.0, dB.ptr, None, 1e-13, rows, R, 0), "row_scale_add lambda")
    assert rel(dA.get(), A + 1e-13 * B) <= 1e-15
    ctx.check(lib.hnh_vec_add_scalar_f64(ctx.h, dyv.ptr, 1e-8, rows, 0), "vec_add")
    assert np.array_equal(dyv.get(), yv + 1e-8)
    ctx.check(lib.hnh_vec_div_f64(ctx.h, dout.ptr, dyv.ptr, dxv.ptr, rows, 0), "vec_div")
    assert np.array_equal(dout.get(), (yv + 1e-8) / xv)


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (100, 37, 29), (513, 130, 70), (1000, 256, 256), (16, 8, 1024), (5, 3, 1), (2100, 300, 72), (4096, 256, 1024), (129, 129, 17)])
def test_gemm_f64_mfma(ctx, M, N, K):
    """hnh_gemm_f64 (v_mfma_f64_16x16x4_f64, gat.hpp:88) vs numpy; asymmetric operands catch row/col swaps."""
    lib = ctx.lib
    rng = np.random.default_rng(M * 7 + N)
    A, B = rng.uniform(-1, 1, (M, K)), rng.uniform(-1, 1, (K, N))
    B[0, :] += np.arange(N)  # asymmetric on purpose
    dA, dB, dC = ctx.upload(A), ctx.upload(B), ctx.upload(np.full((M, N), 7.0))
    ctx.check(lib.hnh_gemm_f64(ctx.h, M, N, K, dA.ptr, dB.ptr, dC.ptr, 0), "gemm")
    assert rel(dC.get(), A @ B) <= 1e-13


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (100, 37, 29), (513, 130, 70), (1000, 256, 256), (5, 3, 1), (2100, 300, 72), (4096, 256, 1024), (257, 129, 17), (256, 128, 64)])
def test_gemm_f64_mfma_tall_workgroups(M, N, K, monkeypatch):
    """HNH_GEMM_WAVES=8: the same contraction through 256 x 128 tiles of 8 waves (two waves per SIMD from one workgroup)."""
    from distributed_sddmm_amd import _kernels as K_
    monkeypatch.setenv("HNH_GEMM_WAVES", "8")
    c = K_.Ctx(0)
    try:
        rng = np.random.default_rng(M * 11 + N)
        A, B = rng.uniform(-1, 1, (M, K)), rng.uniform(-1, 1, (K, N))
        B[0, :] += np.arange(N)
        dA, dB, dC = c.upload(A), c.upload(B), c.upload(np.full((M, N), 7.0))
        c.check(c.lib.hnh_gemm_f64(c.h, M, N, K, dA.ptr, dB.ptr, dC.ptr, 0), "gemm")
        assert rel(dC.get(), A @ B) <= 1e-13
    finally:
        c.close()


def test_gat_elementwise(ctx):
    lib = ctx.lib
    rng = np.random.default_rng(1)
    v = rng.uniform(-1, 1, 10007)
    dv = ctx.upload(v)
    ctx.check(lib.hnh_leaky_relu_f64(ctx.h, dv.ptr, 0.2, len(v), 0), "leaky")
    assert np.array_equal(dv.get(), np.maximum(v, 0) + np.minimum(v, 0) * 0.2)
    src = rng.uniform(-1, 1, (33, 5))
    dst0 = rng.uniform(-1, 1, (33, 12))
    dsrc, ddst = ctx.upload(src), ctx.upload(dst0)
    ctx.check(lib.hnh_relu_store_cols_f64(ctx.h, ddst.ptr, 12, 4, dsrc.ptr, 33, 5, 0), "relu cols")
    want = dst0.copy(); want[:, 4:9] = np.maximum(src, 0)
    assert np.array_equal(ddst.get(), want)


@pytest.mark.parametrize("R", [16, 128, 100])
@pytest.mark.parametrize("hinted", [False, True])
def test_hub_rows_are_split_and_still_exact(ctx, R, hinted):
    """Hub rows (longer than 3 x the block's mean row length, threshold within 256..1024) take the segmented long-row pass
    (SpMM / fused combine with fp64 atomics)."""
    import ctypes as C
    from distributed_sddmm_amd import _kernels as K
    lib = ctx.lib
    rows, cols = 60, 9000
    rng = np.random.default_rng(R)
    lens = rng.integers(0, 40, rows)
    lens[7], lens[31], lens[59] = 5000, 1025, 2600      # three hubs, one just over the threshold
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cidx = np.concatenate([np.sort(rng.choice(cols, n, replace=False)) for n in lens]).astype(np.int32)
    ridx = np.repeat(np.arange(rows, dtype=np.int32), lens)
    nnz = len(cidx)
    X, Y = rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, (cols, R))
    v0, out0 = rng.uniform(-1, 1, nnz), rng.uniform(-1, 1, (rows, R))
    d_rp, d_c, dv, dX, dY, dOut = ctx.upload(rowptr), ctx.upload(cidx), ctx.upload(v0), ctx.upload(X), ctx.upload(Y), ctx.upload(out0)
    mx = C.c_int()
    ctx.check(lib.hnh_csr_max_row_nnz(ctx.h, rows, d_rp.ptr, C.byref(mx), 0), "max_row")
    assert mx.value == 5000
    h_nnz, h_max = (nnz, mx.value) if hinted else (-1, -1)
    ctx.check(lib.hnh_sddmm_csr_ex(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, dX.ptr, dY.ptr, R, h_nnz, h_max, cols, 0), "sddmm")
    assert rel(dv.get(), O.sddmm_local(ridx, cidx, v0, X, Y)) <= TOL
    dv.set(v0)
    ctx.check(lib.hnh_spmm_csr_ex(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, dY.ptr, dOut.ptr, R, h_nnz, h_max, cols, 0), "spmm")
    assert rel(dOut.get(), O.spmm_local(rowptr, cidx, v0, Y, out0)) <= TOL
    for flags in (0, K.FUSED_VALUES_OVERWRITE | K.FUSED_OUT_OVERWRITE):
        dv.set(v0); dOut.set(out0)
        vbase = v0 if flags == 0 else np.zeros(nnz)
        obase = out0 if flags == 0 else np.zeros((rows, R))
        ctx.check(lib.hnh_fused_sddmm_spmm_csr_ex(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dY.ptr, dOut.ptr, R, flags,
                                                  h_nnz, h_max, cols, 0), "fused")
        vals = O.sddmm_local(ridx, cidx, vbase, X, Y)
        assert rel(dv.get(), vals) <= TOL
        assert rel(dOut.get(), O.spmm_local(rowptr, cidx, vals, Y, obase)) <= TOL
    # a wrong "no long rows" hint is the caller's bug, a correct one for a short-row block skips the machinery
    short_rp = np.arange(0, 11, dtype=np.int32) * 3
    d_srp = ctx.upload(short_rp)
    ctx.check(lib.hnh_csr_max_row_nnz(ctx.h, 10, d_srp.ptr, C.byref(mx), 0), "max_row")
    assert mx.value == 3


@pytest.mark.parametrize("atomics", [False, True])
def test_repeat_runs_are_bitwise_reproducible(monkeypatch, atomics):
    """Run-to-run reproducibility.  Rows below the hub-row threshold are finished by ONE group in a fixed order.  Hub rows are cut
    into 256-nonzero segments; each segment writes its partial output row and a second kernel adds a row's segments up in
    segment order — so repeated calls give bit-identical results for every row (like the reference for a fixed thread count).
    HNH_HUB_ATOMICS=1 selects round 1's way (segments combine with hardware fp64 atomics in arrival order): results then agree
    within the parity tolerance, not necessarily bit for bit ("Long rows")."""
    if atomics:
        monkeypatch.setenv("HNH_HUB_ATOMICS", "1")
    from distributed_sddmm_amd import _kernels as K
    ctx = K.Ctx(0)
    lib, R = ctx.lib, 64
    rng = np.random.default_rng(11)
    rows, cols = 300, 9000
    for hubs in (False, True):
        lens = rng.integers(0, 40, rows)
        if hubs:
            lens[5], lens[77], lens[200] = 6000, 1500, 3100
        rowptr = np.zeros(rows + 1, np.int32); rowptr[1:] = np.cumsum(lens)
        cidx = np.concatenate([np.sort(rng.choice(cols, int(n), replace=False)) for n in lens]).astype(np.int32)
        ridx = np.repeat(np.arange(rows, dtype=np.int32), lens)
        X, Y = rng.standard_normal((rows, R)), rng.standard_normal((cols, R))
        d_rp, d_c, dX, dY = ctx.upload(rowptr), ctx.upload(cidx), ctx.upload(X), ctx.upload(Y)
        mx = C.c_int()  # (first use of the fresh context's counters: the hint query and the hub-row pass share them)
        ctx.check(lib.hnh_csr_max_row_nnz(ctx.h, rows, d_rp.ptr, C.byref(mx), 0), "max_row")
        assert mx.value == int(lens.max())
        outs, vals, spmms = [], [], []
        for _ in range(6):
            dv, dOut, dOut2 = ctx.upload(np.zeros(len(cidx))), ctx.upload(np.zeros((rows, R))), ctx.upload(np.zeros((rows, R)))
            ctx.check(lib.hnh_fused_sddmm_spmm_csr_ex(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dY.ptr, dOut.ptr, R, 3,
                                                      len(cidx), int(lens.max()), cols, 0), "fused repeat")
            ctx.check(lib.hnh_spmm_csr_ex(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, dY.ptr, dOut2.ptr, R, len(cidx), int(lens.max()), cols, 0), "spmm repeat")
            ctx.sync()
            outs.append(dOut.get()); vals.append(dv.get()); spmms.append(dOut2.get())
            dv.free(); dOut.free(); dOut2.free()
        mid = O.sddmm_local(ridx, cidx, np.zeros(len(cidx)), X, Y)
        want = O.spmm_local(rowptr, cidx, mid, Y, np.zeros((rows, R)))
        for k in range(6):
            assert np.array_equal(vals[k], vals[0])  # SDDMM values: one group per nonzero batch, always bitwise
            assert rel(outs[k], want) <= TOL and rel(spmms[k], want) <= TOL
            if not (hubs and atomics):
                assert np.array_equal(outs[k], outs[0]) and np.array_equal(spmms[k], spmms[0])
        for d in (d_rp, d_c, dX, dY):
            d.free()
    ctx.close()


@pytest.mark.parametrize("budget_mb", ["0.01", "0"])
def test_hub_scratch_budget_mixes_ordered_and_atomic_segments(monkeypatch, budget_mb):
    """HNH_HUB_SCRATCH_MB bounds the partial-row scratch of the hub-row segments: with room for only SOME of a block's segments
    (0.01 MB = 20 of the 43 here) the first ones are reduced in order and the rest combine with atomics in the same launch; with
    no room at all every segment uses atomics.  Either way the results agree with the oracle, through the per-call entry points
    and through a block descriptor with a structure plan (whose exact segment count sizes the scratch), and a row epilogue that
    cannot ride in the launch runs as its own kernel."""
    monkeypatch.setenv("HNH_HUB_SCRATCH_MB", budget_mb)
    from distributed_sddmm_amd import _kernels as K
    ctx = K.Ctx(0)
    lib, R = ctx.lib, 64
    rng = np.random.default_rng(21)
    rows, cols = 300, 9000
    lens = rng.integers(0, 40, rows)
    lens[5], lens[77], lens[200] = 6000, 1500, 3100
    rowptr = np.zeros(rows + 1, np.int32); rowptr[1:] = np.cumsum(lens)
    cidx = np.concatenate([np.sort(rng.choice(cols, int(n), replace=False)) for n in lens]).astype(np.int32)
    ridx = np.repeat(np.arange(rows, dtype=np.int32), lens)
    X, Y = rng.standard_normal((rows, R)), rng.standard_normal((cols, R))
    d_rp, d_c, dX, dY = ctx.upload(rowptr), ctx.upload(cidx), ctx.upload(X), ctx.upload(Y)
    mid = O.sddmm_local(ridx, cidx, np.zeros(len(cidx)), X, Y)
    want = O.spmm_local(rowptr, cidx, mid, Y, np.zeros((rows, R)))
    plan = C.c_void_p()
    ctx.check(lib.hnh_csr_plan_create(ctx.h, C.byref(plan)), "plan")
    blk = K.CsrBlock(rows, len(cidx), cols, int(lens.max()), 0, d_rp.ptr, d_c.ptr, plan)
    for use_plan in (False, True, True):  # (the second planned call reuses the cached hub list)
        dv, dOut, dDot = ctx.upload(np.zeros(len(cidx))), ctx.upload(np.full((rows, R), 7.0)), ctx.upload(np.zeros(rows))
        ex = K.FusedExtras(0.0, 0.5, dDot.ptr, None, None, 0)
        if use_plan:
            ctx.check(lib.hnh_fused_sddmm_spmm_csr_p(ctx.h, C.byref(blk), dv.ptr, None, dX.ptr, dY.ptr, dOut.ptr, R, 3, C.byref(ex), None, 0), "fused_p")
        else:
            ctx.check(lib.hnh_fused_sddmm_spmm_csr_x(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dY.ptr, dOut.ptr, R, 3, len(cidx),
                                                     int(lens.max()), cols, C.byref(ex), 0), "fused_x")
        ctx.sync()
        out = want + 0.5 * X  # Out += x_scale * X, rowdot = <X, Out>
        assert rel(dv.get(), mid) <= TOL and rel(dOut.get(), out) <= TOL
        assert rel(dDot.get(), np.sum(X * out, axis=1)) <= TOL
        dOut2 = ctx.upload(np.zeros((rows, R)))
        if use_plan:
            ctx.check(lib.hnh_spmm_csr_p(ctx.h, C.byref(blk), dv.ptr, dY.ptr, dOut2.ptr, R, None, 0), "spmm_p")
        else:
            ctx.check(lib.hnh_spmm_csr_ex(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, dY.ptr, dOut2.ptr, R, len(cidx), int(lens.max()), cols, 0), "spmm")
        ctx.sync()
        assert rel(dOut2.get(), want) <= TOL
        for d in (dv, dOut, dDot, dOut2):
            d.free()
    ctx.check(lib.hnh_csr_plan_destroy(ctx.h, plan), "plan destroy")
    for d in (d_rp, d_c, dX, dY):
        d.free()


def test_a_plan_refuses_another_blocks_arrays(ctx):
    """A structure plan belongs to the block it was first used with: handing it to other index arrays is an error, not a silently
    wrong answer."""
    from distributed_sddmm_amd import _kernels as K
    lib, R = ctx.lib, 16
    rowptr = np.arange(0, 41, 4, dtype=np.int32)
    cidx = np.tile(np.arange(4, dtype=np.int32), 10)
    d_rp, d_c, d_c2 = ctx.upload(rowptr), ctx.upload(cidx), ctx.upload(cidx)
    dv, dX, dY = ctx.upload(np.zeros(40)), ctx.upload(np.ones((10, R))), ctx.upload(np.ones((4, R)))
    plan = C.c_void_p()
    ctx.check(lib.hnh_csr_plan_create(ctx.h, C.byref(plan)), "plan")
    blk = K.CsrBlock(10, 40, 4, 4, 0, d_rp.ptr, d_c.ptr, plan)
    ctx.check(lib.hnh_sddmm_csr_p(ctx.h, C.byref(blk), dv.ptr, dX.ptr, dY.ptr, R, 0, None, 0), "sddmm_p")
    other = K.CsrBlock(10, 40, 4, 4, 0, d_rp.ptr, d_c2.ptr, plan)
    assert lib.hnh_sddmm_csr_p(ctx.h, C.byref(other), dv.ptr, dX.ptr, dY.ptr, R, 0, None, 0) != 0
    assert b"another block" in lib.hnh_last_error(ctx.h)
    ctx.sync()
    assert np.allclose(dv.get(), R)
    ctx.check(lib.hnh_csr_plan_destroy(ctx.h, plan), "plan destroy")
    for d in (d_rp, d_c, d_c2, dv, dX, dY):
        d.free()


@pytest.mark.parametrize("R", [16, 128, 100, 257])
@pytest.mark.parametrize("hubs", [False, True])
def test_windowed_passes_equal_the_whole_block(ctx, R, hubs):
    """hnh_*_csr_w: the block processed window by window (column ranges, as the 1.5D dense-shift schedule does while the
    chunks of the fetched blocks land) gives what one pass over the whole block gives — SDDMM, SpMM and the fused pass with
    activation and row epilogue on the last window; with hub rows (left whole, handled with the last window) and for widths
    that take the column-tiled fallback."""
    from distributed_sddmm_amd import _kernels as K
    lib = ctx.lib
    rng = np.random.default_rng(R + hubs)
    rows, cols = 200, 6000
    lens = rng.integers(0, 60, rows)
    if hubs:
        lens[3], lens[150] = 4000, 1300
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cidx = np.concatenate([np.sort(rng.choice(cols, int(n), replace=False)) for n in lens]).astype(np.int32)
    ridx = np.repeat(np.arange(rows, dtype=np.int32), lens)
    nnz = len(cidx)
    X, Y = rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, (cols, R))
    v0, out0 = rng.uniform(-1, 1, nnz), rng.uniform(-1, 1, (rows, R))
    bounds = np.array([1500, 1500, 4100], dtype=np.int32)  # four windows, one of them empty
    nw = len(bounds) + 1
    d_rp, d_c, dX, dY = ctx.upload(rowptr), ctx.upload(cidx), ctx.upload(X), ctx.upload(Y)
    d_split = ctx.upload(np.zeros((len(bounds), rows), np.int32))
    ctx.check(lib.hnh_csr_window_bounds(ctx.h, rows, d_rp.ptr, d_c.ptr, len(bounds), bounds.ctypes.data_as(C.c_void_p), d_split.ptr, 0), "bounds")

    def window(q):
        beg = None if q == 0 else d_split.ptr + (q - 1) * rows * 4
        end = None if q == nw - 1 else d_split.ptr + q * rows * 4
        return K.CsrWindow(beg, end, 1 if q == nw - 1 else 0)

    mx = int(lens.max())
    # SDDMM and SpMM
    dv = ctx.upload(v0)
    for q in range(nw):
        ctx.check(lib.hnh_sddmm_csr_w(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, dX.ptr, dY.ptr, R, nnz, mx, C.byref(window(q)), 0), "sddmm_w")
    assert rel(dv.get(), O.sddmm_local(ridx, cidx, v0, X, Y)) <= TOL
    dv.set(v0)
    dOut = ctx.upload(out0)
    for q in range(nw):
        ctx.check(lib.hnh_spmm_csr_w(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, dY.ptr, dOut.ptr, R, nnz, mx, C.byref(window(q)), 0), "spmm_w")
    assert rel(dOut.get(), O.spmm_local(rowptr, cidx, v0, Y, out0)) <= TOL
    # fused with LeakyReLU, overwrite flags on the first window, the row epilogue on the last
    d_dot = ctx.upload(np.zeros(rows))
    alpha, xs = 0.2, -0.5
    for q in range(nw):
        ex = K.FusedExtras(alpha, xs if q == nw - 1 else 0.0, d_dot.ptr if q == nw - 1 else None)
        flags = K.FUSED_VALUES_OVERWRITE | K.FUSED_LEAKY_RELU | (K.FUSED_OUT_OVERWRITE if q == 0 else 0)
        ctx.check(lib.hnh_fused_sddmm_spmm_csr_w(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dY.ptr, dOut.ptr, R, flags, nnz, mx,
                                                 C.byref(ex), C.byref(window(q)), 0), "fused_w")
    dots = O.sddmm_local(ridx, cidx, np.zeros(nnz), X, Y)
    act = np.where(dots > 0, dots, alpha * dots)
    want = O.spmm_local(rowptr, cidx, act, Y, np.zeros((rows, R))) + xs * X
    assert rel(dv.get(), act) <= TOL
    assert rel(dOut.get(), want) <= TOL
    assert rel(d_dot.get(), np.sum(X * want, axis=1)) <= TOL
    # an epilogue on a window that is not the last one is refused
    ex = K.FusedExtras(alpha, 1.0, None)
    assert lib.hnh_fused_sddmm_spmm_csr_w(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dY.ptr, dOut.ptr, R, 0, nnz, mx, C.byref(ex),
                                          C.byref(window(0)), 0) != 0
    for d in (d_rp, d_c, dX, dY, d_split, dv, dOut, d_dot):
        d.free()


def test_fill_hashed_matches_the_oracle_hash(ctx):
    """hnh_fill_hashed_f64 = oracle.hashed_uniform keyed by the global (row, col) of a sub-block."""
    lib = ctx.lib
    rows, cols, top, left, rg, seed = 37, 5, 1000, 3, 16, 2025
    d = ctx.upload(np.zeros((rows, cols)))
    ctx.check(lib.hnh_fill_hashed_f64(ctx.h, d.ptr, rows, cols, top, left, rg, seed, 0.25, 0), "fill_hashed")
    ii, jj = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    keys = ((top + ii) * rg + left + jj).astype(np.uint64).reshape(-1)
    assert np.array_equal(d.get().reshape(-1), O.hashed_uniform(keys, seed) * 0.25)


def _extras_expected(rowptr, ridx, cidx, v0, sv, X, Y, out0, flags, alpha, x_scale):
    """numpy statement of hnh_fused_sddmm_spmm_csr_x (include/hnh_kernels.h)."""
    from distributed_sddmm_amd import _kernels as K
    nnz, (rows, R) = len(cidx), X.shape
    vbase = np.zeros(nnz) if flags & K.FUSED_VALUES_OVERWRITE else v0
    obase = np.zeros((rows, R)) if flags & K.FUSED_OUT_OVERWRITE else out0
    vals = O.sddmm_local(ridx, cidx, vbase, X, Y)
    if flags & K.FUSED_LEAKY_RELU:
        vals = vals * (sv if sv is not None else 1.0)
        vals = np.where(vals > 0, vals, alpha * vals)
        w = vals
    else:
        w = vals * (sv if sv is not None else 1.0)
    out = O.spmm_local(rowptr, cidx, w, Y, obase) + x_scale * X
    return vals, out, np.einsum("ij,ij->i", X, out)


@pytest.mark.parametrize("R", [1, 2, 7, 16, 64, 100, 128, 192, 256, 301, 512, 600])
def test_fused_extras_activation_and_row_epilogue(ctx, R):
    """LeakyReLU between the halves (gat.hpp:96-99) and the `+ x_scale X`, <X, Out> row epilogue
    (als_conjugate_gradients.cpp:93,282) inside the fused launch; every kernel shape incl. the tiled fallback."""
    from distributed_sddmm_amd import _kernels as K
    lib = ctx.lib
    rows, cols = 211, 190
    rowptr, ridx, cidx = random_block(rows, cols, 3000, seed=R + 77)
    nnz = len(cidx)
    rng = np.random.default_rng(R + 5)
    X, Y = rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, (cols, R))
    v0, out0, sv = rng.uniform(-1, 1, nnz), rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, nnz)
    d_rp, d_c, dv, dX, dY, dOut, dsv = (ctx.upload(a) for a in (rowptr, cidx, v0, X, Y, out0, sv))
    ddot = ctx.upload(np.full(rows, 9.0))
    OW = K.FUSED_VALUES_OVERWRITE | K.FUSED_OUT_OVERWRITE
    for flags, use_sv, alpha, xs, want_dot in [(OW | K.FUSED_LEAKY_RELU, False, 0.2, 0.0, False), (K.FUSED_LEAKY_RELU, True, 0.01, 0.5, True),
                                               (OW, False, 0.0, 1e-3, True), (0, True, 0.0, -2.0, True), (OW, False, 0.0, 0.0, True)]:
        dv.set(v0); dOut.set(out0); ddot.set(np.full(rows, 9.0))
        ex = K.FusedExtras(alpha, xs, ddot.ptr if want_dot else None)
        ctx.check(lib.hnh_fused_sddmm_spmm_csr_x(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, dsv.ptr if use_sv else None, dX.ptr, dY.ptr,
                                                 dOut.ptr, R, flags, nnz, int(np.diff(rowptr).max()), cols, C.byref(ex), 0), "fused_x")
        vals, out, dot = _extras_expected(rowptr, ridx, cidx, v0, sv if use_sv else None, X, Y, out0, flags, alpha, xs)
        assert rel(dv.get(), vals) <= TOL
        assert rel(dOut.get(), out) <= TOL
        if want_dot:
            assert rel(ddot.get(), dot) <= TOL
    # the flag without extras is a caller error
    assert lib.hnh_fused_sddmm_spmm_csr_x(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dY.ptr, dOut.ptr, R, K.FUSED_LEAKY_RELU, -1, -1,
                                          -1, None, 0) != 0
    # standalone epilogue and the CG update
    dOut.set(out0)
    ctx.check(lib.hnh_row_epilogue_f64(ctx.h, dOut.ptr, dX.ptr, 0.25, ddot.ptr, rows, R, 0), "row_epilogue")
    assert rel(dOut.get(), out0 + 0.25 * X) <= TOL and rel(ddot.get(), np.einsum("ij,ij->i", X, out0 + 0.25 * X)) <= TOL
    Xm, Rm, P, MP = (rng.uniform(-1, 1, (rows, R)) for _ in range(4))
    al = rng.uniform(-1, 1, rows)
    dXm, dRm, dP, dMP, dal = (ctx.upload(a) for a in (Xm, Rm, P, MP, al))
    ctx.check(lib.hnh_cg_step_f64(ctx.h, dXm.ptr, dRm.ptr, dP.ptr, dMP.ptr, dal.ptr, ddot.ptr, rows, R, 0), "cg_step")
    r_new = Rm - al[:, None] * MP
    assert rel(dXm.get(), Xm + al[:, None] * P) <= TOL and rel(dRm.get(), r_new) <= TOL
    assert rel(ddot.get(), np.einsum("ij,ij->i", r_new, r_new)) <= TOL
    for d in (d_rp, d_c, dv, dX, dY, dOut, dsv, ddot, dXm, dRm, dP, dMP, dal):
        d.free()


@pytest.mark.parametrize("R", [1, 2, 7, 16, 32, 64, 100, 128, 192, 256, 301, 512, 600])
@pytest.mark.parametrize("variant", ["in_launch", "hub_rows", "windows", "standalone"])
def test_folded_cg_iteration(ctx, R, variant):
    """hnh_cg_update: the rest of a batched-CG iteration (als_conjugate_gradients.cpp:91-139) in the fused call's row epilogue,
    for every kernel shape; with hub rows / column tiles the epilogue is its own launch, with windows it rides on the last one."""
    import cg_common
    cg_common.run(ctx, R, hubs=variant == "hub_rows", windows=3 if variant == "windows" else 0, standalone=variant == "standalone")


@pytest.mark.parametrize("R", [1, 2, 7, 16, 64, 100, 128, 256, 301, 512, 600])
@pytest.mark.parametrize("variant", ["in_launch", "hub_rows", "windows"])
def test_relu_delivery_into_a_column_block(ctx, R, variant):
    """relu_dst: a GAT head's output (gat.hpp:96-101) leaves the fused launch through the ReLU into its column block of the
    layer output; neighbouring blocks stay untouched."""
    import cg_common
    cg_common.run_relu(ctx, R, hubs=variant == "hub_rows", windows=3 if variant == "windows" else 0)


@pytest.mark.parametrize("R", [16, 128, 100])
def test_fused_extras_with_hub_rows(ctx, R):
    """Epilogue appended as its own launch when rows are completed by several groups (hub-row segments); same numbers as
    the in-launch epilogue."""
    from distributed_sddmm_amd import _kernels as K
    lib = ctx.lib
    rng = np.random.default_rng(R + 11)
    rows, cols = 40, 6000
    lens = rng.integers(0, 30, rows); lens[3], lens[20] = 3000, 1100
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cidx = np.concatenate([np.sort(rng.choice(cols, n, replace=False)) for n in lens]).astype(np.int32)
    ridx = np.repeat(np.arange(rows, dtype=np.int32), lens)
    X, Y = rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, (cols, R))
    v0, out0 = rng.uniform(-1, 1, len(cidx)), rng.uniform(-1, 1, (rows, R))
    d_rp, d_c, dv, dX, dY, dOut, ddot = (ctx.upload(a) for a in (rowptr, cidx, v0, X, Y, out0, np.zeros(rows)))
    flags = K.FUSED_VALUES_OVERWRITE | K.FUSED_OUT_OVERWRITE | K.FUSED_LEAKY_RELU
    ex = K.FusedExtras(0.2, 1e-2, ddot.ptr)
    ctx.check(lib.hnh_fused_sddmm_spmm_csr_x(ctx.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dY.ptr, dOut.ptr, R, flags, -1, -1,
                                             cols, C.byref(ex), 0), "fused_x hub")
    vals, out, dot = _extras_expected(rowptr, ridx, cidx, v0, None, X, Y, out0, flags, 0.2, 1e-2)
    assert rel(dv.get(), vals) <= TOL and rel(dOut.get(), out) <= TOL and rel(ddot.get(), dot) <= TOL
    for d in (d_rp, d_c, dv, dX, dY, dOut, ddot):
        d.free()


@pytest.mark.parametrize("R", [8, 128, 100, 256])
@pytest.mark.parametrize("panel_bytes", [20000, 150000])
@pytest.mark.parametrize("hubs", [False, True])
def test_infinity_cache_panels_do_not_change_results(monkeypatch, R, panel_bytes, hubs):
    """With the `cols` hint a pass runs as one launch per column panel of the block (a contiguous piece of every CSR row);
    the panel size is shrunk here so that small blocks get 2..8 panels.  sddmm / spmm / fused (+ extras) must match."""
    monkeypatch.setenv("HNH_PANEL_BYTES", str(panel_bytes))
    monkeypatch.setenv("HNH_MAX_PANELS", "8")  # (the default cap is 4, where the Infinity Cache stops paying for the re-read rows)
    if hubs:
        monkeypatch.setenv("HNH_PANELS_WITH_HUBS", "1")  # off by default: no gain on skewed graphs
    from distributed_sddmm_amd import _kernels as K
    c = K.Ctx(0)
    lib = c.lib
    rng = np.random.default_rng(R)
    if hubs:  # hub rows stay whole (long-row pass, once) while the short rows are panelled
        rows, cols = 120, 3000
        lens = rng.integers(0, 40, rows)
        lens[5], lens[77] = 2500, 1100
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        cidx = np.concatenate([np.sort(rng.choice(cols, n, replace=False)) for n in lens]).astype(np.int32)
        ridx = np.repeat(np.arange(rows, dtype=np.int32), lens)
        panel_bytes *= 10
    else:
        rows, cols = 257, 300
        rowptr, ridx, cidx = random_block(rows, cols, 6000, seed=R + 1)
    nnz = len(cidx)
    X, Y = rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, (cols, R))
    v0, out0 = rng.uniform(-1, 1, nnz), rng.uniform(-1, 1, (rows, R))
    d_rp, d_c, dv, dX, dY, dOut, ddot = (c.upload(a) for a in (rowptr, cidx, v0, X, Y, out0, np.zeros(rows)))
    mx = int(np.diff(rowptr).max())
    c.check(lib.hnh_sddmm_csr_ex(c.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, dX.ptr, dY.ptr, R, nnz, mx, cols, 0), "sddmm")
    assert rel(dv.get(), O.sddmm_local(ridx, cidx, v0, X, Y)) <= TOL
    dv.set(v0)
    c.check(lib.hnh_spmm_csr_ex(c.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, dY.ptr, dOut.ptr, R, nnz, mx, cols, 0), "spmm")
    assert rel(dOut.get(), O.spmm_local(rowptr, cidx, v0, Y, out0)) <= TOL
    OW = K.FUSED_VALUES_OVERWRITE | K.FUSED_OUT_OVERWRITE
    for flags, alpha, xs in [(OW | K.FUSED_LEAKY_RELU, 0.2, 0.0), (OW, 0.0, 1e-3), (0, 0.0, -2.0), (K.FUSED_LEAKY_RELU, 0.1, 0.5)]:
        dv.set(v0); dOut.set(out0)
        ex = K.FusedExtras(alpha, xs, ddot.ptr)
        c.check(lib.hnh_fused_sddmm_spmm_csr_x(c.h, rows, d_rp.ptr, d_c.ptr, dv.ptr, None, dX.ptr, dY.ptr, dOut.ptr, R, flags, nnz, mx, cols,
                                               C.byref(ex), 0), "fused_x")
        vals, out, dot = _extras_expected(rowptr, ridx, cidx, v0, None, X, Y, out0, flags, alpha, xs)
        assert rel(dv.get(), vals) <= TOL and rel(dOut.get(), out) <= TOL and rel(ddot.get(), dot) <= TOL
    for d in (d_rp, d_c, dv, dX, dY, dOut, ddot):
        d.free()
    c.close()


@pytest.mark.parametrize("R", [8, 16, 32])
@pytest.mark.parametrize("shift", [0, 1, 16])
def test_narrow_rows_line_granular_streams(ctx, R, shift):
    """R = 8 / 16 / 32 (16 / 8 / 4 sparse rows per wave) go through the line-granular loop of process_row when colidx / values
    are 128-byte aligned: trips of 16 nonzeros aligned to the lines of `values`, 32-index blocks, clamped first / last trips.
    Rows of every length class here — empty, shorter than a trip, straddling lines, several blocks long, hub rows that go to
    the long-row pass — for SDDMM (accumulating), SpMM (beta = 1), fused (accumulate / overwrite + svalues / LeakyReLU).
    `shift` elements in front of colidx / values: 0 = aligned (line-granular loop), 1 = only 4- / 8-byte aligned (the general
    loop must take over), 16 = values line-aligned, colidx only 64-byte aligned (general loop again)."""
    from distributed_sddmm_amd import _kernels as K
    lib = ctx.lib
    rng = np.random.default_rng(100 * R + shift)
    rows, cols = 157, 3000
    lens = rng.choice([0, 1, 5, 15, 16, 17, 31, 32, 33, 47, 64, 96, 100, 130, 200], rows)
    lens[11], lens[90] = 2600, 700  # hub rows (threshold 3 x mean, within [256, 1024])
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cidx = np.concatenate([np.sort(rng.choice(cols, int(n), replace=False)) for n in lens]).astype(np.int32)
    ridx = np.repeat(np.arange(rows, dtype=np.int32), lens)
    nnz = len(cidx)
    X, Y = rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, (cols, R))
    v0, out0, sv = rng.uniform(-1, 1, nnz), rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, nnz)
    d_rp, dX, dY = ctx.upload(rowptr), ctx.upload(X), ctx.upload(Y)
    d_c = ctx.upload(np.concatenate([np.zeros(shift, np.int32), cidx]))
    dv = ctx.upload(np.concatenate([np.zeros(shift), v0]))
    dsv = ctx.upload(np.concatenate([np.zeros(shift), sv]))
    c_ptr, v_ptr, sv_ptr = d_c.ptr + 4 * shift, dv.ptr + 8 * shift, dsv.ptr + 8 * shift
    mx = int(lens.max())

    def set_v(a):
        dv.set(np.concatenate([np.zeros(shift), a]))

    def get_v():
        return dv.get()[shift:]

    for hinted in (False, True):
        hint = (nnz, mx, cols) if hinted else (-1, -1, -1)
        set_v(v0)
        ctx.check(lib.hnh_sddmm_csr_ex(ctx.h, rows, d_rp.ptr, c_ptr, v_ptr, dX.ptr, dY.ptr, R, *hint, 0), "sddmm")
        assert rel(get_v(), O.sddmm_local(ridx, cidx, v0, X, Y)) <= TOL
        set_v(v0)
        dOut = ctx.upload(out0)
        ctx.check(lib.hnh_spmm_csr_ex(ctx.h, rows, d_rp.ptr, c_ptr, v_ptr, dY.ptr, dOut.ptr, R, *hint, 0), "spmm")
        assert rel(dOut.get(), O.spmm_local(rowptr, cidx, v0, Y, out0)) <= TOL
        assert np.array_equal(get_v(), v0)
        # fused, accumulate semantics
        vals_after = O.sddmm_local(ridx, cidx, v0, X, Y)
        dOut.set(out0)
        ctx.check(lib.hnh_fused_sddmm_spmm_csr_ex(ctx.h, rows, d_rp.ptr, c_ptr, v_ptr, None, dX.ptr, dY.ptr, dOut.ptr, R, 0, *hint, 0), "fused")
        assert rel(get_v(), vals_after) <= TOL
        assert rel(dOut.get(), O.spmm_local(rowptr, cidx, vals_after, Y, out0)) <= TOL
        # fused, overwrite + svalues + LeakyReLU (the activated, scaled weight is what is stored)
        dots = O.sddmm_local(ridx, cidx, np.zeros(nnz), X, Y) * sv
        act = np.where(dots > 0, dots, 0.3 * dots)
        set_v(v0); dOut.set(out0)
        ex = K.FusedExtras(0.3, 0.0, None)
        ctx.check(lib.hnh_fused_sddmm_spmm_csr_x(ctx.h, rows, d_rp.ptr, c_ptr, v_ptr, sv_ptr, dX.ptr, dY.ptr, dOut.ptr, R,
                                                 K.FUSED_VALUES_OVERWRITE | K.FUSED_OUT_OVERWRITE | K.FUSED_LEAKY_RELU, *hint, C.byref(ex), 0), "fused x")
        assert rel(get_v(), act) <= TOL
        assert rel(dOut.get(), O.spmm_local(rowptr, cidx, act, Y, np.zeros((rows, R)))) <= TOL
        dOut.free()
    # nothing in front of the arrays was touched
    assert np.all(d_c.get()[:shift] == 0) and np.all(dv.get()[:shift] == 0)
    for d in (d_rp, d_c, dX, dY, dv, dsv):
        d.free()


@pytest.mark.parametrize("R", [8, 16, 32, 64, 100, 128, 256, 257, 600])
@pytest.mark.parametrize("shift", [0, 3])
def test_sddmm_with_the_hadamard_folded_in(ctx, R, shift):
    """hnh_sddmm_csr_ps: dst[e] (+)= scale[e] * <X[i_e,:], Y[j_e,:]> — storing into garbage (first visits) and accumulating, over the
    whole block and window by window, with hub rows, a structure plan, line-aligned arrays (narrow instances at R = 8 / 16 / 32)
    and arrays that start 3 elements into a line (`shift`; the general loop), widths that take the column-tiled fallback
    (257, 600: every tile scales its partial dot product).  scale = NULL is hnh_sddmm_csr_p; scale aliasing dst is refused."""
    from distributed_sddmm_amd import _kernels as K
    lib = ctx.lib
    rng = np.random.default_rng(7 * R + shift)
    rows, cols = 211, 5000
    lens = rng.choice([0, 1, 7, 16, 17, 33, 64, 90, 130], rows)
    lens[4], lens[120] = 3000, 800  # hub rows
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cidx = np.concatenate([np.sort(rng.choice(cols, int(n), replace=False)) for n in lens]).astype(np.int32)
    ridx = np.repeat(np.arange(rows, dtype=np.int32), lens)
    nnz = len(cidx)
    X, Y = rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, (cols, R))
    sv, v0 = rng.uniform(-1, 1, nnz), rng.uniform(-1, 1, nnz)
    dots = O.sddmm_local(ridx, cidx, np.zeros(nnz), X, Y)
    d_rp, d_c, dX, dY = ctx.upload(rowptr), ctx.upload(cidx), ctx.upload(X), ctx.upload(Y)
    pad = np.zeros(shift)
    d_dst, d_sv = ctx.upload(np.concatenate([pad, np.full(nnz, 1e300)])), ctx.upload(np.concatenate([pad, sv]))
    dst_ptr, sv_ptr = d_dst.ptr + 8 * shift, d_sv.ptr + 8 * shift
    plan = C.c_void_p()
    ctx.check(lib.hnh_csr_plan_create(ctx.h, C.byref(plan)), "plan")
    blk = K.CsrBlock(rows, nnz, cols, int(lens.max()), 0, d_rp.ptr, d_c.ptr, plan)
    # first visit: stored over garbage
    ctx.check(lib.hnh_sddmm_csr_ps(ctx.h, C.byref(blk), dst_ptr, sv_ptr, dX.ptr, dY.ptr, R, K.FUSED_VALUES_OVERWRITE, None, 0), "sddmm_ps")
    assert rel(d_dst.get()[shift:], sv * dots) <= TOL
    # a later visit adds
    d_dst.set(np.concatenate([pad, v0]))
    ctx.check(lib.hnh_sddmm_csr_ps(ctx.h, C.byref(blk), dst_ptr, sv_ptr, dX.ptr, dY.ptr, R, 0, None, 0), "sddmm_ps")
    assert rel(d_dst.get()[shift:], v0 + sv * dots) <= TOL
    # window by window, each window's first visit storing
    bounds = np.array([1200, 1200, 3700], dtype=np.int32)
    nw = len(bounds) + 1
    d_split = ctx.upload(np.zeros((len(bounds), rows), np.int32))
    ctx.check(lib.hnh_csr_window_bounds(ctx.h, rows, d_rp.ptr, d_c.ptr, len(bounds), bounds.ctypes.data_as(C.c_void_p), d_split.ptr, 0), "bounds")
    d_dst.set(np.concatenate([pad, np.full(nnz, -1e300)]))
    noplan = K.CsrBlock(rows, nnz, cols, int(lens.max()), 0, d_rp.ptr, d_c.ptr, None)
    for q in range(nw):
        beg = None if q == 0 else d_split.ptr + (q - 1) * rows * 4
        end = None if q == nw - 1 else d_split.ptr + q * rows * 4
        win = K.CsrWindow(beg, end, 1 if q == nw - 1 else 0)
        ctx.check(lib.hnh_sddmm_csr_ps(ctx.h, C.byref(noplan), dst_ptr, sv_ptr, dX.ptr, dY.ptr, R, K.FUSED_VALUES_OVERWRITE, C.byref(win), 0), "sddmm_ps w")
    assert rel(d_dst.get()[shift:], sv * dots) <= TOL
    # no scale: the plain SDDMM; an aliased scale: refused
    d_dst.set(np.concatenate([pad, v0]))
    ctx.check(lib.hnh_sddmm_csr_ps(ctx.h, C.byref(blk), dst_ptr, None, dX.ptr, dY.ptr, R, 0, None, 0), "sddmm_ps")
    assert rel(d_dst.get()[shift:], v0 + dots) <= TOL
    assert lib.hnh_sddmm_csr_ps(ctx.h, C.byref(blk), dst_ptr, dst_ptr, dX.ptr, dY.ptr, R, 0, None, 0) != 0
    # nothing in front of the arrays was touched, the scale array is unchanged
    assert np.all(d_dst.get()[:shift] == 0) and np.array_equal(d_sv.get()[shift:], sv)
    ctx.check(lib.hnh_csr_plan_destroy(ctx.h, plan), "plan destroy")
    for d in (d_rp, d_c, dX, dY, d_dst, d_sv, d_split):
        d.free()


@pytest.mark.parametrize("R", [320, 384, 448, 512, 640])
@pytest.mark.parametrize("slabs", ["1", "0"])
@pytest.mark.parametrize("hubs", [False, True])
def test_wide_operands_in_128_column_slabs(monkeypatch, R, slabs, hubs):
    """R >= 320 (a multiple of 64): the un-fused SDDMM / SpMM run as 128-column slabs (and a last one of 64 columns), each slab the R = 128 pass with row pitch R and its
    own Infinity-Cache panels (shrunk here so that the small block gets several); later slabs ADD their partial dot products, storing
    first visits and the folded Hadamard scale apply per slab, SpMM slabs store or add their own columns, hub rows go through the long-row
    pass of every slab.  HNH_WIDE_SLABS=0 = the single wide pass; same results."""
    monkeypatch.setenv("HNH_WIDE_SLABS", slabs)
    monkeypatch.setenv("HNH_PANEL_BYTES", "400000")
    monkeypatch.setenv("HNH_MAX_PANELS", "8")
    if hubs:
        monkeypatch.setenv("HNH_PANELS_WITH_HUBS", "1")
    from distributed_sddmm_amd import _kernels as K
    c = K.Ctx(0)
    lib = c.lib
    rng = np.random.default_rng(R + 7)
    rows, cols = 190, 2000
    lens = rng.integers(0, 60, rows)
    if hubs:
        lens[3], lens[150] = 1900, 900
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cidx = np.concatenate([np.sort(rng.choice(cols, int(n), replace=False)) for n in lens]).astype(np.int32)
    ridx = np.repeat(np.arange(rows, dtype=np.int32), lens)
    nnz = len(cidx)
    X, Y = rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, (cols, R))
    sv, v0, out0 = rng.uniform(-1, 1, nnz), rng.uniform(-1, 1, nnz), rng.uniform(-1, 1, (rows, R))
    dots = O.sddmm_local(ridx, cidx, np.zeros(nnz), X, Y)
    d_rp, d_c, dX, dY, dv, d_sv, dOut = (c.upload(a) for a in (rowptr, cidx, X, Y, v0, sv, out0))
    plan = C.c_void_p()
    c.check(lib.hnh_csr_plan_create(c.h, C.byref(plan)), "plan")
    blk = K.CsrBlock(rows, nnz, cols, int(lens.max()), 0, d_rp.ptr, d_c.ptr, plan)
    for rep in range(2):  # (second round: the plan's panel boundaries are reused)
        dv.set(np.full(nnz, 1e300))
        c.check(lib.hnh_sddmm_csr_ps(c.h, C.byref(blk), dv.ptr, d_sv.ptr, dX.ptr, dY.ptr, R, K.FUSED_VALUES_OVERWRITE, None, 0), "sddmm_ps store")
        assert rel(dv.get(), sv * dots) <= TOL
        dv.set(v0)
        c.check(lib.hnh_sddmm_csr_ps(c.h, C.byref(blk), dv.ptr, d_sv.ptr, dX.ptr, dY.ptr, R, 0, None, 0), "sddmm_ps add")
        assert rel(dv.get(), v0 + sv * dots) <= TOL
        dv.set(v0)
        c.check(lib.hnh_sddmm_csr_p(c.h, C.byref(blk), dv.ptr, dX.ptr, dY.ptr, R, 0, None, 0), "sddmm_p")
        assert rel(dv.get(), v0 + dots) <= TOL
        dv.set(v0)
        dOut.set(out0)
        c.check(lib.hnh_spmm_csr_p(c.h, C.byref(blk), dv.ptr, dY.ptr, dOut.ptr, R, None, 0), "spmm_p")
        assert rel(dOut.get(), O.spmm_local(rowptr, cidx, v0, Y, out0)) <= TOL
        dOut.set(np.full((rows, R), -1e300))
        c.check(lib.hnh_spmm_csr_pf(c.h, C.byref(blk), dv.ptr, dY.ptr, dOut.ptr, R, K.FUSED_OUT_OVERWRITE, None, 0), "spmm_pf store")
        assert rel(dOut.get(), O.spmm_local(rowptr, cidx, v0, Y, np.zeros((rows, R)))) <= TOL
    c.check(lib.hnh_csr_plan_destroy(c.h, plan), "plan destroy")
    for d in (d_rp, d_c, dX, dY, dv, d_sv, dOut):
        d.free()
    c.close()


def test_sum_chunked_blocks(ctx):
    """The closing step of the mesh reduce-scatter: dst rows of chunks [q0, q1) += the nb partial blocks of the chunk-major buffer, block order."""
    lib = ctx.lib
    rng = np.random.default_rng(11)
    for R, cuts in ((128, [0, 5, 5, 40, 77]), (7, [0, 3, 30]), (16, [0, 64])):
        nb, rows = 5, cuts[-1]
        dst0 = rng.uniform(-1, 1, (rows, R))
        src = rng.uniform(-1, 1, (nb * rows, R))
        d_dst, d_src = ctx.upload(dst0), ctx.upload(src)
        ch = np.array(cuts, dtype=np.int64)
        nch = len(cuts) - 1
        for q0, q1 in ((0, nch), (1, nch), (0, 1)):
            d_dst.set(dst0)
            ctx.check(lib.hnh_sum_chunked_blocks_f64(ctx.h, d_dst.ptr, d_src.ptr, nb, nch, ch.ctypes.data_as(C.c_void_p), q0, q1, R, 0), "sum_chunked")
            want = dst0.copy()
            for q in range(q0, q1):
                w = cuts[q + 1] - cuts[q]
                for k in range(nb):  # (block order: the sum is taken in this order on the device too)
                    want[cuts[q]:cuts[q + 1]] += src[nb * cuts[q] + k * w: nb * cuts[q] + (k + 1) * w]
            assert np.array_equal(d_dst.get(), want)
        d_dst.free(); d_src.free()


def test_stream_delay_and_paced_copy(ctx):
    """The measurement stand-ins of the overlap probe: hnh_stream_delay_us holds a stream for at least the requested time (and not for many times as long),
    hnh_stream_paced_copy delivers every slice bit for bit and takes at least the modelled time."""
    import time
    lib = ctx.lib
    ctx.sync(1)
    t0 = time.perf_counter()
    ctx.check(lib.hnh_stream_delay_us(ctx.h, 1, 20000.0), "delay")
    ctx.sync(1)
    dt = time.perf_counter() - t0
    assert 0.020 <= dt <= 0.200, dt  # (the upper bound only catches a clock-rate mistake: a loaded box may add scheduling time)
    n = 3 * 4096 + 2  # doubles per slice: not a multiple of the copy tile
    src = np.random.default_rng(3).uniform(-1, 1, n)
    d_src, d_dst = ctx.upload(src), ctx.upload(np.zeros(5 * n))
    for wgs, us in ((1, 0.0), (3, 3000.0)):
        d_dst.set(np.zeros(5 * n))
        t0 = time.perf_counter()
        ctx.check(lib.hnh_stream_paced_copy(ctx.h, 1, d_dst.ptr, d_src.ptr, n * 8, 5, us, wgs), "paced copy")
        ctx.sync(1)
        assert time.perf_counter() - t0 >= us * 1e-6
        assert np.array_equal(d_dst.get(), np.tile(src, 5))
    assert lib.hnh_stream_paced_copy(ctx.h, 1, d_dst.ptr, d_src.ptr, 24, 5, 0.0, 1) != 0  # slices must be multiples of 16 bytes
    d_src.free(); d_dst.free()


@pytest.mark.parametrize("mode", [1, 0])  # HNH_IPC_PULL_KERNEL, HNH_IPC_PULL_ENGINE
def test_ipc_pull_copies_every_alignment_class(ctx, mode):
    """hnh_ipc_pull is a public entry point used for every exchange of the ipc-pull transport, byte-displaced slices included
    (device_alltoallv, staged host exchanges): sources and destinations that are 16-, 8-, 4-byte aligned or not aligned at all, sizes
    that are not multiples of the element the copy uses, several copies per launch — every byte arrives and no byte beside the
    destination changes.  (In-process pointers: what crosses a process boundary is only where `src` came from.)"""
    import ctypes as C
    lib = ctx.lib
    rng = np.random.default_rng(7)
    n_buf = 1 << 16
    src_host = rng.integers(0, 256, n_buf, dtype=np.uint8)
    d_src, d_dst = ctx.upload(src_host), ctx.upload(np.full(n_buf, 0xEE, np.uint8))
    cases = [(0, 0, 4096), (16, 32, 4000), (8, 24, 1000), (8, 8, 1003), (4, 12, 996), (4, 4, 999), (1, 3, 777), (3, 1, 5), (7, 2, 1), (0, 5, 2049)]
    dst_ptrs, src_ptrs, sizes, placed = [], [], [], []
    at_s, at_d = 0, 0
    for so, do, nb in cases:
        s0 = (at_s + 255) // 256 * 256 + so
        d0 = (at_d + 255) // 256 * 256 + do
        src_ptrs.append(d_src.ptr + s0); dst_ptrs.append(d_dst.ptr + d0); sizes.append(nb); placed.append((s0, d0, nb))
        at_s, at_d = s0 + nb + 64, d0 + nb + 64
    n = len(cases)
    a_dst = (C.c_void_p * n)(*dst_ptrs); a_src = (C.c_void_p * n)(*src_ptrs); a_sz = (C.c_size_t * n)(*sizes)
    ctx.check(lib.hnh_ipc_pull(ctx.h, 0, n, a_dst, a_src, a_sz, mode, 4), "hnh_ipc_pull")
    ctx.sync()
    got = d_dst.get()
    want = np.full(n_buf, 0xEE, np.uint8)
    for s0, d0, nb in placed:
        want[d0:d0 + nb] = src_host[s0:s0 + nb]
    assert np.array_equal(got, want), [(s0, d0, nb) for s0, d0, nb in placed if not np.array_equal(got[d0:d0 + nb], src_host[s0:s0 + nb])]
    d_src.free(); d_dst.free()
