"""Shared body of the setup-primitive tests (hnh_tuples_* of include/hnh_kernels.h): the same checks run against the
oracle's C test double on the CPU and against the HIP library on the GPU; expectations are plain numpy."""
import ctypes as C

import numpy as np

from distributed_sddmm_amd import _kernels as K


def make_tuples(n, rows, cols, seed):
    rng = np.random.default_rng(seed)
    t = np.zeros(n, dtype=K.TUPLE_DTYPE)
    t["r"], t["c"] = rng.integers(0, rows, n), rng.integers(0, cols, n)
    t["value"] = rng.uniform(-1, 1, n)
    return t


def key_of(t, kind, **kw):
    r, c = t["r"].astype(np.uint64), t["c"].astype(np.uint64)
    if kind == K.KEY_ROW_COL:
        return (r << np.uint64(32)) | c
    if kind == K.KEY_COL_ROW:
        return (c << np.uint64(32)) | r
    if kind == K.KEY_OWNER:
        rr, cc = (c, r) if kw["transpose"] else (r, c)
        return kw["table"][(rr // np.uint64(kw["rib"])) * np.uint64(kw["ncb"]) + cc // np.uint64(kw["cib"])].astype(np.uint64)
    return c // np.uint64(kw["div"])


def run(api):
    """api: object with upload(np)->handle(.ptr,.get(),.free()), lib, ctx handle `h`, check(rc, what)."""
    lib, h = api.lib, api.h
    rows, cols, n = 1000, 777, 50000
    t0 = make_tuples(n, rows, cols, 3)
    table = np.random.default_rng(4).integers(0, 8, 10 * 7).astype(np.int32)  # 10 x 7 blocks of 100 x 111, 8 owners
    dtab = api.upload(table)
    cases = [(K.KEY_ROW_COL, {}, 64), (K.KEY_COL_ROW, {}, 32 + 10), (K.KEY_COL_DIV, dict(div=100), 4),
             (K.KEY_OWNER, dict(transpose=0, rib=100, cib=111, ncb=7, table=table), 3),
             (K.KEY_OWNER, dict(transpose=1, rib=111, cib=100, ncb=10, table=table), 3)]
    for kind, kw, bits in cases:
        if kind == K.KEY_OWNER and kw["transpose"]:  # 7 x 10 blocks over the transposed matrix
            kw["table"] = table[:70]
        key = K.TupleKey(kind, kw.get("transpose", 0), kw.get("rib", 0), kw.get("cib", 0), kw.get("ncb", 0), dtab.ptr, kw.get("div", 0))
        d = api.upload(t0)
        api.check(lib.hnh_tuples_sort(h, d.ptr, n, C.byref(key), bits, 0), "tuples_sort")
        got = d.get().view(K.TUPLE_DTYPE).reshape(-1)
        order = np.argsort(key_of(t0, kind, **kw), kind="stable")   # the sort is stable
        assert np.array_equal(got, t0[order]), "kind %d" % kind
        # boundaries
        nb = int(key_of(t0, kind, **kw).max()) + 1 if kind in (K.KEY_OWNER, K.KEY_COL_DIV) else 5
        starts = np.zeros(nb + 1, dtype=np.int64)
        api.check(lib.hnh_tuples_bucket_starts(h, d.ptr, n, C.byref(key), nb, starts.ctypes.data_as(C.c_void_p), 0), "bucket_starts")
        want = np.searchsorted(key_of(got, kind, **kw), np.arange(nb + 1, dtype=np.uint64), side="left")
        assert np.array_equal(starts, want)
        d.free()
    # transform: swap, then mod
    d = api.upload(t0)
    api.check(lib.hnh_tuples_transform(h, d.ptr, n, 1, 13, 0, 0), "transform")
    got = d.get().view(K.TUPLE_DTYPE).reshape(-1)
    assert np.array_equal(got["r"], t0["c"] % 13) and np.array_equal(got["c"], t0["r"]) and np.array_equal(got["value"], t0["value"])
    d.free()
    # remap_cols: segment (c / div) * n_sub + (c % div) / sub_div moves to dest[segment], offsets inside a segment stay
    div, sub, nsub = 100, 34, 3
    dest = np.random.default_rng(5).permutation(8 * nsub).astype(np.int64) * 1000   # 777 columns -> 8 blocks x 3 chunks
    d = api.upload(t0)
    api.check(lib.hnh_tuples_remap_cols(h, d.ptr, n, div, sub, nsub, dest.ctypes.data_as(C.c_void_p), len(dest), 0), "remap_cols")
    got = d.get().view(K.TUPLE_DTYPE).reshape(-1)
    c0 = t0["c"].astype(np.int64)
    seg = (c0 // div) * nsub + (c0 % div) // sub
    assert np.array_equal(got["c"].astype(np.int64), dest[seg] + (c0 % div) % sub) and np.array_equal(got["r"], t0["r"])
    # a tuple in a segment without destination (negative entry, or beyond the table) is an error
    bad = dest.copy(); bad[int(seg[0])] = -1
    d2 = api.upload(t0)
    assert lib.hnh_tuples_remap_cols(h, d2.ptr, n, div, sub, nsub, bad.ctypes.data_as(C.c_void_p), len(bad), 0) != 0
    assert lib.hnh_tuples_remap_cols(h, d2.ptr, n, div, sub, nsub, dest.ctypes.data_as(C.c_void_p), 3, 0) != 0
    d.free(); d2.free()
    # to_csr on de-duplicated (row, col)-ordered tuples, with empty rows and one hub row
    keys = np.unique(np.concatenate([key_of(t0, K.KEY_ROW_COL)[t0["r"] % 7 != 3], (np.uint64(5) << np.uint64(32)) | np.arange(cols, dtype=np.uint64)]))
    ts = np.zeros(len(keys), dtype=K.TUPLE_DTYPE)
    ts["r"], ts["c"], ts["value"] = keys >> np.uint64(32), keys & np.uint64(0xffffffff), np.arange(len(keys)) * 0.5
    d = api.upload(ts)
    drp, dci, dv = api.upload(np.zeros(rows + 1, np.int32)), api.upload(np.zeros(len(ts), np.int32)), api.upload(np.zeros(len(ts)))
    mx = C.c_int(-1)
    api.check(lib.hnh_tuples_to_csr(h, d.ptr, len(ts), rows, cols, drp.ptr, dci.ptr, dv.ptr, C.byref(mx), 0), "to_csr")
    want_rp = np.searchsorted(ts["r"], np.arange(rows + 1), side="left").astype(np.int32)
    assert np.array_equal(drp.get().reshape(-1), want_rp) and np.array_equal(dci.get().reshape(-1), ts["c"].astype(np.int32))
    assert np.array_equal(dv.get().reshape(-1), ts["value"]) and mx.value == int(np.diff(want_rp).max()) == cols
    # window bounds on that CSR block: first nonzero of every row with column >= bound
    bounds = np.array([0, 100, 100, 500, 776, 5000], dtype=np.int32)
    dsp = api.upload(np.zeros((len(bounds), rows), np.int32))
    api.check(lib.hnh_csr_window_bounds(h, rows, drp.ptr, dci.ptr, len(bounds), bounds.ctypes.data_as(C.c_void_p), dsp.ptr, 0), "window_bounds")
    ci = ts["c"].astype(np.int64)
    for b, bound in enumerate(bounds):
        want_split = np.array([want_rp[r] + np.searchsorted(ci[want_rp[r]:want_rp[r + 1]], bound, side="left") for r in range(rows)])
        assert np.array_equal(dsp.get().reshape(len(bounds), rows)[b], want_split.astype(np.int32))
    assert lib.hnh_csr_window_bounds(h, rows, drp.ptr, dci.ptr, 2, np.array([5, 3], np.int32).ctypes.data_as(C.c_void_p), dsp.ptr, 0) != 0
    dsp.free()
    # a tuple outside the block is an error, like the reference's MKL call would be
    assert lib.hnh_tuples_to_csr(h, d.ptr, len(ts), rows, cols - 1, drp.ptr, dci.ptr, dv.ptr, C.byref(mx), 0) != 0
    # empty input
    api.check(lib.hnh_tuples_to_csr(h, None, 0, 4, 4, drp.ptr, None, None, C.byref(mx), 0), "to_csr empty")
    assert np.array_equal(drp.get().reshape(-1)[:5], np.zeros(5, np.int32)) and mx.value == 0
    api.check(lib.hnh_tuples_sort(h, None, 0, C.byref(K.TupleKey(K.KEY_ROW_COL, 0, 0, 0, 0, None, 0)), 64, 0), "sort empty")
    # indices beyond 32 bits cannot be keyed
    big = t0[:10].copy(); big["r"][3] = 1 << 33
    db = api.upload(big)
    assert lib.hnh_tuples_sort(h, db.ptr, 10, C.byref(K.TupleKey(K.KEY_ROW_COL, 0, 0, 0, 0, None, 0)), 64, 0) != 0
    for x in (d, drp, dci, dv, dtab, db):
        x.free()
    run_generator(api)


def run_generator(api):
    """hnh_generate_er_keys / hnh_tuples_from_keys / hnh_tuples_relabel against oracle.py (the generator's numpy twin)."""
    from oracle import oracle as O
    lib, h = api.lib, api.h
    m, n, draws, seed = 300, 170, 4000, 99
    rows, cols = O.erdos_renyi_mn(m, n, draws, seed)
    dk = api.upload(np.zeros(draws, dtype=np.uint64))
    cnt = C.c_int64(-1)
    api.check(lib.hnh_generate_er_keys(h, m, n, draws, seed, dk.ptr, C.byref(cnt), 0), "generate_er_keys")
    assert cnt.value == len(rows) < draws  # duplicates were dropped
    keys = dk.get().reshape(-1)[:cnt.value]
    assert np.array_equal(keys, rows.astype(np.uint64) * np.uint64(n) + cols.astype(np.uint64))
    for rank, p in ((0, 1), (1, 3), (2, 3)):
        cnt_local = (cnt.value - rank + p - 1) // p
        dt = api.upload(np.zeros(cnt_local, dtype=K.TUPLE_DTYPE))
        api.check(lib.hnh_tuples_from_keys(h, dk.ptr, n, rank, p, 1.0, dt.ptr, cnt_local, 0), "tuples_from_keys")
        t = dt.get().view(K.TUPLE_DTYPE).reshape(-1)
        assert np.array_equal(t["r"], rows[rank::p].astype(np.uint64)) and np.array_equal(t["c"], cols[rank::p].astype(np.uint64))
        assert np.all(t["value"] == 1.0)
        rl, cl = O.vertex_permutation(m, 7).astype(np.uint64), O.vertex_permutation(n, 8).astype(np.uint64)
        drl, dcl = api.upload(rl), api.upload(cl)
        api.check(lib.hnh_tuples_relabel(h, dt.ptr, cnt_local, drl.ptr, dcl.ptr, 0), "tuples_relabel")
        t2 = dt.get().view(K.TUPLE_DTYPE).reshape(-1)
        assert np.array_equal(t2["r"], rl[rows[rank::p]]) and np.array_equal(t2["c"], cl[cols[rank::p]])
        for x in (dt, drl, dcl):
            x.free()
    dk.free()
    # the skewed initiator (hnh_generate_rmat_keys) against oracle.rmat, scrambled and not
    for logm, edges, abc, scramble in ((9, 6000, (0.57, 0.19, 0.19), 1), (7, 900, (0.45, 0.15, 0.15), 0), (10, 5000, (0.25, 0.25, 0.25), 1)):
        rows, cols = O.rmat(logm, edges, *abc, seed=77, scramble=bool(scramble))
        dk = api.upload(np.zeros(edges, dtype=np.uint64))
        cnt = C.c_int64(-1)
        api.check(lib.hnh_generate_rmat_keys(h, logm, edges, abc[0], abc[1], abc[2], 77, scramble, dk.ptr, C.byref(cnt), 0), "generate_rmat_keys")
        assert cnt.value == len(rows)
        assert np.array_equal(dk.get().reshape(-1)[:cnt.value], rows.astype(np.uint64) * np.uint64(1 << logm) + cols.astype(np.uint64))
        dk.free()
    cnt = C.c_int64(-1)
    assert lib.hnh_generate_rmat_keys(h, 9, 10, 0.6, 0.3, 0.3, 1, 1, None, C.byref(cnt), 0) != 0  # probabilities above one


def run_round6_primitives(api):
    """hnh_spmm_csr_pf (SpMM that STORES fresh output rows), hnh_sum_chunked_blocks_f64 (the closing step of the mesh reduce-scatter) and
    hnh_ctx_device_identity against numpy — the body the CPU test runs on the test double and the GPU test on the HIP library."""
    from oracle import oracle as O
    lib, h = api.lib, api.h
    rng = np.random.default_rng(5)
    # ---- SpMM with the store flag: rows without nonzeros end as zeros, garbage in Out is never read
    rows, cols, R = 90, 400, 48
    lens = rng.integers(0, 30, rows)
    lens[7] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cidx = np.concatenate([np.sort(rng.choice(cols, int(n), replace=False)) for n in lens]).astype(np.int32)
    vals, Y = rng.uniform(-1, 1, len(cidx)), rng.uniform(-1, 1, (cols, R))
    d_rp, d_c, d_v, dY = api.upload(rowptr), api.upload(cidx), api.upload(vals), api.upload(Y)
    blk = K.CsrBlock(rows, len(cidx), cols, int(lens.max()), 0, d_rp.ptr, d_c.ptr, None)
    want = O.spmm_local(rowptr, cidx, vals, Y, np.zeros((rows, R)))
    dOut = api.upload(np.full((rows, R), 1e300))
    api.check(lib.hnh_spmm_csr_pf(h, C.byref(blk), d_v.ptr, dY.ptr, dOut.ptr, R, K.FUSED_OUT_OVERWRITE, None, 0), "spmm_pf store")
    got = dOut.get().reshape(rows, R)
    assert np.max(np.abs(got - want)) <= 1e-11 * np.max(np.abs(want)) and np.all(got[7] == 0.0)
    out0 = rng.uniform(-1, 1, (rows, R))
    dOut2 = api.upload(out0)
    api.check(lib.hnh_spmm_csr_pf(h, C.byref(blk), d_v.ptr, dY.ptr, dOut2.ptr, R, 0, None, 0), "spmm_pf add")
    assert np.max(np.abs(dOut2.get().reshape(rows, R) - (out0 + want))) <= 1e-11 * np.max(np.abs(want))
    assert lib.hnh_spmm_csr_pf(h, C.byref(blk), d_v.ptr, dY.ptr, dOut2.ptr, R, 64, None, 0) != 0  # unknown flag
    # ---- the chunk-major sum, in block order (bit for bit)
    for R2, cuts in ((128, [0, 5, 5, 40, 77]), (7, [0, 3, 30]), (16, [0, 64])):
        nb, nrows = 5, cuts[-1]
        dst0, src = rng.uniform(-1, 1, (nrows, R2)), rng.uniform(-1, 1, (nb * nrows, R2))
        ch = np.array(cuts, dtype=np.int64)
        nch = len(cuts) - 1
        d_src = api.upload(src)
        for q0, q1 in ((0, nch), (1, nch), (0, 1)):
            d_dst = api.upload(dst0)
            api.check(lib.hnh_sum_chunked_blocks_f64(h, d_dst.ptr, d_src.ptr, nb, nch, ch.ctypes.data_as(C.c_void_p), q0, q1, R2, 0), "sum_chunked")
            want2 = dst0.copy()
            for q in range(q0, q1):
                w = cuts[q + 1] - cuts[q]
                for k in range(nb):
                    want2[cuts[q]:cuts[q + 1]] += src[nb * cuts[q] + k * w: nb * cuts[q] + (k + 1) * w]
            assert np.array_equal(d_dst.get().reshape(nrows, R2), want2)
            d_dst.free()
        d_src.free()
    bad = np.array([0, 5, 3], dtype=np.int64)
    assert lib.hnh_sum_chunked_blocks_f64(h, None, None, 2, 2, bad.ctypes.data_as(C.c_void_p), 0, 2, 8, 0) != 0  # cuts that decrease
    # ---- where the context runs
    ordinal, bus = C.c_int(-1), C.create_string_buffer(40)
    api.check(lib.hnh_ctx_device_identity(h, C.byref(ordinal), bus, 40), "device_identity")
    assert ordinal.value == 0 and len(bus.value) >= 7 and b":" in bus.value
    assert lib.hnh_ctx_device_identity(h, C.byref(ordinal), bus, 4) != 0  # a buffer too short for a bus id
    for x in (d_rp, d_c, d_v, dY, dOut, dOut2):
        x.free()
