"""fusedSpMM_out — the out-of-place fused pass with the applications' surrounding work folded in (LeakyReLU between the
halves: gat.hpp:96-99; `+ lambda X` and the row-wise <X, Out>: als_conjugate_gradients.cpp:93,282,295).

CPU ranks (loopback transport + the oracle's C test double): the schedule logic — which launch carries the epilogue,
mesh / relay rings, c > 1 reduce-scatter first, null blocks — against a numpy statement of the result; and the test
double's own `_x` entry points against that same statement, so the GPU parity tests (which use the formula) and the
host-logic tests (which use the double) are pinned to one definition."""
import ctypes as C

import numpy as np
import pytest

import hnh_testlib as T
from distributed_sddmm_amd import _kernels as K
from distributed_sddmm_amd import api as H
from oracle import oracle as O


@pytest.fixture(autouse=True, scope="module")
def cpu_test_double():
    assert H.load_backend(T.ORACLE_BACKEND) == "oracle-cpu-test-double"
    yield


EXTRAS = [(0.2, 0.0, False), (None, 1e-3, True), (0.05, -0.5, True), (None, 0.0, False)]


@pytest.mark.parametrize("p,c", [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)])
@pytest.mark.parametrize("ring", ["mesh", "relay"])
def test_fusion2_out_of_place_with_extras(p, c, ring, monkeypatch):
    monkeypatch.setenv("HNH_RING_MODE", ring)
    case = T.case_inputs("er8_r16")
    for matmode in (H.AMAT, H.BMAT):
        for alpha, xs, dot in EXTRAS:
            per_rank = H.run_spmd(p, lambda w: T.run_fused_out(w, "15d_fusion2", c, case, matmode, alpha, xs, dot))
            assert all(o["supported"] for o in per_rank)
            T.check_fused_out(per_rank, case, matmode, alpha, xs, dot)


@pytest.mark.parametrize("case_name", ["ragged_r8", "rect_r16", "tiny_r8"])
def test_edge_cases(case_name):
    """padded blocks, non-square S (X and Y of different heights), almost-empty S (null blocks still get the epilogue)"""
    case = T.case_inputs(case_name)
    for p, c in [(1, 1), (4, 1), (4, 2), (8, 1)]:
        for matmode in (H.AMAT, H.BMAT):
            per_rank = H.run_spmd(p, lambda w: T.run_fused_out(w, "15d_fusion2", c, case, matmode, 0.2, 0.75, True))
            T.check_fused_out(per_rank, case, matmode, 0.2, 0.75, True)


@pytest.mark.parametrize("alg", [a for a in H.ALGORITHMS if a != "15d_fusion2"])
def test_other_schedules_decline_and_do_nothing(alg):
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(1, lambda w: T.run_fused_out(w, alg, 1, case, H.AMAT, None, 1.0, True))
    assert not any(o["supported"] for o in per_rank)


def test_c_test_double_extras_follow_the_numpy_statement():
    lib = C.CDLL(T.ORACLE_BACKEND)
    ctx = C.c_void_p()
    assert lib.hnh_ctx_create(0, C.byref(ctx)) == 0
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rng = np.random.default_rng(5)
    rows, cols, R = 60, 45, 12
    keys = np.unique(rng.integers(0, rows * cols, 500))
    ridx, cidx = (keys // cols).astype(np.int32), (keys % cols).astype(np.int32)
    rowptr = np.zeros(rows + 1, np.int32); np.add.at(rowptr, ridx + 1, 1); rowptr = np.cumsum(rowptr).astype(np.int32)
    X, Y = rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, (cols, R))
    v0, out0, sv = rng.uniform(-1, 1, len(keys)), rng.uniform(-1, 1, (rows, R)), rng.uniform(-1, 1, len(keys))
    for flags, use_sv, alpha, xs in [(3 | K.FUSED_LEAKY_RELU, False, 0.2, 0.0), (K.FUSED_LEAKY_RELU, True, 0.01, 0.5), (3, False, 0.0, 1e-3),
                                     (0, True, 0.0, -2.0)]:
        v, out, dot = v0.copy(), out0.copy(), np.zeros(rows)
        ex = K.FusedExtras(alpha, xs, dot.ctypes.data)
        assert lib.hnh_fused_sddmm_spmm_csr_x(ctx, C.c_int64(rows), p(rowptr), p(cidx), p(v), p(sv) if use_sv else None, p(X), p(Y), p(out),
                                              R, C.c_uint(flags), C.c_int64(-1), -1, C.c_int64(-1), C.byref(ex), 0) == 0
        vb = np.zeros(len(keys)) if flags & 1 else v0
        ob = np.zeros((rows, R)) if flags & 2 else out0
        vals = O.sddmm_local(ridx, cidx, vb, X, Y)
        if flags & K.FUSED_LEAKY_RELU:
            vals = vals * (sv if use_sv else 1.0)
            vals = np.where(vals > 0, vals, alpha * vals)
            w = vals
        else:
            w = vals * (sv if use_sv else 1.0)
        want = O.spmm_local(rowptr, cidx, w, Y, ob) + xs * X
        assert T.rel(v, vals) <= T.TOL and T.rel(out, want) <= T.TOL and T.rel(dot, np.einsum("ij,ij->i", X, want)) <= T.TOL
    Xm, Rm, P, MP = (rng.uniform(-1, 1, (rows, R)) for _ in range(4))
    al, rs = rng.uniform(-1, 1, rows), np.zeros(rows)
    x2, r2 = Xm.copy(), Rm.copy()
    assert lib.hnh_cg_step_f64(ctx, p(x2), p(r2), p(P), p(MP), p(al), p(rs), C.c_int64(rows), R, 0) == 0
    rn = Rm - al[:, None] * MP
    assert T.rel(x2, Xm + al[:, None] * P) <= T.TOL and T.rel(r2, rn) <= T.TOL and T.rel(rs, np.einsum("ij,ij->i", rn, rn)) <= T.TOL
    lib.hnh_ctx_destroy(ctx)


@pytest.mark.parametrize("alg", ["15d_fusion2", "15d_fusion1"])
@pytest.mark.parametrize("p,ring", [(2, "mesh"), (4, "mesh"), (4, "relay"), (8, "mesh")])
def test_held_operand_is_fetched_once(alg, p, ring, monkeypatch):
    """hold_moving_operand: repeated calls give the same results as without the hint; and the hint really is used — when
    the caller breaks the promise (changes the held matrix) the remote blocks of the FIRST fetch are what later calls see
    (mesh fetch, or a ring of two), while the relay ring of more than two ranks ignores the hint."""
    monkeypatch.setenv("HNH_RING_MODE", ring)
    case = T.case_inputs("er8_r16")

    def body(w):
        sp = H.SpmatLocal.from_global(w, case["M"], case["N"], case["rows"], case["cols"], np.ones(len(case["rows"])))
        d = H.DistributedSparse(w, alg, sp, case["R"], 1)
        subA, subB = d.submatrices(H.AMAT), d.submatrices(H.BMAT)
        A, B = d.like_A_matrix(0.0), d.like_B_matrix(0.0)
        ones, buf = d.like_S_values(1.0), d.like_S_values(0.0)
        a0, b0 = T.fill_local(subA, A.shape, case["A"]), T.fill_local(subB, B.shape, case["B"])
        outs = []
        B.upload(b0)
        for step in range(4):
            if step == 1:
                d.hold_moving_operand(B)
            if step == 3:
                B.upload(2.0 * b0)  # promise broken on purpose: remote ranks keep seeing the blocks fetched at step 1
            A.upload(a0)
            d.fusedSpMM(A, B, ones, buf, H.AMAT)
            outs.append(A.download())
        d.hold_moving_operand(None)
        B.upload(2.0 * b0); A.upload(a0)
        d.fusedSpMM(A, B, ones, buf, H.AMAT)
        outs.append(A.download())
        for h in (A, B, ones, buf):
            h.free()
        d.free(); sp.free()
        return dict(subA=subA, outs=outs)

    per_rank = H.run_spmd(p, body)
    glob = [T.assemble_dense([dict(subA=o["subA"], x=o["outs"][k]) for o in per_rank], "x", "subA", case["M"], case["R"]) for k in range(5)]
    want1, _ = T.fused_out_expected(case, H.AMAT, None, 0.0)
    case2 = dict(case, B=2.0 * case["B"])
    want2, _ = T.fused_out_expected(case2, H.AMAT, None, 0.0)
    for k in range(3):
        assert T.rel(glob[k], want1) <= T.TOL
    assert T.rel(glob[4], want2) <= T.TOL           # released: fresh fetch
    uses_hint = (alg == "15d_fusion2") and (ring == "mesh" or p == 2)
    if uses_hint:
        assert T.rel(glob[3], want2) > 1e-3 and T.rel(glob[3], want1) > 1e-3  # own block new, remote blocks held
    elif alg == "15d_fusion2":
        assert T.rel(glob[3], want2) <= T.TOL       # relay ring of > 2 ranks: hint ignored


@pytest.mark.parametrize("p,ring", [(4, "mesh"), (2, "relay"), (4, "relay")])
def test_held_operand_survives_other_operands(p, ring, monkeypatch):
    """The caller keeps the documented contract (the CONTENTS of the held matrix do not change) but moves another
    operand between two uses of the held one: hold(B); fused(A, B); fused(A, C); fused(A, B).  The landing buffers
    then hold C's blocks, so B has to be fetched again (round-1 advisor finding: the third call multiplied by C)."""
    monkeypatch.setenv("HNH_RING_MODE", ring)
    case = T.case_inputs("er8_r16")

    def body(w):
        sp = H.SpmatLocal.from_global(w, case["M"], case["N"], case["rows"], case["cols"], np.ones(len(case["rows"])))
        d = H.DistributedSparse(w, "15d_fusion2", sp, case["R"], 1)
        subA, subB = d.submatrices(H.AMAT), d.submatrices(H.BMAT)
        A, B, Cm = d.like_A_matrix(0.0), d.like_B_matrix(0.0), d.like_B_matrix(0.0)
        ones, buf = d.like_S_values(1.0), d.like_S_values(0.0)
        a0, b0 = T.fill_local(subA, A.shape, case["A"]), T.fill_local(subB, B.shape, case["B"])
        B.upload(b0)
        Cm.upload(-3.0 * b0)
        d.hold_moving_operand(B)
        outs = []
        for other in (B, Cm, B, B):
            A.upload(a0)
            d.fusedSpMM(A, other, ones, buf, H.AMAT)
            outs.append(A.download())
        d.hold_moving_operand(None)
        for h in (A, B, Cm, ones, buf):
            h.free()
        d.free(); sp.free()
        return dict(subA=subA, outs=outs)

    per_rank = H.run_spmd(p, body)
    glob = [T.assemble_dense([dict(subA=o["subA"], x=o["outs"][k]) for o in per_rank], "x", "subA", case["M"], case["R"]) for k in range(4)]
    wantB, _ = T.fused_out_expected(case, H.AMAT, None, 0.0)
    wantC, _ = T.fused_out_expected(dict(case, B=-3.0 * case["B"]), H.AMAT, None, 0.0)
    assert T.rel(glob[0], wantB) <= T.TOL
    assert T.rel(glob[1], wantC) <= T.TOL
    assert T.rel(glob[2], wantB) <= T.TOL
    assert T.rel(glob[3], wantB) <= T.TOL


@pytest.mark.parametrize("p,ring", [(4, "mesh"), (2, "relay")])
def test_hold_does_not_survive_a_change_of_width(p, ring, monkeypatch):
    """hold(B); fused(A, B); setRValue(R / 2) with a half-width call; setRValue(R); fused(A, B): the landing buffers (and the relay
    ring's spares) were re-allocated twice in between — the pooled allocator may hand back the very same addresses — so B's
    blocks have to travel again although B is still on hold (round-2 advisor finding)."""
    monkeypatch.setenv("HNH_RING_MODE", ring)
    case = T.case_inputs("er8_r16")
    R = case["R"]

    def body(w):
        sp = H.SpmatLocal.from_global(w, case["M"], case["N"], case["rows"], case["cols"], np.ones(len(case["rows"])))
        d = H.DistributedSparse(w, "15d_fusion2", sp, R, 1)
        subA, subB = d.submatrices(H.AMAT), d.submatrices(H.BMAT)
        A, B = d.like_A_matrix(0.0), d.like_B_matrix(0.0)
        ones, buf = d.like_S_values(1.0), d.like_S_values(0.0)
        a0, b0 = T.fill_local(subA, A.shape, case["A"]), T.fill_local(subB, B.shape, case["B"])
        B.upload(b0)
        d.hold_moving_operand(B)
        outs = []
        A.upload(a0)
        d.fusedSpMM(A, B, ones, buf, H.AMAT)
        outs.append(A.download())
        d.setRValue(R // 2)  # narrower operands: every landing / spare buffer is replaced, and filled with other data
        A2, B2 = d.like_A_matrix(0.5), d.like_B_matrix(-7.0)
        d.fusedSpMM(A2, B2, ones, buf, H.AMAT)
        d.setRValue(R)
        A.upload(a0)
        d.fusedSpMM(A, B, ones, buf, H.AMAT)
        outs.append(A.download())
        d.hold_moving_operand(None)
        for h in (A, B, A2, B2, ones, buf):
            h.free()
        d.free(); sp.free()
        return dict(subA=subA, outs=outs)

    per_rank = H.run_spmd(p, body)
    want, _ = T.fused_out_expected(case, H.AMAT, None, 0.0)
    for k in range(2):
        got = T.assemble_dense([dict(subA=o["subA"], x=o["outs"][k]) for o in per_rank], "x", "subA", case["M"], R)
        assert T.rel(got, want) <= T.TOL, k
