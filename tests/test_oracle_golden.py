"""Pins the oracle: the numpy restatement (oracle/oracle.py) and the C restatement behind the kernel ABI
(oracle/hnh_oracle_backend.c) must reproduce the golden vectors that tests/golden/make_golden.py obtained
from the reference itself (all five schedules, several (p, c); deviations recorded in manifest.json)."""
import ctypes as C
import os

import numpy as np
import pytest

import hnh_testlib as T
from oracle import oracle as O

CASES = sorted(T.golden_cases())


def test_manifest_says_every_reference_variant_agreed():
    man = T.golden_cases()
    n_variants = 0
    for name, meta in man.items():
        for variant, dev in meta["deviation_from_canonical"].items():
            assert isinstance(dev, dict), "%s %s: %s" % (name, variant, dev)
            for k, v in dev.items():
                if isinstance(v, str):
                    assert v.startswith("not filled by the reference"), (name, variant, k, v)
                else:
                    assert v <= 1e-12, (name, variant, k, v)
            n_variants += 1
    assert n_variants >= 100  # 4 cases x 5 schedules x several grids


@pytest.mark.parametrize("name", CASES)
def test_numpy_oracle_matches_reference(name):
    c, g = T.case_inputs(name), T.golden_outputs(name)
    rows, cols, vals, a, b = c["rows"], c["cols"], c["vals"], c["A"], c["B"]
    assert T.rel(O.sddmm(rows, cols, vals, a, b), g["sddmmA"]) <= T.TOL
    assert T.rel(O.sddmm(rows, cols, vals, a, b), g["sddmmB"]) <= T.TOL
    assert T.rel(O.spmm_a(rows, cols, vals, b, c["M"]), g["spmmA"]) <= T.TOL
    assert T.rel(O.spmm_b(rows, cols, vals, a, c["N"]), g["spmmB"]) <= T.TOL
    fa, mid = O.fused_a(rows, cols, vals, a, b)
    fb, _ = O.fused_b(rows, cols, vals, a, b)
    assert T.rel(fa, g["fusedA"]) <= T.TOL and T.rel(fb, g["fusedB"]) <= T.TOL
    assert T.rel(mid, g["fusedA_buf"]) <= T.TOL
    fa2, _ = O.fused_a(rows, cols, vals, a, b, ignore_svalues=True)
    fb2, _ = O.fused_b(rows, cols, vals, a, b, ignore_svalues=True)
    assert T.rel(fa2, g["fusedA_fusion2"]) <= T.TOL and T.rel(fb2, g["fusedB_fusion2"]) <= T.TOL
    assert T.rel(np.array(O.fingerprints(rows, cols, c["M"], c["N"], c["R"])), g["fingerprints"]) <= T.TOL
    # the O(nnz) closed form used at full size is pinned to the same reference numbers (chunked path included)
    assert T.rel(np.array(O.fingerprints_closed_form(rows, cols, c["M"], c["N"], c["R"], chunk=97)), g["fingerprints"]) <= T.TOL


@pytest.mark.parametrize("name", CASES)
def test_c_restatement_matches_reference(name):
    """The plain-C kernels (same ABI as the HIP library) against the same golden vectors."""
    lib = C.CDLL(T.ORACLE_BACKEND)
    lib.hnh_backend_name.restype = C.c_char_p
    assert lib.hnh_backend_name() == b"oracle-cpu-test-double"
    c, g = T.case_inputs(name), T.golden_outputs(name)
    m, n, r = c["M"], c["N"], c["R"]
    rows, cols = c["rows"].astype(np.int32), c["cols"].astype(np.int32)
    rowptr = np.zeros(m + 1, np.int32)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    ctx = C.c_void_p()
    assert lib.hnh_ctx_create(0, C.byref(ctx)) == 0
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    a, b = np.ascontiguousarray(c["A"]), np.ascontiguousarray(c["B"])
    vals = np.zeros(len(rows))
    assert lib.hnh_sddmm_csr(ctx, C.c_int64(m), p(rowptr), p(cols), p(vals), p(a), p(b), r, 0) == 0
    assert T.rel(vals * c["vals"], g["sddmmA"]) <= T.TOL
    vals2 = np.zeros(len(rows))
    assert lib.hnh_sddmm_coo(ctx, C.c_int64(len(rows)), p(rows), p(cols), p(vals2), p(a), p(b), r, 0) == 0
    assert np.array_equal(vals, vals2)
    out = np.zeros((m, r))
    sv = np.ascontiguousarray(c["vals"])
    assert lib.hnh_spmm_csr(ctx, C.c_int64(m), p(rowptr), p(cols), p(sv), p(b), p(out), r, 0) == 0
    assert T.rel(out, g["spmmA"]) <= T.TOL
    out[:] = 7.0
    vals[:] = 3.0
    assert lib.hnh_fused_sddmm_spmm_csr(ctx, C.c_int64(m), p(rowptr), p(cols), p(vals), None, p(a), p(b), p(out), r, 3, 0) == 0
    assert T.rel(out, g["fusedA_fusion2"]) <= T.TOL
    # the SDDMM with its closing Hadamard folded in (hnh_sddmm_csr_ps) IS the reference's sddmmA: storing into garbage, and adding
    from distributed_sddmm_amd import _kernels as K
    blk = K.CsrBlock(m, len(rows), n, -1, 0, p(rowptr), p(cols), None)
    dst = np.full(len(rows), 1e300)
    assert lib.hnh_sddmm_csr_ps(ctx, C.byref(blk), p(dst), p(sv), p(a), p(b), r, K.FUSED_VALUES_OVERWRITE, None, 0) == 0
    assert T.rel(dst, g["sddmmA"]) <= T.TOL
    assert lib.hnh_sddmm_csr_ps(ctx, C.byref(blk), p(dst), p(sv), p(a), p(b), r, 0, None, 0) == 0
    assert T.rel(dst, 2 * g["sddmmA"]) <= T.TOL
    assert lib.hnh_sddmm_csr_ps(ctx, C.byref(blk), p(sv), p(sv), p(a), p(b), r, 0, None, 0) != 0  # scale must not alias the destination
    lib.hnh_ctx_destroy(ctx)


def test_closed_form_is_the_reference_at_config2_full_size():
    """The O(nnz) closed form that checks the full-size GPU runs, against the compiled reference's own fingerprints at BASELINE
    config 2's FULL size (ER 2^20, 100 658 766 nonzeros, R = 128; tests/golden/fullsize_reference.json, written by
    tests/golden/make_golden_fullsize.py from oracle/_ref/ref_driver: 195 s on 8 host cores) — about 20 s and 3 GB here."""
    import json
    from distributed_sddmm_amd import api as H
    path = os.path.join(T.GOLDEN, "fullsize_reference.json")
    rec = json.load(open(path)).get("config2_fingerprints")
    assert rec is not None, "tests/golden/fullsize_reference.json has no config2_fingerprints (run tests/golden/make_golden_fullsize.py config2)"
    m = 1 << rec["logm"]
    rows, cols = H.generate_er(m, m, m * rec["edge_factor"], rec["seed"])
    assert len(rows) == rec["nnz"] == 100658766
    closed = np.array(O.fingerprints_closed_form(rows, cols, m, m, rec["R"]))
    assert T.rel(closed, np.array(rec["values"])) <= T.TOL, (closed, rec["values"])
