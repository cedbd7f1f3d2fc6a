"""GAT forward pass (gat.hpp) on CPU ranks: the C++ GAT mirror (dense GEMM, SDDMM, LeakyReLU on the values, SpMM with
replication reuse, ReLU into the head's column block) compared with golden features produced by the REFERENCE's own
gat.hpp (tests/golden/make_golden_gat.py), plus the numpy restatement oracle.gat_forward (pinned by the same golden)."""
import os

import numpy as np
import pytest

import hnh_testlib as T
from distributed_sddmm_amd import api as H
from oracle import oracle as O


@pytest.fixture(autouse=True, scope="module")
def cpu_test_double():
    H.load_backend(T.ORACLE_BACKEND)
    yield


def gold():
    return dict(np.load(os.path.join(T.GOLDEN, "gat_er8_r16.npz")))


def test_numpy_restatement_matches_reference():
    case = T.case_inputs("er8_r16")
    want = O.gat_forward(case["rows"], case["cols"], case["M"], case["A"] * T.GAT_INPUT_SCALE, T.GAT_LAYERS, T.GAT_ALPHA)
    assert T.rel(want, gold()["out"]) <= T.TOL
    assert (want != 0).mean() > 0.3 and np.abs(want).max() > 1e-3, "the test must not be vacuous (the reference's own weights are zero)"


@pytest.mark.parametrize("alg,p,c", [("15d_fusion1", 1, 1), ("15d_fusion1", 4, 1), ("15d_fusion1", 4, 2), ("15d_fusion1", 8, 2),
                                     ("15d_fusion2", 1, 1), ("15d_fusion2", 4, 1),
                                     ("15d_fusion1", 6, 2), ("15d_fusion1", 9, 3), ("15d_fusion2", 5, 1)])  # (grids with remainders)
def test_gat_matches_reference(alg, p, c):
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(p, lambda w: T.run_gat(w, alg, c, case))
    out = T.assemble_dense(per_rank, "gat", "subA", case["M"], T.GAT_LAYERS[-1][1] * T.GAT_LAYERS[-1][2])
    assert T.rel(out, gold()["out"]) <= T.TOL


def test_gat_reproduces_the_reference_even_where_it_is_wrong():
    """Local kernel fusion with c > 1: the reference's SpMM lands on top of the gathered SDDMM operand
    (initial_replicate = false, gat.hpp:100 with 15D_dense_shift.hpp:343-349).  A faithful mirror shows the same."""
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(4, lambda w: T.run_gat(w, "15d_fusion2", 2, case))
    out = T.assemble_dense(per_rank, "gat", "subA", case["M"], T.GAT_LAYERS[-1][1] * T.GAT_LAYERS[-1][2])
    assert T.rel(out, gold()["quirk_fusion2_p4_c2"]) <= T.TOL


def test_pipelined_forward_is_the_serial_forward(monkeypatch):
    """forwardPass pipelines the heads of a layer over two product buffers (product of head j + 1 on the auxiliary stream);
    HNH_GAT_SERIAL=1 is the reference's head-after-head order (gat.hpp:106-112).  Same result bit for bit, with layers of
    different widths (the product buffers are re-shaped between layers) and a one-head layer."""
    m, layers = 1 << 9, [(16, 8, 3), (24, 4, 1), (4, 6, 2)]
    rows, cols = H.generate_er(m, m, m * 6, 3)
    x = O.dense_fill(m, 16, 2) * 4.0
    case = dict(name="gatpipe", M=m, N=m, R=16, rows=rows, cols=cols, vals=np.ones(len(rows)), A=x / T.GAT_INPUT_SCALE, B=x / T.GAT_INPUT_SCALE)

    def forward(p, c, alg):
        per_rank = H.run_spmd(p, lambda w: T.run_gat(w, alg, c, case, layers=layers))
        return T.assemble_dense(per_rank, "gat", "subA", m, layers[-1][1] * layers[-1][2])

    for alg, p, c in [("15d_fusion2", 1, 1), ("15d_fusion1", 2, 2)]:
        monkeypatch.setenv("HNH_GAT_SERIAL", "1")
        serial = forward(p, c, alg)
        monkeypatch.delenv("HNH_GAT_SERIAL")
        assert np.count_nonzero(serial) > serial.size // 10
        assert np.array_equal(forward(p, c, alg), serial)
        assert T.rel(serial, O.gat_forward(rows, cols, m, x, layers, T.GAT_ALPHA)) <= T.TOL
