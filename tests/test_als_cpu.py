"""ALS-CG application (BASELINE config 5; als_conjugate_gradients.{h,cpp}) on CPU ranks: the C++ Distributed_ALS
mirror over every schedule, compared with golden embeddings produced by the REFERENCE's own ALS code
(tests/golden/make_golden_als.py).  Kernels are served by the oracle test double (host logic under test)."""
import pytest

import hnh_testlib as T
from distributed_sddmm_amd import api as H


@pytest.fixture(autouse=True, scope="module")
def cpu_test_double():
    H.load_backend(T.ORACLE_BACKEND)
    yield


@pytest.mark.parametrize("alg,p,c", [("15d_fusion1", 1, 1), ("15d_fusion2", 1, 1), ("15d_fusion2", 4, 2), ("15d_fusion2", 4, 1), ("15d_fusion2", 8, 1), ("15d_fusion1", 4, 1), ("15d_sparse", 4, 1),
                                     ("15d_sparse", 4, 2), ("25d_dense_replicate", 4, 1), ("25d_dense_replicate", 8, 2), ("25d_sparse_replicate", 8, 2),
                                     # grids with remainders: odd counts, three layers, 2 x 2 x 4
                                     ("15d_fusion2", 5, 1), ("15d_fusion2", 6, 2), ("15d_fusion1", 9, 3), ("15d_sparse", 12, 3), ("25d_dense_replicate", 16, 4),
                                     ("25d_sparse_replicate", 16, 4)])
@pytest.mark.parametrize("case_name", ["er8_r16", "ragged_r8"])
def test_als_matches_reference(case_name, alg, p, c):
    case = T.case_inputs(case_name)
    if not T.valid_config(alg, p, c, case["R"]):
        pytest.skip("R not divisible for this grid")
    per_rank = H.run_spmd(p, lambda w: T.run_als(w, alg, c, case, 1, 5))
    T.check_als_against_golden(per_rank, case)


@pytest.mark.parametrize("alg,p,c", [("15d_fusion2", 1, 1), ("15d_fusion2", 4, 2), ("15d_fusion1", 4, 1), ("25d_dense_replicate", 4, 1)])
def test_als_with_separate_cg_updates_matches_reference(monkeypatch, alg, p, c):
    """HNH_ALS_UNFOLDED=1: schedules without an R split normally run a whole CG iteration inside the fused call (hnh_cg_update);
    the variant with the reference's separate update steps (what R-split schedules always use) must give the same embeddings."""
    monkeypatch.setenv("HNH_ALS_UNFOLDED", "1")
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(p, lambda w: T.run_als(w, alg, c, case, 1, 5))
    T.check_als_against_golden(per_rank, case)


def test_run_cg_with_artificial_ground_truth_is_distribution_independent():
    """run_cg(1) with the built-in hashed initialisation: same residual on 1 rank and on a 2 x 2 grid."""
    case = T.case_inputs("er8_r16")

    def body(alg, c):
        def f(w):
            sp = H.SpmatLocal.from_global(w, case["M"], case["N"], case["rows"], case["cols"], case["vals"])
            d = H.DistributedSparse(w, alg, sp, case["R"], c)
            als = H.DistributedALS(d, True, seed=7)
            als.initializeEmbeddings()
            r0 = als.computeResidual()
            als.cg_optimizer(H.AMAT, 3)
            als.cg_optimizer(H.BMAT, 3)
            r1 = als.computeResidual()
            als.free(); d.free(); sp.free()
            return r0, r1
        return f

    a = H.run_spmd(1, body("15d_fusion2", 1))[0]
    b = H.run_spmd(4, body("25d_dense_replicate", 1))[0]
    assert a[1] < a[0]
    assert abs(a[0] - b[0]) <= 1e-9 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-7 * abs(a[0])
