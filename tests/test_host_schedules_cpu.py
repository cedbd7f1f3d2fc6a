"""Host logic on CPU: every schedule class (1.5D dense shift approach 1/2, 1.5D sparse shift, 2.5D Cannon
dense / sparse), every (p, c) up to 8 ranks and a selection up to 18 (odd counts, 3 x 3 / 4 x 4 faces), through the in-process loopback transport — with the kernel
ABI served by the oracle's C test double (explicitly loaded here; the product never does that).  Results
are compared element-wise, by global coordinate, with the golden vectors produced by the reference.

What this covers without a GPU: owner functions and redistribution (a6), block layout and CSR build (a4),
ring schedules incl. event bookkeeping order (a10-a13), sparse/dense shifts (a5, a8), BufferPair hand-back
(a7), grid rank maps (a14), value ownership/length quirks (Appendix C #5), fingerprints (scratch.cpp)."""
import numpy as np
import pytest

import hnh_testlib as T
from distributed_sddmm_amd import api as H

GRIDS = [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2), (8, 4), (8, 8)]


@pytest.fixture(autouse=True, scope="module")
def cpu_test_double():
    assert H.load_backend(T.ORACLE_BACKEND) == "oracle-cpu-test-double"
    yield


def configs(case_name):
    meta = T.golden_cases()[case_name]
    return [(alg, p, c) for alg in H.ALGORITHMS for (p, c) in GRIDS if T.valid_config(alg, p, c, meta["R"])]


@pytest.mark.parametrize("alg,p,c", configs("er8_r16"))
def test_er8_all_schedules(alg, p, c):
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
    T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)


@pytest.mark.parametrize("case_name", ["ragged_r8", "rect_r16", "tiny_r8"])
@pytest.mark.parametrize("alg", H.ALGORITHMS)
def test_edge_cases(case_name, alg):
    """M not divisible by p (padded blocks), non-square S, almost-empty S (null blocks)."""
    case = T.case_inputs(case_name)
    for p, c in [(1, 1), (4, 1), (4, 2), (8, 2)]:
        if not T.valid_config(alg, p, c, case["R"]):
            continue
        per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
        T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)


@pytest.mark.parametrize("case_name", ["er8_r16", "ragged_r8", "rect_r16"])
def test_grids_beyond_eight_and_not_powers_of_two(case_name):
    """p = 3 ... 18, odd counts, 3 x 3 and 4 x 4 faces, three layers: the owner functions, ring lengths and R splits of every schedule
    where the reference's arithmetic has remainders (the reference's outputs do not depend on the grid, so the golden vectors apply)."""
    case = T.case_inputs(case_name)
    ran = 0
    for alg in H.ALGORITHMS:
        for p, c in [(3, 1), (3, 3), (5, 1), (6, 2), (6, 3), (9, 1), (9, 3), (12, 2), (12, 3), (16, 1), (16, 4), (18, 2)]:
            if not T.valid_config(alg, p, c, case["R"]):
                continue
            per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
            T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)
            ran += 1
    assert ran >= 20


def test_value_vector_lengths_follow_the_reference():
    """like_S_values is the ST length under approach 1 and 2.5D dense; json info keys."""
    case = T.case_inputs("rect_r16")

    def body(w):
        sp = H.SpmatLocal.from_global(w, case["M"], case["N"], case["rows"], case["cols"], case["vals"])
        out = {}
        for alg in ("15d_fusion1", "15d_fusion2"):
            d = H.DistributedSparse(w, alg, sp, case["R"], 1)
            out[alg] = (d.info()["nS"], d.info()["nST"], d.json_algorithm_info())
            d.free()
        sp.free()
        return out

    res = H.run_spmd(2, body)
    total = len(case["rows"])
    for alg in ("15d_fusion1", "15d_fusion2"):
        assert sum(r[alg][0] for r in res) == total and sum(r[alg][1] for r in res) == total
        info = res[0][alg][2]
        for key in ("alg_name", "m", "n", "nnz", "r", "adjacency_mode", "p", "c", "dim_interpretations", "dim_values",
                    "nnz_procs", "nnz_tpose_procs"):
            assert key in info
        assert info["nnz"] == total and info["p"] == 2 and sum(info["nnz_procs"]) == total
    # S is split by rows, ST by columns: on a non-square matrix the per-rank counts differ and approach 1 swaps them
    assert [r["15d_fusion1"][0] for r in res] == [r["15d_fusion2"][1] for r in res]


def test_configuration_errors_are_reported():
    case = T.case_inputs("tiny_r8")

    def body(w):
        sp = H.SpmatLocal.from_global(w, case["M"], case["N"], case["rows"], case["cols"], case["vals"])
        errs = []
        for alg, r, c in (("15d_fusion2", 8, 3), ("25d_dense_replicate", 8, 1), ("15d_sparse", 7, 1), ("nope", 8, 1)):
            with pytest.raises(H.HnhError) as e:
                H.DistributedSparse(w, alg, sp, r, c)
            errs.append(str(e.value))
        sp.free()
        return errs

    errs = H.run_spmd(2, body)[0]
    assert "must have c divide num_procs" in errs[0]        # 15D_dense_shift.hpp:60-65
    assert "perfect square" in errs[1]                      # 25D_cannon_dense.hpp:61-67
    assert "divisible by p / c" in errs[2]                  # 15D_sparse_shift.hpp:147-149
    assert "unknown algorithm" in errs[3]


@pytest.mark.parametrize("env", [{"HNH_ACC_HALVES": "0"}, {"HNH_SHIP_INDICES": "1"}, {"HNH_ACC_HALVES": "0", "HNH_SHIP_INDICES": "1"}])
@pytest.mark.parametrize("alg,p,c", [("15d_fusion1", 4, 1), ("15d_fusion1", 8, 2), ("25d_dense_replicate", 4, 1), ("25d_dense_replicate", 8, 2),
                                     ("15d_sparse", 4, 1)])
def test_whole_accumulator_shifts_and_travelling_indices_agree_with_the_reference(monkeypatch, env, alg, p, c):
    """The defaults pipeline a moving ACCUMULATOR in two row halves (one half's shift under the other half's kernel) and keep the
    ring's sparsity structure resident; HNH_ACC_HALVES=0 is the reference's kernel -> shift sequence, HNH_SHIP_INDICES=1 its
    payload (indices travel with the block, which also rules the row halves out for travelling blocks) — same results."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    case = T.case_inputs("ragged_r8")
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
    T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)


@pytest.mark.parametrize("env", [{}, {"HNH_MESH_CHUNKS": "1"}, {"HNH_MESH_TAPER": "3,2,1"}, {"HNH_WINDOW_MERGE": "0"}, {"HNH_ORACLE_EVENTS_PENDING": "2"},
                                 {"HNH_WINDOW_MERGE_CAP": "2", "HNH_ORACLE_EVENTS_PENDING": "3"},
                                 {"HNH_FUSION1_MESH": "0"}, {"HNH_FUSION1_MESH": "0", "HNH_ACC_HALVES": "0"}])
@pytest.mark.parametrize("p,c,case_name", [(4, 1, "er8_r16"), (8, 2, "ragged_r8"), (6, 3, "rect_r16"), (5, 1, "tiny_r8"), (2, 1, "ragged_r8")])
def test_replication_reuse_on_the_mesh_and_on_the_ring_agree_with_the_reference(monkeypatch, env, p, c, case_name):
    """15d_fusion1 (15D_dense_shift.hpp:276-384 with invert): round 6's default — row-merged transposed layout, the SDDMM as row-range passes
    over the landed chunks of the mesh fetch, the SpMM as a mesh reduce-scatter (staging passes, one group of n - 1 transfers per chunk,
    the received partial blocks added in ring-step order) — for chunk shapes and pass groupings (one pass per chunk; arrival queries that
    answer "not yet" so that a pass takes fewer chunks), grids with remainders, null blocks; and the block-by-block rings it replaced
    (HNH_FUSION1_MESH=0: accumulator in two halves / whole).  All against the reference's golden vectors."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    case = T.case_inputs(case_name)
    if not T.valid_config("15d_fusion1", p, c, case["R"]):
        pytest.skip("R not divisible for this grid")
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, "15d_fusion1", c, case))
    T.check_against_golden(T.assemble(per_rank, case), per_rank, case, "15d_fusion1")


@pytest.mark.parametrize("mode", ["relay", "mesh"])
@pytest.mark.parametrize("alg,p,c", [("15d_fusion2", 4, 1), ("15d_fusion2", 8, 1), ("15d_fusion2", 8, 2), ("15d_fusion1", 8, 1), ("15d_fusion1", 4, 1)])
def test_relay_ring_and_mesh_fetch_agree_with_the_reference(monkeypatch, mode, alg, p, c):
    """HNH_RING_MODE: neighbour relay ring vs owner->consumer mesh fetch of the read-only moving operand."""
    monkeypatch.setenv("HNH_RING_MODE", mode)
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
    T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)


def test_perf_counter_keys_match_the_reference():
    """Counter names are the result-file "wire format" (15D_*.hpp:70-74, 25D_cannon_dense.hpp:72-78,
    25D_cannon_sparse.hpp:71-76); with timing_sync the counters are filled with device-complete times."""
    case = T.case_inputs("tiny_r8")
    expect = {
        "15d_fusion1": {"Replication Time", "Cyclic Shift Time", "Computation Time"},
        "15d_fusion2": {"Replication Time", "Cyclic Shift Time", "Computation Time"},
        "15d_sparse": {"Replication Time", "Cyclic Shift Time", "Computation Time"},
        "25d_dense_replicate": {"Dense Cyclic Shift Time", "Sparse Cyclic Shift Time", "Dense Fiber Communication Time",
                                "Computation Time", "Setup Shift Time"},
        "25d_sparse_replicate": {"Dense Cyclic Shift Time", "Sparse Fiber Communication Time", "Computation Time", "Setup Shift Time"},
    }

    def body(w):
        w.set_timing_sync(True)
        out = {}
        for alg in H.ALGORITHMS:
            sp = H.SpmatLocal.from_global(w, case["M"], case["N"], case["rows"], case["cols"], case["vals"])
            d = H.DistributedSparse(w, alg, sp, case["R"], 1)
            d.reset_performance_timers()
            A, B, S, buf = d.like_A_matrix(0.001), d.like_B_matrix(0.001), d.like_S_values(1.0), d.like_S_values(0.0)
            d.initial_shift(A, B, H.K_SDDMM_A)
            d.fusedSpMM(A, B, S, buf, H.AMAT)
            out[alg] = d.json_perf_statistics()
            for x in (A, B, S, buf):
                x.free()
            d.free(); sp.free()
        return out

    res = H.run_spmd(4, body)[0]
    for alg, keys in expect.items():
        assert set(res[alg]) == keys, (alg, res[alg])
        assert res[alg]["Computation Time"] > 0


@pytest.mark.parametrize("chunks", [1, 3, 8])
@pytest.mark.parametrize("ring", ["mesh", "relay"])
def test_chunked_mesh_fetch_any_chunk_count(chunks, ring, monkeypatch):
    """Local kernel fusion keeps S in column chunks of each block so that the fetch of the visiting dense blocks can be
    pipelined chunk by chunk (HNH_MESH_CHUNKS = Q symmetric chunks; the default everywhere else in this suite is six chunks of
    heights (1, 2, 2, 2, 1, 1)): same results for 1 (whole
    blocks), a count that does not divide the block (ragged last chunk, empty chunks on tiny blocks) and the maximum."""
    monkeypatch.setenv("HNH_MESH_CHUNKS", str(chunks))
    monkeypatch.setenv("HNH_RING_MODE", ring)
    for name in ("er8_r16", "ragged_r8", "tiny_r8"):
        case = T.case_inputs(name)
        for p, c in [(1, 1), (2, 1), (2, 2), (4, 2), (8, 1)]:
            per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, "15d_fusion2", c, case))
            T.check_against_golden(T.assemble(per_rank, case), per_rank, case, "15d_fusion2")
    case = T.case_inputs("rect_r16")
    for p in (1, 4):  # a ring of one launches once per chunk (cache panels), several ranks pipeline the fetch on them
        for matmode in (H.AMAT, H.BMAT):
            per_rank = H.run_spmd(p, lambda w: T.run_fused_out(w, "15d_fusion2", 1, case, matmode, 0.2, 0.5, True))
            T.check_fused_out(per_rank, case, matmode, 0.2, 0.5, True)


@pytest.mark.parametrize("taper", ["6,5,4,3,2,1", "1,1", "3,4,4,3,2,1,1", "1,64,1", "5"])
def test_chunked_mesh_fetch_any_chunk_heights(taper, monkeypatch):
    """HNH_MESH_TAPER: the chunks of the mesh fetch with arbitrary relative heights (falling, two equal, seven, one dominating,
    a single chunk) — same results on every golden case, including blocks smaller than the number of fine chunks."""
    monkeypatch.setenv("HNH_MESH_TAPER", taper)
    for name in ("er8_r16", "ragged_r8", "tiny_r8", "rect_r16"):
        case = T.case_inputs(name)
        for p, c in [(2, 1), (4, 2), (8, 1)]:
            per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, "15d_fusion2", c, case))
            T.check_against_golden(T.assemble(per_rank, case), per_rank, case, "15d_fusion2")


def test_bad_chunk_heights_are_refused(monkeypatch):
    case = T.case_inputs("tiny_r8")
    for bad in ("0,1", "1,,2", "a", "1,2,x", ",".join(["1"] * 13)):
        monkeypatch.setenv("HNH_MESH_TAPER", bad)
        with pytest.raises(Exception, match="HNH_MESH_TAPER"):
            H.run_spmd(2, lambda w: T.run_all_ops(w, "15d_fusion2", 1, case))


def test_wrong_length_value_vector_is_refused():
    """like_S_values and like_ST_values differ in length per rank; the reference copies from whichever it is given without
    looking (SpmatLocal.hpp:571-579).  Here a too-short vector is a reported error, not an out-of-bounds read."""
    case = T.case_inputs("rect_r16")

    def body(w):
        sp = H.SpmatLocal.from_global(w, case["M"], case["N"], case["rows"], case["cols"], case["vals"])
        d = H.DistributedSparse(w, "15d_fusion2", sp, case["R"], 1)
        A, B = d.like_A_matrix(0.0), d.like_B_matrix(0.0)
        short = H.Vec.create(w, 1)
        msgs = []
        for call in (lambda: d.spmmA(A, B, short), lambda: d.spmmB(A, B, short)):
            with pytest.raises(H.HnhError) as e:
                call()
            msgs.append(str(e.value))
        for h in (A, B, short):
            h.free()
        d.free(); sp.free()
        return msgs

    for msgs in H.run_spmd(1, body):
        assert all("wrong length" in m for m in msgs)


@pytest.mark.parametrize("alg", H.ALGORITHMS)
def test_matrix_without_nonzeros(alg):
    """A sparse matrix with NO nonzero: every operator is a (collective) no-op that leaves zeros behind, on every grid."""
    m, n, r = 40, 24, 8
    case = T.make_case("empty", m, n, r, np.array([], dtype=np.int64), np.array([], dtype=np.int64), seed=3)
    for p, c in ((1, 1), (4, 1), (8, 2), (4, 4)):
        if not T.valid_config(alg, p, c, r):
            continue
        per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
        T.check_against_oracle(T.assemble(per_rank, case), case, alg)


@pytest.mark.parametrize("mode", ["force", "off"])
@pytest.mark.parametrize("alg,p,c", [("15d_fusion1", 4, 1), ("15d_fusion2", 4, 2), ("15d_fusion2", 1, 1), ("25d_sparse_replicate", 8, 2),
                                     ("15d_sparse", 4, 1)])
def test_borrowed_value_arrays(monkeypatch, mode, alg, p, c):
    """Stationary blocks may read SValues in place (no setCSRValues copy, SpmatLocal.hpp:571-579) and take the SDDMM's result as
    SValues .* dots straight from the kernel (no closing Hadamard, 15D_dense_shift.hpp:366).  HNH_BORROW=force lends whatever the
    alignment, off never does: both give the reference's golden vectors, and the block counts say which path ran.  Travelling
    blocks (sparse shift) never borrow."""
    monkeypatch.setenv("HNH_BORROW", mode)
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
    T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)
    lent_spmm, copied, lent_sddmm, hadamard = (sum(r["borrow"][k] for r in per_rank) for k in range(4))
    if mode == "off" or alg == "15d_sparse":
        assert lent_spmm == 0 and lent_sddmm == 0 and (hadamard > 0 or alg == "25d_sparse_replicate")
        assert copied > 0 or alg == "15d_sparse"  # (the sparse shift copies through setCSRValues, which is not counted)
    elif alg == "25d_sparse_replicate":  # its SDDMM visits the block sqrt(p/c) times: only the SpMM side borrows
        assert lent_spmm > 0 and copied == 0 and lent_sddmm == 0
    else:
        assert lent_spmm > 0 and copied == 0 and lent_sddmm > 0 and hadamard == 0


def test_in_place_sddmm_does_not_borrow(monkeypatch):
    """sddmmA(A, B, S, S): result and SValues are the same vector — the kernel's scale operand must not alias its destination, so
    the call takes the block-values + Hadamard route and still returns S .* dots."""
    monkeypatch.setenv("HNH_BORROW", "force")
    case = T.case_inputs("er8_r16")

    def body(w):
        sp = H.SpmatLocal.from_global(w, case["M"], case["N"], case["rows"], case["cols"], case["vals"])
        d = H.DistributedSparse(w, "15d_fusion2", sp, case["R"], 1)
        A, B = d.like_A_matrix(0.0), d.like_B_matrix(0.0)
        subA, subB = d.submatrices(H.AMAT), d.submatrices(H.BMAT)
        A.upload(T.fill_local(subA, A.shape, case["A"])); B.upload(T.fill_local(subB, B.shape, case["B"]))
        S, res = d.like_S_values(0.5), d.like_S_values(0.0)
        d.sddmmA(A, B, S, res)
        before = d.borrow_stats()
        d.sddmmA(A, B, S, S)
        after = d.borrow_stats()
        out = (res.download(), S.download(), before, after)
        for x in (A, B, S, res):
            x.free()
        d.free(); sp.free()
        return out

    for res, inplace, before, after in H.run_spmd(2, body):
        assert np.array_equal(res, inplace)
        assert before[2] > 0 and after[2] == before[2] and after[3] > before[3]


@pytest.mark.parametrize("merge,cap,pending,want_launches", [
    ("0", None, None, 7),   # one pass per chunk whatever has landed (rounds 2-4): own block + six windows
    (None, None, None, 2),  # everything has landed when the host decides (the double completes at once): own block + ONE pass
    (None, "2", None, 4),   # at most two chunks per pass: (1,2) (2,2) (1,1)
    (None, "4", None, 3),
    (None, None, "2", None),  # every second arrival query answers "not yet": some grouping in between
    (None, None, "3", None),
])
def test_adaptive_chunk_windows_of_the_mesh_fetch(monkeypatch, merge, cap, pending, want_launches):
    """1.5D dense shift, mesh fetch, default chunk heights (1,2,2,2,1,1): a windowed pass covers every chunk whose arrival event has
    completed when the host decides it (Sparse15D_Dense_Shift::walk_merged).  Whatever the grouping — none, everything at once, capped,
    or cut short by arrival events that are not yet complete — the results are the reference's golden vectors (the nonzeros of a row
    are taken in the same order), the number of row-kernel launches per fused call says how the chunks were grouped, and the
    stream-order checker sees no race (the pass waits for the arrival event of its LAST chunk)."""
    for k, v in (("HNH_WINDOW_MERGE", merge), ("HNH_WINDOW_MERGE_CAP", cap), ("HNH_ORACLE_EVENTS_PENDING", pending)):
        if v is None:
            monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv(k, v)
    monkeypatch.setenv("HNH_RING_MODE", "mesh")
    monkeypatch.delenv("HNH_MESH_CHUNKS", raising=False)
    monkeypatch.delenv("HNH_MESH_TAPER", raising=False)
    case = T.case_inputs("er8_r16")
    for p, c in ((4, 1), (8, 2)):
        per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, "15d_fusion2", c, case))
        T.check_against_golden(T.assemble(per_rank, case), per_rank, case, "15d_fusion2")

    def launches(w):
        sp = H.SpmatLocal.from_global(w, case["M"], case["N"], case["rows"], case["cols"], None)
        op = H.DistributedSparse(w, "15d_fusion2", sp, case["R"], 1)
        sp.free()
        A, B = op.like_A_matrix(0.5), op.like_B_matrix(0.25)
        S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
        op.kernel_profile(1)
        op.fusedSpMM(A, B, S, buf, H.AMAT)
        w.sync()
        n = op.kernel_profile(0)[1]
        for x in (A, B, S, buf):
            x.free()
        op.free()
        return n
    got = H.run_spmd(4, launches)
    if want_launches is not None:
        assert got == [want_launches] * 4, got
    else:
        # (the "not yet" answers are dealt over the four rank threads' queries: each rank gets some grouping between the extremes)
        assert all(2 <= n <= 7 for n in got) and any(2 < n < 7 for n in got), got
