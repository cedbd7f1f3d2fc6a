"""GPU parity of the full operator path: C++ schedules -> StandardKernel -> hand-written HIP kernels, for all
five schedules and every (p, c) up to 8 logical ranks on the ONE GPU of the test box (loopback transport:
one host thread per rank, ring transfers as device-to-device copies on the communication stream, so the
double-buffering / event ordering of the overlapped rings is exercised for real).  Compared element-wise
with the golden vectors produced by the reference (tests/golden), tolerance 1e-11 relative."""
import numpy as np
import pytest

import hnh_testlib as T
from distributed_sddmm_amd import api as H

pytestmark = pytest.mark.gpu
GRIDS = [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2), (8, 4), (8, 8)]


@pytest.fixture(autouse=True, scope="module")
def hip_backend():
    assert H.load_backend(None) == "hip-gfx950"  # fails loudly if the HIP library is missing
    yield


def configs(case_name):
    meta = T.golden_cases()[case_name]
    return [(alg, p, c) for alg in H.ALGORITHMS for (p, c) in GRIDS if T.valid_config(alg, p, c, meta["R"])]


@pytest.mark.parametrize("alg,p,c", configs("er8_r16"))
def test_er8_all_schedules_hip(alg, p, c):
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
    assert per_rank[0]["alg_info"]["backend"] == "hip-gfx950"
    T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)


@pytest.mark.parametrize("case_name", ["ragged_r8", "rect_r16", "tiny_r8"])
@pytest.mark.parametrize("alg", H.ALGORITHMS)
def test_edge_cases_hip(case_name, alg):
    case = T.case_inputs(case_name)
    for p, c in [(1, 1), (4, 1), (8, 2)]:
        if not T.valid_config(alg, p, c, case["R"]):
            continue
        per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
        T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)


@pytest.mark.parametrize("alg", H.ALGORITHMS)
def test_matrix_without_nonzeros_hip(alg):
    """A sparse matrix with NO nonzero on the HIP path: every operator leaves zeros behind, on one and on several ranks."""
    m, n, r = 40, 24, 8
    case = T.make_case("empty", m, n, r, np.array([], dtype=np.int64), np.array([], dtype=np.int64), seed=3)
    for p, c in ((1, 1), (4, 1), (8, 2)):
        if not T.valid_config(alg, p, c, r):
            continue
        per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
        T.check_against_oracle(T.assemble(per_rank, case), case, alg)


@pytest.mark.parametrize("alg,p,c", [("15d_fusion2", 1, 1), ("15d_fusion2", 4, 2), ("15d_fusion1", 4, 1), ("15d_sparse", 4, 1),
                                     ("25d_dense_replicate", 4, 1), ("25d_sparse_replicate", 8, 2)])
def test_cfg1_scale_vs_oracle(alg, p, c):
    """BASELINE config 1 shape (ER 2^16, ~1e6 nnz, R = 16) — larger than the golden fixtures: compared with the
    numpy oracle (itself pinned to the reference by tests/test_oracle_golden.py)."""
    from oracle import oracle as O
    log_m, ef, r = 16, 16, 16
    m = 1 << log_m
    rows, cols = H.generate_er(m, m, m * ef, 12345)
    r2, c2 = O.erdos_renyi(log_m, ef)
    assert np.array_equal(rows, r2) and np.array_equal(cols, c2), "native generator must equal the oracle's"
    vals = O.sparse_values(rows, cols, m, 5)
    case = dict(name="cfg1", M=m, N=m, R=r, rows=rows, cols=cols, vals=vals, A=O.dense_fill(m, r, 6), B=O.dense_fill(m, r, 7))
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
    glob = T.assemble(per_rank, case)
    keys = rows * m + cols
    ign = alg == "15d_fusion2"
    assert np.array_equal(glob["sddmmA"][0], keys)
    assert T.rel(glob["sddmmA"][1], O.sddmm(rows, cols, vals, case["A"], case["B"])) <= T.TOL
    assert T.rel(glob["spmmA"], O.spmm_a(rows, cols, vals, case["B"], m)) <= T.TOL
    assert T.rel(glob["spmmB"], O.spmm_b(rows, cols, vals, case["A"], m)) <= T.TOL
    assert T.rel(glob["fusedA"], O.fused_a(rows, cols, vals, case["A"], case["B"], ign)[0]) <= T.TOL
    assert T.rel(glob["fusedB"], O.fused_b(rows, cols, vals, case["A"], case["B"], ign)[0]) <= T.TOL


def test_rccl_world_single_rank():
    """The RCCL transport with one rank (all a 1-GPU box allows): communicator init from a unique id, sub-communicator
    split, self send/recv, all-gather / reduce-scatter of size 1, host collectives staged through the device."""
    case = T.case_inputs("er8_r16")
    w = H.World.rccl(0, 1, 0, H.rccl_unique_id())
    try:
        for alg in H.ALGORITHMS:
            out = T.run_all_ops(w, alg, 1, case)
            assert out["alg_info"]["transport"] == "rccl"
            T.check_against_golden(T.assemble([out], case), [out], case, alg)
        vals, ok = w.grid_probe(1, 1, 1, 3)
        assert ok and vals[:3] == [0, 0, 0]
        # the transport self-tests bench.py --gpus N runs first (here with the one rank a 1-GPU box allows)
        assert all(w.preflight(k, 4096) == 0.0 for k in range(len(H.World.PREFLIGHT)))
    finally:
        w.close()


@pytest.mark.parametrize("mode", ["relay", "mesh"])
@pytest.mark.parametrize("alg,p,c", [("15d_fusion2", 4, 1), ("15d_fusion2", 8, 1), ("15d_fusion2", 8, 2), ("15d_fusion1", 8, 1), ("15d_fusion1", 4, 1)])
def test_relay_ring_and_mesh_fetch_agree_with_the_reference(monkeypatch, mode, alg, p, c):
    """HNH_RING_MODE: neighbour relay ring vs owner->consumer mesh fetch of the read-only moving operand."""
    monkeypatch.setenv("HNH_RING_MODE", mode)
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
    T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)


@pytest.mark.parametrize("alg,p,c", [("15d_fusion2", 1, 1), ("15d_fusion2", 4, 2), ("15d_fusion2", 4, 1), ("15d_fusion2", 8, 1), ("15d_fusion1", 4, 1),
                                     ("15d_sparse", 4, 1), ("25d_dense_replicate", 8, 2), ("25d_sparse_replicate", 8, 2)])
def test_als_cg_matches_reference_hip(alg, p, c):
    """BASELINE config 5 (ALS-CG iteration around fusedSpMM) on the GPU vs the reference's own ALS code."""
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(p, lambda w: T.run_als(w, alg, c, case, 1, 5))
    T.check_als_against_golden(per_rank, case)


@pytest.mark.parametrize("alg,p,c", [("25d_dense_replicate", 8, 2), ("15d_fusion2", 4, 1), ("15d_sparse", 4, 1)])
def test_config4_shape_skewed_graph_hip(alg, p, c):
    """Shape of BASELINE config 4: skewed (R-MAT) graph, R = 256, 2.5D dense-replicate on 8 ranks with c = 2."""
    rows, cols = H.generate_rmat(13, 8192 * 16)
    case = T.make_case("rmat13", 8192, 8192, 256, rows, cols)
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
    T.check_against_oracle(T.assemble(per_rank, case), case, alg)


def test_matrix_market_file_to_25d_dense_on_the_gpu(tmp_path):
    """SURVEY 8(f3), the input side end to end on the HIP path: a symmetric .mtx with duplicate entries is parsed, merged
    (maximum, on the GPU next to the sort), vertex-permuted (random_permute.cpp:42-50) and fed to the 2.5D dense-replicating
    schedule at R = 256 on p = 8, c = 2; every operator result against the oracle on the matrix the file describes."""
    from oracle import oracle as O
    n, r, seed = 600, 256, 9
    path = str(tmp_path / "graph.mtx")
    rows, cols, vals = T.write_symmetric_mtx_with_duplicates(path, n, 4)
    label = O.vertex_permutation(n, seed)
    prow, pcol = label[rows], label[cols]
    order = np.argsort(prow * n + pcol)
    case = dict(name="mtx", M=n, N=n, R=r, rows=prow[order], cols=pcol[order], vals=vals[order], A=O.dense_fill(n, r, 31), B=O.dense_fill(n, r, 32))

    def from_file(w):
        sp = H.SpmatLocal.load_tuples(w, True, -1, -1, path)
        assert sp.info()["dist_nnz"] == len(rows)
        sp.permute(seed)
        return sp

    for alg, p, c in (("25d_dense_replicate", 8, 2), ("15d_fusion2", 4, 1)):
        per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case, make_spmat=from_file))
        T.check_against_oracle(T.assemble(per_rank, case), case, alg)


def test_cpp_dropin_driver(tmp_path):
    """examples/bench_er.cpp — the reference's bench_erdos_renyi.cpp + benchmark_dist.cpp re-written against our class
    headers — runs end to end and appends JSON records with the reference's keys (benchmark_dist.cpp:151-162)."""
    import json
    import os
    import subprocess
    # the class layouts live in the headers: make sure the drivers match the library they are about to load
    subprocess.run(["make", "-C", os.path.join(T.ROOT, "examples"), "bench_er", "bench_file", "bench_heatmap"], check=True, capture_output=True, timeout=600)
    exe = os.path.join(T.ROOT, "examples", "bench_er")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    out = tmp_path / "results.json"
    for alg, fused in (("15d", "fused"), ("25d", "fused"), ("15d_sparse", "unfused")):
        r = subprocess.run([exe, "12", "8", alg, "32", "1", str(out), fused], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    recs = json.loads("[" + out.read_text().rstrip().rstrip(",") + "]")
    assert [r["alg_name"] for r in recs] == ["15d_fusion1", "15d_fusion2", "25d_sparse_replicate", "25d_dense_replicate", "15d_sparse"]
    for r in recs:
        for key in ("elapsed", "overall_throughput", "fused", "num_trials", "alg_name", "alg_info", "application_communication_time", "perf_stats"):
            assert key in r
        assert r["num_trials"] == 5 and r["alg_info"]["backend"] == "hip-gfx950" and "Computation Time" in r["perf_stats"]
    r = subprocess.run([exe, "10", "8", "15d_fusion2", "16", "1", str(out), "fused", "als"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    # bench_file.cpp's driver: same records from a MatrixMarket file ("15d" = 15d_sparse unfused, "25d" = 25d_dense_replicate unfused)
    exe_file = os.path.join(T.ROOT, "examples", "bench_file")
    assert os.path.exists(exe_file), "run __graft_entry__.build()"
    mtx = str(tmp_path / "g.mtx")
    mrows, _, _ = T.write_symmetric_mtx_with_duplicates(mtx, 500, 3)
    out2 = tmp_path / "file_results.json"
    for alg in ("15d", "25d", "15d_fusion2"):
        r = subprocess.run([exe_file, mtx, alg, "64", "1", str(out2), "vanilla"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, HNH_PERMUTE_SEED="5"))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        assert "File reader read %d nonzeros." % len(mrows) in r.stdout
    recs2 = json.loads("[" + out2.read_text().rstrip().rstrip(",") + "]")
    assert [r["alg_name"] for r in recs2] == ["15d_sparse", "25d_dense_replicate", "15d_fusion2"]
    assert all(r["alg_info"]["nnz"] == len(mrows) and r["alg_info"]["m"] == 500 and not r["fused"] for r in recs2)
    # bench_heatmap.cpp's driver: the width sweep of one family on one matrix (a short width list here)
    out3 = tmp_path / "heatmap.json"
    for family, algs in (("15d", ["15d_fusion1", "15d_fusion2", "15d_sparse"]), ("25d", ["25d_sparse_replicate", "25d_dense_replicate"])):
        r = subprocess.run([os.path.join(T.ROOT, "examples", "bench_heatmap"), "11", "8", family, "1", str(out3), "64,192"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    recs3 = json.loads("[" + out3.read_text().rstrip().rstrip(",") + "]")
    assert [(r["alg_name"], r["alg_info"]["r"]) for r in recs3] == [(a, w) for fam in (["15d_fusion1", "15d_fusion2", "15d_sparse"], ["25d_sparse_replicate", "25d_dense_replicate"])
                                                                   for w in (64, 192) for a in fam]
    assert all(r["fused"] == (r["alg_name"] != "25d_sparse_replicate") for r in recs3)
    # examples/verify = scratch.cpp:26-76 verify_operation: the fingerprint trio of every algorithm on the file's matrix, against the
    # oracle's numbers (the compiled reference prints the same ones, tests/test_tools_cpu.py)
    from oracle import oracle as O
    subprocess.run(["make", "-C", os.path.join(T.ROOT, "examples"), "verify"], check=True, capture_output=True, timeout=600)
    r = subprocess.run([os.path.join(T.ROOT, "examples", "verify"), mtx, "all", "32", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    got = np.array([float(ln.split(":")[1]) for ln in r.stdout.splitlines() if "Fingerprint:" in ln]).reshape(5, 3)
    mr, mc, _ = T.write_symmetric_mtx_with_duplicates(str(tmp_path / "again.mtx"), 500, 3)
    want = np.array(O.fingerprints(mr, mc, 500, 500, 32))
    assert np.max(np.abs(got - want) / want) <= 1e-11
    # the GAT application of benchmark_dist.cpp:88-94,133-135 (3 layers, 14 heads, 256 features per head)
    for alg in ("15d_fusion2", "15d_fusion1"):
        r = subprocess.run([exe, "10", "8", alg, "256", "1", str(out), "fused", "gat"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    # examples/c_operator.c: the operator C ABI from plain C11 with its own closed-form check in C (exit status 0 = matches to 1e-11)
    subprocess.run(["make", "-C", os.path.join(T.ROOT, "examples"), "c_operator"], check=True, capture_output=True, timeout=600)
    for alg in ("15d_fusion2", "25d_dense_replicate"):
        r = subprocess.run([os.path.join(T.ROOT, "examples", "c_operator"), "14", "16", alg, "128"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and alg + " on hip-gfx950: 16384 x 16384" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_custom_kernel_plugin_hip():
    """examples/custom_kernel.cpp: a user KernelImplementation that implements only the reference's two pure virtuals
    drives all five schedules on the GPU and reproduces StandardKernel's results (the documented extension point)."""
    import os
    import subprocess
    subprocess.run(["make", "-C", os.path.join(T.ROOT, "examples"), "custom_kernel"], check=True, capture_output=True, timeout=600)
    exe = os.path.join(T.ROOT, "examples", "custom_kernel")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    r = subprocess.run([exe, "", "12", "8", "32"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all schedules ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("alg,p,c", [("15d_fusion1", 1, 1), ("15d_fusion1", 4, 2), ("15d_fusion2", 1, 1), ("15d_fusion2", 4, 1)])
def test_gat_forward_matches_reference_hip(alg, p, c):
    """GAT head = MFMA fp64 GEMM + SDDMM + LeakyReLU + SpMM + ReLU on the GPU vs the reference's own gat.hpp."""
    import os
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(p, lambda w: T.run_gat(w, alg, c, case))
    out = T.assemble_dense(per_rank, "gat", "subA", case["M"], T.GAT_LAYERS[-1][1] * T.GAT_LAYERS[-1][2])
    gold = dict(np.load(os.path.join(T.GOLDEN, "gat_er8_r16.npz")))["out"]
    assert T.rel(out, gold) <= T.TOL


@pytest.mark.parametrize("alg,p,c", [("15d_fusion2", 1, 1), ("15d_fusion2", 4, 1), ("15d_fusion1", 4, 2)])
def test_gat_forward_at_benchmark_widths_against_the_compiled_reference(alg, p, c):
    """GAT at the head widths the in-launch paths are built for (128 and 64 features per head, 256-wide layer input: exact-width
    kernel instances, fp64 MFMA GEMM tiles, ReLU delivery into the layer output) on 2^15 vertices, against the reference's own
    gat.hpp run on the host cores (and the numpy restatement of it)."""
    from oracle import oracle as O
    from oracle import refrun as RR
    m, layers, alpha = 1 << 15, [(128, 128, 2), (256, 64, 3)], 0.2
    rows, cols = H.generate_er(m, m, m * 16, 77)
    x = O.dense_fill(m, 128, 41) * 4.0
    case = dict(name="gat15", M=m, N=m, R=128, rows=rows, cols=cols, vals=np.ones(len(rows)), A=x / T.GAT_INPUT_SCALE, B=x / T.GAT_INPUT_SCALE)
    per_rank = H.run_spmd(p, lambda w: T.run_gat(w, alg, c, case, layers=layers, alpha=alpha))
    out = T.assemble_dense(per_rank, "gat", "subA", m, layers[-1][1] * layers[-1][2])
    want = O.gat_forward(rows, cols, m, x, layers, alpha)
    assert np.count_nonzero(want) > want.size // 10 and np.count_nonzero(want == 0.0) > want.size // 10  # both sides of the ReLU are hit
    assert T.rel(out, want) <= T.TOL
    if RR.available():
        ref = RR.gat(m, rows, cols, 128, x, "15d_fusion1", 1, 1, alpha, layers, timeout=600)
        assert T.rel(out, ref) <= T.TOL


@pytest.mark.parametrize("alg,p,c", [("15d_fusion2", 1, 1), ("15d_fusion2", 2, 1), ("15d_fusion1", 2, 2)])
def test_gat_pipelined_forward_is_the_serial_forward(alg, p, c, monkeypatch):
    """forwardPass runs the product X * W of head j + 1 on HNH_STREAM_AUX beside the attention pass of head j (two product
    buffers, events): the same kernels on the same operands as the reference's head-after-head order (gat.hpp:106-112,
    HNH_GAT_SERIAL=1), so the layer outputs must agree BIT FOR BIT, run after run (a missing event shows up as a difference).
    Three layers of different widths, so the product buffers are re-shaped between layers."""
    from oracle import oracle as O
    m, layers, alpha = 1 << 14, [(128, 128, 3), (384, 64, 4), (256, 32, 2)], 0.2
    rows, cols = H.generate_er(m, m, m * 24, 5)
    x = O.dense_fill(m, 128, 9) * 4.0
    case = dict(name="gatpipe", M=m, N=m, R=128, rows=rows, cols=cols, vals=np.ones(len(rows)), A=x / T.GAT_INPUT_SCALE, B=x / T.GAT_INPUT_SCALE)

    def forward():
        per_rank = H.run_spmd(p, lambda w: T.run_gat(w, alg, c, case, layers=layers, alpha=alpha))
        return T.assemble_dense(per_rank, "gat", "subA", m, layers[-1][1] * layers[-1][2])

    monkeypatch.setenv("HNH_GAT_SERIAL", "1")
    serial = forward()
    monkeypatch.delenv("HNH_GAT_SERIAL")
    assert np.count_nonzero(serial) > serial.size // 10
    for _ in range(3):
        assert np.array_equal(forward(), serial)


def test_operands_in_torch_memory():
    """hnh_dense_wrap: operands that live in PyTorch-owned HBM (non-owning views).  The fused schedule hands its
    result back by copying into the caller's tensor instead of swapping storage (common.h:88-92 semantics)."""
    import torch
    from oracle import oracle as O
    case = T.case_inputs("er8_r16")
    m, r = case["M"], 128
    a, b = O.dense_fill(m, r, 21), O.dense_fill(m, r, 22)
    ta = torch.from_numpy(a).to("cuda:0")
    tb = torch.from_numpy(b).to("cuda:0")
    torch.cuda.synchronize()
    w = H.World.single(0)
    sp = H.SpmatLocal.from_global(w, m, m, case["rows"], case["cols"], np.ones(len(case["rows"])))
    for alg in ("15d_fusion2", "15d_fusion1", "25d_dense_replicate"):
        op = H.DistributedSparse(w, alg, sp, r, 1)
        ta.copy_(torch.from_numpy(a)); torch.cuda.synchronize()
        A, B = H.Dense.wrap(w, ta.data_ptr(), m, r), H.Dense.wrap(w, tb.data_ptr(), m, r)
        S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
        op.initial_shift(A, B, H.K_SDDMM_A)
        op.fusedSpMM(A, B, S, buf, H.AMAT)
        op.de_shift(A, B, H.K_SDDMM_A)
        w.sync()
        want, _ = O.fused_a(case["rows"], case["cols"], np.ones(len(case["rows"])), a, b)
        assert T.rel(ta.cpu().numpy(), want) <= T.TOL, alg
        assert np.array_equal(tb.cpu().numpy(), b), "the other operand must come back unchanged"
        for x in (A, B, S, buf):
            x.free()
        op.free()
    sp.free(); w.close()


@pytest.mark.parametrize("mode", ["relay", "mesh"])
@pytest.mark.parametrize("p,c", [(1, 1), (2, 1), (4, 1), (4, 2), (8, 1), (8, 2)])
def test_fused_out_with_extras_hip(monkeypatch, mode, p, c):
    """Distributed_Sparse::fusedSpMM_out on the HIP kernels: LeakyReLU between the halves, `+ x_scale X` and the row-wise
    <X, Out> in the launch that completes the rows (or appended after the reduce-scatter for c > 1)."""
    monkeypatch.setenv("HNH_RING_MODE", mode)
    for name in ("er8_r16", "rect_r16", "tiny_r8"):
        case = T.case_inputs(name)
        for matmode, alpha, xs, dot in [(H.AMAT, 0.2, 0.0, False), (H.BMAT, None, 1e-3, True), (H.AMAT, 0.05, -0.5, True)]:
            per_rank = H.run_spmd(p, lambda w: T.run_fused_out(w, "15d_fusion2", c, case, matmode, alpha, xs, dot))
            assert all(o["supported"] for o in per_rank)
            T.check_fused_out(per_rank, case, matmode, alpha, xs, dot)


@pytest.mark.parametrize("chunks", [1, 3])
def test_column_chunks_hip(monkeypatch, chunks):
    """Column chunks of S under local kernel fusion (Infinity-Cache panels on a ring of one, pipelined fetch on several
    ranks): same results for any chunk count, incl. one that does not divide the block."""
    monkeypatch.setenv("HNH_MESH_CHUNKS", str(chunks))
    for name in ("er8_r16", "ragged_r8"):
        case = T.case_inputs(name)
        for p, c in [(1, 1), (2, 2), (4, 1), (8, 2)]:
            per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, "15d_fusion2", c, case))
            T.check_against_golden(T.assemble(per_rank, case), per_rank, case, "15d_fusion2")


@pytest.mark.parametrize("merge,cap", [("0", None), (None, None), (None, "2")])
def test_window_groupings_of_the_mesh_fetch_hip(monkeypatch, merge, cap):
    """Adaptive chunk windows (Sparse15D_Dense_Shift::walk_merged) on the HIP library, where arrival events really complete late: one
    pass per chunk (HNH_WINDOW_MERGE=0), the default (a pass takes every chunk that has landed when the host decides it) and at most two
    chunks per pass — the reference's golden vectors whatever the grouping, every operation, ALS and the fused pass with its extras."""
    for k, v in (("HNH_WINDOW_MERGE", merge), ("HNH_WINDOW_MERGE_CAP", cap)):
        if v is None:
            monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv(k, v)
    monkeypatch.setenv("HNH_RING_MODE", "mesh")
    monkeypatch.delenv("HNH_MESH_CHUNKS", raising=False)
    for name in ("er8_r16", "ragged_r8"):
        case = T.case_inputs(name)
        for p, c in [(2, 1), (4, 1), (8, 1), (8, 2)]:
            per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, "15d_fusion2", c, case))
            T.check_against_golden(T.assemble(per_rank, case), per_rank, case, "15d_fusion2")
    case = T.case_inputs("er8_r16")
    T.check_als_against_golden(H.run_spmd(4, lambda w: T.run_als(w, "15d_fusion2", 1, case, 1, 5)), case)
    for matmode in (H.AMAT, H.BMAT):
        per_rank = H.run_spmd(4, lambda w: T.run_fused_out(w, "15d_fusion2", 1, case, matmode, 0.3, 0.7, True))
        T.check_fused_out(per_rank, case, matmode, 0.3, 0.7, True)


@pytest.mark.parametrize("alg", ["15d_fusion2", "15d_fusion1"])
def test_fingerprints_at_scale_against_the_compiled_reference(alg):
    """The HIP path against the REFERENCE ITSELF (oracle/_ref/ref_driver = its unmodified sources + MKL) at 8.4e6 nonzeros,
    R = 128: the scratch.cpp fingerprints (squared norms of SDDMM, SpMM-A, SpMM-B under the dummyInitialize fill) — the
    reference's numbers as committed golden values, or from a run on this box's host cores (HNH_LIVE_REFERENCE=1)."""
    import json
    import os
    from oracle import oracle as O
    from oracle import refrun as RR
    logm, ef, r = 18, 32, 128
    m = 1 << logm
    rows, cols = O.erdos_renyi_mn(m, m, m * ef, 12345)
    gold_path = os.path.join(T.GOLDEN, "fullsize_reference.json")
    rec = json.load(open(gold_path)).get("at_scale_fingerprints") if os.path.exists(gold_path) and os.environ.get("HNH_LIVE_REFERENCE") != "1" else None
    if rec is not None and alg in rec:  # the reference's numbers from the build container (tests/golden/make_golden_fullsize.py)
        assert (rec["logm"], rec["edge_factor"], rec["R"], rec["seed"], rec["nnz"]) == (logm, ef, r, 12345, len(rows))
        ref = dict(zip(("sddmm", "spmmA", "spmmB"), rec[alg]))
    else:
        if not RR.available():
            pytest.skip("neither the golden numbers nor the compiled reference are available on this box")
        ref = RR.fingerprints(m, m, rows, cols, r, alg, 1, 1, timeout=900)
    w = H.World.single(0)
    sp = H.SpmatLocal.load_tuples(w, False, logm, ef)   # the same generator, evaluated on the GPU
    assert sp.info()["dist_nnz"] == len(rows)
    d = H.DistributedSparse(w, alg, sp, r, 1)
    A, B = d.like_A_matrix(0.0), d.like_B_matrix(0.0)
    got = []
    for mode in ("sddmm", "spmmA", "spmmB"):
        d.dummyInitialize(A, H.AMAT); d.dummyInitialize(B, H.BMAT)
        if mode == "sddmm":
            ones, res = d.like_S_values(1.0), d.like_S_values(0.0)
            d.sddmmA(A, B, ones, res); x = res.download(); ones.free(); res.free()
        elif mode == "spmmA":
            ones = d.like_S_values(1.0); d.spmmA(A, B, ones); x = A.download(); ones.free()
        else:
            ones = d.like_ST_values(1.0); d.spmmB(A, B, ones); x = B.download(); ones.free()
        got.append(float(np.sum(x.astype(np.float64) ** 2)))
    want = [ref["sddmm"], ref["spmmA"], ref["spmmB"]]
    assert T.rel(np.array(got), np.array(want)) <= T.TOL, (got, want)
    for h in (A, B):
        h.free()
    d.free(); sp.free(); w.close()


@pytest.mark.parametrize("alg,p,c", [("15d_fusion2", 4, 1), ("15d_fusion2", 8, 2), ("15d_sparse", 4, 1), ("25d_dense_replicate", 8, 2)])
def test_schedules_with_compute_units_set_aside_for_communication(alg, p, c, monkeypatch):
    """HNH_COMM_CUS: the compute stream masked off 16 CUs, the communication stream confined to them (hipExtStreamCreateWithCUMask).
    Same results as ever — the event protocol between a masked compute stream and a masked communication stream holds."""
    monkeypatch.setenv("HNH_COMM_CUS", "16")
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
    T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)


@pytest.mark.parametrize("mode", [None, "force", "off"])
@pytest.mark.parametrize("alg,p,c", [("15d_fusion2", 1, 1), ("15d_fusion2", 4, 2), ("15d_fusion1", 4, 1), ("25d_sparse_replicate", 8, 2)])
def test_borrowed_value_arrays_hip(monkeypatch, mode, alg, p, c):
    """Stationary blocks on the HIP kernels: the SpMM reads the caller's SValues slice in place, the SDDMM writes SValues .* dots
    straight into the caller's result (hnh_sddmm_csr_ps).  Golden vectors at R = 16 — by default a slice is lent there only when it
    starts on a 128-byte line (the narrow row kernels), `force` sends unaligned slices through the general loop — and a wider
    operand (R = 64, hub rows included) against the oracle, where every slice is lent by default."""
    if mode is None:
        monkeypatch.delenv("HNH_BORROW", raising=False)
    else:
        monkeypatch.setenv("HNH_BORROW", mode)
    case = T.case_inputs("er8_r16")
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
    T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)
    stats = [sum(r["borrow"][k] for r in per_rank) for k in range(4)]
    if mode == "off":
        assert stats[0] == 0 and stats[2] == 0
    elif mode == "force":
        assert stats[0] > 0 and stats[1] == 0 and (stats[2] > 0) == (alg != "25d_sparse_replicate")
    if not T.valid_config(alg, p, c, 64):
        return
    rng = np.random.default_rng(5)
    m = 640
    lens = rng.integers(0, 30, m)
    lens[7], lens[300] = 600, 450  # hub rows
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    cols = np.concatenate([np.sort(rng.choice(m, int(k), replace=False)) for k in lens]).astype(np.int64)
    wide = T.make_case("wide_r64", m, m, 64, rows, cols, seed=9)
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, wide))
    T.check_against_oracle(T.assemble(per_rank, wide), wide, alg)
    stats = [sum(r["borrow"][k] for r in per_rank) for k in range(4)]
    if mode != "off":
        assert stats[0] > 0 and stats[1] == 0
        if alg != "25d_sparse_replicate":
            assert stats[2] > 0 and stats[3] == 0


@pytest.mark.parametrize("env", [{}, {"HNH_MESH_TAPER": "1,2,2,2,1,1"}, {"HNH_FUSION1_MESH": "0"}])
@pytest.mark.parametrize("p,c", [(4, 1), (8, 2), (8, 1)])
def test_replication_reuse_on_the_mesh_with_hub_rows_hip(monkeypatch, env, p, c):
    """15d_fusion1's row-merged layout on the HIP kernels with HUB rows and hub columns (a skewed graph): the row-range SDDMM passes and the
    staging passes of the mesh reduce-scatter run the long-row path (segments + ordered reduction; stored output rows start from zero)
    inside a row range of the transposed block; checked against the oracle for every operation, and against the ring it replaced."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(17)
    m, r = 2048, 32
    lens = rng.integers(0, 24, m)
    lens[5], lens[1500] = 1900, 1200  # hub rows: every rank's share of them is above the long-row threshold
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    cols = np.concatenate([np.sort(rng.choice(m, int(k), replace=False)) for k in lens]).astype(np.int64)
    # ... and the same hubs as COLUMNS (the B-side operations see the transpose)
    key = np.unique(np.concatenate([rows * m + cols, cols * m + rows]))
    rows, cols = key // m, key % m
    case = T.make_case("hubs_r32", m, m, r, rows, cols, seed=21)
    per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, "15d_fusion1", c, case))
    assert per_rank[0]["alg_info"]["backend"] == "hip-gfx950"
    T.check_against_oracle(T.assemble(per_rank, case), case, "15d_fusion1")
