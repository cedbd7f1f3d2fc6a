"""Parity at the sizes bench.py reports (BASELINE configs 2, 4 and 5 at full size) on the HIP path.

Checkers (test infrastructure only):
  * oracle.fingerprints_closed_form — the scratch.cpp:26-76 fingerprint trio (squared norms of SDDMM / SpMM-A / SpMM-B under
    the dummyInitialize fill) in O(nnz), pinned to the reference's own numbers in tests/test_oracle_golden.py;
  * the compiled reference itself (oracle/_ref/ref_driver: unmodified sources + MKL/MPICH) at config 2's full size — its
    fingerprints and one ALS step at 2^20 vertices — as committed golden numbers (tests/golden/fullsize_reference.json,
    fullsize_cfg5_als.npz, written by tests/golden/make_golden_fullsize.py in the build container: minutes of host time that the
    GPU box no longer spends on every run).  HNH_LIVE_REFERENCE=1, or a missing fixture, runs the reference on this box instead.
Tolerances: 1e-11 relative (fp64, only the summation order differs), 1e-9 for ALS factors (CG amplifies the summation-order
differences; the reference's own five schedules differ by 1.14e-11, ALS_TOL = 10 x that; tests/golden/als_manifest.json)."""
import numpy as np
import pytest

import hnh_testlib as T
from distributed_sddmm_amd import api as H

pytestmark = pytest.mark.gpu
# the reference does not scale beyond ~32 OpenMP/MKL threads on a big host (profiles/archive/r01_cpu_baseline_sweep.log); with all 256
# hardware threads its full-size runs take three times as long
REF_THREADS = min(32, __import__("os").cpu_count() or 1)
LIVE_REFERENCE = __import__("os").environ.get("HNH_LIVE_REFERENCE") == "1"


def reference_record(key):
    """The compiled reference's numbers for `key` from tests/golden/fullsize_reference.json, or None (then the reference runs here)."""
    import json
    import os
    path = os.path.join(T.GOLDEN, "fullsize_reference.json")
    if LIVE_REFERENCE or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get(key)


@pytest.fixture(autouse=True, scope="module")
def hip_backend():
    assert H.load_backend(None) == "hip-gfx950"
    yield


def device_fingerprints(d):
    """scratch.cpp:26-76 on one rank's share: sum of squares of sddmmA / spmmA / spmmB under dummyInitialize."""
    A, B = d.like_A_matrix(0.0), d.like_B_matrix(0.0)
    out = []
    for mode in ("sddmm", "spmmA", "spmmB"):
        d.dummyInitialize(A, H.AMAT)
        d.dummyInitialize(B, H.BMAT)
        kmode = {"sddmm": H.K_SDDMM_A, "spmmA": H.K_SPMM_A, "spmmB": H.K_SPMM_B}[mode]
        d.initial_shift(A, B, kmode)
        if mode == "sddmm":
            ones, res = d.like_S_values(1.0), d.like_S_values(0.0)
            d.sddmmA(A, B, ones, res)
            x = res.download()
            ones.free(); res.free()
        elif mode == "spmmA":
            ones = d.like_S_values(1.0)
            d.spmmA(A, B, ones)
            d.de_shift(A, B, kmode)
            x = A.download()
            ones.free()
        else:
            ones = d.like_ST_values(1.0)
            d.spmmB(A, B, ones)
            d.de_shift(A, B, kmode)
            x = B.download()
            ones.free()
        out.append(float(np.sum(x.astype(np.float64) ** 2)))
    A.free(); B.free()
    return np.array(out)


@pytest.fixture(scope="module")
def config2():
    """BASELINE config 2's matrix: ER 2^20 x 2^20, edge factor 96 (100 658 766 unique nonzeros), from the host generator."""
    from oracle import oracle as O
    logm, ef = 20, 96
    m = 1 << logm
    rows, cols = H.generate_er(m, m, m * ef, 12345)
    assert len(rows) == 100658766
    return dict(logm=logm, ef=ef, m=m, rows=rows, cols=cols, closed=np.array(O.fingerprints_closed_form(rows, cols, m, m, 128)))


def test_config2_closed_form_is_the_reference(config2):
    """The compiled reference at config 2's FULL size (1.0e8 nonzeros, R = 128; golden numbers, or about a minute of host time
    when run live) gives the fingerprints the closed form predicts — so the next tests' checker is the reference's arithmetic
    at this very size."""
    from oracle import refrun as RR
    rec = reference_record("config2_fingerprints")
    if rec is not None:
        assert (rec["logm"], rec["edge_factor"], rec["R"], rec["seed"], rec["nnz"]) == (config2["logm"], config2["ef"], 128, 12345, len(config2["rows"]))
        want = np.array(rec["values"])
    else:
        if not RR.available():
            pytest.skip("neither the golden numbers nor the compiled reference are available on this box")
        ref = RR.fingerprints(config2["m"], config2["m"], config2["rows"], config2["cols"], 128, "15d_fusion2", 1, 1, timeout=1500,
                              threads=REF_THREADS)
        want = np.array([ref["sddmm"], ref["spmmA"], ref["spmmB"]])
    assert T.rel(config2["closed"], want) <= T.TOL, (config2["closed"], want)


@pytest.mark.parametrize("alg", ["15d_fusion2", "15d_sparse"])
def test_config2_full_size_fingerprints(config2, alg):
    """The headline configuration itself (what bench.py times at N = 1) and one unfused schedule on the same matrix:
    SDDMM, SpMM-A and SpMM-B through the whole operator stack on the GPU, int32 row pointers up to 1.0e8, the
    Infinity-Cache column panels, 262 144 workgroups per launch."""
    w = H.World.single(0)
    sp = H.SpmatLocal.load_tuples(w, False, config2["logm"], config2["ef"])  # the device generator, as bench.py uses it
    assert sp.info()["dist_nnz"] == len(config2["rows"])
    d = H.DistributedSparse(w, alg, sp, 128, 1)
    got = device_fingerprints(d)
    assert T.rel(got, config2["closed"]) <= T.TOL, (got, config2["closed"])
    d.free(); sp.free(); w.close()


def test_config2_full_size_fused_elementwise(config2):
    """fusedSpMM at config 2's size against per-row closed forms with NON-constant operands: A[i,:] = a_i, B[j,:] = b_j
    (row-constant), S = 1  =>  sddmm(i,j) = R a_i b_j and  out[i,:] = R a_i sum_{j in row i} b_j^2  for every column."""
    m, r = config2["m"], 128
    rng = np.random.default_rng(7)
    a, b = rng.uniform(0.5, 1.5, m), rng.uniform(0.5, 1.5, m)
    w = H.World.single(0)
    sp = H.SpmatLocal.load_tuples(w, False, config2["logm"], config2["ef"])
    d = H.DistributedSparse(w, "15d_fusion2", sp, r, 1)
    A, B = d.like_A_matrix(0.0), d.like_B_matrix(0.0)
    A.upload(np.repeat(a[:, None], r, axis=1))
    B.upload(np.repeat(b[:, None], r, axis=1))
    S, buf = d.like_S_values(1.0), d.like_S_values(0.0)
    d.fusedSpMM(A, B, S, buf, H.AMAT)
    got = A.download()
    want_row = r * a * np.bincount(config2["rows"], weights=b[config2["cols"]] ** 2, minlength=m)
    assert np.max(np.abs(got - want_row[:, None])) <= T.TOL * np.max(np.abs(want_row))
    for x in (A, B, S, buf):
        x.free()
    d.free(); sp.free(); w.close()


@pytest.mark.parametrize("alg,c,chunks,ring", [("15d_fusion2", 1, "4", "mesh"), ("15d_fusion2", 2, "2", "mesh"), ("15d_fusion2", 1, "1", "relay"),
                                               ("15d_fusion1", 2, "4", "mesh"), ("15d_fusion1", 1, "3", "mesh"), ("15d_sparse", 2, "4", "mesh"),
                                               ("25d_dense_replicate", 2, "4", "mesh"), ("25d_sparse_replicate", 2, "4", "mesh")])
def test_config3_full_size_on_eight_logical_ranks(config2, monkeypatch, alg, c, chunks, ring):
    """BASELINE config 3 — the schedule an 8-GPU bench.py run executes (1.5D dense shift, local kernel fusion, the same 1.0e8-nonzero
    matrix) — at FULL size on 8 logical ranks sharing the GPU: merged layout, chunked fetch into the landing buffer and windowed
    passes (or the relay ring), with and without replication; and the other four schedules on the same matrix and rank count.
    Checked like bench.py checks itself (closed form of one fused call from constant operands, every rank's rows and columns) and
    by the fingerprint trio."""
    monkeypatch.setenv("HNH_MESH_CHUNKS", chunks)
    monkeypatch.setenv("HNH_RING_MODE", ring)
    m, r, p = config2["m"], 128, 8
    deg = np.bincount(config2["rows"], minlength=m).astype(np.float64)

    def body(w):
        sp = H.SpmatLocal.load_tuples(w, False, config2["logm"], config2["ef"])
        d = H.DistributedSparse(w, alg, sp, r, c)
        sp.free()
        A, B = d.like_A_matrix(0.001), d.like_B_matrix(0.001)
        S, buf = d.like_S_values(1.0), d.like_S_values(0.0)
        d.initial_shift(A, B, H.K_SDDMM_A)
        d.fusedSpMM(A, B, S, buf, H.AMAT)
        d.de_shift(A, B, H.K_SDDMM_A)
        got = A.download().reshape(-1)
        worst, off = 0.0, 0
        for top, left, rc, cc in d.submatrices(H.AMAT):
            blk = got[off:off + rc * cc].reshape(rc, cc)
            off += rc * cc
            keep = max(0, min(rc, m - top))
            worst = max(worst, float(np.max(np.abs(blk[:keep] - deg[top:top + keep, None] * (r * 1e-9)))))
        for x in (A, B, S, buf):
            x.free()
        fp = device_fingerprints(d)
        d.free()
        return worst, fp

    res = H.run_spmd(p, body)
    assert max(x[0] for x in res) <= T.TOL * deg.max() * r * 1e-9
    assert T.rel(np.sum([x[1] for x in res], axis=0), config2["closed"]) <= T.TOL


def test_config4_shape_full_size_25d_dense():
    """BASELINE config 4's shape at full size: a skewed R-MAT graph on 2^22 vertices (2.16e8 unique nonzeros, longest row
    2.4e5 — the stand-in for com-Orkut), R = 256, 2.5D dense-replicating Cannon on p = 8, c = 2 (2 x 2 x 2) through the
    loopback transport; fingerprints against the closed form.  Exercises travelling sparse blocks, hub-row splitting and
    the R split at the size the configuration names."""
    from oracle import oracle as O
    logm, r, p, c = 22, 256, 8, 2
    m = 1 << logm
    rows, cols = H.generate_rmat(logm, 230_000_000)
    want = np.array(O.fingerprints_closed_form(rows, cols, m, m, r))

    def body(w):
        sp = H.SpmatLocal.from_global(w, m, m, rows, cols, None)
        d = H.DistributedSparse(w, "25d_dense_replicate", sp, r, c)
        sp.free()
        fp = device_fingerprints(d)
        d.free()
        return fp

    per_rank = H.run_spmd(p, body)
    got = np.sum(per_rank, axis=0)
    assert T.rel(got, want) <= T.TOL, (got, want)


def test_config5_als_step_at_full_size(config2):
    """BASELINE config 5's application at 2^20 vertices, R = 128: one alternating ALS step (both half-steps, 2 CG iterations
    each = 8 fused calls) on the GPU against the reference's own ALS code (als_conjugate_gradients.cpp) run on host cores from
    the same initial factors and ground truth — its residuals, 512 sampled rows and the column sums of both factors as golden
    numbers (tests/golden/fullsize_cfg5_als.npz), or the whole factors when the reference runs here."""
    import os
    from oracle import oracle as O
    from oracle import refrun as RR
    m, r, rows, cols = config2["m"], 128, config2["rows"], config2["cols"]
    vals = O.sparse_values(rows, cols, m, 5)
    a0, b0 = O.dense_fill(m, r, 11), O.dense_fill(m, r, 12)
    rec, gold, ref = reference_record("config5_als"), None, None
    gold_path = os.path.join(T.GOLDEN, "fullsize_cfg5_als.npz")
    if rec is not None and os.path.exists(gold_path):
        assert (rec["logm"], rec["edge_factor"], rec["R"], rec["steps"], rec["cg_iters"], rec["value_seed"], rec["a_seed"], rec["b_seed"]) == (
            config2["logm"], config2["ef"], r, 1, 2, 5, 11, 12)
        gold = dict(np.load(gold_path))
    else:
        if not RR.available():
            pytest.skip("neither the golden numbers nor the compiled reference are available on this box")
        ref = RR.als(m, m, rows, cols, vals, r, a0, b0, "15d_fusion2", 1, 1, steps=1, cg_iters=2, timeout=1800, threads=REF_THREADS)
    w = H.World.single(0)
    sp = H.SpmatLocal.from_global(w, m, m, rows, cols, vals)
    d = H.DistributedSparse(w, "15d_fusion2", sp, r, 1)
    A, B = d.like_A_matrix(0.0), d.like_B_matrix(0.0)
    # ground truth in the operator's value order: the coordinate probe gives key = i * N + j per value slot, and the values
    # are a pure function of the key (oracle.sparse_values), so no lookup table is needed at 1e8 nonzeros
    pa = np.zeros((m, r)); pa[:, 0] = np.arange(m); pa[:, 1] = 1.0
    pb = np.zeros((m, r)); pb[:, 0] = m; pb[:, 1] = np.arange(m)
    gts = []
    for sddmm, like in ((d.sddmmA, d.like_S_values), (d.sddmmB, d.like_ST_values)):
        A.upload(pa); B.upload(pb)
        ones, res = like(1.0), like(0.0)
        sddmm(A, B, ones, res)
        keys = np.rint(res.download()).astype(np.uint64)
        res.upload(O.hashed_uniform(keys, 5))
        gts.append(res); ones.free()
    del pa, pb
    als = H.DistributedALS(d, False)
    als.set_ground_truth(gts[0], gts[1])
    A.upload(a0); B.upload(b0)
    als.set_embeddings(A, B)
    residuals = [als.computeResidual()]
    als.cg_optimizer(H.AMAT, 2)
    als.cg_optimizer(H.BMAT, 2)
    residuals.append(als.computeResidual())
    als.get_embeddings(A, B)
    ga, gb = A.download(), B.download()
    if gold is not None:
        # the reference's factors on 512 evenly spaced rows element by element, every row through the column sums (an error of
        # tolerance x max|A| in each element of a column is what the sums allow: the element-wise criterion, aggregated)
        idx = gold["rows"]
        T.record_observed("als_fullsize", case="config 2 size, 15d_fusion2, 2 + 2 CG iterations",
                          **{name: float(np.max(np.abs(got[idx] - gold[name])) / float(gold["absmax"][k])) for got, name, k in ((ga, "A", 0), (gb, "B", 1))},
                          **{"colsum_" + name: float(np.max(np.abs(got.sum(axis=0) - gold["colsum_" + name])) / (float(gold["absmax"][k]) * m))
                             for got, name, k in ((ga, "A", 0), (gb, "B", 1))},
                          residuals=T.rel(np.array(residuals), gold["residuals"]))
        for got, name, k in ((ga, "A", 0), (gb, "B", 1)):
            top = float(gold["absmax"][k])
            assert np.max(np.abs(got[idx] - gold[name])) <= T.ALS_TOL * top, (name, np.max(np.abs(got[idx] - gold[name])) / top)
            assert np.max(np.abs(got.sum(axis=0) - gold["colsum_" + name])) <= T.ALS_TOL * top * m, name
            assert abs(np.abs(got).max() - top) <= T.ALS_TOL * top
        assert T.rel(np.array(residuals), gold["residuals"]) <= T.ALS_TOL
    else:
        assert T.rel(ga, ref["A"]) <= T.ALS_TOL, T.rel(ga, ref["A"])
        assert T.rel(gb, ref["B"]) <= T.ALS_TOL, T.rel(gb, ref["B"])
        assert T.rel(np.array(residuals), ref["residuals"]) <= T.ALS_TOL
    als.free()
    for x in (A, B, gts[0], gts[1]):
        x.free()
    d.free(); sp.free(); w.close()


def test_input_side_at_size_matrix_market_to_25d_dense(tmp_path):
    """The input side (row f3) at the size of a real graph file: an R-MAT graph on 2^20 vertices, edge factor 8, written by the
    library's writer as a SYMMETRIC MatrixMarket file (one triangle stored) in which every third entry appears twice and every seventh
    three times (1.2e7 lines, 218 MB), then
      (a) read by 8 logical ranks through SpmatLocal::loadTuples(readFromFile) — the parallel mmap parser, both triangles, duplicates
          merged by the device `maximum` pass (SpmatLocal.hpp:485-498) — into 2.5D dense-replicate p = 8, c = 2 at R = 256
          (bench_file.cpp:23-103 with config 4's schedule): the tuple count equals the generator's, one fusedSpMM from keyed operands
          equals the closed form summed over the generator's nonzeros on every rank's rows;
      (b) timed by `bench.py --workload mtx:<file>`: its own host-side parse (pandas, no code shared with the library's parser) finds the
          same nonzeros, the result check passes and the line reports the parse + set-up seconds.
    /tmp holds the file (tmp_path may be a small tmpfs)."""
    import json
    import os
    import subprocess
    import sys
    import time
    logm, ef, r, p, c = 20, 8, 256, 8, 2
    m = 1 << logm
    gr, gc = H.generate_rmat(logm, m * ef)
    lo = np.unique(np.maximum(gr, gc) * m + np.minimum(gr, gc))  # the distinct lower-triangle coordinates of the symmetrised graph
    dup = np.concatenate([lo, lo[::3], lo[::7]])
    np.random.default_rng(5).shuffle(dup)
    vals = ((dup % 13) - 6.0) * 0.25  # duplicates of a coordinate carry the same value here; the count is what the merge is held to
    path = "/tmp/hnh_rmat20_sym_dup.mtx"
    t0 = time.perf_counter()
    H.write_matrix_market(path, m, m, dup // m, dup % m, vals, symmetric=True)
    write_s = time.perf_counter() - t0
    try:
        size_mb = os.path.getsize(path) / 1e6
        i, j = lo // m, lo % m
        off = i != j
        rows, cols = np.concatenate([i, j[off]]), np.concatenate([j, i[off]])  # what a reader must deliver
        nnz = len(rows)
        assert len(dup) > 1.1e7 and size_mb > 150.0 and nnz > 1.5e7, (len(dup), size_mb, nnz)
        from benchlib.common import keyed
        a_key, b_key, u_key, v_key = keyed(np.arange(m), 1), keyed(np.arange(m), 2), keyed(np.arange(r), 3), keyed(np.arange(r), 4)
        want_row = float(np.dot(u_key, v_key)) * a_key * np.bincount(rows, weights=b_key[cols] ** 2, minlength=m)

        def body(w):
            t0 = time.perf_counter()
            sp = H.SpmatLocal.load_tuples(w, True, 0, 0, path)
            w.sync()
            parse_s = time.perf_counter() - t0
            info = sp.info()
            t0 = time.perf_counter()
            op = H.DistributedSparse(w, "25d_dense_replicate", sp, r, c)
            w.sync()
            setup_s = time.perf_counter() - t0
            sp.free()

            def keyed_local(mat_mode, row_key, col_key):
                parts = []
                for top, left, rc, cc in op.submatrices(mat_mode):
                    blk = np.zeros((rc, cc))
                    keep = int(max(0, min(rc, m - top)))
                    blk[:keep] = row_key[top:top + keep, None] * col_key[None, left:left + cc]
                    parts.append(blk.reshape(-1))
                return np.concatenate(parts)
            A, B = op.like_A_matrix(0.0), op.like_B_matrix(0.0)
            S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
            A.upload(keyed_local(H.AMAT, a_key, u_key).reshape(A.shape))
            B.upload(keyed_local(H.BMAT, b_key, v_key).reshape(B.shape))
            op.initial_shift(A, B, H.K_SDDMM_A)
            op.fusedSpMM(A, B, S, buf, H.AMAT)
            op.de_shift(A, B, H.K_SDDMM_A)
            w.sync()
            got, worst, off_, checked = A.download().reshape(-1), 0.0, 0, 0
            for top, left, rc, cc in op.submatrices(H.AMAT):
                keep = int(max(0, min(rc, m - top)))
                blk = got[off_:off_ + rc * cc].reshape(rc, cc)[:keep]
                off_ += rc * cc
                if keep:
                    worst = max(worst, float(np.max(np.abs(blk - want_row[top:top + keep, None] * v_key[None, left:left + cc]))))
                    checked += keep * cc
            for x in (A, B, S, buf):
                x.free()
            op.free()
            return info["dist_nnz"], info["M"], parse_s, setup_s, worst, checked
        res = H.run_spmd(p, body)
        assert all(x[0] == nnz and x[1] == m for x in res), (res[0][:2], nnz)
        assert sum(x[5] for x in res) == m * r  # every element of the output was compared by the rank that owns it
        assert max(x[4] for x in res) <= T.TOL * float(want_row.max() * v_key.max())
        parse_s, setup_s = max(x[2] for x in res), max(x[3] for x in res)
        print("\n[input side at size] %.0f MB, %d lines -> %d tuples on %d logical ranks: written in %.2f s, parsed + merged in %.2f s, "
              "2.5D dense-replicate operator (R = %d) set up in %.2f s" % (size_mb, len(dup), nnz, p, write_s, parse_s, r, setup_s))
        assert parse_s < 60.0 and setup_s < 60.0
        # (b) the benchmark driver on the same file
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--workload", "mtx:" + path, "--alg", "25d_dense_replicate", "--rvalue", str(r),
                              "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-live-traffic"], env=env, capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["check"]["ok"] and line["check"]["nnz_host_generator"] == nnz == line["config"]["nnz"] and line["data"] == "file"
        assert line["config"]["setup_s"] > 0 and line["config"]["parse_s"] > 0 and line["config"]["algorithm"] == "25d_dense_replicate"
        print("[input side at size] bench.py --workload mtx: parse %.2f s, set-up %.2f s, %.2f ms per fused call" % (
            line["config"]["parse_s"], line["config"]["setup_s"], line["ms_per_step"]))
    finally:
        if os.path.exists(path):
            os.remove(path)
