"""The measurement tools that take `--backend` run end to end on the CPU test double (tiny sizes; the numbers mean nothing):
keeps tools/rank_share_probe.py and tools/overlap_probe.py from rotting between the GPU sessions that need them."""
import os
import subprocess
import sys

import hnh_testlib as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_tool(name, *args):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("HNH_MESH_CHUNKS", "HNH_MESH_TAPER", "HNH_PACE_LINK_GBPS", "HNH_PACE_COPY"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", name), "--backend", T.ORACLE_BACKEND, "--logm", "10", "--ef", "8", "--r", "16",
                          "--p", "4", "--iters", "2", *args], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    return res.stdout


def test_rank_share_probe_runs_on_the_test_double():
    out = run_tool("rank_share_probe.py", "--chunks", "1,4")
    lines = [ln for ln in out.splitlines() if "rank 0 alone" in ln]
    assert len(lines) == 2 and "chunks=1" in lines[0] and "chunks=4" in lines[1]


def test_overlap_probe_runs_on_the_test_double():
    out = run_tool("overlap_probe.py", "--chunks", "2", "--tapers", "3,2,1;1,2,2,2,1,1", "--pace", "60", "--copy-wgs", "0,2")
    assert out.count("unpaced call") == 6  # three shapes, without and with the paced copies
    assert "taper 3,2,1:" in out and "taper 1,2,2,2,1,1:" in out and "Q=2:" in out
    assert out.count("GB/s/link:") == 6  # one rate per shape and variant
    assert out.count("the paced transfers also move their bytes: 2 throttled workgroups per link") == 3


def test_accumulator_overlap_probe_runs_on_the_test_double():
    out = run_tool("overlap_probe_accumulator.py", "--gbps", "0,60")
    assert "15d_fusion1 spmm" in out and "loopback copies only" in out and out.count(" ms") >= 6
    assert "mesh reduce-scatter" in out and "two halves" in out  # the ring (whole / two halves) against the row-merged mesh form


def test_cpp_drivers_run_on_the_test_double(tmp_path):
    """examples/bench_er, bench_file, bench_heatmap (the reference's three `main()`s, bench_erdos_renyi.cpp / bench_file.cpp /
    bench_heatmap.cpp, against this repository's class headers) compile and run end to end on CPU: the drivers load the kernel
    library that sits next to libhnh_host.so, so a directory holding a copy of the host library and the C test double under the
    kernel library's name stands in for a GPU box.  (Their GPU twin is tests/test_schedules_gpu.py::test_cpp_dropin_driver.)"""
    import json
    import shutil
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "bench_er", "bench_file", "bench_heatmap", "verify"], check=True, capture_output=True, timeout=600)
    libdir = tmp_path / "lib"
    libdir.mkdir()
    shutil.copy(os.path.join(ROOT, "distributed_sddmm_amd", "lib", "libhnh_host.so"), libdir / "libhnh_host.so")
    shutil.copy(T.ORACLE_BACKEND, libdir / "libhnh_kernels.so")
    env = dict(os.environ, LD_LIBRARY_PATH=str(libdir), OMP_NUM_THREADS="2", HNH_HOST_SETUP="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)

    def run(exe, *args):
        r = subprocess.run([os.path.join(ROOT, "examples", exe), *args], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        return r.stdout

    def records(path):
        return json.loads("[" + path.read_text().rstrip().rstrip(",") + "]")

    out = tmp_path / "er.json"
    run("bench_er", "9", "8", "15d", "16", "1", str(out), "fused")
    run("bench_er", "9", "8", "15d_sparse", "16", "1", str(out), "unfused")
    run("bench_er", "8", "8", "15d_fusion2", "16", "1", str(out), "fused", "als")
    recs = records(out)
    assert [(r["alg_name"], r["fused"]) for r in recs] == [("15d_fusion1", True), ("15d_fusion2", True), ("15d_sparse", False), ("15d_fusion2", True)]
    assert all(r["alg_info"]["backend"] == "oracle-cpu-test-double" and r["num_trials"] == 5 and "Computation Time" in r["perf_stats"] for r in recs)
    heat = tmp_path / "heat.json"
    run("bench_heatmap", "9", "8", "15d", "1", str(heat), "16,24")
    run("bench_heatmap", "9", "8", "25d", "1", str(heat), "16")
    assert [(r["alg_name"], r["alg_info"]["r"], r["fused"]) for r in records(heat)] == [
        ("15d_fusion1", 16, True), ("15d_fusion2", 16, True), ("15d_sparse", 16, True), ("15d_fusion1", 24, True), ("15d_fusion2", 24, True),
        ("15d_sparse", 24, True), ("25d_sparse_replicate", 16, False), ("25d_dense_replicate", 16, True)]
    mtx = str(tmp_path / "g.mtx")
    mrows, _, _ = T.write_symmetric_mtx_with_duplicates(mtx, 300, 3)
    fout = tmp_path / "file.json"
    assert "File reader read %d nonzeros." % len(mrows) in run("bench_file", mtx, "15d_fusion2", "16", "1", str(fout), "vanilla")
    assert records(fout)[0]["alg_info"]["nnz"] == len(mrows)
    # examples/verify = the reference's own check (scratch.cpp:26-76 verify_operation): three fingerprints that must not depend on
    # the algorithm, and must be the numbers the reference prints for the same matrix
    import numpy as np
    from oracle import oracle as O
    from oracle import refrun as RR
    mrows, mcols, _ = T.write_symmetric_mtx_with_duplicates(mtx, 300, 3)
    want = np.array(O.fingerprints(mrows, mcols, 300, 300, 16))
    text = run("verify", mtx, "all", "16", "1")
    got = np.array([float(ln.split(":")[1]) for ln in text.splitlines() if "Fingerprint:" in ln]).reshape(5, 3)
    assert sorted(ln.split()[1] for ln in text.splitlines() if ln.startswith("==")) == sorted(T.H.ALGORITHMS)
    assert np.max(np.abs(got - want) / want) <= 1e-11
    if RR.available():
        ref = RR.fingerprints(300, 300, mrows, mcols, 16, "15d_sparse", 1, 1)
        assert np.max(np.abs(got - np.array([ref["sddmm"], ref["spmmA"], ref["spmmB"]])) / want) <= 1e-11
    # examples/c_operator.c: the operator C ABI (include/hnh_dist.h) from plain C11 — gcc -std=c11 -pedantic, no C++ on the caller's
    # side — with its own closed-form check of fusedSpMM in C (exit status 0 = matches to 1e-11), every algorithm
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "c_operator"], check=True, capture_output=True, timeout=600)
    for alg in T.H.ALGORITHMS:
        text = run("c_operator", "9", "8", alg, "16")
        assert alg + " on oracle-cpu-test-double: 512 x 512, 4050 nonzeros" in text and "checked on 8192 elements" in text


def test_cpp_verify_across_processes_over_ipc(tmp_path):
    """examples/verify as 2 and 4 PROCESSES over the ipc-pull transport (HNH_TRANSPORT=ipc, the C++ drivers' second way of meeting
    besides RCCL), on the test double: the fingerprints of every algorithm that fits the grid equal the one-process numbers."""
    import shutil
    import time
    import numpy as np
    from oracle import oracle as O
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        import pytest
        pytest.skip("process_vm_readv between own processes is not permitted here")
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "verify"], check=True, capture_output=True, timeout=600)
    libdir = tmp_path / "lib"
    libdir.mkdir()
    shutil.copy(os.path.join(ROOT, "distributed_sddmm_amd", "lib", "libhnh_host.so"), libdir / "libhnh_host.so")
    shutil.copy(T.ORACLE_BACKEND, libdir / "libhnh_kernels.so")
    mtx = str(tmp_path / "g.mtx")
    mrows, mcols, _ = T.write_symmetric_mtx_with_duplicates(mtx, 256, 4)
    want = np.array(O.fingerprints(mrows, mcols, 256, 256, 16))
    for n, c, algs in ((2, 1, ["15d_fusion2", "15d_sparse"]), (4, 2, ["15d_fusion1", "15d_fusion2", "15d_sparse"]), (4, 1, ["25d_dense_replicate", "25d_sparse_replicate"])):
        for alg in algs:
            session = "v%d_%x" % (os.getpid(), time.time_ns())
            procs = []
            for r in range(n):
                env = dict(os.environ, LD_LIBRARY_PATH=str(libdir), OMP_NUM_THREADS="2", HNH_HOST_SETUP="1", RANK=str(r), WORLD_SIZE=str(n),
                           LOCAL_RANK=str(r), HNH_DEVICE="0", HNH_TRANSPORT="ipc", HNH_IPC_SESSION=session, HNH_IPC_WAIT_S="120")
                procs.append(subprocess.Popen([os.path.join(ROOT, "examples", "verify"), mtx, alg, "16", str(c)], env=env, stdout=subprocess.PIPE,
                                              stderr=subprocess.STDOUT, text=True))
            outs = []
            try:
                for p in procs:
                    outs.append(p.communicate(timeout=300)[0])
            finally:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
            assert all(p.returncode == 0 for p in procs), "\n".join(o[-1000:] for o in outs)
            got = np.array([float(ln.split(":")[1]) for ln in outs[0].splitlines() if "Fingerprint:" in ln])
            assert got.shape == (3,) and np.max(np.abs(got - want) / want) <= 1e-11, (alg, n, c, got, want)


def test_matrix_market_writer_parser_and_the_benchmarks_own_reader_agree(tmp_path):
    """hnh_write_matrix_market (formatted by all host cores) -> the library's parallel parser + duplicate merge (8 logical ranks on the
    test double) -> the tuple count of the generator; and benchlib's host-side reader (pandas' C parser, no code shared with the
    library's) finds the same coordinates — general and symmetric files, duplicates, values with 17 digits.  The GPU twin at the size of
    a real graph file is tests/test_fullsize_gpu.py::test_input_side_at_size_matrix_market_to_25d_dense."""
    import numpy as np
    from benchlib.common import read_mtx_coordinates
    H = T.H
    H.load_backend(T.ORACLE_BACKEND)
    m = 1 << 11
    gr, gc = H.generate_rmat(11, m * 6)
    # general file: the generator's coordinates, every fifth twice
    rows, cols = np.concatenate([gr, gr[::5]]), np.concatenate([gc, gc[::5]])
    vals = np.random.default_rng(1).uniform(-1, 1, len(rows))
    general = str(tmp_path / "general.mtx")
    H.write_matrix_market(general, m, m, rows, cols, vals, symmetric=False)
    text = open(general).read().splitlines()
    assert text[0] == "%%MatrixMarket matrix coordinate real general" and text[2].split() == [str(m), str(m), str(len(rows))]
    assert len(text) == 3 + len(rows) and float(text[3].split()[2]) == vals[0]  # (%.17g round-trips a double)
    hr, hc = read_mtx_coordinates(general)
    key = np.unique(gr * m + gc)
    assert np.array_equal(hr * m + hc, key)
    # symmetric file: one triangle stored, duplicates, pattern values
    lo = np.unique(np.maximum(gr, gc) * m + np.minimum(gr, gc))
    dup = np.concatenate([lo, lo[::3]])
    sym = str(tmp_path / "sym.mtx")
    H.write_matrix_market(sym, m, m, dup // m, dup % m, None, symmetric=True)
    i, j = lo // m, lo % m
    want = np.unique(np.concatenate([i * m + j, j * m + i]))
    hr, hc = read_mtx_coordinates(sym)
    assert np.array_equal(hr * m + hc, want)
    for path, count in ((general, len(key)), (sym, len(want))):
        def body(w, path=path):
            sp = H.SpmatLocal.load_tuples(w, True, 0, 0, path)
            info = sp.info()
            sp.free()
            return info["dist_nnz"], info["M"], info["N"]
        assert set(H.run_spmd(8, body)) == {(count, m, m)}
