"""The reference's own `main()`s, unmodified, on the HIP library (the file name makes this the LAST GPU test file).

oracle/_ref/mains/* are bench_erdos_renyi.cpp, bench_file.cpp, bench_heatmap.cpp (+ benchmark_dist.cpp) and scratch.cpp of the
reference compiled in the build container against include/compat and linked with lib/libhnh_host.so (oracle/build_ref_mains.sh,
run by __graft_entry__.build(); the reference's sources do not exist on the GPU box).  Their CPU twin — which also proves that they
compile unchanged — is tests/test_reference_mains_cpu.py.  Here: the same command lines a user of the reference types, on one
MI355X, and bench_erdos_renyi as two processes that share the GPU over the ipc-pull transport."""
import json
import os
import subprocess
import time

import numpy as np
import pytest

import hnh_testlib as T

pytestmark = pytest.mark.gpu
BIN = os.path.join(T.ROOT, "oracle", "_ref", "mains")


@pytest.fixture(scope="module")
def env():
    if not all(os.path.exists(os.path.join(BIN, m)) for m in ("bench_erdos_renyi", "bench_file", "bench_heatmap", "scratch")):
        pytest.skip("oracle/_ref/mains was not built (needs the reference's sources: run __graft_entry__.build() where /root/reference exists)")
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HNH_TRANSPORT", "HNH_ID_FILE"):
        e.pop(k, None)
    return e


def records(path):
    return json.loads("[" + open(path).read().rstrip().rstrip(",") + "]")


def run(env, exe, *args):
    r = subprocess.run([os.path.join(BIN, exe), *args], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return r.stdout


def test_the_references_benchmark_mains_on_hip(env, tmp_path):
    out = tmp_path / "er.json"
    run(env, "bench_erdos_renyi", "12", "8", "15d", "32", "1", str(out))   # bench_erdos_renyi.cpp: logM edgeFactor 15d|25d R c outfile
    run(env, "bench_erdos_renyi", "12", "8", "25d", "32", "1", str(out))
    recs = records(out)
    assert [(r["alg_name"], r["fused"]) for r in recs] == [("15d_fusion1", True), ("15d_fusion2", True), ("25d_sparse_replicate", False), ("25d_dense_replicate", True)]
    for r in recs:
        assert set(r) == {"elapsed", "overall_throughput", "fused", "num_trials", "alg_name", "alg_info", "application_communication_time", "perf_stats"}
        assert r["num_trials"] == 5 and r["alg_info"]["backend"] == "hip-gfx950" and r["alg_info"]["m"] == 4096 and "Computation Time" in r["perf_stats"]
    mtx = str(tmp_path / "g.mtx")
    mrows, _, _ = T.write_symmetric_mtx_with_duplicates(mtx, 500, 3)
    fout = tmp_path / "file.json"
    assert "File reader read %d nonzeros." % len(mrows) in run(env, "bench_file", mtx, "15d", "64", "1", str(fout), "vanilla")
    run(env, "bench_file", mtx, "25d", "64", "1", str(fout), "als")
    assert [(r["alg_name"], r["alg_info"]["nnz"]) for r in records(fout)] == [("15d_sparse", len(mrows)), ("25d_dense_replicate", len(mrows))]
    heat = tmp_path / "heat.json"
    run(env, "bench_heatmap", "10", "8", "25d", "1", str(heat))   # bench_heatmap.cpp:33: R = 64 ... 448
    assert [(r["alg_name"], r["alg_info"]["r"]) for r in records(heat)] == [(a, r) for r in (64, 128, 192, 256, 320, 384, 448)
                                                                            for a in ("25d_sparse_replicate", "25d_dense_replicate")]


def test_scratch_cpp_on_hip_prints_the_references_fingerprints(env, tmp_path):
    """scratch.cpp: 1.5D sparse shift, verify_operation (six significant digits on the stream), then the GAT forward pass."""
    from oracle import oracle as O
    mtx = str(tmp_path / "g.mtx")
    mrows, mcols, _ = T.write_symmetric_mtx_with_duplicates(mtx, 500, 3)
    text = run(env, "scratch", mtx, "32", "1")
    got = np.array([float(ln.split(":")[1]) for ln in text.splitlines() if "Fingerprint:" in ln])
    want = np.array(O.fingerprints(mrows, mcols, 500, 500, 32))
    assert got.shape == (3,) and np.max(np.abs(got - want) / want) <= 1e-5, (got, want)


def test_bench_erdos_renyi_as_two_processes_sharing_the_gpu(env, tmp_path):
    """`mpiexec -n 2 ./bench_erdos_renyi ...` of the reference = two processes with RANK / WORLD_SIZE and a transport in the environment."""
    out = tmp_path / "er2.json"
    session = "z%d_%x" % (os.getpid(), time.time_ns())
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), HNH_DEVICE="0", HNH_TRANSPORT="ipc", HNH_IPC_SESSION=session, HNH_IPC_WAIT_S="120",
                 HNH_IPC_PULL="kernel")  # (copy-engine pulls between processes that share ONE GPU cost ~0.1 s per dependency: tests/test_multigpu_gpu.py)
        procs.append(subprocess.Popen([os.path.join(BIN, "bench_erdos_renyi"), "10", "8", "15d", "32", "1", str(out)], env=e, stdout=subprocess.PIPE,
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
    assert [(r["alg_name"], r["alg_info"]["p"], r["alg_info"]["backend"]) for r in records(out)] == [("15d_fusion1", 2, "hip-gfx950"), ("15d_fusion2", 2, "hip-gfx950")]
