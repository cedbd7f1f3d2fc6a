"""The reference's own `main()`s, UNCHANGED, against this repository's class headers.

bench_erdos_renyi.cpp, bench_file.cpp, bench_heatmap.cpp (+ benchmark_dist.cpp, the harness they call) and scratch.cpp (the
reference's only correctness check, `verify_operation`, plus a GAT forward pass) are copied from /root/reference into a temp dir at
test time — never into the repository — and compiled with nothing but `-I include/compat` (oracle/build_ref_mains.sh): their includes (`benchmark_dist.hpp`,
`15D_dense_shift.hpp` ... `json.hpp`, and through them `<mpi.h>` and `common.h`), `MPI_Init` / `initialize_mpi_datatypes` /
`MPI_Allreduce(MPI_IN_PLACE, ...)` / `MPI_Finalize`, `using json = nlohmann::json`, the `NonzeroDistribution` subclass of scratch.cpp
all resolve.  They then RUN on the kernel test double (a directory holding the host library and the C test double under the kernel
library's name stands in for a GPU box, as in tests/test_tools_cpu.py): one process, and — the launcher's environment
(RANK / WORLD_SIZE, HNH_TRANSPORT=ipc) being what MPI's launcher is to the reference — two and four processes over the ipc-pull
transport.  The fingerprints scratch.cpp prints are the oracle's (= the compiled reference's) for the same file."""
import json
import os
import shutil
import subprocess
import time

import numpy as np
import pytest

import hnh_testlib as T
from test_ipc_world_cpu import can_read_peer_memory

REF = "/root/reference"
MAINS = ("bench_erdos_renyi", "bench_file", "bench_heatmap", "scratch")


@pytest.fixture(scope="module")
def mains(tmp_path_factory):
    if not os.path.exists(os.path.join(REF, "scratch.cpp")):
        pytest.skip("the reference's sources are not on this box")
    # oracle/build_ref_mains.sh: the six reference files copied to a temp dir, compiled with -I include/compat and nothing else changed,
    # linked with lib/libhnh_host.so -> oracle/_ref/mains/ (the same binaries run on the HIP library in tests/test_zz_reference_mains_gpu.py)
    r = subprocess.run(["bash", os.path.join(T.ROOT, "oracle", "build_ref_mains.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "the reference's mains do not compile unchanged against include/compat:\n" + (r.stdout + r.stderr)[-4000:]
    d = tmp_path_factory.mktemp("refmains")
    libdir = d / "lib"
    libdir.mkdir()
    shutil.copy(os.path.join(T.ROOT, "distributed_sddmm_amd", "lib", "libhnh_host.so"), libdir / "libhnh_host.so")
    shutil.copy(T.ORACLE_BACKEND, libdir / "libhnh_kernels.so")
    env = dict(os.environ, LD_LIBRARY_PATH=str(libdir), OMP_NUM_THREADS="2", HNH_HOST_SETUP="1")  # (LD_LIBRARY_PATH goes before the binaries' RUNPATH)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HNH_TRANSPORT", "HNH_ID_FILE"):
        env.pop(k, None)
    mtx = str(d / "g.mtx")
    rows, cols, _ = T.write_symmetric_mtx_with_duplicates(mtx, 256, 4)
    return dict(dir=d, bin=os.path.join(T.ROOT, "oracle", "_ref", "mains"), env=env, mtx=mtx, rows=rows, cols=cols)


def records(path):
    return json.loads("[" + open(path).read().rstrip().rstrip(",") + "]")


def run(mains, exe, *args):
    r = subprocess.run([os.path.join(mains["bin"], exe), *args], env=mains["env"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return r.stdout


def run_ranks(mains, n, exe, *args):
    """`exe` as n processes over the ipc-pull transport (what `mpiexec -n` is to the reference); returns rank 0's output."""
    session = "m%d_%x" % (os.getpid(), time.time_ns())
    procs = []
    for r in range(n):
        env = dict(mains["env"], RANK=str(r), WORLD_SIZE=str(n), LOCAL_RANK=str(r), HNH_DEVICE="0", HNH_TRANSPORT="ipc", HNH_IPC_SESSION=session,
                   HNH_IPC_WAIT_S="120")
        procs.append(subprocess.Popen([os.path.join(mains["bin"], exe), *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1000:] for o in outs)
    assert not [f for f in os.listdir("/dev/shm") if session in f], "the exit handler of the process world left its shared-memory session behind"
    return outs[0]


def test_the_three_benchmark_mains_run_unchanged(mains):
    d = mains["dir"]
    # bench_erdos_renyi.cpp:19-120: logM edgeFactor 15d|25d R c outfile
    run(mains, "bench_erdos_renyi", "9", "8", "15d", "16", "1", str(d / "er.json"))
    run(mains, "bench_erdos_renyi", "8", "8", "25d", "16", "1", str(d / "er.json"))
    recs = records(d / "er.json")
    assert [(r["alg_name"], r["fused"]) for r in recs] == [("15d_fusion1", True), ("15d_fusion2", True), ("25d_sparse_replicate", False), ("25d_dense_replicate", True)]
    for r in recs:
        assert set(r) == {"elapsed", "overall_throughput", "fused", "num_trials", "alg_name", "alg_info", "application_communication_time", "perf_stats"}
        assert r["num_trials"] == 5 and r["alg_info"]["backend"] == "oracle-cpu-test-double" and r["alg_info"]["p"] == 1 and "Computation Time" in r["perf_stats"]
    assert recs[0]["alg_info"]["m"] == 512 and recs[2]["alg_info"]["m"] == 256
    # bench_file.cpp:19-103: file 15d|25d R c outfile app
    assert "File reader read %d nonzeros." % len(mains["rows"]) in run(mains, "bench_file", mains["mtx"], "15d", "16", "1", str(d / "file.json"), "vanilla")
    run(mains, "bench_file", mains["mtx"], "25d", "16", "1", str(d / "file.json"), "als")
    recs = records(d / "file.json")
    assert [(r["alg_name"], r["alg_info"]["nnz"]) for r in recs] == [("15d_sparse", len(mains["rows"])), ("25d_dense_replicate", len(mains["rows"]))]
    # bench_heatmap.cpp:17-109: logM edgeFactor 15d|25d c outfile; R = 64 ... 448
    run(mains, "bench_heatmap", "7", "4", "15d", "1", str(d / "heat.json"))
    recs = records(d / "heat.json")
    assert [(r["alg_name"], r["alg_info"]["r"]) for r in recs] == [(a, r) for r in (64, 128, 192, 256, 320, 384, 448) for a in ("15d_fusion1", "15d_fusion2", "15d_sparse")]


def fingerprints(text):
    got = [float(ln.split(":")[1]) for ln in text.splitlines() if "Fingerprint:" in ln]
    assert len(got) == 3, text[-1500:]
    return np.array(got)


def test_scratch_cpp_prints_the_references_fingerprints(mains):
    """scratch.cpp:78-148: file R c — 1.5D sparse shift, verify_operation, then the GAT forward pass of the benchmark's three layers.
    The stream prints six significant digits."""
    from oracle import oracle as O
    want = np.array(O.fingerprints(mains["rows"], mains["cols"], 256, 256, 16))
    assert np.max(np.abs(fingerprints(run(mains, "scratch", mains["mtx"], "16", "1")) - want) / want) <= 1e-5


@pytest.mark.parametrize("n,c", [(2, 1), (4, 2)])
def test_the_mains_across_processes_over_ipc(mains, n, c):
    """The same binaries as n processes: MPI_Init picks the transport from the launcher's environment, the exit handler of the process
    world ends the session.  Same records (p = n), same fingerprints."""
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    from oracle import oracle as O
    d = mains["dir"]
    out = d / ("er_%d.json" % n)
    run_ranks(mains, n, "bench_erdos_renyi", "9", "8", "15d", "16", str(c), str(out))
    recs = records(out)
    assert [(r["alg_name"], r["alg_info"]["p"], r["alg_info"]["c"]) for r in recs] == [("15d_fusion1", n, c), ("15d_fusion2", n, c)]
    want = np.array(O.fingerprints(mains["rows"], mains["cols"], 256, 256, 16))
    assert np.max(np.abs(fingerprints(run_ranks(mains, n, "scratch", mains["mtx"], "16", str(c))) - want) / want) <= 1e-5


def test_a_rank_that_fails_ends_its_peers_at_once(mains):
    """One rank of two cannot open its input and exits through hnh::fatal (the reference's print-and-exit(1) convention): the other, waiting
    for it inside the collective read, is told through the session's `failed` word and ends with a message within seconds — not at the
    transport's time limit (set to ten minutes here) — and the exit handlers leave no shared-memory session behind."""
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    session = "f%d_%x" % (os.getpid(), time.time_ns())
    procs, t0 = [], time.time()
    for r, path in ((0, mains["mtx"]), (1, str(mains["dir"] / "missing.mtx"))):
        env = dict(mains["env"], RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), HNH_DEVICE="0", HNH_TRANSPORT="ipc", HNH_IPC_SESSION=session, HNH_IPC_WAIT_S="600")
        procs.append(subprocess.Popen([os.path.join(mains["bin"], "scratch"), path, "16", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    try:
        outs = [p.communicate(timeout=120)[0] for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert [p.returncode for p in procs] == [1, 1] and time.time() - t0 < 60
    assert "cannot open matrix file" in outs[1] and "a peer rank of the ipc world gave up" in outs[0], outs
    assert not [f for f in os.listdir("/dev/shm") if session in f]


def test_the_references_launch_line_with_mpiexec(mains):
    """`mpiexec -n 4 ./bench_erdos_renyi 9 8 15d 16 2 out.json` — the reference's launch line, MPICH's launcher, the reference's unmodified
    main — over this engine: MPI_Init reads PMI_RANK / PMI_SIZE / MPI_LOCALRANKID (the launcher only starts the processes; no MPI library
    is linked) and the ranks meet in a session named after the launch (no HNH_IPC_SESSION)."""
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not os.path.exists(mpiexec):
        pytest.skip("no mpiexec on this box")
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    out = mains["dir"] / "er_mpiexec.json"
    env = dict(mains["env"], HNH_TRANSPORT="ipc", HNH_DEVICE="0", HNH_IPC_WAIT_S="120")
    r = subprocess.run([mpiexec, "-n", "4", os.path.join(mains["bin"], "bench_erdos_renyi"), "9", "8", "15d", "16", "2", str(out)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert [(x["alg_name"], x["alg_info"]["p"], x["alg_info"]["c"], x["alg_info"]["transport"]) for x in records(out)] == [
        ("15d_fusion1", 4, 2, "ipc-pull"), ("15d_fusion2", 4, 2, "ipc-pull")]


def test_the_references_launch_line_on_the_default_transport(mains):
    """The same launch line with NOTHING else in the environment: the default transport of a multi-process launch is RcclWorld — rank 0
    leaves the unique id in a file named after the launch, removes it once the communicator exists — here over the test double's
    emulation of RCCL's calling contract between processes (tests/test_rccl_emulation_cpu.py).  Then scratch.cpp the same way."""
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not os.path.exists(mpiexec):
        pytest.skip("no mpiexec on this box")
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    from oracle import oracle as O
    out = mains["dir"] / "er_rccl.json"
    token = "t%d_%x" % (os.getpid(), time.time_ns())
    env = dict(mains["env"], HNH_DEVICE="0", HNH_JOB_TOKEN=token, HNH_ORACLE_COMM_WAIT_S="120")
    env.pop("HNH_TRANSPORT", None)
    r = subprocess.run([mpiexec, "-n", "4", os.path.join(mains["bin"], "bench_erdos_renyi"), "9", "8", "15d", "16", "2", str(out)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert [(x["alg_name"], x["alg_info"]["p"], x["alg_info"]["c"], x["alg_info"]["transport"]) for x in records(out)] == [
        ("15d_fusion1", 4, 2, "rccl"), ("15d_fusion2", 4, 2, "rccl")]
    r = subprocess.run([mpiexec, "-n", "2", os.path.join(mains["bin"], "scratch"), mains["mtx"], "16", "1"], env=dict(env, HNH_JOB_TOKEN=token + "s"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    want = np.array(O.fingerprints(mains["rows"], mains["cols"], 256, 256, 16))
    assert np.max(np.abs(fingerprints(r.stdout) - want) / want) <= 1e-5
    assert not [f for f in os.listdir("/dev/shm") if token in f or f.startswith("hnh_emu_")], "the id file or the emulation's segment was left behind"
