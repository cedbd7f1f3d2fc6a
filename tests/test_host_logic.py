"""CPU-only checks of host-side pieces that need no kernel at all: C-ABI symbol coverage of the host library,
the synthetic generator against its numpy restatement, MatrixMarket input, FlexibleGrid rank maps and
sub-communicators (FlexibleGrid.hpp:41-135) for all six adjacency orders."""
import os
import re

import numpy as np
import pytest

import hnh_testlib as T
from distributed_sddmm_amd import api as H
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True, scope="module")
def cpu_test_double():
    H.load_backend(T.ORACLE_BACKEND)
    yield


def test_host_library_exports_every_declared_symbol():
    txt = open(os.path.join(ROOT, "include", "hnh_dist.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = set(re.findall(r"\b(hnh_[a-z0-9_A-Z]+)\s*\(", txt)) - {"hnh_comm_callbacks"}
    assert declared == set(H.SIGNATURES), declared ^ set(H.SIGNATURES)
    lib = H.lib()
    for name in declared:
        assert getattr(lib, name) is not None


@pytest.mark.parametrize("m,n,draws,seed", [(256, 256, 2048, 12345), (250, 333, 5000, 7), (1 << 14, 1 << 14, 1 << 18, 99), (5, 3, 100, 1)])
def test_native_generator_is_bit_identical_to_the_oracle(m, n, draws, seed):
    r1, c1 = H.generate_er(m, n, draws, seed)
    r2, c2 = O.erdos_renyi_mn(m, n, draws, seed)
    assert np.array_equal(r1, r2) and np.array_equal(c1, c2)
    keys = r1 * n + c1
    assert np.all(np.diff(keys) > 0), "sorted row-major and free of duplicates"


def test_load_tuples_partitions_the_generated_matrix():
    def body(w):
        sp = H.SpmatLocal.load_tuples(w, False, 8, 8)
        info = sp.info()
        sp.free()
        return info

    infos = H.run_spmd(4, body)
    rows, _ = O.erdos_renyi(8, 8)
    assert all(i["dist_nnz"] == len(rows) and i["M"] == 256 and i["N"] == 256 for i in infos)
    assert sum(i["local_nnz"] for i in infos) == len(rows)


def test_matrix_market_reader(tmp_path):
    path = tmp_path / "m.mtx"
    path.write_text("%%MatrixMarket matrix coordinate real symmetric\n% comment\n4 4 4\n1 1 2.0\n3 1 5.0\n3 1 7.0\n4 2 1.5\n")

    def body(w):
        sp = H.SpmatLocal.load_tuples(w, True, -1, -1, str(path))
        info = sp.info()
        sp.free()
        return info

    info = H.run_spmd(1, body)[0]
    # (1,1) ; (3,1)+(1,3) with duplicate -> max kept ; (4,2)+(2,4)
    assert info == {"M": 4, "N": 4, "dist_nnz": 5, "local_nnz": 5}


def test_matrix_market_file_that_ends_on_a_page_boundary(tmp_path):
    """The parser reads a memory mapping: a real-valued file of exactly one (two) pages whose last number is not followed by a
    line feed must not make the value parser run past the mapping."""
    head, body = "%%MatrixMarket matrix coordinate real general\n", "3 3 2\n1 1 2.5\n3 2 1e-3"
    for target in (4096, 8192):
        text = head + "%" + "x" * (target - len(head) - len(body) - 2) + "\n" + body
        assert len(text) == target
        path = tmp_path / ("page_%d.mtx" % target)
        path.write_text(text)

        def body_fn(w, path=path):
            sp = H.SpmatLocal.load_tuples(w, True, -1, -1, str(path))
            info = sp.info()
            sp.free()
            return info

        assert H.run_spmd(1, body_fn)[0] == {"M": 3, "N": 3, "dist_nnz": 2, "local_nnz": 2}


@pytest.mark.parametrize("body", ["3 3 2\n1 2\n3 3 5.0\n", "3 3 2\n1 1 2.5\n3 2", "3 3 2\n1 1 2.5\n3 2   \n"])
def test_matrix_market_line_without_a_value_is_refused(tmp_path, body):
    """A `real` file with an entry line that has no value: the value parser must not borrow the next line's first token (the
    entry count would still add up) nor run past the end of the mapping on the file's last line — the file is malformed."""
    path = tmp_path / "novalue.mtx"
    path.write_text("%%MatrixMarket matrix coordinate real general\n" + body)

    def body_fn(w):
        sp = H.SpmatLocal.load_tuples(w, True, -1, -1, str(path))
        sp.free()

    with pytest.raises(Exception, match="malformed"):
        H.run_spmd(1, body_fn)


@pytest.mark.parametrize("dims,adjacency", [((4, 2, 1), 1), ((2, 2, 2), 3), ((2, 3, 2), 2), ((3, 2, 2), 4), ((2, 2, 3), 5), ((1, 4, 3), 6)])
def test_flexible_grid(dims, adjacency):
    nr, nc, nh = dims
    p = nr * nc * nh
    res = H.run_spmd(p, lambda w: (w.rank,) + tuple(w.grid_probe(nr, nc, nh, adjacency)))
    perms = {1: (0, 1, 2), 2: (0, 2, 1), 3: (1, 0, 2), 4: (1, 2, 0), 5: (2, 0, 1), 6: (2, 1, 0)}[adjacency]
    seen = set()
    for rank, vals, ok in res:
        i, j, k, in_row, in_col, in_fiber, rs, cs, fs = vals
        assert ok, "broadcast self-test (FlexibleGrid.hpp:169-201)"
        t = [i, j, k]
        d = [nr, nc, nh]
        assert rank == t[perms[0]] + t[perms[1]] * d[perms[0]] + t[perms[2]] * d[perms[0]] * d[perms[1]]
        assert (in_row, in_col, in_fiber) == (j, i, k) and (rs, cs, fs) == (nc, nr, nh)
        seen.add((i, j, k))
    assert len(seen) == p


def test_product_backend_is_hip_and_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert H.load_backend(None) == "hip-gfx950"  # the library itself loads (no GPU needed to dlopen)
    with pytest.raises(H.HnhError) as e:
        H.World.single(0)
    assert "no GPU" in str(e.value) or "device" in str(e.value)
    H.load_backend(T.ORACLE_BACKEND)


def test_custom_kernel_plugin_drives_every_schedule():
    """The reference's extension point (README.md:17-18): a KernelImplementation subclass with only sddmm_local and
    spmm_local (examples/custom_kernel.cpp) handed to every schedule constructor — here over the oracle's C test double;
    the local-kernel-fusion schedule then runs on the default fused_local(), the reference's own call pair."""
    import subprocess
    exe = os.path.join(T.ROOT, "examples", "custom_kernel")
    r = subprocess.run(["make", "-C", os.path.join(T.ROOT, "examples"), "custom_kernel"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    r = subprocess.run([exe, T.ORACLE_BACKEND, "8", "8", "16"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all schedules ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stdout.count(" ok") >= 5


def test_failing_rank_does_not_hang_its_peers(monkeypatch):
    """One logical rank of a thread group fails before a collective: its peers, blocked in that collective, give up with an
    error after HNH_THREAD_WAIT_S instead of waiting for ever (round-1 advisor finding)."""
    import time
    monkeypatch.setenv("HNH_THREAD_WAIT_S", "2")

    def body(w):
        if w.rank == 1:
            sp = H.SpmatLocal.from_global(w, 8, 8, np.array([0]), np.array([0]), None)
            with pytest.raises(H.HnhError):
                H.DistributedSparse(w, "15d_fusion2", sp, 8, 3)  # c = 3 does not divide p = 2: configuration error
            return "failed as expected"
        with pytest.raises(H.HnhError, match="did not arrive"):
            w.barrier()  # rank 1 never arrives
        return "gave up"

    t0 = time.time()
    assert H.run_spmd(2, body) == ["gave up", "failed as expected"]
    assert time.time() - t0 < 60


@pytest.mark.parametrize("p", [1, 2, 3, 4, 8])
def test_transport_preflight_on_the_loopback_transport(p):
    """hnh_world_preflight: every transport primitive the schedules use delivers the expected data on p logical ranks, and
    all ranks end with the same communicator-split signature (what bench.py --gpus N checks before it times anything)."""
    def body(w):
        errs = [w.preflight(k, 1000) for k in range(len(H.World.PREFLIGHT))]
        return errs, w.split_signature()

    res = H.run_spmd(p, body)
    assert all(max(errs) == 0.0 for errs, _ in res)
    assert len({sig for _, sig in res}) == 1 and res[0][1][1] == 2 * len(H.World.PREFLIGHT)


def test_the_references_own_harness_compiles_against_the_class_headers(tmp_path):
    """The reference's benchmark_dist.cpp, UNCHANGED (copied from /root/reference at test time, never into the repository),
    compiled against include/compat — its includes (`distributed_sparse.h`, `json.hpp`, `mpi.h` ...), `using json =
    nlohmann::json`, `j_obj["alg_info"] = d_ops->json_algorithm_info()` and `.dump(4)` all resolve — and run on the kernel test
    double: three records with the reference's keys."""
    import json
    import shutil
    import subprocess
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "benchmark_dist.cpp")):
        pytest.skip("the reference's sources are not on this box")
    for f in ("benchmark_dist.cpp", "benchmark_dist.hpp"):
        shutil.copy(os.path.join(ref, f), tmp_path / f)
    (tmp_path / "main.cpp").write_text('''
#include "benchmark_dist.hpp"
#include "world.hpp"
int main(int argc, char** argv) {
    hnh::load_backend(argv[1]);
    hnh::SingleWorld world(hnh::default_backend(), 0);
    hnh::set_current_world(&world);
    {
        SpmatLocal S;
        S.loadTuples(false, 8, 8, "");
        benchmark_algorithm(&S, "15d_fusion2", argv[2], true, 16, 1, "vanilla");
        benchmark_algorithm(&S, "15d_sparse", argv[2], false, 16, 1, "vanilla");
        benchmark_algorithm(&S, "25d_dense_replicate", argv[2], true, 16, 1, "als");
    }
    world.sync_all();
    hnh::set_current_world(nullptr);
    return 0;
}
''')
    lib = os.path.join(T.ROOT, "distributed_sddmm_amd", "lib")
    subprocess.run(["g++", "-O1", "-std=c++17", "-fopenmp", "-w", "-I" + os.path.join(T.ROOT, "include", "compat"),
                    "-I" + os.path.join(T.ROOT, "distributed_sddmm_amd", "csrc", "host"), "-I" + os.path.join(T.ROOT, "include"),
                    "benchmark_dist.cpp", "main.cpp", "-o", "refbench", "-L" + lib, "-lhnh_host", "-ldl", "-lpthread", "-Wl,-rpath," + lib],
                   cwd=tmp_path, check=True, capture_output=True, timeout=600)
    out = tmp_path / "out.json"
    res = subprocess.run([str(tmp_path / "refbench"), T.ORACLE_BACKEND, str(out)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    recs = json.loads("[" + out.read_text().rstrip().rstrip(",") + "]")
    assert [r["alg_name"] for r in recs] == ["15d_fusion2", "15d_sparse", "25d_dense_replicate"]
    for r in recs:
        assert set(r) == {"elapsed", "overall_throughput", "fused", "num_trials", "alg_name", "alg_info", "application_communication_time", "perf_stats"}
        assert r["num_trials"] == 5 and r["alg_info"]["m"] == 256 and r["alg_info"]["p"] == 1 and "Computation Time" in r["perf_stats"]


def test_an_error_after_a_world_was_destroyed_is_still_an_error():
    """hnh::fatal tells the failing thread's CURRENT world (World::note_failure: an ipc session's peers stop waiting at once); the
    thread's current-world pointer may outlive its world — closed here, or made current on another thread — and must then be left
    alone: the next failing call still comes back as an error."""
    import threading
    w = H.World.single(0)
    w.barrier()
    w.close()
    with pytest.raises(H.HnhError):
        H.World.rccl(0, 2, 0, bytes(128))  # (the test double has no RCCL)
    box = {}

    def other_thread():
        box["w"] = H.World.single(0)
        box["w"].barrier()
    t = threading.Thread(target=other_thread)
    t.start()
    t.join()
    box["w"].close()
    with pytest.raises(H.HnhError):
        H.World.rccl(0, 2, 0, bytes(128))
