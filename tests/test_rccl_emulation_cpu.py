"""RcclWorld — the product's default multi-GPU transport — across 2 / 4 / 8 ranks WITHOUT GPUs: the RCCL section of the kernel ABI
emulated by the test double between ranks that are threads of one process or processes of one host (oracle/hnh_oracle_backend.c,
"the RCCL section ... EMULATED": the communicator's state in a shared-memory segment named after the unique id).

What the emulation holds the host layer to is the calling contract RCCL / NCCL document, which is all RcclWorld depends on: the
communicator is formed collectively from one unique id; the point-to-point operations of one ncclGroupStart/End are issued together;
a send and a receive match in the ORDER in which they were issued for their (source, destination) pair, and their sizes must agree;
collectives are called by every rank in the same order; an operation that never finds its partner HANGS on the GPU — here it fails
after HNH_ORACLE_COMM_WAIT_S.  Data moves with memcpy, and the stream-order checker (tests/test_stream_order_cpu.py) sees a send as a
read on the sender's stream, a receive as a write on the receiver's stream behind the matching send, a collective as behind every
rank's contribution — so a schedule that issued its groups in an order RCCL would deadlock on, mismatched a size, or let a kernel
touch a buffer a transfer still owns, fails here.

The GPU twin (tests/test_multigpu_gpu.py::test_schedules_over_rccl, tests/rccl_worker.py) runs the SAME configuration lists on real
GPUs wherever two or more are visible; the driver's box has one, so this file is the only place RcclWorld's multi-rank paths —
explicit-peer groups for sub-communicator collectives, native collectives on the world, host data staged through device memory,
the barrier — have run with more than one rank."""
import ctypes
import os
import threading

import numpy as np
import pytest

import hnh_testlib as T
from distributed_sddmm_amd import api as H
from test_multigpu_gpu import ALL_2, ALL_4, ALL_8


@pytest.fixture(scope="module")
def checker():
    assert H.load_backend(T.ORACLE_BACKEND) == "oracle-cpu-test-double"
    lib = ctypes.CDLL(T.ORACLE_BACKEND)
    lib.hnh_oracle_order_report.restype = ctypes.c_long
    lib.hnh_oracle_order_enable(1)

    def drain():
        buf = ctypes.create_string_buffer(16384)
        n = lib.hnh_oracle_order_report(buf, 16384)
        return n, buf.value.decode()
    assert drain()[0] == 0
    return drain


def over_rccl(n, body):
    """body(world) on n ranks = n threads, each with its own RcclWorld of one communicator; returns the per-rank results."""
    ident = H.rccl_unique_id()
    out, errs = [None] * n, []

    def rank_main(r):
        try:
            w = H.World.rccl(r, n, 0, ident)
            try:
                out[r] = body(w)
            finally:
                w.close()
        except BaseException as e:  # noqa: BLE001
            errs.append("rank %d: %r" % (r, e))
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, "\n".join(errs)
    return out


@pytest.mark.parametrize("nranks,configs", [(2, ALL_2), (4, ALL_4), (8, ALL_8)], ids=["2", "4", "8"])
def test_schedules_over_the_rccl_emulation(checker, monkeypatch, nranks, configs):
    """The configuration lists of the GPU test, both cases: every transport primitive (preflight), then all five schedules (relay ring,
    chunked mesh fetch, replication collectives as explicit-peer groups, travelling sparse blocks, Cannon's two rings) and ALS with
    the held operand, element-wise against the reference's golden vectors; no race of the stream protocol."""
    monkeypatch.setenv("HNH_ORACLE_COMM_WAIT_S", "60")
    for case_name in ("er8_r16", "ragged_r8"):
        case = T.case_inputs(case_name)
        errs = over_rccl(nranks, lambda w: [w.preflight(what, 1 << 12) for what in range(len(H.World.PREFLIGHT))])
        assert all(e <= 1e-9 for per_rank in errs for e in per_rank), errs
        for item in configs.split(";"):
            alg, c, mode, chunks = item.split(":")
            monkeypatch.setenv("HNH_RING_MODE", mode)
            monkeypatch.setenv("HNH_MESH_CHUNKS", chunks)
            if alg.startswith("als@"):
                per_rank = over_rccl(nranks, lambda w: T.run_als(w, alg[4:], int(c), case, 1, 5))
                T.check_als_against_golden(per_rank, case)
                continue
            if not T.valid_config(alg, nranks, int(c), case["R"]):
                continue
            per_rank = over_rccl(nranks, lambda w: T.run_all_ops(w, alg, int(c), case))
            T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)
            assert per_rank[0]["alg_info"]["transport"] == "rccl", item
        n, text = checker()
        assert n == 0, text


def test_gat_and_grids_with_remainders_over_the_rccl_emulation(checker, monkeypatch):
    monkeypatch.setenv("HNH_ORACLE_COMM_WAIT_S", "60")
    case = T.case_inputs("er8_r16")
    gold = dict(np.load(os.path.join(T.GOLDEN, "gat_er8_r16.npz")))
    for alg, p, c in (("15d_fusion1", 4, 2), ("15d_fusion2", 4, 1)):
        per_rank = over_rccl(p, lambda w: T.run_gat(w, alg, c, case))
        out = T.assemble_dense(per_rank, "gat", "subA", case["M"], T.GAT_LAYERS[-1][1] * T.GAT_LAYERS[-1][2])
        assert T.rel(out, gold["out"]) <= T.TOL
    for alg, p, c in (("15d_fusion2", 3, 1), ("15d_fusion1", 6, 2), ("15d_sparse", 8, 2), ("25d_dense_replicate", 16, 4), ("25d_sparse_replicate", 16, 4)):
        per_rank = over_rccl(p, lambda w: T.run_all_ops(w, alg, c, case))
        T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)
    n, text = checker()
    assert n == 0, text


# ---- the emulation itself: it must refuse what RCCL would hang on or corrupt
def raw_comms(n):
    """n (ctx, comm) pairs of one communicator through the kernel ABI, created from n threads (the init is collective)."""
    from distributed_sddmm_amd import _kernels as K
    lib = K.load(T.ORACLE_BACKEND)
    ident = ctypes.create_string_buffer(128)
    assert lib.hnh_comm_unique_id(ident) == 0
    pairs = [None] * n

    def init(r):
        ctx, comm = ctypes.c_void_p(), ctypes.c_void_p()
        assert lib.hnh_ctx_create(0, ctypes.byref(ctx)) == 0
        assert lib.hnh_comm_init(ctx, n, r, ident, ctypes.byref(comm)) == 0, lib.hnh_last_error(ctx)
        pairs[r] = (ctx, comm)
    threads = [threading.Thread(target=init, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(pairs)
    return lib, pairs


def in_threads(fns):
    res = [None] * len(fns)

    def run(i):
        res[i] = fns[i]()
    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(fns))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return res


def test_the_emulation_matches_pairs_in_issue_order_and_refuses_what_rccl_would_hang_on(checker, monkeypatch):
    from distributed_sddmm_amd import _kernels as K
    monkeypatch.setenv("HNH_ORACLE_COMM_WAIT_S", "2")
    lib, pairs = raw_comms(2)
    bufs = []
    for ctx, _ in pairs:
        p = ctypes.c_void_p()
        assert lib.hnh_malloc(ctx, 4096, ctypes.byref(p)) == 0
        bufs.append(p)

    def fill(r, off, n, v):
        assert lib.hnh_fill_f64(pairs[r][0], bufs[r].value + off, n, v, K.STREAM_COMM) == 0

    def read(r, off, n):
        host = np.zeros(n)
        assert lib.hnh_memcpy(pairs[r][0], host.ctypes.data, bufs[r].value + off, 8 * n, K.D2H, K.STREAM_COMM) == 0
        assert lib.hnh_stream_sync(pairs[r][0], K.STREAM_COMM) == 0
        return host
    # two sends 0 -> 1 in one group arrive in the order they were issued (FIFO per pair), whatever order the receives name the buffers in
    fill(0, 0, 16, 1.0); fill(0, 128, 16, 2.0); fill(1, 0, 64, 0.0)

    def rank0():
        ctx, comm = pairs[0]
        assert lib.hnh_comm_group_begin(ctx) == 0
        assert lib.hnh_comm_sendrecv(ctx, comm, bufs[0].value, 128, 1, None, 0, 1, K.STREAM_COMM) == 0
        assert lib.hnh_comm_sendrecv(ctx, comm, bufs[0].value + 128, 128, 1, None, 0, 1, K.STREAM_COMM) == 0
        return lib.hnh_comm_group_end(ctx)

    def rank1():
        ctx, comm = pairs[1]
        assert lib.hnh_comm_group_begin(ctx) == 0
        assert lib.hnh_comm_sendrecv(ctx, comm, None, 0, 0, bufs[1].value + 256, 128, 0, K.STREAM_COMM) == 0   # first receive: the first send
        assert lib.hnh_comm_sendrecv(ctx, comm, None, 0, 0, bufs[1].value, 128, 0, K.STREAM_COMM) == 0
        return lib.hnh_comm_group_end(ctx)
    assert in_threads([rank0, rank1]) == [0, 0]
    assert (read(1, 256, 16) == 1.0).all() and (read(1, 0, 16) == 2.0).all()
    assert checker()[0] == 0
    # sizes of a matching pair differ: refused (undefined on the GPU)
    got = in_threads([lambda: lib.hnh_comm_sendrecv(pairs[0][0], pairs[0][1], bufs[0].value, 128, 1, None, 0, 1, K.STREAM_COMM),
                      lambda: lib.hnh_comm_sendrecv(pairs[1][0], pairs[1][1], None, 0, 0, bufs[1].value, 64, 0, K.STREAM_COMM)])
    assert got[1] != 0 and b"sizes of a matching pair differ" in lib.hnh_last_error(pairs[1][0])
    # a receive whose send is never issued: fails after the time limit instead of hanging
    assert lib.hnh_comm_sendrecv(pairs[1][0], pairs[1][1], None, 0, 0, bufs[1].value, 64, 0, K.STREAM_COMM) != 0
    assert b"never issued the matching ncclSend" in lib.hnh_last_error(pairs[1][0])
    # a collective only one rank calls: the same
    assert lib.hnh_comm_allreduce_f64(pairs[0][0], pairs[0][1], bufs[0], bufs[0], 4, K.STREAM_COMM) != 0
    assert b"not called by every rank" in lib.hnh_last_error(pairs[0][0])
    checker()
    # an id that did not come from hnh_comm_unique_id of this process: no communicator (and no waiting for peers that cannot exist)
    ctx, comm = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.hnh_ctx_create(0, ctypes.byref(ctx)) == 0
    assert lib.hnh_comm_init(ctx, 2, 0, bytes(128), ctypes.byref(comm)) != 0
    assert lib.hnh_ctx_destroy(ctx) == 0


def test_a_kernel_that_touches_a_buffer_in_flight_is_a_race(checker):
    """The checker's view of a transfer: the send reads its buffer ON THE SENDER'S STREAM — a kernel of another stream that overwrites
    the buffer without waiting for that stream is reported; the receiver's kernel behind its receive is not."""
    from distributed_sddmm_amd import _kernels as K
    lib, pairs = raw_comms(2)
    bufs = []
    for ctx, _ in pairs:
        p = ctypes.c_void_p()
        assert lib.hnh_malloc(ctx, 1024, ctypes.byref(p)) == 0
        bufs.append(p)
    assert lib.hnh_fill_f64(pairs[0][0], bufs[0], 128, 1.0, K.STREAM_COMM) == 0

    def rank0():
        ctx, comm = pairs[0]
        assert lib.hnh_comm_sendrecv(ctx, comm, bufs[0], 1024, 1, None, 0, 1, K.STREAM_COMM) == 0
        return lib.hnh_fill_f64(ctx, bufs[0], 128, 2.0, K.STREAM_COMPUTE)     # the compute stream does not know about the send

    def rank1():
        ctx, comm = pairs[1]
        assert lib.hnh_comm_sendrecv(ctx, comm, None, 0, 0, bufs[1], 1024, 0, K.STREAM_COMM) == 0
        return lib.hnh_axpy_f64(ctx, bufs[1], bufs[1], 1.0, 128, K.STREAM_COMM)   # same stream as the receive: ordered
    assert in_threads([rank0, rank1]) == [0, 0]
    n, text = checker()
    # (two reports: against the send, and against the fill on the communication stream that the send was ordered behind)
    assert n == 2 and "RACE hnh_fill_f64 (write, context" in text and "vs ncclSend (read" in text and "ncclRecv" not in text, text


def test_random_configurations_over_the_rccl_emulation(checker, monkeypatch):
    """The fuzz generator of tests/test_fuzz_cpu.py (schedule x grid incl. remainders x sizes incl. M < p x every host switch, ALS twice)
    with its ranks on RcclWorld: a fixed sample here; a one-off exploration of 4 x 500 draws (1 408 valid configurations) was clean."""
    import random
    import test_fuzz_cpu as F
    monkeypatch.setenv("HNH_ORACLE_COMM_WAIT_S", "60")
    monkeypatch.setattr(H, "run_spmd", over_rccl)
    saved = {k: os.environ.get(k) for k in F.KNOBS}
    rng, done = random.Random(21), 0
    try:
        for it in range(14):
            done += 1 if F.one(rng, it) else 0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert done >= 6
    n, text = checker()
    assert n == 0, text


@pytest.mark.parametrize("nranks,configs", [(2, ALL_2), (4, ALL_4)], ids=["2", "4"])
def test_the_gpu_tests_worker_as_processes_over_the_rccl_emulation(nranks, configs):
    """tests/rccl_worker.py — the worker of the GPU test: one PROCESS per rank, the unique id handed round through torch.distributed,
    RcclWorld, the preflight, every configuration against the golden vectors — with the kernel test double in place of the HIP library:
    the emulation's segment lives in shared memory and the bytes move with process_vm_readv."""
    import subprocess
    import sys
    from test_gloo_world import ROOT, free_port
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo",
                   OMP_NUM_THREADS="2", HNH_RCCL_WORKER_DOUBLE="1", HNH_ORACLE_COMM_WAIT_S="120")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), "er8_r16", configs], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "RCCL_OK" in outs[0], outs[0][-3000:]
