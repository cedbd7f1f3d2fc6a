"""The ipc-pull transport across PROCESSES on CPU: IpcWorld's shared-memory control plane (rendezvous, barrier, host
all-gather, message mailboxes, flag words), its group logic and every schedule on top of it, with the oracle's C test double
serving the kernel ABI — there `process_vm_readv` stands in for the mapped peer memory, so the bytes really cross process
boundaries.  The GPU twin (tests/test_multigpu_gpu.py::test_schedules_over_ipc) runs the same worker on the HIP library."""
import os
import subprocess
import sys
import tempfile
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch_ipc(nranks, case, configs, backend="oracle", timeout=600, extra_env=None):
    session = "t%d_%x" % (os.getpid(), time.time_ns())
    with tempfile.TemporaryDirectory(prefix="hnh_ipc_") as outdir:
        procs = []
        for r in range(nranks):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(nranks), OMP_NUM_THREADS="2", HNH_TEST_BACKEND=backend, HNH_IPC_WAIT_S="120",
                       HSA_ENABLE_IPC_MODE_LEGACY="0")
            env.update(extra_env or {})
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_worker.py"), session, outdir, case, configs], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = []
        try:
            for p in procs:
                outs.append(p.communicate(timeout=timeout)[0])
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    return procs, outs


def can_read_peer_memory():
    """The test double's pull is process_vm_readv: needs ptrace permission between the test's own processes."""
    import ctypes
    libc = ctypes.CDLL(None, use_errno=True)
    r, w = os.pipe()
    buf = ctypes.create_string_buffer(b"x" * 8, 8)
    pid = os.fork()
    if pid == 0:
        os.read(r, 1)
        os._exit(0)

    class IoVec(ctypes.Structure):
        _fields_ = [("base", ctypes.c_void_p), ("len", ctypes.c_size_t)]
    out = ctypes.create_string_buffer(8)
    loc, rem = IoVec(ctypes.addressof(out), 8), IoVec(ctypes.addressof(buf), 8)
    got = libc.process_vm_readv(pid, ctypes.byref(loc), 1, ctypes.byref(rem), 1, 0)
    os.write(w, b"x")
    os.waitpid(pid, 0)
    return got == 8


needs_peer_reads = pytest.mark.skipif(not can_read_peer_memory(), reason="process_vm_readv between own processes is not permitted here")

ALL_2 = ("15d_fusion1:1:mesh:4;15d_fusion2:1:mesh:4;15d_fusion2:1:mesh:2;15d_fusion2:1:relay:1;15d_fusion2:2:mesh:4;15d_sparse:1:mesh:4;"
         "15d_sparse:2:mesh:4;25d_dense_replicate:2:mesh:4;25d_sparse_replicate:2:mesh:4;als@15d_fusion2:1:mesh:4;als@15d_sparse:1:mesh:4")
ALL_4 = ("15d_fusion2:1:mesh:4;15d_fusion2:1:relay:1;15d_fusion2:2:mesh:2;15d_fusion1:2:mesh:4;15d_sparse:1:mesh:4;25d_dense_replicate:1:mesh:4;"
         "25d_sparse_replicate:1:mesh:4;als@15d_fusion2:1:mesh:4;als@25d_dense_replicate:1:mesh:4")
ALL_8 = ("15d_fusion2:1:mesh:4;15d_fusion2:1:mesh:8;15d_fusion2:1:relay:1;15d_fusion2:2:mesh:4;15d_fusion2:4:mesh:2;15d_fusion1:1:mesh:4;"
         "15d_sparse:2:mesh:4;25d_dense_replicate:2:mesh:4;25d_sparse_replicate:2:mesh:4;als@15d_fusion2:1:mesh:4")


@needs_peer_reads
@pytest.mark.parametrize("nranks,configs", [(2, ALL_2), (4, ALL_4)])
def test_schedules_over_ipc_processes_cpu(nranks, configs):
    procs, outs = launch_ipc(nranks, "er8_r16", configs)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "IPC_OK" in outs[0], outs[0][-3000:]


@needs_peer_reads
def test_ragged_case_and_eight_ranks_over_ipc_cpu():
    procs, outs = launch_ipc(8, "ragged_r8", "15d_fusion2:1:mesh:4;15d_fusion2:2:mesh:2;25d_dense_replicate:2:mesh:4;15d_sparse:1:mesh:4")
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "IPC_OK" in outs[0], outs[0][-3000:]


@needs_peer_reads
def test_a_missing_rank_ends_the_others_with_an_error():
    """Rank 1 of 2 never shows up: rank 0 gives up after HNH_IPC_WAIT_S with a message instead of hanging."""
    session = "t%d_%x" % (os.getpid(), time.time_ns())
    with tempfile.TemporaryDirectory(prefix="hnh_ipc_") as outdir:
        env = dict(os.environ, RANK="0", WORLD_SIZE="2", HNH_TEST_BACKEND="oracle", HNH_IPC_WAIT_S="3")
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ipc_worker.py"), session, outdir, "er8_r16", "15d_fusion2:1:mesh:4"], env=env,
                             capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "every rank to attach" in (res.stdout + res.stderr)


@needs_peer_reads
def test_a_full_handle_table_is_emptied_between_groups_never_inside_one():
    """HNH_IPC_MAX_OPENED=2 on 4 ranks: every group (3 receives) would overflow the table of mapped peer allocations.  The table is
    emptied BEFORE a group is issued (IpcWorld::make_room), never while the group's earlier receives already hold `base + offset`
    of a mapping (round 4's advisor finding: the eviction inside open_peer unmapped sources of the group in flight) — the schedules'
    results stay the golden ones."""
    procs, outs = launch_ipc(4, "er8_r16", "15d_fusion2:1:mesh:4;15d_fusion2:2:mesh:2;25d_sparse_replicate:1:mesh:4;als@15d_fusion2:1:mesh:4",
                             extra_env={"HNH_IPC_MAX_OPENED": "2"})
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "IPC_OK" in outs[0], outs[0][-3000:]


@needs_peer_reads
def test_a_stale_segment_of_an_earlier_launch_is_not_joined():
    """A shared-memory segment left under the session's name by a launch that was killed (valid size and magic, creator gone, a full
    world attached): the ranks of the next launch must not take it for theirs — rank 0 replaces it, the others drop the stale mapping
    and look again — instead of waiting out the time limit."""
    import struct
    session = "t%d_%x" % (os.getpid(), time.time_ns())
    pid = os.fork()
    if pid == 0:
        os._exit(0)
    os.waitpid(pid, 0)  # `pid` now names no process
    path = "/dev/shm/hnh_ipc_" + session
    with open(path, "wb") as f:
        f.truncate(64 << 20)  # (sparse; larger than the segment)
        f.seek(0)
        f.write(struct.pack("<QiiqI", 0x686e685f69706331, 2, 0, pid, 2))  # magic, nranks, pad, creator_pid, attached = 2
    with tempfile.TemporaryDirectory(prefix="hnh_ipc_") as outdir:
        procs = []
        for r in (1, 0):  # rank 1 first: it finds the leftover before rank 0 replaces it
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", OMP_NUM_THREADS="2", HNH_TEST_BACKEND="oracle", HNH_IPC_WAIT_S="60")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_worker.py"), session, outdir, "er8_r16", "15d_fusion2:1:mesh:4"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
            time.sleep(1.0)
        outs = [p.communicate(timeout=300)[0] for p in procs]
    assert not os.path.exists(path)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "IPC_OK" in outs[1], outs[1][-3000:]
