"""N > 1 across PROCESSES on CPU: world_size 2 and 4 with torch.distributed's gloo backend carrying the
transport callbacks of the host layer (rendezvous on 127.0.0.1)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(nranks, case, configs, timeout=300):
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gloo_worker.py"), case, configs],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return procs, outs


@pytest.mark.parametrize("nranks,configs", [
    (2, "15d_fusion1:1;15d_fusion2:1;15d_fusion2:2;15d_sparse:1;15d_sparse:2;25d_dense_replicate:2;25d_sparse_replicate:2"),
    (4, "15d_fusion2:2;15d_sparse:1;25d_dense_replicate:1;25d_sparse_replicate:1"),
])
def test_schedules_over_gloo(nranks, configs):
    procs, outs = launch(nranks, "er8_r16", configs)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "GLOO_OK" in outs[0], outs[0][-2000:]


def test_cfg1_shape_sparse_shift_world2():
    """BASELINE config 1 plumbing: 1.5D sparse shift, world = 2 on CPU (here on the small fixture, R = 16)."""
    procs, outs = launch(2, "ragged_r8", "15d_sparse:1")
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "GLOO_OK" in outs[0], outs[0][-2000:]


def test_als_and_gat_over_gloo():
    """ALS-CG (and, on 4 processes, the GAT forward pass) over the multi-process transport against the golden factors produced by the reference's own ALS: the R-split
    all-reduce (1.5D sparse shift, 2.5D), the fused epilogue path with the hold hint (1.5D dense, local kernel fusion) and
    the chunked mesh fetch, each on 2 or 4 processes."""
    procs, outs = launch(2, "er8_r16", "als@15d_fusion2:1;als@15d_sparse:1;als@25d_sparse_replicate:2")
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "GLOO_OK" in outs[0], outs[0][-2000:]
    procs, outs = launch(4, "er8_r16", "als@15d_fusion2:1;als@25d_dense_replicate:1;gat@15d_fusion2:1;gat@15d_fusion1:2")
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "GLOO_OK" in outs[0], outs[0][-2000:]
