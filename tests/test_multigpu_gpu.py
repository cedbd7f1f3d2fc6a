"""Several ranks as several PROCESSES.  The RCCL tests need at least two visible MI355X and skip on a one-GPU box; the
ipc-pull transport (receivers copy out of their peers' mapped buffers) also works between processes that SHARE one GPU, so
its schedule tests and the `bench.py --gpus N` runs below exercise the cross-process device-to-device path on any box."""
import json
import os
import subprocess
import sys

import pytest

from test_gloo_world import ROOT, free_port
from bench_line import read_line

pytestmark = pytest.mark.gpu


def gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def launch(nranks, case, configs, timeout=600):
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GLOO_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), case, configs], env=env,
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


ALL_2 = ("15d_fusion1:1:mesh:4;15d_fusion2:1:mesh:4;15d_fusion2:1:mesh:2;15d_fusion2:1:relay:1;15d_fusion2:2:mesh:4;15d_sparse:1:mesh:4;"
         "15d_sparse:2:mesh:4;25d_dense_replicate:2:mesh:4;25d_sparse_replicate:2:mesh:4;als@15d_fusion2:1:mesh:4;als@15d_sparse:1:mesh:4")
ALL_4 = ("15d_fusion2:1:mesh:4;15d_fusion2:1:relay:1;15d_fusion2:2:mesh:2;15d_fusion1:2:mesh:4;15d_sparse:1:mesh:4;25d_dense_replicate:1:mesh:4;"
         "25d_sparse_replicate:1:mesh:4;als@15d_fusion2:1:mesh:4;als@25d_dense_replicate:1:mesh:4")
ALL_8 = ("15d_fusion2:1:mesh:4;15d_fusion2:1:mesh:8;15d_fusion2:1:relay:1;15d_fusion2:2:mesh:4;15d_fusion2:4:mesh:2;15d_fusion1:1:mesh:4;"
         "15d_sparse:2:mesh:4;25d_dense_replicate:2:mesh:4;25d_sparse_replicate:2:mesh:4;als@15d_fusion2:1:mesh:4")


@pytest.mark.parametrize("nranks,configs", [(2, ALL_2), (4, ALL_4), (8, ALL_8)], ids=["2", "4", "8"])
def test_schedules_over_rccl(nranks, configs):
    """All five schedules (relay ring, chunked mesh fetch, replication collectives as explicit-peer groups, ALS with the held
    operand) on `nranks` GPUs over RCCL, element-wise against the reference's golden vectors, after the transport preflight."""
    if gpus() < nranks:
        pytest.skip("needs %d GPUs, this box has %d" % (nranks, gpus()))
    for case in ("er8_r16", "ragged_r8"):
        procs, outs = launch(nranks, case, configs)
        assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
        assert "RCCL_OK" in outs[0], outs[0][-3000:]


# Copy-engine pulls between 8 (4) processes that share ONE GPU cost ~100 ms per cross-process dependency (the hardware scheduler
# time-slices the processes): that variant runs three configurations (dense ring, sparse ring, 2.5D) on one case at 8 ranks and six at
# 4; the pull-kernel variant and the 2-rank runs cover every configuration on both cases.
ENGINE_8 = "15d_fusion2:1:mesh:4;15d_sparse:2:mesh:4;25d_dense_replicate:2:mesh:4"
ENGINE_4 = "15d_fusion2:1:mesh:4;15d_fusion2:2:mesh:2;15d_fusion1:2:mesh:4;15d_sparse:1:mesh:4;25d_sparse_replicate:1:mesh:4;als@15d_fusion2:1:mesh:4"


@pytest.mark.parametrize("nranks,configs,pull,flags,cases",
                         [(2, ALL_2, "engine", "memop", ("er8_r16", "ragged_r8")), (2, ALL_2, "kernel", "kernel", ("er8_r16", "ragged_r8")),
                          (4, ENGINE_4, "engine", "memop", ("er8_r16",)), (4, ALL_4, "kernel", "memop", ("er8_r16", "ragged_r8")),
                          (8, ENGINE_8, "engine", "memop", ("ragged_r8",)), (8, ALL_8, "kernel", "memop", ("er8_r16", "ragged_r8"))],
                         ids=["2-engine-memop", "2-kernel-flagkernels", "4-engine-memop", "4-kernel-memop", "8-engine-memop", "8-kernel-memop"])
def test_schedules_over_ipc(nranks, configs, pull, flags, cases):
    """The ipc-pull transport ACROSS PROCESSES on whatever GPUs are here (one is enough: the processes share it): all five
    schedules (relay ring, chunked mesh fetch, replication collectives as explicit-peer groups, ALS with the held operand) on
    the HIP kernels, every transfer a device-to-device copy out of the peer's mapped buffer ordered by stream flag words —
    element-wise against the reference's golden vectors, after the transport preflight.  Both ways of pulling (copy engines
    on forked streams / one gather-copy kernel) and both ways of signalling (stream memory operations / flag kernels)."""
    if gpus() < 1:
        pytest.skip("needs a GPU")
    from test_ipc_world_cpu import launch_ipc
    for case in cases:
        procs, outs = launch_ipc(nranks, case, configs, backend="hip", timeout=900, extra_env={"HNH_IPC_PULL": pull, "HNH_IPC_FLAGS": flags})
        assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
        assert "IPC_OK" in outs[0], outs[0][-3000:]


def run_bench(n, *extra, timeout=900):
    env = dict(os.environ, GLOO_SOCKET_IFNAME="lo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "HNH_BENCH_WORKER"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--logm", "16",
                           "--edge-factor", "32", "--no-cpu-baseline", *extra], env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [2, 8])
def test_bench_self_launch_on_real_gpus(n):
    """`python bench.py --gpus N` as typed: spawns its N workers, tries RCCL and the ipc-pull transport in child processes, runs
    the preflight, the measured search over transport, replication factor and route, the timed steps, and the row/column-keyed
    result check — which fails if any block travels to the wrong place."""
    if gpus() < n:
        pytest.skip("needs %d GPUs, this box has %d" % (n, gpus()))
    res = run_bench(n)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = read_line(lines[0])
    assert out["n_gpus"] == n and out["backend"] == "hip-gfx950" and out["check"]["ok"] and out["check"]["rows_checked"] == 1 << 16
    assert len(out["preflight"]["primitives_ok"]) == 9 and out["config"]["transport"] in ("rccl", "ipc-pull")
    assert all(v.startswith("ok") for v in out["config"]["transport_trials"].values()), out["config"]["transport_trials"]
    tuned = out["config"]["route_tuning_ms_per_step"]
    assert {k.split()[0] for k in tuned} >= {"c=1", "c=2"} and {k.rsplit("[", 1)[1] for k in tuned} >= {"rccl]", "ipc]", "ipc-kernel]"}


@pytest.mark.parametrize("n", [2, 4] if os.environ.get("HNH_LONG_TESTS") == "1" else [2])  # (4 processes: another 30 s; HNH_LONG_TESTS=1)
def test_bench_processes_share_one_gpu(n):
    """`python bench.py --gpus N` typed as is on a box with ONE GPU: the N worker processes share it.  RCCL refuses that (its
    child-process trial says so and the run goes on without it); the ipc-pull transport moves every block device to device
    between the processes — both of its variants are measured — and the run ends with a checked line: HIP kernels,
    device-resident set-up, preflight, the search over transport variant, replication factor and route, timed steps, the
    row/column-keyed result check."""
    if gpus() != 1:
        pytest.skip("this is the one-GPU behaviour")
    env = dict(os.environ, GLOO_SOCKET_IFNAME="lo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "HNH_BENCH_WORKER"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--logm", "14",
                          "--edge-factor", "16", "--r", "32", "--no-cpu-baseline", "--probe-timeout", "240"], env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = read_line(lines[0])
    assert "incomplete" not in out, (out["incomplete"], out["config"].get("route_tuning_failures"), res.stderr[-3000:])
    assert out["n_gpus"] == n and out["backend"] == "hip-gfx950" and out["config"]["transport"] == "ipc-pull"
    assert out["check"]["ok"] and out["check"]["rows_checked"] == 1 << 14 and out["check"]["rel_err"] <= 1e-11
    trials = out["config"]["transport_trials"]
    assert trials["ipc"].startswith("ok") and not trials["rccl"].startswith("ok")
    tuned = out["config"]["route_tuning_ms_per_step"]
    assert {k.split()[0] for k in tuned} == {"c=%d" % c for c in (1, 2, 4) if n % c == 0}
    assert {k.rsplit("[", 1)[1] for k in tuned} == {"ipc]", "ipc-kernel]"} and all(v is not None for v in tuned.values())
    assert len(out["preflight"]["primitives_ok"]) == 9 and out["preflight"]["transports"] == ["ipc", "ipc-kernel"]


def test_bench_under_torch_distributed_run_on_one_gpu():
    """The driver's way of starting a multi-GPU run — `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` — with two
    workers on a one-GPU box: the elastic agent's environment must not leak into the transport trials' own rendezvous, and rank 0
    prints the one JSON line (over ipc-pull here)."""
    if gpus() != 1:
        pytest.skip("this is the one-GPU behaviour")
    env = dict(os.environ, GLOO_SOCKET_IFNAME="lo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "HNH_BENCH_WORKER"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--logm", "14",
                          "--edge-factor", "16", "--no-cpu-baseline", "--probe-timeout", "240", "--chunks", "2"], env=env,  # (no "--r": the launcher's own
                         # option parser takes it for an abbreviation of its --rdzv-* / --role / --run-path options)
                         capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = read_line(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["transport"] == "ipc-pull" and out["check"]["ok"] and "incomplete" not in out
    assert out["config"]["transport_trials"]["ipc"].startswith("ok (") and "trial failed" not in out["config"]["transport_trials"]["ipc"]


def test_bench_rccl_only_on_one_gpu_ends_with_an_error_line():
    """One GPU, two ranks, --transport rccl: RCCL refuses the second rank on the same device.  The run must END (no hang) with
    one JSON line that says no transport is usable and why, and a non-zero exit code."""
    if gpus() != 1:
        pytest.skip("this is the one-GPU behaviour")
    res = run_bench(2, "--transport", "rccl", "--watchdog", "60", "--launch-timeout", "400", "--probe-timeout", "120", timeout=600)
    assert res.returncode != 0
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = read_line(lines[0])
    assert out["value"] is None and "error" in out and out["n_gpus"] == 2
    assert any("transport" in ph for ph in out["phases"].values()), out
    assert "no usable device-to-device transport" in res.stderr


def test_cpp_verify_across_processes_share_one_gpu(tmp_path):
    """examples/verify (the reference's scratch.cpp check against the class headers) as 2 and 4 PROCESSES over the ipc-pull transport
    on the HIP library, all on device 0 (HNH_TRANSPORT=ipc: the C++ drivers' way of running several ranks without RCCL): same
    fingerprints as the oracle's (the CPU twin in tests/test_tools_cpu.py runs every schedule family)."""
    import time
    import numpy as np
    import hnh_testlib as T
    from oracle import oracle as O
    if gpus() < 1:
        pytest.skip("needs a GPU")
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "verify"], check=True, capture_output=True, timeout=600)
    mtx = str(tmp_path / "g.mtx")
    mrows, mcols, _ = T.write_symmetric_mtx_with_duplicates(mtx, 512, 4)
    want = np.array(O.fingerprints(mrows, mcols, 512, 512, 32))
    for n, c, alg in ((2, 1, "15d_fusion2"), (4, 1, "25d_dense_replicate")):
        session = "g%d_%x" % (os.getpid(), time.time_ns())
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), LOCAL_RANK=str(r), HNH_DEVICE="0", HNH_TRANSPORT="ipc", HNH_IPC_SESSION=session,
                       HNH_IPC_WAIT_S="120", HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="16", OMP_NUM_THREADS="4")
            procs.append(subprocess.Popen([os.path.join(ROOT, "examples", "verify"), mtx, alg, "32", str(c)], env=env, stdout=subprocess.PIPE,
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
