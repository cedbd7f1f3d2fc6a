"""The benchmark driver's control flow (world bootstrap, strong-scaling setup, timed loop, max-over-ranks,
roofline bookkeeping, JSON contract) exercised on CPU ranks over gloo — bench.run() itself, with only the
transport/kernels swapped by the test.  The numbers are meaningless here; the contract is what is checked."""
import json
import os
import subprocess
import sys
import time

import pytest

from test_gloo_world import ROOT, free_port

from bench_line import LINE_LIMIT, read_line  # noqa: E402,F401


@pytest.mark.parametrize("nranks,alg,c,ring", [(1, "15d_fusion2", 1, None), (2, "15d_fusion2", 1, None), (4, "15d_fusion2", 1, None),
                                                (4, "15d_fusion2", 2, None), (2, "15d_fusion1", 1, None), (4, "15d_fusion2", 1, "relay"), (2, "15d_fusion2", 1, "mesh"),
                                                (4, "15d_fusion2", 0, None), (2, "15d_fusion2", 0, "relay")])
def test_bench_contract(nranks, alg, c, ring):
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", GLOO_SOCKET_IFNAME="lo", BENCH_RING_MODE=ring or "")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_worker.py"), alg, str(c)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    line = [ln for ln in outs[0].splitlines() if ln.startswith("BENCH_JSON ")]
    assert len(line) == 1, outs[0][-1500:]
    out = json.loads(line[0][len("BENCH_JSON "):])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in out
    assert out["n_gpus"] == nranks and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "strong"
    assert out["dtype"] == "f64" and out["data"] == "synthetic" and out["vs_baseline"] is None and out["higher_is_better"] is True
    assert out["value"] > 0 and out["config"]["nnz"] > 0 and "workload" in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert rf["launches_per_step"] >= 1 and rf["algorithmic_bytes_per_launch"] > 0
    assert out["backend"] == "oracle-cpu-test-double" and "cpu_baseline" not in out
    # the closed-form result check at the run's own size, on every rank's rows
    chk = out["check"]
    assert chk["ok"] and chk["rel_err"] <= 1e-11 and chk["rows_checked"] == 1 << 10
    assert chk["nnz_operator"] == chk["nnz_host_generator"] == out["config"]["nnz"]
    assert "keyed by global row and column" in chk["what"]  # a mis-routed block changes this answer (constant operands would not)
    if nranks > 1:  # every transport primitive ran before the timed region, and what was measured is recorded
        assert len(out["preflight"]["primitives_ok"]) == 9
        assert out["config"]["transport"] == "callback"  # (the test transport; "rccl" on GPUs)
        cs = [c] if c else [k for k in (1, 2, 4) if nranks % k == 0]
        want = set()
        for k in cs:  # candidates: replication factor x (mesh fetch in 2 / 4 (/ 8) chunks, relay ring); --ring-mode fixes the route
            if nranks // k == 1:
                want.add("c=%d replication only" % k)
                continue
            if ring != "relay":
                want |= {"c=%d mesh/%d chunks" % (k, q) for q in ((2, 3, 4) if k == 1 else (2, 4))}
                want |= {"c=%d mesh/heights %s" % (k, h) for h in (("1,2,2,2,1,1", "3,4,4,3,2,1,1") if k == 1 else ("1,2,2,2,1,1",))}
                if k == 1:  # the default shape with one windowed pass per chunk (HNH_WINDOW_MERGE=0) against the adaptive windows
                    want.add("c=1 mesh/heights 1,2,2,2,1,1/one-pass-per-chunk")
            if ring != "mesh":
                want.add("c=%d relay ring" % k)
            if ring is None:  # the schedule's other fusion strategy joins the search when nothing fixes the route
                want.add("c=%d 15d_fusion1 (replication reuse: SDDMM + SpMM, mesh fetch + mesh reduce-scatter)" % k)
        tuned = alg == "15d_fusion2" and len(want) > 1
        assert ("route_tuning_ms_per_step" in out["config"]) == tuned
        if tuned:
            # what is reported is a COMPLETE measurement (timed steps + check) of either the fastest candidate of the search or the
            # default route measured before it — whichever full measurement was faster
            t = {k.rsplit(" [", 1)[0]: v for k, v in out["config"]["route_tuning_ms_per_step"].items()}  # (names end in " [transport]")
            assert all(k.endswith(" [default]") for k in out["config"]["route_tuning_ms_per_step"])
            assert set(t) == want

            def matches(name):
                bc, broute = name.split(" ", 1)
                if out["config"]["c"] != int(bc[2:]):
                    return False
                if broute.startswith("15d_fusion1"):
                    return out["config"]["algorithm"] == "15d_fusion1" and out["config"]["ring_mode"].startswith("mesh fetch + mesh reduce")
                return (out["config"]["algorithm"] == "15d_fusion2" and
                        out["config"]["ring_mode"] == {"relay ring": "relay", "replication only": None}.get(broute, "mesh") and
                        out["config"]["mesh_chunks"] == (broute.split("/", 1)[1].split(" ", 1)[-1 if "heights" in broute else 0] if broute.startswith("mesh") else None))

            c0 = c or 1
            default = ("c=%d replication only" % c0) if nranks // c0 == 1 else (("c=%d relay ring" % c0) if ring == "relay" else "c=%d mesh/heights 1,2,2,2,1,1" % c0)
            assert matches(min(t, key=t.get)) or matches(default)
            if not c and nranks == 4:
                assert any(k.startswith("c=2") for k in t) and any(k.startswith("c=4") for k in t)
        else:
            assert out["config"]["ring_mode"] == (None if nranks // (c or 1) == 1 else (ring or "mesh"))
    else:
        assert "preflight" not in out


def product_launch(nranks, extra_env=None, probe_timeout="120", extra_args=()):
    """`python bench.py --gpus N` as typed, its workers = tests/bench_product_worker.py: bench.run()'s PRODUCT branch for several GPUs
    with the kernel library, the device selection and the trial script replaced from outside (see the worker)."""
    env = dict(os.environ, OMP_NUM_THREADS="2", GLOO_SOCKET_IFNAME="lo", HNH_ORACLE_COMM_WAIT_S="120", HNH_IPC_WAIT_S="120",
               HNH_BENCH_WORKER=os.path.join(ROOT, "tests", "bench_product_worker.py"))
    env.update(extra_env or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "2", "--warmup", "1", "--logm", "10",
                          "--edge-factor", "8", "--r", "16", "--no-cpu-baseline", "--probe-timeout", probe_timeout, *extra_args], env=env, capture_output=True, text=True,
                         timeout=900)
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, (res.returncode, res.stdout[-1500:], res.stderr[-1500:])
    return res, read_line(lines[0])


@pytest.mark.parametrize("nranks", [2, 4])
def test_the_multi_gpu_product_path_with_every_transport_usable(nranks):
    """What the first run on a node with several GPUs will do, for the first time anywhere (on a one-GPU box RCCL never passes its
    trial): both transports pass their child-process trials, all three variants are created in the benchmark process and run the
    preflight, the default route is measured on each of them, the search continues on the fastest, the winner is measured in full."""
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    res, out = product_launch(nranks)
    assert res.returncode == 0 and "error" not in out, (out.get("error"), res.stderr[-1500:])
    assert out["n_gpus"] == nranks and out["check"]["ok"] and out["value"] > 0
    trials = out["config"]["transport_trials"]
    assert set(trials) == {"rccl", "ipc"} and all(v.startswith("ok") for v in trials.values()), trials
    assert out["preflight"]["transports"] == ["ipc", "ipc-kernel", "rccl"] and len(out["preflight"]["primitives_ok"]) == 9
    tuned = out["config"]["route_tuning_ms_per_step"]
    default = [k for k in tuned if k.startswith("c=1 mesh/heights 1,2,2,2,1,1 [")]
    assert {k.rsplit("[", 1)[1] for k in default} == {"rccl]", "ipc]", "ipc-kernel]"}   # stage 1: the default route on every transport
    assert len(tuned) >= 10 and all(v is not None for v in tuned.values()) and "route_tuning_failures" not in out["config"], tuned
    assert out["config"]["transport"] in ("rccl", "ipc-pull")


def test_the_time_budget_ends_the_search_with_a_complete_line():
    """Candidates that take seconds each (here: 2 s of sleep per candidate) on a run with --budget-s 45: the search stops taking
    candidates while the budget still has room for the winner's full measurement, says so in config.budget_stops, and the line is a
    COMPLETE one (no "incomplete" mark, check ok, phases_s) printed well inside the budget — the launcher's limit is budget + 120."""
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    t0 = time.time()
    res, out = product_launch(2, {"BENCH_PRODUCT_SLOW": "2.0"}, extra_args=("--budget-s", "45"))
    took = time.time() - t0
    assert res.returncode == 0 and "error" not in out and "incomplete" not in out, (out.get("error"), out.get("incomplete"), res.stderr[-1500:])
    assert out["check"]["ok"] and out["value"] > 0 and out["config"]["budget_s"] == 45.0
    stops = out["config"]["budget_stops"]
    assert any("no room for another candidate" in s for s in stops), stops
    tuned = out["config"]["route_tuning_ms_per_step"]
    assert 1 <= len(tuned) < 10, tuned  # (an unbounded search measures 13 or more, see the test above)
    ph = out["phases_s"]
    assert ph["tuning"] >= 2.0 and ph["total"] <= 45.0 and took < 60.0, (ph, took)
    assert {"start_up", "transport_trials", "bring_up", "first_measurement", "tuning", "total"} <= set(ph), ph


def test_sigterm_to_the_launcher_prints_the_line_in_hand():
    """The driver's time limit ends `python bench.py --gpus 2` with SIGTERM in the middle of the route search: the launcher passes it to
    rank 0 first, which prints the complete measurement it already holds, marked "incomplete"; the launcher forwards that ONE line
    and exits 0."""
    import signal
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    env = dict(os.environ, OMP_NUM_THREADS="2", GLOO_SOCKET_IFNAME="lo", HNH_ORACLE_COMM_WAIT_S="120", HNH_IPC_WAIT_S="120",
               HNH_BENCH_WORKER=os.path.join(ROOT, "tests", "bench_product_worker.py"), BENCH_PRODUCT_SLOW="4.0", HNH_BENCH_ANNOUNCE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--logm", "10", "--edge-factor", "8",
                          "--r", "16", "--no-cpu-baseline", "--probe-timeout", "120"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    for ln in p.stderr:  # trials + bring-up + the first measurement take about 5 s here; then the search (4 s per candidate) is under way
        if "a complete measurement is in hand" in ln:
            break
    assert p.poll() is None and time.time() - t0 < 120
    time.sleep(3.0)
    p.send_signal(signal.SIGTERM)
    stdout, stderr = p.communicate(timeout=120)
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, stdout[-1500:], stderr[-1500:])
    out = read_line(lines[0])
    assert out["value"] > 0 and out["check"]["ok"] and "signal 15 sent to the launcher" in out["incomplete"], out.get("incomplete")
    assert out["phases_s"]["total"] > 3.0 and "phases" in out and "exit_codes" in out


def test_a_multi_gpu_line_carries_the_whole_step_roofline_and_the_cpu_baseline(tmp_path):
    """N = 4: `roofline.frac` is SURVEY 8(d)'s whole-step fraction — total B_fused / ms_per_step / (4 x 8 TB/s) — with the kernel-level
    figure kept as `frac_kernel` and the difference of the two times as `exposed_comm_ms`; `cpu_baseline` rides on the line too: the
    record the N = 1 run left on this host is quoted (here: a seeded record), without one the bounded sample leg runs after the line
    is in hand."""
    import socket
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    key = "er10_ef8_r16_s10_t1"
    with open(tmp_path / ("hnh_cpu_baseline_%s_%s.json" % (socket.gethostname(), key)), "w") as f:
        json.dump({"value": 1.25e9, "unit": "nnz*R/s", "cores": 32, "kind": "reference", "sample": "seeded by the test", "_stamp": time.time() - 100.0}, f)
    env = dict(os.environ, OMP_NUM_THREADS="2", GLOO_SOCKET_IFNAME="lo", HNH_ORACLE_COMM_WAIT_S="120", HNH_IPC_WAIT_S="120", HNH_BENCH_CACHE_DIR=str(tmp_path),
               HNH_BENCH_WORKER=os.path.join(ROOT, "tests", "bench_product_worker.py"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)

    def launch(*extra):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--logm", "10", "--edge-factor", "8",
                              "--r", "16", "--cpu-logm", "10", "--cpu-trials", "1", "--probe-timeout", "120", "--no-tune", *extra], env=env,
                             capture_output=True, text=True, timeout=900)
        lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
        assert res.returncode == 0 and len(lines) == 1, (res.returncode, res.stdout[-1500:], res.stderr[-1500:])
        return read_line(lines[0])
    out = launch()
    roof = out["roofline"]
    total = out["config"]["nnz"] * (8 * 16 + 24) + 16 * 16 * out["config"]["M"]
    assert roof["algorithmic_bytes_per_step_all_gpus"] == total
    assert abs(roof["frac_step"] - total / (out["ms_per_step"] * 1e-3) / (4 * 8.0e12)) <= 1e-9 * roof["frac_step"]
    assert roof["frac"] == roof["frac_step"] and "frac_step" in roof["frac_is"] and roof["frac_kernel"] > 0
    assert abs(roof["exposed_comm_ms"] - (out["ms_per_step"] - roof["kernel_ms_per_step"])) < 1e-9
    cb = out["cpu_baseline"]
    assert cb["value"] == 1.25e9 and cb["kind"] == "reference" and "earlier run of bench.py on this host" in cb["cached"], cb
    assert out["phases_s"]["cpu_baseline"] >= 0.0
    # no record on this host: the bounded sample leg of the compiled reference runs, with the line in hand
    os.remove(tmp_path / ("hnh_cpu_baseline_%s_%s.json" % (socket.gethostname(), key)))
    from oracle import refrun as RR
    if RR.available():
        cb = launch()["cpu_baseline"]
        assert cb["kind"] == "reference" and cb["value"] > 0 and "cached" not in cb and cb["cores"] >= 1, cb


def test_the_multi_gpu_product_path_when_rccl_cannot_be_created():
    """RCCL fails in its child-process trial (here: the transport constructor raises): the verdict says so, the benchmark process never
    tries it, and the run completes over the ipc-pull transport."""
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    res, out = product_launch(2, {"BENCH_PRODUCT_BREAK": "rccl"})
    assert res.returncode == 0 and out["check"]["ok"], (out.get("error"), res.stderr[-1500:])
    trials = out["config"]["transport_trials"]
    assert trials["ipc"].startswith("ok") and not trials["rccl"].startswith("ok") and "exit code" in trials["rccl"], trials
    assert out["config"]["transport"] == "ipc-pull" and out["preflight"]["transports"] == ["ipc", "ipc-kernel"]
    assert not [k for k in out["config"]["route_tuning_ms_per_step"] if k.endswith("[rccl]")]


def test_the_multi_gpu_product_path_when_a_transport_trial_hangs():
    """A transport whose child-process trial never answers (an RCCL bootstrap stuck on this node, say) is ended at --probe-timeout, recorded
    as a hang, and left alone; the hang never reaches the benchmark process."""
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    res, out = product_launch(2, {"BENCH_PRODUCT_BREAK": "rccl-hang"}, probe_timeout="6")
    assert res.returncode == 0 and out["check"]["ok"], (out.get("error"), res.stderr[-1500:])
    trials = out["config"]["transport_trials"]
    assert "no answer within 6 s (hang)" in trials["rccl"] and trials["ipc"].startswith("ok"), trials
    assert out["config"]["transport"] == "ipc-pull"


def test_the_multi_gpu_product_path_when_a_transport_fails_its_preflight_on_one_rank():
    """RCCL passes its trial, is created in the benchmark process, and then fails the preflight ON ONE RANK ONLY: the ranks agree (a
    transport that failed anywhere failed), it is marked dead, the other rank does not wait for it, and the run completes over ipc-pull."""
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    res, out = product_launch(2, {"BENCH_PRODUCT_BREAK": "rccl-preflight", "HNH_ORACLE_COMM_WAIT_S": "5"})  # (the stuck rank's transport gives up after 5 s)
    assert res.returncode == 0 and out["check"]["ok"], (out.get("error"), res.stderr[-1500:])
    trials = out["config"]["transport_trials"]
    assert trials["ipc"].startswith("ok") and trials["rccl"].startswith("preflight failed in the benchmark process"), trials
    assert out["preflight"]["transports"] == ["ipc", "ipc-kernel"] and out["config"]["transport"] == "ipc-pull"
    assert not [k for k in out["config"]["route_tuning_ms_per_step"] if k.endswith("[rccl]")]


def test_the_multi_gpu_product_path_when_a_later_transport_hangs():
    """Only one transport is brought up before the first complete measurement.  The second one hangs while it is created (one rank never
    arrives): the watchdog ends the run — with the line measured on the first transport, marked incomplete, not with an error."""
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    res, out = product_launch(2, {"BENCH_PRODUCT_BREAK": "ipc-hang-late"}, extra_args=("--watchdog", "8"))
    assert res.returncode == 0 and out["value"] > 0 and out["check"]["ok"], (out.get("error"), res.stderr[-1500:])
    assert out["config"]["transport"] == "rccl" and "transport creation (ipc)" in out["incomplete"], out.get("incomplete")


def test_the_multi_gpu_product_path_when_every_trial_fails():
    """No transport passes its child-process trial — the trial machinery itself may be what is broken: the transports are tried in the
    benchmark process after all rather than giving up without a number."""
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    res, out = product_launch(2, {"BENCH_PRODUCT_BREAK": "all-trials"})
    assert res.returncode == 0 and out["check"]["ok"] and out["value"] > 0, (out.get("error"), res.stderr[-1500:])
    assert all("trial failed" in v and "created in the benchmark process" in v for v in out["config"]["transport_trials"].values())
    assert out["preflight"]["transports"] == ["ipc", "ipc-kernel", "rccl"]


def test_the_multi_gpu_product_path_when_a_transport_dies_in_the_search():
    """The fastest transport measures the default route, then one of its candidates raises on ONE rank while the other waits inside the
    transport (which gives up after its own time limit here): the candidate is recorded with its reason, that transport is not used
    again, and the line that was measured stays."""
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    res, out = product_launch(2, {"BENCH_PRODUCT_BREAK": "rccl-dies-in-search", "HNH_ORACLE_COMM_WAIT_S": "5", "HNH_IPC_WAIT_S": "5"})
    assert res.returncode == 0 and out["check"]["ok"] and out["value"] > 0, (out.get("error"), res.stderr[-1500:])
    tuned, failed = out["config"]["route_tuning_ms_per_step"], out["config"].get("route_tuning_failures", {})
    broken = [k for k in tuned if k.startswith("c=1 mesh/4 chunks [")]
    assert len(broken) == 1 and tuned[broken[0]] is None and broken[0] in failed, (tuned, failed)
    transport = broken[0].rsplit("[", 1)[1]
    assert tuned["c=1 mesh/heights 1,2,2,2,1,1 [" + transport] is not None  # its default route had been measured before it died


def test_the_multi_gpu_product_path_under_torch_distributed_run():
    """The driver's launch line — python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    <script> --gpus N ... — around the product branch: the transport trials scrub the elastic agent's environment (their children host
    their own store), one JSON line comes out."""
    from test_ipc_world_cpu import can_read_peer_memory
    if not can_read_peer_memory():
        pytest.skip("process_vm_readv between own processes is not permitted here")
    env = dict(os.environ, OMP_NUM_THREADS="2", GLOO_SOCKET_IFNAME="lo", HNH_ORACLE_COMM_WAIT_S="120", HNH_IPC_WAIT_S="120")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                          str(free_port()), os.path.join(ROOT, "tests", "bench_product_worker.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--logm", "9",
                          "--edge-factor", "8", "--rvalue", "16", "--no-cpu-baseline", "--probe-timeout", "120"], env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, (res.returncode, res.stdout[-1000:], res.stderr[-1500:])
    out = read_line(lines[0])
    assert out["n_gpus"] == 2 and out["check"]["ok"] and out["config"]["R"] == 16 if "R" in out["config"] else out["check"]["ok"]
    assert all(v.startswith("ok") for v in out["config"]["transport_trials"].values())


def self_launch(extra_env, nranks=2, extra_args=()):
    env = dict(os.environ, OMP_NUM_THREADS="2", GLOO_SOCKET_IFNAME="lo", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "2", "--warmup", "1", "--logm", "10",
                           "--edge-factor", "8", "--r", "16", "--no-cpu-baseline", *extra_args], env=env, capture_output=True, text=True, timeout=600)


def test_bench_launches_its_own_workers():
    """`python bench.py --gpus 2` as typed (no WORLD_SIZE): bench.py starts one worker per rank itself, forwards rank 0's single
    JSON line and exits 0.  The workers here are tests/bench_worker.py = bench.run() over gloo with the kernel test double."""
    res = self_launch({"HNH_BENCH_WORKER": os.path.join(ROOT, "tests", "bench_worker.py")})
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = read_line(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["check"]["ok"] and "error" not in out
    assert {k.split()[0] for k in out["config"]["route_tuning_ms_per_step"]} == {"c=1", "c=2"}


def test_bench_self_launch_reports_a_failing_rank():
    """A worker that fails: still ONE JSON line, with "error", the rank, the phase it was in and every exit code; non-zero exit."""
    res = self_launch({"HNH_BENCH_WORKER": os.path.join(ROOT, "tests", "bench_worker.py"), "BENCH_WORKER_FAIL_RANK": "1"}, extra_args=("--no-tune",))
    assert res.returncode == 7, (res.returncode, res.stderr[-1500:])
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = read_line(lines[0])
    assert out["value"] is None and out["failed_rank"] == 1 and out["exit_codes"] == [0, 7] and "rank 1" in out["error"]
    assert out["phase"] and set(out["phases"]) == {"0", "1"}
    assert out["line_of_rank0"]["n_gpus"] == 2  # what rank 0 had measured before the failure is kept, marked as part of an error


def test_bench_self_launch_ends_a_hung_run():
    """A rank that hangs before it gets anywhere: at --launch-timeout the launcher ends exactly the workers it started and prints
    ONE JSON line that says where every rank was; non-zero exit."""
    res = self_launch({"HNH_BENCH_WORKER": os.path.join(ROOT, "tests", "bench_worker.py"), "BENCH_WORKER_HANG_RANK": "1"},
                      extra_args=("--no-tune", "--launch-timeout", "14", "--watchdog", "9"))
    assert res.returncode != 0
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-1500:]
    out = read_line(lines[0])
    assert out["value"] is None and "error" in out and set(out["phases"]) == {"0", "1"}
    assert out["phases"]["1"].startswith("start-up")  # the hung rank never reached the benchmark body


def test_bench_without_a_gpu_fails_loudly_with_one_line():
    """The product path (no test worker) on a box without a GPU: no CPU fallback — one JSON line with "error", non-zero exit."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    res = self_launch({})
    assert res.returncode != 0
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = read_line(lines[0])
    assert out["value"] is None and "error" in out and out["failed_rank"] in (0, 1) and "transport creation" in out["phase"]
    assert "no GPU visible" in res.stderr


def test_a_failing_route_candidate_is_recorded_and_the_line_survives():
    """One candidate of the route search throws: the ranks agree to drop it (recorded as null with the reason), the transport
    it ran on is not used again, and the run still ends with the measured line of the default route, exit code 0."""
    res = self_launch({"HNH_BENCH_WORKER": os.path.join(ROOT, "tests", "bench_worker.py"), "BENCH_WORKER_POISON": "mesh/4 chunks"})
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = read_line(lines[0])
    assert out["value"] > 0 and out["check"]["ok"] and "error" not in out
    tuned, failed = out["config"]["route_tuning_ms_per_step"], out["config"]["route_tuning_failures"]
    assert tuned["c=1 mesh/4 chunks [default]"] is None and "c=1 mesh/4 chunks [default]" in failed
    assert any(v is not None for v in tuned.values())  # candidates before the failure were measured
    assert all("skipped" in why or "poisoned" in why or "another rank" in why for why in failed.values())
    # what was timed is a route that was measured, not the failed one
    assert out["config"]["mesh_chunks"] != "4"


def test_a_candidate_that_hangs_still_leaves_the_line_in_hand():
    """A candidate fails on ONE rank only, so the other rank waits inside the transport for ever: its watchdog ends the wait, and
    because the default route had been measured in full BEFORE the search, rank 0 prints that line, marked incomplete — the run
    does not end without a number."""
    res = self_launch({"HNH_BENCH_WORKER": os.path.join(ROOT, "tests", "bench_worker.py"), "BENCH_WORKER_POISON": "mesh/4 chunks",
                       "BENCH_WORKER_POISON_RANK": "1"}, extra_args=("--watchdog", "12"))
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-2000:]
    out = read_line(lines[0])
    assert res.returncode == 0 and out["value"] > 0 and out["check"]["ok"]
    assert "stuck in phase 'route tuning: c=1 mesh/4 chunks [default]'" in out["incomplete"]
    assert out["config"]["mesh_chunks"] == "1,2,2,2,1,1"  # the default route's measurement


def run_worker_directly(n, *cli, timeout=600):
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2",
                   GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_worker.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
                                       "--no-cpu-baseline", "--no-secondary", *cli], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-1500:] for o in outs)
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][0][-1500:]
    return read_line(lines[0])


@pytest.mark.parametrize("n,cli,app", [(1, ("--workload", "rmat", "--logm", "9", "--edge-factor", "8", "--r", "16"), "vanilla"),
                                       (2, ("--workload", "rmat", "--app", "als", "--logm", "9", "--edge-factor", "8", "--r", "16", "--no-tune"), "als"),
                                       (1, ("--app", "gat", "--logm", "7", "--edge-factor", "4"), "gat"),
                                       (2, ("--app", "gat", "--logm", "7", "--edge-factor", "4", "--no-tune"), "gat")])
def test_workloads_and_applications_of_the_reference_harness(n, cli, app):
    """--workload rmat and --app als | gat (benchmark_dist.cpp:88-141) through bench.run() on CPU ranks: the line names them and
    carries the application's own check (ALS: the residual falls; GAT: the rank-one closed form, layer by layer)."""
    out = run_worker_directly(n, *cli)
    assert out["config"]["app"] == app and out["value"] > 0 and out["check"]["ok"], out.get("check")
    if "rmat" in cli:
        assert "R-MAT" in out["config"]["workload"]
    if app == "als":
        assert out["check"]["residual_after"] < out["check"]["residual_before"]
    if app == "gat":
        assert out["check"]["rel_err"] <= 1e-9 and "closed form" in out["check"]["what"]


def test_matrix_market_workload(tmp_path):
    """--workload mtx:<file> (bench_file.cpp:23-28): the file is read by the library's parallel parser on every rank, the result
    check sums over the same file parsed independently with scipy."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hnh_testlib as T
    mtx = str(tmp_path / "g.mtx")
    T.write_symmetric_mtx_with_duplicates(mtx, 300, 3)
    out = run_worker_directly(2, "--workload", "mtx:" + mtx, "--r", "16", "--no-tune")
    assert out["data"] == "file" and "g.mtx" in out["config"]["workload"] and out["check"]["ok"] and out["check"]["rows_checked"] == out["config"]["M"]


def test_secondary_workloads_are_listed_with_their_checks():
    """N = 1: the other workloads of the reference's harness ride in the same JSON line under "secondary", each with its byte
    model, a fraction and a check of its own (toy sizes here; the code is the GPU run's)."""
    port = free_port()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2",
               HNH_BENCH_SECONDARY_SMALL="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_worker.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                          "--logm", "9", "--edge-factor", "8", "--r", "16"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    printed = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][0]
    out = read_line(printed)
    # the printed line carries one row per entry — id, ms, frac — and the count of entries and failed checks
    table = json.loads(printed)
    assert [r["id"] for r in table["secondary"]] == [e["id"] for e in out["secondary"]] and table["secondary_checks"] == {"entries": 19, "failed": []}
    assert all(set(r) <= {"id", "ms", "frac", "sddmm", "spmm", "frac_wall", "launches"} and r["ms"] > 0 for r in table["secondary"])
    assert len(json.dumps(table["secondary"])) <= 1800
    sec = out["secondary"]
    assert len(sec) == 19 and not [e for e in sec if "error" in e], [e.get("error") for e in sec]
    by_name = {e["workload"]: e for e in sec}
    # one rank's share of configs 3 / 4 / 5 and config 1 as typed: rank 0 of p logical ranks alone (held blocks / solo replay)
    shares = [v for k, v in by_name.items() if k.startswith("rank share, config 3")]
    assert sorted((e["p"], e["chunks"]) for e in shares) == [(2, "1,2,2,2,1,1"), (4, "1,2,2,2,1,1"), (8, "1"), (8, "1,2,2,2,1,1")]
    for e in shares:
        assert e["held"]["wall_ms"] > 0 and e["solo"]["wall_ms"] > 0 and e["algorithmic_bytes_rank"] > 0 and e["nnz_rank"] > 0
        # own block + one pass per chunk window (a single pass over the fetched blocks when there is one chunk)
        # one pass per chunk: own block + one launch per chunk window; everything landed: own block + ONE pass over the fetched blocks
        assert e["held"]["launches"] == (2 if e["chunks"] == "1" else 7) and e["held_all_landed"]["launches"] == 2 and e["solo"]["launches"] >= 2, e
    for key in ("rank share, config 4", "rank share, config 5", "rank share, 2.5D sparse-replicate", "rank share, 1.5D dense shift by replication reuse", "config 1 as typed"):
        e = next(v for k, v in by_name.items() if k.startswith(key))
        assert e["solo"]["wall_ms"] > 0 and e["solo"]["launches"] > 0 and e["all_ranks_on_this_gpu_ms"] > 0 and e["algorithmic_bytes_rank"] > 0, e
    for r in (8, 16, 128, 256, 512):
        e = next(v for k, v in by_name.items() if "R=%d:" % r in k)
        assert e["check"]["ok"] and all(e[op]["ms"] > 0 and e[op]["algorithmic_bytes"] > 0 for op in ("fused", "sddmm", "spmm"))
        assert e["sddmm"]["call_ms"] > 0 and e["spmm"]["call_ms"] > 0  # whole sddmmA / spmmA calls
        b = e["borrowed_value_arrays"]  # one stationary block at offset 0: SValues read in place, results written in place
        assert b["spmm_lent"] > 0 and b["sddmm_in_place"] > 0 and b["spmm_copied"] == 0 and b["sddmm_hadamard"] == 0
    assert next(v for k, v in by_name.items() if "ALS" in k)["check"]["ok"]
    assert next(v for k, v in by_name.items() if k.startswith("GAT"))["check"]["rel_err"] <= 1e-9
    assert next(v for k, v in by_name.items() if k.startswith("R-MAT"))["check"]["ok"]
    assert next(v for k, v in by_name.items() if "config 4" in k)["check"]["ok"]
    knl = next(v for k, v in by_name.items() if "printed weak-scaling point" in k)  # (the comparison itself is only made at the printed size)
    assert knl["check"]["ok"] and knl["seconds_for_5_fusedmm"] > 0 and knl["schedule"].startswith("15d_sparse") and "reference_printed" not in knl


def test_a_signal_before_the_first_measurement_still_leaves_one_json_line():
    """SIGTERM while rank 0 holds no complete measurement yet (a driver that ends the run during set-up): the sigwait() thread prints
    the contract's error line instead of dying silently; exit code 143."""
    import signal
    code = ("import sys, time\nsys.path.insert(0, %r)\nfrom benchlib import guards, common\ncommon.claim_stdout()\n"
            "fb = guards.Fallback(0, guards.Phases(), lambda why: common.emit({'metric': 'fused SDDMM+SpMM nnz*R/s', 'value': None, 'error': why}))\n"
            "fb.watch_sigterm()\nprint('armed', file=sys.stderr, flush=True)\ntime.sleep(120)\n" % ROOT)
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.stderr.readline().strip() == "armed"
    p.send_signal(signal.SIGTERM)
    out, _ = p.communicate(timeout=60)
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert p.returncode == 143 and len(lines) == 1, (p.returncode, out)
    rec = json.loads(lines[0])
    assert rec["value"] is None and "before its first complete measurement" in rec["error"] and len(lines[0]) < LINE_LIMIT


def test_the_line_stays_bounded_whatever_goes_into_it(tmp_path, monkeypatch):
    """benchlib/line.py on its own: a record with hundreds of search candidates, a hundred secondary entries, kilobyte strings, NaN and
    infinity still renders to ONE strict-JSON line of at most 8192 bytes that keeps the contract's keys, `roofline` and `cpu_baseline`;
    what was left out is named in `shed`, and the record file holds everything.  A line that was rendered before (the launcher forwarding
    rank 0's) keeps its secondary table and amends the record instead of replacing it."""
    from benchlib import line as L
    monkeypatch.setenv("HNH_BENCH_RECORD", str(tmp_path / "rec.json"))
    long = "x" * 5000
    rec = {"backend": "hip-gfx950", "metric": "fused SDDMM+SpMM nnz*R/s", "value": 1.0e12, "unit": "nnz*R/s", "n_gpus": 8, "steps": 20, "warmup": 5,
           "ms_per_step": 2.5, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": long, "nnz": 1, "route_tuning_ms_per_step": {"candidate %d %s" % (k, long[:100]): float(k) for k in range(300)},
                      "route_tuning_failures": {"c %d" % k: long for k in range(50)}, "transport_trials": {"rccl": long, "ipc": long},
                      "ranks": [[r, 1000 + r, r, "0000:%02x:00.0" % r, 8, r, r] for r in range(8)], "budget_stops": [long] * 5},
           "roofline": {"bound": "hbm", "achieved": float("nan"), "peak": 8000.0, "unit": "GB/s", "frac": float("inf"), "traffic": None, "model": long},
           "cpu_baseline": {"value": 3.6e9, "unit": "nnz*R/s", "cores": 32, "kind": "reference", "sample": long},
           "check": {"what": long, "ok": True}, "phases_s": {"p%d" % k: 1.0 for k in range(40)}, "preflight": {"primitives_ok": [long] * 9},
           "secondary": [{"id": "entry_%d" % k, "workload": long, "ms": 1.0 + k, "frac": 0.5, "check": {"ok": k != 3}} for k in range(100)]}
    data, path = L.render(rec)
    assert len(data) <= L.LINE_LIMIT and data.endswith(b"\n") and data.count(b"\n") == 1
    line = read_line(data.decode().strip(), want_record=False)
    assert line["value"] == 1.0e12 and line["roofline"]["achieved"] is None and line["roofline"]["frac"] is None  # strict JSON: no NaN / Infinity
    assert line["cpu_baseline"]["value"] == 3.6e9 and len(line["config"]["workload"]) <= 200 and line["shed"], sorted(line)
    with open(path) as f:
        full = json.load(f)
    assert len(full["secondary"]) == 100 and len(full["config"]["route_tuning_ms_per_step"]) == 300 and full["config"]["workload"] == long
    # a second pass over the printed line (the launcher): nothing is lost from the record, its remark is added
    again = dict(line, incomplete="run ended by signal 15 sent to the launcher", exit_codes=[0] * 8)
    data2, _ = L.render(again)
    assert len(data2) <= L.LINE_LIMIT
    with open(path) as f:
        full2 = json.load(f)
    assert len(full2["secondary"]) == 100 and "signal 15" in full2["incomplete"] and full2["exit_codes"] == [0] * 8
    # a moderately sized record sheds nothing and keeps its whole secondary table
    small = dict(rec, config={"workload": "w"}, check={"ok": True}, phases_s={"total": 1.0}, preflight=None,
                 roofline=dict(rec["roofline"], model="m"), cpu_baseline=dict(rec["cpu_baseline"], sample="s"),
                 secondary=[{"id": "e%d" % k, "workload": "w", "ms": 1.0, "frac": 0.5} for k in range(19)])
    small.pop("preflight")
    d3 = json.loads(L.render(small)[0])
    assert "shed" not in d3 and len(d3["secondary"]) == 19 and d3["secondary_checks"] == {"entries": 19, "failed": []}
