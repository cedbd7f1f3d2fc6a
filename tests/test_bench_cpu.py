"""The benchmark driver's control flow (world bootstrap, strong-scaling setup, timed loop, max-over-ranks,
roofline bookkeeping, JSON contract) exercised on CPU ranks over gloo — bench.run() itself, with only the
transport/kernels swapped by the test.  The numbers are meaningless here; the contract is what is checked."""
import json
import os
import subprocess
import sys

import pytest

from test_gloo_world import ROOT, free_port


@pytest.mark.parametrize("nranks,alg,c,ring", [(1, "15d_fusion2", 1, None), (2, "15d_fusion2", 1, None), (4, "15d_fusion2", 1, None),
                                                (4, "15d_fusion2", 2, None), (2, "15d_fusion1", 1, None), (4, "15d_fusion2", 1, "relay"), (2, "15d_fusion2", 1, "mesh")])
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
    if nranks > 1:  # every transport primitive ran before the timed region, and what was measured is recorded
        assert len(out["preflight"]["primitives_ok"]) == 9
        tuned = alg == "15d_fusion2" and ring != "relay" and nranks // c > 1
        assert out["config"]["transport"] == "rccl"
        assert ("route_tuning_ms_per_step" in out["config"]) == tuned
        if tuned:  # the route that was timed is the fastest of the measured candidates (mesh fetch in 2 / 4 / 8 chunks, relay ring)
            t = out["config"]["route_tuning_ms_per_step"]
            # --ring-mode mesh fixes the route and leaves the chunk count to be measured
            assert set(t) == {"mesh/2 chunks", "mesh/4 chunks", "mesh/8 chunks"} | (set() if ring == "mesh" else {"relay ring"})
            best = min(t, key=t.get)
            assert out["config"]["ring_mode"] == ("relay" if best == "relay ring" else "mesh")
            assert out["config"]["mesh_chunks"] == (None if best == "relay ring" else int(best.split("/")[1].split()[0]))
        else:
            assert out["config"]["ring_mode"] == (ring or "mesh")
    else:
        assert "preflight" not in out
