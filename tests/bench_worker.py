"""Worker of tests/test_bench_cpu.py: runs bench.run() — the real benchmark driver — on CPU ranks with the
transport swapped for gloo callbacks and the kernel ABI for the oracle test double (explicitly, here)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import hnh_testlib as T  # noqa: E402


def cpu_world(H, dist, rank, n, local_rank):
    assert H.load_backend(T.ORACLE_BACKEND) == "oracle-cpu-test-double"
    if n == 1:
        return H.World.single(0), (lambda: None)
    from gloo_worker import make_callbacks
    cpu_world.cb = make_callbacks()  # keep the ctypes thunks alive
    return H.World.callback(rank, n, 0, cpu_world.cb), (lambda: None)


if __name__ == "__main__":
    n = int(os.environ["WORLD_SIZE"])
    if len(sys.argv) > 1 and sys.argv[1].startswith("--"):
        # started by bench.py's own launcher (HNH_BENCH_WORKER points here): bench.py's command line, untouched
        args = bench.parse(sys.argv[1:])
        assert args.gpus == n
        if os.environ.get("BENCH_WORKER_HANG_RANK") == os.environ["RANK"]:  # a rank that never gets anywhere: the launcher's time limit, exercised
            import time
            time.sleep(3600)
        poison = os.environ.get("BENCH_WORKER_POISON")
        if poison:  # one candidate of the route search fails — on every rank, or on BENCH_WORKER_POISON_RANK only (the others then wait for it)
            real_build = bench.Bench.build

            def build(self, route):
                if poison in bench.route_name(route) and os.environ.get("BENCH_WORKER_POISON_RANK", os.environ["RANK"]) == os.environ["RANK"]:
                    raise RuntimeError("poisoned candidate (test)")
                return real_build(self, route)
            bench.Bench.build = build
        bench.run(args, make_world=cpu_world)  # rank 0 prints the JSON line itself
        if os.environ.get("BENCH_WORKER_FAIL_RANK") == os.environ["RANK"]:  # the launcher's failure report, exercised
            sys.exit(7)
        sys.exit(0)
    args = argparse.Namespace(gpus=n, steps=2, warmup=1, logm=10, edge_factor=8, r=16, alg=sys.argv[1], c=int(sys.argv[2]) or None,
                              no_cpu_baseline=True, cpu_logm=10, cpu_trials=1, ring_mode=os.environ.get("BENCH_RING_MODE") or None,
                              chunks=None, no_cpu_full=True, no_check=False, no_preflight=False, no_tune=False, watchdog=120.0, no_live_traffic=True,
                              nchannels=None, workload="er", app="vanilla", transport="auto", no_secondary=True, probe_timeout=60.0)
    out = bench.run(args, make_world=cpu_world)
    if out is not None:
        print("BENCH_JSON " + json.dumps(out), flush=True)
