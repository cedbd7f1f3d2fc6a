"""bench.py's command line."""
import argparse


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--logm", type=int, default=20)
    ap.add_argument("--edge-factor", type=int, default=96)
    ap.add_argument("--r", "--rvalue", dest="r", type=int, default=128, help="embedding width R (under torch.distributed.run say --rvalue: its own "
                    "parser takes a bare --r as an ambiguous abbreviation of its --rdzv-* / --role / --run-path options)")
    ap.add_argument("--alg", default="15d_fusion2")
    ap.add_argument("--workload", default="er", help="er = Erdos-Renyi 2^logm, edge factor (the default); rmat = skewed R-MAT of the same size "
                    "(stand-in for a SuiteSparse graph, BASELINE config 4); mtx:<path> = a MatrixMarket file (bench_file.cpp:23-28)")
    ap.add_argument("--app", choices=["vanilla", "als", "gat"], default="vanilla", help="what a step is (benchmark_dist.cpp:117-141): vanilla = one "
                    "fusedSpMM; als = one alternating ALS step by batched CG (run_cg(1)); gat = one GAT forward pass (layers of benchmark_dist.cpp:88-94)")
    ap.add_argument("--c", type=int, default=None, help="replication factor of the 1.5D/2.5D schedule (the reference's command-line "
                    "argument, bench_erdos_renyi.cpp:23-28).  Not given: 1 on one GPU; on several GPUs the candidates 1 / 2 / 4 that "
                    "divide N are MEASURED together with the route (below) and the fastest is timed")
    ap.add_argument("--transport", choices=["auto", "rccl", "ipc"], default="auto", help="several GPUs: device-to-device transport.  auto = "
                    "both are probed in child processes, the usable ones are measured (ipc with copy engines and with a pull kernel) and the "
                    "fastest is timed")
    ap.add_argument("--ring-mode", choices=["mesh", "relay"], default=None,
                    help="route of the 1.5D dense shift's moving operand: mesh = every block straight from its owner (default), "
                         "relay = the reference's neighbour ring (sets HNH_RING_MODE)")
    ap.add_argument("--chunks", default=None, help="chunks of the pipelined mesh fetch: a number Q = symmetric chunks of heights "
                    "(1, 2, .., 2, 1) (HNH_MESH_CHUNKS), or a comma list of heights, e.g. 1,2,2,2,1,1 (HNH_MESH_TAPER)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-logm", type=int, default=18, help="the CPU baseline's sample: ER 2^cpu_logm at the line's edge factor and R")
    ap.add_argument("--cpu-trials", type=int, default=2)
    ap.add_argument("--cpu-full", action="store_true", help="CPU baseline: also run the sweep's winner once at the GPU line's full size and report "
                    "that as cpu_baseline.value (about a minute more, nearly all of it the reference's set-up)")
    ap.add_argument("--no-cpu-full", action="store_true", help=argparse.SUPPRESS)  # (the default since round 6; kept for old command lines)
    ap.add_argument("--no-check", action="store_true", help="skip the closed-form result check (outside the timed region)")
    ap.add_argument("--no-preflight", action="store_true", help="skip the transport self-tests of a multi-GPU run")
    ap.add_argument("--no-tune", action="store_true", help="several GPUs: keep the default route (first usable transport, mesh fetch, default "
                    "chunk heights) instead of measuring transports, replication factors, chunk shapes and the relay ring")
    ap.add_argument("--no-secondary", action="store_true", help="one GPU: skip the secondary workloads (R-MAT, config 4's schedule, ALS, GAT, other widths)")
    ap.add_argument("--watchdog", type=float, default=240.0, help="seconds a multi-GPU phase may take before the rank reports "
                    "where it is stuck and exits (with the best line measured so far, if there is one)")
    ap.add_argument("--probe-timeout", type=float, default=300.0, help="several GPUs: seconds a transport's child-process trial may take")
    ap.add_argument("--no-live-traffic", action="store_true", help="one GPU: do not run the two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) "
                    "behind roofline.traffic; the tracked profiles/hbm_traffic.json is quoted instead")
    ap.add_argument("--nchannels", type=int, default=None, help="several GPUs: pin RCCL's channel count (NCCL_MIN/MAX_NCHANNELS); every "
                    "channel is a workgroup that competes with the row kernel for CUs and HBM")
    ap.add_argument("--budget-s", type=float, default=1200.0, help="seconds the whole run may take, counted from process start (the driver ends a "
                    "run at 1800 s): optional work — further route candidates, other transports, the CPU baseline's sample leg at N > 1 — is only "
                    "started while it fits into what is left; the measured line is never optional")
    ap.add_argument("--launch-timeout", type=float, default=None, help="self-launched run (--gpus N without WORLD_SIZE): seconds before the "
                    "launcher ends its workers and reports the phase each one was in (default: --budget-s + 120)")
    ap.add_argument("--probe-transport", default=None, help=argparse.SUPPRESS)  # internal: the child-process trial of one transport
    args = ap.parse_args(argv)
    if args.launch_timeout is None:
        args.launch_timeout = args.budget_s + 120.0
    return args
