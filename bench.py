#!/usr/bin/env python3
"""Headline benchmark: fused SDDMM -> SpMM (Distributed_Sparse::fusedSpMM, the reference's
benchmark_dist.cpp:117-149 loop) on an Erdős–Rényi matrix, nnz*R per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload er|rmat|mtx:<path>] [--app vanilla|als|gat]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1 : BASELINE config 2 — ER 2^20 x 2^20 (edge factor 96, ~1.0066e8 nnz), R = 128, one MI355X, the local
        fused kernel behind `15d_fusion2` (no shift).  A "step" = one fusedSpMM(A, B, S, buf, Amat) call.
N > 1 : BASELINE config 3 — the SAME global matrix strong-scaled over N GPUs with the 1.5D dense-shifting
        schedule, one process per GPU.  Two device-to-device transports stand behind the same schedules: RCCL
        send/recv groups over xGMI and the ipc-pull transport (receivers copy out of their peers' mapped buffers);
        each is first tried in a CHILD process (a transport that fails or hangs there is left alone), then
        transport, replication factor and route are measured and the fastest is timed.
Other workloads / applications of the reference's harness (benchmark_dist.cpp:88-141, bench_file.cpp:23-103):
--workload rmat | mtx:<file> (configs 4) and --app als | gat (config 5) select them for the timed line; the default
N = 1 run also times a bounded instance of each and lists them under "secondary" (outside the timed region).
Inputs are resident in HBM before the timed region (A = B = 0.001, S = 1 as benchmark_dist.cpp:102-106).

The code lives in benchlib/: timed.py is the timed path (build, step, measure, the line), run.py the order of a run; cli, guards
(watchdog, time budget, line in hand), transports, search, checks, secondary, baseline and launcher are around them.

One JSON line is printed by rank 0.  Extra objects:
  roofline     — the dominant kernel (fused row pass): algorithmic bytes per launch / average launch
                 duration measured live with HIP events on the compute stream, against 8.0 TB/s HBM.
  roofline     — ... `frac` = `frac_kernel` on one GPU; on several GPUs `frac` = `frac_step` = total B_fused / ms_per_step / (N x 8 TB/s)
                 (exposed communication counts), with `frac_kernel`, `kernel_ms_per_step` and `exposed_comm_ms` beside it.
  cpu_baseline — the reference itself (oracle/_ref/ref_driver = unmodified reference sources + MKL/MPICH)
                 timed on this box's host cores: measured by the N = 1 run (thread sweep on a bounded sample, then the same
                 workload once), quoted from that run's record by the N > 1 runs that follow on the host (else its sample leg).
  phases_s     — wall seconds per phase of the run (start-up, transport trials, bring-up, first measurement, tuning, ...);
                 --budget-s (default 1200) bounds the whole run: optional work is only started while it fits.
  secondary    — N = 1: R-MAT (hub rows), config 4's schedule on 8 logical ranks, one ALS-CG step, the GAT forward pass,
                 narrow and wide operands, the one point the reference's tree prints a time for; each with its
                 own byte model, fraction of 8 TB/s and a closed-form check.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from benchlib import common  # noqa: E402,F401  (first: the run's clock starts here)
from benchlib.cli import parse  # noqa: E402,F401
from benchlib.common import route_name  # noqa: E402,F401
from benchlib.run import main, run  # noqa: E402,F401
from benchlib.timed import Bench  # noqa: E402,F401

if __name__ == "__main__":
    main()
