"""TEST INFRASTRUCTURE (oracle tooling): the compiled reference's fused throughput vs (MPI ranks x OpenMP/MKL threads)
on this box — how the thread counts tried by bench.py's cpu_baseline leg were chosen (profiles/archive/r01_cpu_baseline_sweep.log)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_sddmm_amd import api as H
from oracle import refrun as RR
logm = int(sys.argv[1]) if len(sys.argv) > 1 else 18
m = 1 << logm
rows, cols = H.generate_er(m, m, m * 96, 12345)
cfgs = [tuple(int(v) for v in x.split("x")) for x in (sys.argv[2] if len(sys.argv) > 2 else "1x32,4x32,8x16,8x32,16x16,32x8").split(",")]
for p, t in cfgs:
    r = RR.bench(m, m, rows, cols, 128, "15d_fusion2", p, 1, True, 2, threads=t)
    print("ranks %2d x threads %3d: elapsed %.3f s  %.3e nnz*R/s  (Computation Time %.3f s, shift %.3f s)" % (
        p, t, r["elapsed"], r["nnz_R_per_s"], r["perf_stats"]["Computation Time"], r["perf_stats"]["Cyclic Shift Time"]), flush=True)
