"""Development tool: the compiled reference's fused throughput vs OpenMP/MKL thread count on this box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_sddmm_amd import api as H
from oracle import refrun as RR
logm = int(sys.argv[1]) if len(sys.argv) > 1 else 18
m = 1 << logm
rows, cols = H.generate_er(m, m, m * 96, 12345)
for t in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16,32,64,128,256").split(",")]:
    for p in (1,):
        r = RR.bench(m, m, rows, cols, 128, "15d_fusion2", p, 1, True, 2, threads=t)
        print("threads %3d ranks %d: elapsed %.3f s  %.3e nnz*R/s  (Computation Time %.3f s)" % (t, p, r["elapsed"], r["nnz_R_per_s"], r["perf_stats"]["Computation Time"]), flush=True)
