"""Development tool: time the GAT forward pass (benchmark_dist.cpp:91-96 layer spec) and its fp64 MFMA GEMM on one GPU."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_sddmm_amd import api as H, _kernels as K
assert H.load_backend(None) == "hip-gfx950"
logm = int(sys.argv[1]) if len(sys.argv) > 1 else 18
algs = [sys.argv[2]] if len(sys.argv) > 2 else ["15d_fusion2", "15d_fusion1"]  # fused head in one launch / the reference's call sequence
w = H.World.single(0)
RMAT = int(os.environ.get("HNH_PROFILE_RMAT_EDGES", "0"))  # > 0: a skewed R-MAT graph with this many edge draws (hub rows) instead of Erdos-Renyi
if RMAT > 0:
    rr, cc = H.generate_rmat(logm, RMAT)
    print("R-MAT 2^%d vertices, %d unique nonzeros, longest row %d" % (logm, len(rr), int(np.bincount(rr).max())))
    sp = H.SpmatLocal.from_global(w, 1 << logm, 1 << logm, rr, cc, None)
else:
    sp = H.SpmatLocal.load_tuples(w, False, logm, 32)
nnz = sp.info()["dist_nnz"]
layers = [(256, 256, 4), (1024, 256, 4), (1024, 256, 6)]   # benchmark_dist.cpp:93-95
for alg in algs:
    op = H.DistributedSparse(w, alg, sp, 256, 1)
    gnn = H.GAT(op, layers, 0.2)
    rng = np.random.default_rng(0)
    for li, (fin, fph, heads) in enumerate(layers):
        for h in range(heads):
            k, n = gnn.weight_shape(li, h)
            gnn.set_weight(li, h, rng.uniform(-1, 1, (k, n)) / k)
    x = H.Dense.create(w, *gnn.buffer_shape(0)); x.fill(0.01); gnn.set_input(x)
    gnn.forwardPass(); w.sync()
    t = time.perf_counter()
    for _ in range(3): gnn.forwardPass()
    w.sync(); dt = (time.perf_counter() - t) / 3
    heads_total = sum(l[2] for l in layers)
    print("GAT forward [%s] (2^%d vertices, %d nnz, %d heads): %.1f ms, %.2f ms per head" % (alg, logm, nnz, heads_total, dt * 1e3, dt * 1e3 / heads_total))
    gnn.free(); x.free(); op.free()
if os.environ.get("HNH_PROFILE_NO_GEMM"): sys.exit(0)
# GEMM alone: M x 1024 times 1024 x 256
ctx = K.Ctx(0); lib = ctx.lib
M, Kd, N = 1 << logm, 1024, 256
dA, dB, dC = K.DevArray(ctx, (M, Kd), np.float64), K.DevArray(ctx, (Kd, N), np.float64), K.DevArray(ctx, (M, N), np.float64)
lib.hnh_fill_f64(ctx.h, dA.ptr, M * Kd, 0.5, 0); lib.hnh_fill_f64(ctx.h, dB.ptr, Kd * N, 0.25, 0)
lib.hnh_gemm_f64(ctx.h, M, N, Kd, dA.ptr, dB.ptr, dC.ptr, 0); ctx.sync()
t = time.perf_counter()
for _ in range(3): lib.hnh_gemm_f64(ctx.h, M, N, Kd, dA.ptr, dB.ptr, dC.ptr, 0)
ctx.sync(); dt = (time.perf_counter() - t) / 3
print("gemm_f64 %d x %d x %d: %.2f ms -> %.1f TFLOP/s fp64 (MI355X fp64 matrix peak 78.6)" % (M, N, Kd, dt * 1e3, 2.0 * M * N * Kd / dt / 1e12))
