"""Development tool: where does an ALS alternating step spend its time? (config 5 shape on one GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_sddmm_amd import api as H
assert H.load_backend(None) == "hip-gfx950"
w = H.World.single(0)
logm, ef, r = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 96, 128
alg = sys.argv[2] if len(sys.argv) > 2 else "15d_fusion2"
sp = H.SpmatLocal.load_tuples(w, False, logm, ef)
nnz = sp.info()["dist_nnz"]
op = H.DistributedSparse(w, alg, sp, r, 1)
t = time.perf_counter(); als = H.DistributedALS(op, True); w.sync(); print("ctor (artificial ground truth) %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); als.initializeEmbeddings(); w.sync(); print("initializeEmbeddings %.3f s" % (time.perf_counter() - t))
for it in (1, 10, 10):
    t = time.perf_counter(); als.cg_optimizer(H.AMAT, it); w.sync(); dt = time.perf_counter() - t
    print("cg_optimizer(Amat, %d) %.3f s -> %.2f ms per CG iteration incl. setup (fused kernel alone ~15-16 ms)" % (it, dt, dt / (it + 1) * 1e3))
A, B, S, buf = op.like_A_matrix(0.001), op.like_B_matrix(0.001), op.like_S_values(1.0), op.like_S_values(0.0)
op.fusedSpMM(A, B, S, buf, H.AMAT); w.sync()
t = time.perf_counter()
for _ in range(5): op.fusedSpMM(A, B, S, buf, H.AMAT)
w.sync(); print("fusedSpMM %.2f ms" % ((time.perf_counter() - t) / 5 * 1e3))
