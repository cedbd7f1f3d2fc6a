"""Development tool: where does an ALS alternating step spend its time? (config 5 shape on one GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_sddmm_amd import api as H
assert H.load_backend(None) == "hip-gfx950"
w = H.World.single(0)
logm, ef, r = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 96, 128
alg = sys.argv[2] if len(sys.argv) > 2 else "15d_fusion2"
RMAT = int(os.environ.get("HNH_PROFILE_RMAT_EDGES", "0"))  # > 0: a skewed R-MAT graph with this many edge draws (hub rows) instead of Erdos-Renyi
if RMAT > 0:
    rr, cc = H.generate_rmat(logm, RMAT)
    print("R-MAT 2^%d vertices, %d unique nonzeros, longest row %d" % (logm, len(rr), int(__import__("numpy").bincount(rr).max())))
    sp = H.SpmatLocal.from_global(w, 1 << logm, 1 << logm, rr, cc, None)
else:
    sp = H.SpmatLocal.load_tuples(w, False, logm, ef)
nnz = sp.info()["dist_nnz"]
op = H.DistributedSparse(w, alg, sp, r, 1)
for unfolded in (False, True):
    # HNH_ALS_UNFOLDED is read when the ALS object is made: off = the whole CG iteration rides in the fused call's row epilogue
    # (hnh_cg_update), on = the separate dense update launches that R-split schedules need
    if unfolded:
        os.environ["HNH_ALS_UNFOLDED"] = "1"
    else:
        os.environ.pop("HNH_ALS_UNFOLDED", None)
    label = "separate CG update launches" if unfolded else "CG updates folded into the fused call"
    t = time.perf_counter(); als = H.DistributedALS(op, True); w.sync(); print("[%s] ctor (artificial ground truth) %.3f s" % (label, time.perf_counter() - t))
    t = time.perf_counter(); als.initializeEmbeddings(); w.sync(); print("[%s] initializeEmbeddings %.3f s" % (label, time.perf_counter() - t))
    took = {}
    for it in (1, 1, 10, 10):
        t = time.perf_counter(); als.cg_optimizer(H.AMAT, it); w.sync(); took[it] = time.perf_counter() - t
        print("[%s] cg_optimizer(Amat, %d) %.3f s" % (label, it, took[it]))
    print("[%s] -> %.2f ms per CG iteration ((t(10) - t(1)) / 9; the fused call alone is timed below)" % (label, (took[10] - took[1]) / 9 * 1e3))
    als.free()
A, B, S, buf = op.like_A_matrix(0.001), op.like_B_matrix(0.001), op.like_S_values(1.0), op.like_S_values(0.0)
op.fusedSpMM(A, B, S, buf, H.AMAT); w.sync()
t = time.perf_counter()
for _ in range(5): op.fusedSpMM(A, B, S, buf, H.AMAT)
w.sync(); print("fusedSpMM %.2f ms" % ((time.perf_counter() - t) / 5 * 1e3))
