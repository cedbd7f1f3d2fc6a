"""Development tool: ONE rank's share of any schedule at a configuration's size, alone on the GPU in solo replay (World::set_solo: every
message the rank would receive is replaced by a device copy of what it would send — same bytes, streams, events): wall time per call,
event-bracketed row-kernel time and launches, and what is left over ("outside the row kernels": value copies, zero fills, Hadamard
passes, the copies that stand for its shifts, host enqueue and event waits).  Under rocprofv3 --kernel-trace --memory-copy-trace the
last calls give the timeline that says which of those it is (tools/rocpd_timeline.py --busy).

    python tools/rank_share_schedule.py --alg 25d_dense_replicate --p 8 --c 2 --kind rmat --logm 20 --ef 44 --r 256 [--op fused|sddmm|spmm]"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--alg", default="25d_dense_replicate")
ap.add_argument("--p", type=int, default=8)
ap.add_argument("--c", type=int, default=2)
ap.add_argument("--kind", choices=["er", "rmat"], default="rmat")
ap.add_argument("--logm", type=int, default=20)
ap.add_argument("--ef", type=int, default=44)
ap.add_argument("--r", type=int, default=256)
ap.add_argument("--op", choices=["fused", "sddmm", "spmm"], default="fused")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--backend", default=None)
a = ap.parse_args()
import numpy as np  # noqa: E402
from distributed_sddmm_amd import api as H  # noqa: E402

name = H.load_backend(a.backend)
assert a.backend or name == "hip-gfx950"


def body(w):
    if a.kind == "er":
        sp = H.SpmatLocal.load_tuples(w, False, a.logm, a.ef)
    else:
        rows, cols = H.generate_rmat(a.logm, (1 << a.logm) * a.ef)
        sp = H.SpmatLocal.from_global(w, 1 << a.logm, 1 << a.logm, rows, cols, np.ones(len(rows)))
    gnnz = sp.info()["dist_nnz"]
    op = H.DistributedSparse(w, a.alg, sp, a.r, a.c)
    sp.free()
    A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
    S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
    call = {"fused": lambda: op.fusedSpMM(A, B, S, buf, H.AMAT), "sddmm": lambda: op.sddmmA(A, B, S, buf), "spmm": lambda: op.spmmA(A, B, S)}[a.op]
    call()
    w.sync()
    w.barrier()
    out = None
    try:
        if w.rank == 0:
            w.set_solo(True)
            call()
            w.sync()
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    call()
                w.sync()
                t = (time.perf_counter() - t0) * 1e3 / a.iters
                best = t if best is None else min(best, t)
            op.kernel_profile(1)
            for _ in range(a.iters):
                call()
            w.sync()
            kms, launches = op.kernel_profile(0)
            # the calls a trace should look at: a marker-free way to find them is "the last iters calls of the run"
            for _ in range(a.iters):
                call()
            w.sync()
            out = (best, kms / a.iters, launches // a.iters, gnnz)
    finally:
        if w.rank == 0:
            w.set_solo(False)
        w.barrier()
    for x in (A, B, S, buf):
        x.free()
    op.free()
    return out


wall, kms, launches, gnnz = H.run_spmd(a.p, body)[0]
m = 1 << a.logm
unfused = gnnz * (16 * a.r + 44) + 16 * a.r * m
by = {"fused": unfused if a.alg != "15d_fusion2" else gnnz * (8 * a.r + 24) + 16 * a.r * m, "sddmm": gnnz * (8 * a.r + 20) + 8 * a.r * m,
      "spmm": gnnz * (8 * a.r + 12) + 16 * a.r * m}[a.op] / a.p
print("%s p=%d c=%d, %s 2^%d ef %d (%d nnz), R=%d, %s: rank 0 alone: wall %.3f ms, row kernels %.3f ms in %d launches, outside the row kernels %.3f ms; "
      "byte model / p %.3e B: frac_kernel %.3f frac_wall %.3f" % (a.alg, a.p, a.c, a.kind, a.logm, a.ef, gnnz, a.r, a.op, wall, kms, launches, wall - kms, by,
                                                                  by / (kms * 1e-3) / 8e12 if kms > 0 else 0.0, by / (wall * 1e-3) / 8e12))
