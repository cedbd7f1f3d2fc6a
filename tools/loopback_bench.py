"""Development tool: BASELINE config 3's schedule at FULL size on ONE GPU through the loopback transport
(p logical ranks = p host threads sharing the device).  Not a multi-GPU measurement — every rank's kernels and
every ring transfer share one HBM — but it runs the exact multi-rank code path (redistribution, p block
columns per rank, ring/mesh transfers with event ordering) at 1e8 nnz, and shows per-launch kernel efficiency."""
import argparse
import os
import sys
import time

# N logical ranks = 2 N streams on ONE device: HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams
# that share a queue serialise — give every stream its own (has to be set before the HIP runtime starts)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_sddmm_amd import api as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--p", type=int, default=8); ap.add_argument("--c", type=int, default=1); ap.add_argument("--alg", default="15d_fusion2")
ap.add_argument("--logm", type=int, default=20); ap.add_argument("--ef", type=int, default=96); ap.add_argument("--r", type=int, default=128)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--rmat-edges", type=int, default=0, help="> 0: skewed R-MAT graph on 2^logm vertices with this many edge draws "
                "(BASELINE config 4's stand-in for com-Orkut) instead of the ER matrix")
a = ap.parse_args()
assert H.load_backend(None) == "hip-gfx950"


RMAT = None
if a.rmat_edges > 0:
    t_gen = time.time()
    RMAT = H.generate_rmat(a.logm, a.rmat_edges)
    print("R-MAT 2^%d vertices, %d unique nonzeros, longest row %d (generated on the host in %.0f s)"
          % (a.logm, len(RMAT[0]), int(np.bincount(RMAT[0]).max()), time.time() - t_gen))


def body(w):
    if RMAT is not None:
        sp = H.SpmatLocal.from_global(w, 1 << a.logm, 1 << a.logm, RMAT[0], RMAT[1], None)
    else:
        sp = H.SpmatLocal.load_tuples(w, False, a.logm, a.ef)
    nnz = sp.info()["dist_nnz"]
    op = H.DistributedSparse(w, a.alg, sp, a.r, a.c)
    sp.free()
    A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
    S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
    op.fusedSpMM(A, B, S, buf, H.AMAT); w.sync(); w.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        op.fusedSpMM(A, B, S, buf, H.AMAT)
    w.sync(); w.barrier()
    dt = (time.perf_counter() - t0) / a.steps
    op.reset_performance_timers()
    for _ in range(a.steps):
        op.fusedSpMM(A, B, S, buf, H.AMAT)
    w.sync(); w.barrier()
    stats = op.json_perf_statistics()   # device-timed phases, mean over ranks, summed over a.steps calls
    op.kernel_profile(1)
    op.fusedSpMM(A, B, S, buf, H.AMAT); w.sync()
    kms, kl = op.kernel_profile(0)
    chk = float(np.sum(A.download()[:4]))
    for x in (A, B, S, buf):
        x.free()
    op.free()
    return nnz, dt, kms, kl, chk, stats


t = time.time()
res = H.run_spmd(a.p, body)
nnz, dt = res[0][0], max(r[1] for r in res)
print("p=%d c=%d %s ring=%s: %.2f ms per fused call -> %.3e nnz*R/s on ONE gpu (setup+run %.0f s); per-rank serialized kernel time %.2f ms over %d launches; checksum %.6e"
      % (a.p, a.c, a.alg, os.environ.get("HNH_RING_MODE", "mesh"), dt * 1e3, nnz * a.r / dt, time.time() - t, res[0][2], res[0][3], res[0][4]))
print("   device-timed phases per fused call, mean over ranks (ms): " + ", ".join("%s %.2f" % (k, v * 1e3 / a.steps) for k, v in res[0][5].items())
      + "   [compute-stream and communication-stream phases run concurrently; %.2f ms elapsed per call]" % (dt * 1e3))
