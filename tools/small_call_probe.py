"""Development tool: where does the time of a SMALL operator call go?  BASELINE config 1 as typed (ER 2^16, edge factor 16, R = 16,
1.5D sparse shift, 2 logical ranks): rank 0 replays its side of the fused call alone (World::set_solo) `--iters` times; wall time per
call against the event-bracketed kernel time.  Under `rocprofv3 --hip-trace --kernel-trace --stats` the API and kernel tables say
what the rest is (tools/gpu_jobs/r05_job4.sh).

    python tools/small_call_probe.py [--alg 15d_sparse] [--p 2] [--logm 16] [--ef 16] [--r 16] [--iters 200]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_sddmm_amd import api as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--alg", default="15d_sparse")
ap.add_argument("--p", type=int, default=2)
ap.add_argument("--c", type=int, default=1)
ap.add_argument("--logm", type=int, default=16)
ap.add_argument("--ef", type=int, default=16)
ap.add_argument("--r", type=int, default=16)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--backend", default=None)
a = ap.parse_args()
name = H.load_backend(a.backend)
assert a.backend or name == "hip-gfx950"


def body(w):
    sp = H.SpmatLocal.load_tuples(w, False, a.logm, a.ef)
    op = H.DistributedSparse(w, a.alg, sp, a.r, a.c)
    sp.free()
    A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
    S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
    call = lambda: op.fusedSpMM(A, B, S, buf, H.AMAT)  # noqa: E731
    call()
    w.sync()
    w.barrier()
    out = None
    try:
        if w.rank == 0:
            w.set_solo(True)
            for _ in range(10):
                call()
            w.sync()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                call()
            enq = (time.perf_counter() - t0) / a.iters
            w.sync()
            wall = (time.perf_counter() - t0) / a.iters
            op.kernel_profile(1)
            for _ in range(a.iters):
                call()
            w.sync()
            kms, launches = op.kernel_profile(0)
            out = (wall * 1e6, enq * 1e6, kms / a.iters * 1e3, launches // a.iters)
    finally:
        if w.rank == 0:
            w.set_solo(False)
        w.barrier()
    for x in (A, B, S, buf):
        x.free()
    op.free()
    return out


wall, enq, kern, launches = H.run_spmd(a.p, body)[0]
print("%s p=%d c=%d ER 2^%d ef %d R=%d: %.1f us per fused call (host enqueue %.1f us), %.1f us in %d local kernel launches"
      % (a.alg, a.p, a.c, a.logm, a.ef, a.r, wall, enq, kern or 0.0, launches))
