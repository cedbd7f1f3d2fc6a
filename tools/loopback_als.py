"""Development tool: one ALS half-step (cg_optimizer(Amat, 10) = 12 fused passes) of config 5's shape under the 1.5D
dense-shift schedule on p logical ranks sharing ONE GPU (loopback transport) — with and without the
hold_moving_operand hint (HNH_NO_HOLD=1 makes Distributed_Sparse ignore it).  Not a multi-GPU measurement: it shows the
control flow at full size and how many bytes the hint takes off the transport."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_sddmm_amd import api as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--p", type=int, default=8); ap.add_argument("--logm", type=int, default=20); ap.add_argument("--ef", type=int, default=96)
ap.add_argument("--r", type=int, default=128); ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
assert H.load_backend(None) == "hip-gfx950"


def body(w):
    sp = H.SpmatLocal.load_tuples(w, False, a.logm, a.ef)
    op = H.DistributedSparse(w, "15d_fusion2", sp, a.r, 1)
    sp.free()
    als = H.DistributedALS(op, True)
    als.initializeEmbeddings()
    als.cg_optimizer(H.AMAT, 1); w.sync(); w.barrier()
    op.reset_performance_timers()
    t0 = time.perf_counter()
    als.cg_optimizer(H.AMAT, a.iters)
    w.sync(); w.barrier()
    dt = time.perf_counter() - t0
    stats = op.json_perf_statistics()
    res = als.computeResidual()
    als.free(); op.free()
    return dt, stats, res


res = H.run_spmd(a.p, body)
dt = max(r[0] for r in res)
print("p=%d hold=%s: cg_optimizer(Amat, %d) %.1f ms = %.2f ms per fused pass; rank-0 timers %s; residual %.6e"
      % (a.p, "off" if os.environ.get("HNH_NO_HOLD") else "on", a.iters, dt * 1e3, dt * 1e3 / (a.iters + 2), res[0][1], res[0][2]))
