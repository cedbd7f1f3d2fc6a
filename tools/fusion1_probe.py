"""Development tool: ONE rank's share of the 1.5D dense shift by replication reuse (15d_fusion1), alone on the GPU — its SDDMM, its SpMM
and the pair, in the block-by-block ring of rounds 1-5 (HNH_FUSION1_MESH=0: two-half accumulator ring) and on the mesh (round 6:
row-merged layout, row-range SDDMM passes, mesh reduce-scatter SpMM), for a few chunk shapes.

p logical ranks (loopback transport, one device) build the operator at full size and run one collective call; then rank 0 repeats the
call BY ITSELF in solo replay (World::set_solo: every message it would receive is replaced by a device copy of what it would send — same
bytes, streams and events), so what is timed is the rank's kernel sequence plus the HBM side of its exchange.  What it cannot show: links.

    python tools/fusion1_probe.py [--p 8] [--r 128] [--chunks "1,2,2,2,1,1;1,2,1;1"] [--iters 5]"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--p", type=int, default=8)
ap.add_argument("--logm", type=int, default=20)
ap.add_argument("--ef", type=int, default=96)
ap.add_argument("--r", type=int, default=128)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--chunks", default="1,2,2,2,1,1;1,2,1;1", help="chunk shapes of the mesh variant, ';'-separated (HNH_MESH_TAPER lists)")
ap.add_argument("--backend", default=None)
a = ap.parse_args()
from distributed_sddmm_amd import api as H  # noqa: E402

name = H.load_backend(a.backend)
assert a.backend or name == "hip-gfx950"
HBM = 8.0e12


def measure():
    def body(w):
        sp = H.SpmatLocal.load_tuples(w, False, a.logm, a.ef)
        gnnz = sp.info()["dist_nnz"]
        op = H.DistributedSparse(w, "15d_fusion1", sp, a.r, 1)
        sp.free()
        A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
        S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
        calls = {"sddmm": lambda: op.sddmmA(A, B, S, buf), "spmm": lambda: op.spmmA(A, B, S), "pair": lambda: op.fusedSpMM(A, B, S, buf, H.AMAT)}
        for c in calls.values():
            c()
        w.sync()
        w.barrier()
        out = None
        try:
            if w.rank == 0:
                w.set_solo(True)
                out = {}
                for k, c in calls.items():
                    c()
                    w.sync()
                    best = None
                    for _ in range(2):
                        t0 = time.perf_counter()
                        for _ in range(a.iters):
                            c()
                        w.sync()
                        t = (time.perf_counter() - t0) * 1e3 / a.iters
                        best = t if best is None else min(best, t)
                    op.kernel_profile(1)
                    for _ in range(a.iters):
                        c()
                    w.sync()
                    kms, launches = op.kernel_profile(0)
                    out[k] = (best, kms / a.iters, launches // a.iters)
        finally:
            if w.rank == 0:
                w.set_solo(False)
            w.barrier()
        for x in (A, B, S, buf):
            x.free()
        op.free()
        return out, gnnz
    res = H.run_spmd(a.p, body)
    return res[0]


m = 1 << a.logm
variants = [("ring, two halves (HNH_FUSION1_MESH=0)", {"HNH_FUSION1_MESH": "0"}, None)]
for spec in a.chunks.split(";"):
    variants.append(("mesh, chunk heights %s" % spec, {"HNH_FUSION1_MESH": "1"}, spec))
print("15d_fusion1, one rank of %d alone (solo replay), ER 2^%d edge factor %d, R=%d; byte models per rank = global / p: SDDMM nnz(8R+20)+8RM, "
      "SpMM nnz(8R+12)+16RM, pair = their sum" % (a.p, a.logm, a.ef, a.r))
for vname, env, spec in variants:
    for k in ("HNH_FUSION1_MESH", "HNH_MESH_TAPER", "HNH_MESH_CHUNKS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    if spec:
        os.environ["HNH_MESH_TAPER"] = spec
    out, gnnz = measure()
    by = {"sddmm": (gnnz * (8 * a.r + 20) + 8 * a.r * m) / a.p, "spmm": (gnnz * (8 * a.r + 12) + 16 * a.r * m) / a.p}
    by["pair"] = by["sddmm"] + by["spmm"]
    print("  %s" % vname)
    for k in ("sddmm", "spmm", "pair"):
        wall, kms, launches = out[k]
        print("    %-6s wall %6.3f ms  row kernels %6.3f ms in %2d launches  frac_kernel %.3f  frac_wall %.3f  outside the row kernels %.3f ms" % (
            k, wall, kms, launches, by[k] / (kms * 1e-3) / HBM if kms > 0 else 0.0, by[k] / (wall * 1e-3) / HBM, wall - kms))
