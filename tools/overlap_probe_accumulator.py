"""Development tool: what the two-half accumulator ring buys (dense_shift_15d.hpp: ring_readwrite_halves) — the SpMM of the
1.5D dense shift under replication reuse (15d_fusion1), whose MOVING buffer is the output accumulator, so that a step's
shift cannot start before the step's kernel has finished.  p logical ranks share ONE GPU (loopback transport); every
message is followed by a hold of the receiving stream for as long as it would need to cross one xGMI link at a modelled rate
(HNH_PACE_LINK_GBPS in the measurement build of the host library), so the event protocol is timed against transfers of a
known duration.  Compared: HNH_ACC_HALVES=0 (the reference's kernel -> shift -> kernel, 15D_dense_shift.hpp:343-356), 1
(kernel on one half of the rows while the other half travels) and, round 6, the MESH REDUCE-SCATTER of the row-merged layout
(HNH_FUSION1_MESH=1, the default: every rank computes its partial result for every block chunk by chunk, each chunk's n - 1 pieces
travel to their owners over n - 1 links at once, the owners add what they receive).  The link rate is a MODEL parameter; the p ranks' kernels
share the GPU's HBM, so per-rank kernel times are p times what a rank alone would see — both variants alike.

    python tools/overlap_probe_accumulator.py [--p 4] [--logm 20] [--r 128] [--gbps 0,40,60,100]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HNH_HOST_LIB_DEV", os.path.join(ROOT, "distributed_sddmm_amd", "lib", "libhnh_host_aids.so"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")  # p ranks x (compute + communication) streams must not share hardware queues
ap = argparse.ArgumentParser()
ap.add_argument("--p", type=int, default=4)
ap.add_argument("--logm", type=int, default=20)
ap.add_argument("--ef", type=int, default=96)
ap.add_argument("--r", type=int, default=128)
ap.add_argument("--alg", default="15d_fusion1")
ap.add_argument("--op", choices=["spmm", "sddmm", "fused"], default="spmm", help="spmmA (the accumulator travels), sddmmA (the moving operand is read-only: "
                "relay ring of whole blocks against the chunked mesh fetch with row-range passes), or the pair (fusedSpMM of this schedule)")
ap.add_argument("--gbps", default="0,40,60,100", help="modelled GB/s per link and direction; 0 = the loopback copies alone")
ap.add_argument("--calls", type=int, default=3)
ap.add_argument("--iters", type=int, default=None, help="alias of --calls")
ap.add_argument("--backend", default=None, help="kernel library to load (default: the HIP library; tests pass the CPU test double)")
a = ap.parse_args()
if a.iters:
    a.calls = a.iters
from distributed_sddmm_amd import api as H  # noqa: E402

name = H.load_backend(a.backend)
assert a.backend or name == "hip-gfx950"
rates = [float(x) for x in a.gbps.split(",")]
# the variants of the SpMM whose moving buffer is the accumulator: the reference's ring, the ring in two row halves (round 4), and the
# mesh reduce-scatter of the row-merged layout (round 6; 15d_fusion1 only — the 2.5D schedule has no mesh form)
VARIANTS = [("kernel->shift", {"HNH_FUSION1_MESH": "0", "HNH_ACC_HALVES": "0"}), ("two halves", {"HNH_FUSION1_MESH": "0", "HNH_ACC_HALVES": "1"})]
if a.alg == "15d_fusion1":
    VARIANTS.append(("mesh reduce-scatter", {"HNH_FUSION1_MESH": "1"}))
results = {}
for vname, venv in VARIANTS:
    for k in ("HNH_FUSION1_MESH", "HNH_ACC_HALVES"):
        os.environ.pop(k, None)
    os.environ.update(venv)
    os.environ.pop("HNH_PACE_LINK_GBPS", None)

    def body(w):
        sp = H.SpmatLocal.load_tuples(w, False, a.logm, a.ef)
        op = H.DistributedSparse(w, a.alg, sp, a.r, 1)
        sp.free()
        A, B, S, buf = op.like_A_matrix(0.001), op.like_B_matrix(0.001), op.like_S_values(1.0), op.like_S_values(0.0)
        call = {"spmm": lambda: op.spmmA(A, B, S), "sddmm": lambda: op.sddmmA(A, B, S, buf), "fused": lambda: op.fusedSpMM(A, B, S, buf, H.AMAT)}[a.op]
        out = {}
        for g in rates:
            w.sync(); w.barrier()
            if w.rank == 0:
                if g > 0:
                    os.environ["HNH_PACE_LINK_GBPS"] = repr(g)
                else:
                    os.environ.pop("HNH_PACE_LINK_GBPS", None)
            w.barrier()
            call()
            w.sync(); w.barrier()
            t0 = time.perf_counter()
            for _ in range(a.calls):
                call()
            w.sync(); w.barrier()
            out[g] = (time.perf_counter() - t0) / a.calls * 1e3
        info = op.info()
        for x in (A, B, S, buf):
            x.free()
        op.free()
        return out, info

    res = H.run_spmd(a.p, body)
    results[vname] = {g: max(r[0][g] for r in res) for g in rates}
    info = res[0][1]
m = 1 << a.logm
block_mib = (m // a.p) * a.r * 8 / 2 ** 20
print("%s %s, ER 2^%d edge factor %d, R=%d, %d logical ranks on one GPU; a dense block is %.0f MiB; the ring shifts it %d times over ONE link, "
      "the mesh forms move %d blocks over %d links at once (a group of messages is paced as its longest message)" % (
          a.alg, a.op, a.logm, a.ef, a.r, a.p, block_mib, a.p, a.p - 1, a.p - 1))
names = [v[0] for v in VARIANTS]
print("%-28s " % "GB/s per link (modelled)" + " ".join("%19s" % v for v in names) + "   last vs two halves   one block at that rate")
for g in rates:
    ts = [results[v][g] for v in names]
    print("%-28s " % ("loopback copies only" if g == 0 else "%.0f" % g) + " ".join("%16.2f ms" % t for t in ts) +
          "   %+17.1f %%   %s" % (100.0 * (ts[-1] - ts[1]) / ts[1], "-" if g == 0 else "%.2f ms" % (block_mib * 2 ** 20 / (g * 1e9) * 1e3)))
