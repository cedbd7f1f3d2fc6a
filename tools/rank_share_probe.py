"""Development tool: ONE rank's share of BASELINE config 3, alone on the GPU.

p logical ranks (loopback transport, one device) build the 1.5D dense-shift operator at full size; the moving operand
is put on hold (`hold_moving_operand`, the ALS hint) and one collective fusedSpMM fills every rank's landing buffers.
After that a held operand needs no transfer, so rank 0 can repeat the call BY ITSELF while the other ranks wait at a
barrier: what is timed is exactly the kernel sequence a rank of a p-GPU job runs per fused call (local block + the
fetched blocks, chunk by chunk), with the device to itself — the per-rank roofline fraction of the multi-block path.

    python tools/rank_share_probe.py [--p 8] [--chunks 4] [--r 128] [--iters 10]
"""
import argparse
import os
import sys
import time

# N logical ranks = 2 N streams on ONE device: HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams
# that share a queue serialise — give every stream its own (has to be set before the HIP runtime starts)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--p", type=int, default=8)
ap.add_argument("--c", type=int, default=1)
ap.add_argument("--logm", type=int, default=20)
ap.add_argument("--ef", type=int, default=96)
ap.add_argument("--r", type=int, default=128)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--backend", default=None, help="kernel library to load (default: the HIP library; tests pass the CPU test double)")
ap.add_argument("--chunks", default="", help="comma list of HNH_MESH_CHUNKS values to sweep (default: the library default)")
a = ap.parse_args()

from distributed_sddmm_amd import api as H  # noqa: E402

name = H.load_backend(a.backend)
assert a.backend or name == "hip-gfx950"


def body(w):
    sp = H.SpmatLocal.load_tuples(w, False, a.logm, a.ef)
    op = H.DistributedSparse(w, "15d_fusion2", sp, a.r, a.c)
    sp.free()
    A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
    S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
    op.hold_moving_operand(B)
    op.walk_windows_when_held(2)  # the held blocks are resident, but walk the chunk windows one by one, as a call whose chunks arrive one by one does
    op.fusedSpMM(A, B, S, buf, H.AMAT)  # collective: fills the landing buffers
    w.sync()
    w.barrier()
    out = None
    if w.rank == 0:
        info = op.info()
        A.fill(0.001)
        op.fusedSpMM(A, B, S, buf, H.AMAT)
        w.sync()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            op.fusedSpMM(A, B, S, buf, H.AMAT)
        w.sync()
        wall = (time.perf_counter() - t0) / a.iters
        op.kernel_profile(1)
        for _ in range(a.iters):
            op.fusedSpMM(A, B, S, buf, H.AMAT)
        w.sync()
        kms, launches = op.kernel_profile(0)
        rows = info["localArows"] * a.c
        alg = info["nS"] * (8 * a.r + 24) + 16 * a.r * rows
        out = (wall, kms / a.iters, launches // a.iters, alg, info["nS"], rows)
    w.barrier()
    op.hold_moving_operand(None)
    for x in (A, B, S, buf):
        x.free()
    op.free()
    return out


for q in (a.chunks.split(",") if a.chunks else [""]):
    if q:
        os.environ["HNH_MESH_CHUNKS"] = q
    res = H.run_spmd(a.p, body)[0]
    wall, kms, launches, alg, nnz, rows = res
    kms = kms or wall * 1e3  # the CPU test double has no event timing
    print("p=%d c=%d chunks=%s R=%d: rank 0 alone: %.3f ms wall per fused call, %.3f ms in %d kernel launches; local nnz %d, rows %d, "
          "algorithmic %.3f GB -> %.2f TB/s = %.1f%% of 8 TB/s (kernel time), %.1f%% (wall)"
          % (a.p, a.c, q or "default", a.r, wall * 1e3, kms, launches, nnz, rows, alg / 1e9, alg / (kms * 1e-3) / 1e12,
             100 * alg / (kms * 1e-3) / 8e12, 100 * alg / wall / 8e12), flush=True)
