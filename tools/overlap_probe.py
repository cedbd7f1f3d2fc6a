"""Development tool: does the event protocol of the chunked mesh fetch HIDE a transfer?  Measured on ONE GPU.

Set-up as tools/rank_share_probe.py: p logical ranks build BASELINE config 3 (1.5D dense shift, local kernel fusion), the
moving operand is held so that every rank's landing buffer is filled once, then rank 0 repeats the fused call alone.  The
difference: with HNH_PACE_LINK_GBPS the communication stream is held, chunk by chunk, for exactly as long as that chunk
would need to cross one xGMI link at the given rate (hnh_stream_delay_us: an idle-spinning wave, no copies, no HBM
traffic), and the compute stream waits for each chunk's event as in a fetching call.  So the call is timed against
transfers of a KNOWN duration with the rank's kernels alone on the device — which the loopback transport cannot show,
because there the copies of 8 logical ranks compete with the kernels for the same HBM.

The link rate is MODELLED (a parameter), not measured: this is evidence about the stream/event protocol
(dense_shift_15d.hpp: fetch_into_landing / walk_merged), not a scaling number.

    python tools/overlap_probe.py [--p 8] [--chunks 1,2,4] [--pace 40,60,75] [--r 128] [--iters 10]

Per (Q, rate) it prints the measured time per call next to
    serial   = T_fetch + T_kernels                     (what "kernel -> Sendrecv -> Barrier", 15D_dense_shift.hpp:343-356, costs)
    model    = the pipeline played through: chunk q (height (1, 2, .., 2, 1) / (2Q - 2) of a block) lands at the running sum of the
               chunk transfer times, window q's kernel starts when it has landed and the previous kernel has ended, the own
               block's kernel runs first; for T_own = 0 this is max(T_fetch, T_rem) + min(T_fetch, T_rem) / (2Q - 2)
with T_fetch = one block / rate (the n-1 blocks use n-1 links at once), T_own / T_rem = kernel time of the own block and of
the windowed passes over the fetched blocks (HIP events, unpaced run).  Q = 1 has no chunks: model = T_fetch + T_rem, the
own block's kernel being the only thing that overlaps."""
import argparse
import os
import sys
import time

# N logical ranks = 2 N streams on ONE device: HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams
# that share a queue serialise — give every stream its own (has to be set before the HIP runtime starts)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the paced stand-ins live in the measurement build of the host library only
os.environ.setdefault("HNH_HOST_LIB_DEV", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "distributed_sddmm_amd", "lib", "libhnh_host_aids.so"))

ap = argparse.ArgumentParser()
ap.add_argument("--p", type=int, default=8)
ap.add_argument("--logm", type=int, default=20)
ap.add_argument("--ef", type=int, default=96)
ap.add_argument("--r", type=int, default=128)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--backend", default=None, help="kernel library to load (default: the HIP library; tests pass the CPU test double)")
ap.add_argument("--chunks", default="1,2,4")
ap.add_argument("--pace", default="40,60,75", help="modelled GB/s per xGMI link and direction")
ap.add_argument("--tapers", default="", help="semicolon list of HNH_MESH_TAPER weight lists to measure besides --chunks, e.g. \"6,5,4,3,2,1;3,4,4,3,2,1,1\"")
ap.add_argument("--copy-wgs", default="0", help="comma list; > 0: the paced stand-in also MOVES the bytes (HNH_PACE_COPY: chunk q of the own block "
                "read once per peer, written to each peer's place in the landing buffer, by this many throttled workgroups per link), so "
                "the rank's kernels also meet the HBM traffic (896 MiB in + 896 MiB out per call) and the workgroups of a real exchange")
a = ap.parse_args()

os.environ.pop("HNH_PACE_LINK_GBPS", None)
from distributed_sddmm_amd import api as H  # noqa: E402

name = H.load_backend(a.backend)
assert a.backend or name == "hip-gfx950"
paces = [float(x) for x in a.pace.split(",") if x]


def body(w):
    sp = H.SpmatLocal.load_tuples(w, False, a.logm, a.ef)
    op = H.DistributedSparse(w, "15d_fusion2", sp, a.r, 1)
    sp.free()
    A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
    S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
    op.hold_moving_operand(B)
    op.walk_windows_when_held(1)  # windows are walked (and the paced stand-ins waited for) although the blocks are resident
    op.fusedSpMM(A, B, S, buf, H.AMAT)  # collective: fills the landing buffers
    w.sync()
    w.barrier()
    out = None
    if w.rank == 0:
        info = op.info()

        def timed():
            A.fill(0.001)
            op.fusedSpMM(A, B, S, buf, H.AMAT)
            w.sync()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                op.fusedSpMM(A, B, S, buf, H.AMAT)
            w.sync()
            return (time.perf_counter() - t0) / a.iters * 1e3

        unpaced = timed()
        # kernel time per launch position (own block first, then the windows), HIP events on the compute stream
        op.kernel_profile(1)
        op.fusedSpMM(A, B, S, buf, H.AMAT)
        w.sync()
        k_all, launches = op.kernel_profile(0)
        # the own block alone: the first launch of a call; measured by a call with the profile on and the fetched block skipped
        # is not available through the API, so split by nonzero share (own block = 1/p of the rank's nonzeros, same kernel)
        t_own = k_all / a.p if a.p > 1 else k_all
        t_rem = k_all - t_own
        rows = []
        for g in paces:
            os.environ["HNH_PACE_LINK_GBPS"] = repr(g)
            rows.append((g, timed()))
        os.environ.pop("HNH_PACE_LINK_GBPS", None)
        out = (unpaced, k_all or unpaced, launches, t_own or unpaced / a.p, t_rem or unpaced * (a.p - 1) / a.p, rows, info["localBrows"])
    w.barrier()
    op.hold_moving_operand(None)
    for x in (A, B, S, buf):
        x.free()
    op.free()
    return out


print("one rank of p=%d alone on the GPU, ER 2^%d ef %d, R=%d; link rate is a MODEL parameter (paced communication stream, no copies)" %
      (a.p, a.logm, a.ef, a.r), flush=True)
shapes = [(q, None) for q in a.chunks.split(",") if q] + [(None, t) for t in a.tapers.split(";") if t]
for (q, taper), wgs in [(sh, w) for w in a.copy_wgs.split(",") for sh in shapes]:
    if taper is None:
        os.environ["HNH_MESH_CHUNKS"] = q
        os.environ.pop("HNH_MESH_TAPER", None)
        Q = int(q)
        weights = [1.0] if Q == 1 else [(1.0 if i in (0, Q - 1) else 2.0) for i in range(Q)]
    else:
        os.environ["HNH_MESH_TAPER"] = taper
        weights = [float(x) for x in taper.split(",")]
        Q = len(weights)
    if int(wgs) > 0:
        os.environ["HNH_PACE_COPY"] = wgs
        print("-- the paced transfers also move their bytes: %s throttled workgroups per link --" % wgs, flush=True)
    else:
        os.environ.pop("HNH_PACE_COPY", None)
    unpaced, k_all, launches, t_own, t_rem, rows, brows = H.run_spmd(a.p, body)[0]
    block_bytes = brows * a.r * 8
    label = "Q=%d" % Q if taper is None else "taper %s" % taper
    print("%s: unpaced call %.3f ms (%d launches, %.3f ms of kernels: own block ~%.3f, fetched blocks ~%.3f)" % (label, unpaced, launches, k_all, t_own, t_rem), flush=True)
    for g, t in rows:
        tf = block_bytes / (g * 1e9) * 1e3
        serial = tf + unpaced
        # pipeline simulation with the tapered chunk heights (1, 2, .., 2, 1) / (2Q - 2): chunk q lands at the running sum of its
        # transfer times; window q's kernel starts when chunk q has landed and the previous kernel is done
        frac = [x / sum(weights) for x in weights]
        t_comm, t_comp = 0.0, t_own
        for f in frac:
            t_comm += tf * f
            t_comp = max(t_comp, t_comm) + t_rem * f
        model = t_comp
        print("   %5.1f GB/s/link: T_fetch %.3f ms | measured %.3f ms | serial (fetch + kernels) %.3f ms | model %.3f ms | hidden %.3f ms = %.0f%% of "
              "min(T_fetch, T_kernels)" % (g, tf, t, serial, model, serial - t, 100.0 * (serial - t) / min(tf, unpaced)), flush=True)
