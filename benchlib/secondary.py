"""One GPU: the rest of the reference's harness, bounded, outside the timed region — every entry with its own byte model, fraction of
8 TB/s and closed-form check.  None of them touches the headline."""
import argparse
import os
import time

from . import checks, search, timed
from .common import GAT_LAYERS, HBM_PEAK, Workload, fused_bytes, keyed
from .guards import Watchdog


def secondary(args, b):
    """The rest of the reference's harness on one GPU, each entry bounded to a few seconds and carrying its own byte model and
    check: (i) R-MAT (hub rows), (ii) config 4's schedule — 2.5D dense-replicate, p = 8, c = 2, R = 256 — on 8 logical ranks sharing
    this GPU through the loopback transport, (iii) one ALS-CG step, (iv) the GAT forward pass, (v) fused / SDDMM / SpMM at R = 8, 16, 256.
    Every failure is recorded in its entry; none of them touches the headline."""
    import numpy as np
    H, torch = b.H, b.torch
    world = b.world()
    out = []
    small = os.environ.get("HNH_BENCH_SECONDARY_SMALL") is not None  # (the CPU test of this function: same code, toy sizes)

    def entry(ident, name, fn):
        """`ident` = the entry's row in the printed line's table (benchlib/line.py); the entry itself goes to the record file"""
        t0 = time.perf_counter()
        try:
            e = fn()
        except Exception as ex:  # noqa: BLE001
            e = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
        e = dict({"id": ident, "workload": name}, **e)
        e["seconds"] = round(time.perf_counter() - t0, 1)
        out.append(e)

    def kernel_time(op, fn, calls):
        """event-bracketed device time of the local kernels of `calls` invocations (ms per invocation), after one warm-up"""
        fn()
        world.sync()
        op.kernel_profile(1)
        for _ in range(calls):
            fn()
        world.sync()
        ms, launches = op.kernel_profile(0)
        return ms / calls, max(1, launches // calls)

    def call_time(fn, calls):
        """wall time per WHOLE operator call (ms), device drained on both sides: the local kernels plus whatever the operation does around
        them (value copies, zero fills, the closing Hadamard of an SDDMM)"""
        fn()
        world.sync()
        t0 = time.perf_counter()
        for _ in range(calls):
            fn()
        world.sync()
        return (time.perf_counter() - t0) * 1e3 / calls

    def frac_of(bytes_alg, ms):
        return bytes_alg / (ms * 1e-3) / HBM_PEAK

    # (v) other widths on the headline matrix and operator (the structure plans and blocks are the headline's)
    if args.app == "vanilla" and b.op is not None:
        op, m, nnz = b.op, b.m, b.nnz
        host = b.wl.host_nonzeros(H)
        for r in (8, 16, 128, 256, 512):
            def widths(r=r):
                op.setRValue(r)
                A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
                S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
                res = {"R": r}
                if r == 512:
                    res["note"] = ("wide operands: the un-fused SDDMM / SpMM run as 128-column slabs (each slab = the R = 128 pass with row pitch R and its own "
                                   "Infinity-Cache panels); the fused pass needs the whole dot product and stays one wide pass at the DRAM copy rate")
                if r == 8:
                    res["note"] = ("a 64-byte dense row costs a whole 128-byte request between L2 and the fabric, and the row kernels are bound by the "
                                   "number of such requests (about 55 G/s at every width, DESIGN section 3.2): the byte model cannot exceed ~0.55 here")
                try:
                    ms, _ = kernel_time(op, lambda: op.fusedSpMM(A, B, S, buf, H.AMAT), 3)
                    res["fused"] = {"ms": ms, "algorithmic_bytes": fused_bytes(nnz, r, m), "frac": frac_of(fused_bytes(nnz, r, m), ms)}
                    ms, _ = kernel_time(op, lambda: op.sddmmA(A, B, S, buf), 3)
                    by = nnz * (8 * r + 20) + 8 * r * m
                    res["sddmm"] = {"ms": ms, "algorithmic_bytes": by, "frac": frac_of(by, ms), "call_ms": call_time(lambda: op.sddmmA(A, B, S, buf), 3)}
                    ms, _ = kernel_time(op, lambda: op.spmmA(A, B, S), 3)
                    by = nnz * (8 * r + 12) + 16 * r * m
                    res["spmm"] = {"ms": ms, "algorithmic_bytes": by, "frac": frac_of(by, ms), "call_ms": call_time(lambda: op.spmmA(A, B, S), 3)}
                    res["borrowed_value_arrays"] = dict(zip(("spmm_lent", "spmm_copied", "sddmm_in_place", "sddmm_hadamard"), op.borrow_stats()))
                    if host is not None:  # closed forms with the keyed operands: sddmm(i,j) = a_i b_j (u.v); spmm[i,k] = v_k sum_j b_j
                        grows, gcols = host
                        a_key, b_key = keyed(np.arange(m), 1), keyed(np.arange(m), 2)
                        u_key, v_key = keyed(np.arange(r), 3), keyed(np.arange(r), 4)
                        A.upload(a_key[:, None] * u_key[None, :])
                        B.upload(b_key[:, None] * v_key[None, :])
                        op.sddmmA(A, B, S, buf)
                        got = buf.download()
                        w = float(np.dot(u_key, v_key))
                        # the block's value order is row-major (one block on one rank), like the generator's
                        e1 = float(np.max(np.abs(got - w * a_key[grows] * b_key[gcols])) / (w * 2.25))
                        op.spmmA(A, B, S)
                        gotA = A.download()
                        want = np.bincount(grows, weights=b_key[gcols], minlength=m)
                        e2 = float(np.max(np.abs(gotA - want[:, None] * v_key[None, :])) / float(want.max() * v_key.max()))
                        res["check"] = {"what": "sddmmA and spmmA from keyed operands against a_i b_j (u.v) and v_k sum_{j in row i} b_j",
                                        "rel_err_sddmm": e1, "rel_err_spmm": e2, "ok": bool(e1 <= 1e-11 and e2 <= 1e-11)}
                finally:
                    for x in (A, B, S, buf):
                        x.free()
                return res
            entry("width_R%d" % r, "the headline matrix at R=%d: fused / SDDMM / SpMM kernels through the operator (ms = device time of the local kernels, "
                  "call_ms = the whole sddmmA / spmmA call)" % r, widths)
        b.op.setRValue(args.r)

    # (iii) one ALS step, (iv) the GAT forward pass — on the headline's matrix and transport, a fresh operator each
    def app_entry(app):
        def run_it():
            sub = timed.Bench(argparse.Namespace(**dict(vars(args), app=app, steps=1, warmup=0, no_check=False)), H, torch, None, 0, 1, Watchdog(0, 0, False), b.wl)
            sub.transports = {"single": dict(b.transports["single"])}
            sub.nnz, sub.m = b.nnz, b.m
            small = None
            if app == "gat":  # the forward pass's buffers are 2^logm x 1536: a bounded instance (2^18 vertices, edge factor 32 as in profiles/)
                small = Workload(b.wl.kind if b.wl.kind != "mtx" else "er", min(args.logm, 18), min(args.edge_factor, 32))
                sub.wl = small
                sub.transports["single"]["sp"] = None
            try:
                sub.build(("single", 1, "none", None))
                ms = search.quick_time(sub, 1)
                chk = checks.check_app(sub)
                info = sub.op.info()
                if app == "als":
                    by = 24 * fused_bytes(sub.nnz, args.r, sub.m)
                    what = "24 fused calls (2 half-steps x (2 + 10 CG iterations)) with the CG updates in the row epilogue"
                else:
                    by = sum(h * fused_bytes(sub.nnz, f, sub.m) for _, f, h in GAT_LAYERS)
                    what = "14 fused heads (SDDMM -> LeakyReLU -> SpMM -> ReLU delivery) + 14 fp64 MFMA GEMMs, the product of head j + 1 on a second compute stream beside the attention pass of head j"
                res = {"ms": ms, "what": what, "nnz": sub.nnz, "M": sub.m, "R": info["R"], "algorithmic_bytes_fused_calls": by,
                       "frac_whole_step": frac_of(by, ms), "check": chk}
                if app == "als":
                    # what the fused-call model leaves out: the 20 CG iterations' own row streams — x and r read and written, p written
                    # (als_conjugate_gradients.cpp:112-139) — which ride in the fused call's row epilogue here: 5 x 8 R M bytes each
                    # (profiles/r05_als_step_kernel_stats.csv: the launch that carries them takes 8.51 ms against 7.39 ms without)
                    by_cg = by + 20 * 5 * 8 * args.r * sub.m
                    res.update({"algorithmic_bytes_with_cg_row_streams": by_cg, "frac_with_cg_row_streams": frac_of(by_cg, ms)})
                else:
                    # what the fused-call model leaves out: the 14 dense products X * W_h — their compulsory bytes (X read, the product written;
                    # W is small) and their fp64 matrix-core work, which shares the chip with the attention passes
                    gemm_bytes = sum(h * 8 * sub.m * (fin + f) for fin, f, h in GAT_LAYERS)
                    gemm_flops = sum(h * 2 * sub.m * fin * f for fin, f, h in GAT_LAYERS)
                    res.update({"algorithmic_bytes_with_gemm_operands": by + gemm_bytes, "frac_with_gemm_operands": frac_of(by + gemm_bytes, ms),
                                "gemm_tflops_over_the_whole_pass": gemm_flops / (ms * 1e-3) / 1e12, "gemm_flops": gemm_flops,
                                "note": "two roofline-bound kernels side by side: the attention passes alone take ~37 ms (HBM gathers), the products alone ~23 ms "
                                        "(84 % of the fp64 matrix peak); what they share is the L2 request stream (DESIGN section 4)"})
                return res
            finally:
                sub.free_current()
                if small is not None and sub.transports["single"]["sp"] is not None:
                    sub.transports["single"]["sp"].free()
        return run_it

    entry("als_step", "one alternating ALS-CG step (run_cg(1), benchmark_dist.cpp:134-137) on the headline matrix, R=%d" % args.r, app_entry("als"))
    entry("gat_forward", "GAT forward pass (layers of benchmark_dist.cpp:88-94) on a bounded instance of the workload", app_entry("gat"))

    # (i) R-MAT with hub rows, fused at the headline width
    def rmat():
        wl = Workload("rmat", 9 if small else 20, 8 if small else 44)
        sub = timed.Bench(argparse.Namespace(**dict(vars(args), app="vanilla", steps=1, warmup=0, no_check=False)), H, torch, None, 0, 1, Watchdog(0, 0, False), wl)
        sub.transports = {"single": dict(b.transports["single"], sp=None)}
        try:
            sub.build(("single", 1, "none", None))
            ms, launches = kernel_time(sub.op, sub.step, 5)
            chk = checks.check(sub)
            deg = np.bincount(wl.host_nonzeros(H)[0], minlength=sub.m)
            by = fused_bytes(sub.nnz, args.r, sub.m)
            return {"ms": ms, "nnz": sub.nnz, "M": sub.m, "R": args.r, "longest_row": int(deg.max()), "algorithmic_bytes": by, "frac": frac_of(by, ms),
                    "note": "hot columns are cache-resident on a skewed graph: the gather model can exceed 100 %",
                    "check": {k: chk[k] for k in ("rel_err", "rows_checked", "ok")}}
        finally:
            sub.free_current()
            if sub.transports["single"]["sp"] is not None:
                sub.transports["single"]["sp"].free()
    entry("rmat_fused", "R-MAT 2^%d, edge factor %d (hub rows: long-row pass with ordered reduction), fused R=%d" % ((9, 8, args.r) if small else (20, 44, args.r)), rmat)

    # (ii) config 4's schedule and width on 8 logical ranks that share this GPU (loopback transport: device-to-device copies)
    def cfg4():
        logm, ef, r = (8, 8, 32) if small else (18, 32, 256)
        rows, cols = H.generate_rmat(logm, (1 << logm) * ef)
        m = 1 << logm
        a_key, b_key = keyed(np.arange(m), 1), keyed(np.arange(m), 2)
        u_key, v_key = keyed(np.arange(r), 3), keyed(np.arange(r), 4)
        want_row = float(np.dot(u_key, v_key)) * a_key * np.bincount(rows, weights=b_key[cols] ** 2, minlength=m)

        def body(w):
            sp = H.SpmatLocal.from_global(w, m, m, rows, cols, np.ones(len(rows)))
            op = H.DistributedSparse(w, "25d_dense_replicate", sp, r, 2)
            sp.free()
            A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
            S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
            op.fusedSpMM(A, B, S, buf, H.AMAT)
            w.sync()
            w.barrier()
            t0 = time.perf_counter()
            for _ in range(3):
                op.fusedSpMM(A, B, S, buf, H.AMAT)
            w.sync()
            w.barrier()
            ms = (time.perf_counter() - t0) / 3 * 1e3

            def keyed_local(mat_mode, row_key, col_key):
                parts = []
                for top, left, rc, cc in op.submatrices(mat_mode):
                    blk = np.zeros((rc, cc))
                    keep = int(max(0, min(rc, m - top)))
                    blk[:keep] = row_key[top:top + keep, None] * col_key[None, left:left + cc]
                    parts.append(blk.reshape(-1))
                return np.concatenate(parts)
            A.upload(keyed_local(H.AMAT, a_key, u_key).reshape(A.shape))
            B.upload(keyed_local(H.BMAT, b_key, v_key).reshape(B.shape))
            S.fill(1.0)
            op.initial_shift(A, B, H.K_SDDMM_A)  # Cannon's skew (25D_cannon_dense.hpp:222-248)
            op.fusedSpMM(A, B, S, buf, H.AMAT)
            op.de_shift(A, B, H.K_SDDMM_A)
            w.sync()
            got, worst, off = A.download().reshape(-1), 0.0, 0
            for top, left, rc, cc in op.submatrices(H.AMAT):
                keep = int(max(0, min(rc, m - top)))
                blk = got[off:off + rc * cc].reshape(rc, cc)[:keep]
                off += rc * cc
                if keep:
                    worst = max(worst, float(np.max(np.abs(blk - want_row[top:top + keep, None] * v_key[None, left:left + cc]))))
            for x in (A, B, S, buf):
                x.free()
            op.free()
            return ms, worst
        res = H.run_spmd(8, body)
        ms = max(x[0] for x in res)
        err = max(x[1] for x in res) / float(want_row.max() * v_key.max())
        by = len(rows) * (16 * r + 44) + 16 * r * m  # the unfused pair this schedule runs (SURVEY 8d B_unfused)
        return {"ms": ms, "nnz": int(len(rows)), "M": m, "R": r, "schedule": "25d_dense_replicate p=8 c=2 (2 x 2 x 2), 8 logical ranks on ONE GPU, loopback copies",
                "algorithmic_bytes": by, "frac_all_ranks_on_this_one_gpu": frac_of(by, ms),
                "note": "all 8 ranks' kernels AND their device-to-device copies share this one GPU: a correctness-at-shape and cost figure, not a scaling "
                        "claim — one rank's efficiency with the GPU to itself is the 'rank share, config 4' entry below",
                "check": {"rel_err": err, "ok": bool(err <= 1e-11)}}
    entry("cfg4_8ranks_one_gpu", "config 4's shape, bounded: R-MAT 2^%d, edge factor %d, R=%d, 2.5D dense-replicate on 8 logical ranks" % ((8, 8, 32) if small else (18, 32, 256)), cfg4)

    # (vi) the one throughput the reference's own tree prints for this path: the p = 1 point of its weak-scaling
    # experiment 1 — `15d_sparse`, fused, 5 FusedMM calls in 0.8375 s on one Cori KNL node (ipdps_chart_generator.ipynb:564), at the size
    # its own throughput line implies (:573,589: 2^16 rows, 32 nonzeros per row, R = 256) — timed the reference's way (benchmark_dist.cpp:
    # 117-149: wall time of 5 calls) on this GPU.  Other hardware, a printed cell output, not a controlled comparison: context only.
    def knl_point():
        logm, ef, r = (8, 8, 32) if small else (16, 32, 256)
        wl = Workload("er", logm, ef)
        sub = timed.Bench(argparse.Namespace(**dict(vars(args), app="vanilla", alg="15d_sparse", r=r, steps=5, warmup=0, no_check=False)), H, torch, None, 0, 1,
                    Watchdog(0, 0, False), wl)
        sub.transports = {"single": dict(b.transports["single"], sp=None)}
        try:
            sub.build(("single", 1, "none", None))
            sub.step()
            world.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                sub.step()
            world.sync()
            s5 = time.perf_counter() - t0
            chk = checks.check(sub)
            by = sub.nnz * (16 * r + 44) + 16 * r * sub.m  # this schedule runs the SDDMM + SpMM pair (SURVEY 8d B_unfused)
            ref_s, ref_rate = 0.8375, (2 ** 16) * 32 * 256 * 5 / 0.8375
            res = {"seconds_for_5_fusedmm": s5, "ms": s5 / 5 * 1e3, "nnz": sub.nnz, "M": sub.m, "R": r, "schedule": "15d_sparse, fused, p = 1, c = 1",
                   "nnzR_per_s": sub.nnz * r * 5 / s5, "algorithmic_bytes": by, "frac": frac_of(by, s5 / 5 * 1e3),
                   "check": {k: chk[k] for k in ("rel_err", "rows_checked", "ok")}}
            if not small:
                res["reference_printed"] = {"seconds_for_5_fusedmm": ref_s, "nnzR_per_s": ref_rate, "hardware": "one Cori KNL node, 1 MPI rank",
                                            "source": "ipdps_chart_generator.ipynb:564 (time), :573,589 (the size its throughput line implies)",
                                            "speedup": ref_s / s5}
            return res
        finally:
            sub.free_current()
            if sub.transports["single"]["sp"] is not None:
                sub.transports["single"]["sp"].free()
    entry("knl_point", "the reference's printed weak-scaling point at p = 1: ER 2^%d, %d nonzeros per row, R=%d, 15d_sparse fused, 5 FusedMM timed the reference's way"
          % ((8, 8, 32) if small else (16, 32, 256)), knl_point)

    # ---- ONE RANK'S SHARE of the multi-GPU configurations (BASELINE configs 3, 4, 5) and config 1 as typed.  p logical ranks (host threads
    # over the loopback transport) build the operator at the configuration's size and run one collective call; then rank 0 repeats the
    # call BY ITSELF while its peers wait at a barrier, in one of two measurement modes of the library:
    #   held    hold_moving_operand + walk_windows_when_held: the fetched blocks are resident, the call walks own block + one windowed
    #           pass per chunk — the rank's KERNEL sequence only;
    #   solo    World::set_solo: every message the rank would receive is replaced by a device copy of what it would send (same bytes,
    #           streams, events) — the kernel sequence PLUS the HBM side of its exchange, overlapped as the schedule overlaps them.
    # What neither shows: the links.  Byte model per rank = the rank's own nonzeros and rows in the SURVEY 8(d) model (15d_fusion2: fused),
    # or the global unfused model / p where the schedule splits R (every nonzero is visited by several ranks with a slice of the columns).
    def solo_section(w, fn):
        """rank 0 alone: wall ms per call and event-bracketed kernel ms + launches per call; its peers wait at the closing barrier"""
        res = None
        try:
            if w.rank == 0:
                res = fn()
        finally:
            if w.rank == 0:
                w.set_solo(False)
            w.barrier()
        return res

    def time_alone(w, op, call, iters):
        call()
        w.sync()
        t0 = time.perf_counter()
        call()
        w.sync()
        one = time.perf_counter() - t0
        if one * iters < 0.02:  # calls of config 1's size (0.2 ms): a batch of five is one scheduler hiccup away from nonsense — 20 ms per batch
            iters = min(400, int(0.02 / max(one, 1e-6)) + 1)
        wall = None
        for _ in range(2 if one * iters >= 0.02 else 3):  # the better of two (three) batches: one host hiccup (a few ms on some boxes) would otherwise weigh on five calls
            t0 = time.perf_counter()
            for _ in range(iters):
                call()
            w.sync()
            t = (time.perf_counter() - t0) * 1e3 / iters
            wall = t if wall is None else min(wall, t)
        op.kernel_profile(1)
        for _ in range(iters):
            call()
        w.sync()
        kms, launches = op.kernel_profile(0)
        return wall, (kms / iters) or wall, launches // iters  # (the CPU test double has no event timing)

    def share_of_config3(p, chunk_spec):
        logm, ef, r = (9, 8, 16) if small else (args.logm, args.edge_factor, args.r)
        saved = {k: os.environ.get(k) for k in ("HNH_MESH_CHUNKS", "HNH_MESH_TAPER", "HNH_RING_MODE")}
        os.environ["HNH_RING_MODE"] = "mesh"
        from .common import set_chunk_spec
        set_chunk_spec(chunk_spec)

        def body(w):
            sp = H.SpmatLocal.load_tuples(w, False, logm, ef)
            op = H.DistributedSparse(w, "15d_fusion2", sp, r, 1)
            sp.free()
            A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
            S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
            call = lambda: op.fusedSpMM(A, B, S, buf, H.AMAT)  # noqa: E731
            op.hold_moving_operand(B)
            op.walk_windows_when_held(2)
            call()  # collective: fills the landing buffers
            w.sync()
            w.barrier()

            def alone():
                info = op.info()
                held = time_alone(w, op, call, 5)      # one windowed pass per chunk: the sequence of a call whose chunks arrive one by one
                op.walk_windows_when_held(1)
                landed = time_alone(w, op, call, 5)    # adaptive windows with everything landed: own block + one pass
                op.hold_moving_operand(None)
                op.walk_windows_when_held(0)
                w.set_solo(True)
                solo = time_alone(w, op, call, 5)
                return held, landed, solo, info["nS"], info["localArows"]
            res = solo_section(w, alone)
            for x in (A, B, S, buf):
                x.free()
            op.free()
            return res
        try:
            held, landed, solo, nnz_rank, rows = H.run_spmd(p, body)[0]
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        by = fused_bytes(nnz_rank, r, rows)
        return {"p": p, "chunks": chunk_spec, "R": r, "nnz_rank": int(nnz_rank), "rows_rank": int(rows), "algorithmic_bytes_rank": by,
                "held": {"what": "kernel sequence only (fetched blocks resident), ONE windowed pass PER CHUNK: what a call runs whose chunks arrive "
                                 "one by one (links slower than the kernels)", "wall_ms": held[0], "kernel_ms": held[1],
                         "launches": held[2], "frac": frac_of(by, held[1]), "frac_wall": frac_of(by, held[0])},
                "held_all_landed": {"what": "kernel sequence only, adaptive windows with every chunk landed when the host decides: own block + one "
                                            "pass over all fetched blocks (links faster than the kernels)", "wall_ms": landed[0],
                                    "kernel_ms": landed[1], "launches": landed[2], "frac": frac_of(by, landed[1]), "frac_wall": frac_of(by, landed[0])},
                "solo": {"what": "kernel sequence + the HBM side of the exchange (%d blocks of %d x %d copied device to device per call, overlapped "
                                 "through the schedule's events; adaptive windows: a pass takes every chunk whose copy has completed)" % (p - 1, rows, r),
                         "wall_ms": solo[0], "kernel_ms": solo[1], "launches": solo[2],
                         "frac_wall": frac_of(by, solo[0])}}

    for p, spec in ((8, "1,2,2,2,1,1"), (8, "1"), (4, "1,2,2,2,1,1"), (2, "1,2,2,2,1,1")):
        entry("rank_cfg3_p%d%s" % (p, "" if "," in spec else "_q" + spec), "rank share, config 3: one rank of %d (15d_fusion2, c = 1, mesh fetch, chunk heights %s) alone on this GPU, %s"
              % (p, spec, "toy size" if small else "ER 2^%d, edge factor %d, R=%d" % (args.logm, args.edge_factor, args.r)),
              lambda p=p, spec=spec: share_of_config3(p, spec))

    def share_of(alg, p, c, logm, ef, r, kind, unfused, als_iters=0):
        """one rank of `alg` on p logical ranks, solo replay; als_iters > 0: cg_optimizer(Amat, als_iters) instead of fusedSpMM"""
        def body(w):
            if kind == "er":
                sp = H.SpmatLocal.load_tuples(w, False, logm, ef)
            else:
                rows, cols = H.generate_rmat(logm, (1 << logm) * ef)
                sp = H.SpmatLocal.from_global(w, 1 << logm, 1 << logm, rows, cols, np.ones(len(rows)))
            gnnz = sp.info()["dist_nnz"]
            op = H.DistributedSparse(w, alg, sp, r, c)
            sp.free()
            als = None
            if als_iters:
                als = H.DistributedALS(op, True)
                als.initializeEmbeddings()
                call = lambda: als.cg_optimizer(H.AMAT, als_iters)  # noqa: E731
                ncalls = als_iters + 2  # right-hand side + initial residual + the iterations (als_conjugate_gradients.cpp:38-141)
            else:
                A, B = op.like_A_matrix(0.001), op.like_B_matrix(0.001)
                S, buf = op.like_S_values(1.0), op.like_S_values(0.0)
                call = lambda: op.fusedSpMM(A, B, S, buf, H.AMAT)  # noqa: E731
                ncalls = 1
            # everybody together first: wall time of the collective call with all p ranks' kernels and copies on this ONE GPU
            call()
            w.sync()
            w.barrier()
            t0 = time.perf_counter()
            for _ in range(3):
                call()
            w.sync()
            w.barrier()
            together = (time.perf_counter() - t0) * 1e3 / 3

            def alone():
                w.set_solo(True)
                return time_alone(w, op, call, 5 if not als_iters else 2)
            res = solo_section(w, alone)
            if als is not None:
                als.free()
            else:
                for x in (A, B, S, buf):
                    x.free()
            op.free()
            return res, together, gnnz, ncalls
        out_ = H.run_spmd(p, body)
        (wall, kms, launches), together, gnnz, ncalls = out_[0][0], max(o[1] for o in out_), out_[0][2], out_[0][3]
        m = 1 << logm
        by = ((gnnz * (16 * r + 44) + 16 * r * m) if unfused else fused_bytes(gnnz, r, m)) * ncalls / p
        return {"schedule": "%s p=%d c=%d" % (alg, p, c), "R": r, "nnz": int(gnnz), "M": m,
                "algorithmic_bytes_rank": by, "model": ("global %s model / p" % ("unfused (SDDMM + SpMM pair)" if unfused else "fused")) +
                (" x %d fused calls (2 + %d CG iterations)" % (ncalls, als_iters) if als_iters else ""),
                "solo": {"what": "rank 0 alone: its kernel sequence + the HBM side of its exchange", "wall_ms": wall, "kernel_ms": kms,
                         "launches": launches, "frac_wall": frac_of(by, wall), "frac_kernel": frac_of(by, kms),
                         "host_and_wait_ms": wall - kms},
                "all_ranks_on_this_gpu_ms": together}

    c4 = (8, 8, 32) if small else (20, 44, 256)  # (the R-MAT of entry (i): 4.3e7 nonzeros, longest row 8e4 — a rank's kernels take milliseconds)
    entry("rank_cfg4", "rank share, config 4: one rank of 8 (2.5D dense-replicate 2 x 2 x 2: R/2 columns, transposed blocks, accumulator in two halves), "
          "R-MAT 2^%d, edge factor %d, R=%d" % c4, lambda: share_of("25d_dense_replicate", 8, 2, c4[0], c4[1], c4[2], "rmat", True))
    c5 = (9, 8, 16) if small else (args.logm, args.edge_factor, args.r)
    entry("rank_cfg5", "rank share, config 5: one rank of 8 through a CG half-step (cg_optimizer(Amat, 10): 12 fused calls with the CG updates in the row "
          "epilogue, fixed factor held), 15d_fusion2, ER 2^%d, edge factor %d, R=%d" % c5,
          lambda: share_of("15d_fusion2", 8, 1, c5[0], c5[1], c5[2], "er", False, als_iters=10))
    c13 = (9, 8, 16) if small else (args.logm, args.edge_factor, args.r)
    entry("rank_25d_sparse", "rank share, 2.5D sparse-replicate (the one schedule the configurations do not name): one rank of 8 (2 x 2 x 2: S stationary and "
          "replicated, both dense operands move, R/4 columns per rank), SDDMM + SpMM pair, ER 2^%d, edge factor %d, R=%d" % c13,
          lambda: share_of("25d_sparse_replicate", 8, 2, c13[0], c13[1], c13[2], "er", True))
    entry("rank_15d_fusion1", "rank share, 1.5D dense shift by replication reuse (15d_fusion1): one rank of 8, SDDMM + SpMM pair, the SpMM's accumulator travelling in two "
          "row halves, ER 2^%d, edge factor %d, R=%d" % c13, lambda: share_of("15d_fusion1", 8, 1, c13[0], c13[1], c13[2], "er", True))
    c1 = (8, 8, 16) if small else (16, 16, 16)
    entry("cfg1_as_typed", "config 1 as typed: ER 2^%d, edge factor %d, R=%d, 15d_sparse, 2 logical ranks (bench_erdos_renyi.cpp:19-120): one fusedSpMM, kernel "
          "time against call time" % c1, lambda: share_of("15d_sparse", 2, 1, c1[0], c1[1], c1[2], "er", True))
    return out
