"""TEST INFRASTRUCTURE ONLY — launch helper for oracle/_ref/ref_driver (the compiled reference).

Used by tests/golden/make_golden.py (here, where /root/reference exists, to produce the committed
golden vectors), by tests that compare against the reference binary when it is present, and by
bench.py's `cpu_baseline` leg (kind = "reference").  The product never imports this.

The binary links MKL 2021.4 + MPICH 3.3.2 from /opt/conda/lib.  That directory also holds an old
libstdc++, so it must never be on a library search path: a private directory of symlinks is created
under /tmp at run time and passed through LD_LIBRARY_PATH (SURVEY.md Appendix B pitfalls).
"""
from __future__ import annotations

import glob
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
CONDA_LIB = os.environ.get("HNH_CONDA_LIB", "/opt/conda/lib")
MPIEXEC = os.environ.get("HNH_MPIEXEC", "/opt/conda/bin/mpiexec")
ALGS = ("15d_fusion1", "15d_fusion2", "15d_sparse", "25d_dense_replicate", "25d_sparse_replicate")


PREEXEC = None  # (bench.py's cpu_baseline leg: children get SIGTERM / SIGINT unblocked)


def available() -> bool:
    return (os.path.exists(os.path.join(REF_DIR, "ref_driver"))
            and os.path.exists(os.path.join(CONDA_LIB, "libmkl_core.so.1"))
            and os.path.exists(MPIEXEC))


def _deps_dir() -> str:
    d = os.path.join(tempfile.gettempdir(), "hnh_ref_deps_%d" % os.getuid())
    os.makedirs(d, exist_ok=True)
    pats = ("libmpi.so*", "libmkl_*.so*", "libgfortran.so.4", "libquadmath.so.0")
    for pat in pats:
        for f in glob.glob(os.path.join(CONDA_LIB, pat)):
            dst = os.path.join(d, os.path.basename(f))
            if not os.path.lexists(dst):
                try:
                    os.symlink(f, dst)
                except FileExistsError:
                    pass
    return d


def write_case(path: str, m: int, n: int, rows, cols, vals, r: int, a=None, b=None) -> None:
    with open(path, "wb") as f:
        f.write(b"HNHCASE1")
        np.array([m, n, len(rows), r], dtype=np.int64).tofile(f)
        np.ascontiguousarray(rows, dtype=np.int64).tofile(f)
        np.ascontiguousarray(cols, dtype=np.int64).tofile(f)
        np.ascontiguousarray(vals, dtype=np.float64).tofile(f)
        if a is not None:
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
            np.ascontiguousarray(b, dtype=np.float64).tofile(f)


def run(args, p: int, alg: str, threads: int | None = None, timeout: float = 600.0, retries: int = 3) -> str:
    """Run ref_driver on `p` MPI ranks; returns stdout.  2.5D dense uses the patched build and is
    retried on timeout (reference races)."""
    exe = os.path.join(REF_DIR, "ref_driver_patched" if alg == "25d_dense_replicate" else "ref_driver")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = _deps_dir()
    env["OMP_NUM_THREADS"] = str(threads if threads else max(1, (os.cpu_count() or 1) // p))
    env["MKL_NUM_THREADS"] = env["OMP_NUM_THREADS"]
    env.pop("LD_PRELOAD", None)
    cmd = [MPIEXEC, "-n", str(p), exe] + [str(a) for a in args]
    last = None
    for _ in range(retries):
        try:
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, preexec_fn=PREEXEC)
        except subprocess.TimeoutExpired as e:  # known reference hang; try again
            last = e
            continue
        if out.returncode != 0:
            raise RuntimeError("ref_driver failed (%d): %s\n%s" % (out.returncode, out.stdout[-2000:], out.stderr[-2000:]))
        return out.stdout
    raise RuntimeError("ref_driver timed out %d times: %r" % (retries, last))


def _assemble_dense(prefix, p, name, which, nrows, r):
    out = np.zeros((nrows, r))
    for rank in range(p):
        subs = np.fromfile("%s.r%d.sub%s.i64" % (prefix, rank, which), dtype=np.int64).reshape(-1, 4)
        loc = np.fromfile("%s.r%d.%s" % (prefix, rank, name), dtype=np.float64)
        off = 0
        for top, left, rc, cc in subs:
            blk = loc[off:off + rc * cc].reshape(rc, cc)
            off += rc * cc
            keep = max(0, min(rc, nrows - top))
            out[top:top + keep, left:left + cc] = blk[:keep]
    return out


def _assemble_sparse(prefix, p, name, keyname):
    keys, vals = [], []
    for rank in range(p):
        keys.append(np.fromfile("%s.r%d.%s" % (prefix, rank, keyname), dtype=np.int64))
        vals.append(np.fromfile("%s.r%d.%s" % (prefix, rank, name), dtype=np.float64))
    keys, vals = np.concatenate(keys), np.concatenate(vals)
    order = np.argsort(keys, kind="stable")
    return keys[order], vals[order]


def dump(m, n, rows, cols, vals, r, a, b, alg: str, p: int, c: int, timeout: float = 120.0) -> dict:
    """All six operator results of the reference for one (alg, p, c), keyed globally.

    Returns dict with sddmmA/sddmmB/fusedA_buf/fusedB_buf as (keys, values) sorted by key = i*N + j,
    and spmmA/spmmB/fusedA/fusedB as global dense matrices."""
    with tempfile.TemporaryDirectory(prefix="hnh_ref_") as td:
        case = os.path.join(td, "case.bin")
        write_case(case, m, n, rows, cols, vals, r, a, b)
        prefix = os.path.join(td, "out")
        run(["dump", case, alg, c, prefix], p, alg, timeout=timeout)
        res = {}
        for name, kn in (("sddmmA", "keysS"), ("sddmmB", "keysST"), ("fusedA_buf", "keysS"), ("fusedB_buf", "keysST")):
            res[name] = _assemble_sparse(prefix, p, name + ".f64", kn + ".i64")
        for name, which, nr in (("spmmA", "A", m), ("fusedA", "A", m), ("spmmB", "B", n), ("fusedB", "B", n)):
            res[name] = _assemble_dense(prefix, p, name + ".f64", which, nr, r)
        return res


def fingerprints(m, n, rows, cols, r, alg: str, p: int, c: int, timeout: float = 120.0, threads: int | None = None) -> dict:
    with tempfile.TemporaryDirectory(prefix="hnh_ref_") as td:
        case = os.path.join(td, "case.bin")
        write_case(case, m, n, rows, cols, np.ones(len(rows)), r)
        out = run(["fp", case, alg, c], p, alg, threads=threads, timeout=timeout)
        return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])


def bench(m, n, rows, cols, r, alg: str, p: int, c: int, fused: bool, trials: int,
          threads: int | None = None, timeout: float = 1800.0) -> dict:
    with tempfile.TemporaryDirectory(prefix="hnh_ref_") as td:
        case = os.path.join(td, "case.bin")
        write_case(case, m, n, rows, cols, np.ones(len(rows)), r)
        out = run(["bench", case, alg, c, int(fused), trials], p, alg, threads=threads, timeout=timeout)
        return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])


def sweep(m, n, rows, cols, r, alg: str, p: int, c: int, fused: bool, trials: int, reps: int, thread_counts, timeout: float = 600.0) -> dict:
    """The timing loop of bench() at several OpenMP/MKL thread counts behind one set-up (ref_driver sweep): per count one warm-up
    call and `reps` batches of `trials` timed calls; "points" = [{"threads", "elapsed" (best batch), "batches", "nnz_R_per_s"}]."""
    with tempfile.TemporaryDirectory(prefix="hnh_ref_") as td:
        case = os.path.join(td, "case.bin")
        write_case(case, m, n, rows, cols, np.ones(len(rows)), r)
        counts = sorted(set(int(t) for t in thread_counts))
        out = run(["sweep", case, alg, c, int(fused), trials, reps, ",".join(str(t) for t in counts)], p, alg, threads=max(counts), timeout=timeout, retries=1)
        return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])


def als(m, n, rows, cols, vals, r, a, b, alg: str, p: int, c: int, steps: int, cg_iters: int, timeout: float = 300.0,
        threads: int | None = None) -> dict:
    """ALS-CG of the reference (als_conjugate_gradients.cpp) with ground truth = `vals`, embeddings initialised
    from (a, b): returns the global A, B after `steps` alternating steps and the residual history."""
    with tempfile.TemporaryDirectory(prefix="hnh_ref_") as td:
        case = os.path.join(td, "case.bin")
        write_case(case, m, n, rows, cols, vals, r, a, b)
        prefix = os.path.join(td, "out")
        run(["als", case, alg, c, prefix, steps, cg_iters], p, alg, threads=threads, timeout=timeout)
        return {"A": _assemble_dense(prefix, p, "alsA.f64", "A", m, r), "B": _assemble_dense(prefix, p, "alsB.f64", "B", n, r),
                "residuals": np.fromfile(prefix + ".r0.residuals.f64", dtype=np.float64)}


def gat(m, rows, cols, r_in, a, alg: str, p: int, c: int, alpha: float, layers, timeout: float = 300.0):
    """GAT forward pass of the reference (gat.hpp:106-112) on a square graph: input features `a` (m x r_in),
    layers = [(in, features_per_head, heads), ...]; returns the final global feature matrix."""
    with tempfile.TemporaryDirectory(prefix="hnh_ref_") as td:
        case = os.path.join(td, "case.bin")
        write_case(case, m, m, rows, cols, np.ones(len(rows)), r_in, a, a)
        prefix = os.path.join(td, "out")
        spec = ",".join("%d:%d:%d" % tuple(l) for l in layers)
        run(["gat", case, alg, c, prefix, repr(float(alpha)), spec], p, alg, timeout=timeout)
        return _assemble_dense(prefix, p, "gat.f64", "A", m, layers[-1][1] * layers[-1][2])
