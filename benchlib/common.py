"""Shared pieces of the benchmark driver: the one-line stdout contract, chunk-shape and route names, the keyed operands of the
result checks, the workloads (benchmark_dist.cpp / bench_erdos_renyi.cpp / bench_file.cpp) and the SURVEY 8(d) byte model."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T_START = time.monotonic()  # the run's clock: --budget-s and phases_s count from here (bench.py imports this module first)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
GAT_LAYERS = [(256, 256, 4), (1024, 256, 4), (1024, 256, 6)]  # benchmark_dist.cpp:88-94: (input features, features per head, heads)

_JSON_FD = None


def claim_stdout():
    """stdout carries exactly ONE line, the JSON result: everything else this process writes to file descriptor 1 (the host
    library mirrors the reference's console messages, e.g. "R-mat generator created ... nonzeros") goes to stderr instead."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    """Print THE line: a dict goes out compacted to at most line.LINE_LIMIT bytes (benchlib/line.py), its full form into the record file
    beside it; the write is repeated until every byte is out (a pipe may take a long line in pieces)."""
    from . import line as L
    data, _ = L.render(obj)
    if _JSON_FD is None:
        sys.stdout.flush()
        fd = sys.stdout.fileno()
    else:
        fd = _JSON_FD
    view = memoryview(data)
    while len(view):
        try:
            view = view[os.write(fd, view):]
        except InterruptedError:
            continue


def unblock_signals():
    """preexec_fn of every child this driver starts: rank 0 keeps SIGTERM / SIGINT blocked for its sigwait() thread (guards.Fallback),
    and a signal mask is inherited across exec."""
    import signal
    signal.pthread_sigmask(signal.SIG_UNBLOCK, {signal.SIGTERM, signal.SIGINT})


DEFAULT_CHUNKS = "1,2,2,2,1,1"  # the library's default shape of the mesh fetch (dense_shift_15d.hpp)


_merge_switched_off_here = False
STATIC_WINDOWS = "/one-pass-per-chunk"  # suffix of a chunk spec: HNH_WINDOW_MERGE=0 (the route search measures it against the adaptive windows)


def set_chunk_spec(spec):
    """A chunk spec is a number (Q symmetric chunks, HNH_MESH_CHUNKS) or a comma list of heights (HNH_MESH_TAPER); with the suffix
    STATIC_WINDOWS the windowed passes take exactly one chunk each (HNH_WINDOW_MERGE=0) instead of every chunk that has landed."""
    global _merge_switched_off_here
    if spec.endswith(STATIC_WINDOWS):
        spec = spec[:-len(STATIC_WINDOWS)]
        os.environ["HNH_WINDOW_MERGE"] = "0"
        _merge_switched_off_here = True
    elif _merge_switched_off_here:  # (a HNH_WINDOW_MERGE the USER set is never undone)
        os.environ.pop("HNH_WINDOW_MERGE", None)
        _merge_switched_off_here = False
    if "," in spec:
        os.environ["HNH_MESH_TAPER"] = spec
        os.environ.pop("HNH_MESH_CHUNKS", None)
    else:
        os.environ["HNH_MESH_CHUNKS"] = spec
        os.environ.pop("HNH_MESH_TAPER", None)


FUSION1_CHUNKS = "1,2,1"  # ... and of 15d_fusion1's row-merged layout (staging passes + transfer groups)


def current_chunk_spec(alg=None):
    return os.environ.get("HNH_MESH_TAPER") or os.environ.get("HNH_MESH_CHUNKS") or (FUSION1_CHUNKS if alg == "15d_fusion1" else DEFAULT_CHUNKS)


def route_name(route):
    """route = (transport, c, mode, chunk spec)"""
    tr, c, mode, q = route
    mesh = ("mesh/heights %s" % q) if (q and "," in str(q)) else ("mesh/%s chunks" % q)
    return "c=%d %s [%s]" % (c, {"mesh": mesh, "relay": "relay ring", "none": "replication only",
                                 "fusion1": "15d_fusion1 (replication reuse: SDDMM + SpMM, mesh fetch + mesh reduce-scatter)"}[mode], tr)


def keyed(idx, salt):
    """Deterministic value in [0.5, 1.5) per global index (multiplicative hash): the operands of the result check."""
    import numpy as np
    h = (idx.astype(np.uint64) * np.uint64(2654435761) + np.uint64(salt) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    return 0.5 + h.astype(np.float64) / 4294967296.0

def read_mtx_coordinates(path):
    """(rows, cols) of a MatrixMarket coordinate file as the library must deliver them — 0-based, both triangles of a symmetric file,
    duplicates merged, sorted row-major — parsed on the host by pandas' C reader (scipy.io.mmread where pandas is missing): a parse that
    shares no code with the library's own parser, fast enough for SuiteSparse-sized files (1.7e7 lines in a few seconds)."""
    import numpy as np
    try:
        import pandas as pd
    except ImportError:
        import scipy.io
        a = scipy.io.mmread(path).tocsr()
        a.sum_duplicates()
        a = a.tocoo()
        return a.row.astype(np.int64), a.col.astype(np.int64)
    with open(path, "rb") as f:
        header = f.readline().decode("ascii", "replace").lower()
        if not header.startswith("%%matrixmarket") or "coordinate" not in header:
            raise SystemExit("bench.py: %s is not a MatrixMarket coordinate file" % path)
        symmetric = any(w in header for w in ("symmetric", "hermitian", "skew"))
        skip, line = 1, f.readline()
        while line.startswith(b"%") or not line.strip():
            skip, line = skip + 1, f.readline()
        skip += 1  # the size line
        n = int(line.split()[1])
    df = pd.read_csv(path, sep=r"\s+", header=None, skiprows=skip, usecols=[0, 1], dtype=np.int64, engine="c")
    r, c = df[0].to_numpy() - 1, df[1].to_numpy() - 1
    if symmetric:
        off = r != c
        r, c = np.concatenate([r, c[off]]), np.concatenate([c, r[off]])
    key = np.unique(r * n + c)
    return key // n, key % n


class Workload:
    """The sparse matrix of the run (benchmark_dist.cpp / bench_erdos_renyi.cpp / bench_file.cpp): how every rank gets its
    tuples, and the host copy of the nonzeros the result checks sum over."""

    def __init__(self, spec, logm, edge_factor):
        self.spec, self.logm, self.ef = spec, logm, edge_factor
        self.kind = "mtx" if spec.startswith("mtx:") else spec
        if self.kind not in ("er", "rmat", "mtx"):
            raise SystemExit("bench.py --workload %r: use er, rmat or mtx:<path>" % spec)
        self.path = spec[4:] if self.kind == "mtx" else None
        self._host = None

    def host_nonzeros(self, H):
        """(rows, cols) of the global matrix on the host — the generators are deterministic and bit-identical to the device
        ones; a file is parsed with scipy when it is small enough."""
        if self._host is None:
            m = 1 << self.logm
            if self.kind == "er":
                self._host = H.generate_er(m, m, m * self.ef, 12345)
            elif self.kind == "rmat":
                self._host = H.generate_rmat(self.logm, m * self.ef)
            else:
                if os.path.getsize(self.path) > 4 << 30:
                    return None
                self._host = read_mtx_coordinates(self.path)
        return self._host

    def load(self, H, world):
        if self.kind == "er":
            return H.SpmatLocal.load_tuples(world, False, self.logm, self.ef)
        if self.kind == "mtx":
            return H.SpmatLocal.load_tuples(world, True, 0, 0, self.path)
        # the skewed initiator of the same generator call (SpmatLocal.hpp:502-505 with {.57, .19, .19, .05}), evaluated on the device like the
        # Erdos-Renyi one (hnh_generate_rmat_keys); the host twin (H.generate_rmat) is what the result check sums over
        saved = os.environ.get("HNH_RMAT")
        os.environ["HNH_RMAT"] = "0.57,0.19,0.19"
        try:
            return H.SpmatLocal.load_tuples(world, False, self.logm, self.ef)
        finally:
            if saved is None:
                os.environ.pop("HNH_RMAT", None)
            else:
                os.environ["HNH_RMAT"] = saved

    def describe(self, nnz):
        if self.kind == "er":
            return "Erdos-Renyi 2^%d, edge factor %d (%d nnz)" % (self.logm, self.ef, nnz)
        if self.kind == "rmat":
            return "R-MAT 2^%d (a,b,c = .57,.19,.19), edge factor %d (%d nnz)" % (self.logm, self.ef, nnz)
        return "MatrixMarket file %s (%d nnz)" % (os.path.basename(self.path), nnz)


def fused_bytes(nnz, r, rows):
    return nnz * (8 * r + 24) + 16 * r * rows
def error_line(args, message, **extra):
    """The one JSON line of a run that failed: the contract's keys with value null, plus what went wrong and where."""
    out = {"metric": "fused SDDMM+SpMM nnz*R/s", "value": None, "unit": "nnz*R/s", "n_gpus": args.gpus, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic", "config": {"workload": "%s 2^%d, edge factor %d, R=%d, %s, %s on %d x MI355X" % (
               args.workload, args.logm, args.edge_factor, args.r, args.app, args.alg, args.gpus)}, "error": message}
    out.update(extra)
    return out
