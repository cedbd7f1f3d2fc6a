"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy/scipy) of the reference's hot path.

Nothing in the product imports this module; only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may (and only as the checker).

What is restated, and which reference lines it follows (all paths under /root/reference):

* sddmm()      global meaning of Distributed_Sparse::sddmmA/B (distributed_sparse.h:284-290):
               per nonzero e=(i,j): out[e] = Sval[e] * <A[i,:], B[j,:]>, i.e. the COO loop of
               StandardKernel::sddmm_local (sparse_kernels.cpp:44-55) followed by the Hadamard product
               with SValues that every schedule applies (e.g. 15D_dense_shift.hpp:366).
* spmm_a()     spmmA (distributed_sparse.h:274-277): A = S * B with alpha = 1 on a zeroed output
               (mkl_sparse_d_mm call, sparse_kernels.cpp:95-107; MKL 2021.4 is a closed third-party
               dependency, its documented contract C = alpha*op(S)*B + beta*C is what is restated).
* spmm_b()     spmmB (distributed_sparse.h:279-282): B = S^T * A.
* fused_a/b()  fusedSpMM (distributed_sparse.h:296-312): SDDMM, zero the output, SpMM with the SDDMM
               values.  `ignore_svalues=True` restates the 1.5D dense-shift "local kernel fusion"
               override, which never multiplies by Svalues (15D_dense_shift.hpp:189,203-217, SURVEY
               Appendix C #4).
* dummy_fill() Distributed_Sparse::dummyInitialize pattern value(row, col) = row*R + col
               (distributed_sparse.h:338-342), used by the scratch.cpp:26-76 fingerprints.

Pinning: tests/test_oracle_golden.py checks every function here against tests/golden/*.npz, which
were produced by running the reference itself (oracle/_ref/ref_driver, all five schedules, several
(p, c)) through tests/golden/make_golden.py.  Tolerance 1e-11 relative (fp64, summation order only).

Also here: the deterministic synthetic inputs shared by the oracle, the reference driver and the HIP
path (Erdős–Rényi generator replacing CombBLAS GenGraph500Data with initiator .25/.25/.25/.25,
SpmatLocal.hpp:502-505; dense fills keyed by global (row, col)).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """The splitmix64 output function applied to `x + GOLDEN` (uint64, wrapping)."""
    with np.errstate(over="ignore"):
        z = (x.astype(np.uint64) + _GOLDEN).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def erdos_renyi_mn(m: int, n: int, draws: int, seed: int = 12345):
    """`draws` i.i.d. uniform (row, col) pairs on an m x n grid, de-duplicated, sorted row-major.

    draw k: row = splitmix64(seed + 2k*G) % m, col = splitmix64(seed + (2k+1)*G) % n.
    Returns (rows int64, cols int64).  The native generator (hnh_generate_er) is bit-identical.
    """
    with np.errstate(over="ignore"):
        k = np.arange(draws, dtype=np.uint64)
        base = np.uint64(seed) + (k * np.uint64(2)) * _GOLDEN
        r = splitmix64(base) % np.uint64(m)
        c = splitmix64(base + _GOLDEN) % np.uint64(n)
        keys = np.unique(r * np.uint64(n) + c)
    return (keys // np.uint64(n)).astype(np.int64), (keys % np.uint64(n)).astype(np.int64)


def erdos_renyi(log_m: int, edge_factor: int, seed: int = 12345):
    """The bench_erdos_renyi.cpp:23-24 parameterisation: M = N = 2^logM, M*edgeFactor draws."""
    m = 1 << log_m
    return erdos_renyi_mn(m, m, m * edge_factor, seed)


def rmat(log_m: int, edges: int, a: float = 0.57, b: float = 0.19, c: float = 0.19, seed: int = 12345, scramble: bool = True):
    """Graph500-style R-MAT stand-in for skewed real graphs (BASELINE config 4); twin of hnh::rmat_keys."""
    n = np.uint64(1 << log_m)
    with np.errstate(over="ignore"):
        k = np.arange(edges, dtype=np.uint64)
        r = np.zeros(edges, dtype=np.uint64)
        col = np.zeros(edges, dtype=np.uint64)
        for lvl in range(log_m):
            h = splitmix64(np.uint64(seed) + (k * np.uint64(log_m) + np.uint64(lvl)) * _GOLDEN)
            u = (h >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
            rb = (u >= a + b).astype(np.uint64)
            cb = (((u >= a) & (u < a + b)) | (u >= a + b + c)).astype(np.uint64)
            r = (r << np.uint64(1)) | rb
            col = (col << np.uint64(1)) | cb
        if scramble:
            r = (r * np.uint64(0x9E3779B1) + np.uint64(0x7F4A7C15)) & (n - np.uint64(1))
            col = (col * np.uint64(0x9E3779B1) + np.uint64(0x7F4A7C15)) & (n - np.uint64(1))
        keys = np.unique(r * n + col)
    return (keys // n).astype(np.int64), (keys % n).astype(np.int64)


def vertex_permutation(n: int, seed: int) -> np.ndarray:
    """new_label[v] = position of v when vertices are ordered by splitmix64(seed + v*G); twin of hnh::vertex_permutation."""
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) + np.arange(n, dtype=np.uint64) * _GOLDEN)
    order = np.argsort(h, kind="stable")
    label = np.empty(n, dtype=np.int64)
    label[order] = np.arange(n)
    return label


def hashed_uniform(keys: np.ndarray, seed: int) -> np.ndarray:
    """uniform(-1, 1) fp64 as a pure function of (key, seed)."""
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + keys.astype(np.uint64) * _GOLDEN)
    return (h >> np.uint64(11)).astype(np.float64) * (2.0 ** -52) - 1.0


def dense_fill(nrows: int, r: int, seed: int) -> np.ndarray:
    """uniform(-1,1)/R keyed by (global row, col, seed) — the ALS initialisation shape
    (als_conjugate_gradients.cpp:143-146 uses setRandom()/R)."""
    keys = np.arange(nrows * r, dtype=np.uint64)
    return (hashed_uniform(keys, seed) / r).reshape(nrows, r)


def sparse_values(rows: np.ndarray, cols: np.ndarray, n: int, seed: int) -> np.ndarray:
    """uniform(-1,1) S values keyed by the global coordinate (row*N + col)."""
    return hashed_uniform(rows.astype(np.uint64) * np.uint64(n) + cols.astype(np.uint64), seed)


def dummy_fill(nrows: int, r: int) -> np.ndarray:
    """distributed_sparse.h:338-342: value(row, col) = row*R + col in global coordinates."""
    return (np.arange(nrows, dtype=np.float64)[:, None] * r + np.arange(r, dtype=np.float64)[None, :])


# ------------------------------------------------------------------ global semantics (row a9 of §8)

def sddmm(rows, cols, svals, a, b, chunk: int = 1 << 20) -> np.ndarray:
    out = np.empty(len(rows), dtype=np.float64)
    for s in range(0, len(rows), chunk):
        e = slice(s, s + chunk)
        out[e] = np.einsum("ij,ij->i", a[rows[e]], b[cols[e]])
    return out * svals


def _csr(rows, cols, vals, m, n):
    return sp.csr_matrix((vals, (rows, cols)), shape=(m, n))


def spmm_a(rows, cols, vals, b, m) -> np.ndarray:
    return _csr(rows, cols, vals, m, b.shape[0]) @ b


def spmm_b(rows, cols, vals, a, n) -> np.ndarray:
    return _csr(rows, cols, vals, a.shape[0], n).T.tocsr() @ a


def fused_a(rows, cols, svals, a, b, ignore_svalues: bool = False):
    sv = np.ones_like(svals) if ignore_svalues else svals
    mid = sddmm(rows, cols, sv, a, b)
    return spmm_a(rows, cols, mid, b, a.shape[0]), mid


def fused_b(rows, cols, svals, a, b, ignore_svalues: bool = False):
    sv = np.ones_like(svals) if ignore_svalues else svals
    mid = sddmm(rows, cols, sv, a, b)
    return spmm_b(rows, cols, mid, a, b.shape[0]), mid


def fingerprints(rows, cols, m, n, r):
    """scratch.cpp:26-76: squared norms of sddmmA / spmmA / spmmB under dummy_fill, S = 1."""
    a, b = dummy_fill(m, r), dummy_fill(n, r)
    ones = np.ones(len(rows))
    f1 = float(np.sum(sddmm(rows, cols, ones, a, b) ** 2))
    f2 = float(np.sum(spmm_a(rows, cols, ones, b, m) ** 2))
    f3 = float(np.sum(spmm_b(rows, cols, ones, a, n) ** 2))
    return f1, f2, f3


def fingerprints_closed_form(rows, cols, m, n, r, chunk: int = 1 << 24):
    """The same three fingerprints (scratch.cpp:26-76) WITHOUT forming a dense operand, so that they can be had at any size
    in O(nnz) time and memory — the checker of the full-size GPU tests.  Under dummy_fill (distributed_sparse.h:338-342)
    A[i,k] = iR + k and B[j,k] = jR + k, hence with S1 = sum_k k, S2 = sum_k k^2 over k < R:
        sddmm(i,j)  = sum_k (iR + k)(jR + k)        = R^3 ij + R S1 (i + j) + S2
        spmmA[i,k]  = sum_{j in row i} (jR + k)      = R sj(i) + deg(i) k,     sj(i) = sum of the row's column indices
        sum_k spmmA[i,k]^2                           = R (R sj)^2 + 2 (R sj) deg S1 + deg^2 S2      (spmmB: rows <-> columns)
    Pinned against `fingerprints` above and the reference's own numbers in tests/test_oracle_golden.py."""
    big_r = float(r)
    s1 = r * (r - 1) / 2.0
    s2 = (r - 1) * r * (2 * r - 1) / 6.0
    f1 = 0.0
    for s in range(0, len(rows), chunk):
        i = rows[s:s + chunk].astype(np.float64)
        j = cols[s:s + chunk].astype(np.float64)
        v = big_r ** 3 * i * j + big_r * s1 * (i + j) + s2
        f1 += float(np.sum(v * v))

    def dense_side(own, other, count):
        deg = np.bincount(own, minlength=count).astype(np.float64)
        so = big_r * np.bincount(own, weights=other.astype(np.float64), minlength=count)
        return float(np.sum(big_r * so * so + 2.0 * so * deg * s1 + deg * deg * s2))

    return f1, dense_side(rows, cols, m), dense_side(cols, rows, n)


# ------------------------------------------------------------------ GAT forward (gat.hpp:83-112)

def gat_weight(layer: int, head: int, k: int, n: int, seed: int = 31) -> np.ndarray:
    """W(layer, head) of shape (k, n): hashed uniform(-1, 1) / k — what oracle/ref_driver.cpp's gat mode installs."""
    kk, jj = np.meshgrid(np.arange(k, dtype=np.uint64), np.arange(n, dtype=np.uint64), indexing="ij")
    keys = ((np.uint64(layer * 64 + head) * np.uint64(65536) + kk) * np.uint64(65536) + jj)
    return hashed_uniform(keys.reshape(-1), seed).reshape(k, n) / k


def gat_forward(rows, cols, m, x, layers, alpha: float, seed: int = 31) -> np.ndarray:
    """gat.hpp:83-112 for schedules that do not split R.  Per head: A = X W (:88); e = SDDMM(A, A) with S = 1
    (:92); LeakyReLU(e) (:96-97); H = SpMM(e, A) (:100); output columns of the head = ReLU(H) (:103)."""
    ones = np.ones(len(rows))
    for li, (fin, fph, heads) in enumerate(layers):
        assert x.shape[1] == fin
        out = np.zeros((m, fph * heads))
        for h in range(heads):
            a = x @ gat_weight(li, h, fin, fph, seed)
            e = sddmm(rows, cols, ones, a, a)
            e = np.maximum(e, 0.0) + np.minimum(e, 0.0) * alpha
            out[:, h * fph:(h + 1) * fph] = np.maximum(spmm_a(rows, cols, e, a, m), 0.0)
        x = out
    return x


# ------------------------------------------------------------------ kernel-level semantics (rows a1/a2)

def sddmm_local(row_idx, col_idx, values, x, y) -> np.ndarray:
    """sparse_kernels.cpp:44-55 — values[i] += <X[row_idx[i],:], Y[col_idx[i],:]> (accumulates)."""
    return values + np.einsum("ij,ij->i", x[row_idx], y[col_idx])


def spmm_local(rowptr, col_idx, values, x, y) -> np.ndarray:
    """sparse_kernels.cpp:95-107 — Y += S_blk * X (alpha = 1, beta = 1), CSR block."""
    rows = len(rowptr) - 1
    s = sp.csr_matrix((values, col_idx, rowptr), shape=(rows, x.shape[0]))
    return y + s @ x
