"""ctypes binding of include/hnh_dist.h (lib/libhnh_host.so): the HnH operator surface for Python callers.

Pure plumbing for tests/ and bench.py — the schedules, sparse storage and transports are C++
(csrc/host), the kernels are HIP (csrc/hip).  Method names follow the reference's classes
(Distributed_Sparse::sddmmA / spmmA / fusedSpMM / like_A_matrix ..., distributed_sparse.h:32-388).
There is no compute path in Python and no CPU fallback: with the default backend every constructor
below raises unless the HIP library loads and a GPU is present.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import threading

import numpy as np

from . import _kernels

HERE = os.path.dirname(os.path.abspath(__file__))
# HNH_HOST_LIB_DEV: the measurement tools load libhnh_host_aids.so (the same library + the paced stand-ins, see
# include/hnh_measurement_aids.h); nothing else sets it
HOST_LIB = os.environ.get("HNH_HOST_LIB_DEV") or os.path.join(HERE, "lib", "libhnh_host.so")

K_SDDMM_A, K_SPMM_A, K_SPMM_B, K_SDDMM_B = 0, 1, 2, 3
AMAT, BMAT = 0, 1
ALGORITHMS = ("15d_fusion1", "15d_fusion2", "15d_sparse", "25d_dense_replicate", "25d_sparse_replicate")

_vp, _i32, _i64, _dbl, _sz, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t, C.c_uint64
_pi64, _pdbl, _pvp, _pi32 = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_void_p), C.POINTER(C.c_int)

SENDRECV_CB = C.CFUNCTYPE(_i32, _vp, _vp, _sz, _i32, _vp, _sz, _i32)
BARRIER_CB = C.CFUNCTYPE(_i32, _vp)
ALLGATHER_CB = C.CFUNCTYPE(_i32, _vp, _vp, _vp, _sz)


class CommCallbacks(C.Structure):
    _fields_ = [("user", _vp), ("sendrecv", SENDRECV_CB), ("barrier", BARRIER_CB), ("allgather", ALLGATHER_CB)]


SIGNATURES = {
    "hnh_host_last_error": (C.c_char_p, []),
    "hnh_backend_load": (_i32, [C.c_char_p]),
    "hnh_host_backend_name": (C.c_char_p, []),
    "hnh_world_create_single": (_i32, [_i32, _pvp]),
    "hnh_thread_group_create": (_i32, [_i32, _pvp]),
    "hnh_thread_group_destroy": (_i32, [_vp]),
    "hnh_world_create_thread": (_i32, [_vp, _i32, _i32, _pvp]),
    "hnh_rccl_unique_id": (_i32, [_vp]),
    "hnh_world_create_rccl": (_i32, [_i32, _i32, _i32, _vp, _pvp]),
    "hnh_world_create_ipc": (_i32, [_i32, _i32, _i32, C.c_char_p, _pvp]),
    "hnh_world_create_callback": (_i32, [_i32, _i32, _i32, C.POINTER(CommCallbacks), _pvp]),
    "hnh_world_destroy": (_i32, [_vp]),
    "hnh_world_rank": (_i32, [_vp]),
    "hnh_world_size": (_i32, [_vp]),
    "hnh_world_barrier": (_i32, [_vp]),
    "hnh_world_sync": (_i32, [_vp]),
    "hnh_world_set_timing_sync": (_i32, [_vp, _i32]),
    "hnh_world_set_solo": (_i32, [_vp, _i32]),
    "hnh_world_stream": (_vp, [_vp, _i32]),
    "hnh_world_ctx": (_vp, [_vp]),
    "hnh_world_grid_probe": (_i32, [_vp, _i32, _i32, _i32, _i32, _pi32, _pi32]),
    "hnh_world_preflight": (_i32, [_vp, _i32, _i64, C.POINTER(_dbl)]),
    "hnh_world_split_signature": (_i32, [_vp, C.POINTER(_u64), _pi32]),
    "hnh_world_identities": (_i32, [_vp, _vp]),
    "hnh_spmat_create": (_i32, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _pvp]),
    "hnh_spmat_load_tuples": (_i32, [_vp, _i32, _i32, _i32, C.c_char_p, _pvp]),
    "hnh_spmat_info": (_i32, [_vp, _pi64]),
    "hnh_spmat_permute": (_i32, [_vp, _u64]),
    "hnh_spmat_destroy": (_i32, [_vp]),
    "hnh_er_generate": (_i32, [_u64, _u64, _u64, _u64, _pvp, _pi64]),
    "hnh_rmat_generate": (_i32, [_i32, _u64, _dbl, _dbl, _dbl, _u64, _i32, _pvp, _pi64]),
    "hnh_er_fetch": (_i32, [_vp, _vp, _vp]),
    "hnh_write_matrix_market": (_i32, [C.c_char_p, _i64, _i64, _i64, _vp, _vp, _vp, _i32]),
    "hnh_dist_create": (_i32, [_vp, C.c_char_p, _vp, _i32, _i32, _pvp]),
    "hnh_dist_destroy": (_i32, [_vp]),
    "hnh_dist_info": (_i32, [_vp, _pi64]),
    "hnh_dist_submatrices": (_i32, [_vp, _i32, _vp, _i32]),
    "hnh_dist_set_r": (_i32, [_vp, _i32]),
    "hnh_dist_json": (_i32, [_vp, _i32, C.c_char_p, _sz]),
    "hnh_dist_reset_timers": (_i32, [_vp]),
    "hnh_dist_kernel_profile": (_i32, [_vp, _i32, _pdbl, _pi64]),
    "hnh_dist_borrow_stats": (_i32, [_vp, _pi64]),
    "hnh_dense_create": (_i32, [_vp, _i64, _i64, _dbl, _pvp]),
    "hnh_dense_wrap": (_i32, [_vp, _vp, _i64, _i64, _pvp]),
    "hnh_dense_like": (_i32, [_vp, _i32, _dbl, _pvp]),
    "hnh_dense_shape": (_i32, [_vp, _pi64]),
    "hnh_dense_data": (_vp, [_vp]),
    "hnh_dense_upload": (_i32, [_vp, _vp]),
    "hnh_dense_download": (_i32, [_vp, _vp]),
    "hnh_dense_fill": (_i32, [_vp, _dbl]),
    "hnh_dense_copy": (_i32, [_vp, _vp]),
    "hnh_dense_destroy": (_i32, [_vp]),
    "hnh_dense_dummy_initialize": (_i32, [_vp, _vp, _i32]),
    "hnh_vec_create": (_i32, [_vp, _i64, _dbl, _pvp]),
    "hnh_vec_like": (_i32, [_vp, _i32, _dbl, _pvp]),
    "hnh_vec_size": (_i64, [_vp]),
    "hnh_vec_data": (_vp, [_vp]),
    "hnh_vec_upload": (_i32, [_vp, _vp]),
    "hnh_vec_download": (_i32, [_vp, _vp]),
    "hnh_vec_fill": (_i32, [_vp, _dbl]),
    "hnh_vec_destroy": (_i32, [_vp]),
    "hnh_dist_initial_shift": (_i32, [_vp, _vp, _vp, _i32]),
    "hnh_dist_de_shift": (_i32, [_vp, _vp, _vp, _i32]),
    "hnh_dist_sddmmA": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "hnh_dist_sddmmB": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "hnh_dist_spmmA": (_i32, [_vp, _vp, _vp, _vp]),
    "hnh_dist_spmmB": (_i32, [_vp, _vp, _vp, _vp]),
    "hnh_dist_fusedSpMM": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32]),
    "hnh_dist_algorithm": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32]),
    "hnh_dist_hold_moving_operand": (_i32, [_vp, _vp]),
    "hnh_dist_walk_windows_when_held": (_i32, [_vp, _i32]),
    "hnh_dist_fusedSpMM_out": (_i32, [_vp, _vp, _vp, _i32, _vp, _i32, C.c_double, C.c_double, _vp, C.POINTER(C.c_int)]),
    "hnh_als_create": (_i32, [_vp, _i32, _u64, _pvp]),
    "hnh_als_destroy": (_i32, [_vp]),
    "hnh_als_set_ground_truth": (_i32, [_vp, _vp, _vp]),
    "hnh_als_initialize_embeddings": (_i32, [_vp]),
    "hnh_als_set_embeddings": (_i32, [_vp, _vp, _vp]),
    "hnh_als_get_embeddings": (_i32, [_vp, _vp, _vp]),
    "hnh_als_cg_optimizer": (_i32, [_vp, _i32, _i32]),
    "hnh_als_run_cg": (_i32, [_vp, _i32]),
    "hnh_als_compute_residual": (_i32, [_vp, _pdbl]),
    "hnh_gat_create": (_i32, [_vp, _i32, _pi32, _dbl, _pvp]),
    "hnh_gat_destroy": (_i32, [_vp]),
    "hnh_gat_weight_shape": (_i32, [_vp, _i32, _i32, _pi64]),
    "hnh_gat_set_weight": (_i32, [_vp, _i32, _i32, _vp]),
    "hnh_gat_set_input": (_i32, [_vp, _vp]),
    "hnh_gat_get_output": (_i32, [_vp, _vp]),
    "hnh_gat_buffer_shape": (_i32, [_vp, _i32, _pi64]),
    "hnh_gat_forward": (_i32, [_vp]),
}

_lib = None
_lock = threading.Lock()


class HnhError(RuntimeError):
    pass


def lib() -> C.CDLL:
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(HOST_LIB):
                raise HnhError("host library %s is missing: run __graft_entry__.build()" % HOST_LIB)
            if os.environ.get("HNH_NO_TORCH") != "1":
                try:  # one HIP runtime per process: torch's bundled one must be loaded first if torch is used at all
                    import torch  # noqa: F401
                except ImportError:
                    pass
            l = C.CDLL(HOST_LIB)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(l, name)
                fn.restype, fn.argtypes = res, args
            _lib = l
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise HnhError("%s failed (%d): %s" % (what, rc, lib().hnh_host_last_error().decode(errors="replace")))


def load_backend(path: str | None = None) -> str:
    """Select the implementation of the kernel ABI.  None = the product HIP library (the default).
    Tests pass oracle/liboracle_backend.so explicitly to exercise host logic without a GPU."""
    _check(lib().hnh_backend_load(path.encode() if path else None), "hnh_backend_load")
    return backend_name()


def backend_name() -> str:
    return lib().hnh_host_backend_name().decode()


def generate_er(m: int, n: int, draws: int, seed: int = 12345):
    """The shared synthetic generator (native, OpenMP); bit-identical to oracle.erdos_renyi_mn."""
    h, cnt = _vp(), _i64()
    _check(lib().hnh_er_generate(m, n, draws, seed, C.byref(h), C.byref(cnt)), "hnh_er_generate")
    rows, cols = np.empty(cnt.value, np.int64), np.empty(cnt.value, np.int64)
    _check(lib().hnh_er_fetch(h, rows.ctypes.data, cols.ctypes.data), "hnh_er_fetch")
    return rows, cols


def generate_rmat(logm: int, edges: int, a: float = 0.57, b: float = 0.19, c: float = 0.19, seed: int = 12345, scramble: bool = True):
    """Graph500-style R-MAT (skewed degrees), de-duplicated and sorted row-major; twin of oracle.rmat."""
    h, cnt = _vp(), _i64()
    _check(lib().hnh_rmat_generate(logm, edges, a, b, c, seed, int(scramble), C.byref(h), C.byref(cnt)), "hnh_rmat_generate")
    rows, cols = np.empty(cnt.value, np.int64), np.empty(cnt.value, np.int64)
    _check(lib().hnh_er_fetch(h, rows.ctypes.data, cols.ctypes.data), "hnh_er_fetch")
    return rows, cols


def write_matrix_market(path: str, m: int, n: int, rows, cols, values=None, symmetric: bool = False):
    """hnh_write_matrix_market: the entries as a MatrixMarket coordinate file, formatted by all host cores."""
    rows, cols = np.ascontiguousarray(rows, np.int64), np.ascontiguousarray(cols, np.int64)
    vals = None if values is None else np.ascontiguousarray(values, np.float64)
    _check(lib().hnh_write_matrix_market(path.encode(), m, n, len(rows), rows.ctypes.data, cols.ctypes.data,
                                         None if vals is None else vals.ctypes.data, int(symmetric)), "hnh_write_matrix_market")


# --------------------------------------------------------------------------------------------- worlds
class World:
    def __init__(self, handle, keepalive=None):
        self.h = handle
        self._keep = keepalive
        self.rank = lib().hnh_world_rank(handle)
        self.size = lib().hnh_world_size(handle)

    @classmethod
    def single(cls, device: int = 0) -> "World":
        h = _vp()
        _check(lib().hnh_world_create_single(device, C.byref(h)), "hnh_world_create_single")
        return cls(h)

    @classmethod
    def thread(cls, group: "ThreadGroup", rank: int, device: int = 0) -> "World":
        h = _vp()
        _check(lib().hnh_world_create_thread(group.h, rank, device, C.byref(h)), "hnh_world_create_thread")
        return cls(h, group)

    @classmethod
    def rccl(cls, rank: int, nranks: int, device: int, unique_id: bytes) -> "World":
        h = _vp()
        buf = C.create_string_buffer(unique_id, _kernels.UNIQUE_ID_BYTES)
        _check(lib().hnh_world_create_rccl(rank, nranks, device, buf, C.byref(h)), "hnh_world_create_rccl")
        return cls(h)

    @classmethod
    def ipc(cls, rank: int, nranks: int, device: int, session: str) -> "World":
        """One process per GPU of one node, no RCCL: receivers pull out of their peers' mapped buffers.  `session` is the same
        string on every rank (ipc_session_id() on rank 0, handed round by the launcher)."""
        h = _vp()
        _check(lib().hnh_world_create_ipc(rank, nranks, device, session.encode(), C.byref(h)), "hnh_world_create_ipc")
        return cls(h)

    @classmethod
    def callback(cls, rank: int, nranks: int, device: int, callbacks: CommCallbacks) -> "World":
        h = _vp()
        _check(lib().hnh_world_create_callback(rank, nranks, device, C.byref(callbacks), C.byref(h)), "hnh_world_create_callback")
        return cls(h, callbacks)

    def barrier(self):
        _check(lib().hnh_world_barrier(self.h), "barrier")

    def sync(self):
        _check(lib().hnh_world_sync(self.h), "sync")

    def set_solo(self, on: bool):
        """World::set_solo — solo replay (measurement entry point, loopback transport): this rank runs its side of every collective alone."""
        _check(lib().hnh_world_set_solo(self.h, int(on)), "set_solo")

    def set_timing_sync(self, on: bool):
        lib().hnh_world_set_timing_sync(self.h, int(on))

    def grid_probe(self, nr, nc, nh, adjacency):
        out, ok = (C.c_int * 9)(), C.c_int()
        _check(lib().hnh_world_grid_probe(self.h, nr, nc, nh, adjacency, out, C.byref(ok)), "grid_probe")
        return list(out), bool(ok.value)

    PREFLIGHT = ("ring sendrecv", "mesh group (n-1 pairs)", "allgather (layer)", "reduce_scatter (layer)", "allreduce (layer)",
                 "allgatherv + reduce_scatter_v", "device alltoallv", "allgather (world, native)", "reduce_scatter (world, native)")

    def preflight(self, what: int, count: int = 4096) -> float:
        """hnh_world_preflight: one transport primitive on known data; returns the largest deviation (collective)."""
        err = _dbl()
        _check(lib().hnh_world_preflight(self.h, what, count, C.byref(err)), "preflight: " + self.PREFLIGHT[what])
        return err.value

    def identities(self):
        """Collective: where every rank of the world runs — [{"rank", "pid", "device_ordinal", "pci_bus_id", + "comm_count", "comm_rank",
        "comm_device" when the transport has a communicator (RCCL)}], the same list on every rank (hnh_world_identities)."""
        class Rec(C.Structure):
            _fields_ = [("rank", _i32), ("pid", _i32), ("device_ordinal", _i32), ("comm_count", _i32), ("comm_rank", _i32), ("comm_device", _i32),
                        ("pci_bus_id", C.c_char * 40)]
        n = lib().hnh_world_size(self.h)
        buf = (Rec * n)()
        _check(lib().hnh_world_identities(self.h, C.cast(buf, _vp)), "identities")
        out = []
        for r in buf:
            rec = {"rank": r.rank, "pid": r.pid, "device_ordinal": r.device_ordinal, "pci_bus_id": r.pci_bus_id.decode("ascii", "replace")}
            if r.comm_count != -1:  # (-1: the transport has no communicator; -2: RCCL would not say)
                rec.update(comm_count=r.comm_count, comm_rank=r.comm_rank, comm_device=r.comm_device)
            out.append(rec)
        return out

    def split_signature(self):
        sig, n = _u64(), _i32()
        _check(lib().hnh_world_split_signature(self.h, C.byref(sig), C.byref(n)), "split_signature")
        return int(sig.value), int(n.value)

    def close(self):
        if self.h:
            _check(lib().hnh_world_destroy(self.h), "world_destroy")
            self.h = None


def ipc_session_id() -> str:
    """A name no other job on this node uses (made on rank 0, given to every rank)."""
    import os
    import time
    return "%d_%x" % (os.getpid(), time.time_ns())


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(_kernels.UNIQUE_ID_BYTES)
    _check(lib().hnh_rccl_unique_id(buf), "hnh_rccl_unique_id")
    return buf.raw


class ThreadGroup:
    def __init__(self, nranks: int):
        self.h = _vp()
        self.n = nranks
        _check(lib().hnh_thread_group_create(nranks, C.byref(self.h)), "thread_group_create")

    def close(self):
        if self.h:
            lib().hnh_thread_group_destroy(self.h)
            self.h = None


def run_spmd(nranks: int, fn, device: int = 0):
    """Run fn(world) on `nranks` logical ranks = host threads sharing one device (loopback transport).
    Returns the list of results; re-raises the first exception."""
    group = ThreadGroup(nranks)
    results, errors = [None] * nranks, [None] * nranks

    def body(r):
        w = None
        try:
            w = World.thread(group, r, device)
            results[r] = fn(w)
        except BaseException as e:  # noqa: BLE001
            errors[r] = e
        finally:
            if w is not None and errors[r] is None:
                try:
                    w.close()
                except BaseException as e:  # noqa: BLE001
                    errors[r] = e

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    alive = [t for t in threads if t.is_alive()]
    for e in errors:
        if e is not None:
            raise e
    if alive:
        raise HnhError("SPMD ranks hung (a peer probably failed)")
    group.close()
    return results


# --------------------------------------------------------------------------------------------- data
class Dense:
    def __init__(self, world: World, handle):
        self.w, self.h = world, handle

    @classmethod
    def create(cls, world: World, rows: int, cols: int, fill: float = 0.0) -> "Dense":
        h = _vp()
        _check(lib().hnh_dense_create(world.h, rows, cols, fill, C.byref(h)), "dense_create")
        return cls(world, h)

    @classmethod
    def wrap(cls, world: World, device_ptr: int, rows: int, cols: int) -> "Dense":
        h = _vp()
        _check(lib().hnh_dense_wrap(world.h, device_ptr, rows, cols, C.byref(h)), "dense_wrap")
        return cls(world, h)

    @property
    def shape(self):
        o = (C.c_int64 * 2)()
        lib().hnh_dense_shape(self.h, o)
        return int(o[0]), int(o[1])

    @property
    def data_ptr(self) -> int:
        return lib().hnh_dense_data(self.h)

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        assert arr.shape == self.shape, (arr.shape, self.shape)
        _check(lib().hnh_dense_upload(self.h, arr.ctypes.data), "dense_upload")

    def download(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float64)
        _check(lib().hnh_dense_download(self.h, out.ctypes.data), "dense_download")
        return out

    def fill(self, v: float):
        _check(lib().hnh_dense_fill(self.h, v), "dense_fill")

    def copy_from(self, other: "Dense"):
        _check(lib().hnh_dense_copy(self.h, other.h), "dense_copy")

    def free(self):
        if self.h:
            _check(lib().hnh_dense_destroy(self.h), "dense_destroy")
            self.h = None


class Vec:
    def __init__(self, world: World, handle):
        self.w, self.h = world, handle

    @classmethod
    def create(cls, world: World, n: int, fill: float = 0.0) -> "Vec":
        h = _vp()
        _check(lib().hnh_vec_create(world.h, n, fill, C.byref(h)), "vec_create")
        return cls(world, h)

    def __len__(self):
        return int(lib().hnh_vec_size(self.h))

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        assert arr.size == len(self)
        _check(lib().hnh_vec_upload(self.h, arr.ctypes.data), "vec_upload")

    def download(self) -> np.ndarray:
        out = np.empty(len(self), dtype=np.float64)
        _check(lib().hnh_vec_download(self.h, out.ctypes.data), "vec_download")
        return out

    def fill(self, v: float):
        _check(lib().hnh_vec_fill(self.h, v), "vec_fill")

    def free(self):
        if self.h:
            _check(lib().hnh_vec_destroy(self.h), "vec_destroy")
            self.h = None


class SpmatLocal:
    """SpmatLocal (SpmatLocal.hpp:267-606): the tuples this rank holds before redistribution."""

    def __init__(self, world: World, handle):
        self.w, self.h = world, handle

    @classmethod
    def from_tuples(cls, world: World, m: int, n: int, dist_nnz: int, rows, cols, vals=None) -> "SpmatLocal":
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        v = None if vals is None else np.ascontiguousarray(vals, dtype=np.float64)
        h = _vp()
        _check(lib().hnh_spmat_create(world.h, m, n, dist_nnz, len(rows), rows.ctypes.data, cols.ctypes.data,
                                      None if v is None else v.ctypes.data, C.byref(h)), "spmat_create")
        return cls(world, h)

    @classmethod
    def from_global(cls, world: World, m: int, n: int, rows, cols, vals=None) -> "SpmatLocal":
        """Every rank passes the same global tuple list and keeps the strided slice e % p == rank."""
        sl = slice(world.rank, None, world.size)
        return cls.from_tuples(world, m, n, len(rows), rows[sl], cols[sl], None if vals is None else vals[sl])

    @classmethod
    def load_tuples(cls, world: World, read_from_file: bool, log_m: int, nnz_per_row: int, filename: str = "") -> "SpmatLocal":
        h = _vp()
        _check(lib().hnh_spmat_load_tuples(world.h, int(read_from_file), log_m, nnz_per_row, filename.encode(), C.byref(h)),
               "spmat_load_tuples")
        return cls(world, h)

    def permute(self, seed: int):
        """Seeded random relabelling of rows/columns (load balance on real graphs)."""
        _check(lib().hnh_spmat_permute(self.h, seed), "spmat_permute")

    def info(self):
        o = (C.c_int64 * 4)()
        lib().hnh_spmat_info(self.h, o)
        return {"M": int(o[0]), "N": int(o[1]), "dist_nnz": int(o[2]), "local_nnz": int(o[3])}

    def free(self):
        if self.h:
            _check(lib().hnh_spmat_destroy(self.h), "spmat_destroy")
            self.h = None


class DistributedSparse:
    """A Distributed_Sparse subclass chosen by name as in benchmark_dist.cpp:45-82, plus its StandardKernel."""

    def __init__(self, world: World, alg: str, spmat: SpmatLocal, r: int, c: int):
        self.w, self.alg = world, alg
        self.h = _vp()
        _check(lib().hnh_dist_create(world.h, alg.encode(), spmat.h, r, c, C.byref(self.h)), "dist_create(%s)" % alg)

    def info(self) -> dict:
        o = (C.c_int64 * 16)()
        _check(lib().hnh_dist_info(self.h, o), "dist_info")
        names = ["M", "N", "R", "p", "c", "localArows", "localAcols", "localBrows", "localBcols", "nS", "nST", "r_split",
                 "dist_nnz", "proc_rank", "nAsub", "nBsub"]
        return {k: int(v) for k, v in zip(names, o)}

    def submatrices(self, matmode: int) -> np.ndarray:
        n = self.info()["nAsub" if matmode == AMAT else "nBsub"]
        out = np.empty((n, 4), dtype=np.int64)
        _check(lib().hnh_dist_submatrices(self.h, matmode, out.ctypes.data, n), "dist_submatrices")
        return out

    def like_A_matrix(self, v: float = 0.0) -> Dense:
        h = _vp()
        _check(lib().hnh_dense_like(self.h, AMAT, v, C.byref(h)), "like_A_matrix")
        return Dense(self.w, h)

    def like_B_matrix(self, v: float = 0.0) -> Dense:
        h = _vp()
        _check(lib().hnh_dense_like(self.h, BMAT, v, C.byref(h)), "like_B_matrix")
        return Dense(self.w, h)

    def like_S_values(self, v: float = 0.0) -> Vec:
        h = _vp()
        _check(lib().hnh_vec_like(self.h, 0, v, C.byref(h)), "like_S_values")
        return Vec(self.w, h)

    def like_ST_values(self, v: float = 0.0) -> Vec:
        h = _vp()
        _check(lib().hnh_vec_like(self.h, 1, v, C.byref(h)), "like_ST_values")
        return Vec(self.w, h)

    def setRValue(self, r: int):
        _check(lib().hnh_dist_set_r(self.h, r), "setRValue")

    def dummyInitialize(self, m: Dense, matmode: int):
        _check(lib().hnh_dense_dummy_initialize(self.h, m.h, matmode), "dummyInitialize")

    def initial_shift(self, a: Dense | None, b: Dense | None, mode: int):
        _check(lib().hnh_dist_initial_shift(self.h, a.h if a else None, b.h if b else None, mode), "initial_shift")

    def de_shift(self, a: Dense | None, b: Dense | None, mode: int):
        _check(lib().hnh_dist_de_shift(self.h, a.h if a else None, b.h if b else None, mode), "de_shift")

    def sddmmA(self, a, b, s, result):
        _check(lib().hnh_dist_sddmmA(self.h, a.h, b.h, s.h, result.h), "sddmmA")

    def sddmmB(self, a, b, s, result):
        _check(lib().hnh_dist_sddmmB(self.h, a.h, b.h, s.h, result.h), "sddmmB")

    def spmmA(self, a, b, s):
        _check(lib().hnh_dist_spmmA(self.h, a.h, b.h, s.h), "spmmA")

    def spmmB(self, a, b, s):
        _check(lib().hnh_dist_spmmB(self.h, a.h, b.h, s.h), "spmmB")

    def fusedSpMM(self, a, b, s, buf, matmode: int):
        _check(lib().hnh_dist_fusedSpMM(self.h, a.h, b.h, s.h, buf.h, matmode), "fusedSpMM")

    def hold_moving_operand(self, m=None):
        """Distributed_Sparse::hold_moving_operand(m) / release_moving_operand() (m = None)."""
        _check(lib().hnh_dist_hold_moving_operand(self.h, m.h if m else None), "hold_moving_operand")

    def walk_windows_when_held(self, on=True):
        """Distributed_Sparse::walk_windows_when_held: a held operand's resident blocks are walked by chunk windows, as a fetching call
        does (measurement entry point: one rank's kernel sequence alone on a GPU).  on = 1 / True: adaptive windows (everything has
        landed: one pass); on = 2: one pass per chunk."""
        _check(lib().hnh_dist_walk_windows_when_held(self.h, int(on)), "walk_windows_when_held")

    def fusedSpMM_out(self, a, b, matmode: int, out, leaky_alpha=None, x_scale: float = 0.0, rowdot=None) -> bool:
        """Distributed_Sparse::fusedSpMM_out; False (nothing done) when the schedule has no single fused pass."""
        ok = C.c_int(0)
        _check(lib().hnh_dist_fusedSpMM_out(self.h, a.h, b.h, matmode, out.h, int(leaky_alpha is not None), float(leaky_alpha or 0.0),
                                            float(x_scale), rowdot.h if rowdot else None, C.byref(ok)), "fusedSpMM_out")
        return bool(ok.value)

    def algorithm(self, a, b, s, result, mode: int, initial_replicate: bool):
        _check(lib().hnh_dist_algorithm(self.h, a.h, b.h, s.h, result.h if result else None, mode, int(initial_replicate)), "algorithm")

    def json_algorithm_info(self) -> dict:
        buf = C.create_string_buffer(1 << 16)
        _check(lib().hnh_dist_json(self.h, 0, buf, len(buf)), "json_algorithm_info")
        return json.loads(buf.value.decode())

    def json_perf_statistics(self) -> dict:
        buf = C.create_string_buffer(1 << 14)
        _check(lib().hnh_dist_json(self.h, 1, buf, len(buf)), "json_perf_statistics")
        return json.loads(buf.value.decode())

    def reset_performance_timers(self):
        _check(lib().hnh_dist_reset_timers(self.h), "reset_performance_timers")

    def kernel_profile(self, enable: int = -1):
        """enable: 1 start / 0 stop (both reset the counters), -1 just read.  Returns (ms, launches) BEFORE the reset."""
        ms, n = C.c_double(), C.c_int64()
        _check(lib().hnh_dist_kernel_profile(self.h, enable, C.byref(ms), C.byref(n)), "kernel_profile")
        return ms.value, n.value

    def borrow_stats(self):
        """(SpMM arrays lent, SpMM arrays copied, SDDMM results written in place, SDDMM results by a Hadamard pass) — block counts."""
        out = (C.c_int64 * 4)()
        _check(lib().hnh_dist_borrow_stats(self.h, out), "borrow_stats")
        return tuple(out)

    def free(self):
        if self.h:
            _check(lib().hnh_dist_destroy(self.h), "dist_destroy")
            self.h = None


class DistributedALS:
    """Distributed_ALS (als_conjugate_gradients.{h,cpp}): ALS by batched CG around fusedSpMM."""

    def __init__(self, op: DistributedSparse, artificial_groundtruth: bool = False, seed: int = 2022):
        self.op = op
        self.h = _vp()
        _check(lib().hnh_als_create(op.h, int(artificial_groundtruth), seed, C.byref(self.h)), "als_create")

    def set_ground_truth(self, gt_s: Vec, gt_st: Vec):
        _check(lib().hnh_als_set_ground_truth(self.h, gt_s.h, gt_st.h), "als_set_ground_truth")

    def initializeEmbeddings(self):
        _check(lib().hnh_als_initialize_embeddings(self.h), "initializeEmbeddings")

    def set_embeddings(self, a: Dense, b: Dense):
        _check(lib().hnh_als_set_embeddings(self.h, a.h, b.h), "als_set_embeddings")

    def get_embeddings(self, a: Dense, b: Dense):
        _check(lib().hnh_als_get_embeddings(self.h, a.h, b.h), "als_get_embeddings")

    def cg_optimizer(self, matmode: int, cg_max_iter: int):
        _check(lib().hnh_als_cg_optimizer(self.h, matmode, cg_max_iter), "cg_optimizer")

    def run_cg(self, steps: int):
        _check(lib().hnh_als_run_cg(self.h, steps), "run_cg")

    def computeResidual(self) -> float:
        out = C.c_double()
        _check(lib().hnh_als_compute_residual(self.h, C.byref(out)), "computeResidual")
        return out.value

    def free(self):
        if self.h:
            _check(lib().hnh_als_destroy(self.h), "als_destroy")
            self.h = None


class GAT:
    """GAT (gat.hpp): multi-head graph-attention forward pass on top of a DistributedSparse."""

    def __init__(self, op: DistributedSparse, layers, leaky_relu_alpha: float = 0.2):
        self.op, self.layers = op, [tuple(l) for l in layers]
        spec = (C.c_int * (3 * len(layers)))(*[x for l in self.layers for x in l])
        self.h = _vp()
        _check(lib().hnh_gat_create(op.h, len(layers), spec, leaky_relu_alpha, C.byref(self.h)), "gat_create")

    def weight_shape(self, layer: int, head: int):
        o = (C.c_int64 * 2)()
        _check(lib().hnh_gat_weight_shape(self.h, layer, head, o), "gat_weight_shape")
        return int(o[0]), int(o[1])

    def set_weight(self, layer: int, head: int, w: np.ndarray):
        w = np.ascontiguousarray(w, dtype=np.float64)
        assert w.shape == self.weight_shape(layer, head)
        _check(lib().hnh_gat_set_weight(self.h, layer, head, w.ctypes.data), "gat_set_weight")

    def buffer_shape(self, index: int):
        o = (C.c_int64 * 2)()
        _check(lib().hnh_gat_buffer_shape(self.h, index, o), "gat_buffer_shape")
        return int(o[0]), int(o[1])

    def set_input(self, x: Dense):
        _check(lib().hnh_gat_set_input(self.h, x.h), "gat_set_input")

    def get_output(self, out: Dense):
        _check(lib().hnh_gat_get_output(self.h, out.h), "gat_get_output")

    def forwardPass(self):
        _check(lib().hnh_gat_forward(self.h), "forwardPass")

    def free(self):
        if self.h:
            _check(lib().hnh_gat_destroy(self.h), "gat_destroy")
            self.h = None
