"""ctypes binding of include/hnh_kernels.h (lib/libhnh_kernels.so).  Fails loudly if the HIP library
is missing or no GPU is present — there is no fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HNH_KERNEL_LIB_DEV") or os.path.join(HERE, "lib", "libhnh_kernels.so")  # override: kernel tuning experiments only

OK = 0
STREAM_COMPUTE, STREAM_COMM, STREAM_AUX = 0, 1, 2
H2D, D2H, D2D = 0, 1, 2
FUSED_VALUES_OVERWRITE, FUSED_OUT_OVERWRITE, FUSED_LEAKY_RELU = 1, 2, 4
UNIQUE_ID_BYTES = 128

_vp, _i32, _i64, _dbl, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t

# name -> (restype, argtypes): every symbol include/hnh_kernels.h declares
SIGNATURES = {
    "hnh_backend_name": (C.c_char_p, []),
    "hnh_ctx_create": (_i32, [_i32, C.POINTER(_vp)]),
    "hnh_ctx_destroy": (_i32, [_vp]),
    "hnh_last_error": (C.c_char_p, [_vp]),
    "hnh_ctx_stream": (_vp, [_vp, _i32]),
    "hnh_ctx_device_identity": (_i32, [_vp, C.POINTER(_i32), C.c_char_p, _i32]),
    "hnh_malloc": (_i32, [_vp, _sz, C.POINTER(_vp)]),
    "hnh_free": (_i32, [_vp, _vp]),
    "hnh_memcpy": (_i32, [_vp, _vp, _vp, _sz, _i32, _i32]),
    "hnh_memset": (_i32, [_vp, _vp, _i32, _sz, _i32]),
    "hnh_stream_sync": (_i32, [_vp, _i32]),
    "hnh_event_create": (_i32, [_vp, C.POINTER(_vp)]),
    "hnh_event_destroy": (_i32, [_vp, _vp]),
    "hnh_event_record": (_i32, [_vp, _vp, _i32]),
    "hnh_event_wait": (_i32, [_vp, _vp, _i32]),
    "hnh_event_sync": (_i32, [_vp, _vp]),
    "hnh_event_query": (_i32, [_vp, _vp, _vp]),
    "hnh_event_elapsed_ms": (_i32, [_vp, _vp, _vp, C.POINTER(C.c_float)]),
    "hnh_sddmm_coo": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32]),
    "hnh_sddmm_csr": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32]),
    "hnh_spmm_csr": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32]),
    "hnh_fused_sddmm_spmm_csr": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, C.c_uint, _i32]),
    "hnh_sddmm_csr_ex": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i64, _i32]),
    "hnh_spmm_csr_ex": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i64, _i32]),
    "hnh_fused_sddmm_spmm_csr_ex": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, C.c_uint, _i64, _i32, _i64, _i32]),
    "hnh_panel_count": (_i32, [_vp, _i64, _i64, _i64, _i32, _i32]),
    "hnh_csr_window_bounds": (_i32, [_vp, _i64, _vp, _vp, _i32, _vp, _vp, _i32]),
    "hnh_sddmm_csr_w": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _i32]),
    "hnh_spmm_csr_w": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _i32]),
    "hnh_fused_sddmm_spmm_csr_w": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, C.c_uint, _i64, _i32, _vp, _vp, _i32]),
    "hnh_csr_max_row_nnz": (_i32, [_vp, _i64, _vp, C.POINTER(C.c_int), _i32]),
    "hnh_fused_sddmm_spmm_csr_x": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, C.c_uint, _i64, _i32, _i64, _vp, _i32]),
    "hnh_row_epilogue_f64": (_i32, [_vp, _vp, _vp, C.c_double, _vp, _i64, _i32, _i32]),
    "hnh_row_epilogue_x": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32]),
    "hnh_cg_step_f64": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32]),
    "hnh_tuples_sort": (_i32, [_vp, _vp, _i64, _vp, _i32, _i32]),
    "hnh_tuples_bucket_starts": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _i32]),
    "hnh_tuples_transform": (_i32, [_vp, _vp, _i64, _i32, C.c_uint64, C.c_uint64, _i32]),
    "hnh_tuples_remap_cols": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i32]),
    "hnh_tuples_dedup_max": (_i32, [_vp, _vp, _i64, C.POINTER(C.c_int64), _i32]),
    "hnh_tuples_take_strided": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _i32]),
    "hnh_generate_er_keys": (_i32, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _vp, C.POINTER(C.c_int64), _i32]),
    "hnh_generate_rmat_keys": (_i32, [_vp, _i32, C.c_uint64, _dbl, _dbl, _dbl, C.c_uint64, _i32, _vp, C.POINTER(C.c_int64), _i32]),
    "hnh_tuples_from_keys": (_i32, [_vp, _vp, C.c_uint64, _i64, _i64, C.c_double, _vp, _i64, _i32]),
    "hnh_tuples_relabel": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32]),
    "hnh_tuples_to_csr": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, C.POINTER(C.c_int), _i32]),
    "hnh_fill_f64": (_i32, [_vp, _vp, _i64, _dbl, _i32]),
    "hnh_hadamard_f64": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32]),
    "hnh_axpy_f64": (_i32, [_vp, _vp, _vp, _dbl, _i64, _i32]),
    "hnh_expand_rowptr": (_i32, [_vp, _i64, _vp, _vp, _i32]),
    "hnh_rowdot_f64": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32]),
    "hnh_row_scale_add_f64": (_i32, [_vp, _vp, _vp, _dbl, _vp, _vp, _dbl, _i64, _i32, _i32]),
    "hnh_vec_add_scalar_f64": (_i32, [_vp, _vp, _dbl, _i64, _i32]),
    "hnh_vec_div_f64": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32]),
    "hnh_fill_hashed_f64": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, C.c_uint64, _dbl, _i32]),
    "hnh_gemm_f64": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _i32]),
    "hnh_leaky_relu_f64": (_i32, [_vp, _vp, _dbl, _i64, _i32]),
    "hnh_relu_store_cols_f64": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i32]),
    "hnh_comm_unique_id": (_i32, [_vp]),
    "hnh_comm_init": (_i32, [_vp, _i32, _i32, _vp, C.POINTER(_vp)]),
    "hnh_comm_split": (_i32, [_vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "hnh_comm_destroy": (_i32, [_vp, _vp]),
    "hnh_comm_identity": (_i32, [_vp, _vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "hnh_comm_sendrecv": (_i32, [_vp, _vp, _vp, _sz, _i32, _vp, _sz, _i32, _i32]),
    "hnh_comm_group_begin": (_i32, [_vp]),
    "hnh_comm_group_end": (_i32, [_vp]),
    "hnh_comm_allgather": (_i32, [_vp, _vp, _vp, _vp, _sz, _i32]),
    "hnh_comm_reduce_scatter_f64": (_i32, [_vp, _vp, _vp, _vp, _sz, _i32]),
    "hnh_comm_allreduce_f64": (_i32, [_vp, _vp, _vp, _vp, _sz, _i32]),
    "hnh_ipc_export": (_i32, [_vp, _vp, _vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "hnh_ipc_open": (_i32, [_vp, _vp, C.c_uint64, C.POINTER(_vp)]),
    "hnh_ipc_close": (_i32, [_vp, _vp]),
    "hnh_ipc_pull": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32]),
    "hnh_ipc_flags_register": (_i32, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "hnh_ipc_flags_unregister": (_i32, [_vp, _vp]),
    "hnh_stream_write_flag": (_i32, [_vp, _i32, _vp, C.c_uint64]),
    "hnh_stream_wait_flag": (_i32, [_vp, _i32, _vp, C.c_uint64]),
    "hnh_csr_plan_create": (_i32, [_vp, C.POINTER(_vp)]),
    "hnh_csr_plan_destroy": (_i32, [_vp, _vp]),
    "hnh_sddmm_csr_p": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, C.c_uint, _vp, _i32]),
    "hnh_sddmm_csr_ps": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, C.c_uint, _vp, _i32]),
    "hnh_spmm_csr_p": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32]),
    "hnh_sum_chunked_blocks_f64": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32]),
    "hnh_spmm_csr_pf": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, C.c_uint, _vp, _i32]),
    "hnh_fused_sddmm_spmm_csr_p": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, C.c_uint, _vp, _vp, _i32]),
}


# include/hnh_measurement_aids.h (tools and their tests only)
AIDS_SIGNATURES = {
    "hnh_stream_delay_us": (_i32, [_vp, _i32, C.c_double]),
    "hnh_stream_paced_copy": (_i32, [_vp, _i32, _vp, _vp, C.c_size_t, _i32, C.c_double, _i32]),
    "hnh_stream_pace_begin": (_i32, [_vp, _i32]),
    "hnh_stream_pace_end": (_i32, [_vp, _i32, C.c_double]),
}


class CsrBlock(C.Structure):
    """struct hnh_csr_block"""
    _fields_ = [("rows", C.c_int64), ("nnz", C.c_int64), ("cols", C.c_int64), ("max_row_nnz", C.c_int32), ("reserved", C.c_int32),
                ("rowptr", C.c_void_p), ("col_idx", C.c_void_p), ("plan", C.c_void_p)]

class CgUpdate(C.Structure):
    """struct hnh_cg_update"""
    _fields_ = [("x", C.c_void_p), ("r", C.c_void_p), ("p", C.c_void_p), ("rsold", C.c_void_p), ("eps", C.c_double)]


class FusedExtras(C.Structure):
    """struct hnh_fused_extras"""
    _fields_ = [("leaky_alpha", C.c_double), ("x_scale", C.c_double), ("rowdot", C.c_void_p), ("cg", C.POINTER(CgUpdate)),
                ("relu_dst", C.c_void_p), ("relu_ld", C.c_int64)]


class TupleKey(C.Structure):
    """struct hnh_tuple_key"""
    _fields_ = [("kind", C.c_int), ("transpose", C.c_int), ("rows_in_block", C.c_int64), ("cols_in_block", C.c_int64),
                ("n_col_blocks", C.c_int64), ("owner_table", C.c_void_p), ("div", C.c_int64)]


KEY_ROW_COL, KEY_COL_ROW, KEY_OWNER, KEY_COL_DIV = 0, 1, 2, 3
TUPLE_DTYPE = [("r", "<u8"), ("c", "<u8"), ("value", "<f8")]  # struct hnh_tuple


class CsrWindow(C.Structure):
    """struct hnh_csr_window"""
    _fields_ = [("beg", _vp), ("end", _vp), ("last", _i32)]


_lib = None


def load(path: str | None = None) -> C.CDLL:
    """dlopen the HIP kernel library and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError("HIP kernel library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % p)
    # One HIP runtime per process: PyTorch bundles its own libamdhip64/librccl (same SONAMEs as
    # /opt/rocm's).  If torch is going to be used (bench.py, torch.distributed bootstrap) it must be
    # imported BEFORE this library so that both bind to the same runtime; loading in the other order
    # leaves two allocators fighting at process exit.
    if os.environ.get("HNH_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(p)  # RTLD_LOCAL: the oracle test double exports the same symbol names
    for name, (res, args) in list(SIGNATURES.items()) + list(AIDS_SIGNATURES.items()):
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if path is None:
        _lib = lib
    return lib


class HnhError(RuntimeError):
    pass


class Ctx:
    """One hnh_ctx (device + compute/comm streams).  Raises if there is no GPU."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = _vp()
        rc = self.lib.hnh_ctx_create(device, C.byref(h))
        if rc != OK:
            raise HnhError("hnh_ctx_create(device=%d) failed with status %d: no usable MI355X (HIP) device" % (device, rc))
        self.h = h

    def check(self, rc: int, what: str = ""):
        if rc != OK:
            raise HnhError("%s failed (%d): %s" % (what, rc, self.lib.hnh_last_error(self.h).decode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.hnh_ctx_destroy(self.h)
            self.h = None

    # ---- memory helpers
    def alloc(self, nbytes: int) -> int:
        p = _vp()
        self.check(self.lib.hnh_malloc(self.h, nbytes, C.byref(p)), "hnh_malloc")
        return p.value

    def free(self, ptr: int):
        self.check(self.lib.hnh_free(self.h, ptr), "hnh_free")

    def upload(self, arr: np.ndarray) -> "DevArray":
        return DevArray.from_host(self, arr)

    def sync(self, stream: int = STREAM_COMPUTE):
        self.check(self.lib.hnh_stream_sync(self.h, stream), "hnh_stream_sync")


class DevArray:
    """A typed device allocation owned by Python (test / bench plumbing only)."""

    def __init__(self, ctx: Ctx, shape, dtype):
        self.ctx, self.shape, self.dtype = ctx, tuple(np.atleast_1d(shape)), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = ctx.alloc(self.nbytes)

    @classmethod
    def from_host(cls, ctx: Ctx, arr: np.ndarray) -> "DevArray":
        arr = np.ascontiguousarray(arr)
        d = cls(ctx, arr.shape, arr.dtype)
        d.set(arr)
        return d

    def set(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes
        self.ctx.check(self.ctx.lib.hnh_memcpy(self.ctx.h, self.ptr, arr.ctypes.data, self.nbytes, H2D, STREAM_COMPUTE), "h2d")
        self.ctx.sync()

    def get(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        self.ctx.sync()
        self.ctx.check(self.ctx.lib.hnh_memcpy(self.ctx.h, out.ctypes.data, self.ptr, self.nbytes, D2H, STREAM_COMPUTE), "d2h")
        self.ctx.sync()
        return out

    def free(self):
        if self.ptr:
            self.ctx.free(self.ptr)
            self.ptr = None
