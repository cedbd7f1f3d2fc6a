// Function table over the C ABI of include/hnh_kernels.h.  The host layer never links the HIP library:
// it dlopen()s it (default: libhnh_kernels.so next to this library) and fails loudly if that is not
// possible.  There is NO built-in CPU implementation; the only other implementation of this ABI in the
// repository is the test double under oracle/, which tests load explicitly by path.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include "hnh_kernels.h"
#ifdef HNH_MEASUREMENT_AIDS
#include "hnh_measurement_aids.h"
#endif

namespace hnh {

struct Backend {
    void* dl = nullptr;
    std::string path, name;

#define HNH_FN(sym) decltype(&::sym) sym = nullptr;
    HNH_FN(hnh_backend_name)
    HNH_FN(hnh_ctx_create) HNH_FN(hnh_ctx_destroy) HNH_FN(hnh_last_error) HNH_FN(hnh_ctx_stream) HNH_FN(hnh_ctx_device_identity)
    HNH_FN(hnh_malloc) HNH_FN(hnh_free) HNH_FN(hnh_memcpy) HNH_FN(hnh_memset) HNH_FN(hnh_stream_sync)
    HNH_FN(hnh_event_create) HNH_FN(hnh_event_destroy) HNH_FN(hnh_event_record) HNH_FN(hnh_event_wait)
    HNH_FN(hnh_event_sync) HNH_FN(hnh_event_query) HNH_FN(hnh_event_elapsed_ms)
    HNH_FN(hnh_sddmm_coo) HNH_FN(hnh_sddmm_csr) HNH_FN(hnh_spmm_csr) HNH_FN(hnh_fused_sddmm_spmm_csr)
    HNH_FN(hnh_sddmm_csr_ex) HNH_FN(hnh_spmm_csr_ex) HNH_FN(hnh_fused_sddmm_spmm_csr_ex) HNH_FN(hnh_csr_max_row_nnz)
    HNH_FN(hnh_fused_sddmm_spmm_csr_x) HNH_FN(hnh_row_epilogue_f64) HNH_FN(hnh_row_epilogue_x) HNH_FN(hnh_cg_step_f64)
    HNH_FN(hnh_tuples_sort) HNH_FN(hnh_tuples_bucket_starts) HNH_FN(hnh_tuples_transform) HNH_FN(hnh_tuples_to_csr)
    HNH_FN(hnh_csr_window_bounds) HNH_FN(hnh_sddmm_csr_w) HNH_FN(hnh_spmm_csr_w) HNH_FN(hnh_fused_sddmm_spmm_csr_w) HNH_FN(hnh_tuples_remap_cols) HNH_FN(hnh_tuples_dedup_max) HNH_FN(hnh_tuples_take_strided)
    HNH_FN(hnh_panel_count) HNH_FN(hnh_generate_er_keys) HNH_FN(hnh_generate_rmat_keys) HNH_FN(hnh_tuples_from_keys) HNH_FN(hnh_tuples_relabel)
    HNH_FN(hnh_fill_f64) HNH_FN(hnh_hadamard_f64) HNH_FN(hnh_axpy_f64) HNH_FN(hnh_expand_rowptr) HNH_FN(hnh_sum_chunked_blocks_f64)
    HNH_FN(hnh_rowdot_f64) HNH_FN(hnh_row_scale_add_f64) HNH_FN(hnh_vec_add_scalar_f64) HNH_FN(hnh_vec_div_f64) HNH_FN(hnh_fill_hashed_f64)
    HNH_FN(hnh_gemm_f64) HNH_FN(hnh_leaky_relu_f64) HNH_FN(hnh_relu_store_cols_f64)
    HNH_FN(hnh_comm_unique_id) HNH_FN(hnh_comm_init) HNH_FN(hnh_comm_split) HNH_FN(hnh_comm_destroy) HNH_FN(hnh_comm_identity)
    HNH_FN(hnh_comm_sendrecv) HNH_FN(hnh_comm_group_begin) HNH_FN(hnh_comm_group_end) HNH_FN(hnh_comm_allgather) HNH_FN(hnh_comm_reduce_scatter_f64)
    HNH_FN(hnh_comm_allreduce_f64)
    HNH_FN(hnh_ipc_export) HNH_FN(hnh_ipc_open) HNH_FN(hnh_ipc_close) HNH_FN(hnh_ipc_pull) HNH_FN(hnh_ipc_flags_register) HNH_FN(hnh_ipc_flags_unregister)
    HNH_FN(hnh_stream_write_flag) HNH_FN(hnh_stream_wait_flag)
    HNH_FN(hnh_csr_plan_create) HNH_FN(hnh_csr_plan_destroy) HNH_FN(hnh_sddmm_csr_p) HNH_FN(hnh_sddmm_csr_ps) HNH_FN(hnh_spmm_csr_p) HNH_FN(hnh_spmm_csr_pf) HNH_FN(hnh_fused_sddmm_spmm_csr_p)
#ifdef HNH_MEASUREMENT_AIDS
    HNH_FN(hnh_stream_delay_us) HNH_FN(hnh_stream_paced_copy) HNH_FN(hnh_stream_pace_begin) HNH_FN(hnh_stream_pace_end)
#endif
#undef HNH_FN
};

// Loads (once per path) and returns the backend; path == nullptr or "" selects the product HIP library.
// Calls hnh::fatal() (print + exit(1) / exception) if the library or any symbol is missing.
Backend* load_backend(const char* path);
Backend* default_backend();           // the most recently loaded one; loads the HIP library on first use
std::string default_backend_path();   // <dir of libhnh_host.so>/libhnh_kernels.so

}  // namespace hnh
