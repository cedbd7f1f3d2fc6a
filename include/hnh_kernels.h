/*
 * hnh_kernels.h — C ABI of the MI355X (gfx950) local-kernel library  (libhnh_kernels.so)
 *
 * This is the drop-in boundary for the hot path of PASSIONLab/distributed_sddmm ("HnH"): the bodies of
 * the reference's plugin class `StandardKernel` (sparse_kernels.h:84-99) marshal to these entry points.
 * Plain pointers and sizes only; every data pointer is a DEVICE pointer unless a comment says "host".
 * Every function returns an int status (HNH_OK == 0); no exceptions cross this boundary; the text of
 * the last error of a context is available through hnh_last_error().
 *
 * Conventions (all from the reference, file:line relative to /root/reference):
 *   - dense matrices are row-major fp64 with leading dimension == number of columns == R
 *     (common.h:13; asserted equal for A and B at sparse_kernels.cpp:21);
 *   - a sparse block is CSR (rowptr / col_idx / values) of the *active* buffer of a CSRLocal
 *     (SpmatLocal.hpp:55-62,171-179).  Device indices are 32-bit (per-rank nnz < 2^31 is already a
 *     constraint of the reference, which counts nonzeros in `int`: SpmatLocal.hpp:68);
 *   - SDDMM ACCUMULATES into `values` (sparse_kernels.cpp:54) and SpMM uses alpha = 1, beta = 1
 *     (sparse_kernels.cpp:97,104): callers zero the destination first, exactly as in the reference.
 *
 * `stream` arguments select one of the context's HIP streams: HNH_STREAM_COMPUTE, HNH_STREAM_COMM or
 * HNH_STREAM_AUX (a second compute stream for work that may run BESIDE the compute stream's: the GAT
 * forward pass puts the next head's MFMA-bound GEMM there while the HBM-bound fused pass of the current
 * head runs; ordered against the others with events like any stream).
 * All kernels and copies are asynchronous with respect to the host.
 */
#ifndef HNH_KERNELS_H
#define HNH_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HNH_OK 0
#define HNH_ERR_INVALID 1     /* bad argument (null pointer, negative size, R <= 0 ...) */
#define HNH_ERR_DEVICE 2      /* HIP / RCCL runtime error, see hnh_last_error()           */
#define HNH_ERR_NOMEM 3
#define HNH_ERR_UNSUPPORTED 4

#define HNH_STREAM_COMPUTE 0
#define HNH_STREAM_COMM 1
#define HNH_STREAM_AUX 2
#define HNH_STREAMS 3

#define HNH_COPY_H2D 0
#define HNH_COPY_D2H 1
#define HNH_COPY_D2D 2

/* flags of hnh_fused_sddmm_spmm_csr */
#define HNH_FUSED_VALUES_OVERWRITE 1u /* values[e]  = dot  instead of  values[e] += dot (caller knows they are zero) */
#define HNH_FUSED_OUT_OVERWRITE 2u    /* Out[i,:]   = sum  instead of  Out[i,:] += sum  (caller knows it is zero)    */
#define HNH_FUSED_LEAKY_RELU 4u       /* _x entry points: activation between the two halves, see hnh_fused_extras     */

typedef struct hnh_ctx hnh_ctx; /* opaque, one per rank: device ordinal, two streams, scratch */

/* Identifies the implementation behind this ABI; the product library returns "hip-gfx950". */
const char* hnh_backend_name(void);

/* ---- context, memory, streams, events ------------------------------------------------------------
 * Replaces: host allocation of Eigen matrices / std::vector storage in the reference
 * (common.h:13, SpmatLocal.hpp:55-62); the reference has no device. */
int hnh_ctx_create(int device, hnh_ctx** out);
int hnh_ctx_destroy(hnh_ctx* ctx);
const char* hnh_last_error(hnh_ctx* ctx);
void* hnh_ctx_stream(hnh_ctx* ctx, int stream); /* the raw hipStream_t, for interop (RCCL, torch) */
/* Which physical device the context runs on: the ordinal it was created with and the device's PCI bus id ("0000:c1:00.0"; at most
 * len - 1 characters + NUL, len >= 16).  Two ranks of one job that report the same bus id share ONE GPU — what a benchmark line of
 * N processes has to be able to prove about itself (the reference's MPI ranks are host processes: no counterpart there). */
int hnh_ctx_device_identity(hnh_ctx* ctx, int* ordinal, char* pci_bus_id, int len);
int hnh_malloc(hnh_ctx* ctx, size_t bytes, void** out);
int hnh_free(hnh_ctx* ctx, void* ptr);
int hnh_memcpy(hnh_ctx* ctx, void* dst, const void* src, size_t bytes, int kind, int stream);
int hnh_memset(hnh_ctx* ctx, void* dst, int byte, size_t bytes, int stream);
int hnh_stream_sync(hnh_ctx* ctx, int stream);
int hnh_event_create(hnh_ctx* ctx, void** event);
int hnh_event_destroy(hnh_ctx* ctx, void* event);
int hnh_event_record(hnh_ctx* ctx, void* event, int stream);
int hnh_event_wait(hnh_ctx* ctx, void* event, int stream); /* `stream` waits for `event` (device side) */
int hnh_event_sync(hnh_ctx* ctx, void* event);            /* host waits for `event`                    */
int hnh_event_query(hnh_ctx* ctx, void* event, int* done); /* host asks: *done = 1 when `event` has completed, never blocks */
int hnh_event_elapsed_ms(hnh_ctx* ctx, void* start, void* stop, float* ms);
/* ---- local kernels --------------------------------------------------------------------------------
 * hnh_sddmm_coo — replaces StandardKernel::sddmm_local (sparse_kernels.cpp:13-57), COO view:
 *     for e in [0, nnz):  values[e] += < X[row_idx[e], :], Y[col_idx[e], :] >
 *   X, Y are the (possibly role-swapped, sparse_kernels.cpp:29-37) dense operands, R columns each. */
int hnh_sddmm_coo(hnh_ctx* ctx, int64_t nnz, const int32_t* row_idx, const int32_t* col_idx, double* values,
                  const double* X, const double* Y, int R, int stream);

/* hnh_sddmm_csr — same arithmetic as hnh_sddmm_coo on the CSR view of the same block (row_idx is the
 *   expansion of rowptr, SpmatLocal.hpp:139-147); the X row is staged once per sparse row. */
int hnh_sddmm_csr(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                  const double* X, const double* Y, int R, int stream);

/* hnh_spmm_csr — replaces StandardKernel::spmm_local's mkl_sparse_d_mm call (sparse_kernels.cpp:95-121):
 *     Out[i, :] += sum_{e in row i} values[e] * X[col_idx[e], :]        (alpha = 1, beta = 1)
 *   Amat mode: Out = A, X = B; Bmat mode (block stored transposed): Out = B, X = A. */
int hnh_spmm_csr(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, const double* values,
                 const double* X, double* Out, int R, int stream);

/* hnh_fused_sddmm_spmm_csr — the "local kernel fusion" pair of calls in
 *   Sparse15D_Dense_Shift::fusedSpMM (15D_dense_shift.hpp:203-217) as ONE pass over the block:
 *     for each row i, each nonzero e=(i,j):  d = < X[i,:], Y[j,:] >;  values[e] (+)= d;
 *                                            Out[i,:] (+)= w_e * Y[j,:]
 *   with w_e = values[e] after the update (the reference never applies Svalues on this path), or
 *   w_e = svalues[e] * values[e] when `svalues` is non-null (extension; not used for parity).
 *   One gather of Y[j,:] serves both the dot product and the axpy. X and Out must not alias. */
int hnh_fused_sddmm_spmm_csr(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx,
                             double* values, const double* svalues, const double* X, const double* Y,
                             double* Out, int R, unsigned flags, int stream);

/* Variants for callers that know the block: `nnz` = rowptr[rows] and `max_row_nnz` = its longest row (either may
 * be -1 = unknown).  Hub rows — rows longer than 3 x the block's mean row length, taken in steps of 64 and within
 * [256, 1024] (1024 when nnz is unknown) — are cut into 256-nonzero
 * segments that a second small launch spreads over the whole chip (SpMM / fused segments write partial output rows
 * that a third launch adds up in a fixed order: results are bit-identical run to run); with max_row_nnz below that
 * threshold none of that machinery runs.  The plain entry points above pass -1, -1.
 * hnh_csr_max_row_nnz computes the hint (one device reduction + 4-byte synchronous copy).
 * cols = number of rows of the gathered dense operand (= columns of the sparse block), or -1.  When it is given and the
 * operand is larger than ~768 MiB the pass runs as several launches, one per ~512 MiB COLUMN PANEL of the block
 * (column indices are sorted within a CSR row, so a panel is a contiguous piece of every row; the per-row boundaries are
 * found by a small kernel first): a launch then revisits a cache-sized part of the operand and about half of its gathers
 * hit the 256 MiB Infinity Cache — measured 16.8 -> 14.8 ms at config 2.  Same arithmetic, same results. */
int hnh_sddmm_csr_ex(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                     const double* X, const double* Y, int R, int64_t nnz, int max_row_nnz, int64_t cols, int stream);
int hnh_spmm_csr_ex(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, const double* values,
                    const double* X, double* Out, int R, int64_t nnz, int max_row_nnz, int64_t cols, int stream);
int hnh_fused_sddmm_spmm_csr_ex(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx,
                                double* values, const double* svalues, const double* X, const double* Y, double* Out,
                                int R, unsigned flags, int64_t nnz, int max_row_nnz, int64_t cols, int stream);
int hnh_csr_max_row_nnz(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, int* out_host, int stream);
/* number of kernel launches (column panels) a row pass with these hints is run as; 1 = a single launch (profiling aid) */
int hnh_panel_count(hnh_ctx* ctx, int64_t rows, int64_t nnz, int64_t cols, int R, int max_row_nnz);

/* Row WINDOWS.  A window restricts a row pass to a contiguous piece [beg[r], end[r]) of every CSR row (column indices are
 * sorted within a row, so "the nonzeros whose column lies in [c0, c1)" is such a piece).  The 1.5D dense-shift schedule keeps
 * all blocks fetched from the other ranks as ONE CSR block whose columns index the landing buffer in arrival order, and runs
 * one windowed pass per arrived chunk, overlapping kernels with the fetch (15D_dense_shift.hpp:199-227 walks the same
 * nonzeros block by block); the kernel library's own Infinity-Cache panels are the same mechanism with automatic bounds.
 *   hnh_csr_window_bounds  split[b * rows + r] = first nonzero of row r with column >= bounds_host[b]  (b < nbounds <= 15)
 *   *_w entry points       as the _ex / _x entry points, on the window only.  Hub rows (see the _ex entry points) are left
 *                          whole: they are skipped by every window and processed — over their whole length — by the call
 *                          whose window has `last` set, which is also the call that applies a row epilogue. */
typedef struct hnh_csr_window {
    const int32_t* beg; /* device, rows entries; NULL = rowptr */
    const int32_t* end; /* device, rows entries; NULL = rowptr + 1 */
    int last;           /* non-zero: the pass's last window */
} hnh_csr_window;
int hnh_csr_window_bounds(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, int nbounds,
                          const int32_t* bounds_host, int32_t* split, int stream);
int hnh_sddmm_csr_w(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values, const double* X,
                    const double* Y, int R, int64_t nnz, int max_row_nnz, const hnh_csr_window* window, int stream);
int hnh_spmm_csr_w(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, const double* values, const double* X,
                   double* Out, int R, int64_t nnz, int max_row_nnz, const hnh_csr_window* window, int stream);

/* Extras of the fused pass — what the reference's applications do immediately around their SDDMM->SpMM pair,
 * folded into the same launch while the operands are still in registers:
 *   leaky_alpha  with HNH_FUSED_LEAKY_RELU: the SpMM half uses (and values[] keeps)
 *                w = LeakyReLU(svalues[e] * (values[e] + <X[i,:],Y[j,:]>))   — gat.hpp:96-99 (SDDMM, activation, SpMM)
 *   x_scale      != 0:  Out[i,:] += x_scale * X[i,:]                         — als_conjugate_gradients.cpp:282,295 (+ lambda * X)
 *   rowdot       != NULL: rowdot[i] = <X[i,:], Out[i,:]> of the FINAL row     — als_conjugate_gradients.cpp:93 (batch_dot_product(p, Mp))
 * The epilogue (x_scale, rowdot) runs inside the launch when one group completes the output row; with hub rows
 * (segments combined after the launch) or column tiles it is appended as a row-wise
 * launch — same result either way.  hnh_row_epilogue_f64 is that launch on its own.
 *
 *   cg != NULL: the REST of one batched-CG iteration (als_conjugate_gradients.cpp:91-139) runs on the finished row too,
 *   with p = the fused call's row operand X (cg->p must be that pointer, writable) and Mp = the final output row:
 *       bdot  = <p[i,:], Mp[i,:]> + eps;      rs = rsold[i] + eps;       alpha = rs / bdot             (:91-101)
 *       x[i,:] += alpha * p[i,:];             r[i,:] -= alpha * Mp[i,:];                                 (:112-118)
 *       rsnew = <r[i,:], r[i,:]>;             p[i,:] = r[i,:] + (rsnew / rs) * p[i,:];   rsold[i] = rsnew   (:120-139)
 *   Row i of p is read only by the group that owns output row i, so updating it in place is safe.  Only valid when the
 *   whole row lives on this rank (no R split: the reference all-reduces bdot and rsnew otherwise, :95-97,122-124).
 *   Replaces five further dense passes per CG iteration by loads/stores of rows that are already in registers. */
typedef struct hnh_cg_update {
    double* x;     /* the factor being optimised (rows x R) */
    double* r;     /* residual (rows x R) */
    double* p;     /* search direction = the fused call's X */
    double* rsold; /* rows: <r, r> of the previous iteration in, of this one out */
    double eps;    /* nan_avoidance_constant (:44) */
} hnh_cg_update;
typedef struct hnh_fused_extras {
    double leaky_alpha;
    double x_scale;
    double* rowdot;
    const hnh_cg_update* cg;
    /* relu_dst != NULL: the finished row goes, through a ReLU, into a column block of a wider matrix —
     *   relu_dst[i * relu_ld + j] = max(Out[i, j], 0), j < R     (gat.hpp:101: `buffers[i+1].middleCols(...) = A.cwiseMax(0)`)
     * — and Out is scratch (it holds partial sums between the launches of a pass; its final contents are undefined).
     * relu_dst already points at the block's first column.  Not together with cg. */
    double* relu_dst;
    int64_t relu_ld;
} hnh_fused_extras;
int hnh_fused_sddmm_spmm_csr_x(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                               const double* svalues, const double* X, const double* Y, double* Out, int R, unsigned flags,
                               int64_t nnz, int max_row_nnz, int64_t cols, const hnh_fused_extras* extras, int stream);
int hnh_fused_sddmm_spmm_csr_w(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                               const double* svalues, const double* X, const double* Y, double* Out, int R, unsigned flags,
                               int64_t nnz, int max_row_nnz, const hnh_fused_extras* extras, const hnh_csr_window* window, int stream);
int hnh_row_epilogue_f64(hnh_ctx* ctx, double* Out, const double* X, double x_scale, double* rowdot, int64_t rows, int R, int stream);
/* the same launch taking the whole extras record (x_scale, rowdot, cg; leaky_alpha is not an epilogue and is ignored) */
int hnh_row_epilogue_x(hnh_ctx* ctx, double* Out, const double* X, const hnh_fused_extras* extras, int64_t rows, int R, int stream);

/* ---- element-wise helpers (K3-K5 of SURVEY §2.4) ----------------------------------------------------
 * hnh_fill_f64      — SpmatLocal::setValuesConstant (SpmatLocal.hpp:595-605), DenseMatrix::setZero
 * hnh_hadamard_f64  — `SValues.cwiseProduct(choice->getCSRValues())` (15D_dense_shift.hpp:366)
 * hnh_axpy_f64      — y += alpha * x : local reduction step of MPI_Reduce_scatter / `tmp *= 0.0`
 * hnh_expand_rowptr — rebuilds COO row_idx from rowptr (SpmatLocal.hpp:139-147) */
int hnh_fill_f64(hnh_ctx* ctx, double* dst, int64_t n, double value, int stream);
int hnh_hadamard_f64(hnh_ctx* ctx, double* out, const double* a, const double* b, int64_t n, int stream);
int hnh_axpy_f64(hnh_ctx* ctx, double* y, const double* x, double alpha, int64_t n, int stream);
int hnh_expand_rowptr(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, int32_t* row_idx, int stream);
/* The closing step of a mesh reduce-scatter (replaces the additions the reference's accumulator collects on its way round the ring,
 * 15D_dense_shift.hpp:331-356): `src` holds `nblocks` partial copies of a block of cuts[nchunks] rows x R in CHUNK-MAJOR order — the
 * rows [cuts[q], cuts[q+1]) of block k start at row nblocks * cuts[q] + k * (cuts[q+1] - cuts[q]) — and
 *     dst[i,:] += src(block 0, row i) + src(block 1, row i) + ... (in block order)   for the rows i of chunks [q0, q1).
 * cuts_host: nchunks + 1 non-decreasing row numbers on the HOST, cuts[0] = 0; nchunks <= HNH_MAX_CHUNKS. */
#define HNH_MAX_CHUNKS 12
int hnh_sum_chunked_blocks_f64(hnh_ctx* ctx, double* dst, const double* src, int nblocks, int nchunks, const int64_t* cuts_host, int q0, int q1,
                               int R, int stream);

/* ---- setup on the device: routing, ordering and CSR conversion of the matrix's (row, col, value) tuples ---------------
 * Replaces host code of the reference's SpmatLocal.hpp: getOwner + the Alltoallv pack (:45-52, :404-420), the
 * column-major std::sort (:454), divideIntoBlockCols (:541-563) and the MKL COO->CSR conversion in CSRLocal's
 * constructor (:117-147).  `hnh_tuple` is the reference's spcoord_t (common.h:27-33); arrays of it live in DEVICE memory.
 *
 * A key maps a tuple to an unsigned integer:
 *   HNH_KEY_ROW_COL  (r << 32) | c   — CSR order;   HNH_KEY_COL_ROW  (c << 32) | r — column-major (indices must fit 32 bits)
 *   HNH_KEY_OWNER    owner_table[(R / rows_in_block) * n_col_blocks + (C / cols_in_block)] with (R, C) = (r, c), or (c, r)
 *                    when `transpose` — NonzeroDistribution::getOwner; owner_table is a DEVICE array
 *   HNH_KEY_COL_DIV  c / div         — block column of divideIntoBlockCols
 * hnh_tuples_sort          stable in-place sort by key (LSD radix sort of (key, index) pairs + one gather); key_bits =
 *                          number of significant key bits (<= 0: 64), fewer bits = fewer radix passes
 * hnh_tuples_bucket_starts for tuples whose keys are non-decreasing: starts_host[b] = first index with key >= b,
 *                          b = 0..nbuckets (send counts per owner, block-column boundaries); synchronous
 * hnh_tuples_transform     r <-> c when swap_rc, then r %= rmod, c %= cmod (0 = leave): transposition and the
 *                          "make indices block-local" loops of the schedule constructors (15D_dense_shift.hpp:96-100)
 * hnh_tuples_to_csr        tuples in HNH_KEY_ROW_COL order -> rowptr (rows + 1), col_idx, values of one block; reports the
 *                          longest row; HNH_ERR_INVALID when a tuple lies outside rows x cols; synchronous */
typedef struct hnh_tuple {
    uint64_t r, c;
    double value;
} hnh_tuple;
#define HNH_KEY_ROW_COL 0
#define HNH_KEY_COL_ROW 1
#define HNH_KEY_OWNER 2
#define HNH_KEY_COL_DIV 3
typedef struct hnh_tuple_key {
    int kind;
    int transpose;               /* HNH_KEY_OWNER */
    int64_t rows_in_block;       /* HNH_KEY_OWNER */
    int64_t cols_in_block;       /* HNH_KEY_OWNER */
    int64_t n_col_blocks;        /* HNH_KEY_OWNER */
    const int32_t* owner_table;  /* HNH_KEY_OWNER, device */
    int64_t div;                 /* HNH_KEY_COL_DIV */
} hnh_tuple_key;
int hnh_tuples_sort(hnh_ctx* ctx, hnh_tuple* tuples, int64_t n, const hnh_tuple_key* key, int key_bits, int stream);
int hnh_tuples_bucket_starts(hnh_ctx* ctx, const hnh_tuple* sorted, int64_t n, const hnh_tuple_key* key, int64_t nbuckets,
                             int64_t* starts_host, int stream);
int hnh_tuples_transform(hnh_ctx* ctx, hnh_tuple* tuples, int64_t n, int swap_rc, uint64_t rmod, uint64_t cmod, int stream);
/* hnh_tuples_remap_cols  piecewise relabelling of column indices: with seg = (c / div) * n_sub + (c % div) / sub_div,
 *                        c <- dest_host[seg] + (c % div) % sub_div.  The 1.5D dense-shift schedule maps (block column, chunk)
 *                        to the position of that chunk in its landing buffer with it; a negative dest marks a segment that
 *                        must be empty (HNH_ERR_INVALID otherwise); synchronous */
int hnh_tuples_remap_cols(hnh_ctx* ctx, hnh_tuple* tuples, int64_t n, int64_t div, int64_t sub_div, int64_t n_sub,
                          const int64_t* dest_host, int64_t ndest, int stream);
int hnh_tuples_to_csr(hnh_ctx* ctx, const hnh_tuple* sorted, int64_t n, int64_t rows, int64_t cols, int32_t* rowptr,
                      int32_t* col_idx, double* values, int* max_row_nnz_host, int stream);
/* File input on the device (replaces the duplicate handling of CombBLAS ParallelReadMM(..., maximum<double>()), SpmatLocal.hpp:485-498):
 * hnh_tuples_dedup_max     tuples in HNH_KEY_ROW_COL order: every run of equal (row, col) collapses to ONE tuple carrying the
 *                          run's MAXIMUM value; compacted in place, the new count goes to *n_unique_host; synchronous
 * hnh_tuples_take_strided  out[i] = src[first + i * stride]: a rank's strided slice of the whole tuple list */
int hnh_tuples_dedup_max(hnh_ctx* ctx, hnh_tuple* sorted, int64_t n, int64_t* n_unique_host, int stream);
int hnh_tuples_take_strided(hnh_ctx* ctx, const hnh_tuple* src, int64_t first, int64_t stride, hnh_tuple* out, int64_t n_out, int stream);
/* Synthetic input on the device (replaces CombBLAS GenGraph500Data with initiator {.25,.25,.25,.25}, SpmatLocal.hpp:502-505):
 * hnh_generate_er_keys   the counter-based Erdos-Renyi generator of er_generator.hpp / oracle.py:erdos_renyi_mn, bit for bit:
 *                        draw k -> key = (splitmix64(seed + 2kG) % m) * n + splitmix64(seed + (2k+1)G) % n; `keys` (device,
 *                        capacity `draws`) receives the SORTED, DE-DUPLICATED keys, their count goes to *n_unique_host; synchronous
 * hnh_tuples_from_keys   out[i] = (key / ncols, key % ncols, value) for key = keys[first + i * stride]: a rank's strided slice
 * hnh_tuples_relabel     r = row_label[r], c = col_label[c]: random vertex relabelling for load balance
 *                        (PermEdges / RenameVertices, SpmatLocal.hpp:506-507; random_permute.cpp) */
int hnh_generate_er_keys(hnh_ctx* ctx, uint64_t m, uint64_t n, uint64_t draws, uint64_t seed, uint64_t* keys, int64_t* n_unique_host,
                         int stream);
/* hnh_generate_rmat_keys: the same for a SKEWED initiator (GenGraph500Data's general case, SpmatLocal.hpp:502-505 with initiator
 * {a, b, c, 1 - a - b - c}): the counter-based R-MAT generator of er_generator.hpp: rmat_keys / oracle.py:rmat, bit for bit — edge k
 * descends logm levels, level l picking its quadrant from u = (splitmix64(seed + (k logm + l) G) >> 11) 2^-53; `scramble` multiplies both
 * vertex numbers by an odd constant mod 2^logm (hubs spread over the rows).  `keys` (device, capacity `edges`) receives the sorted,
 * de-duplicated keys row * 2^logm + col; synchronous. */
int hnh_generate_rmat_keys(hnh_ctx* ctx, int logm, uint64_t edges, double a, double b, double c, uint64_t seed, int scramble, uint64_t* keys,
                           int64_t* n_unique_host, int stream);
int hnh_tuples_from_keys(hnh_ctx* ctx, const uint64_t* keys, uint64_t ncols, int64_t first, int64_t stride, double value,
                         hnh_tuple* out, int64_t n_out, int stream);
int hnh_tuples_relabel(hnh_ctx* ctx, hnh_tuple* tuples, int64_t n, const uint64_t* row_label, const uint64_t* col_label, int stream);

/* ---- row-wise dense helpers of the ALS-CG application around fusedSpMM (als_conjugate_gradients.cpp) -----------
 * hnh_rowdot_f64         — batch_dot_product (:9-11):  out[i] = sum_j A[i,j] * B[i,j]
 * hnh_row_scale_add_f64  — scale_matrix_rows + matrix add/sub (:13-29, :117-123, :137):
 *                            Y[i,:] = ya * (yv ? yv[i] : 1) * Y[i,:]  +  xa * (xv ? xv[i] : 1) * X[i,:]
 * hnh_vec_add_scalar_f64 — `v.array() += c` (:99-100);  hnh_vec_div_f64 — cwiseQuotient (:102,136) */
int hnh_rowdot_f64(hnh_ctx* ctx, const double* A, const double* B, double* out, int64_t rows, int R, int stream);
int hnh_row_scale_add_f64(hnh_ctx* ctx, double* Y, const double* yv, double ya, const double* X, const double* xv, double xa,
                          int64_t rows, int R, int stream);
int hnh_vec_add_scalar_f64(hnh_ctx* ctx, double* v, double c, int64_t n, int stream);
/* hnh_cg_step_f64 — the three dense passes in the middle of one CG iteration (:117-127) as one:
 *   X[i,:] += alpha[i] * P[i,:];   Rm[i,:] -= alpha[i] * MP[i,:];   rsnew[i] = <Rm[i,:], Rm[i,:]> */
int hnh_cg_step_f64(hnh_ctx* ctx, double* X, double* Rm, const double* P, const double* MP, const double* alpha, double* rsnew,
                    int64_t rows, int R, int stream);
/* hnh_fill_hashed_f64 — distribution-independent stand-in for Eigen's setRandom() (als_conjugate_gradients.cpp:143-146):
 *   dst[i, j] = scale * uniform(-1, 1) hashed from the GLOBAL element (top_row + i, left_col + j) of an R_global-wide matrix:
 *   key = (top_row + i) * R_global + left_col + j;  h = splitmix64(seed * 0xD1342543DE82EF95 + key * 0x9E3779B97F4A7C15);
 *   value = ((h >> 11) * 2^-52 - 1) * scale      (twin: oracle/oracle.py:hashed_uniform) */
int hnh_fill_hashed_f64(hnh_ctx* ctx, double* dst, int64_t rows, int64_t cols, int64_t top_row, int64_t left_col, int64_t R_global,
                        uint64_t seed, double scale, int stream);
int hnh_vec_div_f64(hnh_ctx* ctx, double* out, const double* num, const double* den, int64_t n, int stream);

/* ---- dense helpers of the GAT application (gat.hpp:83-104) -----------------------------------------------------
 * hnh_gemm_f64            — `buffers[i] * wMats[j]` (gat.hpp:88): C[M x N] = A[M x K] * B[K x N], row-major, on the fp64
 *                           matrix cores (v_mfma_f64_16x16x4_f64): the one true dense contraction of the path
 * hnh_leaky_relu_f64      — `x.max(0) + x.min(0) * alpha` on the SDDMM values (gat.hpp:96-97), in place
 * hnh_relu_store_cols_f64 — `dst.middleCols(col0, cols) = src.array().max(0)` (gat.hpp:103) */
int hnh_gemm_f64(hnh_ctx* ctx, int64_t M, int64_t N, int64_t K, const double* A, const double* B, double* C, int stream);
int hnh_leaky_relu_f64(hnh_ctx* ctx, double* v, double alpha, int64_t n, int stream);
int hnh_relu_store_cols_f64(hnh_ctx* ctx, double* dst, int64_t ld_dst, int64_t col0, const double* src, int64_t rows,
                            int64_t cols, int stream);

/* ---- block descriptors and structure plans ----------------------------------------------------------------
 * A sparse block's STRUCTURE (rowptr / col_idx) is fixed once SpmatLocal has built it (SpmatLocal.hpp:78-188); only its
 * values change.  Everything the row passes derive from the structure alone — the per-row boundaries of the Infinity-Cache
 * panels, the hub-row work list (rows above the long-row threshold, cut into segments) and its exact size — is kept in an
 * opaque plan that the owner of the block creates once and hands to every call, so steady-state calls launch row kernels
 * only.  A plan is filled on first use (per panel count / threshold) and stays valid for as long as the two index arrays keep
 * their contents; blocks whose indices are overwritten in place (a travelling block that ships its indices) pass plan = NULL
 * and get the per-call behaviour of the _ex / _x / _w entry points.
 *   hnh_csr_block   everything a call needs to know about the block: sizes, hints (as the _ex entry points), arrays, plan;
 *   *_p             the _ex / _x / _w entry points on a block descriptor; window == NULL = the whole block (cols >= 0 then
 *                   enables the cache panels), extras == NULL = none.
 * HNH_HUB_SCRATCH_MB (environment, default 2048) bounds the partial-row scratch of the hub-row segments per stream; segments
 * beyond it combine with atomics (exact within the parity tolerance, not bit-reproducible run to run). */
typedef struct hnh_csr_plan hnh_csr_plan;
typedef struct hnh_csr_block {
    int64_t rows, nnz, cols;   /* nnz = rowptr[rows]; cols = rows of the gathered operand, or -1 */
    int32_t max_row_nnz;       /* longest row, or an upper bound, or -1 = unknown */
    int32_t reserved;
    const int32_t* rowptr;     /* device */
    const int32_t* col_idx;    /* device */
    hnh_csr_plan* plan;        /* or NULL */
} hnh_csr_block;
int hnh_csr_plan_create(hnh_ctx* ctx, hnh_csr_plan** out);
int hnh_csr_plan_destroy(hnh_ctx* ctx, hnh_csr_plan* plan);
/* flags of hnh_sddmm_csr_p: HNH_FUSED_VALUES_OVERWRITE = the (window's) values are known to be zero — the reference zeroes them
 * before its SDDMM loop accumulates (distributed_sparse.h:280 -> sparse_kernels.cpp:54) — so values[e] = dot may be STORED instead
 * of read, added to and stored: the caller skips its zero fill, the kernel the read (whole lines written, none fetched). */
int hnh_sddmm_csr_p(hnh_ctx* ctx, const hnh_csr_block* block, double* values, const double* X, const double* Y, int R,
                    unsigned flags, const hnh_csr_window* window, int stream);
/* hnh_sddmm_csr_ps: the SDDMM with the Hadamard product that ends every sddmmA / sddmmB of the reference folded in
 * (`SValues.cwiseProduct(choice->getCSRValues())`, 15D_dense_shift.hpp:366): dst[e] (+)= scale[e] * dot[e], so a caller whose
 * nonzeros are each visited once per operation points dst at its slice of the RESULT vector and scale at the same slice of
 * SValues, and neither the block's own values nor a closing element-wise pass are touched (24 B per nonzero saved; 8 added).
 * scale == NULL: hnh_sddmm_csr_p.  scale must not alias dst. */
int hnh_sddmm_csr_ps(hnh_ctx* ctx, const hnh_csr_block* block, double* dst, const double* scale, const double* X, const double* Y,
                     int R, unsigned flags, const hnh_csr_window* window, int stream);
int hnh_spmm_csr_p(hnh_ctx* ctx, const hnh_csr_block* block, const double* values, const double* X, double* Out, int R,
                   const hnh_csr_window* window, int stream);
/* hnh_spmm_csr_pf: hnh_spmm_csr_p with flags.  HNH_FUSED_OUT_OVERWRITE = the output rows are known to hold nothing yet — a staging
 * buffer that every row of the block writes exactly once — so Out[i,:] = sum is STORED instead of read, added to and stored (the
 * reference's mkl_sparse_d_mm call has beta = 1, sparse_kernels.cpp:95-121, onto a buffer it zeroed; here neither the zero fill nor the
 * read happen).  Rows without nonzeros store zeros.  flags == 0: hnh_spmm_csr_p. */
int hnh_spmm_csr_pf(hnh_ctx* ctx, const hnh_csr_block* block, const double* values, const double* X, double* Out, int R, unsigned flags,
                    const hnh_csr_window* window, int stream);
int hnh_fused_sddmm_spmm_csr_p(hnh_ctx* ctx, const hnh_csr_block* block, double* values, const double* svalues, const double* X,
                               const double* Y, double* Out, int R, unsigned flags, const hnh_fused_extras* extras,
                               const hnh_csr_window* window, int stream);

/* ---- RCCL ring / collectives over xGMI ---------------------------------------------------------------
 * Replace the MPI calls of the shift schedules:
 *   hnh_comm_sendrecv        — MPI_Sendrecv in shiftDenseMatrix (distributed_sparse.h:351-361) and the
 *                              Isend/Irecv set of CSRLocal::shiftCSR (SpmatLocal.hpp:200-259), issued as one
 *                              ncclGroup of explicit-peer send+recv (no MPI_ANY_SOURCE);
 *   hnh_comm_allgather       — MPI_Allgather (15D_dense_shift.hpp:194-195,310-311; 25D_cannon_dense.hpp:265);
 *   hnh_comm_reduce_scatter  — MPI_Reduce_scatter, equal counts (15D_dense_shift.hpp:240-242,378-380).
 * A communicator is created from a 128-byte unique id made on rank 0 (hnh_comm_unique_id) and handed to
 * the other ranks by the launcher (torch.distributed store, MPI, a file ...). */
#define HNH_UNIQUE_ID_BYTES 128
int hnh_comm_unique_id(void* id_host /* HNH_UNIQUE_ID_BYTES, host */);
int hnh_comm_init(hnh_ctx* ctx, int nranks, int rank, const void* id_host, void** comm);
int hnh_comm_split(hnh_ctx* ctx, void* comm, int color, int key, void** newcomm);
int hnh_comm_destroy(hnh_ctx* ctx, void* comm);
/* What the COMMUNICATOR says about itself (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): its size, this rank's index in it and
 * the device it is bound to. */
int hnh_comm_identity(hnh_ctx* ctx, void* comm, int* nranks, int* rank, int* device);
int hnh_comm_sendrecv(hnh_ctx* ctx, void* comm, const void* sendbuf, size_t sendbytes, int dst, void* recvbuf,
                      size_t recvbytes, int src, int stream);
/* Everything enqueued between begin and end is issued as ONE RCCL group (ncclGroupStart/End), so that
 * send/recv pairs towards several peers progress concurrently over their separate xGMI links. */
int hnh_comm_group_begin(hnh_ctx* ctx);
int hnh_comm_group_end(hnh_ctx* ctx);
int hnh_comm_allgather(hnh_ctx* ctx, void* comm, const void* sendbuf, void* recvbuf, size_t bytes_per_rank,
                       int stream);
int hnh_comm_reduce_scatter_f64(hnh_ctx* ctx, void* comm, const double* sendbuf, double* recvbuf,
                                size_t count_per_rank, int stream);
int hnh_comm_allreduce_f64(hnh_ctx* ctx, void* comm, const double* sendbuf, double* recvbuf, size_t count,
                           int stream);

/* ---- peer-to-peer PULL over mapped peer memory (single node; no RCCL) -------------------------------------
 * The second device-to-device transport behind the same schedules (host side: IpcWorld, world.hpp).  Replaces the same
 * MPI calls as the RCCL section — MPI_Sendrecv of shiftDenseMatrix (distributed_sparse.h:351-361) and the Isend/Irecv
 * set of CSRLocal::shiftCSR (SpmatLocal.hpp:200-259) — with the RECEIVER copying straight out of the sender's buffer:
 *   hnh_ipc_export   the allocation that holds `ptr`, as a 64-byte handle another PROCESS of this node can open
 *                    (hipIpcGetMemHandle of the allocation's base), plus ptr's offset in it;
 *   hnh_ipc_open     maps a peer's allocation into this process (hipIpcOpenMemHandle; over xGMI when the peer owns
 *                    another GPU, plainly when both processes share one); returns the base of the mapping;
 *   hnh_ipc_pull     n copies peer -> local issued together from `stream`: mode HNH_IPC_PULL_ENGINE = one
 *                    hipMemcpyAsync per source on auxiliary streams forked from and joined to `stream` (the copy
 *                    engines: no compute units, every source's link busy at once), HNH_IPC_PULL_KERNEL = ONE
 *                    gather-copy launch on `stream` (`wgs_per_copy` workgroups per source, loads over the links);
 *   hnh_ipc_flags_*  a host shared-memory region (the processes' common mmap) made visible to the device; its
 *                    64-bit words order the processes' STREAMS without a host round trip:
 *   hnh_stream_write_flag / hnh_stream_wait_flag   `stream` stores `value` into a word after everything enqueued so
 *                    far / holds `stream` until the word is >= `value` (stream memory operations of the command
 *                    processor, or one-lane kernels with system-scope atomics: HNH_IPC_FLAGS=memop|kernel).
 * Values only grow, so a wait never depends on WHEN the peer enqueued its write. */
#define HNH_IPC_HANDLE_BYTES 64
#define HNH_IPC_PULL_ENGINE 0
#define HNH_IPC_PULL_KERNEL 1
#define HNH_IPC_MAX_PULL 16
int hnh_ipc_export(hnh_ctx* ctx, const void* ptr, void* handle_host /* HNH_IPC_HANDLE_BYTES */, uint64_t* offset,
                   uint64_t* alloc_bytes);
int hnh_ipc_open(hnh_ctx* ctx, const void* handle_host, uint64_t alloc_bytes, void** base);
int hnh_ipc_close(hnh_ctx* ctx, void* base);
int hnh_ipc_pull(hnh_ctx* ctx, int stream, int n, void* const* dst, const void* const* src, const size_t* bytes,
                 int mode, int wgs_per_copy);
int hnh_ipc_flags_register(hnh_ctx* ctx, void* host_shm, size_t bytes, void** device_view);
int hnh_ipc_flags_unregister(hnh_ctx* ctx, void* host_shm);
int hnh_stream_write_flag(hnh_ctx* ctx, int stream, void* flag_device, uint64_t value);
int hnh_stream_wait_flag(hnh_ctx* ctx, int stream, void* flag_device, uint64_t value);

#ifdef __cplusplus
}
#endif
#endif /* HNH_KERNELS_H */
