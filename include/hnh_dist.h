/*
 * hnh_dist.h — C ABI of the host layer (libhnh_host.so): the HnH operator surface as opaque handles.
 *
 * The host layer itself is C++ and mirrors the reference's classes by name (Distributed_Sparse,
 * Sparse15D_Dense_Shift, Sparse15D_Sparse_Shift, Sparse25D_Cannon_Dense, Sparse25D_Cannon_Sparse, SpmatLocal,
 * CSRLocal, StandardKernel, FlexibleGrid, BufferPair — headers under distributed_sddmm_amd/csrc/host/, see
 * INTEGRATION.md); C++ callers include those headers directly.  This C ABI exposes the same operations
 * to non-C++ callers (the Python tests and bench.py bind it with ctypes).  Each function cites the
 * reference member it forwards to (file:line under /root/reference).
 *
 * Conventions: int status return (0 == HNH_OK, codes of hnh_kernels.h); no exception crosses the ABI;
 * hnh_host_last_error() returns the calling thread's last message.  Handles are not thread-safe; every
 * call must be made by the thread that drives the handle's rank.  All calls of one operation are
 * collective over the ranks of the world, like the reference's MPI-based methods.
 */
#ifndef HNH_DIST_H
#define HNH_DIST_H

#include <stddef.h>
#include <stdint.h>

#include "hnh_kernels.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hnh_world hnh_world;               /* one rank of a process group (MPI_COMM_WORLD analogue) */
typedef struct hnh_thread_group hnh_thread_group; /* shared state of an in-process ("loopback") group       */
typedef struct hnh_spmat hnh_spmat;               /* SpmatLocal                                             */
typedef struct hnh_dist hnh_dist;                 /* Distributed_Sparse subclass + its StandardKernel       */
typedef struct hnh_dense hnh_dense;               /* DenseMatrix (device resident)                          */
typedef struct hnh_vec hnh_vec;                   /* VectorXd   (device resident)                          */

/* KernelMode (sparse_kernels.h:13) and MatMode (common.h:21) */
#define HNH_K_SDDMM_A 0
#define HNH_K_SPMM_A 1
#define HNH_K_SPMM_B 2
#define HNH_K_SDDMM_B 3
#define HNH_AMAT 0
#define HNH_BMAT 1

/* transport callbacks for hnh_world_create_callback (blocking; pointers are in the backend's memory space) */
#ifndef HNH_COMM_CALLBACKS_DEFINED
#define HNH_COMM_CALLBACKS_DEFINED
typedef struct hnh_comm_callbacks {
    void* user;
    int (*sendrecv)(void* user, const void* sendbuf, size_t sendbytes, int dst, void* recvbuf, size_t recvbytes, int src);
    int (*barrier)(void* user);
    int (*allgather)(void* user, const void* send, void* recv, size_t bytes_per_rank);
} hnh_comm_callbacks;
#endif

const char* hnh_host_last_error(void);

/* Selects the library implementing hnh_kernels.h.  NULL / "" = the product HIP library next to
 * libhnh_host.so.  The product never loads anything else; tests pass the path of the oracle test double. */
int hnh_backend_load(const char* path);
const char* hnh_host_backend_name(void);

/* ---- process groups (replace MPI_Init / MPI_COMM_WORLD; benchmark_dist.cpp:35-36) */
int hnh_world_create_single(int device, hnh_world** out);
int hnh_thread_group_create(int nranks, hnh_thread_group** out);
int hnh_thread_group_destroy(hnh_thread_group* g);
int hnh_world_create_thread(hnh_thread_group* g, int rank, int device, hnh_world** out);
int hnh_rccl_unique_id(void* id128_host);
int hnh_world_create_rccl(int rank, int nranks, int device, const void* id128_host, hnh_world** out);
/* One process per GPU of ONE node without RCCL: receivers pull out of their peers' mapped buffers (hnh_kernels.h, "ipc").
 * `session` = the same string on every rank of the job (it names the node-local shared-memory rendezvous). */
int hnh_world_create_ipc(int rank, int nranks, int device, const char* session, hnh_world** out);
int hnh_world_create_callback(int rank, int nranks, int device, const hnh_comm_callbacks* cb, hnh_world** out);
int hnh_world_destroy(hnh_world* w);
int hnh_world_rank(hnh_world* w);
int hnh_world_size(hnh_world* w);
int hnh_world_barrier(hnh_world* w);
int hnh_world_sync(hnh_world* w);                      /* drain compute + communication streams */
/* World::set_solo (an addition, a MEASUREMENT entry point; loopback transport only): solo replay — while on, the rank runs its own
 * side of every collective call alone (each receive = a device copy of what it would send, host collectives and barriers local), so
 * one rank's kernel sequence + the HBM side of its exchange can be timed with the GPU to itself.  Results of such calls are void. */
int hnh_world_set_solo(hnh_world* w, int on);
int hnh_world_set_timing_sync(hnh_world* w, int on);   /* perf counters synchronise first (reference-like attribution) */
void* hnh_world_stream(hnh_world* w, int stream);      /* raw hipStream_t */
hnh_ctx* hnh_world_ctx(hnh_world* w);                  /* the rank's kernel-level context */
/* FlexibleGrid(nr, nc, nh, adjacency) (FlexibleGrid.hpp:41-94): out9 = i, j, k, rankInRow, rankInCol, rankInFiber,
 * row size, col size, fiber size; *ok = result of the broadcast self test (FlexibleGrid.hpp:169-201). */
/* Transport self-test (collective): runs ONE communication primitive on `count` doubles with known contents and reports the
 * largest deviation from the expected values.  bench.py --gpus N runs every one of them under a watchdog before the timed
 * region, so a transport problem shows up as "rank r, primitive X" instead of a hang (no reference counterpart). */
enum {
    HNH_PREFLIGHT_RING = 0,                 /* relay step: send to ring rank +1, receive from -1 */
    HNH_PREFLIGHT_MESH = 1,                 /* n - 1 explicit-peer pairs in one group (mesh fetch) */
    HNH_PREFLIGHT_ALLGATHER = 2,            /* over a layer sub-communicator */
    HNH_PREFLIGHT_REDUCE_SCATTER = 3,
    HNH_PREFLIGHT_ALLREDUCE = 4,
    HNH_PREFLIGHT_VARIABLE = 5,             /* allgatherv + reduce_scatter_v, ragged counts */
    HNH_PREFLIGHT_ALLTOALLV = 6,            /* device all-to-all of the set-up pipeline */
    HNH_PREFLIGHT_ALLGATHER_WORLD = 7,      /* native RCCL collective on the world communicator */
    HNH_PREFLIGHT_REDUCE_SCATTER_WORLD = 8,
    HNH_PREFLIGHT_COUNT = 9
};
int hnh_world_preflight(hnh_world* w, int what, int64_t count, double* max_err);
/* running hash / number of the communicator splits so far: identical on every rank iff they split in the same order */
int hnh_world_split_signature(hnh_world* w, uint64_t* signature, int* count);
int hnh_world_grid_probe(hnh_world* w, int nr, int nc, int nh, int adjacency, int* out9, int* ok);
/* Where every rank of the world runs (collective; `out` takes hnh_world_size() records, the same on every rank): process id, device
 * ordinal and PCI bus id of the rank's GPU and — RCCL transport — what its communicator reports about itself (ncclCommCount,
 * ncclCommUserRank, ncclCommCuDevice; -1 for the other transports).  A job whose ranks report fewer distinct bus ids than ranks shares
 * GPUs: bench.py prints these records so that its N > 1 line certifies where it ran (the reference's ranks are host processes). */
typedef struct hnh_rank_identity {
    int32_t rank, pid, device_ordinal;
    int32_t comm_count, comm_rank, comm_device;
    char pci_bus_id[40];
} hnh_rank_identity;
int hnh_world_identities(hnh_world* w, hnh_rank_identity* out);

/* ---- sparse input (SpmatLocal.hpp:267-606) */
/* Wraps tuples that this rank holds initially (any distribution; coords / M / N / dist_nnz as the reference's
 * drivers fill them).  Host arrays, copied. */
int hnh_spmat_create(hnh_world* w, int64_t M, int64_t N, int64_t dist_nnz, int64_t local_nnz, const int64_t* rows,
                     const int64_t* cols, const double* values, hnh_spmat** out);
/* SpmatLocal::loadTuples(readFromFile, logM, nnz_per_row, filename) (SpmatLocal.hpp:467-533) */
int hnh_spmat_load_tuples(hnh_world* w, int read_from_file, int logM, int nnz_per_row, const char* filename, hnh_spmat** out);
int hnh_spmat_info(hnh_spmat* s, int64_t out4[4]); /* M, N, dist_nnz, local tuple count */
/* seeded random relabelling of rows/columns for load balance (random_permute.cpp; twin: oracle.vertex_permutation) */
int hnh_spmat_permute(hnh_spmat* s, uint64_t seed);
int hnh_spmat_destroy(hnh_spmat* s);
/* the shared synthetic generator (bit-identical to oracle/oracle.py:erdos_renyi_mn) */
int hnh_er_generate(uint64_t m, uint64_t n, uint64_t draws, uint64_t seed, void** handle, int64_t* count);
/* skewed R-MAT stand-in for real graphs (initiator a, b, c, 1-a-b-c; twin: oracle/oracle.py:rmat) */
int hnh_rmat_generate(int logm, uint64_t edges, double a, double b, double c, uint64_t seed, int scramble, void** handle,
                      int64_t* count);
int hnh_er_fetch(void* handle, int64_t* rows, int64_t* cols); /* also frees the handle */
/* (an addition) writes the entries as a MatrixMarket coordinate file (1-based; values NULL = all 1; symmetric != 0: the header says so
 * and the caller passes one triangle), formatted by all host cores — the files the input side (loadTuples(readFromFile = true),
 * SpmatLocal.hpp:485-498) is tested and benchmarked with. */
int hnh_write_matrix_market(const char* path, int64_t M, int64_t N, int64_t n, const int64_t* rows, const int64_t* cols, const double* values,
                            int symmetric);

/* ---- operator construction (benchmark_dist.cpp:45-82): alg in
 *   "15d_fusion1" | "15d_fusion2" | "15d_sparse" | "25d_dense_replicate" | "25d_sparse_replicate" */
int hnh_dist_create(hnh_world* w, const char* alg, hnh_spmat* s, int R, int c, hnh_dist** out);
int hnh_dist_destroy(hnh_dist* d);
/* out16 = M, N, R, p, c, localArows, localAcols, localBrows, localBcols, len(like_S_values), len(like_ST_values),
 *         r_split, dist_nnz, proc_rank, #aSubmatrices, #bSubmatrices   (distributed_sparse.h:34-76) */
int hnh_dist_info(hnh_dist* d, int64_t out16[16]);
/* a/bSubmatrices (distributed_sparse.h:56-57): 4 ints each (topRow, leftCol, rowCount, colCount) */
int hnh_dist_submatrices(hnh_dist* d, int matmode, int64_t* out, int capacity_entries);
int hnh_dist_set_r(hnh_dist* d, int R); /* setRValue */
/* which: 0 = json_algorithm_info (distributed_sparse.h:131-179), 1 = json_perf_statistics (:245-261) */
int hnh_dist_json(hnh_dist* d, int which, char* buf, size_t capacity);
int hnh_dist_reset_timers(hnh_dist* d); /* reset_performance_timers */
/* HIP-event time of the local kernels launched by the operator's StandardKernel since enabling */
int hnh_dist_kernel_profile(hnh_dist* d, int enable, double* total_ms, int64_t* launches);
/* Borrowed value arrays of stationary blocks (an addition; HNH_BORROW=off|force, default = where it pays): block-level counts since
 * construction, S and ST together: [0] SpMM value arrays read in place from the caller's vector, [1] copied as setCSRValues does
 * (SpmatLocal.hpp:571-579), [2] SDDMM results written as SValues .* dots by the kernel, [3] by the closing Hadamard pass
 * (15D_dense_shift.hpp:366) */
int hnh_dist_borrow_stats(hnh_dist* d, int64_t out4[4]);

/* ---- dense operands / value vectors (device resident) */
int hnh_dense_create(hnh_world* w, int64_t rows, int64_t cols, double fill, hnh_dense** out);
int hnh_dense_wrap(hnh_world* w, void* device_ptr, int64_t rows, int64_t cols, hnh_dense** out); /* non-owning */
int hnh_dense_like(hnh_dist* d, int matmode, double fill, hnh_dense** out); /* like_A_matrix / like_B_matrix (:197-203) */
int hnh_dense_shape(hnh_dense* m, int64_t out2[2]);
void* hnh_dense_data(hnh_dense* m); /* device pointer; may change after an operation that hands storage back */
int hnh_dense_upload(hnh_dense* m, const double* host);
int hnh_dense_download(hnh_dense* m, double* host);
int hnh_dense_fill(hnh_dense* m, double value);
int hnh_dense_copy(hnh_dense* dst, hnh_dense* src);
int hnh_dense_destroy(hnh_dense* m);
int hnh_dense_dummy_initialize(hnh_dist* d, hnh_dense* m, int matmode); /* dummyInitialize (:322-346) */
int hnh_vec_create(hnh_world* w, int64_t n, double fill, hnh_vec** out);
int hnh_vec_like(hnh_dist* d, int which /* 0 = like_S_values, 1 = like_ST_values (:189-195) */, double fill, hnh_vec** out);
int64_t hnh_vec_size(hnh_vec* v);
void* hnh_vec_data(hnh_vec* v);
int hnh_vec_upload(hnh_vec* v, const double* host);
int hnh_vec_download(hnh_vec* v, double* host);
int hnh_vec_fill(hnh_vec* v, double value);
int hnh_vec_destroy(hnh_vec* v);

/* ---- operations (distributed_sparse.h:268-320) */
int hnh_dist_initial_shift(hnh_dist* d, hnh_dense* A, hnh_dense* B, int kernel_mode);
int hnh_dist_de_shift(hnh_dist* d, hnh_dense* A, hnh_dense* B, int kernel_mode);
int hnh_dist_sddmmA(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S, hnh_vec* result);
int hnh_dist_sddmmB(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S, hnh_vec* result);
int hnh_dist_spmmA(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S);
int hnh_dist_spmmB(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S);
int hnh_dist_fusedSpMM(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S, hnh_vec* sddmm_buffer, int matmode);
/* Distributed_Sparse::hold_moving_operand / release_moving_operand (an addition): a promise that the CONTENTS of `m` stay
 * the same until released (m = NULL), so a schedule may keep the blocks of it that it fetched from other ranks (the fixed
 * factor of an ALS half-step, als_conjugate_gradients.cpp:38-141).  Ignored by schedules that cannot use it. */
int hnh_dist_hold_moving_operand(hnh_dist* d, hnh_dense* m_or_null);
/* Distributed_Sparse::walk_windows_when_held (an addition, a MEASUREMENT entry point): with a held operand's blocks resident a call
 * runs one pass over them; on = 1 makes it walk the chunk windows as a fetching call does (own block, then windowed passes over what
 * has landed — everything, here), on = 2 one windowed pass per chunk (the sequence of a call whose chunks arrive one by one), so that
 * one rank's kernel sequence of a p-rank job can be timed alone on a GPU.  Same results either way. */
int hnh_dist_walk_windows_when_held(hnh_dist* d, int on);
/* Distributed_Sparse::fusedSpMM_out (an addition): out-of-place fusedSpMM with the applications' surrounding work in the
 * same pass — LeakyReLU between the halves (gat.hpp:96-99), Out += x_scale * X and rowdot[i] = <X[i,:], Out[i,:]>
 * (als_conjugate_gradients.cpp:93,282,295).  *supported = 0 and nothing done when the schedule has no single fused pass. */
int hnh_dist_fusedSpMM_out(hnh_dist* d, hnh_dense* A, hnh_dense* B, int matmode, hnh_dense* Out, int leaky, double leaky_alpha,
                           double x_scale, hnh_vec* rowdot_or_null, int* supported);
int hnh_dist_algorithm(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S, hnh_vec* result_or_null, int kernel_mode,
                       int initial_replicate);

/* ---- ALS by batched conjugate gradients around fusedSpMM (als_conjugate_gradients.{h,cpp}; BASELINE config 5) */
typedef struct hnh_als hnh_als; /* Distributed_ALS */
/* Distributed_ALS(d_ops, artificial_groundtruth) (.cpp:148-184); random fills are hashes of global coordinates */
int hnh_als_create(hnh_dist* d, int artificial_groundtruth, uint64_t seed, hnh_als** out);
int hnh_als_destroy(hnh_als* a);
int hnh_als_set_ground_truth(hnh_als* a, hnh_vec* gt_S_order, hnh_vec* gt_ST_order); /* members ground_truth{,_transpose} */
int hnh_als_initialize_embeddings(hnh_als* a);                                         /* initializeEmbeddings (.cpp:221-233) */
int hnh_als_set_embeddings(hnh_als* a, hnh_dense* A, hnh_dense* B);                   /* members A, B (copied in)   */
int hnh_als_get_embeddings(hnh_als* a, hnh_dense* A, hnh_dense* B);                   /* members A, B (copied out)  */
int hnh_als_cg_optimizer(hnh_als* a, int matmode, int cg_max_iter);                    /* cg_optimizer (.cpp:38-141) */
int hnh_als_run_cg(hnh_als* a, int n_alternating_steps);                               /* run_cg (.cpp:235-263)      */
int hnh_als_compute_residual(hnh_als* a, double* out);                                 /* computeResidual (.cpp:201-219) */

/* ---- GAT forward pass (gat.hpp; BASELINE config 5's second application) */
typedef struct hnh_gat hnh_gat; /* GAT */
/* GAT(layers, d_ops) (gat.hpp:57-81): spec3 = {input_features, features_per_head, num_heads} per layer */
int hnh_gat_create(hnh_dist* d, int nlayers, const int* spec3, double leaky_relu_alpha, hnh_gat** out);
int hnh_gat_destroy(hnh_gat* g);
int hnh_gat_weight_shape(hnh_gat* g, int layer, int head, int64_t out2[2]);       /* layers[l].wMats[h] */
int hnh_gat_set_weight(hnh_gat* g, int layer, int head, const double* host);      /* row-major, host */
int hnh_gat_set_input(hnh_gat* g, hnh_dense* X);                                  /* buffers[0] (copied in)  */
int hnh_gat_get_output(hnh_gat* g, hnh_dense* out);                               /* buffers.back() (copied) */
int hnh_gat_buffer_shape(hnh_gat* g, int index, int64_t out2[2]);                 /* buffers[index]          */
int hnh_gat_forward(hnh_gat* g);                                                  /* forwardPass (gat.hpp:106-112) */

#ifdef __cplusplus
}
#endif
#endif /* HNH_DIST_H */
