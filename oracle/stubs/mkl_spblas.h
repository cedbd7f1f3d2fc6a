// TEST INFRASTRUCTURE ONLY (oracle build). Hand-written declarations of the six Intel MKL
// inspector-executor sparse BLAS entry points the reference calls (SpmatLocal.hpp:117-137,171-179,
// 187,192; sparse_kernels.cpp:95,109).  MKL's shared libraries (2021.4, ILP64, gnu_thread) are in
// /opt/conda/lib but its headers are not; the enum values below are MKL's ABI values, confirmed by a
// 3x4 transpose-convert + d_mm known-answer test (oracle/tests in Makefile target `mkl_kat`).
#pragma once
#include <cstdint>

#ifdef MKL_ILP64
typedef int64_t MKL_INT;
#else
typedef int MKL_INT;
#endif

extern "C" {
struct sparse_matrix;
typedef struct sparse_matrix* sparse_matrix_t;
typedef enum { SPARSE_STATUS_SUCCESS = 0 } sparse_status_t;
typedef enum { SPARSE_INDEX_BASE_ZERO = 0, SPARSE_INDEX_BASE_ONE = 1 } sparse_index_base_t;
typedef enum { SPARSE_OPERATION_NON_TRANSPOSE = 10, SPARSE_OPERATION_TRANSPOSE = 11,
               SPARSE_OPERATION_CONJUGATE_TRANSPOSE = 12 } sparse_operation_t;
typedef enum { SPARSE_MATRIX_TYPE_GENERAL = 20 } sparse_matrix_type_t;
typedef enum { SPARSE_FILL_MODE_LOWER = 40, SPARSE_FILL_MODE_UPPER = 41, SPARSE_FILL_MODE_FULL = 42 } sparse_fill_mode_t;
typedef enum { SPARSE_DIAG_NON_UNIT = 50, SPARSE_DIAG_UNIT = 51 } sparse_diag_type_t;
typedef enum { SPARSE_LAYOUT_ROW_MAJOR = 101, SPARSE_LAYOUT_COLUMN_MAJOR = 102 } sparse_layout_t;
struct matrix_descr { sparse_matrix_type_t type; sparse_fill_mode_t mode; sparse_diag_type_t diag; };

sparse_status_t mkl_sparse_d_create_coo(sparse_matrix_t* A, sparse_index_base_t indexing, MKL_INT rows,
                                        MKL_INT cols, MKL_INT nnz, MKL_INT* row_indx, MKL_INT* col_indx,
                                        double* values);
sparse_status_t mkl_sparse_convert_csr(sparse_matrix_t source, sparse_operation_t operation,
                                       sparse_matrix_t* dest);
sparse_status_t mkl_sparse_d_export_csr(sparse_matrix_t source, sparse_index_base_t* indexing, MKL_INT* rows,
                                        MKL_INT* cols, MKL_INT** rows_start, MKL_INT** rows_end,
                                        MKL_INT** col_indx, double** values);
sparse_status_t mkl_sparse_d_create_csr(sparse_matrix_t* A, sparse_index_base_t indexing, MKL_INT rows,
                                        MKL_INT cols, MKL_INT* rows_start, MKL_INT* rows_end,
                                        MKL_INT* col_indx, double* values);
sparse_status_t mkl_sparse_d_mm(sparse_operation_t operation, double alpha, sparse_matrix_t A,
                                struct matrix_descr descr, sparse_layout_t layout, const double* B,
                                MKL_INT columns, MKL_INT ldb, double beta, double* C, MKL_INT ldc);
sparse_status_t mkl_sparse_destroy(sparse_matrix_t A);
}
