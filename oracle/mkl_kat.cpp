// TEST INFRASTRUCTURE ONLY. Known-answer test pinning the hand-declared MKL ABI in stubs/mkl_spblas.h:
// a 3x4 COO matrix is converted with SPARSE_OPERATION_TRANSPOSE (as CSRLocal does for transposed
// blocks, SpmatLocal.hpp:108-118), exported, and multiplied with beta = 1 (sparse_kernels.cpp:95-107).
#include <cstdio>
#include <cmath>
#include "mkl_spblas.h"
int main() {
    // S = [[1,2,0,0],[0,0,3,0],[0,0,0,4]]  (3x4), given unsorted
    MKL_INT r[] = {2, 0, 1, 0}, c[] = {3, 1, 2, 0};
    double v[] = {4, 2, 3, 1};
    sparse_matrix_t coo, csr;
    if (mkl_sparse_d_create_coo(&coo, SPARSE_INDEX_BASE_ZERO, 3, 4, 4, r, c, v)) return 1;
    if (mkl_sparse_convert_csr(coo, SPARSE_OPERATION_TRANSPOSE, &csr)) return 2;
    sparse_index_base_t ib; MKL_INT R, C, *rs, *re, *ci; double* vals;
    if (mkl_sparse_d_export_csr(csr, &ib, &R, &C, &rs, &re, &ci, &vals)) return 3;
    if (R != 4 || C != 3) return 4;
    // S^T rows: 0:{(0,1)} 1:{(0,2)} 2:{(1,3)} 3:{(2,4)}
    const MKL_INT exp_ci[] = {0, 0, 1, 2}; const double exp_v[] = {1, 2, 3, 4};
    for (int i = 0; i < 4; i++) if (rs[i] != i || ci[i] != exp_ci[i] || vals[i] != exp_v[i]) return 5;
    double B[3 * 2] = {1, 2, 3, 4, 5, 6}, Cm[4 * 2] = {10, 10, 10, 10, 10, 10, 10, 10};
    matrix_descr d; d.type = SPARSE_MATRIX_TYPE_GENERAL; d.mode = SPARSE_FILL_MODE_FULL; d.diag = SPARSE_DIAG_NON_UNIT;
    if (mkl_sparse_d_mm(SPARSE_OPERATION_NON_TRANSPOSE, 1.0, csr, d, SPARSE_LAYOUT_ROW_MAJOR, B, 2, 2, 1.0, Cm, 2)) return 6;
    const double exp_C[] = {11, 12, 12, 14, 19, 22, 30, 34};  // 10 + S^T * B
    for (int i = 0; i < 8; i++) if (std::fabs(Cm[i] - exp_C[i]) > 0) return 7;
    std::printf("mkl_kat: ok\n");
    return 0;
}
