/* The operator C ABI (include/hnh_dist.h) from plain C11 — the shape of binding a cgo / Rust-FFI / JNI caller would write.
 *
 *     c_operator <logM> <edgeFactor> <algorithm> <R>
 *
 * What a user of the reference does in C++ (README.md "How do I use it?": load a matrix, pick an algorithm, get the buffers adapted to
 * it, run the operation), spelled with handles and status codes:
 *     SpmatLocal S; S.loadTuples(false, logM, edgeFactor, "")          -> hnh_spmat_load_tuples
 *     new Sparse15D_Dense_Shift(&S, R, c, 2, &local_ops)               -> hnh_dist_create("15d_fusion2", ...)
 *     d_ops->like_A_matrix / like_B_matrix / like_S_values             -> hnh_dense_like / hnh_vec_like
 *     d_ops->fusedSpMM(A, B, S, result, Amat)                          -> hnh_dist_fusedSpMM
 * and then checks the answer itself: A and B are filled with a_i u_k and b_j v_k (hashes of the GLOBAL indices, placed with
 * hnh_dist_submatrices, so the check holds for every algorithm and is independent of the library's layout), for which one fused
 * call leaves A[i,k] = (u.v) a_i v_k sum_{j in row i} b_j^2 — summed here over the nonzeros hnh_er_generate returns.
 * One process, one GPU (device HNH_DEVICE, default 0).  Exit status 0 = the result matches to 1e-11.                                    */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "hnh_dist.h"

#define TRY(call)                                                                       \
    do {                                                                                \
        int st_ = (call);                                                               \
        if (st_ != HNH_OK) {                                                            \
            fprintf(stderr, "%s -> status %d: %s\n", #call, st_, hnh_host_last_error()); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static double keyed(uint64_t idx, uint64_t salt) { /* deterministic value in [0.5, 1.5) per index */
    const uint64_t h = (idx * 2654435761ULL + salt * 0x9E3779B1ULL) & 0xFFFFFFFFULL;
    return 0.5 + (double)h / 4294967296.0;
}

/* fills a local operand of `d` (matmode 0 = A, 1 = B) with row_key(global row) * col_key(global column), block by block */
static int fill_keyed(hnh_dist* d, hnh_dense* mat, int matmode, uint64_t row_salt, uint64_t col_salt, int64_t m) {
    int64_t shape[2], sub[4 * 64], info[16];
    TRY(hnh_dense_shape(mat, shape));
    TRY(hnh_dist_info(d, info));
    const int nsub = (int)info[14 + matmode]; /* #aSubmatrices, #bSubmatrices */
    TRY(hnh_dist_submatrices(d, matmode, sub, 64));
    double* host = calloc((size_t)(shape[0] * shape[1]), sizeof(double));
    if (!host) return 1;
    int64_t off = 0;
    for (int s = 0; s < nsub; s++) {
        const int64_t top = sub[4 * s], left = sub[4 * s + 1], rc = sub[4 * s + 2], cc = sub[4 * s + 3];
        for (int64_t i = 0; i < rc && top + i < m; i++)
            for (int64_t k = 0; k < cc; k++) host[off + i * cc + k] = keyed((uint64_t)(top + i), row_salt) * keyed((uint64_t)(left + k), col_salt);
        off += rc * cc;
    }
    const int st = hnh_dense_upload(mat, host);
    free(host);
    return st;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: c_operator logM edgeFactor algorithm R\n");
        return 2;
    }
    const int logm = atoi(argv[1]), ef = atoi(argv[2]), r = atoi(argv[4]);
    const char* alg = argv[3];
    const int64_t m = (int64_t)1 << logm;
    const char* dev = getenv("HNH_DEVICE");

    TRY(hnh_backend_load(NULL)); /* the kernel library next to libhnh_host.so; fails loudly without it or without a GPU */
    hnh_world* w = NULL;
    hnh_spmat* s = NULL;
    hnh_dist* d = NULL;
    hnh_dense *A = NULL, *B = NULL;
    hnh_vec *S = NULL, *buf = NULL;
    TRY(hnh_world_create_single(dev ? atoi(dev) : 0, &w));
    TRY(hnh_spmat_load_tuples(w, 0, logm, ef, "", &s));
    TRY(hnh_dist_create(w, alg, s, r, 1, &d));
    TRY(hnh_dense_like(d, 0, 0.0, &A));
    TRY(hnh_dense_like(d, 1, 0.0, &B));
    TRY(hnh_vec_like(d, 0, 1.0, &S));
    TRY(hnh_vec_like(d, 0, 0.0, &buf));
    if (fill_keyed(d, A, 0, 1, 3, m) != HNH_OK || fill_keyed(d, B, 1, 2, 4, m) != HNH_OK) {
        fprintf(stderr, "filling the operands failed: %s\n", hnh_host_last_error());
        return 1;
    }
    TRY(hnh_dist_initial_shift(d, A, B, 0 /* k_sddmmA */));
    TRY(hnh_dist_fusedSpMM(d, A, B, S, buf, 0 /* Amat */));
    TRY(hnh_dist_de_shift(d, A, B, 0));
    TRY(hnh_world_sync(w));

    /* the closed form, from the generator's own nonzeros (the same draws hnh_spmat_load_tuples made, seed 12345) */
    void* gen = NULL;
    int64_t nnz = 0;
    TRY(hnh_er_generate((uint64_t)m, (uint64_t)m, (uint64_t)(m * ef), 12345, &gen, &nnz));
    int64_t* rows = malloc((size_t)nnz * sizeof(int64_t));
    int64_t* cols = malloc((size_t)nnz * sizeof(int64_t));
    double* rowsum = calloc((size_t)m, sizeof(double));
    if (!rows || !cols || !rowsum) return 1;
    TRY(hnh_er_fetch(gen, rows, cols));
    for (int64_t e = 0; e < nnz; e++) {
        const double b = keyed((uint64_t)cols[e], 2);
        rowsum[rows[e]] += b * b;
    }
    double uv = 0.0;
    for (int k = 0; k < r; k++) uv += keyed((uint64_t)k, 3) * keyed((uint64_t)k, 4);

    int64_t shape[2], sub[4 * 64], dinfo[16];
    TRY(hnh_dense_shape(A, shape));
    double* got = malloc((size_t)(shape[0] * shape[1]) * sizeof(double));
    if (!got) return 1;
    TRY(hnh_dense_download(A, got));
    TRY(hnh_dist_info(d, dinfo));
    const int nsub = (int)dinfo[14];
    TRY(hnh_dist_submatrices(d, 0, sub, 64));
    double worst = 0.0, top_val = 0.0;
    int64_t off = 0, checked = 0;
    for (int sidx = 0; sidx < nsub; sidx++) {
        const int64_t top = sub[4 * sidx], left = sub[4 * sidx + 1], rc = sub[4 * sidx + 2], cc = sub[4 * sidx + 3];
        for (int64_t i = 0; i < rc && top + i < m; i++)
            for (int64_t k = 0; k < cc; k++) {
                const double want = uv * keyed((uint64_t)(top + i), 1) * rowsum[top + i] * keyed((uint64_t)(left + k), 4);
                const double err = fabs(got[off + i * cc + k] - want);
                if (err > worst) worst = err;
                if (want > top_val) top_val = want;
                checked++;
            }
        off += rc * cc;
    }
    int64_t info[4];
    TRY(hnh_spmat_info(s, info));
    printf("%s on %s: %lld x %lld, %lld nonzeros, R = %d: fusedSpMM checked on %lld elements, max deviation %.3e of the largest entry\n", alg,
           hnh_host_backend_name(), (long long)info[0], (long long)info[1], (long long)info[2], r, (long long)checked, worst / top_val);
    const int ok = (info[2] == nnz) && (worst <= 1e-11 * top_val);

    free(got); free(rows); free(cols); free(rowsum);
    TRY(hnh_vec_destroy(S));
    TRY(hnh_vec_destroy(buf));
    TRY(hnh_dense_destroy(A));
    TRY(hnh_dense_destroy(B));
    TRY(hnh_dist_destroy(d));
    TRY(hnh_spmat_destroy(s));
    TRY(hnh_world_destroy(w));
    return ok ? 0 : 1;
}
