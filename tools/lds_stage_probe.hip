// Development probe (not product code): settles "register tile vs LDS-staged dense-row tile" for the fused
// SDDMM -> SpMM row pass by measurement (north_star names LDS-staged tiles; DESIGN.md section 3 argues for registers).
//
// Both kernels do the product kernel's work for R = 128 on a synthetic Erdos-Renyi-like matrix (DEG nonzeros per row,
// column = hash(row, e) mod M computed on the fly, so no index stream dilutes the comparison): one wave per sparse row,
// per batch of U = 8 nonzeros gather 8 dense rows of 1 KiB, 8 dot products with the row operand, write the 8 values,
// 8 axpys into the accumulator; the output row is written once.
//   reg : gathered rows land in VGPRs (global_load_dwordx4), as the product kernel does;
//   lds : gathered rows land in LDS through the asynchronous global_load_lds_dwordx4 (8 KiB of LDS per wave, 32 KiB per
//         workgroup), and the arithmetic reads them back with ds_read_b128 — the LDS-staged tile.
// Nothing is shared between rows on such a matrix, so LDS can only add a round trip; the numbers show how much.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_stage_probe tools/lds_stage_probe.hip && tools/lds_stage_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                    \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                                \
        }                                                                                           \
    } while (0)

constexpr int R = 128, U = 8, DEG = 96;

__device__ __forceinline__ uint32_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void fused_like_kernel(int64_t rows, uint32_t m, const double2* __restrict__ X, const double2* __restrict__ Y,
                                                          double2* __restrict__ Out, double* __restrict__ values) {
    __shared__ double2 stage[4][U][64];  // 4 waves x 8 rows x 1 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const double2 x = X[row * 64 + lane];
    double2 acc = make_double2(0.0, 0.0);
    for (int e = 0; e < DEG; e += U) {
        double2 y[U];
        uint32_t col[U];
#pragma unroll
        for (int u = 0; u < U; u++) col[u] = mix((uint64_t)row * DEG + e + u) % m;
        if constexpr (USE_LDS) {
#pragma unroll
            for (int u = 0; u < U; u++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Y + (uint64_t)col[u] * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(&stage[wave][u][0]), 16, 0, 0);
            __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) lgkmcnt(0): the rows have landed
#pragma unroll
            for (int u = 0; u < U; u++) y[u] = stage[wave][u][lane];
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) y[u] = Y[(uint64_t)col[u] * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const double d = wave_sum(x.x * y[u].x + x.y * y[u].y);
            if (lane == 0) values[row * DEG + e + u] = d;
            acc.x = fma(d, y[u].x, acc.x);
            acc.y = fma(d, y[u].y, acc.y);
        }
    }
    Out[row * 64 + lane] = acc;
}

template <bool USE_LDS>
float run(int64_t rows, uint32_t m, const double2* X, const double2* Y, double2* Out, double* values) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((fused_like_kernel<USE_LDS>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, 0, rows, m, X, Y, Out, values);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t row_bytes = R * sizeof(double);
    for (int logm : {19, 20}) {  // gathered operand of 512 MiB (about the product's cache panel) and 1 GiB
        const int64_t rows = 1ll << 20;
        const uint32_t m = 1u << logm;
        double2 *X, *Y, *Out;
        double* values;
        CHECK(hipMalloc(&X, rows * row_bytes)); CHECK(hipMalloc(&Out, rows * row_bytes));
        CHECK(hipMalloc(&Y, (size_t)m * row_bytes)); CHECK(hipMalloc(&values, rows * DEG * sizeof(double)));
        CHECK(hipMemset(X, 0, rows * row_bytes)); CHECK(hipMemset(Y, 0, (size_t)m * row_bytes));
        const double bytes = (double)rows * DEG * (row_bytes + 8) + 2.0 * rows * row_bytes;
        const float t_reg = run<false>(rows, m, X, Y, Out, values), t_lds = run<true>(rows, m, X, Y, Out, values);
        printf("rows 2^20 x %d nnz, gathered operand %4zu MiB:  register tile %.3f ms (%.2f TB/s)   LDS-staged tile %.3f ms (%.2f TB/s)   LDS/reg = %.3f\n",
               DEG, (size_t)m * row_bytes >> 20, t_reg, bytes / t_reg / 1e9, t_lds, bytes / t_lds / 1e9, t_lds / t_reg);
        CHECK(hipFree(X)); CHECK(hipFree(Y)); CHECK(hipFree(Out)); CHECK(hipFree(values));
    }
    return 0;
}
