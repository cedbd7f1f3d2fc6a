// Development probe (not product code): what can the MI355X memory system deliver for the access pattern of the
// local SDDMM/SpMM kernels — random gathers of whole dense rows (ROW_BYTES contiguous bytes, 16 B per lane) out of a
// working set of a given size?  Sweeps the working set from L2-resident (16 MiB) over Infinity-Cache-resident
// (<= 256 MiB) to DRAM-resident (GiBs), for row widths 128 B .. 2 KiB, with U rows in flight per lane group, and
// prints GB/s.  A sequential read of 4 GiB is the streaming reference.  The numbers bound what `row_kernel` can
// reach at a given R and size of the gathered operand ("what binds the headline kernel").
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/gather_probe tools/gather_probe.hip && tools/gather_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                    \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                                \
        }                                                                                           \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}

// LPR lanes own one gathered row of LPR*16 bytes; every group performs `per_group` gathers, U at a time.
template <int LPR, int U>
__global__ __launch_bounds__(256) void gather_kernel(const double2* __restrict__ base, uint32_t ws_rows, int per_group,
                                                     double* __restrict__ sink, uint64_t seed) {
    extern __shared__ char occupancy_limiter[];  // dynamic LDS only limits how many workgroups a CU can hold
    if (per_group < 0) sink[1] = occupancy_limiter[0];
    const int lig = threadIdx.x % LPR;
    const uint64_t group = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
    double2 acc = make_double2(0.0, 0.0);
    for (int it = 0; it < per_group; it += U) {
        double2 y[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t row = mix(seed + group * (uint64_t)per_group + it + u) % ws_rows;
            y[u] = base[(uint64_t)row * LPR + lig];
        }
#pragma unroll
        for (int u = 0; u < U; u++) { acc.x += y[u].x; acc.y += y[u].y; }
    }
    if (acc.x == 123.456) sink[0] = acc.y;  // never true; keeps the loads alive
}

__global__ __launch_bounds__(256) void stream_kernel(const double2* __restrict__ base, uint64_t n16, double* __restrict__ sink) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    double2 acc = make_double2(0.0, 0.0);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * 4) {
        double2 y[4];
#pragma unroll
        for (int u = 0; u < 4; u++) y[u] = (i + u * stride < n16) ? base[i + u * stride] : make_double2(0, 0);
#pragma unroll
        for (int u = 0; u < 4; u++) { acc.x += y[u].x; acc.y += y[u].y; }
    }
    if (acc.x == 123.456) sink[0] = acc.y;
}

template <int LPR, int U>
double run_gather(const double2* buf, size_t ws_bytes, double* sink, double total_gb, int wgs_per_cu = 8) {
    const uint32_t ws_rows = (uint32_t)(ws_bytes / (LPR * 16));
    const int groups_per_block = 256 / LPR;
    const int blocks = 256 * 32;  // 32 workgroups per CU in the grid
    const uint64_t groups = (uint64_t)blocks * groups_per_block;
    int per_group = (int)(total_gb * 1e9 / (LPR * 16) / groups);
    per_group = (per_group / U) * U;
    if (per_group < U) per_group = U;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0));
        // 160 KiB of LDS per CU: a workgroup that asks for 160 KiB / k of it leaves room for exactly k workgroups (k waves per SIMD)
        const size_t lds = wgs_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / wgs_per_cu) - 64;
        hipLaunchKernelGGL((gather_kernel<LPR, U>), dim3(blocks), dim3(256), lds, 0, buf, ws_rows, per_group, sink, 0x1234567ULL * (rep + 1));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)groups * per_group * LPR * 16;
    return bytes / (best * 1e-3) / 1e9;
}

int main() {
    const size_t max_ws = (size_t)4 << 30;
    double2* buf; double* sink;
    CHECK(hipMalloc(&buf, max_ws)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, max_ws));
    // streaming reference
    {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(stream_kernel, dim3(256 * 32), dim3(256), 0, 0, buf, max_ws / 16, sink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("stream read 4 GiB: %.0f GB/s\n", max_ws / (best * 1e-3) / 1e9);
    }
    const size_t sizes_mib[] = {16, 64, 128, 192, 256, 384, 512, 1024, 2048, 4096};
    printf("%-28s", "working set (MiB):");
    for (size_t s : sizes_mib) printf("%8zu", s);
    printf("\n");
#define ROW(LPR, U, label)                                                         \
    {                                                                              \
        printf("%-28s", label);                                                    \
        for (size_t s : sizes_mib) {                                               \
            printf("%8.0f", run_gather<LPR, U>(buf, s << 20, sink, 40.0));         \
            fflush(stdout);                                                        \
        }                                                                          \
        printf("   GB/s\n");                                                       \
    }
    ROW(64, 8, "row 1024 B (R=128), U=8")
    ROW(64, 4, "row 1024 B (R=128), U=4")
    ROW(64, 16, "row 1024 B (R=128), U=16")
    ROW(32, 8, "row  512 B (R=64),  U=8")
    ROW(16, 8, "row  256 B (R=32),  U=8")
    ROW(16, 16, "row  256 B (R=32),  U=16")
    ROW(8, 8, "row  128 B (R=16),  U=8")
    ROW(8, 16, "row  128 B (R=16),  U=16")
    // how many loads in flight does it take?  waves per SIMD (via an LDS allocation that limits workgroups per CU) x loads per lane
    printf("\nrows of 1 KiB, GB/s by waves per SIMD and gathers in flight per wave (working set 512 MiB | 1024 MiB):\n");
    for (int k : {1, 2, 3, 4, 6, 8}) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_kernel<64, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_kernel<64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_kernel<64, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        printf("  %d waves/SIMD:  U=4 %6.0f | %6.0f    U=8 %6.0f | %6.0f    U=16 %6.0f | %6.0f\n", k,
               run_gather<64, 4>(buf, (size_t)512 << 20, sink, 20.0, k), run_gather<64, 4>(buf, (size_t)1024 << 20, sink, 20.0, k),
               run_gather<64, 8>(buf, (size_t)512 << 20, sink, 20.0, k), run_gather<64, 8>(buf, (size_t)1024 << 20, sink, 20.0, k),
               run_gather<64, 16>(buf, (size_t)512 << 20, sink, 20.0, k), run_gather<64, 16>(buf, (size_t)1024 << 20, sink, 20.0, k));
        fflush(stdout);
    }
    return 0;
}
