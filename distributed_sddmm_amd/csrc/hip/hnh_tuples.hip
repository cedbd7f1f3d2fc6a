// Setup pipeline on the device: the (row, col, value) tuples of the sparse matrix are routed, ordered and turned
// into CSR blocks without going back to the host (hnh_tuples_* of include/hnh_kernels.h).
//
// What this replaces in the reference (all host code, SpmatLocal.hpp): getOwner + the Alltoallv pack (:45-52,
// :404-420), std::sort(column_major) (:454), divideIntoBlockCols (:541-563), and the MKL COO -> CSR conversion in the
// CSRLocal constructor (:117-147).  At config-2 size (1e8 tuples, twice: S and S^T) those are ~9 s of host time on a
// 256-thread box; here they are radix sorts over 64-bit keys (rocPRIM, ~10 ms each on MI355X), a few streaming
// kernels, and binary searches for the bucket boundaries.
//
// Sorting 24-byte tuples: sort (key, index) pairs, then gather the tuples through the permutation — the radix
// passes then move 12 bytes per element instead of 32.  Every sort is stable (LSD radix), which is what lets the
// callers compose orders (e.g. CSR order inside column blocks).
#include <cstring>  // rocPRIM's texture iterator uses memset without including it

#include <rocprim/rocprim.hpp>

#include "hnh_ctx.hpp"

namespace {

constexpr int kBlock = 256;

struct KeyDesc {
    int kind, transpose;
    long long rows_in_block, cols_in_block, n_col_blocks, div;
    const int32_t* owner_table;
};

__device__ __forceinline__ unsigned long long key_of(const hnh_tuple& t, const KeyDesc& k) {
    switch (k.kind) {
        case HNH_KEY_ROW_COL: return (t.r << 32) | (t.c & 0xffffffffull);
        case HNH_KEY_COL_ROW: return (t.c << 32) | (t.r & 0xffffffffull);
        case HNH_KEY_OWNER: {
            const unsigned long long rb = (k.transpose ? t.c : t.r) / (unsigned long long)k.rows_in_block;
            const unsigned long long cb = (k.transpose ? t.r : t.c) / (unsigned long long)k.cols_in_block;
            return (unsigned long long)(unsigned)k.owner_table[rb * (unsigned long long)k.n_col_blocks + cb];
        }
        default: return t.c / (unsigned long long)k.div;  // HNH_KEY_COL_DIV
    }
}

__global__ __launch_bounds__(kBlock) void make_keys_kernel(const hnh_tuple* __restrict__ t, long long n, KeyDesc k,
                                                           unsigned long long* __restrict__ keys, unsigned* __restrict__ idx,
                                                           int* __restrict__ bad) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const hnh_tuple v = t[i];
        if ((k.kind == HNH_KEY_ROW_COL || k.kind == HNH_KEY_COL_ROW) && ((v.r >> 32) != 0 || (v.c >> 32) != 0)) *bad = 1;
        keys[i] = key_of(v, k);
        idx[i] = (unsigned)i;
    }
}

__global__ __launch_bounds__(kBlock) void gather_kernel(const hnh_tuple* __restrict__ src, const unsigned* __restrict__ perm, long long n,
                                                        hnh_tuple* __restrict__ dst) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) dst[i] = src[perm[i]];
}

// starts[b] = first index whose key is >= b, b = 0 .. nbuckets (keys non-decreasing)
__global__ __launch_bounds__(kBlock) void bucket_starts_kernel(const hnh_tuple* __restrict__ t, long long n, KeyDesc k, long long nbuckets,
                                                               long long* __restrict__ starts) {
    const long long b = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (b > nbuckets) return;
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (key_of(t[mid], k) < (unsigned long long)b) lo = mid + 1;
        else hi = mid;
    }
    starts[b] = lo;
}

__global__ __launch_bounds__(kBlock) void transform_kernel(hnh_tuple* t, long long n, int swap_rc, unsigned long long rmod,
                                                           unsigned long long cmod) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        hnh_tuple v = t[i];
        if (swap_rc) { const unsigned long long x = v.r; v.r = v.c; v.c = x; }
        if (rmod) v.r %= rmod;
        if (cmod) v.c %= cmod;
        t[i] = v;
    }
}

__global__ __launch_bounds__(kBlock) void remap_cols_kernel(hnh_tuple* t, long long n, unsigned long long div, unsigned long long sub_div,
                                                            unsigned long long n_sub, const long long* __restrict__ dest, long long ndest,
                                                            int* __restrict__ bad) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const unsigned long long c = t[i].c, in = c % div;
        const unsigned long long seg = (c / div) * n_sub + in / sub_div;
        if ((long long)seg >= ndest || dest[seg] < 0) { *bad = 1; continue; }
        t[i].c = (unsigned long long)dest[seg] + in % sub_div;
    }
}

// de-duplication with maximum (tuples sorted by (row, col)): a tuple is the head of its run when it differs from its
// predecessor; heads are numbered by an inclusive scan; every head walks its (short) run for the maximum
__global__ __launch_bounds__(kBlock) void head_flags_kernel(const hnh_tuple* __restrict__ t, long long n, unsigned* __restrict__ flag) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        flag[i] = (i == 0 || t[i].r != t[i - 1].r || t[i].c != t[i - 1].c) ? 1u : 0u;
}
__global__ __launch_bounds__(kBlock) void compact_max_kernel(const hnh_tuple* __restrict__ t, long long n, const unsigned* __restrict__ flag,
                                                             const unsigned* __restrict__ pos, hnh_tuple* __restrict__ out) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        if (!flag[i]) continue;
        hnh_tuple v = t[i];
        for (long long j = i + 1; j < n && !flag[j]; j++) v.value = fmax(v.value, t[j].value);
        out[pos[i] - 1] = v;
    }
}
__global__ __launch_bounds__(kBlock) void take_strided_kernel(const hnh_tuple* __restrict__ src, long long first, long long step,
                                                              hnh_tuple* __restrict__ out, long long n_out) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n_out; i += stride) out[i] = src[first + i * step];
}

// tuples in (row, col) order -> col_idx / values; rowptr[r] = first tuple of row >= r (binary search per row);
// max_row[0] = longest row; bad[0] = 1 when a tuple lies outside rows x cols
__global__ __launch_bounds__(kBlock) void unzip_kernel(const hnh_tuple* __restrict__ t, long long n, long long rows, long long cols,
                                                       int32_t* __restrict__ col_idx, double* __restrict__ values, int* __restrict__ bad) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const hnh_tuple v = t[i];
        if ((long long)v.r >= rows || (long long)v.c >= cols) *bad = 1;
        col_idx[i] = (int32_t)v.c;
        values[i] = v.value;
    }
}

__global__ __launch_bounds__(kBlock) void rowptr_kernel(const hnh_tuple* __restrict__ t, long long n, long long rows,
                                                        int32_t* __restrict__ rowptr) {
    const long long r = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (r > rows) return;
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if ((long long)t[mid].r < r) lo = mid + 1;
        else hi = mid;
    }
    rowptr[r] = (int32_t)lo;
}

__global__ __launch_bounds__(kBlock) void max_row_kernel(const int32_t* __restrict__ rowptr, long long rows, int* __restrict__ out) {
    const long long stride = (long long)gridDim.x * kBlock;
    int m = 0;
    for (long long r = (long long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += stride) m = max(m, rowptr[r + 1] - rowptr[r]);
    for (int off = 32; off >= 1; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

__device__ __forceinline__ unsigned long long splitmix64_dev(unsigned long long x) {
    unsigned long long z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// draw k of the counter-based Erdos-Renyi generator (er_generator.hpp): key = row * n + col
__global__ __launch_bounds__(kBlock) void er_keys_kernel(unsigned long long m, unsigned long long n, unsigned long long draws,
                                                         unsigned long long seed, unsigned long long* __restrict__ keys) {
    const unsigned long long G = 0x9E3779B97F4A7C15ull;
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < draws; k += stride) {
        const unsigned long long base = seed + (2 * k) * G;
        keys[k] = (splitmix64_dev(base) % m) * n + splitmix64_dev(base + G) % n;
    }
}

// edge k of the counter-based R-MAT generator (er_generator.hpp: rmat_keys — the same doubles, the same comparisons, bit for bit): logm
// levels, each picking a quadrant with probabilities (a, b, c, 1 - a - b - c) from its own draw; optional multiplicative scramble of
// both vertex numbers (the hubs are then spread over the rows instead of sitting at the low numbers)
__global__ __launch_bounds__(kBlock) void rmat_keys_kernel(int logm, unsigned long long edges, double a, double ab, double abc, unsigned long long seed,
                                                           int scramble, unsigned long long* __restrict__ keys) {
    const unsigned long long G = 0x9E3779B97F4A7C15ull, n = 1ull << logm, mask = n - 1;
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < edges; k += stride) {
        unsigned long long r = 0, col = 0;
        for (int l = 0; l < logm; l++) {
            const double u = (double)(splitmix64_dev(seed + (k * (unsigned long long)logm + (unsigned long long)l) * G) >> 11) * 0x1.0p-53;
            const unsigned long long rb = (u >= ab) ? 1 : 0;
            const unsigned long long cb = ((u >= a && u < ab) || (u >= abc)) ? 1 : 0;
            r = (r << 1) | rb;
            col = (col << 1) | cb;
        }
        if (scramble) {
            r = (r * 0x9E3779B1ull + 0x7F4A7C15ull) & mask;
            col = (col * 0x9E3779B1ull + 0x7F4A7C15ull) & mask;
        }
        keys[k] = r * n + col;
    }
}

__global__ __launch_bounds__(kBlock) void tuples_from_keys_kernel(const unsigned long long* __restrict__ keys, unsigned long long ncols,
                                                                  long long first, long long stride_keys, double value,
                                                                  hnh_tuple* __restrict__ out, long long n_out) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n_out; i += stride) {
        const unsigned long long key = keys[first + i * stride_keys];
        hnh_tuple t;
        t.r = key / ncols; t.c = key % ncols; t.value = value;
        out[i] = t;
    }
}

__global__ __launch_bounds__(kBlock) void relabel_kernel(hnh_tuple* t, long long n, const unsigned long long* __restrict__ row_label,
                                                         const unsigned long long* __restrict__ col_label) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        hnh_tuple v = t[i];
        v.r = row_label[v.r];
        v.c = col_label[v.c];
        t[i] = v;
    }
}

unsigned grid_for(long long n) {
    long long b = (n + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > 256 * 32) b = 256 * 32;  // grid-stride beyond 32 workgroups per CU
    return (unsigned)b;
}

int desc_from(hnh_ctx* ctx, const hnh_tuple_key* key, KeyDesc* d, const char* who) {
    if (!key) return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": null key");
    d->kind = key->kind; d->transpose = key->transpose;
    d->rows_in_block = key->rows_in_block; d->cols_in_block = key->cols_in_block; d->n_col_blocks = key->n_col_blocks;
    d->div = key->div; d->owner_table = key->owner_table;
    switch (key->kind) {
        case HNH_KEY_ROW_COL: case HNH_KEY_COL_ROW: return HNH_OK;
        case HNH_KEY_OWNER:
            if (key->rows_in_block <= 0 || key->cols_in_block <= 0 || key->n_col_blocks <= 0 || !key->owner_table)
                return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": incomplete owner key");
            return HNH_OK;
        case HNH_KEY_COL_DIV:
            if (key->div <= 0) return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": column divisor must be positive");
            return HNH_OK;
        default: return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": unknown key kind");
    }
}

struct Scratch {  // frees on scope exit (setup path: plain hipMalloc / hipFree, synchronised by the caller's stream sync)
    void* p = nullptr;
    ~Scratch() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" {

int hnh_tuples_sort(hnh_ctx* ctx, hnh_tuple* tuples, int64_t n, const hnh_tuple_key* key, int key_bits, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_sort: negative size");
    KeyDesc d;
    if (int rc = desc_from(ctx, key, &d, "hnh_tuples_sort")) return rc;
    if (n <= 1) return HNH_OK;
    if (!tuples) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_sort: null pointer");
    if (n > 0xffffffffLL) return hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "hnh_tuples_sort: more than 2^32 tuples in one call");
    if (key_bits <= 0 || key_bits > 64) key_bits = 64;
    hipStream_t st = ctx->streams[stream];
    const size_t un = (size_t)n;
    size_t tmp_bytes = 0;
    HNH_TRY_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                               (unsigned*)nullptr, (unsigned*)nullptr, un, 0, (unsigned)key_bits, st));
    // one allocation: keys in/out, index in/out, flag, radix scratch, tuple copy
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_k0 = 0, o_k1 = o_k0 + align(un * 8), o_i0 = o_k1 + align(un * 8), o_i1 = o_i0 + align(un * 4), o_bad = o_i1 + align(un * 4),
                 o_tmp = o_bad + 256, o_cp = o_tmp + align(tmp_bytes), total = o_cp + align(un * sizeof(hnh_tuple));
    Scratch s;
    HNH_TRY_HIP(ctx, hipMalloc(&s.p, total));
    char* base = static_cast<char*>(s.p);
    auto* k0 = reinterpret_cast<unsigned long long*>(base + o_k0);
    auto* k1 = reinterpret_cast<unsigned long long*>(base + o_k1);
    auto* i0 = reinterpret_cast<unsigned*>(base + o_i0);
    auto* i1 = reinterpret_cast<unsigned*>(base + o_i1);
    int* bad = reinterpret_cast<int*>(base + o_bad);
    auto* cp = reinterpret_cast<hnh_tuple*>(base + o_cp);
    HNH_TRY_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(int), st));
    hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, tuples, (long long)n, d, k0, i0, bad);
    HNH_TRY_HIP(ctx, hipGetLastError());
    HNH_TRY_HIP(ctx, rocprim::radix_sort_pairs(base + o_tmp, tmp_bytes, k0, k1, i0, i1, un, 0, (unsigned)key_bits, st));
    HNH_TRY_HIP(ctx, hipMemcpyAsync(cp, tuples, un * sizeof(hnh_tuple), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, cp, i1, (long long)n, tuples);
    HNH_TRY_HIP(ctx, hipGetLastError());
    int h_bad = 0;
    HNH_TRY_HIP(ctx, hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    HNH_TRY_HIP(ctx, hipStreamSynchronize(st));  // scratch dies with this scope
    if (h_bad) return hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "hnh_tuples_sort: a row or column index does not fit 32 bits");
    return HNH_OK;
}

int hnh_tuples_bucket_starts(hnh_ctx* ctx, const hnh_tuple* sorted, int64_t n, const hnh_tuple_key* key, int64_t nbuckets,
                             int64_t* starts_host, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0 || nbuckets < 0 || !starts_host) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_bucket_starts: bad argument");
    KeyDesc d;
    if (int rc = desc_from(ctx, key, &d, "hnh_tuples_bucket_starts")) return rc;
    if (n == 0) {
        for (int64_t b = 0; b <= nbuckets; b++) starts_host[b] = 0;
        return HNH_OK;
    }
    if (!sorted) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_bucket_starts: null pointer");
    hipStream_t st = ctx->streams[stream];
    Scratch s;
    HNH_TRY_HIP(ctx, hipMalloc(&s.p, (size_t)(nbuckets + 1) * sizeof(long long)));
    hipLaunchKernelGGL(bucket_starts_kernel, dim3((unsigned)((nbuckets + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, sorted, (long long)n,
                       d, (long long)nbuckets, static_cast<long long*>(s.p));
    HNH_TRY_HIP(ctx, hipGetLastError());
    static_assert(sizeof(long long) == sizeof(int64_t), "int64_t layout");
    HNH_TRY_HIP(ctx, hipMemcpyAsync(starts_host, s.p, (size_t)(nbuckets + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    return hnh::check_hip(ctx, hipStreamSynchronize(st), "hipStreamSynchronize");
}

int hnh_tuples_transform(hnh_ctx* ctx, hnh_tuple* tuples, int64_t n, int swap_rc, uint64_t rmod, uint64_t cmod, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_transform: negative size");
    if (n == 0 || (!swap_rc && !rmod && !cmod)) return HNH_OK;
    if (!tuples) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_transform: null pointer");
    hipLaunchKernelGGL(transform_kernel, dim3(grid_for(n)), dim3(kBlock), 0, ctx->streams[stream], tuples, (long long)n, swap_rc,
                       (unsigned long long)rmod, (unsigned long long)cmod);
    return hnh::check_hip(ctx, hipGetLastError(), "transform_kernel launch");
}

int hnh_tuples_dedup_max(hnh_ctx* ctx, hnh_tuple* sorted, int64_t n, int64_t* n_unique_host, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0 || !n_unique_host) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_dedup_max: bad argument");
    *n_unique_host = 0;
    if (n == 0) return HNH_OK;
    if (!sorted) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_dedup_max: null pointer");
    if (n > 0xffffffffLL) return hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "hnh_tuples_dedup_max: more than 2^32 tuples");
    hipStream_t st = ctx->streams[stream];
    const size_t un = (size_t)n;
    size_t scan_bytes = 0;
    HNH_TRY_HIP(ctx, rocprim::inclusive_scan(nullptr, scan_bytes, (unsigned*)nullptr, (unsigned*)nullptr, un, rocprim::plus<unsigned>(), st));
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_flag = 0, o_pos = o_flag + up(un * sizeof(unsigned)), o_out = o_pos + up(un * sizeof(unsigned)),
                 o_tmp = o_out + up(un * sizeof(hnh_tuple));
    Scratch s;
    HNH_TRY_HIP(ctx, hipMalloc(&s.p, o_tmp + up(scan_bytes)));
    char* base = static_cast<char*>(s.p);
    unsigned* flag = reinterpret_cast<unsigned*>(base + o_flag);
    unsigned* pos = reinterpret_cast<unsigned*>(base + o_pos);
    hnh_tuple* out = reinterpret_cast<hnh_tuple*>(base + o_out);
    hipLaunchKernelGGL(head_flags_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, sorted, (long long)n, flag);
    HNH_TRY_HIP(ctx, rocprim::inclusive_scan(base + o_tmp, scan_bytes, flag, pos, un, rocprim::plus<unsigned>(), st));
    hipLaunchKernelGGL(compact_max_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, sorted, (long long)n, flag, pos, out);
    HNH_TRY_HIP(ctx, hipGetLastError());
    unsigned count = 0;
    HNH_TRY_HIP(ctx, hipMemcpyAsync(&count, pos + (un - 1), sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
    HNH_TRY_HIP(ctx, hipMemcpyAsync(sorted, out, (size_t)count * sizeof(hnh_tuple), hipMemcpyDeviceToDevice, st));
    HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
    *n_unique_host = (int64_t)count;
    return HNH_OK;
}

int hnh_tuples_take_strided(hnh_ctx* ctx, const hnh_tuple* src, int64_t first, int64_t stride, hnh_tuple* out, int64_t n_out, int stream) {
    HNH_ENTER(ctx, stream);
    if (n_out < 0 || first < 0 || stride <= 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_take_strided: bad argument");
    if (n_out == 0) return HNH_OK;
    if (!src || !out) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_take_strided: null pointer");
    hipLaunchKernelGGL(take_strided_kernel, dim3(grid_for(n_out)), dim3(kBlock), 0, ctx->streams[stream], src, (long long)first, (long long)stride,
                       out, (long long)n_out);
    return hnh::check_hip(ctx, hipGetLastError(), "take_strided_kernel launch");
}

int hnh_tuples_remap_cols(hnh_ctx* ctx, hnh_tuple* tuples, int64_t n, int64_t div, int64_t sub_div, int64_t n_sub, const int64_t* dest_host,
                          int64_t ndest, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0 || div <= 0 || sub_div <= 0 || n_sub <= 0 || ndest <= 0 || !dest_host) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_remap_cols: bad argument");
    if (n == 0) return HNH_OK;
    if (!tuples) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_remap_cols: null pointer");
    hipStream_t st = ctx->streams[stream];
    Scratch s;
    HNH_TRY_HIP(ctx, hipMalloc(&s.p, (size_t)ndest * sizeof(long long) + sizeof(int)));
    long long* dtable = static_cast<long long*>(s.p);
    int* bad = reinterpret_cast<int*>(dtable + ndest);
    HNH_TRY_HIP(ctx, hipMemcpyAsync(dtable, dest_host, (size_t)ndest * sizeof(long long), hipMemcpyHostToDevice, st));
    HNH_TRY_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(int), st));
    hipLaunchKernelGGL(remap_cols_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, tuples, (long long)n, (unsigned long long)div,
                       (unsigned long long)sub_div, (unsigned long long)n_sub, dtable, (long long)ndest, bad);
    HNH_TRY_HIP(ctx, hipGetLastError());
    int h = 0;
    HNH_TRY_HIP(ctx, hipMemcpyAsync(&h, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
    if (h) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_remap_cols: a tuple lies in a segment that has no destination");
    return HNH_OK;
}

int hnh_tuples_to_csr(hnh_ctx* ctx, const hnh_tuple* sorted, int64_t n, int64_t rows, int64_t cols, int32_t* rowptr, int32_t* col_idx,
                      double* values, int* max_row_nnz_host, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0 || rows < 0 || cols < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_to_csr: negative size");
    if (n > 0x7fffffffLL) return hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "hnh_tuples_to_csr: block has more than 2^31 nonzeros");
    if (!rowptr || (n > 0 && (!sorted || !col_idx || !values))) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_to_csr: null pointer");
    hipStream_t st = ctx->streams[stream];
    Scratch s;
    HNH_TRY_HIP(ctx, hipMalloc(&s.p, 2 * sizeof(int)));
    int* flags = static_cast<int*>(s.p);  // [0] = bad tuple, [1] = longest row
    HNH_TRY_HIP(ctx, hipMemsetAsync(flags, 0, 2 * sizeof(int), st));
    if (n > 0) hipLaunchKernelGGL(unzip_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, sorted, (long long)n, (long long)rows, (long long)cols,
                                  col_idx, values, flags);
    hipLaunchKernelGGL(rowptr_kernel, dim3((unsigned)((rows + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, sorted, (long long)n,
                       (long long)rows, rowptr);
    if (rows > 0) hipLaunchKernelGGL(max_row_kernel, dim3(grid_for(rows)), dim3(kBlock), 0, st, rowptr, (long long)rows, flags + 1);
    HNH_TRY_HIP(ctx, hipGetLastError());
    int h[2] = {0, 0};
    HNH_TRY_HIP(ctx, hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, st));
    HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
    if (h[0]) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_to_csr: nonzero outside its block");
    if (max_row_nnz_host) *max_row_nnz_host = h[1];
    return HNH_OK;
}

}  // extern "C"

namespace {
// draws of a generator -> sorted unique keys in place (radix sort + unique), their number read back
template <typename Fill>
int generate_sorted_unique(hnh_ctx* ctx, hipStream_t st, uint64_t draws, unsigned bits, uint64_t* keys, int64_t* n_unique_host, Fill&& fill) {
    const size_t un = (size_t)draws;
    size_t sort_bytes = 0, uniq_bytes = 0;
    HNH_TRY_HIP(ctx, rocprim::radix_sort_keys(nullptr, sort_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, un, 0, bits, st));
    HNH_TRY_HIP(ctx, rocprim::unique(nullptr, uniq_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t*)nullptr, un,
                                     rocprim::equal_to<unsigned long long>(), st));
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t tmp_bytes = sort_bytes > uniq_bytes ? sort_bytes : uniq_bytes;
    const size_t o_alt = 0, o_cnt = o_alt + align(un * 8), o_tmp = o_cnt + 256, total = o_tmp + align(tmp_bytes);
    Scratch s;
    HNH_TRY_HIP(ctx, hipMalloc(&s.p, total));
    char* base = static_cast<char*>(s.p);
    auto* alt = reinterpret_cast<unsigned long long*>(base + o_alt);
    auto* cnt = reinterpret_cast<size_t*>(base + o_cnt);
    auto* k = reinterpret_cast<unsigned long long*>(keys);
    fill(k);
    HNH_TRY_HIP(ctx, hipGetLastError());
    HNH_TRY_HIP(ctx, rocprim::radix_sort_keys(base + o_tmp, sort_bytes, k, alt, un, 0, bits, st));
    HNH_TRY_HIP(ctx, rocprim::unique(base + o_tmp, uniq_bytes, alt, k, cnt, un, rocprim::equal_to<unsigned long long>(), st));
    size_t h_cnt = 0;
    HNH_TRY_HIP(ctx, hipMemcpyAsync(&h_cnt, cnt, sizeof(size_t), hipMemcpyDeviceToHost, st));
    HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
    *n_unique_host = (int64_t)h_cnt;
    return HNH_OK;
}
}  // namespace

extern "C" {

int hnh_generate_er_keys(hnh_ctx* ctx, uint64_t m, uint64_t n, uint64_t draws, uint64_t seed, uint64_t* keys, int64_t* n_unique_host,
                         int stream) {
    HNH_ENTER(ctx, stream);
    if (!n_unique_host || m == 0 || n == 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_generate_er_keys: bad argument");
    *n_unique_host = 0;
    if (draws == 0) return HNH_OK;
    if (!keys) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_generate_er_keys: null pointer");
    if (m > (~0ull) / n) return hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "hnh_generate_er_keys: m * n overflows 64 bits");
    hipStream_t st = ctx->streams[stream];
    unsigned bits = 1;
    while (bits < 64 && (1ull << bits) < m * n) bits++;
    return generate_sorted_unique(ctx, st, draws, bits, keys, n_unique_host, [&](unsigned long long* k) {
        hipLaunchKernelGGL(er_keys_kernel, dim3(grid_for((long long)draws)), dim3(kBlock), 0, st, (unsigned long long)m, (unsigned long long)n,
                           (unsigned long long)draws, (unsigned long long)seed, k);
    });
}

int hnh_generate_rmat_keys(hnh_ctx* ctx, int logm, uint64_t edges, double a, double b, double c, uint64_t seed, int scramble, uint64_t* keys,
                           int64_t* n_unique_host, int stream) {
    HNH_ENTER(ctx, stream);
    if (!n_unique_host || logm < 1 || logm > 31 || a < 0.0 || b < 0.0 || c < 0.0 || a + b + c > 1.0)
        return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_generate_rmat_keys: bad argument");
    *n_unique_host = 0;
    if (edges == 0) return HNH_OK;
    if (!keys) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_generate_rmat_keys: null pointer");
    hipStream_t st = ctx->streams[stream];
    const double ab = a + b, abc = a + b + c;  // (the sums the host generator compares with: formed once, in the same order)
    return generate_sorted_unique(ctx, st, edges, (unsigned)(2 * logm), keys, n_unique_host, [&](unsigned long long* k) {
        hipLaunchKernelGGL(rmat_keys_kernel, dim3(grid_for((long long)edges)), dim3(kBlock), 0, st, logm, (unsigned long long)edges, a, ab, abc,
                           (unsigned long long)seed, scramble, k);
    });
}

int hnh_tuples_from_keys(hnh_ctx* ctx, const uint64_t* keys, uint64_t ncols, int64_t first, int64_t stride_keys, double value,
                         hnh_tuple* out, int64_t n_out, int stream) {
    HNH_ENTER(ctx, stream);
    if (n_out < 0 || first < 0 || stride_keys <= 0 || ncols == 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_from_keys: bad argument");
    if (n_out == 0) return HNH_OK;
    if (!keys || !out) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_from_keys: null pointer");
    hipLaunchKernelGGL(tuples_from_keys_kernel, dim3(grid_for(n_out)), dim3(kBlock), 0, ctx->streams[stream],
                       reinterpret_cast<const unsigned long long*>(keys), (unsigned long long)ncols, (long long)first, (long long)stride_keys,
                       value, out, (long long)n_out);
    return hnh::check_hip(ctx, hipGetLastError(), "tuples_from_keys_kernel launch");
}

int hnh_tuples_relabel(hnh_ctx* ctx, hnh_tuple* tuples, int64_t n, const uint64_t* row_label, const uint64_t* col_label, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_relabel: negative size");
    if (n == 0) return HNH_OK;
    if (!tuples || !row_label || !col_label) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_tuples_relabel: null pointer");
    hipLaunchKernelGGL(relabel_kernel, dim3(grid_for(n)), dim3(kBlock), 0, ctx->streams[stream], tuples, (long long)n,
                       reinterpret_cast<const unsigned long long*>(row_label), reinterpret_cast<const unsigned long long*>(col_label));
    return hnh::check_hip(ctx, hipGetLastError(), "relabel_kernel launch");
}

}  // extern "C"
