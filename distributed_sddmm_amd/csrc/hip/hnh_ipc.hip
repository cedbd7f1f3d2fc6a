// Peer-to-peer pull over mapped peer memory — the "ipc" entry points of include/hnh_kernels.h.
//
// The reference moves the dense operand with MPI_Sendrecv (distributed_sparse.h:351-361) and the sparse block with
// Isend/Irecv sets (SpmatLocal.hpp:200-259).  The RCCL transport (hnh_comm.hip) turns those into send/recv kernels on both
// sides; here the RECEIVER copies straight out of the sender's buffer, which it has mapped once through an interprocess
// memory handle — over xGMI when the peer owns another GPU of the node, plainly when two processes share one GPU (which is
// what lets the cross-process path be tested on a one-GPU box).  Ordering between the two processes' streams comes from
// 64-bit flag words in a host shared-memory region both have registered with HIP: values only grow, a waiter asks for
// "at least v", so nothing depends on when the peer enqueued its write and the host never has to wait for the device.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "hnh_ctx.hpp"

static_assert(sizeof(hipIpcMemHandle_t) <= HNH_IPC_HANDLE_BYTES, "memory handle size");

namespace {

__global__ void flag_write_kernel(unsigned long long* f, unsigned long long v) {
    __threadfence_system();
    __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void flag_wait_kernel(const unsigned long long* f, unsigned long long v) {
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) __builtin_amdgcn_s_sleep(8);
}

typedef double v2d __attribute__((ext_vector_type(2)));

struct PullList {
    char* dst[HNH_IPC_MAX_PULL];
    const char* src[HNH_IPC_MAX_PULL];
    size_t bytes[HNH_IPC_MAX_PULL];
};

// One launch pulls every source of a group: copy c is cut into `wgs` contiguous parts, workgroup (c, part) streams its part in
// tiles of 256 lanes x 8 x 16 bytes — eight independent 16-byte loads in flight per lane before the first store, which is what
// a load over a link needs to keep the link busy.  Sources that are not 16-byte aligned fall back to 8-byte elements.
template <typename V>
__device__ __forceinline__ void pull_part(V* __restrict__ d, const V* __restrict__ s, size_t lo, size_t hi) {
    constexpr int kU = 8;
    for (size_t base = lo; base < hi; base += (size_t)256 * kU) {
        V v[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < hi) v[u] = __builtin_nontemporal_load(s + i);
        }
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < hi) d[i] = v[u];
        }
    }
}

__global__ __launch_bounds__(256) void pull_kernel(PullList list, int wgs) {
    const int c = (int)blockIdx.x / wgs, part = (int)blockIdx.x % wgs;
    char* dst = list.dst[c];
    const char* src = list.src[c];
    const size_t bytes = list.bytes[c];
    const bool wide = (((uintptr_t)dst | (uintptr_t)src | bytes) & 15) == 0;
    if (wide) {
        const size_t n = bytes / 16, per = ((n + (size_t)wgs - 1) / (size_t)wgs + 255) / 256 * 256;
        const size_t lo = per * (size_t)part, hi = lo + per < n ? lo + per : n;
        if (lo < hi) pull_part(reinterpret_cast<v2d*>(dst), reinterpret_cast<const v2d*>(src), lo, hi);
    } else if ((((uintptr_t)dst | (uintptr_t)src) & 7) == 0) {
        const size_t n = bytes / 8, per = ((n + (size_t)wgs - 1) / (size_t)wgs + 255) / 256 * 256;
        const size_t lo = per * (size_t)part, hi = lo + per < n ? lo + per : n;
        if (lo < hi) pull_part(reinterpret_cast<double*>(dst), reinterpret_cast<const double*>(src), lo, hi);
        if (part == 0 && threadIdx.x < (bytes & 7)) dst[n * 8 + threadIdx.x] = src[n * 8 + threadIdx.x];
    } else if ((((uintptr_t)dst | (uintptr_t)src) & 3) == 0) {  // e.g. an int32 slice at an odd element offset (device_alltoallv displacements)
        const size_t n = bytes / 4, per = ((n + (size_t)wgs - 1) / (size_t)wgs + 255) / 256 * 256;
        const size_t lo = per * (size_t)part, hi = lo + per < n ? lo + per : n;
        if (lo < hi) pull_part(reinterpret_cast<int*>(dst), reinterpret_cast<const int*>(src), lo, hi);
        if (part == 0 && threadIdx.x < (bytes & 3)) dst[n * 4 + threadIdx.x] = src[n * 4 + threadIdx.x];
    } else {  // byte-displaced: nothing may be assumed
        const size_t per = ((bytes + (size_t)wgs - 1) / (size_t)wgs + 255) / 256 * 256;
        const size_t lo = per * (size_t)part, hi = lo + per < bytes ? lo + per : bytes;
        if (lo < hi) pull_part(dst, src, lo, hi);
    }
}

bool flag_kernels(hnh_ctx* ctx) {
    if (ctx->flag_kernels < 0) {
        const char* m = std::getenv("HNH_IPC_FLAGS");
        int can = 0;
        const bool memops = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, ctx->device) == hipSuccess && can != 0;
        ctx->flag_kernels = (m && std::strcmp(m, "kernel") == 0) || !memops ? 1 : 0;
        (void)hipGetLastError();
    }
    return ctx->flag_kernels == 1;
}

int ensure_aux(hnh_ctx* ctx, int k) {
    if (!ctx->aux_fork) HNH_TRY_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
    if (!ctx->aux[k]) {
        HNH_TRY_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux[k], hipStreamNonBlocking));
        HNH_TRY_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_join[k], hipEventDisableTiming));
    }
    return HNH_OK;
}

}  // namespace

extern "C" {

int hnh_ipc_export(hnh_ctx* ctx, const void* ptr, void* handle_host, uint64_t* offset, uint64_t* alloc_bytes) {
    if (!ctx || !ptr || !handle_host || !offset || !alloc_bytes) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    hipDeviceptr_t base = nullptr;
    size_t range = 0;
    HNH_TRY_HIP(ctx, hipMemGetAddressRange(&base, &range, (hipDeviceptr_t)ptr));
    hipIpcMemHandle_t h;
    HNH_TRY_HIP(ctx, hipIpcGetMemHandle(&h, (void*)base));
    std::memset(handle_host, 0, HNH_IPC_HANDLE_BYTES);
    std::memcpy(handle_host, &h, sizeof(h));
    *offset = (uint64_t)((const char*)ptr - (const char*)base);
    *alloc_bytes = (uint64_t)range;
    return HNH_OK;
}

int hnh_ipc_open(hnh_ctx* ctx, const void* handle_host, uint64_t alloc_bytes, void** base) {
    if (!ctx || !handle_host || !base) return HNH_ERR_INVALID;
    (void)alloc_bytes;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle_host, sizeof(h));
    *base = nullptr;
    return hnh::check_hip(ctx, hipIpcOpenMemHandle(base, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
}

int hnh_ipc_close(hnh_ctx* ctx, void* base) {
    if (!ctx) return HNH_ERR_INVALID;
    if (!base) return HNH_OK;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return hnh::check_hip(ctx, hipIpcCloseMemHandle(base), "hipIpcCloseMemHandle");
}

int hnh_ipc_pull(hnh_ctx* ctx, int stream, int n, void* const* dst, const void* const* src, const size_t* bytes, int mode, int wgs_per_copy) {
    HNH_ENTER(ctx, stream);
    if (n <= 0) return HNH_OK;
    if (!dst || !src || !bytes || n > HNH_IPC_MAX_PULL) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_ipc_pull: bad argument");
    hipStream_t st = ctx->streams[stream];
    if (mode == HNH_IPC_PULL_KERNEL) {
        PullList list;
        int m = 0;
        for (int i = 0; i < n; i++) {
            if (bytes[i] == 0) continue;
            if (!dst[i] || !src[i]) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_ipc_pull: null pointer");
            list.dst[m] = static_cast<char*>(dst[i]);
            list.src[m] = static_cast<const char*>(src[i]);
            list.bytes[m] = bytes[i];
            m++;
        }
        if (m == 0) return HNH_OK;
        const int wgs = wgs_per_copy < 1 ? 16 : (wgs_per_copy > 256 ? 256 : wgs_per_copy);
        hipLaunchKernelGGL(pull_kernel, dim3((unsigned)(m * wgs)), dim3(256), 0, st, list, wgs);
        return hnh::check_hip(ctx, hipGetLastError(), "pull_kernel");
    }
    if (mode != HNH_IPC_PULL_ENGINE) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_ipc_pull: bad mode");
    int live = 0;
    for (int i = 0; i < n; i++) live += bytes[i] != 0;
    if (live <= 1) {  // nothing to run side by side: stay on the stream
        for (int i = 0; i < n; i++)
            if (bytes[i]) HNH_TRY_HIP(ctx, hipMemcpyAsync(dst[i], src[i], bytes[i], hipMemcpyDeviceToDevice, st));
        return HNH_OK;
    }
    // fork: every auxiliary stream waits for what `stream` has enqueued so far, copies its sources, and `stream` waits for all
    // of them.  The auxiliary streams never hold a cross-process wait themselves (those stay on `stream`).
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    int st_fork = ensure_aux(ctx, 0);
    if (st_fork != HNH_OK) return st_fork;
    HNH_TRY_HIP(ctx, hipEventRecord(ctx->aux_fork, st));
    bool used[hnh_ctx::kAuxStreams] = {false};
    int k = 0;
    for (int i = 0; i < n; i++) {
        if (bytes[i] == 0) continue;
        const int a = k++ % hnh_ctx::kAuxStreams;
        int rc = ensure_aux(ctx, a);
        if (rc != HNH_OK) return rc;
        if (!used[a]) HNH_TRY_HIP(ctx, hipStreamWaitEvent(ctx->aux[a], ctx->aux_fork, 0));
        used[a] = true;
        HNH_TRY_HIP(ctx, hipMemcpyAsync(dst[i], src[i], bytes[i], hipMemcpyDeviceToDevice, ctx->aux[a]));
    }
    for (int a = 0; a < hnh_ctx::kAuxStreams; a++) {
        if (!used[a]) continue;
        HNH_TRY_HIP(ctx, hipEventRecord(ctx->aux_join[a], ctx->aux[a]));
        HNH_TRY_HIP(ctx, hipStreamWaitEvent(st, ctx->aux_join[a], 0));
    }
    return HNH_OK;
}

int hnh_ipc_flags_register(hnh_ctx* ctx, void* host_shm, size_t bytes, void** device_view) {
    if (!ctx || !host_shm || !bytes || !device_view) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    HNH_TRY_HIP(ctx, hipHostRegister(host_shm, bytes, hipHostRegisterMapped));
    *device_view = nullptr;
    return hnh::check_hip(ctx, hipHostGetDevicePointer(device_view, host_shm, 0), "hipHostGetDevicePointer");
}

int hnh_ipc_flags_unregister(hnh_ctx* ctx, void* host_shm) {
    if (!ctx || !host_shm) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return hnh::check_hip(ctx, hipHostUnregister(host_shm), "hipHostUnregister");
}

int hnh_stream_write_flag(hnh_ctx* ctx, int stream, void* flag_device, uint64_t value) {
    HNH_ENTER(ctx, stream);
    if (!flag_device) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_stream_write_flag: null flag");
    if (flag_kernels(ctx)) {
        hipLaunchKernelGGL(flag_write_kernel, dim3(1), dim3(1), 0, ctx->streams[stream], static_cast<unsigned long long*>(flag_device), (unsigned long long)value);
        return hnh::check_hip(ctx, hipGetLastError(), "flag_write_kernel");
    }
    return hnh::check_hip(ctx, hipStreamWriteValue64(ctx->streams[stream], flag_device, value, 0), "hipStreamWriteValue64");
}

int hnh_stream_wait_flag(hnh_ctx* ctx, int stream, void* flag_device, uint64_t value) {
    HNH_ENTER(ctx, stream);
    if (!flag_device) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_stream_wait_flag: null flag");
    if (flag_kernels(ctx)) {
        hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(1), 0, ctx->streams[stream], static_cast<const unsigned long long*>(flag_device), (unsigned long long)value);
        return hnh::check_hip(ctx, hipGetLastError(), "flag_wait_kernel");
    }
    return hnh::check_hip(ctx, hipStreamWaitValue64(ctx->streams[stream], flag_device, value, hipStreamWaitValueGte, ~0ull), "hipStreamWaitValue64");
}

}  // extern "C"
