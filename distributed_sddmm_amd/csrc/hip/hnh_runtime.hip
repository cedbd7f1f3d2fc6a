// Context / memory / stream / event entry points of include/hnh_kernels.h (gfx950, HIP runtime only).
#include <hip/hip_runtime.h>
#include <new>
#include <cstdlib>

#include "hnh_ctx.hpp"
#include "hnh_measurement_aids.h"

namespace {
// one lane spins on the constant-rate (100 MHz) clock; s_sleep keeps it off the issue ports
__global__ void delay_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
// a transfer of modelled duration that INCLUDES whatever real work sits between the two marks: stamp the clock, ..., hold until
// `ticks` have passed since the stamp
__global__ void stamp_kernel(unsigned long long* t) { *t = wall_clock64(); }
__global__ void hold_until_kernel(const unsigned long long* t, unsigned long long ticks) {
    const unsigned long long t0 = *t;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
// slice s = blockIdx.x / wgs copies its share of src -> dst + s * slice_bytes in tiles, never ahead of the modelled rate
__global__ __launch_bounds__(256) void paced_copy_kernel(char* dst, const char* src, size_t slice_bytes, int wgs, unsigned long long ticks) {
    const int slice = (int)blockIdx.x / wgs, part = (int)blockIdx.x % wgs;
    const size_t vecs = slice_bytes / 16, per = (vecs + (size_t)wgs - 1) / (size_t)wgs;
    const size_t lo = per * (size_t)part, hi = lo + per < vecs ? lo + per : vecs;
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(dst + (size_t)slice * slice_bytes);
    constexpr size_t kTile = 256 * 8;  // 32 KiB per workgroup and step
    const size_t tiles = hi > lo ? (hi - lo + kTile - 1) / kTile : 0;
    const unsigned long long t0 = wall_clock64();
    for (size_t t = 0; t < tiles; t++) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const size_t i = lo + t * kTile + (size_t)u * 256 + threadIdx.x;
            if (i < hi) v[u] = s[i];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const size_t i = lo + t * kTile + (size_t)u * 256 + threadIdx.x;
            if (i < hi) d[i] = v[u];
        }
        const unsigned long long due = ticks * (t + 1) / tiles;
        while (wall_clock64() - t0 < due) __builtin_amdgcn_s_sleep(16);
    }
}
}  // namespace

extern "C" {

int hnh_stream_paced_copy(hnh_ctx* ctx, int stream, void* dst_base, const void* src, size_t slice_bytes, int nslices, double microseconds,
                          int wgs_per_slice) {
    HNH_ENTER(ctx, stream);
    if (slice_bytes == 0 || nslices <= 0) return HNH_OK;
    if (!dst_base || !src || slice_bytes % 16 != 0 || wgs_per_slice < 1 || wgs_per_slice > 64 || nslices > 64 || !(microseconds >= 0.0) ||
        microseconds > 5.0e6)
        return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_stream_paced_copy: bad argument");
    hipLaunchKernelGGL(paced_copy_kernel, dim3((unsigned)(nslices * wgs_per_slice)), dim3(256), 0, ctx->streams[stream], static_cast<char*>(dst_base),
                       static_cast<const char*>(src), slice_bytes, wgs_per_slice, (unsigned long long)(microseconds * 100.0));
    return hnh::check_hip(ctx, hipGetLastError(), "paced_copy_kernel");
}

int hnh_stream_pace_begin(hnh_ctx* ctx, int stream) {
    HNH_ENTER(ctx, stream);
    if (!ctx->pace_stamp[stream]) HNH_TRY_HIP(ctx, hipMalloc((void**)&ctx->pace_stamp[stream], sizeof(unsigned long long)));
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ctx->streams[stream], ctx->pace_stamp[stream]);
    return hnh::check_hip(ctx, hipGetLastError(), "stamp_kernel");
}

int hnh_stream_pace_end(hnh_ctx* ctx, int stream, double microseconds) {
    HNH_ENTER(ctx, stream);
    if (!ctx->pace_stamp[stream]) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_stream_pace_end without hnh_stream_pace_begin");
    if (!(microseconds > 0.0)) return HNH_OK;
    if (microseconds > 5.0e6) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_stream_pace_end: more than 5 s");
    hipLaunchKernelGGL(hold_until_kernel, dim3(1), dim3(1), 0, ctx->streams[stream], ctx->pace_stamp[stream], (unsigned long long)(microseconds * 100.0));
    return hnh::check_hip(ctx, hipGetLastError(), "hold_until_kernel");
}

int hnh_stream_delay_us(hnh_ctx* ctx, int stream, double microseconds) {
    HNH_ENTER(ctx, stream);
    if (!(microseconds > 0.0)) return HNH_OK;
    if (microseconds > 5.0e6) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_stream_delay_us: more than 5 s");
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(1), 0, ctx->streams[stream], (unsigned long long)(microseconds * 100.0));
    return hnh::check_hip(ctx, hipGetLastError(), "delay_kernel");
}

const char* hnh_backend_name(void) { return "hip-gfx950"; }

int hnh_ctx_device_identity(hnh_ctx* ctx, int* ordinal, char* pci_bus_id, int len) {
    if (!ctx || !ordinal || !pci_bus_id || len < 16) return HNH_ERR_INVALID;
    *ordinal = ctx->device;
    pci_bus_id[0] = 0;
    return hnh::check_hip(ctx, hipDeviceGetPCIBusId(pci_bus_id, len, ctx->device), "hipDeviceGetPCIBusId");
}

int hnh_ctx_create(int device, hnh_ctx** out) {
    if (!out) return HNH_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return HNH_ERR_DEVICE;  // no GPU: fail loudly
    if (device < 0 || device >= count) return HNH_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return HNH_ERR_DEVICE;
    hnh_ctx* ctx = new (std::nothrow) hnh_ctx();
    if (!ctx) return HNH_ERR_NOMEM;
    ctx->device = device;
    ctx->no_panels = std::getenv("HNH_NO_PANELS") != nullptr;
    ctx->hub_atomics = std::getenv("HNH_HUB_ATOMICS") != nullptr;
    if (const char* hs = std::getenv("HNH_HUB_SCRATCH_MB")) {
        const double mb = std::atof(hs);
        if (mb >= 0.0 && mb <= 262144.0) ctx->hub_scratch_bytes = (size_t)(mb * 1024.0 * 1024.0);
    }
    if (const char* k = std::getenv("HNH_ROW_WAVES_CAP")) {
        const long v = std::strtol(k, nullptr, 10);
        if (v >= 0 && v <= 7) ctx->row_waves_cap = (int)v;
    }
    if (const char* gw = std::getenv("HNH_GEMM_WAVES")) ctx->gemm_waves = std::atoi(gw) == 8 ? 8 : 4;
    if (const char* ge = std::getenv("HNH_GEMM_LDS_EXTRA")) ctx->gemm_lds_extra = std::max(0, std::min(64 * 1024, std::atoi(ge)));
    ctx->panels_with_hubs = std::getenv("HNH_PANELS_WITH_HUBS") != nullptr;
    if (const char* nr = std::getenv("HNH_NARROW_ROWS")) ctx->narrow_rows = std::atoi(nr) != 0;
    if (const char* lg = std::getenv("HNH_LONG_GRID")) {
        const long v = std::strtol(lg, nullptr, 10);
        if (v >= 64 && v <= 65536) ctx->long_grid = (int)v;
    }
    if (const char* lr = std::getenv("HNH_LONG_ROW")) {
        const long v = std::strtol(lr, nullptr, 10);
        if (v >= 64 && v <= 1984) ctx->long_row_override = (int)(v / 64 * 64);
    }
    if (const char* mp = std::getenv("HNH_MAX_PANELS")) {
        const int v = std::atoi(mp);
        if (v >= 1 && v <= 8) ctx->max_panels = v;
    }
    if (const char* ws = std::getenv("HNH_WIDE_SLABS")) ctx->wide_slabs = std::atoi(ws) != 0;
    if (const char* sm = std::getenv("HNH_SLAB_MIN_R")) {
        const int v = std::atoi(sm);
        if (v >= 128 && v <= 4096) ctx->slab_min_r = v;
    }
    if (const char* pb = std::getenv("HNH_PANEL_BYTES")) {
        const double v = std::atof(pb);
        if (v >= 1.0) ctx->panel_bytes = v;
    }
    // HNH_COMM_PRIORITY=1: the communication stream gets the highest priority the device offers.  Opt-in: on one GPU shared by 8
    // logical ranks it makes no measurable difference (profiles/archive/r02_loopback_p8_comm_priority.log).  A masked stream (below) is
    // created by hipExtStreamCreateWithCUMask, which takes neither a priority nor hipStreamNonBlocking: the mask supersedes both.
    int least = 0, greatest = 0;
    const bool has_priorities = std::getenv("HNH_COMM_PRIORITY") != nullptr &&
                                hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least;
    // Compute units of the compute stream (HNH_COMM_CUS=<n>, default 0 = no mask).  The row kernels are bound by the memory side,
    // not by CUs: with 16 of the 256 CUs masked off their stream the fused pass and SpMM run 0.3-2 % FASTER depending on the box
    // (fewer requesters queueing at the fabric; profiles/archive/r03_kbench_cus_off_x_waves_cap.log, r04_job1_bench_masked_default.json),
    // while arithmetic-heavier launches lose — the stand-alone SDDMM 5 %, the fp64 GEMM its share of the matrix cores.  With a
    // mask the library therefore decides per operation: streams[HNH_STREAM_COMPUTE] is masked and the entry points that want the
    // whole chip (hnh_sddmm_*, hnh_gemm_f64) fork onto `wide`, an unmasked stream, and join back (hnh::WideLaunch) — callers keep
    // seeing ONE compute stream.  The masked-off CUs are where a communication stream's work (RCCL channels, pull kernels) finds
    // free slots at once; HNH_COMM_CUS_EXCLUSIVE=1 additionally confines the communication stream to them.
    // NOT the default, for robustness: a process that had created a CU-masked stream hung in the HIP runtime's exit handler
    // (a queue-teardown ioctl) in 4 of 80 runs when a large spinning OpenMP pool was alive at exit — 0 of 80 without the mask, with
    // 8 OpenMP threads or with OMP_WAIT_POLICY=passive (profiles/r04_masked_stream_exit_hang.log).  One percent is not worth that.
    int comm_cus = 0;
    if (const char* cc = std::getenv("HNH_COMM_CUS")) comm_cus = std::atoi(cc);
    static const bool comm_exclusive = std::getenv("HNH_COMM_CUS_EXCLUSIVE") != nullptr;
    hipDeviceProp_t prop;
    uint32_t mask_compute[16] = {0}, mask_comm[16] = {0};
    int mask_words = 0, placed = 0;
    if (comm_cus > 0 && hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount >= 2 * comm_cus &&
        prop.multiProcessorCount <= 512) {
        const int cus = prop.multiProcessorCount;
        mask_words = (cus + 31) / 32;
        // the reserved CUs are spread over the mask with a stride coprime to the CU count (so the walk visits every CU before it
        // repeats), which lands them evenly on the XCDs whether the mask's bits enumerate the XCDs interleaved or block by block
        auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
        int stride = cus / comm_cus + 1;
        while (gcd(stride, cus) != 1) stride++;
        for (int k = 0; placed < comm_cus && k < cus; k++) {
            const int i = (int)(((long)k * stride) % cus);
            mask_comm[i / 32] |= 1u << (i % 32);
            placed++;
        }
        for (int i = 0; i < cus; i++)
            if (!(mask_comm[i / 32] & (1u << (i % 32)))) mask_compute[i / 32] |= 1u << (i % 32);
    }
    // HNH_AUX_PRIORITY=high|low: priority of HNH_STREAM_AUX (the GAT pipeline's dense products beside the attention passes of the
    // compute stream); default: the same priority as the compute stream
    int aux_priority = 0;
    bool aux_has_priority = false;
    if (const char* ap = std::getenv("HNH_AUX_PRIORITY")) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
            aux_has_priority = true;
            aux_priority = (ap[0] == 'h') ? hi : lo;
        }
    }
    auto plain_stream = [&](hipStream_t* st, bool comm) {
        hipError_t e = hipErrorUnknown;
        if (comm && has_priorities) e = hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest);
        if (!comm && aux_has_priority && st == &ctx->streams[HNH_STREAM_AUX]) e = hipStreamCreateWithPriority(st, hipStreamNonBlocking, aux_priority);
        if (e != hipSuccess) e = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
        return e;
    };
    bool masked = false;
    if (mask_words > 0) {
        masked = hipExtStreamCreateWithCUMask(&ctx->streams[HNH_STREAM_COMPUTE], (uint32_t)mask_words, mask_compute) == hipSuccess;
        if (masked) {
            hipError_t e = comm_exclusive ? hipExtStreamCreateWithCUMask(&ctx->streams[HNH_STREAM_COMM], (uint32_t)mask_words, mask_comm)
                                          : plain_stream(&ctx->streams[HNH_STREAM_COMM], true);
            if (e == hipSuccess) e = plain_stream(&ctx->streams[HNH_STREAM_AUX], false);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->wide, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->wide_fork, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->wide_join, hipEventDisableTiming);
            if (e != hipSuccess) {  // half a set-up is no set-up: start again without masks
                for (hipStream_t* st : {&ctx->streams[0], &ctx->streams[1], &ctx->streams[2], &ctx->wide})
                    if (*st) { (void)hipStreamDestroy(*st); *st = nullptr; }
                if (ctx->wide_fork) { (void)hipEventDestroy(ctx->wide_fork); ctx->wide_fork = nullptr; }
                if (ctx->wide_join) { (void)hipEventDestroy(ctx->wide_join); ctx->wide_join = nullptr; }
                masked = false;
            }
        }
    }
    if (masked) {
        ctx->comm_cus = placed;
    } else {
        for (int s = 0; s < HNH_STREAMS; s++)
            if (plain_stream(&ctx->streams[s], s == HNH_STREAM_COMM) != hipSuccess) {
                delete ctx;
                return HNH_ERR_DEVICE;
            }
    }
    (void)hipGetLastError();  // (a refused priority request above is not this call's error)
    *out = ctx;
    return HNH_OK;
}

int hnh_ctx_destroy(hnh_ctx* ctx) {
    if (!ctx) return HNH_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    for (int s = 0; s < HNH_STREAMS; s++) {
        if (ctx->streams[s]) { (void)hipStreamSynchronize(ctx->streams[s]); (void)hipStreamDestroy(ctx->streams[s]); }
        if (ctx->long_items[s]) (void)hipFree(ctx->long_items[s]);
        if (ctx->long_count[s]) (void)hipFree(ctx->long_count[s]);
        if (ctx->long_rows[s]) (void)hipFree(ctx->long_rows[s]);
        if (ctx->long_partials[s]) (void)hipFree(ctx->long_partials[s]);
        if (ctx->panel_split[s]) (void)hipFree(ctx->panel_split[s]);
        if (ctx->pace_stamp[s]) (void)hipFree(ctx->pace_stamp[s]);
    }
    for (int a = 0; a < hnh_ctx::kAuxStreams; a++) {
        if (ctx->aux[a]) { (void)hipStreamSynchronize(ctx->aux[a]); (void)hipStreamDestroy(ctx->aux[a]); }
        if (ctx->aux_join[a]) (void)hipEventDestroy(ctx->aux_join[a]);
    }
    if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
    if (ctx->wide) { (void)hipStreamSynchronize(ctx->wide); (void)hipStreamDestroy(ctx->wide); }
    if (ctx->wide_fork) (void)hipEventDestroy(ctx->wide_fork);
    if (ctx->wide_join) (void)hipEventDestroy(ctx->wide_join);
    delete ctx;
    return HNH_OK;
}

const char* hnh_last_error(hnh_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

void* hnh_ctx_stream(hnh_ctx* ctx, int stream) {
    if (!ctx || !hnh::valid_stream(stream)) return nullptr;
    return (void*)ctx->streams[stream];
}

int hnh_malloc(hnh_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) return hnh::fail(ctx, HNH_ERR_NOMEM, "hipMalloc: out of memory");
    return hnh::check_hip(ctx, e, "hipMalloc");
}

int hnh_free(hnh_ctx* ctx, void* ptr) {
    if (!ctx) return HNH_ERR_INVALID;
    if (!ptr) return HNH_OK;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return hnh::check_hip(ctx, hipFree(ptr), "hipFree");
}

int hnh_memcpy(hnh_ctx* ctx, void* dst, const void* src, size_t bytes, int kind, int stream) {
    HNH_ENTER(ctx, stream);
    if (bytes == 0) return HNH_OK;
    if (!dst || !src) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_memcpy: null pointer");
    hipMemcpyKind k;
    switch (kind) {
        case HNH_COPY_H2D: k = hipMemcpyHostToDevice; break;
        case HNH_COPY_D2H: k = hipMemcpyDeviceToHost; break;
        case HNH_COPY_D2D: k = hipMemcpyDeviceToDevice; break;
        default: return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_memcpy: bad kind");
    }
    return hnh::check_hip(ctx, hipMemcpyAsync(dst, src, bytes, k, ctx->streams[stream]), "hipMemcpyAsync");
}

int hnh_memset(hnh_ctx* ctx, void* dst, int byte, size_t bytes, int stream) {
    HNH_ENTER(ctx, stream);
    if (bytes == 0) return HNH_OK;
    if (!dst) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_memset: null pointer");
    return hnh::check_hip(ctx, hipMemsetAsync(dst, byte, bytes, ctx->streams[stream]), "hipMemsetAsync");
}

int hnh_stream_sync(hnh_ctx* ctx, int stream) {
    HNH_ENTER(ctx, stream);
    return hnh::check_hip(ctx, hipStreamSynchronize(ctx->streams[stream]), "hipStreamSynchronize");
}

int hnh_event_create(hnh_ctx* ctx, void** event) {
    if (!ctx || !event) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t ev;
    HNH_TRY_HIP(ctx, hipEventCreate(&ev));  // timing enabled: bench.py measures kernels with these
    *event = (void*)ev;
    return HNH_OK;
}

int hnh_event_destroy(hnh_ctx* ctx, void* event) {
    if (!ctx) return HNH_ERR_INVALID;
    if (!event) return HNH_OK;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return hnh::check_hip(ctx, hipEventDestroy((hipEvent_t)event), "hipEventDestroy");
}

int hnh_event_record(hnh_ctx* ctx, void* event, int stream) {
    HNH_ENTER(ctx, stream);
    if (!event) return hnh::fail(ctx, HNH_ERR_INVALID, "null event");
    return hnh::check_hip(ctx, hipEventRecord((hipEvent_t)event, ctx->streams[stream]), "hipEventRecord");
}

int hnh_event_wait(hnh_ctx* ctx, void* event, int stream) {
    HNH_ENTER(ctx, stream);
    if (!event) return hnh::fail(ctx, HNH_ERR_INVALID, "null event");
    return hnh::check_hip(ctx, hipStreamWaitEvent(ctx->streams[stream], (hipEvent_t)event, 0), "hipStreamWaitEvent");
}

int hnh_event_sync(hnh_ctx* ctx, void* event) {
    if (!ctx || !event) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return hnh::check_hip(ctx, hipEventSynchronize((hipEvent_t)event), "hipEventSynchronize");
}

int hnh_event_query(hnh_ctx* ctx, void* event, int* done) {
    if (!ctx || !event || !done) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    const hipError_t e = hipEventQuery((hipEvent_t)event);
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        *done = 0;
        return HNH_OK;
    }
    *done = 1;
    return hnh::check_hip(ctx, e, "hipEventQuery");
}

int hnh_event_elapsed_ms(hnh_ctx* ctx, void* start, void* stop, float* ms) {
    if (!ctx || !start || !stop || !ms) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return hnh::check_hip(ctx, hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop), "hipEventElapsedTime");
}

}  // extern "C"
