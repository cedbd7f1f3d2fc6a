// Internal: the per-rank context behind the opaque `hnh_ctx` of include/hnh_kernels.h.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "hnh_kernels.h"

struct hnh_ctx {
    int device = 0;
    hipStream_t streams[HNH_STREAMS] = {nullptr};  // [HNH_STREAM_COMPUTE], [HNH_STREAM_COMM], [HNH_STREAM_AUX]
    std::string last_error;
    // work list of the long-row pass (one per stream): items = (row, segment), count lives on the device
    void* long_items[HNH_STREAMS] = {nullptr};
    int* long_count[HNH_STREAMS] = {nullptr};  // two ints: number of items, number of hub rows
    size_t long_cap[HNH_STREAMS] = {0};
    // the hub rows themselves, (row, first item, segments), and the segments' partial output rows (items x row pitch doubles)
    void* long_rows[HNH_STREAMS] = {nullptr};
    size_t long_rows_cap[HNH_STREAMS] = {0};
    void* long_partials[HNH_STREAMS] = {nullptr};
    size_t long_partials_bytes[HNH_STREAMS] = {0};
    size_t hub_scratch_bytes = (size_t)2 << 30;  // HNH_HUB_SCRATCH_MB: bound of long_partials per stream; segments beyond it use atomics
    bool hub_atomics = false;  // HNH_HUB_ATOMICS=1: combine hub-row segments with fp64 atomics (round 1's way) instead of the ordered reduction
    // per-row panel boundaries of the Infinity-Cache panels (one per stream): (panels - 1) x rows int32
    void* panel_split[HNH_STREAMS] = {nullptr};
    size_t panel_cap[HNH_STREAMS] = {0};
    bool no_panels = false;  // HNH_NO_PANELS=1: A/B switch
    int row_waves_cap = -1;  // HNH_ROW_WAVES_CAP=k: 1..7 = at most k waves per SIMD for every row-kernel launch, 0 = never cap; unset = the
                             // library's rule (uniform blocks of certain widths run at 5, see row_occupancy_pad in hnh_kernels.hip)
    int long_grid = 1024;           // workgroups of the hub-row segment pass (HNH_LONG_GRID, measurement aid)
    int comm_cus = 0;               // compute units masked off streams[HNH_STREAM_COMPUTE] (HNH_COMM_CUS; default 0 = no mask, see hnh_runtime.hip)
    // the unmasked twin of the compute stream for launches that want every CU (hnh::WideLaunch); null when nothing is masked
    hipStream_t wide = nullptr;
    hipEvent_t wide_fork = nullptr, wide_join = nullptr;
    bool narrow_rows = true;        // HNH_NARROW_ROWS=0: A/B switch — R = 8 / 16 / 32 through the general row loop instead of the line-granular one
    bool panels_with_hubs = false;  // HNH_PANELS_WITH_HUBS=1: panel the short rows of blocks that also have hub rows
    int long_row_override = 0;      // HNH_LONG_ROW=<multiple of 64, 64..1984>: fixed hub-row threshold instead of the adaptive one (measurement aid)
    double panel_bytes = 512.0 * 1024.0 * 1024.0;  // bytes of the gathered operand per panel (HNH_PANEL_BYTES; tests shrink it)
    // Most panels a pass is cut into (HNH_MAX_PANELS, 1 .. 8).  Every panel re-reads the row operand and read-modify-writes the output
    // (3 dense rows per sparse row and panel: counter traffic 1.03 / 1.09 / 1.21 x the byte model at 2 / 4 / 8 panels), which the
    // Infinity Cache has to earn back.  Where the balance tips depends on the BOX: R = 512 (8 x 512 MiB) on three leases — 4 panels
    // 66.7 ms against 8 panels 70.2 ms; 6 panels 71.8 against 4 panels 73.2; 8 panels 73.5, 6 panels 74.3, 4 panels 76.5 — the same
    // launch sequence differs by more between boxes than between panel counts (profiles/r05_wide_panels*.log).  Six is never more than
    // 1-2 % from the best of any of them; R <= 384 is not affected (its natural count is at most six).
    int max_panels = 6;
    bool wide_slabs = true;  // un-fused SDDMM / SpMM of R >= slab_min_r (a multiple of 128) run as 128-column slabs (HNH_WIDE_SLABS=0: one wide pass)
    int slab_min_r = 320;    // HNH_SLAB_MIN_R (a multiple of 64 from here on is cut into slabs; 256 keeps its four panels: 93.3 vs 92.8 % SDDMM, 89.6 vs 93.3 % SpMM)
    // peer-to-peer pull (hnh_ipc.hip): auxiliary streams the copy-engine pulls of one group are spread over (created on first use),
    // the fork event recorded on the issuing stream and one join event per auxiliary stream
    static constexpr int kAuxStreams = 8;
    hipStream_t aux[kAuxStreams] = {nullptr};
    hipEvent_t aux_fork = nullptr, aux_join[kAuxStreams] = {nullptr};
    unsigned long long* pace_stamp[HNH_STREAMS] = {nullptr};  // measurement aid (hnh_stream_pace_begin / _end): the clock at the begin mark
    int gemm_waves = 4;     // HNH_GEMM_WAVES=8: hnh_gemm_f64 with 256 x 128 tiles of 8 waves instead of 128 x 128 tiles of 4
    int gemm_lds_extra = 0; // HNH_GEMM_LDS_EXTRA=<bytes>: unused dynamic LDS added to every GEMM workgroup's request (occupancy experiments)
    int flag_kernels = -1;  // HNH_IPC_FLAGS=kernel: flag words are written / awaited by one-lane kernels instead of stream memory operations
};

namespace hnh {

inline int fail(hnh_ctx* ctx, int code, const std::string& what) {
    if (ctx) ctx->last_error = what;
    return code;
}

inline int check_hip(hnh_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return HNH_OK;
    return fail(ctx, HNH_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

inline bool valid_stream(int s) { return s >= 0 && s < HNH_STREAMS; }

// Scope of an operation that runs on ALL compute units although the compute stream is masked: the constructor makes the wide
// stream wait for what the compute stream has enqueued so far, `stream()` is where the operation's launches go, finish()
// makes the compute stream wait for them.  Everywhere else (no mask, or the communication stream) it is the stream itself.
struct WideLaunch {
    hnh_ctx* ctx;
    hipStream_t st;
    bool forked = false;
    int status = HNH_OK;
    WideLaunch(hnh_ctx* c, int sidx) : ctx(c), st(c->streams[sidx]) {
        if (sidx != HNH_STREAM_COMPUTE || c->wide == nullptr) return;
        status = check_hip(c, hipEventRecord(c->wide_fork, st), "hipEventRecord");
        if (status == HNH_OK) status = check_hip(c, hipStreamWaitEvent(c->wide, c->wide_fork, 0), "hipStreamWaitEvent");
        forked = status == HNH_OK;
    }
    hipStream_t stream() const { return forked ? ctx->wide : st; }
    int finish(int rc) {
        if (!forked) return rc;
        forked = false;
        int j = check_hip(ctx, hipEventRecord(ctx->wide_join, ctx->wide), "hipEventRecord");
        if (j == HNH_OK) j = check_hip(ctx, hipStreamWaitEvent(st, ctx->wide_join, 0), "hipStreamWaitEvent");
        return rc != HNH_OK ? rc : j;
    }
};

}  // namespace hnh

#define HNH_TRY_HIP(ctx, expr)                                   \
    do {                                                         \
        int _st = hnh::check_hip((ctx), (expr), #expr);          \
        if (_st != HNH_OK) return _st;                           \
    } while (0)

#define HNH_ENTER(ctx, stream)                                                        \
    if (!(ctx)) return HNH_ERR_INVALID;                                               \
    if (!hnh::valid_stream(stream)) return hnh::fail((ctx), HNH_ERR_INVALID, "bad stream selector"); \
    HNH_TRY_HIP((ctx), hipSetDevice((ctx)->device))
