// Hand-written CDNA4 (gfx950) local kernels behind include/hnh_kernels.h.
//
// What they replace (reference = PASSIONLab/distributed_sddmm, CPU only):
//   sddmm_*  : StandardKernel::sddmm_local, OpenMP loop over COO nonzeros  (sparse_kernels.cpp:13-57)
//   spmm_csr : StandardKernel::spmm_local -> mkl_sparse_d_mm(alpha=1,beta=1) (sparse_kernels.cpp:59-127)
//   fused    : the back-to-back sddmm/spmm pair of the "local kernel fusion" schedule
//              (15D_dense_shift.hpp:203-217) as one pass with a single gather of each dense row.
//
// Design (MI355X first):
//   * The path is HBM/gather bound (0.49 flop/B), so no MFMA: the work is laid out so that every
//     vector-memory instruction of a wave reads whole, contiguous dense rows.
//   * "Group per sparse row": a group of LPR lanes (power of two, <= 64) owns one CSR row; lane l of the
//     group holds elements (v*LPR + l)*W .. +W of a dense row for v < VEC, W = 2 doubles (one 16-byte
//     dwordx4 load).  For R = 128 that is LPR = 64, VEC = 1: one wave per row, one 1 KiB fully coalesced
//     load instruction per nonzero.  For R = 16 it is LPR = 8: eight rows per wave.
//   * The row operand X[i,:] and the output accumulator live in registers for the whole sparse row (the
//     register file is the "LDS tile": nothing is shared between groups, so staging through LDS would
//     only add a round trip).  The output row is written once, without atomics.
//   * U nonzeros are in flight per group (U independent 16-byte loads per lane issued before the first
//     use), and their U dot products are reduced together with a transposed butterfly: (U-1) + log2(LPR/U)
//     cross-lane exchanges instead of U*log2(LPR).
//   * With LPR = 64 the row index is wave-uniform (readfirstlane), so rowptr / col_idx / svalues go
//     through the scalar cache and the vector memory pipe carries only dense rows and `values`.
//   * Launch: rows/(256/LPR) workgroups of 256 threads (cfg2: 262 144 workgroups >> 256 CUs); consecutive
//     workgroups own consecutive rows, so the CSR index stream is read in order.
#include <hip/hip_runtime.h>
#include <new>
#include <cstdint>
#include "hnh_ctx.hpp"

namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------- small device helpers

template <int W>
__device__ __forceinline__ void load_w(double (&dst)[W], const double* __restrict__ p) {
    if constexpr (W == 2) {
        const double2 t = *reinterpret_cast<const double2*>(p);
        dst[0] = t.x;
        dst[1] = t.y;
    } else {
        dst[0] = *p;
    }
}

// Load W doubles from global memory at (64-bit base address) + (32-bit byte offset).  The explicit global address space
// keeps the access a global_load (an integer-to-generic-pointer cast would turn it into a flat_load, which also ties up
// the LDS counter), and the base + zero-extended-offset form lets a wave-uniform base stay in scalar registers.
typedef const __attribute__((address_space(1))) char* gbytes_t;
template <int W>
__device__ __forceinline__ void load_w_global(double (&dst)[W], uint64_t base, unsigned byte_off) {
    gbytes_t p = reinterpret_cast<gbytes_t>(base) + byte_off;
    if constexpr (W == 2) {
        typedef double dv2 __attribute__((ext_vector_type(2)));
        const dv2 t = *reinterpret_cast<const __attribute__((address_space(1))) dv2*>(p);
        dst[0] = t.x;
        dst[1] = t.y;
    } else {
        dst[0] = *reinterpret_cast<const __attribute__((address_space(1))) double*>(p);
    }
}

template <int W>
__device__ __forceinline__ void store_w(double* __restrict__ p, const double (&src)[W]) {
    if constexpr (W == 2) {
        *reinterpret_cast<double2*>(p) = make_double2(src[0], src[1]);
    } else {
        *p = src[0];
    }
}

// Streaming (use-once) data — the row operand, the output row, the per-nonzero values — can be marked non-temporal so
// that it does not displace the gathered operand's rows from L2 / the Infinity Cache (tuning knob HNH_NT_STREAM).
#ifndef HNH_NT_STREAM
#define HNH_NT_STREAM 0
#endif
typedef double d2_t __attribute__((ext_vector_type(2)));
template <int W>
__device__ __forceinline__ void load_w_stream(double (&dst)[W], const double* __restrict__ p) {
#if HNH_NT_STREAM
    if constexpr (W == 2) {
        const d2_t t = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(p));
        dst[0] = t.x;
        dst[1] = t.y;
    } else {
        dst[0] = __builtin_nontemporal_load(p);
    }
#else
    load_w<W>(dst, p);
#endif
}
template <int W>
__device__ __forceinline__ void store_w_stream(double* __restrict__ p, const double (&src)[W]) {
#if HNH_NT_STREAM
    if constexpr (W == 2) {
        d2_t t; t.x = src[0]; t.y = src[1];
        __builtin_nontemporal_store(t, reinterpret_cast<d2_t*>(p));
    } else {
        __builtin_nontemporal_store(src[0], p);
    }
#else
    store_w<W>(p, src);
#endif
}
__device__ __forceinline__ double load_stream(const double* p) {
#if HNH_NT_STREAM
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void store_stream(double* p, double v) {
#if HNH_NT_STREAM
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }

// Broadcast the value held by lane `src` of each LPR-lane group to the whole group.
template <int LPR>
__device__ __forceinline__ double group_bcast(double v, int src) {
    if constexpr (LPR == 1) {
        return v;
    } else if constexpr (LPR == 64) {
        // wave-wide: result is uniform, keep it in SGPRs so the following FMAs take a scalar operand
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    } else {
        return __shfl(v, src, LPR);
    }
}

// Transposed butterfly: on entry d[0..U) are this lane's partial sums of U independent reductions over
// the LPR lanes of its group; on exit the return value is the complete sum of reduction number
// (lane_in_group / (LPR / U)).  S is the exchange stride of the current level (start at LPR / 2).
template <int U, int S>
__device__ __forceinline__ double multi_reduce(double (&d)[U], int lig) {
    if constexpr (U == 1) {
        double v = d[0];
#pragma unroll
        for (int s = S; s >= 1; s >>= 1) v += shfl_xor_f64(v, s);
        return v;
    } else {
        static_assert(S >= 1, "needs U <= LPR");
        const bool upper = (lig & S) != 0;
        double n[U / 2];
#pragma unroll
        for (int j = 0; j < U / 2; j++) {
            const double send = upper ? d[j] : d[j + U / 2];
            const double keep = upper ? d[j + U / 2] : d[j];
            n[j] = keep + shfl_xor_f64(send, S);
        }
        return multi_reduce<U / 2, S / 2>(n, lig);
    }
}

template <int LPR, int U>
__device__ __forceinline__ double group_multi_reduce(double (&d)[U], int lig) {
    if constexpr (LPR == 1) {
        static_assert(U == 1, "LPR == 1 needs U == 1");
        return d[0];
    } else {
        return multi_reduce<U, LPR / 2>(d, lig);
    }
}

// nonzeros in flight per group
#ifndef HNH_UNROLL1
#define HNH_UNROLL1 8  // nonzeros in flight per group when one 16-byte chunk per lane covers the row (tuning knob)
#endif
template <int LPR, int VEC>
struct Unroll {
    static constexpr int byvec = VEC == 1 ? HNH_UNROLL1 : (VEC == 2 ? 4 : (VEC <= 4 ? 2 : 1));
    static constexpr int value = byvec < LPR ? byvec : LPR;
};

// kFusedCg = kFused whose row epilogue also performs the CG updates of hnh_cg_update or the ReLU delivery of
// hnh_fused_extras::relu_dst; its own instance so that the plain fused kernel's register allocation (and with it its
// occupancy) is not touched by code it never runs
enum class Op { kSddmm, kSpmm, kFused, kFusedCg };
constexpr bool fused_op(Op o) { return o == Op::kFused || o == Op::kFusedCg; }

// ---------------------------------------------------------------- the row kernel (sddmm / spmm / fused)
//
// Columns handled: [col0, col0 + ncols) of rows of length `ld`; EXACT means ncols == W*LPR*VEC.
//
// Long rows.  Real graphs have hub rows (an R-MAT stand-in at 2^20 vertices has rows of > 80 000 nonzeros);
// a single group walking such a row serialises the whole launch behind it (measured: 27 % of the roofline
// instead of 81 %).  Rows longer than a threshold are therefore skipped by the row kernel and cut into segments
// of kLongSeg nonzeros that are spread over the whole chip by a second, small launch (`long_row_kernel`);
// SDDMM segments are independent; SpMM / fused segments write their partial output rows to scratch and a third launch
// (`reduce_long_kernel`) adds a row's segments up in a fixed order, so results do not depend on which segment finishes
// first (HNH_HUB_ATOMICS=1: hardware fp64 atomics instead, same speed, arrival-order sums).  Callers that know the
// block's longest row (the host layer does) pass it as a hint and short-row matrices never pay for any of this.
// The threshold adapts to the block: 3 x its mean row length, in steps of 64 within [kLongRowMin, kLongRowMax].  On a skewed
// graph (R-MAT 2^20, mean 85) rows of 256..1024 nonzeros walked by one wave each still cost 5 % (tail and imbalance inside
// the CUs; 1024 -> 256: 12.29 -> 11.65 ms at R = 128, 26.1 -> 24.8 ms at R = 256); on a uniform matrix whose mean is that
// long the same 256 would push every row through the segment path and lose the cache panels (Erdos-Renyi with 300 per row:
// 44.0 -> 50.5 ms), hence "relative to the mean" (profiles/archive/r02_kbench_rmat_longrow_threshold.log, r02_kbench_er_ef300_*).
constexpr int kLongRowMin = 256;
constexpr int kLongRowMax = 1024;
constexpr int kLongSeg = 256;
constexpr unsigned kLongRowShift = 16;  // flag bits 16..20 carry threshold / 64 next to kInternalSplitLong
__host__ __device__ constexpr int long_row_of(unsigned flags) { return (int)((flags >> kLongRowShift) & 0x1fu) * 64; }
constexpr unsigned kInternalSplitLong = 0x100u;  // flag bit, never set by callers
constexpr unsigned kInternalEpilogue = 0x200u;   // flag bit: apply Extras::x_scale / rowdot when the output row is stored

// Optional extras of the fused pass (hnh_fused_extras): an activation between the two halves and a row epilogue.
struct Extras {
    double leaky_alpha = 0.0;  // HNH_FUSED_LEAKY_RELU: weight = dot > 0 ? dot : leaky_alpha * dot
    double x_scale = 0.0;      // epilogue: Out[i,:] += x_scale * X[i,:]
    double* rowdot = nullptr;  // epilogue: rowdot[i] = <X[i,:], Out[i,:]>
    // epilogue: the rest of a batched-CG iteration on the finished row (hnh_cg_update; cg_x == nullptr: off).  cg_p is the
    // row operand X itself, writable.
    double* cg_x = nullptr;
    double* cg_r = nullptr;
    double* cg_p = nullptr;
    double* cg_rsold = nullptr;
    double cg_eps = 0.0;
    // epilogue: deliver max(row, 0) to relu_dst[row * relu_ld + column] instead of storing the row to Out (GAT head output)
    double* relu_dst = nullptr;
    int64_t relu_ld = 0;
};

template <int LPR>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
    return v;
}

// Gather pipeline.  HNH_PIPE = 1: a batch of U nonzeros is handled as two half batches whose gathers are issued one
// half ahead, so that while the dot products / cross-lane reductions / axpys of one half run, the other half's loads
// are still in flight (the wave never sits with zero outstanding loads).  Same registers as one batch of U.
#ifndef HNH_PIPE
#define HNH_PIPE 1
#endif

template <bool B>
struct BoolTag { static constexpr bool value = B; };

template <Op OP, int LPR, int VEC, int W, bool EXACT, bool NARROW = false>
__device__ __forceinline__ void process_row(int64_t row, int beg, int end, bool atomic_out, const int32_t* __restrict__ colidx,
                                            double* values, const double* __restrict__ svalues, const double* __restrict__ X,
                                            const double* __restrict__ Y, double* __restrict__ Out, int64_t ld, int col0,
                                            int ncols, unsigned flags, int lig, const Extras& ex, double* part_row = nullptr) {
    constexpr int UFULL = Unroll<LPR, VEC>::value;
    constexpr bool PIPE = (HNH_PIPE != 0) && UFULL >= 4;
    constexpr int U = PIPE ? UFULL / 2 : UFULL;  // nonzeros per (half) batch
    constexpr int SUB = LPR / U;                 // lanes that end up holding the same reduced value
    bool act[VEC];
    int64_t coff[VEC];
    unsigned lane_off[VEC];  // byte offset of this lane's chunk v inside a dense row (from column col0)
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        const int c = (v * LPR + lig) * W;
        act[v] = EXACT ? true : (c < ncols);
        coff[v] = (int64_t)col0 + c;
        lane_off[v] = (unsigned)c * (unsigned)sizeof(double);
    }

    double x[VEC][W];    // SDDMM row operand X[row, :]
    double acc[VEC][W];  // SpMM accumulator Out[row, :]
#pragma unroll
    for (int v = 0; v < VEC; v++) {
#pragma unroll
        for (int w = 0; w < W; w++) { x[v][w] = 0.0; acc[v][w] = 0.0; }
        if (act[v]) {
            if constexpr (OP == Op::kFusedCg) {
                // with the CG updates the row operand IS cg_p, which the epilogue rewrites: read it through that (unrestricted)
                // pointer, so that no access of this launch goes through the __restrict__ alias X of memory it stores to
                const double* xsrc = (ex.cg_x != nullptr) ? ex.cg_p : X;
                load_w_stream<W>(x[v], xsrc + row * ld + coff[v]);
            } else if constexpr (OP != Op::kSpmm) {
                load_w_stream<W>(x[v], X + row * ld + coff[v]);
            }
            if constexpr (OP != Op::kSddmm) {
                if (!atomic_out && !(flags & HNH_FUSED_OUT_OVERWRITE)) load_w_stream<W>(acc[v], Out + row * ld + coff[v]);
            }
        }
    }
    // the gathered operand: byte address of column col0 of its row 0, and its row pitch in bytes
    const uint64_t g_base = reinterpret_cast<uint64_t>(((OP == Op::kSpmm) ? X : Y) + col0);
    const uint64_t ld_bytes = (uint64_t)(unsigned)(ld * (int64_t)sizeof(double));

    // column indices of the U nonzeros starting at e (-1 beyond `end` when the batch is not FULL).  One wave per row:
    // e is wave-uniform and the indices come through the scalar cache; several rows per wave: ONE coalesced load per
    // group (lane u holds index u), handed round with cross-lane moves instead of U vector loads of the same line.
    auto load_idx = [&](auto full, int e, int (&c)[U]) {
        constexpr bool FULL = decltype(full)::value;
        if constexpr (LPR == 64) {
#pragma unroll
            for (int u = 0; u < U; u++) c[u] = (FULL || e + u < end) ? colidx[e + u] : -1;
        } else {
            const int my = e + (lig % U);
            const int cv = (FULL || my < end) ? colidx[my] : -1;
#pragma unroll
            for (int u = 0; u < U; u++) c[u] = __shfl(cv, u, LPR);
        }
    };
    auto gather = [&](auto full, const int (&c)[U], double (&y)[U][VEC][W]) {
        constexpr bool FULL = decltype(full)::value;
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int v = 0; v < VEC; v++) {
#pragma unroll
                for (int w = 0; w < W; w++) y[u][v][w] = 0.0;
                if ((FULL || c[u] >= 0) && act[v]) {
                    // row base (wave-uniform when one wave owns the row: scalar registers) + a 32-bit lane offset
                    uint64_t rowp = g_base + (uint64_t)(unsigned)c[u] * ld_bytes;
                    if constexpr (LPR == 64) {  // keep it in scalar registers: global_load with SGPR base + VGPR offset
                        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)rowp);
                        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(rowp >> 32));
                        rowp = ((uint64_t)hi << 32) | lo;
                    }
                    unsigned off = lane_off[v];
                    // opaque to the optimiser: keeps the zero-extension of the offset next to the load (instruction selection
                    // works per basic block; hoisted out of the loop it would cost a 64-bit VGPR address per gather)
                    if constexpr (LPR == 64) asm volatile("" : "+v"(off));
                    load_w_global<W>(y[u][v], rowp, off);
                }
            }
        }
    };
    // dot products, value update and axpys of the U nonzeros starting at e whose dense rows are in y
    auto compute = [&](auto full, int e, const double (&y)[U][VEC][W]) {
        constexpr bool FULL = decltype(full)::value;
        double wgt;  // weight of nonzero (e + lig / SUB), valid in every lane of its SUB-lane subgroup
        const int mine = e + lig / SUB;
        const bool have = FULL || mine < end;
        if constexpr (OP == Op::kSpmm) {
            wgt = have ? load_stream(values + mine) : 0.0;
            if (svalues != nullptr && have) wgt *= svalues[mine];
        } else {
            double d[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                double s = 0.0;
#pragma unroll
                for (int v = 0; v < VEC; v++)
#pragma unroll
                    for (int w = 0; w < W; w++) s = fma(x[v][w], y[u][v][w], s);
                d[u] = s;
            }
            wgt = group_multi_reduce<LPR, U>(d, lig);
            if (have) {
                const bool overwrite = (OP != Op::kSpmm) && (flags & HNH_FUSED_VALUES_OVERWRITE);
                if constexpr (OP == Op::kSddmm) {  // stand-alone SDDMM with its closing Hadamard folded in: the result is scale .* dots
                    if (svalues != nullptr) wgt *= svalues[mine];
                }
                if (!overwrite) wgt += load_stream(values + mine);
                if (fused_op(OP) && (flags & HNH_FUSED_LEAKY_RELU)) {  // the activated weight is what gets stored
                    if (svalues != nullptr) wgt *= svalues[mine];
                    wgt = wgt > 0.0 ? wgt : ex.leaky_alpha * wgt;
                    if (lig % SUB == 0) store_stream(values + mine, wgt);
                } else {
                    if (lig % SUB == 0) store_stream(values + mine, wgt);
                    if (fused_op(OP) && svalues != nullptr) wgt *= svalues[mine];
                }
            } else {
                wgt = 0.0;
            }
        }
        if constexpr (OP != Op::kSddmm) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double wu = group_bcast<LPR>(wgt, u * SUB);
#pragma unroll
                for (int v = 0; v < VEC; v++)
#pragma unroll
                    for (int w = 0; w < W; w++) acc[v][w] = fma(wu, y[u][v][w], acc[v][w]);
            }
            // pin the accumulation here: left alone, the optimiser sinks the whole axpy chain to the end of the trip
            // (its result is not needed earlier), which keeps every gather buffer alive and doubles the register count
#pragma unroll
            for (int v = 0; v < VEC; v++)
#pragma unroll
                for (int w = 0; w < W; w++) asm volatile("" : "+v"(acc[v][w]));
        }
    };
    const BoolTag<true> kFull;
    const BoolTag<false> kMasked;

    if constexpr (NARROW) {
        // ---- narrow rows (R = 8 / 16 / 32: 16 / 8 / 4 sparse rows share a wave), line-granular CSR streams.
        // What bounds these widths is the number of 128-byte line requests that reach the fabric (about 55 G lines/s, counters in
        // profiles/archive/r03_narrow_R16_pmc_by_kernel.txt): every gathered dense row is one such request, and so is every piece of
        // `colidx` / `values` a group touches — with 256 .. 512 groups per CU each walking its own sparse row, a stream line is
        // long gone from L1 AND L2 (4 MiB per XCD turn over in about 5 us) when the group's next trip comes back for its second
        // half, so the loop above fetches every stream line 2 .. 8 times (+26 % requests at R = 16).  Here a group moves through its
        // row in trips of 16 nonzeros ALIGNED to the 128-byte lines of `values`; it reads the line of 16 values with one group-wide
        // load (each lane 128 / LPR bytes), the 32 column indices of an aligned 128-byte block once per two trips, and writes the
        // 16 results of a trip as one line.  Positions of the first / last trip that lie outside [beg, end) gather row 0 of the
        // dense operand (an L1 hit, no traffic) and are zeroed, so the gathers stay unconditional.  Requires colidx / values to
        // be 128-byte aligned (checked by the dispatcher).
        static_assert(VEC == 1 && W == 2 && EXACT && (LPR == 4 || LPR == 8 || LPR == 16), "narrow instances");
        constexpr int T = 16;           // nonzeros per trip = one line of `values`
        constexpr int UQ = 4;           // nonzeros per quarter batch (reduced together by the transposed butterfly)
        constexpr int SUBQ = LPR / UQ;  // lanes holding the same reduced dot product
        constexpr int IPL = 32 / LPR;   // column indices per lane of a 32-index block (one line of `colidx`)
        constexpr int VPL = T / LPR > 0 ? T / LPR : 1;  // values per lane of a trip's line (LPR = 16: one)
        const bool vals_overwrite = (OP != Op::kSpmm) && (flags & HNH_FUSED_VALUES_OVERWRITE);
        const bool reads_values = (OP == Op::kSpmm) || !vals_overwrite;

        auto load_block = [&](int b, int (&dst)[IPL]) {  // the aligned block of 32 column indices starting at nonzero b
            const int32_t* p = colidx + b + lig * IPL;
            if constexpr (IPL == 8) {
                const int4 t0 = *reinterpret_cast<const int4*>(p), t1 = *reinterpret_cast<const int4*>(p + 4);
                dst[0] = t0.x; dst[1] = t0.y; dst[2] = t0.z; dst[3] = t0.w;
                dst[4] = t1.x; dst[5] = t1.y; dst[6] = t1.z; dst[7] = t1.w;
            } else if constexpr (IPL == 4) {
                const int4 t0 = *reinterpret_cast<const int4*>(p);
                dst[0] = t0.x; dst[1] = t0.y; dst[2] = t0.z; dst[3] = t0.w;
            } else {
                const int2 t0 = *reinterpret_cast<const int2*>(p);
                dst[0] = t0.x; dst[1] = t0.y;
            }
        };
        auto load_line = [&](const double* base, int e0, double (&dst)[VPL]) {  // the aligned line of 16 values starting at nonzero e0
            const double* p = base + e0 + lig * VPL;
            if constexpr (VPL == 4) {
                const double2 t0 = *reinterpret_cast<const double2*>(p), t1 = *reinterpret_cast<const double2*>(p + 2);
                dst[0] = t0.x; dst[1] = t0.y; dst[2] = t1.x; dst[3] = t1.y;
            } else if constexpr (VPL == 2) {
                const double2 t0 = *reinterpret_cast<const double2*>(p);
                dst[0] = t0.x; dst[1] = t0.y;
            } else {
                dst[0] = *p;
            }
        };
        // value of nonzero k (0 .. 15) of the line held in `line`, as seen by every lane of the group (k may differ per lane)
        auto line_value = [&](const double (&line)[VPL], int k) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < VPL; j++) {
                const double t = __shfl(line[j], k / VPL, LPR);
                if (k % VPL == j) r = t;
            }
            return r;
        };

        int e0 = (beg < end) ? (beg & ~(T - 1)) : end;  // (an empty row / window: no trip at all)
        int blk = e0 & ~31;
        int ib[IPL], ibn[IPL];
        double vv[VPL], vvn[VPL], sv[VPL];
#pragma unroll
        for (int j = 0; j < IPL; j++) { ib[j] = 0; ibn[j] = 0; }
#pragma unroll
        for (int j = 0; j < VPL; j++) { vv[j] = 0.0; vvn[j] = 0.0; sv[j] = 1.0; }
        if (e0 < end) {
            load_block(blk, ib);
            if (blk + 32 < end) load_block(blk + 32, ibn);
            if (reads_values) load_line(values, e0, vv);
        }
        for (; e0 < end; e0 += T) {
            if (e0 == blk + 32) {  // second trip of the block done: the prefetched block becomes current, the one after it is requested
                blk += 32;
#pragma unroll
                for (int j = 0; j < IPL; j++) ib[j] = ibn[j];
                if (blk + 32 < end) load_block(blk + 32, ibn);
            }
            if (reads_values && e0 + T < end) load_line(values, e0 + T, vvn);  // next trip's values travel with this trip's gathers
            if (svalues != nullptr) load_line(svalues, e0, sv);
            const bool part = (e0 < beg) || (e0 + T > end);
            const int half = (e0 - blk) / IPL;  // first lane holding this trip's half of the index block
            int c[T];
#pragma unroll
            for (int k = 0; k < T; k++) {
                c[k] = __shfl(ib[k % IPL], half + k / IPL, LPR);
                if (part && !(e0 + k >= beg && e0 + k < end)) c[k] = 0;
            }
            double ya[UQ][1][W], yb[UQ][1][W];
            double wq[4];  // per quarter: the value to store for nonzero 4 q + lig / SUBQ
            auto gatherq = [&](int q, double (&y)[UQ][1][W]) {
#pragma unroll
                for (int u = 0; u < UQ; u++) {
                    const uint64_t rowp = g_base + (uint64_t)(unsigned)c[4 * q + u] * ld_bytes;
                    load_w_global<W>(y[u][0], rowp, lane_off[0]);
                }
            };
            auto computeq = [&](int q, double (&y)[UQ][1][W]) {
                if (part) {
#pragma unroll
                    for (int u = 0; u < UQ; u++) {
                        const int pos = e0 + 4 * q + u;
                        if (!(pos >= beg && pos < end)) { y[u][0][0] = 0.0; y[u][0][1] = 0.0; }
                    }
                }
                const int kmine = 4 * q + lig / SUBQ;  // the nonzero whose reduced value this lane ends up holding
                const bool have = !part || (e0 + kmine >= beg && e0 + kmine < end);
                double wgt;
                if constexpr (OP == Op::kSpmm) {
                    wgt = line_value(vv, kmine);
                    if (svalues != nullptr) wgt *= line_value(sv, kmine);
                    if (!have) wgt = 0.0;
                } else {
                    double d[UQ];
#pragma unroll
                    for (int u = 0; u < UQ; u++) d[u] = fma(x[0][1], y[u][0][1], x[0][0] * y[u][0][0]);
                    wgt = group_multi_reduce<LPR, UQ>(d, lig);
                    if constexpr (OP == Op::kSddmm) {  // (the folded Hadamard of the stand-alone SDDMM, as in the general loop)
                        if (svalues != nullptr) wgt *= line_value(sv, kmine);
                    }
                    if (!vals_overwrite) wgt += line_value(vv, kmine);
                    if (fused_op(OP) && (flags & HNH_FUSED_LEAKY_RELU)) {  // the activated weight is what gets stored
                        if (svalues != nullptr) wgt *= line_value(sv, kmine);
                        wgt = wgt > 0.0 ? wgt : ex.leaky_alpha * wgt;
                        wq[q] = wgt;
                    } else {
                        wq[q] = wgt;
                        if (fused_op(OP) && svalues != nullptr) wgt *= line_value(sv, kmine);
                    }
                    if (!have) wgt = 0.0;
                }
                if constexpr (OP != Op::kSddmm) {
#pragma unroll
                    for (int u = 0; u < UQ; u++) {
                        const double wu = __shfl(wgt, u * SUBQ, LPR);
                        acc[0][0] = fma(wu, y[u][0][0], acc[0][0]);
                        acc[0][1] = fma(wu, y[u][0][1], acc[0][1]);
                    }
                    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]));
                }
            };
            gatherq(0, ya);
            gatherq(1, yb);
            __builtin_amdgcn_sched_barrier(0);
            computeq(0, ya);
            __builtin_amdgcn_sched_barrier(0);
            gatherq(2, ya);
            __builtin_amdgcn_sched_barrier(0);
            computeq(1, yb);
            __builtin_amdgcn_sched_barrier(0);
            gatherq(3, yb);
            __builtin_amdgcn_sched_barrier(0);
            computeq(2, ya);
            computeq(3, yb);
            if constexpr (OP != Op::kSpmm) {
                // the trip's 16 results as one line: lane l stores values VPL l .. VPL l + VPL - 1; result k = 4 q + u sits in lanes
                // [u SUBQ, (u + 1) SUBQ) of wq[q]
                double outv[VPL];
#pragma unroll
                for (int j = 0; j < VPL; j++) {
                    const int k = lig * VPL + j;
                    outv[j] = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const double t = __shfl(wq[q], (k & 3) * SUBQ, LPR);
                        if ((k >> 2) == q) outv[j] = t;
                    }
                }
                double* vp = values + e0 + lig * VPL;
                if (!part) {
                    if constexpr (VPL == 4) {
                        *reinterpret_cast<double2*>(vp) = make_double2(outv[0], outv[1]);
                        *reinterpret_cast<double2*>(vp + 2) = make_double2(outv[2], outv[3]);
                    } else if constexpr (VPL == 2) {
                        *reinterpret_cast<double2*>(vp) = make_double2(outv[0], outv[1]);
                    } else {
                        *vp = outv[0];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < VPL; j++) {
                        const int pos = e0 + lig * VPL + j;
                        if (pos >= beg && pos < end) vp[j] = outv[j];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < VPL; j++) vv[j] = vvn[j];
        }
    }

    if constexpr (!NARROW) {
    int e = beg;
    if constexpr (PIPE) {
        // Four half batches A, B, C, D per trip through two register buffers: gathers of B fly while A is computed, C's
        // (into A's buffer) while B is computed, D's while C is.  Every load a trip consumes is issued inside the trip —
        // nothing but the column indices is carried round the loop — and every gather of the steady state is issued
        // unconditionally, so the waits in front of a half's arithmetic cover exactly that half's loads.
        double ya[U][VEC][W], yb[U][VEC][W];
        if (e + 4 * U <= end) {
            int c0[U], c1[U], c2[U], c3[U];
            load_idx(kFull, e, c0);
            load_idx(kFull, e + U, c1);
            load_idx(kFull, e + 2 * U, c2);
            load_idx(kFull, e + 3 * U, c3);
            for (;;) {
                gather(kFull, c0, ya);
                gather(kFull, c1, yb);
                const bool more = e + 8 * U <= end;
                if (more) {  // the next trip's indices travel with this trip's gathers
                    load_idx(kFull, e + 4 * U, c0);
                    load_idx(kFull, e + 5 * U, c1);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the issue order (the scheduler would sink gathers below the waits)
                compute(kFull, e, ya);
                __builtin_amdgcn_sched_barrier(0);
                gather(kFull, c2, ya);
                __builtin_amdgcn_sched_barrier(0);
                compute(kFull, e + U, yb);
                __builtin_amdgcn_sched_barrier(0);
                gather(kFull, c3, yb);
                if (more) {
                    load_idx(kFull, e + 6 * U, c2);
                    load_idx(kFull, e + 7 * U, c3);
                }
                __builtin_amdgcn_sched_barrier(0);
                compute(kFull, e + 2 * U, ya);
                compute(kFull, e + 3 * U, yb);
                e += 4 * U;
                if (!more) break;
            }
        }
        // fewer than 4U nonzeros left: one unmasked double half when at least 2U remain ...
        if (e + 2 * U <= end) {
            int c0[U], c1[U];
            load_idx(kFull, e, c0);
            load_idx(kFull, e + U, c1);
            gather(kFull, c0, ya);
            gather(kFull, c1, yb);
            __builtin_amdgcn_sched_barrier(0);
            compute(kFull, e, ya);
            compute(kFull, e + U, yb);
            e += 2 * U;
        }
        // ... then masked double halves (both halves' gathers issued together): at most one trip
        while (e < end) {
            int c0[U], c1[U];
            load_idx(kMasked, e, c0);
            load_idx(kMasked, e + U, c1);
            gather(kMasked, c0, ya);
            gather(kMasked, c1, yb);
            compute(kMasked, e, ya);
            compute(kMasked, e + U, yb);  // nothing to do (all slots masked) when fewer than U nonzeros were left
            e += 2 * U;
        }
    } else {
        if (e + U <= end) {
            int c[U];
            load_idx(kFull, e, c);
            for (;;) {
                double y[U][VEC][W];
                gather(kFull, c, y);
                const bool more = e + 2 * U <= end;
                if (more) load_idx(kFull, e + U, c);  // next batch's indices travel with this batch's gathers
                compute(kFull, e, y);
                e += U;
                if (!more) break;
            }
        }
        if (e < end) {
            int c[U];
            double y[U][VEC][W];
            load_idx(kMasked, e, c);
            gather(kMasked, c, y);
            compute(kMasked, e, y);
        }
    }
    }  // !NARROW

    if constexpr (OP != Op::kSddmm) {
        if (fused_op(OP) && (flags & kInternalEpilogue) && !atomic_out) {  // the row is complete in this launch
            double part = 0.0;
#pragma unroll
            for (int v = 0; v < VEC; v++)
#pragma unroll
                for (int w = 0; w < W; w++) {
                    acc[v][w] = fma(ex.x_scale, x[v][w], acc[v][w]);
                    part = fma(x[v][w], acc[v][w], part);
                }
            if (ex.rowdot != nullptr || (OP == Op::kFusedCg && ex.cg_x != nullptr)) part = group_sum<LPR>(part);
            if (ex.rowdot != nullptr && lig == 0) ex.rowdot[row] = part;
            if constexpr (OP == Op::kFusedCg) {
              if (ex.relu_dst != nullptr) {  // the finished row leaves through a ReLU into a column block of a wider matrix
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    double y[W];
#pragma unroll
                    for (int w = 0; w < W; w++) y[w] = fmax(acc[v][w], 0.0);
                    if (act[v]) store_w_stream<W>(ex.relu_dst + row * ex.relu_ld + coff[v], y);
                }
                return;  // Out is scratch
              }
              if (ex.cg_x != nullptr) {
                // x = p (search direction), acc = Mp, part = <p, Mp>: the remaining CG updates of this row, see hnh_cg_update
                const double rs = ex.cg_rsold[row] + ex.cg_eps;
                const double alpha = rs / (part + ex.cg_eps);
                double rr[VEC][W];
                double rsnew = 0.0;
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    double xs[W];
#pragma unroll
                    for (int w = 0; w < W; w++) { xs[w] = 0.0; rr[v][w] = 0.0; }
                    if (act[v]) {
                        load_w_stream<W>(xs, ex.cg_x + row * ld + coff[v]);
                        load_w_stream<W>(rr[v], ex.cg_r + row * ld + coff[v]);
                    }
#pragma unroll
                    for (int w = 0; w < W; w++) {
                        xs[w] = fma(alpha, x[v][w], xs[w]);
                        rr[v][w] = fma(-alpha, acc[v][w], rr[v][w]);
                        rsnew = fma(rr[v][w], rr[v][w], rsnew);
                    }
                    if (act[v]) {
                        store_w_stream<W>(ex.cg_x + row * ld + coff[v], xs);
                        store_w_stream<W>(ex.cg_r + row * ld + coff[v], rr[v]);
                    }
                }
                rsnew = group_sum<LPR>(rsnew);
                const double beta = rsnew / rs;
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    double pn[W];
#pragma unroll
                    for (int w = 0; w < W; w++) pn[w] = fma(beta, x[v][w], rr[v][w]);
                    if (act[v]) store_w_stream<W>(ex.cg_p + row * ld + coff[v], pn);
                }
                if (lig == 0) ex.cg_rsold[row] = rsnew;
              }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            if (!act[v]) continue;
            if (atomic_out) {
                if (part_row != nullptr) {  // this segment's partial output row; reduce_long_kernel adds the segments up in order
                    store_w_stream<W>(part_row + coff[v], acc[v]);
                } else {
#pragma unroll
                    for (int w = 0; w < W; w++) unsafeAtomicAdd(Out + row * ld + coff[v] + w, acc[v][w]);
                }
            } else {
                store_w_stream<W>(Out + row * ld + coff[v], acc[v]);
            }
        }
    }
}

#ifndef HNH_ROW_WAVES
#define HNH_ROW_WAVES 1  // waves per SIMD the row kernels are compiled for (register budget 512 / waves)
#endif
template <Op OP, int LPR, int VEC, int W, bool EXACT, bool NARROW = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(HNH_ROW_WAVES, 8))) void row_kernel(int64_t rows, const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ beg_ptr, const int32_t* __restrict__ end_ptr,
                                                     const int32_t* __restrict__ colidx, double* values,
                                                     const double* __restrict__ svalues,
                                                     const double* __restrict__ X, const double* __restrict__ Y,
                                                     double* __restrict__ Out, int64_t ld, int col0, int ncols,
                                                     unsigned flags, Extras ex) {
    constexpr int GROUPS = kBlock / LPR;
    const int tid = threadIdx.x;
    const int lig = tid % LPR;
    int64_t row = (int64_t)blockIdx.x * GROUPS + tid / LPR;
    if constexpr (LPR == 64) row = ((int64_t)blockIdx.x * GROUPS) + __builtin_amdgcn_readfirstlane(tid >> 6);
    if (row >= rows) return;

    // the row's nonzeros handled by this launch: the whole row (beg_ptr = rowptr, end_ptr = rowptr + 1) or the part of it
    // that falls into one column panel (per-row boundaries from panel_split_kernel)
    int beg = beg_ptr[row];
    int end = end_ptr[row];
    if constexpr (LPR == 64) {
        beg = __builtin_amdgcn_readfirstlane(beg);
        end = __builtin_amdgcn_readfirstlane(end);
    }
    if (flags & kInternalSplitLong) {
        // hub rows (judged by the WHOLE row, also when this launch covers one column panel of it) are left to
        // long_row_kernel; its segments are ADDED to the output row, so an overwritten output row has to start from zero
        int full = rowptr[row + 1] - rowptr[row];
        if constexpr (LPR == 64) full = __builtin_amdgcn_readfirstlane(full);
        if (full > long_row_of(flags)) {
            // (the row's epilogue, if any, runs in reduce_long_kernel once its segments are added up)
            if (OP != Op::kSddmm && (flags & HNH_FUSED_OUT_OVERWRITE)) { end = beg; flags &= ~kInternalEpilogue; }
            else return;
        }
    }
    if (OP == Op::kSddmm && beg == end) return;
    process_row<OP, LPR, VEC, W, EXACT, NARROW>(row, beg, end, false, colidx, values, svalues, X, Y, Out, ld, col0, ncols, flags, lig, ex);
}

// Infinity-Cache panels.  A launch that gathers from ALL rows of the dense operand revisits them at random across a
// working set several times the 256 MiB memory-side cache; restricting a launch to the nonzeros of one column panel
// (~512 MiB of the operand) lets about half of its gathers hit that cache (measured -12 % at config 2).  Column indices
// are sorted within a CSR row, so a panel is a contiguous piece of every row: split[(q-1) * rows + r] = first nonzero
// of row r with column >= q * width, q = 1 .. panels-1; launch q works on [split[q-1][r], split[q][r]).
__global__ __launch_bounds__(kBlock) void panel_split_kernel(int64_t rows, const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ colidx, int panels, int width,
                                                             int32_t* __restrict__ split) {
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    int lo = beg;
    for (int q = 1; q < panels; q++) {
        const int bound = q * width;
        int hi = end;  // boundaries are non-decreasing in q: continue from the previous one
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (colidx[mid] < bound) lo = mid + 1;
            else hi = mid;
        }
        split[(int64_t)(q - 1) * rows + row] = lo;
    }
}

// Caller-defined windows (hnh_csr_window_bounds): split[b * rows + r] = first nonzero of row r with column >= bound b.
constexpr int kMaxWindowBounds = 15;
struct WindowBounds {
    int32_t v[kMaxWindowBounds];
    int n;
};
__global__ __launch_bounds__(kBlock) void window_bounds_kernel(int64_t rows, const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ colidx, WindowBounds wb,
                                                               int32_t* __restrict__ split) {
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= rows) return;
    const int end = rowptr[row + 1];
    int lo = rowptr[row];
    for (int b = 0; b < wb.n; b++) {  // bounds are non-decreasing: continue from the previous one
        int hi = end;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (colidx[mid] < wb.v[b]) lo = mid + 1;
            else hi = mid;
        }
        split[(int64_t)b * rows + row] = lo;
    }
}

// element-wise passes of the column-tiled fused fallback, restricted to a window: 0 = zero, 1 = LeakyReLU, 2 = *= svalues.
// Hub rows follow the row passes' rule: untouched by every window but the last, whole with the last one.
__global__ __launch_bounds__(kBlock) void window_values_kernel(int64_t rows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ beg,
                                                               const int32_t* __restrict__ end, double* __restrict__ values,
                                                               const double* __restrict__ svalues, int mode, double alpha, int long_row, bool last) {
    const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 64;
    const int lane = threadIdx.x % 64;
    if (row >= rows) return;
    int b = beg[row], e2 = end[row];
    if (long_row > 0 && rowptr[row + 1] - rowptr[row] > long_row) {  // long_row == 0: this pass has no hub rows
        if (!last) return;
        b = rowptr[row];
        e2 = rowptr[row + 1];
    }
    for (int e = b + lane; e < e2; e += 64) {
        if (mode == 0) values[e] = 0.0;
        else if (mode == 1) { const double x = values[e]; values[e] = fmax(x, 0.0) + fmin(x, 0.0) * alpha; }
        else values[e] *= svalues[e];
    }
}

// One work item = kLongSeg consecutive nonzeros of a long row; items are listed by build_long_list_kernel.
template <Op OP, int LPR, int VEC, int W, bool EXACT, bool NARROW = false>
__global__ __launch_bounds__(kBlock) void long_row_kernel(const int2* __restrict__ items, const int* __restrict__ item_count,
                                                          int capacity, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ colidx, double* values,
                                                          const double* __restrict__ svalues, const double* __restrict__ X,
                                                          const double* __restrict__ Y, double* __restrict__ Out, int64_t ld,
                                                          int col0, int ncols, unsigned flags, Extras ex, double* partials,
                                                          int partial_items) {
    constexpr int GROUPS = kBlock / LPR;
    const int tid = threadIdx.x;
    const int lig = tid % LPR;
    int count = *item_count;
    if (count > capacity) count = capacity;
    const int ngroups = (int)gridDim.x * GROUPS;
    for (int it = (int)blockIdx.x * GROUPS + tid / LPR; it < count; it += ngroups) {
        const int2 item = items[it];
        const int64_t row = item.x;
        const int rbeg = rowptr[row], rend = rowptr[row + 1];
        const int beg = rbeg + item.y * kLongSeg;
        const int end = (beg + kLongSeg < rend) ? beg + kLongSeg : rend;
        double* part_row = (partials != nullptr && it < partial_items) ? partials + (int64_t)it * ld : nullptr;
        process_row<OP, LPR, VEC, W, EXACT, NARROW>(row, beg, end, true, colidx, values, svalues, X, Y, Out, ld, col0, ncols, flags, lig, ex, part_row);
    }
}

template <int LPR, int W>
__device__ __forceinline__ void row_epilogue_row(double* Out, const double* X, const Extras& ex, int64_t row, int R, int lig);

// Out[row, col0 .. col0 + ncols) += the sum of the row's segments' partial rows in a FIXED order: the deterministic replacement
// of atomically combined segments.  One workgroup per hub row (a row's segments are consecutive items): wave w adds up the w-th
// quarter of the segments front to back (8 loads in flight), then wave 0 adds the four quarter sums in wave order.
template <int W>
__global__ __launch_bounds__(kBlock) void reduce_long_kernel(const int4* __restrict__ hub_rows, const int* __restrict__ counts, int capacity_rows,
                                                             const double* __restrict__ partials, int partial_items,
                                                             double* Out, int64_t ld, int col0, int ncols, const double* X, Extras ex,
                                                             bool epilogue) {
    // epilogue: the fused pass's row epilogue (x_scale / rowdot / CG updates / ReLU delivery, see row_epilogue_row) on the hub row
    // once its segments are added up — the short rows of the block got theirs inside the row kernel (whole rows: ld == R)
    constexpr int WAVES = kBlock / 64;
    __shared__ double quarter[WAVES][64][W];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int nrows = counts[1];
    if (nrows > capacity_rows) nrows = capacity_rows;
    for (int e = (int)blockIdx.x; e < nrows; e += (int)gridDim.x) {
        const int4 h = hub_rows[e];  // (row, first item, segments, -)
        int nseg = h.z;
        if (h.y + nseg > partial_items) nseg = partial_items > h.y ? partial_items - h.y : 0;  // (the rest went the atomic way)
        const int per = (nseg + WAVES - 1) / WAVES;
        const int s_beg = wave * per < nseg ? wave * per : nseg;
        const int s_end = s_beg + per < nseg ? s_beg + per : nseg;
        double* o = Out + (int64_t)h.x * ld + col0;
        const double* p0 = partials + (int64_t)h.y * ld + col0;
        for (int c0 = 0; c0 < ncols; c0 += 64 * W) {
            const int c = c0 + lane * W;
            const bool live = c < ncols;  // ncols is a multiple of W
            double sum[W];
#pragma unroll
            for (int w = 0; w < W; w++) sum[w] = 0.0;
            int s2 = s_beg;
            for (; s2 + 8 <= s_end; s2 += 8) {
                double y[8][W];
#pragma unroll
                for (int u = 0; u < 8; u++) {
#pragma unroll
                    for (int w = 0; w < W; w++) y[u][w] = 0.0;
                    if (live) load_w<W>(y[u], p0 + (int64_t)(s2 + u) * ld + c);
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int w = 0; w < W; w++) sum[w] += y[u][w];
            }
            for (; s2 < s_end; s2++) {
                double y[W];
                if (live) {
                    load_w<W>(y, p0 + (int64_t)s2 * ld + c);
#pragma unroll
                    for (int w = 0; w < W; w++) sum[w] += y[w];
                }
            }
#pragma unroll
            for (int w = 0; w < W; w++) quarter[wave][lane][w] = sum[w];
            __syncthreads();
            if (wave == 0 && live) {
                double y[W];
                load_w<W>(y, o + c);
#pragma unroll
                for (int w = 0; w < W; w++) {
                    double t = quarter[0][lane][w];
#pragma unroll
                    for (int q = 1; q < WAVES; q++) t += quarter[q][lane][w];
                    y[w] += t;
                }
                store_w<W>(o + c, y);
            }
            __syncthreads();
        }
        if (epilogue && wave == 0) row_epilogue_row<64, W>(Out, X, ex, (int64_t)h.x, (int)ld, lane);  // wave 0 reads back what it stored
    }
}

__global__ __launch_bounds__(kBlock) void build_long_list_kernel(int64_t rows, const int32_t* __restrict__ rowptr,
                                                                 int2* __restrict__ items, int* __restrict__ item_count,
                                                                 int capacity, int long_row, int4* __restrict__ hub_rows, int capacity_rows) {
    // one thread per row; a WAVE reserves its hub rows' item and row slots with one atomic each (a skewed graph has tens of
    // thousands of hub rows: one atomic per row on the same two counters took 0.3 ms per pass on R-MAT 2^20)
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int nseg = 0;
    if (row < rows) {
        const int len = rowptr[row + 1] - rowptr[row];
        if (len > long_row) nseg = (len + kLongSeg - 1) / kLongSeg;
    }
    const unsigned long long hubs = __ballot(nseg > 0);
    if (hubs == 0ull) return;
    int incl = nseg;  // inclusive prefix sum of nseg over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    const int total = __shfl(incl, 63, 64);
    int base_items = 0, base_rows = 0;
    if (lane == 0) {
        base_items = atomicAdd(item_count, total);  // item_count[0]: items, [1]: hub rows
        base_rows = atomicAdd(item_count + 1, __popcll(hubs));
    }
    base_items = __shfl(base_items, 0, 64);
    base_rows = __shfl(base_rows, 0, 64);
    if (nseg == 0) return;
    const int base = base_items + incl - nseg;
    for (int s2 = 0; s2 < nseg; s2++)
        if (base + s2 < capacity) items[base + s2] = make_int2((int)row, s2);
    const int slot = base_rows + __popcll(hubs & ((1ull << lane) - 1ull));
    if (slot < capacity_rows) hub_rows[slot] = make_int4((int)row, base, nseg, 0);
}

__global__ __launch_bounds__(kBlock) void max_row_nnz_kernel(int64_t rows, const int32_t* __restrict__ rowptr, int* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int m = 0;
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < rows; row += stride) {
        const int len = rowptr[row + 1] - rowptr[row];
        m = len > m ? len : m;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// ---------------------------------------------------------------- COO SDDMM (no rowptr available)
// A group owns U consecutive nonzeros at a time; both dense rows are gathered per nonzero.
template <int LPR, int VEC, int W, bool EXACT>
__global__ __launch_bounds__(kBlock) void sddmm_coo_kernel(int64_t nnz, const int32_t* __restrict__ rowidx,
                                                           const int32_t* __restrict__ colidx, double* values,
                                                           const double* __restrict__ X,
                                                           const double* __restrict__ Y, int64_t ld, int col0,
                                                           int ncols) {
    constexpr int U = Unroll<LPR, VEC>::value > 4 ? 4 : Unroll<LPR, VEC>::value;
    constexpr int SUB = LPR / U;
    constexpr int GROUPS = kBlock / LPR;
    const int tid = threadIdx.x;
    const int lig = tid % LPR;
    const int64_t group = (int64_t)blockIdx.x * GROUPS + tid / LPR;
    const int64_t ngroups = (int64_t)gridDim.x * GROUPS;
    for (int64_t e = group * U; e < nnz; e += ngroups * U) {
        double d[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            double s = 0.0;
            if (e + u < nnz) {
                const int64_t r = rowidx[e + u], c = colidx[e + u];
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    const int cc = (v * LPR + lig) * W;
                    if (EXACT || cc < ncols) {
                        double a[W], b[W];
                        load_w<W>(a, X + r * ld + col0 + cc);
                        load_w<W>(b, Y + c * ld + col0 + cc);
#pragma unroll
                        for (int w = 0; w < W; w++) s = fma(a[w], b[w], s);
                    }
                }
            }
            d[u] = s;
        }
        const double tot = group_multi_reduce<LPR, U>(d, lig);
        const int64_t mine = e + lig / SUB;
        if (mine < nnz && lig % SUB == 0) values[mine] += tot;
    }
}

// ---------------------------------------------------------------- element-wise
// Streaming element-wise kernels.  Each thread keeps kEwUnroll independent 16-byte accesses in flight per loop trip
// (a grid-stride loop with one 8-byte access per trip reached only 4.3-4.8 TB/s; the copy engine does 5.1, memset 6.5).
constexpr int kEwUnroll = 4;

template <typename F>
__device__ __forceinline__ void ew_apply2(int64_t n, F&& f) {  // f(i, count): elements [i, i + count), count in {1, 2}
    const int64_t pairs = n / 2, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t p0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; p0 < pairs; p0 += stride * kEwUnroll) {
#pragma unroll
        for (int u = 0; u < kEwUnroll; u++) {
            const int64_t p = p0 + u * stride;
            if (p < pairs) f(2 * p, 2);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) f(n - 1, 1);
}

__global__ __launch_bounds__(kBlock) void fill_kernel(double* __restrict__ dst, int64_t n, double v, bool vec) {
    if (vec) {
        ew_apply2(n, [&](int64_t i, int c) {
            if (c == 2) *reinterpret_cast<double2*>(dst + i) = make_double2(v, v);
            else dst[i] = v;
        });
    } else {
        const int64_t stride = (int64_t)gridDim.x * kBlock;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) dst[i] = v;
    }
}

__global__ __launch_bounds__(kBlock) void hadamard_kernel(double* out, const double* a, const double* b, int64_t n, bool vec) {
    if (vec) {
        const int64_t pairs = n / 2, stride = (int64_t)gridDim.x * kBlock;
        for (int64_t p0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; p0 < pairs; p0 += stride * kEwUnroll) {
            double2 x[kEwUnroll], y[kEwUnroll];
#pragma unroll
            for (int u = 0; u < kEwUnroll; u++) {
                const int64_t p = p0 + u * stride;
                if (p < pairs) { x[u] = reinterpret_cast<const double2*>(a)[p]; y[u] = reinterpret_cast<const double2*>(b)[p]; }
            }
#pragma unroll
            for (int u = 0; u < kEwUnroll; u++) {
                const int64_t p = p0 + u * stride;
                if (p < pairs) reinterpret_cast<double2*>(out)[p] = make_double2(x[u].x * y[u].x, x[u].y * y[u].y);
            }
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = a[n - 1] * b[n - 1];
    } else {
        const int64_t stride = (int64_t)gridDim.x * kBlock;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = a[i] * b[i];
    }
}

__global__ __launch_bounds__(kBlock) void axpy_kernel(double* y, const double* __restrict__ x, double alpha, int64_t n, bool vec) {
    if (vec) {
        const int64_t pairs = n / 2, stride = (int64_t)gridDim.x * kBlock;
        for (int64_t p0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; p0 < pairs; p0 += stride * kEwUnroll) {
            double2 xv[kEwUnroll], yv[kEwUnroll];
#pragma unroll
            for (int u = 0; u < kEwUnroll; u++) {
                const int64_t p = p0 + u * stride;
                if (p < pairs) { xv[u] = reinterpret_cast<const double2*>(x)[p]; yv[u] = reinterpret_cast<const double2*>(y)[p]; }
            }
#pragma unroll
            for (int u = 0; u < kEwUnroll; u++) {
                const int64_t p = p0 + u * stride;
                if (p < pairs) reinterpret_cast<double2*>(y)[p] = make_double2(fma(alpha, xv[u].x, yv[u].x), fma(alpha, xv[u].y, yv[u].y));
            }
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = fma(alpha, x[n - 1], y[n - 1]);
    } else {
        const int64_t stride = (int64_t)gridDim.x * kBlock;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) y[i] = fma(alpha, x[i], y[i]);
    }
}

// dst[i,:] += sum_k src[chunk-major row of (block k, row i), :] for the rows of chunks [q0, q1) — the closing step of the mesh
// reduce-scatter (hnh_sum_chunked_blocks_f64): the nb partial blocks a rank has received are added to its own rows in block order
// k = 0 .. nb - 1, whatever the launch geometry (bit-identical run to run).  Each thread owns one 16-byte piece of an output row and
// keeps its nb + 1 loads in flight together.
struct ChunkCuts {
    long long cut[HNH_MAX_CHUNKS + 1];
};
template <bool VEC2>
__global__ __launch_bounds__(kBlock) void sum_chunks_kernel(double* dst, const double* __restrict__ src, int nb, ChunkCuts cc, int q0, int q1, int R) {
    const long long row0 = cc.cut[q0], row1 = cc.cut[q1];
    constexpr int W = VEC2 ? 2 : 1;
    const long long per_row = R / W, items = (row1 - row0) * per_row, stride = (long long)gridDim.x * kBlock;
    for (long long t = (long long)blockIdx.x * kBlock + threadIdx.x; t < items; t += stride) {
        const long long i = row0 + t / per_row;
        const int col = (int)(t % per_row) * W;
        int q = q0;
        while (q + 1 < q1 && cc.cut[q + 1] <= i) q++;
        const long long w = cc.cut[q + 1] - cc.cut[q];
        const double* s = src + ((long long)nb * cc.cut[q] + (i - cc.cut[q])) * R + col;
        double* d = dst + i * R + col;
        if constexpr (VEC2) {
            double2 acc = *reinterpret_cast<const double2*>(d);
            for (int k = 0; k < nb; k++) {
                const double2 v = *reinterpret_cast<const double2*>(s + (long long)k * w * R);
                acc.x += v.x;
                acc.y += v.y;
            }
            *reinterpret_cast<double2*>(d) = acc;
        } else {
            double acc = *d;
            for (int k = 0; k < nb; k++) acc += s[(long long)k * w * R];
            *d = acc;
        }
    }
}

__global__ __launch_bounds__(kBlock) void expand_rowptr_kernel(int64_t rows, const int32_t* __restrict__ rowptr,
                                                               int32_t* __restrict__ rowidx) {
    // one wave per row (rows are short on the target matrices)
    const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / 64;
    const int lane = threadIdx.x % 64;
    if (row >= rows) return;
    for (int e = rowptr[row] + lane; e < rowptr[row + 1]; e += 64) rowidx[e] = (int32_t)row;
}

// out[i] = <A[i,:], B[i,:]>; a group of LPR lanes per row, 16-byte loads when W == 2
template <int LPR, int W>
__global__ __launch_bounds__(kBlock) void rowdot_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                        double* __restrict__ out, int64_t rows, int R) {
    constexpr int GROUPS = kBlock / LPR;
    const int lig = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * GROUPS + threadIdx.x / LPR;
    if (row >= rows) return;
    const double* a = A + row * R;
    const double* b = B + row * R;
    double s = 0.0;
    for (int c = lig * W; c < R; c += LPR * W) {
        double x[W], y[W];
        load_w<W>(x, a + c);
        load_w<W>(y, b + c);
#pragma unroll
        for (int w = 0; w < W; w++) s = fma(x[w], y[w], s);
    }
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) s += shfl_xor_f64(s, m);
    if (lig == 0) out[row] = s;
}

// Out[i,:] += x_scale * X[i,:];  rowdot[i] = <X[i,:], Out[i,:]>;  optionally the CG updates of hnh_cg_update — the fused pass's
// row epilogue as its own launch (rows completed by several launches or by atomically combined hub-row segments)
template <int LPR, int W>
__device__ __forceinline__ void row_epilogue_row(double* Out, const double* X, const Extras& ex, int64_t row, int R, int lig) {
    double* o = Out + row * R;
    const double* xr = X + row * R;
    double s = 0.0;
    for (int c = lig * W; c < R; c += LPR * W) {
        double x[W], y[W];
        load_w<W>(x, xr + c);
        load_w<W>(y, o + c);
#pragma unroll
        for (int w = 0; w < W; w++) {
            y[w] = fma(ex.x_scale, x[w], y[w]);
            s = fma(x[w], y[w], s);
        }
        if (ex.x_scale != 0.0) store_w<W>(o + c, y);
    }
    s = group_sum<LPR>(s);
    if (lig == 0 && ex.rowdot != nullptr) ex.rowdot[row] = s;
    if (ex.relu_dst != nullptr) {
        double* d = ex.relu_dst + row * ex.relu_ld;
        for (int c = lig * W; c < R; c += LPR * W) {
            double y[W];
            load_w<W>(y, o + c);
#pragma unroll
            for (int w = 0; w < W; w++) y[w] = fmax(y[w], 0.0);
            store_w<W>(d + c, y);
        }
        return;
    }
    if (ex.cg_x == nullptr) return;
    // second sweep over the row (it is in L1/L2 now): x += alpha p, r -= alpha Mp, <r, r>; third: p = r + beta p
    const double rs = ex.cg_rsold[row] + ex.cg_eps;
    const double alpha = rs / (s + ex.cg_eps);
    double* xs_row = ex.cg_x + row * R;
    double* r_row = ex.cg_r + row * R;
    double* p_row = ex.cg_p + row * R;
    double rsnew = 0.0;
    for (int c = lig * W; c < R; c += LPR * W) {
        double p[W], mp[W], xs[W], r[W];
        load_w<W>(p, p_row + c);
        load_w<W>(mp, o + c);
        load_w<W>(xs, xs_row + c);
        load_w<W>(r, r_row + c);
#pragma unroll
        for (int w = 0; w < W; w++) {
            xs[w] = fma(alpha, p[w], xs[w]);
            r[w] = fma(-alpha, mp[w], r[w]);
            rsnew = fma(r[w], r[w], rsnew);
        }
        store_w<W>(xs_row + c, xs);
        store_w<W>(r_row + c, r);
    }
    rsnew = group_sum<LPR>(rsnew);
    const double beta = rsnew / rs;
    for (int c = lig * W; c < R; c += LPR * W) {
        double p[W], r[W];
        load_w<W>(p, p_row + c);
        load_w<W>(r, r_row + c);  // this lane's own store above
#pragma unroll
        for (int w = 0; w < W; w++) p[w] = fma(beta, p[w], r[w]);
        store_w<W>(p_row + c, p);
    }
    if (lig == 0) ex.cg_rsold[row] = rsnew;
}

template <int LPR, int W>
__global__ __launch_bounds__(kBlock) void row_epilogue_kernel(double* __restrict__ Out, const double* X, Extras ex, int64_t rows, int R) {
    constexpr int GROUPS = kBlock / LPR;
    const int lig = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * GROUPS + threadIdx.x / LPR;
    if (row >= rows) return;
    row_epilogue_row<LPR, W>(Out, X, ex, row, R, lig);
}

// One CG update (als_conjugate_gradients.cpp:117-127):  X[i,:] += alpha[i] P[i,:];  Rm[i,:] -= alpha[i] MP[i,:];
// rsnew[i] = <Rm[i,:], Rm[i,:]>  — three dense passes of the reference in one
template <int LPR, int W>
__global__ __launch_bounds__(kBlock) void cg_step_kernel(double* __restrict__ X, double* __restrict__ Rm, const double* __restrict__ P,
                                                         const double* __restrict__ MP, const double* __restrict__ alpha,
                                                         double* __restrict__ rsnew, int64_t rows, int R) {
    constexpr int GROUPS = kBlock / LPR;
    const int lig = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * GROUPS + threadIdx.x / LPR;
    if (row >= rows) return;
    const double a = alpha[row];
    const int64_t base = row * R;
    double s = 0.0;
    for (int c = lig * W; c < R; c += LPR * W) {
        double x[W], r[W], p[W], mp[W];
        load_w<W>(x, X + base + c);
        load_w<W>(r, Rm + base + c);
        load_w<W>(p, P + base + c);
        load_w<W>(mp, MP + base + c);
#pragma unroll
        for (int w = 0; w < W; w++) {
            x[w] = x[w] + a * p[w];
            r[w] = r[w] - a * mp[w];
            s = fma(r[w], r[w], s);
        }
        store_w<W>(X + base + c, x);
        store_w<W>(Rm + base + c, r);
    }
    s = group_sum<LPR>(s);
    if (lig == 0) rsnew[row] = s;
}

// Y[i,:] = ya * yv[i] * Y[i,:] + xa * xv[i] * X[i,:]   (null vector = ones); kEwUnroll chunks in flight per thread
template <int W>
__global__ __launch_bounds__(kBlock) void row_scale_add_kernel(double* __restrict__ Y, const double* __restrict__ yv, double ya,
                                                               const double* __restrict__ X, const double* __restrict__ xv,
                                                               double xa, int64_t rows, int R) {
    const int64_t total = rows * (int64_t)R / W;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < total; i0 += stride * kEwUnroll) {
        double y[kEwUnroll][W], x[kEwUnroll][W], fy[kEwUnroll], fx[kEwUnroll];
#pragma unroll
        for (int u = 0; u < kEwUnroll; u++) {
            const int64_t i = i0 + u * stride;
            if (i < total) {
                const int64_t e = i * W, row = e / R;
                fy[u] = ya * (yv ? yv[row] : 1.0);
                fx[u] = xa * (xv ? xv[row] : 1.0);
                load_w<W>(y[u], Y + e);
                load_w<W>(x[u], X + e);
            }
        }
#pragma unroll
        for (int u = 0; u < kEwUnroll; u++) {
            const int64_t i = i0 + u * stride;
            if (i < total) {
#pragma unroll
                for (int w = 0; w < W; w++) y[u][w] = fy[u] * y[u][w] + fx[u] * x[u][w];
                store_w<W>(Y + i * W, y[u]);
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void vec_add_scalar_kernel(double* v, double c, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) v[i] += c;
}

__device__ __forceinline__ unsigned long long splitmix64_dev(unsigned long long x) {
    unsigned long long z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(kBlock) void fill_hashed_kernel(double* __restrict__ dst, int64_t rows, int64_t cols, int64_t top_row,
                                                             int64_t left_col, int64_t rg, unsigned long long seed, double scale) {
    const int64_t stride = (int64_t)gridDim.x * kBlock, total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        const unsigned long long key = (unsigned long long)((top_row + i / cols) * rg + left_col + i % cols);
        const unsigned long long h = splitmix64_dev(seed * 0xD1342543DE82EF95ull + key * 0x9E3779B97F4A7C15ull);
        dst[i] = ((double)(h >> 11) * 0x1.0p-52 - 1.0) * scale;
    }
}

__global__ __launch_bounds__(kBlock) void vec_div_kernel(double* out, const double* num, const double* den, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = num[i] / den[i];
}

// ---------------------------------------------------------------- fp64 GEMM on the matrix cores (GAT's X * W)
// C[M x N] = A[M x K] * B[K x N], all row-major.  Block tile 128 x 128, K step 16, 4 waves in a 2 x 2 grid; each wave
// owns 64 x 64 = 4 x 4 accumulators of v_mfma_f64_16x16x4_f64, so one k-step of 4 feeds 16 MFMAs from 8 LDS reads
// (4 A fragments + 4 B fragments) — the LDS port, not the matrix core, is what a small wave tile saturates.
// Fragment layout (cdna_hip_programming.md, "f64 MFMA does NOT use these maps"):
//   A: lane l holds A[i = l & 15][k = l >> 4];  B: lane l holds B[k = l >> 4][j = l & 15];
//   C/D: register r of lane l is C[row = (l >> 4) + 4 r][col = l & 15].
// Both tiles sit in LDS k-major ([k][m], [k][n]) so the 16 lanes of a fragment row read 128 contiguous bytes.
// Pipeline: the next K tile is fetched from global memory into registers while the MFMAs of the current one run, then
// stored into the other LDS buffer — one barrier per K tile.
// Launch: 1-D grid; consecutive slots of one XCD (blockIdx.x % 8) are the column blocks of the SAME row block, so the
// A panel is read from HBM once and served from that XCD's L2 for the other column blocks.
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int kGemmBM = 128, kGemmBN = 128, kGemmBK = 16, kGemmLd = 128 + 4;
constexpr int kXcds = 8;

// WM = wave rows of the workgroup: 2 = the 128 x 128 tile of 4 waves described above (two workgroups of it share a CU);
// 4 = a 256 x 128 tile of 8 waves (HNH_GEMM_WAVES=8): two waves per SIMD come from ONE workgroup, so the kernel keeps its latency
// hiding when it shares the CUs with another stream's kernel (the GAT pipeline) — and reads each B tile for twice the rows.
template <int WM>
__global__ __launch_bounds__(128 * WM) void gemm_f64_kernel(int64_t M, int64_t N, int64_t K, const double* __restrict__ A,
                                                            const double* __restrict__ B, double* __restrict__ C, int64_t row_blocks,
                                                            int col_blocks, bool vec_ok) {
    constexpr int BM = 64 * WM, LDA = BM + 4, T = 128 * WM;
    constexpr int BPT = kGemmBK * kGemmBN / T;  // doubles of the B tile per thread (8 or 4)
    constexpr int BLANES = kGemmBN / BPT;       // threads per B tile row
    __shared__ double As[2][kGemmBK][LDA];
    __shared__ double Bs[2][kGemmBK][kGemmLd];
    // XCD-aware tile assignment
    const int64_t id = blockIdx.x;
    const int64_t xcd = id % kXcds, slot = id / kXcds;
    const int64_t rb = (slot / col_blocks) * kXcds + xcd;
    const int cb = (int)(slot % col_blocks);
    if (rb >= row_blocks) return;
    const int64_t row0 = rb * BM, col0 = (int64_t)cb * kGemmBN;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    d4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};

    // global -> register staging: A: row am, 8 consecutive k;  B: row bk, BPT consecutive n
    const int am = tid % BM, ak = (tid / BM) * 8;
    const int bk = tid / BLANES, bn = (tid % BLANES) * BPT;
    const bool interior = vec_ok && row0 + BM <= M && col0 + kGemmBN <= N;
    double ra[8], rbv[BPT];
    auto fetch = [&](int64_t k0) {
        if (interior && k0 + kGemmBK <= K) {
            const double* ap = A + (row0 + am) * K + k0 + ak;
            const double* bp = B + (k0 + bk) * N + col0 + bn;
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                const double2 va = *reinterpret_cast<const double2*>(ap + q);
                ra[q] = va.x; ra[q + 1] = va.y;
            }
#pragma unroll
            for (int q = 0; q < BPT; q += 2) {
                const double2 vb = *reinterpret_cast<const double2*>(bp + q);
                rbv[q] = vb.x; rbv[q + 1] = vb.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int64_t gr = row0 + am, gk = k0 + ak + q;
                ra[q] = (gr < M && gk < K) ? A[gr * K + gk] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < BPT; q++) {
                const int64_t hk = k0 + bk, hc = col0 + bn + q;
                rbv[q] = (hk < K && hc < N) ? B[hk * N + hc] : 0.0;
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 8; q++) As[buf][ak + q][am] = ra[q];
#pragma unroll
        for (int q = 0; q < BPT; q += 2) *reinterpret_cast<double2*>(&Bs[buf][bk][bn + q]) = make_double2(rbv[q], rbv[q + 1]);
    };

    const int64_t ktiles = (K + kGemmBK - 1) / kGemmBK;
    if (ktiles > 0) {
        fetch(0);
        stage(0);
    }
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;
    for (int64_t t = 0; t < ktiles; t++) {
        const int buf = (int)(t & 1);
        if (t + 1 < ktiles) fetch((t + 1) * kGemmBK);  // in flight during the MFMAs below
#pragma unroll
        for (int kk = 0; kk < kGemmBK; kk += 4) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = As[buf][kk + fk][wm + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = Bs[buf][kk + fk][wn + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < ktiles) stage(buf ^ 1);  // the other buffer: its readers finished before the previous barrier
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t row = row0 + wm + i * 16 + fk + 4 * r, col = col0 + wn + j * 16 + fr;
                if (row < M && col < N) C[row * N + col] = acc[i][j][r];
            }
}

__global__ __launch_bounds__(kBlock) void leaky_relu_kernel(double* v, double alpha, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double x = v[i];
        v[i] = fmax(x, 0.0) + fmin(x, 0.0) * alpha;
    }
}

__global__ __launch_bounds__(kBlock) void relu_store_cols_kernel(double* __restrict__ dst, int64_t ld, int64_t col0,
                                                                 const double* __restrict__ src, int64_t rows, int64_t cols) {
    const int64_t stride = (int64_t)gridDim.x * kBlock, total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / cols, c = i % cols;
        dst[r * ld + col0 + c] = fmax(src[i], 0.0);
    }
}

int ew_grid(int64_t n) {
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;  // 256 CUs x 8 resident blocks x 4 waves of blocks, grid-stride the rest
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ---------------------------------------------------------------- dispatch

struct Shape {
    int lpr = 0, vec = 0, w = 0;
    bool exact = false;  // false: tiled fallback (LPR 64, VEC 1, tile = 64 * w columns)
};

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool aligned128(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 127u) == 0; }

Shape pick_shape(int R, bool vec_ok) {
    Shape s;
    s.w = (R % 2 == 0 && vec_ok) ? 2 : 1;
    if (s.w == 2) {
        const int chunks = R / 2;
        if (chunks <= 64 && (chunks & (chunks - 1)) == 0) { s.lpr = chunks; s.vec = 1; s.exact = true; return s; }
        if (chunks % 64 == 0 && chunks / 64 <= 4) { s.lpr = 64; s.vec = chunks / 64; s.exact = true; return s; }
        if (chunks % 32 == 0 && (chunks / 32 == 3 || chunks / 32 == 5 || chunks / 32 == 7)) {
            s.lpr = 32; s.vec = chunks / 32; s.exact = true; return s;
        }
    }
    s.lpr = 64; s.vec = 1; s.exact = false;
    return s;
}

}  // namespace

// Structure-only results of the row passes for ONE block whose index arrays keep their contents (hnh_csr_plan of hnh_kernels.h).
struct hnh_csr_plan {
    // whose structure this is (checked on every use: a plan handed to another block is a caller error, not a silent wrong answer)
    const int32_t* rowptr = nullptr;
    const int32_t* colidx = nullptr;
    int64_t rows = -1, nnz = -1;
    struct Split {  // per-row panel boundaries for one (panels, width)
        int panels = 0, width = 0;
        int32_t* split = nullptr;
        unsigned long age = 0;
    } splits[3];
    unsigned long clock = 0;
    // hub-row work list for one threshold
    int threshold = 0;  // 0 = not built
    int2* items = nullptr;
    int* count = nullptr;
    int4* hub_rows = nullptr;
    int n_items = 0, n_hub_rows = 0;  // exact, read back once
};

namespace {

struct LongCtl {
    bool enabled = false;
    int2* items = nullptr;
    int* count = nullptr;
    int capacity = 0;
    int threshold = 0;  // rows longer than this are the list's (multiple of 64)
    int4* hub_rows = nullptr;   // (row, first item, segments) per hub row
    int capacity_rows = 0;
    double* partials = nullptr;  // items x (row pitch) doubles; nullptr: segments combine with atomics
    int partial_items = 0;
    size_t lds_pad = 0;  // unused LDS the row kernel's workgroups ask for, to cap their number per CU (row_occupancy_pad)
};

// Occupancy of the row kernels.  The memory system serves scattered rows a little FASTER when fewer waves compete for it: the
// pure-gather probe gains 2 % going from 8 to 2-3 waves per SIMD (profiles/archive/r02_gather_probe_occupancy.log), and the row kernels,
// which register use would let run at 6-8 waves, gain 2-3 % at 4-5 on a uniform matrix (Erdos-Renyi, R = 32 / 128 / 256;
// R = 64 loses 0.8 %) — but lose 9 % on a skewed one, where waves finish at very different times and more of them are needed
// to keep the CU busy (profiles/archive/r02_kbench_waves_cap_sweep.log).  So: blocks whose longest row is at most twice the mean, widths
// that were measured to gain, 5 waves per SIMD.  A workgroup is one wave per SIMD, so the cap is an LDS request the kernel never
// touches: 160 KiB / 5 per workgroup leaves room for exactly 5 of them on a CU.
size_t row_occupancy_pad(const hnh_ctx* ctx, const Shape& s, int64_t rows, int64_t nnz, int max_row_nnz) {
    if (ctx->row_waves_cap == 0) return 0;
    if (ctx->row_waves_cap > 0) return (size_t)(160 * 1024 / ctx->row_waves_cap) - 64;
    const bool measured_width = s.exact && ((s.lpr == 64 && (s.vec == 1 || s.vec == 2)) || (s.lpr == 16 && s.vec == 1));
    if (!measured_width || rows <= 0 || nnz <= 0 || max_row_nnz < 0) return 0;
    if ((int64_t)max_row_nnz * rows > 2 * nnz) return 0;  // not uniform
    return (size_t)(160 * 1024 / 5) - 64;
}

// Rows longer than this go to the long-row pass (see kLongRowMin); nnz < 0 = unknown
int long_row_threshold(const hnh_ctx* ctx, int64_t rows, int64_t nnz) {
    if (ctx->long_row_override > 0) return ctx->long_row_override;
    if (rows <= 0 || nnz < 0) return kLongRowMax;
    const int64_t t = (3 * nnz / rows + 63) / 64 * 64;
    return (int)(t < kLongRowMin ? kLongRowMin : (t > kLongRowMax ? kLongRowMax : t));
}

// Decides whether this call needs the long-row pass and, if so, builds the (row, segment) work list on the
// device.  max_row_nnz: the caller's knowledge of the longest row (< 0 = unknown -> the list is always built);
// nnz: number of nonzeros (< 0 = unknown -> read back from rowptr[rows], one 4-byte synchronous copy).
// out_pitch: row pitch (in doubles) of the output the segments add to, 0 for SDDMM — sizes the partial-row scratch.
// build_list = false: the launch only has to SKIP the hub rows (a window that is not the pass's last one) — threshold only.
int partial_scratch(hnh_ctx* ctx, hipStream_t st, int sidx, size_t items_bound, int64_t out_pitch, LongCtl* lc);

int prepare_long(hnh_ctx* ctx, hipStream_t st, int sidx, int64_t rows, const int32_t* rowptr, int64_t nnz, int max_row_nnz,
                 int64_t out_pitch, LongCtl* lc, bool build_list = true, hnh_csr_plan* plan = nullptr) {
    if (max_row_nnz >= 0 && max_row_nnz <= (ctx->long_row_override > 0 ? ctx->long_row_override : kLongRowMin)) return HNH_OK;
    if (nnz < 0) {
        int last = 0;
        HNH_TRY_HIP(ctx, hipMemcpyAsync(&last, rowptr + rows, sizeof(int), hipMemcpyDeviceToHost, st));
        HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
        nnz = last;
    }
    const int threshold = long_row_threshold(ctx, rows, nnz);
    if (nnz <= threshold || (max_row_nnz >= 0 && max_row_nnz <= threshold)) return HNH_OK;
    if (!build_list) {
        lc->enabled = true;
        lc->threshold = threshold;
        return HNH_OK;
    }
    const size_t cap = (size_t)(nnz / kLongSeg + nnz / threshold + 16);
    if (plan != nullptr) {
        // the list depends on rowptr and the threshold only: built once, its exact size read back once, then no call launches
        // build_long_list_kernel again (it was 0.3 ms of a 6.7 ms fused call on R-MAT 2^20)
        if (plan->threshold != threshold) {
            HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
            for (void* p : {(void*)plan->items, (void*)plan->count, (void*)plan->hub_rows})
                if (p) HNH_TRY_HIP(ctx, hipFree(p));
            plan->items = nullptr; plan->count = nullptr; plan->hub_rows = nullptr; plan->threshold = 0;
            const size_t cap_rows = (size_t)(nnz / threshold + 16);
            HNH_TRY_HIP(ctx, hipMalloc((void**)&plan->items, cap * sizeof(int2)));
            HNH_TRY_HIP(ctx, hipMalloc((void**)&plan->count, 2 * sizeof(int)));
            HNH_TRY_HIP(ctx, hipMalloc((void**)&plan->hub_rows, cap_rows * sizeof(int4)));
            HNH_TRY_HIP(ctx, hipMemsetAsync(plan->count, 0, 2 * sizeof(int), st));
            hipLaunchKernelGGL(build_long_list_kernel, dim3((unsigned)((rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, rows, rowptr, plan->items,
                               plan->count, (int)cap, threshold, plan->hub_rows, (int)cap_rows);
            if (int rc = hnh::check_hip(ctx, hipGetLastError(), "build_long_list_kernel launch")) return rc;
            int counts[2] = {0, 0};
            HNH_TRY_HIP(ctx, hipMemcpyAsync(counts, plan->count, sizeof counts, hipMemcpyDeviceToHost, st));
            HNH_TRY_HIP(ctx, hipStreamSynchronize(st));  // once per block: later calls may run on any stream
            plan->n_items = counts[0];
            plan->n_hub_rows = counts[1];
            plan->threshold = threshold;
        }
        if (plan->n_items == 0) return HNH_OK;  // no row above the threshold after all (the hint was an upper bound)
        lc->items = plan->items;
        lc->count = plan->count;
        lc->capacity = plan->n_items;
        lc->hub_rows = plan->hub_rows;
        lc->capacity_rows = plan->n_hub_rows;
        lc->enabled = true;
        lc->threshold = threshold;
        if (out_pitch > 0 && !ctx->hub_atomics) return partial_scratch(ctx, st, sidx, (size_t)plan->n_items, out_pitch, lc);
        return HNH_OK;
    }
    if (ctx->long_cap[sidx] < cap) {
        HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
        if (ctx->long_items[sidx]) HNH_TRY_HIP(ctx, hipFree(ctx->long_items[sidx]));
        ctx->long_items[sidx] = nullptr;
        HNH_TRY_HIP(ctx, hipMalloc(&ctx->long_items[sidx], cap * sizeof(int2)));
        ctx->long_cap[sidx] = cap;
    }
    if (!ctx->long_count[sidx]) HNH_TRY_HIP(ctx, hipMalloc((void**)&ctx->long_count[sidx], 2 * sizeof(int)));
    const size_t cap_rows = (size_t)(nnz / threshold + 16);
    if (ctx->long_rows_cap[sidx] < cap_rows) {
        HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
        if (ctx->long_rows[sidx]) HNH_TRY_HIP(ctx, hipFree(ctx->long_rows[sidx]));
        ctx->long_rows[sidx] = nullptr;
        HNH_TRY_HIP(ctx, hipMalloc(&ctx->long_rows[sidx], cap_rows * sizeof(int4)));
        ctx->long_rows_cap[sidx] = cap_rows;
    }
    if (out_pitch > 0 && !ctx->hub_atomics)
        if (int rc = partial_scratch(ctx, st, sidx, cap, out_pitch, lc)) return rc;
    lc->items = static_cast<int2*>(ctx->long_items[sidx]);
    lc->count = ctx->long_count[sidx];
    lc->capacity = (int)cap;
    lc->hub_rows = static_cast<int4*>(ctx->long_rows[sidx]);
    lc->capacity_rows = (int)cap_rows;
    lc->enabled = true;
    lc->threshold = threshold;
    HNH_TRY_HIP(ctx, hipMemsetAsync(lc->count, 0, 2 * sizeof(int), st));
    const int64_t blocks = (rows + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(build_long_list_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, rows, rowptr, lc->items, lc->count, lc->capacity, threshold,
                       lc->hub_rows, lc->capacity_rows);
    return hnh::check_hip(ctx, hipGetLastError(), "build_long_list_kernel launch");
}

// Partial output rows of the hub-row segments (ordered reduction instead of atomics: results do not depend on the order in which
// segments finish): one row of `out_pitch` doubles per segment, allocated once per stream and kept.  `items_bound` is the exact
// segment count when a plan knows it, else a bound that holds for ANY matrix of the block's size (nnz * pitch / 16 bytes at worst),
// which at wide rows would be more scratch than the CSR block itself: past ctx->hub_scratch_bytes (HNH_HUB_SCRATCH_MB, default
// 2 GiB = the segments of 5e8 hub-row nonzeros at R = 128) the remaining segments combine with atomics — exact within the
// parity tolerance, no longer bit-reproducible.
int partial_scratch(hnh_ctx* ctx, hipStream_t st, int sidx, size_t items_bound, int64_t out_pitch, LongCtl* lc) {
    size_t items = items_bound;
    if (items * (size_t)out_pitch * sizeof(double) > ctx->hub_scratch_bytes) items = ctx->hub_scratch_bytes / ((size_t)out_pitch * sizeof(double));
    const size_t need = items * (size_t)out_pitch * sizeof(double);
    if (ctx->long_partials_bytes[sidx] < need) {
        HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
        if (ctx->long_partials[sidx]) HNH_TRY_HIP(ctx, hipFree(ctx->long_partials[sidx]));
        ctx->long_partials[sidx] = nullptr;
        ctx->long_partials_bytes[sidx] = 0;
        HNH_TRY_HIP(ctx, hipMalloc(&ctx->long_partials[sidx], need));
        ctx->long_partials_bytes[sidx] = need;
    }
    lc->partials = items > 0 ? static_cast<double*>(ctx->long_partials[sidx]) : nullptr;
    lc->partial_items = (int)items;
    return HNH_OK;
}

template <Op OP, int LPR, int VEC, int W, bool EXACT, bool NARROW = false>
int launch_row(hnh_ctx* ctx, hipStream_t st, const LongCtl& lc, int64_t rows, const int32_t* rowptr, const int32_t* beg_ptr,
               const int32_t* end_ptr, const int32_t* colidx, double* values, const double* svalues, const double* X, const double* Y,
               double* Out, int64_t ld, int col0, int ncols, unsigned flags, const Extras& ex, bool run_long = true) {
    constexpr int GROUPS = kBlock / LPR;
    const int64_t blocks = (rows + GROUPS - 1) / GROUPS;
    if (blocks <= 0) return HNH_OK;
    if (blocks > 0x7fffffffLL) return hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "too many rows for one launch");
    if (lc.enabled) flags |= kInternalSplitLong | ((unsigned)(lc.threshold / 64) << kLongRowShift);
    const size_t lds_pad = lc.lds_pad;
    if (lds_pad > 48 * 1024)  // (only the measurement knob asks for that much)
        HNH_TRY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&row_kernel<OP, LPR, VEC, W, EXACT, NARROW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pad));
    hipLaunchKernelGGL((row_kernel<OP, LPR, VEC, W, EXACT, NARROW>), dim3((unsigned)blocks), dim3(kBlock), lds_pad, st, rows, rowptr, beg_ptr, end_ptr,
                       colidx, values, svalues, X, Y, Out, ld, col0, ncols, flags, ex);
    if (int rc = hnh::check_hip(ctx, hipGetLastError(), "row_kernel launch")) return rc;
    if (lc.enabled && run_long) {  // hub rows once per pass (after the last column panel), over their whole length
        // (the segments are plain fused work whatever epilogue the closing launch carries: the kFused instance)
        constexpr Op LOP = (OP == Op::kFusedCg) ? Op::kFused : OP;
        // 256 CUs x 4 resident workgroups; items are spread round-robin over all groups of the grid
        double* partials = (OP != Op::kSddmm) ? lc.partials : nullptr;
        hipLaunchKernelGGL((long_row_kernel<LOP, LPR, VEC, W, EXACT, NARROW>), dim3((unsigned)ctx->long_grid), dim3(kBlock), 0, st, lc.items, lc.count,
                           lc.capacity, rowptr, colidx, values, svalues, X, Y, Out, ld, col0, ncols, flags & ~kInternalEpilogue, ex, partials,
                           lc.partial_items);
        if (int rc = hnh::check_hip(ctx, hipGetLastError(), "long_row_kernel launch")) return rc;
        if (partials != nullptr) {
            const bool epi = fused_op(OP) && (flags & kInternalEpilogue);  // (dispatch_row grants it only for whole rows)
            // 16-byte accesses: to the output row, and with an epilogue to every row operand it reads or writes
            const bool pairs = (ld % 2 == 0) && (col0 % 2 == 0) && (ncols % 2 == 0) && aligned16(Out) &&
                               (!epi || (aligned16(X) && (ex.cg_x == nullptr || (aligned16(ex.cg_x) && aligned16(ex.cg_r) && aligned16(ex.cg_p))) &&
                                         (ex.relu_dst == nullptr || (aligned16(ex.relu_dst) && ex.relu_ld % 2 == 0))));
            if (pairs)
                hipLaunchKernelGGL(reduce_long_kernel<2>, dim3(2048), dim3(kBlock), 0, st, lc.hub_rows, lc.count, lc.capacity_rows, partials,
                                   lc.partial_items, Out, ld, col0, ncols, X, ex, epi);
            else
                hipLaunchKernelGGL(reduce_long_kernel<1>, dim3(2048), dim3(kBlock), 0, st, lc.hub_rows, lc.count, lc.capacity_rows, partials,
                                   lc.partial_items, Out, ld, col0, ncols, X, ex, epi);
            return hnh::check_hip(ctx, hipGetLastError(), "reduce_long_kernel launch");
        }
        return HNH_OK;
    }
    return HNH_OK;
}

// one launch of the instance that fits (shape, R) over the row pieces [beg_ptr[r], end_ptr[r]); returns -1 when R needs
// column tiles instead
template <Op OP>
int launch_shape(hnh_ctx* ctx, hipStream_t st, const LongCtl& lc, const Shape& s, int64_t rows, const int32_t* rowptr,
                 const int32_t* beg_ptr, const int32_t* end_ptr, const int32_t* colidx, double* values, const double* svalues,
                 const double* X, const double* Y, double* Out, int R, unsigned flags, const Extras& ex, bool run_long = true) {
#define HNH_CASE(L, V)                                                                                                     \
    if (s.lpr == L && s.vec == V)                                                                                          \
        return launch_row<OP, L, V, 2, true>(ctx, st, lc, rows, rowptr, beg_ptr, end_ptr, colidx, values, svalues, X, Y, Out, R, 0, R, flags, ex, run_long);
    if (s.exact) {
        // narrow rows with line-aligned CSR streams: the line-granular instances (process_row, NARROW)
        if (ctx->narrow_rows && (s.lpr == 4 || s.lpr == 8 || s.lpr == 16) && s.vec == 1 && aligned128(colidx) && aligned128(values) &&
            (svalues == nullptr || aligned128(svalues))) {
#define HNH_NARROW_CASE(L)                                                                                                 \
    if (s.lpr == L)                                                                                                        \
        return launch_row<OP, L, 1, 2, true, true>(ctx, st, lc, rows, rowptr, beg_ptr, end_ptr, colidx, values, svalues, X, Y, Out, R, 0, R, flags, ex, run_long);
            HNH_NARROW_CASE(4) HNH_NARROW_CASE(8) HNH_NARROW_CASE(16)
#undef HNH_NARROW_CASE
        }
        HNH_CASE(1, 1) HNH_CASE(2, 1) HNH_CASE(4, 1) HNH_CASE(8, 1) HNH_CASE(16, 1) HNH_CASE(32, 1) HNH_CASE(64, 1)
        HNH_CASE(64, 2) HNH_CASE(64, 3) HNH_CASE(64, 4) HNH_CASE(32, 3) HNH_CASE(32, 5) HNH_CASE(32, 7)
        return hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "no kernel instance for this shape");
    }
#undef HNH_CASE
    if constexpr (OP == Op::kFusedCg) return hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "the CG epilogue has exact-width instances only");
    // Widths that are not a supported exact multiple: one bounds-checked pass when the row fits the widest
    // instance (R <= 512 even / 256 odd)
#define HNH_NX(V, WW)                                                                                                      \
    if (s.w == WW && R <= 64 * WW * V)                                                                                     \
        return launch_row<OP, 64, V, WW, false>(ctx, st, lc, rows, rowptr, beg_ptr, end_ptr, colidx, values, svalues, X, Y, Out, R, 0, R, flags, ex, run_long);
    if constexpr (OP != Op::kFusedCg) {
        HNH_NX(1, 2) HNH_NX(2, 2) HNH_NX(4, 2) HNH_NX(1, 1) HNH_NX(2, 1) HNH_NX(4, 1)
    }
#undef HNH_NX
    return -1;
}

// launch_shape for the launch that completes the output rows: with the in-launch epilogue (kInternalEpilogue in `flags`) and CG
// updates requested, the kFusedCg instance runs instead of kFused
template <Op OP>
int launch_closing(hnh_ctx* ctx, hipStream_t st, const LongCtl& lc, const Shape& s, int64_t rows, const int32_t* rowptr,
                   const int32_t* beg_ptr, const int32_t* end_ptr, const int32_t* colidx, double* values, const double* svalues,
                   const double* X, const double* Y, double* Out, int R, unsigned flags, const Extras& ex, bool run_long = true) {
    if constexpr (OP == Op::kFused) {
        if ((flags & kInternalEpilogue) && (ex.cg_x != nullptr || ex.relu_dst != nullptr))
            return launch_shape<Op::kFusedCg>(ctx, st, lc, s, rows, rowptr, beg_ptr, end_ptr, colidx, values, svalues, X, Y, Out, R, flags, ex, run_long);
    }
    return launch_shape<OP>(ctx, st, lc, s, rows, rowptr, beg_ptr, end_ptr, colidx, values, svalues, X, Y, Out, R, flags, ex, run_long);
}

constexpr int kMaxPanels = 8;
int panel_count(const hnh_ctx* ctx, int64_t cols, int R) {
    if (cols <= 0 || ctx->no_panels) return 1;
    const long p = std::lround((double)cols * (double)R * sizeof(double) / ctx->panel_bytes);
    const long most = ctx->max_panels < kMaxPanels ? ctx->max_panels : kMaxPanels;
    return (int)(p < 1 ? 1 : (p > most ? most : p));
}

// cols: number of rows of the gathered operand (= columns of the sparse block), or < 0 when unknown (no panels)
template <Op OP>
int dispatch_row(hnh_ctx* ctx, hipStream_t st, int sidx, const Shape& s, int64_t rows, int64_t nnz, int max_row_nnz, int64_t cols,
                 const int32_t* rowptr, const int32_t* colidx, double* values, const double* svalues, const double* X,
                 const double* Y, double* Out, int R, unsigned flags, const Extras& ex = Extras(), bool* epilogue_done = nullptr,
                 const hnh_csr_window* win = nullptr, hnh_csr_plan* plan = nullptr) {
    if (plan != nullptr) {
        if (plan->rowptr == nullptr) {  // first use: the plan is this block's from now on
            plan->rowptr = rowptr;
            plan->colidx = colidx;
            plan->rows = rows;
            plan->nnz = nnz;
        } else if (plan->rowptr != rowptr || plan->colidx != colidx || plan->rows != rows || plan->nnz != nnz) {
            return hnh::fail(ctx, HNH_ERR_INVALID, "the structure plan belongs to another block");
        }
    }
    LongCtl lc;
    if (int rc = prepare_long(ctx, st, sidx, rows, rowptr, nnz, max_row_nnz, (OP != Op::kSddmm) ? (int64_t)R : 0, &lc,
                              win == nullptr || win->last != 0, plan))
        return rc;
    if (!lc.enabled || ctx->row_waves_cap > 0) lc.lds_pad = row_occupancy_pad(ctx, s, rows, nnz, max_row_nnz);  // (hub rows = a skewed block)
    const bool single_pass = s.exact || R <= 64 * s.w * 4;
    // the epilogue can ride in the launches that complete the rows: short rows are completed by ONE group of the row kernel, hub rows
    // by reduce_long_kernel once their segments' partial rows are added up (not when the segments combine with atomics); no
    // column tiles (and, for the CG updates, an exact-width instance)
    const bool extra_rows_ok = s.w == 1 || ((ex.cg_x == nullptr || (aligned16(ex.cg_x) && aligned16(ex.cg_r))) &&
                                            (ex.relu_dst == nullptr || (aligned16(ex.relu_dst) && ex.relu_ld % 2 == 0)));  // 16-byte row accesses
    const bool hubs_ok = !lc.enabled || (lc.partials != nullptr && lc.partial_items >= lc.capacity);
    const bool epilogue_in_launch = hubs_ok && single_pass && ((ex.cg_x == nullptr && ex.relu_dst == nullptr) || (s.exact && extra_rows_ok));
    if (win != nullptr) {
        // a caller-defined window of every row; hub rows stay whole and go to the long-row pass with the pass's last window,
        // which is also where a row epilogue can run inside the launch
        const int32_t* beg_ptr = win->beg ? win->beg : rowptr;
        const int32_t* end_ptr = win->end ? win->end : rowptr + 1;
        const bool last = win->last != 0;
        if (epilogue_done != nullptr) *epilogue_done = last && epilogue_in_launch;
        const unsigned f = flags | ((epilogue_done != nullptr && *epilogue_done) ? kInternalEpilogue : 0u);
        const int rcw = launch_closing<OP>(ctx, st, lc, s, rows, rowptr, beg_ptr, end_ptr, colidx, values, svalues, X, Y, Out, R, f, ex, last);
        if (rcw != -1) return rcw;
        if (OP == Op::kFused) return hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "fused fallback is composed by the caller");
        const int wtile = 64 * s.w;
        for (int col0 = 0; col0 < R; col0 += wtile) {
            const int ncols = (R - col0 < wtile) ? (R - col0) : wtile;
            int rc;
            const unsigned ft = (col0 == 0) ? flags : (flags & ~HNH_FUSED_VALUES_OVERWRITE);  // later tiles add their partial dot products
            if (s.w == 2)
                rc = launch_row<OP, 64, 1, 2, false>(ctx, st, lc, rows, rowptr, beg_ptr, end_ptr, colidx, values, svalues, X, Y, Out, R, col0, ncols, ft, ex, last);
            else
                rc = launch_row<OP, 64, 1, 1, false>(ctx, st, lc, rows, rowptr, beg_ptr, end_ptr, colidx, values, svalues, X, Y, Out, R, col0, ncols, ft, ex, last);
            if (rc != HNH_OK) return rc;
        }
        return HNH_OK;
    }
    // otherwise the caller appends row_epilogue_kernel
    if (epilogue_done != nullptr) *epilogue_done = epilogue_in_launch;
    const unsigned epi = (epilogue_done != nullptr && *epilogue_done) ? kInternalEpilogue : 0u;

    // Infinity-Cache panels (see panel_split_kernel): single-pass widths only.  The mechanism handles hub rows (they stay
    // whole and go to the long-row pass once, after the last panel; HNH_PANELS_WITH_HUBS=1), but on skewed graphs the hot
    // columns are cache-resident anyway and panels cost 2.5 % (R-MAT 2^20: 6.85 -> 7.03 ms), so such blocks keep one launch.
    // WIDE OPERANDS IN 128-COLUMN SLABS (round 6; un-fused SDDMM / SpMM, R >= 320 a multiple of 64: slabs of 128 columns and a last one of 64).  At those widths the gathered operand is
    // 3 - 4 GiB at config 2's size, nothing of it survives in the 256 MiB Infinity Cache and the pass runs at the DRAM copy rate (72 - 79 % of
    // 8 TB/s whatever the panel count).  A slab of 128 columns IS the R = 128 problem with row pitch R: its panels are again
    // 512 MiB of the gathered operand and every gather moves 1 KiB — so each slab runs as the R = 128 pass does (panels and all), slab after
    // slab.  SpMM slabs are independent columns of the output; the SDDMM's slabs after the first ADD their partial dot products (16 B per
    // nonzero and slab, and the index stream once more per slab: ~1 % of the slab's gathers).  The fused pass needs the whole dot before
    // its second half and keeps its single pass.  HNH_WIDE_SLABS=0 restores the single wide pass.  Measured at config 2's size
    // (profiles/r06_job5_kbench_wide_slabs.log): SpMM 84.2 -> 92.5 % of 8 TB/s at R = 384, 77.3 -> 91.9 % at R = 512; SDDMM 89.8 -> 92.1 %, 86.0 -> 92.0 %.
    const int slab_w = (ctx->wide_slabs && (OP == Op::kSddmm || OP == Op::kSpmm) && s.w == 2 && R >= ctx->slab_min_r && R % 64 == 0) ? 128 : 0;
    const int nslabs = slab_w ? (R + slab_w - 1) / slab_w : 1;
    const int panels = slab_w ? ((!lc.enabled || ctx->panels_with_hubs) ? panel_count(ctx, cols, slab_w) : 1)
                              : ((single_pass && (!lc.enabled || ctx->panels_with_hubs)) ? panel_count(ctx, cols, R) : 1);
    if (panels > 1 || slab_w) {
        const size_t need = (size_t)(panels - 1) * (size_t)rows * sizeof(int32_t);
        const int width = (int)((cols + panels - 1) / panels);
        int32_t* split = nullptr;
        if (panels == 1) {
            // (one panel per slab: the rows' own boundaries)
        } else if (plan != nullptr) {
            // the boundaries depend on the structure and (panels, width) only: computed once per block and width class
            hnh_csr_plan::Split* slot = nullptr;
            for (auto& sp : plan->splits)
                if (sp.split != nullptr && sp.panels == panels && sp.width == width) slot = &sp;
            if (slot == nullptr) {
                slot = &plan->splits[0];
                for (auto& sp : plan->splits)
                    if (sp.split == nullptr || (slot->split != nullptr && sp.age < slot->age)) slot = &sp;
                HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
                if (slot->split) HNH_TRY_HIP(ctx, hipFree(slot->split));
                slot->split = nullptr;
                HNH_TRY_HIP(ctx, hipMalloc((void**)&slot->split, need));
                slot->panels = panels;
                slot->width = width;
                hipLaunchKernelGGL(panel_split_kernel, dim3((unsigned)((rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, rows, rowptr, colidx,
                                   panels, width, slot->split);
                if (int rc = hnh::check_hip(ctx, hipGetLastError(), "panel_split_kernel launch")) return rc;
                HNH_TRY_HIP(ctx, hipStreamSynchronize(st));  // once per block and width class: later calls may run on any stream
            }
            slot->age = ++plan->clock;
            split = slot->split;
        } else {
            if (ctx->panel_cap[sidx] < need) {
                HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
                if (ctx->panel_split[sidx]) HNH_TRY_HIP(ctx, hipFree(ctx->panel_split[sidx]));
                ctx->panel_split[sidx] = nullptr;
                HNH_TRY_HIP(ctx, hipMalloc(&ctx->panel_split[sidx], need));
                ctx->panel_cap[sidx] = need;
            }
            split = static_cast<int32_t*>(ctx->panel_split[sidx]);
            hipLaunchKernelGGL(panel_split_kernel, dim3((unsigned)((rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, rows, rowptr, colidx, panels,
                               width, split);
            if (int rc = hnh::check_hip(ctx, hipGetLastError(), "panel_split_kernel launch")) return rc;
        }
        for (int sl = 0; sl < nslabs; sl++)
            for (int q = 0; q < panels; q++) {
                const int32_t* beg_ptr = (q == 0) ? rowptr : split + (size_t)(q - 1) * rows;
                const int32_t* end_ptr = (q == panels - 1) ? rowptr + 1 : split + (size_t)q * rows;
                unsigned f = flags;
                if (q > 0) f &= ~HNH_FUSED_OUT_OVERWRITE;  // later panels add to the rows the first one wrote
                if (q == panels - 1) f |= epi;             // the last panel completes the rows
                int rc;
                if (slab_w) {
                    if (sl > 0) f &= ~HNH_FUSED_VALUES_OVERWRITE;  // later slabs add their partial dot products
                    if (R - sl * slab_w >= slab_w)
                        rc = launch_row<OP, 64, 1, 2, true>(ctx, st, lc, rows, rowptr, beg_ptr, end_ptr, colidx, values, svalues, X, Y, Out, R, sl * slab_w,
                                                            slab_w, f, ex, q == panels - 1);
                    else  // (the last 64 columns of a width that is an odd multiple of 64: the R = 64 instance, two rows per wave)
                        rc = launch_row<OP, 32, 1, 2, true>(ctx, st, lc, rows, rowptr, beg_ptr, end_ptr, colidx, values, svalues, X, Y, Out, R, sl * slab_w, 64,
                                                            f, ex, q == panels - 1);
                } else {
                    rc = launch_closing<OP>(ctx, st, lc, s, rows, rowptr, beg_ptr, end_ptr, colidx, values, svalues, X, Y, Out, R, f, ex, q == panels - 1);
                }
                if (rc) return rc;
            }
        return HNH_OK;
    }

    const int rc1 = launch_closing<OP>(ctx, st, lc, s, rows, rowptr, rowptr, rowptr + 1, colidx, values, svalues, X, Y, Out, R, flags | epi, ex);
    if (rc1 != -1) return rc1;
    // ... else column tiles; SDDMM partial dot products accumulate into `values` tile by tile
    if (OP == Op::kFused) return hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "fused fallback is composed by the caller");
    const int tile = 64 * s.w;
    for (int col0 = 0; col0 < R; col0 += tile) {
        const int ncols = (R - col0 < tile) ? (R - col0) : tile;
        int rc;
        const unsigned ft = (col0 == 0) ? flags : (flags & ~HNH_FUSED_VALUES_OVERWRITE);  // later tiles add their partial dot products
        if (s.w == 2)
            rc = launch_row<OP, 64, 1, 2, false>(ctx, st, lc, rows, rowptr, rowptr, rowptr + 1, colidx, values, svalues, X, Y, Out, R, col0, ncols, ft, ex);
        else
            rc = launch_row<OP, 64, 1, 1, false>(ctx, st, lc, rows, rowptr, rowptr, rowptr + 1, colidx, values, svalues, X, Y, Out, R, col0, ncols, ft, ex);
        if (rc != HNH_OK) return rc;
    }
    return HNH_OK;
}

template <int LPR, int VEC, int W, bool EXACT>
int launch_coo(hnh_ctx* ctx, hipStream_t st, int64_t nnz, const int32_t* rowidx, const int32_t* colidx,
               double* values, const double* X, const double* Y, int64_t ld, int col0, int ncols) {
    constexpr int GROUPS = kBlock / LPR;
    constexpr int U = Unroll<LPR, VEC>::value > 4 ? 4 : Unroll<LPR, VEC>::value;
    int64_t blocks = (nnz + (int64_t)GROUPS * U - 1) / ((int64_t)GROUPS * U);
    if (blocks <= 0) return HNH_OK;
    if (blocks > (1 << 20)) blocks = 1 << 20;  // grid-stride beyond that
    hipLaunchKernelGGL((sddmm_coo_kernel<LPR, VEC, W, EXACT>), dim3((unsigned)blocks), dim3(kBlock), 0, st, nnz, rowidx,
                       colidx, values, X, Y, ld, col0, ncols);
    return hnh::check_hip(ctx, hipGetLastError(), "sddmm_coo_kernel launch");
}

int check_common(hnh_ctx* ctx, int64_t n, int R, const char* who) {
    if (n < 0) return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": negative size");
    if (R <= 0) return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": R must be positive");
    return HNH_OK;
}

}  // namespace

extern "C" {

int hnh_panel_count(hnh_ctx* ctx, int64_t rows, int64_t nnz, int64_t cols, int R, int max_row_nnz) {
    if (!ctx || R <= 0) return 1;
    if ((max_row_nnz < 0 || max_row_nnz > long_row_threshold(ctx, rows, nnz)) && !ctx->panels_with_hubs) return 1;  // hub rows (or unknown): one launch
    const Shape s = pick_shape(R, true);
    if (!(s.exact || R <= 64 * s.w * 4)) return 1;
    return panel_count(ctx, cols, R);
}

int hnh_sddmm_csr_ex(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                     const double* X, const double* Y, int R, int64_t nnz, int max_row_nnz, int64_t cols, int stream) {
    HNH_ENTER(ctx, stream);
    if (int rc = check_common(ctx, rows, R, "hnh_sddmm_csr")) return rc;
    if (rows == 0) return HNH_OK;
    if (!rowptr || !col_idx || !values || !X || !Y) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_sddmm_csr: null pointer");
    const Shape s = pick_shape(R, aligned16(X) && aligned16(Y));
    hnh::WideLaunch wide(ctx, stream);  // the stand-alone SDDMM does more arithmetic per byte than the other row passes: all CUs
    if (wide.status != HNH_OK) return wide.status;
    return wide.finish(dispatch_row<Op::kSddmm>(ctx, wide.stream(), stream, s, rows, nnz, max_row_nnz, cols, rowptr, col_idx, values, nullptr, X,
                                                Y, nullptr, R, 0u));
}

int hnh_sddmm_csr(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                  const double* X, const double* Y, int R, int stream) {
    return hnh_sddmm_csr_ex(ctx, rows, rowptr, col_idx, values, X, Y, R, -1, -1, -1, stream);
}

int hnh_spmm_csr(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, const double* values,
                 const double* X, double* Out, int R, int stream) {
    return hnh_spmm_csr_ex(ctx, rows, rowptr, col_idx, values, X, Out, R, -1, -1, -1, stream);
}

int hnh_spmm_csr_ex(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, const double* values,
                    const double* X, double* Out, int R, int64_t nnz, int max_row_nnz, int64_t cols, int stream) {
    HNH_ENTER(ctx, stream);
    if (int rc = check_common(ctx, rows, R, "hnh_spmm_csr")) return rc;
    if (rows == 0) return HNH_OK;
    if (!rowptr || !col_idx || !values || !X || !Out) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_spmm_csr: null pointer");
    if (X == Out) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_spmm_csr: X and Out alias");
    const Shape s = pick_shape(R, aligned16(X) && aligned16(Out));
    return dispatch_row<Op::kSpmm>(ctx, ctx->streams[stream], stream, s, rows, nnz, max_row_nnz, cols, rowptr, col_idx,
                                   const_cast<double*>(values), nullptr, X, nullptr, Out, R, 0u);
}

int hnh_fused_sddmm_spmm_csr(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                             const double* svalues, const double* X, const double* Y, double* Out, int R,
                             unsigned flags, int stream) {
    return hnh_fused_sddmm_spmm_csr_ex(ctx, rows, rowptr, col_idx, values, svalues, X, Y, Out, R, flags, -1, -1, -1, stream);
}

int hnh_fused_sddmm_spmm_csr_ex(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                                const double* svalues, const double* X, const double* Y, double* Out, int R,
                                unsigned flags, int64_t nnz_in, int max_row_nnz, int64_t cols, int stream) {
    return hnh_fused_sddmm_spmm_csr_x(ctx, rows, rowptr, col_idx, values, svalues, X, Y, Out, R, flags, nnz_in, max_row_nnz, cols, nullptr, stream);
}

}  // extern "C"

namespace {
// the row epilogue as its own launch (hub rows / column tiles / several launches per output row)
int launch_row_epilogue(hnh_ctx* ctx, hipStream_t st, double* Out, const double* X, const Extras& ex, int64_t rows, int R) {
    if (rows == 0 || (ex.x_scale == 0.0 && ex.rowdot == nullptr && ex.cg_x == nullptr && ex.relu_dst == nullptr)) return HNH_OK;
    const bool w2 = (R % 2 == 0) && aligned16(Out) && aligned16(X) && (ex.cg_x == nullptr || (aligned16(ex.cg_x) && aligned16(ex.cg_r))) &&
                    (ex.relu_dst == nullptr || (aligned16(ex.relu_dst) && ex.relu_ld % 2 == 0));
    const int chunks = w2 ? R / 2 : R;
#define HNH_EP(L)                                                                                                          \
    {                                                                                                                      \
        const int64_t blocks = (rows + (kBlock / L) - 1) / (kBlock / L);                                                   \
        if (w2) hipLaunchKernelGGL((row_epilogue_kernel<L, 2>), dim3((unsigned)blocks), dim3(kBlock), 0, st, Out, X, ex, rows, R); \
        else hipLaunchKernelGGL((row_epilogue_kernel<L, 1>), dim3((unsigned)blocks), dim3(kBlock), 0, st, Out, X, ex, rows, R);   \
    }
    if (chunks >= 64) HNH_EP(64) else if (chunks >= 16) HNH_EP(16) else if (chunks >= 4) HNH_EP(4) else HNH_EP(1)
#undef HNH_EP
    return hnh::check_hip(ctx, hipGetLastError(), "row_epilogue_kernel launch");
}

int fused_impl(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values, const double* svalues,
               const double* X, const double* Y, double* Out, int R, unsigned flags, int64_t nnz_in, int max_row_nnz, int64_t cols,
               const hnh_fused_extras* extras, const hnh_csr_window* win, int stream, hnh_csr_plan* plan = nullptr);

int check_extras(hnh_ctx* ctx, unsigned flags, const hnh_fused_extras* extras, const double* X, const double* Out, Extras* ex, bool* want_epilogue,
                 const char* who) {
    if (flags & ~(HNH_FUSED_VALUES_OVERWRITE | HNH_FUSED_OUT_OVERWRITE | HNH_FUSED_LEAKY_RELU))
        return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": unknown flag");
    if ((flags & HNH_FUSED_LEAKY_RELU) && !extras)
        return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": HNH_FUSED_LEAKY_RELU needs extras->leaky_alpha");
    if (extras) {
        ex->leaky_alpha = extras->leaky_alpha;
        ex->x_scale = extras->x_scale;
        ex->rowdot = extras->rowdot;
        if (const hnh_cg_update* cg = extras->cg) {
            if (!cg->x || !cg->r || !cg->p || !cg->rsold) return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": hnh_cg_update with a null pointer");
            if (X != nullptr && cg->p != X) return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": hnh_cg_update.p must be the row operand X");
            if (cg->x == cg->r || cg->x == cg->p || cg->r == cg->p || cg->x == Out || cg->r == Out || cg->p == Out)
                return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": hnh_cg_update operands alias");
            ex->cg_x = cg->x;
            ex->cg_r = cg->r;
            ex->cg_p = cg->p;
            ex->cg_rsold = cg->rsold;
            ex->cg_eps = cg->eps;
        }
        if (extras->relu_dst != nullptr) {
            if (extras->cg != nullptr) return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": relu_dst and cg exclude each other");
            if (extras->relu_ld <= 0) return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": relu_dst needs its row pitch relu_ld");
            if (extras->relu_dst == Out || extras->relu_dst == X) return hnh::fail(ctx, HNH_ERR_INVALID, std::string(who) + ": relu_dst aliases an operand");
            ex->relu_dst = extras->relu_dst;
            ex->relu_ld = extras->relu_ld;
        }
    }
    *want_epilogue = extras && (extras->x_scale != 0.0 || extras->rowdot != nullptr || extras->cg != nullptr || extras->relu_dst != nullptr);
    return HNH_OK;
}
}  // namespace

extern "C" {

int hnh_fused_sddmm_spmm_csr_x(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                               const double* svalues, const double* X, const double* Y, double* Out, int R, unsigned flags,
                               int64_t nnz_in, int max_row_nnz, int64_t cols, const hnh_fused_extras* extras, int stream) {
    return fused_impl(ctx, rows, rowptr, col_idx, values, svalues, X, Y, Out, R, flags, nnz_in, max_row_nnz, cols, extras, nullptr, stream);
}

int hnh_fused_sddmm_spmm_csr_w(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                               const double* svalues, const double* X, const double* Y, double* Out, int R, unsigned flags,
                               int64_t nnz_in, int max_row_nnz, const hnh_fused_extras* extras, const hnh_csr_window* window, int stream) {
    if (!window) return ctx ? hnh::fail(ctx, HNH_ERR_INVALID, "hnh_fused_sddmm_spmm_csr_w: null window") : HNH_ERR_INVALID;
    return fused_impl(ctx, rows, rowptr, col_idx, values, svalues, X, Y, Out, R, flags, nnz_in, max_row_nnz, -1, extras, window, stream);
}

}  // extern "C"

namespace {
int fused_impl(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values, const double* svalues,
               const double* X, const double* Y, double* Out, int R, unsigned flags, int64_t nnz_in, int max_row_nnz, int64_t cols,
               const hnh_fused_extras* extras, const hnh_csr_window* win, int stream, hnh_csr_plan* plan) {
    HNH_ENTER(ctx, stream);
    if (int rc = check_common(ctx, rows, R, "hnh_fused_sddmm_spmm_csr")) return rc;
    Extras ex;
    bool want_epilogue = false;
    if (int rc = check_extras(ctx, flags, extras, X, Out, &ex, &want_epilogue, "hnh_fused_sddmm_spmm_csr")) return rc;
    if (rows == 0) return HNH_OK;
    if (!rowptr || !col_idx || !values || !X || !Y || !Out)
        return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_fused_sddmm_spmm_csr: null pointer");
    if (X == Out || Y == Out) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_fused_sddmm_spmm_csr: Out aliases an input");
    hipStream_t st = ctx->streams[stream];
    const Shape s = pick_shape(R, aligned16(X) && aligned16(Y) && aligned16(Out));
    const bool closes_rows = (win == nullptr) || win->last != 0;  // the call that completes the output rows runs the epilogue
    if (want_epilogue && !closes_rows) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_fused_sddmm_spmm_csr_w: a row epilogue belongs to the last window");
    if (s.exact || R <= 256 * s.w) {  // one pass: an exact instance, or a bounds-checked one wide enough for the whole row
        bool done = false;
        if (int rc = dispatch_row<Op::kFused>(ctx, st, stream, s, rows, nnz_in, max_row_nnz, cols, rowptr, col_idx, values, svalues, X, Y, Out, R,
                                              flags, ex, want_epilogue ? &done : nullptr, win, plan))
            return rc;
        if (want_epilogue && !done) return launch_row_epilogue(ctx, st, Out, X, ex, rows, R);
        return HNH_OK;
    }
    // Tiled fallback (R odd or not a supported multiple): the dot product needs the whole row before the
    // axpy can start, so compose the two column-tiled passes; same arithmetic, one extra gather.
    int64_t nnz = nnz_in;
    if (nnz < 0) {
        int last = 0;
        HNH_TRY_HIP(ctx, hipMemcpyAsync(&last, rowptr + rows, sizeof(int), hipMemcpyDeviceToHost, st));
        HNH_TRY_HIP(ctx, hipStreamSynchronize(st));
        nnz = last;
    }
    // element-wise steps between the passes touch the window's values only (other windows hold finished or pending results)
    const int32_t* wbeg = (win && win->beg) ? win->beg : rowptr;
    const int32_t* wend = (win && win->end) ? win->end : rowptr + 1;
    auto on_values = [&](int mode) {
        if (win == nullptr) {
            if (mode == 0) return hnh::check_hip(ctx, hipMemsetAsync(values, 0, sizeof(double) * (size_t)nnz, st), "hipMemsetAsync");
            if (mode == 1) hipLaunchKernelGGL(leaky_relu_kernel, dim3(ew_grid(nnz)), dim3(kBlock), 0, st, values, ex.leaky_alpha, nnz);
            else hipLaunchKernelGGL(hadamard_kernel, dim3(ew_grid(nnz)), dim3(kBlock), 0, st, values, values, svalues, nnz, false);
        } else {
            const int64_t blocks = (rows * 64 + kBlock - 1) / kBlock;
            const int threshold = long_row_threshold(ctx, rows, nnz);  // as prepare_long decides for the row passes
            const int split_long = (max_row_nnz >= 0 && max_row_nnz <= threshold) ? 0 : threshold;
            hipLaunchKernelGGL(window_values_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, rows, rowptr, wbeg, wend, values, svalues, mode,
                               ex.leaky_alpha, split_long, win->last != 0);
        }
        return hnh::check_hip(ctx, hipGetLastError(), "element-wise pass of the fused fallback");
    };
    if (flags & HNH_FUSED_VALUES_OVERWRITE)
        if (int rc = on_values(0)) return rc;
    if (flags & HNH_FUSED_OUT_OVERWRITE) HNH_TRY_HIP(ctx, hipMemsetAsync(Out, 0, sizeof(double) * (size_t)rows * R, st));
    if (int rc = dispatch_row<Op::kSddmm>(ctx, st, stream, s, rows, nnz, max_row_nnz, cols, rowptr, col_idx, values, nullptr, X, Y, nullptr, R, 0u,
                                          Extras(), nullptr, win, plan))
        return rc;
    if (flags & HNH_FUSED_LEAKY_RELU) {
        if (svalues) {
            if (int rc = on_values(2)) return rc;
            svalues = nullptr;
        }
        if (int rc = on_values(1)) return rc;
    }
    if (int rc = dispatch_row<Op::kSpmm>(ctx, st, stream, s, rows, nnz, max_row_nnz, cols, rowptr, col_idx, values, svalues, Y, nullptr, Out, R, 0u,
                                         Extras(), nullptr, win, plan))
        return rc;
    if (want_epilogue) return launch_row_epilogue(ctx, st, Out, X, ex, rows, R);
    return HNH_OK;
}
}  // namespace

extern "C" {

int hnh_csr_window_bounds(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, int nbounds, const int32_t* bounds_host,
                          int32_t* split, int stream) {
    HNH_ENTER(ctx, stream);
    if (rows < 0 || nbounds < 0 || nbounds > kMaxWindowBounds) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_csr_window_bounds: bad size");
    if (rows == 0 || nbounds == 0) return HNH_OK;
    if (!rowptr || !col_idx || !bounds_host || !split) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_csr_window_bounds: null pointer");
    WindowBounds wb;
    wb.n = nbounds;
    for (int b = 0; b < nbounds; b++) {
        if (b > 0 && bounds_host[b] < bounds_host[b - 1]) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_csr_window_bounds: bounds must not decrease");
        wb.v[b] = bounds_host[b];
    }
    hipLaunchKernelGGL(window_bounds_kernel, dim3((unsigned)((rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->streams[stream], rows, rowptr,
                       col_idx, wb, split);
    return hnh::check_hip(ctx, hipGetLastError(), "window_bounds_kernel launch");
}

int hnh_sddmm_csr_w(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values, const double* X,
                    const double* Y, int R, int64_t nnz, int max_row_nnz, const hnh_csr_window* window, int stream) {
    HNH_ENTER(ctx, stream);
    if (int rc = check_common(ctx, rows, R, "hnh_sddmm_csr_w")) return rc;
    if (rows == 0) return HNH_OK;
    if (!rowptr || !col_idx || !values || !X || !Y || !window) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_sddmm_csr_w: null pointer");
    const Shape s = pick_shape(R, aligned16(X) && aligned16(Y));
    hnh::WideLaunch wide(ctx, stream);
    if (wide.status != HNH_OK) return wide.status;
    return wide.finish(dispatch_row<Op::kSddmm>(ctx, wide.stream(), stream, s, rows, nnz, max_row_nnz, -1, rowptr, col_idx, values, nullptr, X, Y,
                                                nullptr, R, 0u, Extras(), nullptr, window));
}

int hnh_spmm_csr_w(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, const double* values, const double* X,
                   double* Out, int R, int64_t nnz, int max_row_nnz, const hnh_csr_window* window, int stream) {
    HNH_ENTER(ctx, stream);
    if (int rc = check_common(ctx, rows, R, "hnh_spmm_csr_w")) return rc;
    if (rows == 0) return HNH_OK;
    if (!rowptr || !col_idx || !values || !X || !Out || !window) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_spmm_csr_w: null pointer");
    if (X == Out) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_spmm_csr_w: X and Out alias");
    const Shape s = pick_shape(R, aligned16(X) && aligned16(Out));
    return dispatch_row<Op::kSpmm>(ctx, ctx->streams[stream], stream, s, rows, nnz, max_row_nnz, -1, rowptr, col_idx,
                                   const_cast<double*>(values), nullptr, X, nullptr, Out, R, 0u, Extras(), nullptr, window);
}

int hnh_csr_plan_create(hnh_ctx* ctx, hnh_csr_plan** out) {
    if (!ctx || !out) return HNH_ERR_INVALID;
    *out = new (std::nothrow) hnh_csr_plan();
    return *out ? HNH_OK : hnh::fail(ctx, HNH_ERR_NOMEM, "hnh_csr_plan_create: out of memory");
}

int hnh_csr_plan_destroy(hnh_ctx* ctx, hnh_csr_plan* plan) {
    if (!ctx) return HNH_ERR_INVALID;
    if (!plan) return HNH_OK;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    for (int s = 0; s < HNH_STREAMS; s++)
        if (ctx->streams[s]) HNH_TRY_HIP(ctx, hipStreamSynchronize(ctx->streams[s]));
    if (ctx->wide) HNH_TRY_HIP(ctx, hipStreamSynchronize(ctx->wide));
    for (auto& sp : plan->splits)
        if (sp.split) (void)hipFree(sp.split);
    for (void* p : {(void*)plan->items, (void*)plan->count, (void*)plan->hub_rows})
        if (p) (void)hipFree(p);
    delete plan;
    return HNH_OK;
}

int hnh_sddmm_csr_p(hnh_ctx* ctx, const hnh_csr_block* b, double* values, const double* X, const double* Y, int R, unsigned flags,
                    const hnh_csr_window* window, int stream) {
    return hnh_sddmm_csr_ps(ctx, b, values, nullptr, X, Y, R, flags, window, stream);
}

int hnh_sddmm_csr_ps(hnh_ctx* ctx, const hnh_csr_block* b, double* values, const double* scale, const double* X, const double* Y, int R,
                     unsigned flags, const hnh_csr_window* window, int stream) {
    HNH_ENTER(ctx, stream);
    if (!b) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_sddmm_csr_p: null block");
    if (flags & ~HNH_FUSED_VALUES_OVERWRITE) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_sddmm_csr_p: unknown flag");
    if (int rc = check_common(ctx, b->rows, R, "hnh_sddmm_csr_p")) return rc;
    if (b->rows == 0) return HNH_OK;
    if (!b->rowptr || !b->col_idx || !values || !X || !Y) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_sddmm_csr_p: null pointer");
    if (scale == values) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_sddmm_csr_ps: scale aliases values");
    const Shape s = pick_shape(R, aligned16(X) && aligned16(Y));
    hnh::WideLaunch wide(ctx, stream);
    if (wide.status != HNH_OK) return wide.status;
    return wide.finish(dispatch_row<Op::kSddmm>(ctx, wide.stream(), stream, s, b->rows, b->nnz, b->max_row_nnz, window ? -1 : b->cols, b->rowptr, b->col_idx,
                                                values, scale, X, Y, nullptr, R, flags, Extras(), nullptr, window, b->plan));
}

int hnh_spmm_csr_pf(hnh_ctx* ctx, const hnh_csr_block* b, const double* values, const double* X, double* Out, int R, unsigned flags,
                    const hnh_csr_window* window, int stream) {
    HNH_ENTER(ctx, stream);
    if (!b) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_spmm_csr_p: null block");
    if (int rc = check_common(ctx, b->rows, R, "hnh_spmm_csr_p")) return rc;
    if (flags & ~HNH_FUSED_OUT_OVERWRITE) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_spmm_csr_pf: unknown flag");
    if ((flags & HNH_FUSED_OUT_OVERWRITE) && window) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_spmm_csr_pf: a window of a block cannot overwrite its rows");
    if (b->rows == 0) return HNH_OK;
    if (!b->rowptr || !b->col_idx || !values || !X || !Out) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_spmm_csr_p: null pointer");
    if (X == Out) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_spmm_csr_p: X and Out alias");
    const Shape s = pick_shape(R, aligned16(X) && aligned16(Out));
    return dispatch_row<Op::kSpmm>(ctx, ctx->streams[stream], stream, s, b->rows, b->nnz, b->max_row_nnz, window ? -1 : b->cols, b->rowptr, b->col_idx,
                                   const_cast<double*>(values), nullptr, X, nullptr, Out, R, flags, Extras(), nullptr, window, b->plan);
}

int hnh_spmm_csr_p(hnh_ctx* ctx, const hnh_csr_block* b, const double* values, const double* X, double* Out, int R, const hnh_csr_window* window,
                   int stream) {
    return hnh_spmm_csr_pf(ctx, b, values, X, Out, R, 0u, window, stream);
}

int hnh_fused_sddmm_spmm_csr_p(hnh_ctx* ctx, const hnh_csr_block* b, double* values, const double* svalues, const double* X, const double* Y,
                               double* Out, int R, unsigned flags, const hnh_fused_extras* extras, const hnh_csr_window* window, int stream) {
    if (!ctx) return HNH_ERR_INVALID;
    if (!b) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_fused_sddmm_spmm_csr_p: null block");
    return fused_impl(ctx, b->rows, b->rowptr, b->col_idx, values, svalues, X, Y, Out, R, flags, b->nnz, b->max_row_nnz, window ? -1 : b->cols, extras,
                      window, stream, b->plan);
}

int hnh_row_epilogue_f64(hnh_ctx* ctx, double* Out, const double* X, double x_scale, double* rowdot, int64_t rows, int R, int stream) {
    HNH_ENTER(ctx, stream);
    if (int rc = check_common(ctx, rows, R, "hnh_row_epilogue_f64")) return rc;
    if (rows == 0) return HNH_OK;
    if (!Out || !X || Out == X) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_row_epilogue_f64: bad operand");
    Extras ex;
    ex.x_scale = x_scale;
    ex.rowdot = rowdot;
    return launch_row_epilogue(ctx, ctx->streams[stream], Out, X, ex, rows, R);
}

int hnh_row_epilogue_x(hnh_ctx* ctx, double* Out, const double* X, const hnh_fused_extras* extras, int64_t rows, int R, int stream) {
    HNH_ENTER(ctx, stream);
    if (int rc = check_common(ctx, rows, R, "hnh_row_epilogue_x")) return rc;
    if (!extras) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_row_epilogue_x: null extras");
    Extras ex;
    bool want = false;
    if (int rc = check_extras(ctx, 0u, extras, X, Out, &ex, &want, "hnh_row_epilogue_x")) return rc;
    if (rows == 0 || !want) return HNH_OK;
    if (!Out || !X || Out == X) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_row_epilogue_x: bad operand");
    return launch_row_epilogue(ctx, ctx->streams[stream], Out, X, ex, rows, R);
}

int hnh_cg_step_f64(hnh_ctx* ctx, double* X, double* Rm, const double* P, const double* MP, const double* alpha, double* rsnew,
                    int64_t rows, int R, int stream) {
    HNH_ENTER(ctx, stream);
    if (int rc = check_common(ctx, rows, R, "hnh_cg_step_f64")) return rc;
    if (rows == 0) return HNH_OK;
    if (!X || !Rm || !P || !MP || !alpha || !rsnew) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_cg_step_f64: null pointer");
    hipStream_t st = ctx->streams[stream];
    const bool w2 = (R % 2 == 0) && aligned16(X) && aligned16(Rm) && aligned16(P) && aligned16(MP);
    const int chunks = w2 ? R / 2 : R;
#define HNH_CG(L)                                                                                                          \
    {                                                                                                                      \
        const int64_t blocks = (rows + (kBlock / L) - 1) / (kBlock / L);                                                   \
        if (w2) hipLaunchKernelGGL((cg_step_kernel<L, 2>), dim3((unsigned)blocks), dim3(kBlock), 0, st, X, Rm, P, MP, alpha, rsnew, rows, R); \
        else hipLaunchKernelGGL((cg_step_kernel<L, 1>), dim3((unsigned)blocks), dim3(kBlock), 0, st, X, Rm, P, MP, alpha, rsnew, rows, R);   \
    }
    if (chunks >= 64) HNH_CG(64) else if (chunks >= 16) HNH_CG(16) else if (chunks >= 4) HNH_CG(4) else HNH_CG(1)
#undef HNH_CG
    return hnh::check_hip(ctx, hipGetLastError(), "cg_step_kernel launch");
}

int hnh_sddmm_coo(hnh_ctx* ctx, int64_t nnz, const int32_t* row_idx, const int32_t* col_idx, double* values,
                  const double* X, const double* Y, int R, int stream) {
    HNH_ENTER(ctx, stream);
    if (int rc = check_common(ctx, nnz, R, "hnh_sddmm_coo")) return rc;
    if (nnz == 0) return HNH_OK;
    if (!row_idx || !col_idx || !values || !X || !Y) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_sddmm_coo: null pointer");
    hnh::WideLaunch wide(ctx, stream);
    if (wide.status != HNH_OK) return wide.status;
    hipStream_t st = wide.stream();
    const Shape s = pick_shape(R, aligned16(X) && aligned16(Y));
#define HNH_CASE(L, V) \
    if (s.lpr == L && s.vec == V) return wide.finish(launch_coo<L, V, 2, true>(ctx, st, nnz, row_idx, col_idx, values, X, Y, R, 0, R));
    if (s.exact) {
        HNH_CASE(1, 1) HNH_CASE(2, 1) HNH_CASE(4, 1) HNH_CASE(8, 1) HNH_CASE(16, 1) HNH_CASE(32, 1) HNH_CASE(64, 1)
        HNH_CASE(64, 2) HNH_CASE(64, 3) HNH_CASE(64, 4) HNH_CASE(32, 3) HNH_CASE(32, 5) HNH_CASE(32, 7)
        return wide.finish(hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "no kernel instance for this shape"));
    }
#undef HNH_CASE
    const int tile = 64 * s.w;
    for (int col0 = 0; col0 < R; col0 += tile) {
        const int ncols = (R - col0 < tile) ? (R - col0) : tile;
        int rc = (s.w == 2) ? launch_coo<64, 1, 2, false>(ctx, st, nnz, row_idx, col_idx, values, X, Y, R, col0, ncols)
                            : launch_coo<64, 1, 1, false>(ctx, st, nnz, row_idx, col_idx, values, X, Y, R, col0, ncols);
        if (rc != HNH_OK) return wide.finish(rc);
    }
    return wide.finish(HNH_OK);
}

int hnh_fill_f64(hnh_ctx* ctx, double* dst, int64_t n, double value, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_fill_f64: negative size");
    if (n == 0) return HNH_OK;
    if (!dst) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_fill_f64: null pointer");
    if (value == 0.0) return hnh::check_hip(ctx, hipMemsetAsync(dst, 0, sizeof(double) * (size_t)n, ctx->streams[stream]), "hipMemsetAsync");
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n / 2 / kEwUnroll + 1)), dim3(kBlock), 0, ctx->streams[stream], dst, n, value, aligned16(dst));
    return hnh::check_hip(ctx, hipGetLastError(), "fill_kernel launch");
}

int hnh_hadamard_f64(hnh_ctx* ctx, double* out, const double* a, const double* b, int64_t n, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_hadamard_f64: negative size");
    if (n == 0) return HNH_OK;
    if (!out || !a || !b) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_hadamard_f64: null pointer");
    hipLaunchKernelGGL(hadamard_kernel, dim3(ew_grid(n / 2 / kEwUnroll + 1)), dim3(kBlock), 0, ctx->streams[stream], out, a, b, n,
                       aligned16(out) && aligned16(a) && aligned16(b));
    return hnh::check_hip(ctx, hipGetLastError(), "hadamard_kernel launch");
}

int hnh_axpy_f64(hnh_ctx* ctx, double* y, const double* x, double alpha, int64_t n, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_axpy_f64: negative size");
    if (n == 0) return HNH_OK;
    if (!y || !x) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_axpy_f64: null pointer");
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n / 2 / kEwUnroll + 1)), dim3(kBlock), 0, ctx->streams[stream], y, x, alpha, n,
                       aligned16(y) && aligned16(x));
    return hnh::check_hip(ctx, hipGetLastError(), "axpy_kernel launch");
}

int hnh_sum_chunked_blocks_f64(hnh_ctx* ctx, double* dst, const double* src, int nblocks, int nchunks, const int64_t* cuts_host, int q0, int q1,
                               int R, int stream) {
    HNH_ENTER(ctx, stream);
    if (nblocks < 0 || nchunks < 1 || nchunks > HNH_MAX_CHUNKS || !cuts_host || q0 < 0 || q1 > nchunks || q0 > q1 || R <= 0)
        return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_sum_chunked_blocks_f64: bad argument");
    ChunkCuts cc;
    for (int q = 0; q <= nchunks; q++) {
        cc.cut[q] = cuts_host[q];
        if (cuts_host[q] < 0 || (q > 0 && cuts_host[q] < cuts_host[q - 1])) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_sum_chunked_blocks_f64: cuts must not decrease");
    }
    const int64_t rows = cc.cut[q1] - cc.cut[q0];
    if (rows == 0 || nblocks == 0) return HNH_OK;
    if (!dst || !src) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_sum_chunked_blocks_f64: null pointer");
    const bool vec = (R % 2 == 0) && aligned16(dst) && aligned16(src);
    const int64_t items = rows * (vec ? R / 2 : R);
    if (vec) hipLaunchKernelGGL(sum_chunks_kernel<true>, dim3(ew_grid(items)), dim3(kBlock), 0, ctx->streams[stream], dst, src, nblocks, cc, q0, q1, R);
    else hipLaunchKernelGGL(sum_chunks_kernel<false>, dim3(ew_grid(items)), dim3(kBlock), 0, ctx->streams[stream], dst, src, nblocks, cc, q0, q1, R);
    return hnh::check_hip(ctx, hipGetLastError(), "sum_chunks_kernel launch");
}

int hnh_rowdot_f64(hnh_ctx* ctx, const double* A, const double* B, double* out, int64_t rows, int R, int stream) {
    HNH_ENTER(ctx, stream);
    if (int rc = check_common(ctx, rows, R, "hnh_rowdot_f64")) return rc;
    if (rows == 0) return HNH_OK;
    if (!A || !B || !out) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_rowdot_f64: null pointer");
    hipStream_t st = ctx->streams[stream];
    const bool w2 = (R % 2 == 0) && aligned16(A) && aligned16(B);
    const int chunks = w2 ? R / 2 : R;
#define HNH_RD(L)                                                                                                   \
    {                                                                                                               \
        const int64_t blocks = (rows + (kBlock / L) - 1) / (kBlock / L);                                            \
        if (w2) hipLaunchKernelGGL((rowdot_kernel<L, 2>), dim3((unsigned)blocks), dim3(kBlock), 0, st, A, B, out, rows, R); \
        else hipLaunchKernelGGL((rowdot_kernel<L, 1>), dim3((unsigned)blocks), dim3(kBlock), 0, st, A, B, out, rows, R);   \
    }
    if (chunks >= 64) HNH_RD(64) else if (chunks >= 32) HNH_RD(32) else if (chunks >= 16) HNH_RD(16) else if (chunks >= 8) HNH_RD(8)
    else if (chunks >= 4) HNH_RD(4) else if (chunks >= 2) HNH_RD(2) else HNH_RD(1)
#undef HNH_RD
    return hnh::check_hip(ctx, hipGetLastError(), "rowdot_kernel launch");
}

int hnh_row_scale_add_f64(hnh_ctx* ctx, double* Y, const double* yv, double ya, const double* X, const double* xv, double xa,
                          int64_t rows, int R, int stream) {
    HNH_ENTER(ctx, stream);
    if (int rc = check_common(ctx, rows, R, "hnh_row_scale_add_f64")) return rc;
    if (rows == 0) return HNH_OK;
    if (!Y || !X) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_row_scale_add_f64: null pointer");
    hipStream_t st = ctx->streams[stream];
    if ((R % 2 == 0) && aligned16(Y) && aligned16(X))
        hipLaunchKernelGGL((row_scale_add_kernel<2>), dim3(ew_grid(rows * R / 2 / kEwUnroll + 1)), dim3(kBlock), 0, st, Y, yv, ya, X, xv, xa, rows, R);
    else
        hipLaunchKernelGGL((row_scale_add_kernel<1>), dim3(ew_grid(rows * R / kEwUnroll + 1)), dim3(kBlock), 0, st, Y, yv, ya, X, xv, xa, rows, R);
    return hnh::check_hip(ctx, hipGetLastError(), "row_scale_add_kernel launch");
}

int hnh_vec_add_scalar_f64(hnh_ctx* ctx, double* v, double c, int64_t n, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_vec_add_scalar_f64: negative size");
    if (n == 0) return HNH_OK;
    if (!v) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_vec_add_scalar_f64: null pointer");
    hipLaunchKernelGGL(vec_add_scalar_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, ctx->streams[stream], v, c, n);
    return hnh::check_hip(ctx, hipGetLastError(), "vec_add_scalar_kernel launch");
}

int hnh_fill_hashed_f64(hnh_ctx* ctx, double* dst, int64_t rows, int64_t cols, int64_t top_row, int64_t left_col, int64_t R_global,
                        uint64_t seed, double scale, int stream) {
    HNH_ENTER(ctx, stream);
    if (rows < 0 || cols < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_fill_hashed_f64: negative size");
    if (rows == 0 || cols == 0) return HNH_OK;
    if (!dst) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_fill_hashed_f64: null pointer");
    hipLaunchKernelGGL(fill_hashed_kernel, dim3(ew_grid(rows * cols)), dim3(kBlock), 0, ctx->streams[stream], dst, rows, cols, top_row,
                       left_col, R_global, (unsigned long long)seed, scale);
    return hnh::check_hip(ctx, hipGetLastError(), "fill_hashed_kernel launch");
}

int hnh_vec_div_f64(hnh_ctx* ctx, double* out, const double* num, const double* den, int64_t n, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_vec_div_f64: negative size");
    if (n == 0) return HNH_OK;
    if (!out || !num || !den) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_vec_div_f64: null pointer");
    hipLaunchKernelGGL(vec_div_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, ctx->streams[stream], out, num, den, n);
    return hnh::check_hip(ctx, hipGetLastError(), "vec_div_kernel launch");
}

int hnh_gemm_f64(hnh_ctx* ctx, int64_t M, int64_t N, int64_t K, const double* A, const double* B, double* C, int stream) {
    HNH_ENTER(ctx, stream);
    if (M < 0 || N < 0 || K < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_gemm_f64: negative size");
    if (M == 0 || N == 0) return HNH_OK;
    if (!C || (K > 0 && (!A || !B))) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_gemm_f64: null pointer");
    const bool tall = ctx->gemm_waves == 8;
    const int64_t bm = tall ? 2 * kGemmBM : kGemmBM;
    const int64_t row_blocks = (M + bm - 1) / bm, col_blocks = (N + kGemmBN - 1) / kGemmBN;
    const int64_t grid = ((row_blocks + kXcds - 1) / kXcds) * kXcds * col_blocks;  // row blocks padded to whole XCD rounds
    if (grid > 0x7fffffffLL || col_blocks > 0x7fffffffLL) return hnh::fail(ctx, HNH_ERR_UNSUPPORTED, "hnh_gemm_f64: matrix too large");
    const bool vec_ok = (K % 2 == 0) && (N % 2 == 0) && aligned16(A) && aligned16(B);
    hnh::WideLaunch wide(ctx, stream);  // a dense contraction wants every matrix core
    if (wide.status != HNH_OK) return wide.status;
    const unsigned extra = (unsigned)ctx->gemm_lds_extra;  // (an LDS request the kernel never touches: caps its workgroups per CU)
    if (tall)
        hipLaunchKernelGGL(gemm_f64_kernel<4>, dim3((unsigned)grid), dim3(512), extra, wide.stream(), M, N, K, A, B, C, row_blocks, (int)col_blocks, vec_ok);
    else
        hipLaunchKernelGGL(gemm_f64_kernel<2>, dim3((unsigned)grid), dim3(256), extra, wide.stream(), M, N, K, A, B, C, row_blocks, (int)col_blocks, vec_ok);
    return wide.finish(hnh::check_hip(ctx, hipGetLastError(), "gemm_f64_kernel launch"));
}

int hnh_leaky_relu_f64(hnh_ctx* ctx, double* v, double alpha, int64_t n, int stream) {
    HNH_ENTER(ctx, stream);
    if (n < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_leaky_relu_f64: negative size");
    if (n == 0) return HNH_OK;
    if (!v) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_leaky_relu_f64: null pointer");
    hipLaunchKernelGGL(leaky_relu_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, ctx->streams[stream], v, alpha, n);
    return hnh::check_hip(ctx, hipGetLastError(), "leaky_relu_kernel launch");
}

int hnh_relu_store_cols_f64(hnh_ctx* ctx, double* dst, int64_t ld_dst, int64_t col0, const double* src, int64_t rows,
                            int64_t cols, int stream) {
    HNH_ENTER(ctx, stream);
    if (rows < 0 || cols < 0 || col0 < 0 || col0 + cols > ld_dst) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_relu_store_cols_f64: bad shape");
    if (rows == 0 || cols == 0) return HNH_OK;
    if (!dst || !src) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_relu_store_cols_f64: null pointer");
    hipLaunchKernelGGL(relu_store_cols_kernel, dim3(ew_grid(rows * cols)), dim3(kBlock), 0, ctx->streams[stream], dst, ld_dst, col0,
                       src, rows, cols);
    return hnh::check_hip(ctx, hipGetLastError(), "relu_store_cols_kernel launch");
}

int hnh_csr_max_row_nnz(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, int* out_host, int stream) {
    HNH_ENTER(ctx, stream);
    if (rows < 0 || !out_host) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_csr_max_row_nnz: bad argument");
    *out_host = 0;
    if (rows == 0) return HNH_OK;
    if (!rowptr) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_csr_max_row_nnz: null pointer");
    hipStream_t st = ctx->streams[stream];
    if (!ctx->long_count[stream]) HNH_TRY_HIP(ctx, hipMalloc((void**)&ctx->long_count[stream], 2 * sizeof(int)));  // shared with prepare_long (two counters)
    HNH_TRY_HIP(ctx, hipMemsetAsync(ctx->long_count[stream], 0, sizeof(int), st));
    hipLaunchKernelGGL(max_row_nnz_kernel, dim3(ew_grid(rows)), dim3(kBlock), 0, st, rows, rowptr, ctx->long_count[stream]);
    HNH_TRY_HIP(ctx, hipGetLastError());
    HNH_TRY_HIP(ctx, hipMemcpyAsync(out_host, ctx->long_count[stream], sizeof(int), hipMemcpyDeviceToHost, st));
    return hnh::check_hip(ctx, hipStreamSynchronize(st), "hipStreamSynchronize");
}

int hnh_expand_rowptr(hnh_ctx* ctx, int64_t rows, const int32_t* rowptr, int32_t* row_idx, int stream) {
    HNH_ENTER(ctx, stream);
    if (rows < 0) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_expand_rowptr: negative size");
    if (rows == 0) return HNH_OK;
    if (!rowptr || !row_idx) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_expand_rowptr: null pointer");
    const int64_t blocks = (rows * 64 + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(expand_rowptr_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->streams[stream], rows, rowptr, row_idx);
    return hnh::check_hip(ctx, hipGetLastError(), "expand_rowptr_kernel launch");
}

}  // extern "C"
