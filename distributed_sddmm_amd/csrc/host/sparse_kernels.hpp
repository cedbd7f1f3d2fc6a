// Local-kernel plugin point — same surface as the reference's sparse_kernels.h:
//   class KernelImplementation { virtual sddmm_local, virtual spmm_local, triple_function }  (:15-79)
//   class StandardKernel : KernelImplementation                                              (:84-99)
// Every schedule receives a KernelImplementation* and calls it only through triple_function()
// (15D_dense_shift.hpp:343-349 etc.), so user plugins (README.md:17-18) keep working.  StandardKernel's
// bodies marshal to the hand-written HIP kernels behind the C ABI of include/hnh_kernels.h.
//
// One addition: fused_local(), the back-to-back sddmm/spmm pair that the "local kernel fusion" schedule
// issues per visiting block (15D_dense_shift.hpp:203-217).  The default implementation is literally that
// pair of virtual calls; StandardKernel overrides it with the single-pass HIP kernel.
#pragma once
#include <vector>

#include "common.hpp"
#include "dense.hpp"
#include "spmat_local.hpp"

using hnh::DenseMatrix;
using hnh::VectorXd;

class KernelImplementation {
public:
    virtual ~KernelImplementation() {}

    // Performs an operation that looks like a local SDDMM and returns the number of nonzeros processed
    virtual size_t sddmm_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, int block, int offset) = 0;

    // S is m x n, A is m x r, B is n x r.  Amat: A += S B.  Bmat: B += S^T A (block stored transposed).
    virtual size_t spmm_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, MatMode mode, int block) = 0;

    // values(block) += <A rows, B rows>;  Out += S_block(values) * B        (Out must not alias A or B)
    // flags: HNH_FUSED_VALUES_OVERWRITE / HNH_FUSED_OUT_OVERWRITE tell the implementation that the block's
    // values / Out are to be treated as zero on entry, so it may overwrite instead of read-modify-write.
    // extras (optional, hnh_kernels.h): with HNH_FUSED_LEAKY_RELU the block's values are activated between the two
    // halves; x_scale / rowdot are applied to the finished rows of Out — ALWAYS, also when the block is absent —
    // so a schedule hands them to exactly one call: the last one that touches Out.
    // Default: literally the two virtual calls the reference makes (15D_dense_shift.hpp:203-217).
    virtual size_t fused_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, DenseMatrix& Out, int block, unsigned flags,
                               const hnh_fused_extras* extras = nullptr) {
        hnh::World* w = S.world;
        CSRLocal* blk = S.csr_blocks[block];
        size_t n = 0;
        if (flags & HNH_FUSED_OUT_OVERWRITE) Out.setZero();
        if (blk != nullptr) {
            if (flags & HNH_FUSED_VALUES_OVERWRITE)
                w->check(w->be->hnh_fill_f64(w->ctx, blk->getActive()->values, blk->num_coords, 0.0, HNH_STREAM_COMPUTE), "hnh_fill_f64");
            n += sddmm_local(S, A, B, block, 0);
            if (flags & HNH_FUSED_LEAKY_RELU)
                w->check(w->be->hnh_leaky_relu_f64(w->ctx, blk->getActive()->values, extras->leaky_alpha, blk->num_coords, HNH_STREAM_COMPUTE),
                         "hnh_leaky_relu_f64");
            n += spmm_local(S, Out, B, Amat, block);
        }
        row_epilogue(w, A, Out, extras);
        return n;
    }

    // Row windows (CSRLocal::window): a schedule may select one column range of a block before calling the kernels, to
    // work on data that arrives piece by piece.  An implementation that honours CSRLocal::window says so here; for the
    // others (plugins written against the reference's two pure virtuals) the schedule waits for the whole block instead.
    virtual bool handles_windows() const { return false; }
    // Window RANGES (CSRLocal::window .. window_end, round 5's adaptive windows): the schedule may select SEVERAL consecutive windows
    // for one pass — [window, window_end) — when their data has landed together.  Only an implementation that reads the selection
    // through CSRLocal::window_args() (or honours window_end itself) may say so; a plugin that reads CSRLocal::window alone keeps
    // the default and gets exactly one window per pass (window_end == window + 1 always), as in rounds 2-4.
    virtual bool handles_window_ranges() const { return false; }

    // Row parts (CSRLocal::row_part): the SpMM of the selected half of a block's rows.  An implementation that honours it says so
    // here; for the others the schedule keeps the whole-block step (kernel, then shift).
    virtual bool handles_row_parts() const { return false; }

    // An implementation that honours CSRLocal::values_fresh says so here: the schedules then skip zeroing the block values before
    // an SDDMM (distributed_sparse.h:280's setValuesConstant) and mark every first visit instead.
    virtual bool overwrites_fresh_values() const { return false; }

    // An implementation that honours CSRLocal::out_fresh says so here: the SpMM of rows whose output nobody has written yet STORES its sums,
    // and the schedules skip zeroing such an output buffer first.
    virtual bool stores_fresh_output() const { return false; }

    // Row RANGES (CSRLocal::select_row_range): SDDMM / SpMM of rows [r0, r1) of a block — the row operand and the output are addressed
    // from the range's first row (CSRLocal::part_first_row()).  An implementation that honours it says so here; for the others the 1.5D
    // replication-reuse schedule keeps the reference's block-by-block ring.
    virtual bool handles_row_ranges() const { return false; }

    // An implementation that honours CSRLocal::spmm_values and CSRLocal::sddmm_dst / sddmm_scale says so here: the schedules then
    // let stationary blocks read SValues in place (no setCSRValues copy) and write `SValues .* dots` straight into the result of an
    // SDDMM (no closing Hadamard pass).  Requires overwrites_fresh_values().
    virtual bool borrows_value_arrays() const { return false; }

    static bool wants_epilogue(const hnh_fused_extras* extras) {
        return extras && (extras->x_scale != 0.0 || extras->rowdot != nullptr || extras->cg != nullptr || extras->relu_dst != nullptr);
    }
    static void row_epilogue(hnh::World* w, DenseMatrix& X, DenseMatrix& Out, const hnh_fused_extras* extras) {
        if (!wants_epilogue(extras)) return;
        w->check(w->be->hnh_row_epilogue_x(w->ctx, Out.data(), X.data(), extras, Out.rows(), (int)Out.cols(), HNH_STREAM_COMPUTE),
                 "hnh_row_epilogue_x");
    }

    size_t triple_function(KernelMode mode, SpmatLocal& S, DenseMatrix& localA, DenseMatrix& localB, int block, int offset) {
        size_t nnz_processed = 0;
        if (mode == k_sddmmA || mode == k_sddmmB) nnz_processed += sddmm_local(S, localA, localB, block, offset);
        else if (mode == k_spmmA) nnz_processed += spmm_local(S, localA, localB, Amat, block);
        else if (mode == k_spmmB) nnz_processed += spmm_local(S, localA, localB, Bmat, block);
        return nnz_processed;
    }
};

// Exactly the algebra on the box (sparse_kernels.h:81-83), on the GPU.
class StandardKernel : public KernelImplementation {
public:
    // Accumulated device time of the kernels launched through this object, measured with HIP events on
    // the compute stream when profiling is enabled (bench.py's roofline leg).  The event pairs are only RECORDED while the calls
    // run — the host never waits inside a profiled call, so a schedule whose control flow depends on what has completed when
    // (the adaptive windows of the 1.5D dense shift) makes the same decisions profiled as timed — and are read by resolve_profile().
    bool profile = false;
    double kernel_ms = 0.0;
    long kernel_launches = 0;
    void resolve_profile();  // waits for the recorded pairs and adds their elapsed times to kernel_ms

    bool handles_windows() const override { return true; }
    bool handles_window_ranges() const override { return true; }
    bool overwrites_fresh_values() const override { return true; }
    bool handles_row_parts() const override { return true; }
    bool handles_row_ranges() const override { return true; }
    bool stores_fresh_output() const override { return true; }
    bool borrows_value_arrays() const override { return true; }
    size_t sddmm_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, int block, int offset) override;
    size_t spmm_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, MatMode mode, int block) override;
    size_t fused_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, DenseMatrix& Out, int block, unsigned flags,
                       const hnh_fused_extras* extras = nullptr) override;
    ~StandardKernel() override;

private:
    std::vector<std::pair<void*, void*>> pairs_;  // (start, stop) events; the first `used_` are recorded and not yet read
    size_t used_ = 0;
    void* tick_ = nullptr;  // 8 device bytes: the tiny dispatch in front of every start event (see begin())
    hnh::World* evw_ = nullptr;
    void begin(hnh::World* w);
    void end(hnh::World* w, long launches = 1);
};
