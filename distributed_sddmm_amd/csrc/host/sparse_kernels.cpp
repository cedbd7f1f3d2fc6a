#include "sparse_kernels.hpp"

// SDDMM of operands up to this many columns goes through the nonzero-balanced COO kernel (StandardKernel::sddmm_local)
static constexpr int64_t kCooSddmmMaxWidth = 16;

// StandardKernel::sddmm_local — sparse_kernels.cpp:13-57 of the reference:
//   values[i] += <Arow(row_idx[i]), Brow(col_idx[i])>, with A and B swapped when the block is stored
//   transposed (:29-37); a null block is a no-op (:25-27); always reports 0 nonzeros (:23,56).
size_t StandardKernel::sddmm_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, int block, int offset) {
    (void)offset;  // ignored by the reference as well (15D_sparse_shift.hpp:245 TODO)
    if (A.cols() != B.cols()) hnh::fatal("Error, SDDMM operands must have the same number of columns!");
    size_t processed = 0;
    CSRLocal* blk = S.csr_blocks[block];
    if (blk == nullptr || blk->num_coords == 0) return processed;
    double* Xptr = (blk->transpose ? B.data() : A.data()) + blk->part_first_row() * A.cols();  // (a row range reads its own rows of the row operand)
    double* Yptr = blk->transpose ? A.data() : B.data();
    CSRHandle* active = blk->getActive();
    hnh::World* w = S.world;
    begin(w);
    hnh_csr_window win;
    const hnh_csr_block desc = blk->block_args();
    const unsigned fresh = blk->values_fresh ? HNH_FUSED_VALUES_OVERWRITE : 0u;  // first visit: store instead of read-add-store
    // (a lent destination: the products scale .* dots go straight into the caller's result vector, CSRLocal::sddmm_dst)
    double* dst = blk->sddmm_dst ? blk->sddmm_dst : active->values;
    const double* scale = blk->sddmm_dst ? blk->sddmm_scale : nullptr;
    if (blk->window_args(&win)) {  // one column range of the block (the schedule walks them as their data arrives)
        w->check(w->be->hnh_sddmm_csr_ps(w->ctx, &desc, dst, scale, Xptr, Yptr, (int)A.cols(), fresh, &win, HNH_STREAM_COMPUTE), "hnh_sddmm_csr_ps");
        end(w);
        return processed;
    }
    if (A.cols() <= kCooSddmmMaxWidth && !fresh && scale == nullptr && blk->range_sel < 0) {  // (the COO kernel accumulates: first visits take the storing CSR pass)
        // narrow operands, ACCUMULATING visit (the travelling blocks of 15d_sparse / 2.5D dense after their first step): several sparse
        // rows share a wave in the row kernel and the wave runs as long as its longest row; the COO kernel deals nonzeros out evenly
        // instead.  Re-measured in round 5 with the line-granular row loop (config-2 size, profiles/r05_kbench_narrow.log): accumulating
        // row pass 2.062 / 2.132 ms at R = 8 / 16 against 2.025 / 2.055 ms here — the row loop already reads the value line with the
        // index line and writes whole lines, what it cannot shed is the longest-row effect — so these two widths stay on this kernel
        // (1.8 % / 3.7 %); STORING visits (every first visit) are faster through the row pass (1.925 / 2.003 ms) and never come here; from
        // R = 32 the row pass wins either way (3.825 vs 3.978 ms)
        const int32_t* row_idx = blk->ensure_row_idx(HNH_STREAM_COMPUTE);
        w->check(w->be->hnh_sddmm_coo(w->ctx, blk->num_coords, row_idx, active->col_idx, active->values, Xptr, Yptr, (int)A.cols(),
                                      HNH_STREAM_COMPUTE),
                 "hnh_sddmm_coo");
        end(w);
        return processed;
    }
    w->check(w->be->hnh_sddmm_csr_ps(w->ctx, &desc, dst, scale, Xptr, Yptr, (int)A.cols(), fresh, nullptr, HNH_STREAM_COMPUTE), "hnh_sddmm_csr_ps");
    end(w, profile ? w->be->hnh_panel_count(w->ctx, desc.rows, desc.nnz, desc.cols, (int)A.cols(), desc.max_row_nnz) : 1);
    return processed;
}

// StandardKernel::spmm_local — sparse_kernels.cpp:59-127: C += S_blk * X with alpha = beta = 1.
size_t StandardKernel::spmm_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, MatMode mode, int block) {
    size_t processed = 0;
    CSRLocal* blk = S.csr_blocks[block];
    if (blk == nullptr) return processed;
    if (mode == Amat && blk->transpose) hnh::fatal("Error, local matrix is transposed, can't perform SpmmA");
    else if (mode == Bmat && !blk->transpose) hnh::fatal("Error, local matrix is not transposed, can't perform SpmmB");
    CSRHandle* active = blk->getActive();
    hnh::World* w = S.world;
    const double* X = (mode == Amat) ? B.data() : A.data();
    double* Out = ((mode == Amat) ? A.data() : B.data()) + blk->part_first_row() * A.cols();  // (a row part writes its own rows of the output)
    if (blk->num_coords == 0) {  // nothing to multiply; fresh output rows still have to hold zeros afterwards
        if (blk->out_fresh) {
            const hnh_csr_block d0 = blk->block_args();
            w->check(w->be->hnh_fill_f64(w->ctx, Out, d0.rows * A.cols(), 0.0, HNH_STREAM_COMPUTE), "hnh_fill_f64");
        }
        return processed;
    }
    const double* vals = blk->spmm_values ? blk->spmm_values : active->values;  // (lent: the caller's SValues slice, read in place)
    begin(w);
    hnh_csr_window win;
    const hnh_csr_block desc = blk->block_args();
    if (blk->window_args(&win)) {
        w->check(w->be->hnh_spmm_csr_p(w->ctx, &desc, vals, X, Out, (int)A.cols(), &win, HNH_STREAM_COMPUTE), "hnh_spmm_csr_p");
        end(w);
        return processed;
    }
    // (a fresh output: the selected rows of Out are STORED, not added to — the staging rows of the mesh reduce-scatter are written once)
    w->check(w->be->hnh_spmm_csr_pf(w->ctx, &desc, vals, X, Out, (int)A.cols(), blk->out_fresh ? HNH_FUSED_OUT_OVERWRITE : 0u, nullptr, HNH_STREAM_COMPUTE),
             "hnh_spmm_csr_pf");
    end(w, profile ? w->be->hnh_panel_count(w->ctx, desc.rows, desc.nnz, desc.cols, (int)A.cols(), desc.max_row_nnz) : 1);
    return processed;
}

// One pass for the sddmm/spmm pair of 15D_dense_shift.hpp:203-217 (block not transposed: approach 2).
size_t StandardKernel::fused_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, DenseMatrix& Out, int block, unsigned flags,
                                   const hnh_fused_extras* extras) {
    if (A.cols() != B.cols() || Out.cols() != A.cols()) hnh::fatal("Error, fused operands must have the same number of columns!");
    if ((flags & HNH_FUSED_LEAKY_RELU) && !extras) hnh::fatal("Error, HNH_FUSED_LEAKY_RELU needs extras!");
    CSRLocal* blk = S.csr_blocks[block];
    hnh::World* w = S.world;
    if (blk == nullptr || blk->num_coords == 0) {  // nothing to multiply; the flags' contract and the row epilogue still apply
        if (flags & HNH_FUSED_OUT_OVERWRITE) Out.setZero();  // "treat Out as zero on entry": it must not keep stale rows
        row_epilogue(w, A, Out, extras);
        return 0;
    }
    if (blk->transpose) hnh::fatal("Error, local matrix is transposed, can't perform the fused SDDMM+SpMM");
    CSRHandle* active = blk->getActive();
    begin(w);
    hnh_csr_window win;
    const hnh_csr_block desc = blk->block_args();
    if (blk->window_args(&win)) {
        if (wants_epilogue(extras) && !win.last) hnh::fatal("Error, the row epilogue belongs to the block's last window!");
        w->check(w->be->hnh_fused_sddmm_spmm_csr_p(w->ctx, &desc, active->values, nullptr, A.data(), B.data(), Out.data(), (int)A.cols(), flags, extras,
                                                   &win, HNH_STREAM_COMPUTE),
                 "hnh_fused_sddmm_spmm_csr_p");
        end(w);
        return 0;
    }
    w->check(w->be->hnh_fused_sddmm_spmm_csr_p(w->ctx, &desc, active->values, nullptr, A.data(), B.data(), Out.data(), (int)A.cols(), flags, extras,
                                               nullptr, HNH_STREAM_COMPUTE),
             "hnh_fused_sddmm_spmm_csr_p");
    end(w, profile ? w->be->hnh_panel_count(w->ctx, desc.rows, desc.nnz, desc.cols, (int)A.cols(), desc.max_row_nnz) : 1);
    return 0;
}

void StandardKernel::begin(hnh::World* w) {
    if (!profile) return;
    if (evw_ != nullptr && evw_ != w) hnh::fatal("Error, a profiled StandardKernel belongs to one world!");
    evw_ = w;
    if (used_ == pairs_.size()) {
        if (used_ >= 4096) resolve_profile();  // (bounded: a very long profiled section reads its pairs now and reuses them)
        else pairs_.emplace_back(w->event_create(), w->event_create());
    }
    // A start event right behind a cross-stream wait is stamped when the stream's EARLIER work completes, not when the wait is satisfied
    // (measured: the two-half accumulator ring's 16 launches summed to 4.5 ms inside a 2.9 ms call, profiles/r06_job4_fusion1_rank_share.log):
    // the span would count the wait for a transfer as kernel time.  A dispatch cannot start before the wait is satisfied, so a tiny one
    // (8 bytes filled) goes in front of the start event: its end is the earliest moment the kernel could have started.
    if (!tick_) tick_ = w->dmalloc(8);
    w->memset0(tick_, 8, HNH_STREAM_COMPUTE);
    w->event_record(pairs_[used_].first, HNH_STREAM_COMPUTE);
}

void StandardKernel::end(hnh::World* w, long launches) {
    if (!profile) return;
    w->event_record(pairs_[used_].second, HNH_STREAM_COMPUTE);
    used_++;
    kernel_launches += launches;  // a row pass may run as several column-panel launches (hnh_panel_count)
}

void StandardKernel::resolve_profile() {
    if (used_ == 0) return;
    hnh::World* w = evw_;
    w->check(w->be->hnh_event_sync(w->ctx, pairs_[used_ - 1].second), "hnh_event_sync");  // (same stream: the last pair completes last)
    for (size_t k = 0; k < used_; k++) {
        float ms = 0.f;
        w->check(w->be->hnh_event_elapsed_ms(w->ctx, pairs_[k].first, pairs_[k].second, &ms), "hnh_event_elapsed_ms");
        kernel_ms += ms;
    }
    used_ = 0;
}

StandardKernel::~StandardKernel() {
    if (evw_) {
        for (auto& pr : pairs_) {
            evw_->event_destroy(pr.first);
            evw_->event_destroy(pr.second);
        }
        if (tick_) evw_->dfree(tick_);
    }
}
