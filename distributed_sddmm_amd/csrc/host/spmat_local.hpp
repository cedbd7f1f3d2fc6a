// Local sparse storage + redistribution — same public surface as the reference's SpmatLocal.hpp (L3):
//   NonzeroDistribution (SpmatLocal.hpp:34-53), CSRHandle (:55-62), CSRLocal (:64-264), SpmatLocal (:267-606).
//
// MI355X layout.  A block's three arrays live in HBM:
//     values   fp64  [max_nnz]      rowStart int32 [rows + 1]      col_idx int32 [max_nnz]
// (the reference keeps MKL_INT = int64 indices and an MKL handle per buffer; int32 halves the index
// stream, and per-rank nnz < 2^31 is already required by the reference's `int` counters, :68).  A COO
// row_idx array is NOT stored: the HIP kernels walk rowStart, and hnh_expand_rowptr can rebuild it.
// Blocks that never move (dense-shift schedules) hold ONE buffer; blocks that travel around a ring
// (sparse-shift / Cannon) hold two, padded to the largest block on their ring, exactly like the
// reference's `buffer[2]` + `max_nnz` (:153-185).
//
// COO -> CSR (the reference: mkl_sparse_d_create_coo + mkl_sparse_convert_csr, :117-137) is a host-side
// counting sort by row followed by a per-row sort by column; it also rewrites the caller's tuples into
// CSR order in place (and with r/c swapped when transposing) like :139-147 does.
#pragma once
#include <parallel/algorithm>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "common.hpp"
#include "dense.hpp"
#include "world.hpp"

typedef enum { csr, coo, both } ShiftMode;  // kept for API parity; every mode ships values + col_idx + rowStart

class NonzeroDistribution {
public:
    hnh::World* world = nullptr;  // the reference's `MPI_Comm world`
    int rows_in_block = 1, cols_in_block = 1;
    virtual ~NonzeroDistribution() {}
    virtual int blockOwner(int row_block, int col_block) = 0;
    // processor that is supposed to own nonzero (r, c) (SpmatLocal.hpp:45-52)
    int getOwner(int r, int c, int transpose) {
        return transpose ? blockOwner(c / rows_in_block, r / cols_in_block) : blockOwner(r / rows_in_block, c / cols_in_block);
    }
};

class CSRHandle {
public:
    double* values = nullptr;     // device
    int32_t* col_idx = nullptr;   // device
    int32_t* rowStart = nullptr;  // device, rows + 1 entries
    int32_t* row_idx = nullptr;   // device, built on demand by CSRLocal::ensure_row_idx()
    bool row_idx_valid = false;   // row_idx describes the rowStart currently in this handle
};

class CSRLocal {
public:
    int64_t rows, cols;
    int max_nnz, num_coords;
    bool transpose;
    int active;
    CSRHandle* buffer;  // [2]; buffer[1] has null arrays when the block never shifts
    bool shifting;
    hnh::World* world;
    // Longest row of this block, and an upper bound over every block that can occupy these buffers after a
    // shift (set by the schedule from a ring all-gather).  Passed to the kernels as the long-row hint.
    int max_row_nnz = 0, ring_max_row_nnz = 0;
    int row_hint() const { return shifting ? ring_max_row_nnz : max_row_nnz; }

    // Ring-resident sparsity structure.  The pattern never changes, and a GPU has room for the col_idx / rowStart
    // arrays of EVERY block of its ring (config 3 sized: 8 x 6.8 MB), so a travelling block can leave its indices
    // at home: after replicate_ring_indices() each shift moves `values` only (8 B per nonzero instead of 12 B plus
    // a row pointer array) and just re-points the active handle at the resident indices of the arriving block.
    // Row windows (hnh_csr_window of hnh_kernels.h): the block's columns are cut at `bounds` into n_windows column ranges;
    // win_split[(q - 1) * rows + r] = first nonzero of row r in window q (device).  `window` selects the one the local
    // kernels work on (-1 = the whole block); the 1.5D dense-shift schedule walks them as the fetched chunks arrive.
    int32_t* win_split = nullptr;
    int n_windows = 1;
    int window = -1;
    int window_end = -1;  // one past the last selected window when several consecutive ones are taken in one pass (-1: just `window`)
    // Set by a schedule around an SDDMM call: the values of the selected window are known to be zero (nobody has written them
    // since the operation began), so a kernel that says overwrites_fresh_values() may store its results instead of adding to them
    // — and the schedule has skipped the zero fill.  Kernels that do not know the hint are never given unfilled values.
    bool values_fresh = false;
    // The SpMM twin, set by a schedule around an SpMM whose output rows (of the selected row range) nobody has written yet: a kernel
    // that says stores_fresh_output() stores the sums instead of adding them to what is there — and the schedule has skipped the zero
    // fill of that buffer (the staging rows of 1.5D replication reuse's mesh reduce-scatter are each written exactly once).
    bool out_fresh = false;
    // Borrowed value arrays.  A schedule whose kernel says borrows_value_arrays() may, for ONE operation on a block that never
    // shifts, point the kernels at slices of the CALLER's vectors instead of the block's own value array:
    //   spmm_values               the SpMM reads its nonzero values here — the copy of setCSRValues (SpmatLocal.hpp:571-579) is skipped;
    //   sddmm_dst / sddmm_scale   the SDDMM stores scale[e] * dot[e] into dst[e] (result and SValues slices) — the block's own values
    //                             and the closing Hadamard pass (15D_dense_shift.hpp:366) are not touched.
    // Set through SpmatLocal::lendCSRValues / lendSddmmTargets, cleared by reclaimValueArrays() before the operation returns.
    const double* spmm_values = nullptr;
    double* sddmm_dst = nullptr;
    const double* sddmm_scale = nullptr;
    void set_windows(const std::vector<int32_t>& bounds) {
        if (win_split) world->dfree(win_split);
        win_split = nullptr;
        n_windows = (int)bounds.size() + 1;
        window = -1;
        if (bounds.empty()) return;
        win_split = static_cast<int32_t*>(world->dmalloc(bounds.size() * (size_t)std::max<int64_t>(rows, 1) * sizeof(int32_t)));
        world->check(world->be->hnh_csr_window_bounds(world->ctx, rows, buffer[0].rowStart, buffer[0].col_idx, (int)bounds.size(), bounds.data(),
                                                      win_split, HNH_STREAM_COMPUTE),
                     "hnh_csr_window_bounds");
    }
    // the kernel-ABI description of the selected window; false when the whole block is selected
    bool window_args(hnh_csr_window* w) const {
        if (window < 0 || n_windows <= 1) return false;
        const int last = (window_end > window ? window_end : window + 1) - 1;  // the last window of the range (consecutive windows are one column range)
        w->beg = (window == 0) ? nullptr : win_split + (size_t)(window - 1) * (size_t)rows;
        w->end = (last == n_windows - 1) ? nullptr : win_split + (size_t)last * (size_t)rows;
        w->last = (last == n_windows - 1) ? 1 : 0;
        return true;
    }

    struct RingIndex {
        int32_t* col_idx = nullptr;
        int32_t* rowStart = nullptr;
        int32_t* row_idx = nullptr;  // COO view of the same structure, built on first use (ensure_row_idx)
        int nnz = 0;
        hnh_csr_plan* plan = nullptr;  // structure-only work of the row passes for this ring block (block_args)
        int part_nnz0 = -1;            // row parts of this ring block (select_row_part)
        hnh_csr_plan* part_plan[2] = {nullptr, nullptr};
    };
    std::vector<RingIndex> ring_index;  // by ring position of the block's origin; empty = indices travel (reference behaviour)
    int active_slot = -1;               // ring position whose block occupies the active buffer (ring-resident indices only)

    // Device build (the default): `dev_tuples` are this block's tuples in DEVICE memory, in any order.  One
    // radix sort into CSR order and one unzip pass; like the reference's constructor, the caller's tuples end up in CSR
    // order (and transposed when `transpose`), SpmatLocal.hpp:139-147.
    struct FromDevice {};
    CSRLocal(FromDevice, int64_t blockRows, int64_t blockCols, int64_t max_nnz_in, spcoord_t* dev_tuples, int num_coords_in,
             bool transpose_in, bool shifting_in = true)
        : rows(blockRows), cols(blockCols), max_nnz((int)max_nnz_in), num_coords(num_coords_in), transpose(transpose_in),
          active(0), shifting(shifting_in), world(hnh::current_world()) {
        if (num_coords > max_nnz) hnh::fatal("Error, block holds more nonzeros than its padded capacity!");
        hnh::PhaseTimer pt_all("CSRLocal ctor (device)");
        hnh::Backend* be = world->be;
        hnh_tuple* t = reinterpret_cast<hnh_tuple*>(dev_tuples);
        if (transpose) {
            std::swap(rows, cols);
            world->check(be->hnh_tuples_transform(world->ctx, t, num_coords, 1, 0, 0, HNH_STREAM_COMPUTE), "hnh_tuples_transform");
        }
        hnh_tuple_key key{};
        key.kind = HNH_KEY_ROW_COL;
        int row_bits = 1;
        while (row_bits < 32 && ((int64_t)1 << row_bits) < rows) row_bits++;
        world->check(be->hnh_tuples_sort(world->ctx, t, num_coords, &key, 32 + row_bits, HNH_STREAM_COMPUTE), "hnh_tuples_sort");
        allocate_buffers();
        world->check(be->hnh_tuples_to_csr(world->ctx, t, num_coords, rows, cols, buffer[0].rowStart, buffer[0].col_idx, buffer[0].values,
                                           &max_row_nnz, HNH_STREAM_COMPUTE),
                     "hnh_tuples_to_csr (nonzero outside its block?)");
        ring_max_row_nnz = max_row_nnz;
        if (shifting) {
            world->copy(buffer[1].values, buffer[0].values, (size_t)num_coords * sizeof(double), HNH_COPY_D2D, HNH_STREAM_COMPUTE);
            world->copy(buffer[1].col_idx, buffer[0].col_idx, (size_t)num_coords * sizeof(int32_t), HNH_COPY_D2D, HNH_STREAM_COMPUTE);
            world->copy(buffer[1].rowStart, buffer[0].rowStart, ((size_t)rows + 1) * sizeof(int32_t), HNH_COPY_D2D, HNH_STREAM_COMPUTE);
        }
    }

    void allocate_buffers() {
        buffer = new CSRHandle[2];
        const size_t cap = (size_t)std::max(max_nnz, 1);
        for (int t = 0; t < (shifting ? 2 : 1); t++) {
            buffer[t].values = static_cast<double*>(world->dmalloc(cap * sizeof(double)));
            buffer[t].col_idx = static_cast<int32_t*>(world->dmalloc(cap * sizeof(int32_t)));
            buffer[t].rowStart = static_cast<int32_t*>(world->dmalloc(((size_t)rows + 1) * sizeof(int32_t)));
        }
    }

    // Host build (HNH_HOST_SETUP=1; the reference's constructor restated): `coords` are HOST tuples.
    CSRLocal(int64_t blockRows, int64_t blockCols, int64_t max_nnz_in, spcoord_t* coords, int num_coords_in, bool transpose_in,
             bool shifting_in = true)
        : rows(blockRows), cols(blockCols), max_nnz((int)max_nnz_in), num_coords(num_coords_in), transpose(transpose_in),
          active(0), shifting(shifting_in), world(hnh::current_world()) {
        if (num_coords > max_nnz) hnh::fatal("Error, block holds more nonzeros than its padded capacity!");
        hnh::PhaseTimer pt_all("CSRLocal ctor");
        if (transpose) {
            std::swap(rows, cols);
#pragma omp parallel for
            for (int e = 0; e < num_coords; e++) std::swap(coords[e].r, coords[e].c);
        }
        hnh::PhaseTimer* pt = new hnh::PhaseTimer("  csr: count+scatter");
        // counting sort by row, then order each row by column
        std::vector<int32_t> rowStart((size_t)rows + 1, 0);
        for (int e = 0; e < num_coords; e++) {
            if ((int64_t)coords[e].r >= rows || (int64_t)coords[e].c >= cols) hnh::fatal("Error, nonzero outside its block!");
            rowStart[coords[e].r + 1]++;
        }
        for (int64_t r = 0; r < rows; r++) {
            max_row_nnz = std::max(max_row_nnz, (int)rowStart[r + 1]);
            rowStart[r + 1] += rowStart[r];
        }
        ring_max_row_nnz = max_row_nnz;
        std::vector<spcoord_t> sorted((size_t)num_coords);
        {
            std::vector<int32_t> cursor(rowStart.begin(), rowStart.end() - 1);
            for (int e = 0; e < num_coords; e++) sorted[cursor[coords[e].r]++] = coords[e];
        }
        delete pt;
        pt = new hnh::PhaseTimer("  csr: row sorts");
#pragma omp parallel for schedule(dynamic, 1024)
        for (int64_t r = 0; r < rows; r++)
            std::sort(sorted.begin() + rowStart[r], sorted.begin() + rowStart[r + 1],
                      [](const spcoord_t& a, const spcoord_t& b) { return a.c < b.c; });
        delete pt;
        pt = new hnh::PhaseTimer("  csr: unzip + upload");
        std::vector<int32_t> col((size_t)std::max(num_coords, 1));
        std::vector<double> val((size_t)std::max(num_coords, 1));
#pragma omp parallel for
        for (int e = 0; e < num_coords; e++) {
            coords[e] = sorted[e];  // caller's tuples end up in CSR order (SpmatLocal.hpp:139-147)
            col[e] = (int32_t)sorted[e].c;
            val[e] = sorted[e].value;
        }

        allocate_buffers();
        for (int t = 0; t < (shifting ? 2 : 1); t++) {
            world->copy(buffer[t].values, val.data(), (size_t)num_coords * sizeof(double), HNH_COPY_H2D, HNH_STREAM_COMPUTE);
            world->copy(buffer[t].col_idx, col.data(), (size_t)num_coords * sizeof(int32_t), HNH_COPY_H2D, HNH_STREAM_COMPUTE);
            world->copy(buffer[t].rowStart, rowStart.data(), ((size_t)rows + 1) * sizeof(int32_t), HNH_COPY_H2D, HNH_STREAM_COMPUTE);
        }
        world->sync(HNH_STREAM_COMPUTE);  // host staging vectors die here
        delete pt;
    }

    ~CSRLocal() {
        for (int t = 0; t < 2; t++) {
            world->dfree(buffer[t].values);
            if (ring_index.empty()) {  // otherwise the handles point into ring_index
                world->dfree(buffer[t].col_idx);
                world->dfree(buffer[t].rowStart);
            }
            world->dfree(buffer[t].row_idx);
        }
        if (static_plan) world->be->hnh_csr_plan_destroy(world->ctx, static_plan);
        for (RowRange& rr : row_ranges)
            if (rr.plan) world->be->hnh_csr_plan_destroy(world->ctx, rr.plan);
        for (hnh_csr_plan* pp : part_plan)
            if (pp) world->be->hnh_csr_plan_destroy(world->ctx, pp);
        for (RingIndex& ri : ring_index) {
            if (ri.plan) world->be->hnh_csr_plan_destroy(world->ctx, ri.plan);
            for (hnh_csr_plan* pp : ri.part_plan)
                if (pp) world->be->hnh_csr_plan_destroy(world->ctx, pp);
            world->dfree(ri.col_idx);
            world->dfree(ri.rowStart);
            world->dfree(ri.row_idx);
        }
        world->dfree(win_split);
        delete[] buffer;
    }

    // One-time exchange at construction: every rank of `ring` receives the index arrays of every other rank's
    // block (all transfers in one group).  nnz_per_slot[s] = nonzeros of the block owned by ring position s.
    void replicate_ring_indices(const hnh::Comm& ring, const std::vector<int>& nnz_per_slot) {
        if (!shifting) hnh::fatal("Error, only travelling blocks keep ring-resident indices!");
        const int n = ring.size(), me = ring.me;
        const size_t rp_bytes = ((size_t)rows + 1) * sizeof(int32_t);
        ring_index.resize(n);
        for (int s2 = 0; s2 < n; s2++) {
            ring_index[s2].nnz = nnz_per_slot[s2];
            ring_index[s2].col_idx = static_cast<int32_t*>(world->dmalloc((size_t)std::max(nnz_per_slot[s2], 1) * sizeof(int32_t)));
            ring_index[s2].rowStart = static_cast<int32_t*>(world->dmalloc(rp_bytes));
        }
        world->copy(ring_index[me].col_idx, buffer[0].col_idx, (size_t)num_coords * sizeof(int32_t), HNH_COPY_D2D, HNH_STREAM_COMPUTE);
        world->copy(ring_index[me].rowStart, buffer[0].rowStart, rp_bytes, HNH_COPY_D2D, HNH_STREAM_COMPUTE);
        world->group_begin();
        for (int k = 1; k < n; k++) {
            const int dst = (me + k) % n, src = (me - k + n) % n;
            world->sendrecv(ring, buffer[0].col_idx, (size_t)num_coords * sizeof(int32_t), dst, ring_index[src].col_idx,
                            (size_t)nnz_per_slot[src] * sizeof(int32_t), src, HNH_STREAM_COMPUTE);
            world->sendrecv(ring, buffer[0].rowStart, rp_bytes, dst, ring_index[src].rowStart, rp_bytes, src, HNH_STREAM_COMPUTE);
        }
        world->group_end();
        world->sync(HNH_STREAM_COMPUTE);
        for (int t = 0; t < 2; t++) {  // the per-buffer index arrays are no longer needed
            world->dfree(buffer[t].col_idx);
            world->dfree(buffer[t].rowStart);
            buffer[t].col_idx = ring_index[me].col_idx;
            buffer[t].rowStart = ring_index[me].rowStart;
        }
        active_slot = me;
    }
    CSRLocal(const CSRLocal&) = delete;
    CSRLocal& operator=(const CSRLocal&) = delete;

    CSRHandle* getActive() { return buffer + active; }

    // The kernel-ABI description of the block in the active buffer, with the plan that caches what the row passes derive from
    // its STRUCTURE alone (cache-panel boundaries, hub-row work list): one plan per structure that keeps its contents — the
    // block itself when it never shifts, each ring position when the ring's structure is resident; a block that ships its
    // indices (HNH_SHIP_INDICES=1: the arrays are overwritten by every shift) gets none.  The structure is fixed once the
    // constructor has run (the reference's SpmatLocal.hpp:78-188), so a plan never needs invalidating.
    hnh_csr_plan* static_plan = nullptr;

    // Row parts.  When a block's OUTPUT is a buffer that travels (the SpMM accumulator of 1.5D replication reuse), the schedule runs
    // the block's rows in two halves so that one half's shift overlaps the other half's kernel: `row_part` selects rows
    // [0, part_rows0) or [part_rows0, rows) for the next SpMM (-1 = all rows).  A part is a CSR view of its own (row pointers from
    // rowStart + first row, the same col_idx / values arrays), with its own structure plan.  Blocks that never shift only.
    int row_part = -1;
    int64_t part_rows0 = 0;
    int part_nnz0 = -1;  // nonzeros of part 0 (read back once, when the parts are first used)
    hnh_csr_plan* part_plan[2] = {nullptr, nullptr};
    int64_t part_first_row() const { return range_sel >= 0 ? row_ranges[(size_t)range_sel].r0 : (row_part == 1 ? part_rows0 : 0); }
    // Row RANGES (round 6): the same view for any rows [r0, r1) of a block that never shifts.  The row-merged layout of 1.5D replication
    // reuse keeps all fetched blocks as ONE transposed CSR block whose rows are the rows of the chunk-major landing buffer: chunk q of
    // every fetched block is then a contiguous row range, and a pass over the chunks that have landed is a pass over one range — no
    // window boundaries, nothing re-streamed.  Each distinct range keeps its nonzero count (read back once) and its own plan.
    struct RowRange {
        int64_t r0, r1;
        int nnz_before, nnz;
        hnh_csr_plan* plan;
    };
    std::vector<RowRange> row_ranges;
    int range_sel = -1;
    void select_row_range(int64_t r0, int64_t r1) {
        if (r0 < 0) {
            range_sel = -1;
            return;
        }
        if (shifting) hnh::fatal("Error, row ranges need a block whose index arrays keep their contents!");
        if (r1 < r0 || r1 > rows) hnh::fatal("Error, row range outside the block!");
        for (size_t k = 0; k < row_ranges.size(); k++)
            if (row_ranges[k].r0 == r0 && row_ranges[k].r1 == r1) {
                range_sel = (int)k;
                return;
            }
        int32_t v[2] = {0, 0};
        world->copy(&v[0], getActive()->rowStart + r0, sizeof(int32_t), HNH_COPY_D2H, HNH_STREAM_COMPUTE);
        world->copy(&v[1], getActive()->rowStart + r1, sizeof(int32_t), HNH_COPY_D2H, HNH_STREAM_COMPUTE);
        world->sync(HNH_STREAM_COMPUTE);
        row_ranges.push_back(RowRange{r0, r1, v[0], v[1] - v[0], nullptr});
        range_sel = (int)row_ranges.size() - 1;
    }
    // whether this block can be run in row parts: its structure must keep its contents (never shifts, or ring-resident indices)
    bool supports_row_parts() const { return !shifting || !ring_index.empty(); }
    void select_row_part(int part) {
        if (part >= 0) {
            if (!supports_row_parts()) hnh::fatal("Error, row parts need a block whose index arrays keep their contents!");
            part_rows0 = rows / 2;
            int* known = ring_index.empty() ? &part_nnz0 : &ring_index[(size_t)active_slot].part_nnz0;
            if (*known < 0) {
                int32_t v = 0;
                world->copy(&v, getActive()->rowStart + part_rows0, sizeof(int32_t), HNH_COPY_D2H, HNH_STREAM_COMPUTE);
                world->sync(HNH_STREAM_COMPUTE);
                *known = v;
            }
        }
        row_part = part;
    }

    hnh_csr_block block_args() {
        CSRHandle* h = getActive();
        hnh_csr_plan** slot = !ring_index.empty() ? &ring_index[(size_t)active_slot].plan : (shifting ? nullptr : &static_plan);
        hnh_csr_block b;
        b.rows = rows;
        b.nnz = num_coords;
        b.cols = cols;
        b.max_row_nnz = row_hint();
        b.reserved = 0;
        b.rowptr = h->rowStart;
        b.col_idx = h->col_idx;
        if (range_sel >= 0) {
            RowRange& rr = row_ranges[(size_t)range_sel];
            slot = &rr.plan;
            b.rows = rr.r1 - rr.r0;
            b.nnz = rr.nnz;
            b.rowptr = h->rowStart + rr.r0;  // (row pointers are offsets into the block's col_idx / values: still valid)
        } else if (row_part >= 0) {
            const bool ring = !ring_index.empty();
            slot = ring ? &ring_index[(size_t)active_slot].part_plan[row_part] : &part_plan[row_part];
            const int nnz0 = ring ? ring_index[(size_t)active_slot].part_nnz0 : part_nnz0;
            b.rows = row_part == 0 ? part_rows0 : rows - part_rows0;
            b.nnz = row_part == 0 ? nnz0 : num_coords - nnz0;
            b.rowptr = h->rowStart + part_first_row();  // (row pointers are offsets into the block's col_idx / values: still valid)
        }
        if (slot != nullptr && *slot == nullptr) world->check(world->be->hnh_csr_plan_create(world->ctx, slot), "hnh_csr_plan_create");
        b.plan = slot ? *slot : nullptr;
        return b;
    }

    // COO row indices of the block in the active buffer (the nonzero-balanced SDDMM of narrow operands walks them): built
    // once per structure — per ring position when the ring's structure is resident, again after a shift when indices travel.
    int32_t* ensure_row_idx(int stream = HNH_STREAM_COMPUTE) {
        if (!ring_index.empty()) {
            RingIndex& ri = ring_index[(size_t)active_slot];
            if (!ri.row_idx) {
                ri.row_idx = static_cast<int32_t*>(world->dmalloc((size_t)std::max(ri.nnz, 1) * sizeof(int32_t)));
                world->check(world->be->hnh_expand_rowptr(world->ctx, rows, ri.rowStart, ri.row_idx, stream), "hnh_expand_rowptr");
            }
            return ri.row_idx;
        }
        CSRHandle* h = getActive();
        if (!h->row_idx) h->row_idx = static_cast<int32_t*>(world->dmalloc((size_t)std::max(max_nnz, 1) * sizeof(int32_t)));
        if (!h->row_idx_valid) {
            world->check(world->be->hnh_expand_rowptr(world->ctx, rows, h->rowStart, h->row_idx, stream), "hnh_expand_rowptr");
            h->row_idx_valid = true;
        }
        return h->row_idx;
    }

    // Cyclic shift of the block around a ring (SpmatLocal.hpp:200-259): the active buffer goes to comm
    // index `dst`, the block of comm index `src` lands in the passive buffer, then the two swap roles.
    // Stream-ordered on `stream`; `tag` is unused (explicit peers, no wildcard matching).
    void shiftCSR(int src, int dst, const hnh::Comm& comm, int nnz_to_receive, int tag, ShiftMode mode,
                  int stream = HNH_STREAM_COMM, int incoming_slot = -1) {
        (void)tag;
        (void)mode;
        if (!shifting) hnh::fatal("Error, this sparse block was built without a second buffer and cannot shift!");
        if (nnz_to_receive > max_nnz) hnh::fatal("Error, incoming sparse block exceeds the padded capacity!");
        CSRHandle* send = buffer + active;
        CSRHandle* recv = buffer + 1 - active;
        if (!ring_index.empty()) {  // indices are resident: only the values travel
            if (incoming_slot < 0 || incoming_slot >= (int)ring_index.size()) hnh::fatal("Error, shiftCSR needs the arriving block's ring position!");
            if (ring_index[incoming_slot].nnz != nnz_to_receive) hnh::fatal("Error, arriving block does not match its resident indices!");
            world->sendrecv(comm, send->values, (size_t)num_coords * sizeof(double), dst, recv->values,
                            (size_t)nnz_to_receive * sizeof(double), src, stream);
            recv->col_idx = ring_index[incoming_slot].col_idx;
            recv->rowStart = ring_index[incoming_slot].rowStart;
            num_coords = nnz_to_receive;
            active = 1 - active;
            active_slot = incoming_slot;
            return;
        }
        world->group_begin();  // the three arrays travel as one RCCL group
        world->sendrecv(comm, send->values, (size_t)num_coords * sizeof(double), dst, recv->values,
                        (size_t)nnz_to_receive * sizeof(double), src, stream);
        world->sendrecv(comm, send->col_idx, (size_t)num_coords * sizeof(int32_t), dst, recv->col_idx,
                        (size_t)nnz_to_receive * sizeof(int32_t), src, stream);
        world->sendrecv(comm, send->rowStart, ((size_t)rows + 1) * sizeof(int32_t), dst, recv->rowStart,
                        ((size_t)rows + 1) * sizeof(int32_t), src, stream);
        world->group_end();
        recv->row_idx_valid = false;  // another block's structure arrives in this handle
        num_coords = nnz_to_receive;
        active = 1 - active;
    }
};

class SpmatLocal {
public:
    std::vector<spcoord_t> coords;  // unzipped tuples (host): what loaders / generators fill, and the HNH_HOST_SETUP=1 pipeline

    // Setup on the device (default).  redistribute_nonzeros() uploads the input's tuples once and its
    // result is DEVICE-RESIDENT: owner routing, the all-to-all, the column-major sort, the block-column split, index
    // localisation and the CSR conversion all run on the GPU (hnh_tuples_* of hnh_kernels.h) and `coords` stays empty.
    hnh::DeviceArray dcoords;
    bool resident = false;
    size_t n_resident = 0;
    size_t num_tuples() const { return resident ? n_resident : coords.size(); }
    hnh_tuple* dptr() const { return static_cast<hnh_tuple*>(dcoords.ptr()); }
    static bool device_setup() { return std::getenv("HNH_HOST_SETUP") == nullptr; }
    static_assert(sizeof(spcoord_t) == sizeof(hnh_tuple), "spcoord_t must have the layout of hnh_tuple");

    // r %= rmod, c %= cmod (0 = leave): the "make indices block-local" loops of the schedule constructors
    // (15D_dense_shift.hpp:96-100, 25D_cannon_dense.hpp:117-125)
    void localize(uint64_t rmod, uint64_t cmod) {
        if (resident) {
            world->check(world->be->hnh_tuples_transform(world->ctx, dptr(), (int64_t)n_resident, 0, rmod, cmod, HNH_STREAM_COMPUTE),
                         "hnh_tuples_transform");
            return;
        }
#pragma omp parallel for
        for (size_t e = 0; e < coords.size(); e++) {
            if (rmod) coords[e].r %= rmod;
            if (cmod) coords[e].c %= cmod;
        }
    }

    // the schedules drop the tuples once the CSR blocks exist
    void release_tuples() {
        std::vector<spcoord_t>().swap(coords);
        if (resident) world->sync(HNH_STREAM_COMPUTE);
        dcoords.reset();
        resident = false;
        n_resident = 0;
    }

    // Replicates root's tuples over `comm` (2.5D sparse replication: bottom face -> fibers, 25D_cannon_sparse.hpp:67-77)
    void broadcast_tuples(const hnh::Comm& comm, int root) {
        hnh::World* w = world;
        int n = (int)num_tuples();
        w->host_bcast(comm, root, &n, sizeof(int));
        if (resident) {
            if (comm.me != root) {
                dcoords = hnh::DeviceArray(w, (size_t)n * sizeof(spcoord_t));
                n_resident = (size_t)n;
            }
            w->sync(HNH_STREAM_COMPUTE);
            w->device_bcast(comm, root, dcoords.ptr(), (size_t)n * sizeof(spcoord_t), HNH_STREAM_COMM);
        } else {
            if (comm.me != root) coords.resize((size_t)n);
            w->host_bcast(comm, root, coords.data(), coords.size() * sizeof(spcoord_t));
        }
    }

    uint64_t M = 0, N = 0, dist_nnz = 0;  // global properties
    bool initialized;

    int owned_coords_start = 0, owned_coords_end = 0;
    std::vector<int> layer_coords_start, layer_coords_sizes;
    bool coordinate_ownership_initialized;
    bool csr_initialized;

    std::vector<uint64_t> blockStarts;
    std::vector<CSRLocal*> csr_blocks;
    hnh::World* world;

    SpmatLocal() : initialized(false), coordinate_ownership_initialized(false), csr_initialized(false), world(hnh::current_world()) {}
    ~SpmatLocal() {
        for (CSRLocal* b : csr_blocks) delete b;
    }
    SpmatLocal(const SpmatLocal&) = delete;
    SpmatLocal& operator=(const SpmatLocal&) = delete;

    // One CSRLocal per non-empty block column (max_nnz == -1; stationary), or a single padded block
    // that will travel around a ring (SpmatLocal.hpp:314-338).
    // `widths` (optional, max_nnz == -1 only): the column count of every block when they differ
    void initializeCSRBlocks(int blockRows, int blockCols, int max_nnz, bool transpose, const std::vector<int64_t>* widths = nullptr) {
        auto width = [&](size_t i) { return widths ? (*widths)[i] : (int64_t)blockCols; };
        if (resident) {
            if (max_nnz == -1) {
                for (size_t i = 0; i + 1 < blockStarts.size(); i++) {
                    const int n = (int)(blockStarts[i + 1] - blockStarts[i]);
                    csr_blocks.push_back(n > 0 ? new CSRLocal(CSRLocal::FromDevice(), blockRows, width(i), n,
                                                              reinterpret_cast<spcoord_t*>(dptr() + blockStarts[i]), n, transpose, false)
                                               : nullptr);
                }
            } else {
                const int n = (int)(blockStarts[1] - blockStarts[0]);
                csr_blocks.push_back(new CSRLocal(CSRLocal::FromDevice(), blockRows, blockCols, max_nnz, reinterpret_cast<spcoord_t*>(dptr()), n,
                                                  transpose, true));
            }
            csr_initialized = true;
            return;
        }
        if (max_nnz == -1) {
            for (size_t i = 0; i + 1 < blockStarts.size(); i++) {
                const int n = (int)(blockStarts[i + 1] - blockStarts[i]);
                csr_blocks.push_back(n > 0 ? new CSRLocal(blockRows, width(i), n, coords.data() + blockStarts[i], n, transpose, false)
                                           : nullptr);
            }
        } else {
            const int n = (int)(blockStarts[1] - blockStarts[0]);
            csr_blocks.push_back(new CSRLocal(blockRows, blockCols, max_nnz, coords.data(), n, transpose, true));
        }
        csr_initialized = true;
    }

    void own_all_coordinates() {
        owned_coords_start = 0;
        owned_coords_end = (int)num_tuples();
        layer_coords_start = {0, (int)num_tuples()};
        layer_coords_sizes = {(int)num_tuples()};
        coordinate_ownership_initialized = true;
    }

    void shard_across_layers(int num_layers, int current_layer) {
        divideIntoSegments((int)num_tuples(), num_layers, layer_coords_start, layer_coords_sizes);
        owned_coords_start = layer_coords_start[current_layer];
        owned_coords_end = layer_coords_start[current_layer + 1];
        coordinate_ownership_initialized = true;
    }

    // Routes every tuple to its owner under `dist` (optionally transposing), all-to-all, then sorts the
    // received tuples column-major (SpmatLocal.hpp:389-462).
    SpmatLocal* redistribute_nonzeros(NonzeroDistribution* dist, bool transpose, bool in_place) {
        if (device_setup()) return redistribute_on_device(dist, transpose, in_place);
        if (resident) hnh::fatal("Error, the host setup path cannot take device-resident tuples!");
        hnh::World* w = dist->world ? dist->world : world;
        const int p = w->size;
        hnh::PhaseTimer pt_all("redistribute_nonzeros");
        hnh::PhaseTimer* pt = new hnh::PhaseTimer("  redist: owners + pack");
        std::vector<size_t> sendcounts(p, 0), recvcounts(p, 0);
        std::vector<int> owner(coords.size());
#pragma omp parallel for
        for (size_t e = 0; e < coords.size(); e++) owner[e] = dist->getOwner((int)coords[e].r, (int)coords[e].c, transpose);
        for (size_t e = 0; e < coords.size(); e++) {
            if (owner[e] < 0 || owner[e] >= p) hnh::fatal("Error, nonzero distribution produced an invalid owner!");
            sendcounts[owner[e]]++;
        }
        std::vector<size_t> offsets(p + 1, 0);
        for (int r = 0; r < p; r++) offsets[r + 1] = offsets[r] + sendcounts[r];
        std::vector<spcoord_t> sendbuf(coords.size());
        {
            std::vector<size_t> cursor(offsets.begin(), offsets.end() - 1);
            for (size_t e = 0; e < coords.size(); e++) {
                spcoord_t& t = sendbuf[cursor[owner[e]]++];
                t.r = transpose ? coords[e].c : coords[e].r;
                t.c = transpose ? coords[e].r : coords[e].c;
                t.value = coords[e].value;
            }
        }
        delete pt;
        pt = new hnh::PhaseTimer("  redist: alltoallv");
        std::vector<size_t> all_counts((size_t)p * p);
        w->host_allgather(sendcounts.data(), all_counts.data(), (size_t)p * sizeof(size_t));
        for (int r = 0; r < p; r++) recvcounts[r] = all_counts[(size_t)r * p + w->rank];
        std::vector<size_t> sb(p), sd(p), rb(p), rd(p);
        size_t total = 0;
        for (int r = 0; r < p; r++) {
            sb[r] = sendcounts[r] * sizeof(spcoord_t);
            sd[r] = offsets[r] * sizeof(spcoord_t);
            rb[r] = recvcounts[r] * sizeof(spcoord_t);
            rd[r] = total * sizeof(spcoord_t);
            total += recvcounts[r];
        }
        const uint64_t newM = transpose ? N : M, newN = transpose ? M : N;
        std::vector<spcoord_t> received(total);
        w->host_alltoallv(sendbuf.data(), sb, sd, received.data(), rb, rd);

        SpmatLocal* result = in_place ? this : new SpmatLocal();
        result->M = newM;
        result->N = newN;
        result->dist_nnz = dist_nnz;
        result->initialized = true;
        result->coords.swap(received);
        delete pt;
        pt = new hnh::PhaseTimer("  redist: column-major sort");
        __gnu_parallel::sort(result->coords.begin(), result->coords.end(), column_major);
        delete pt;
        return result;
    }

    // The same routing on the GPU.  The owner of a tuple is a table lookup (the distribution's blockOwner() evaluated once
    // per block on the host); tuples are radix-sorted by owner, which is the Alltoallv pack; boundaries between owners
    // come from binary searches; the exchange moves device buffers peer to peer; the received tuples are radix-sorted
    // column-major.  The input (host tuples of a loader, or an already resident matrix) is left untouched.
    SpmatLocal* redistribute_on_device(NonzeroDistribution* dist, bool transpose, bool in_place) {
        hnh::World* w = dist->world ? dist->world : world;
        hnh::Backend* be = w->be;
        const int p = w->size;
        hnh::PhaseTimer pt_all("redistribute_nonzeros (device)");
        hnh::PhaseTimer* pt = new hnh::PhaseTimer("  redist: upload + owner sort");
        const size_t n = num_tuples();
        const uint64_t newM = transpose ? N : M, newN = transpose ? M : N;

        hnh::DeviceArray work(w, std::max<size_t>(n, 1) * sizeof(spcoord_t));
        if (n) w->copy(work.ptr(), resident ? dcoords.ptr() : (const void*)coords.data(), n * sizeof(spcoord_t),
                       resident ? HNH_COPY_D2D : HNH_COPY_H2D, HNH_STREAM_COMPUTE);
        hnh_tuple* wt = static_cast<hnh_tuple*>(work.ptr());

        const int64_t rib = dist->rows_in_block, cib = dist->cols_in_block;
        if (rib <= 0 || cib <= 0) hnh::fatal("Error, nonzero distribution has empty blocks!");
        const int64_t nrb = std::max<int64_t>(1, ((int64_t)newM + rib - 1) / rib), ncb = std::max<int64_t>(1, ((int64_t)newN + cib - 1) / cib);
        std::vector<int32_t> table((size_t)(nrb * ncb));
        for (int64_t rb = 0; rb < nrb; rb++)
            for (int64_t cb = 0; cb < ncb; cb++) {
                const int o = dist->blockOwner((int)rb, (int)cb);
                if (o < 0 || o >= p) hnh::fatal("Error, nonzero distribution produced an invalid owner!");
                table[(size_t)(rb * ncb + cb)] = o;
            }
        hnh::DeviceArray dtable(w, table.size() * sizeof(int32_t));
        w->copy(dtable.ptr(), table.data(), table.size() * sizeof(int32_t), HNH_COPY_H2D, HNH_STREAM_COMPUTE);

        hnh_tuple_key okey{};
        okey.kind = HNH_KEY_OWNER;
        okey.transpose = transpose ? 1 : 0;
        okey.rows_in_block = rib; okey.cols_in_block = cib; okey.n_col_blocks = ncb;
        okey.owner_table = static_cast<const int32_t*>(dtable.ptr());
        int owner_bits = 1;
        while ((1 << owner_bits) < p) owner_bits++;
        w->check(be->hnh_tuples_sort(w->ctx, wt, (int64_t)n, &okey, owner_bits, HNH_STREAM_COMPUTE), "hnh_tuples_sort (owner)");
        std::vector<int64_t> starts((size_t)p + 1, 0);
        w->check(be->hnh_tuples_bucket_starts(w->ctx, wt, (int64_t)n, &okey, p, starts.data(), HNH_STREAM_COMPUTE), "hnh_tuples_bucket_starts");
        if ((size_t)starts[p] != n) hnh::fatal("Error, nonzero distribution produced an invalid owner!");
        if (transpose) w->check(be->hnh_tuples_transform(w->ctx, wt, (int64_t)n, 1, 0, 0, HNH_STREAM_COMPUTE), "hnh_tuples_transform");
        w->sync(HNH_STREAM_COMPUTE);
        delete pt;

        pt = new hnh::PhaseTimer("  redist: alltoallv (device)");
        std::vector<size_t> sendcounts(p), recvcounts(p), all_counts((size_t)p * p);
        for (int r = 0; r < p; r++) sendcounts[r] = (size_t)(starts[r + 1] - starts[r]);
        w->host_allgather(sendcounts.data(), all_counts.data(), (size_t)p * sizeof(size_t));
        std::vector<size_t> sb(p), sd(p), rb(p), rd(p);
        size_t total = 0;
        for (int r = 0; r < p; r++) {
            recvcounts[r] = all_counts[(size_t)r * p + w->rank];
            sb[r] = sendcounts[r] * sizeof(spcoord_t);
            sd[r] = (size_t)starts[r] * sizeof(spcoord_t);
            rb[r] = recvcounts[r] * sizeof(spcoord_t);
            rd[r] = total * sizeof(spcoord_t);
            total += recvcounts[r];
        }
        hnh::DeviceArray received(w, std::max<size_t>(total, 1) * sizeof(spcoord_t));
        w->device_alltoallv(work.ptr(), sb, sd, received.ptr(), rb, rd, HNH_STREAM_COMM);
        work.reset();
        delete pt;

        pt = new hnh::PhaseTimer("  redist: column-major sort (device)");
        SpmatLocal* result = in_place ? this : new SpmatLocal();
        result->M = newM;
        result->N = newN;
        result->dist_nnz = dist_nnz;
        result->initialized = true;
        std::vector<spcoord_t>().swap(result->coords);
        result->dcoords = std::move(received);
        result->resident = true;
        result->n_resident = total;
        hnh_tuple_key ckey{};
        ckey.kind = HNH_KEY_COL_ROW;
        int col_bits = 1;
        while (col_bits < 32 && ((uint64_t)1 << col_bits) < newN) col_bits++;
        w->check(be->hnh_tuples_sort(w->ctx, result->dptr(), (int64_t)total, &ckey, 32 + col_bits, HNH_STREAM_COMPUTE), "hnh_tuples_sort (column-major)");
        delete pt;
        return result;
    }

    // Synthetic / file input.  The reference uses CombBLAS (GenGraph500Data with initiator .25 x4 = ER,
    // ParallelReadMM; SpmatLocal.hpp:467-533).  Ours: a counter-based generator that every rank
    // evaluates identically (see er_generator.hpp), each rank keeping a strided slice — any initial
    // distribution is legal because redistribute_nonzeros() follows.
    void loadTuples(bool readFromFile, int logM, int nnz_per_row, std::string filename);
    // Seeded random relabelling of rows / columns for load balance (random_permute.cpp, SpmatLocal.hpp:506-507);
    // also applied by loadTuples when HNH_PERMUTE_SEED is set.
    void permuteVertices(uint64_t seed);

    // Tuples must be column-major sorted.  Splits them into block columns of `blockWidth` and (optionally)
    // makes column indices block-local (SpmatLocal.hpp:541-563).
    void divideIntoBlockCols(int blockWidth, int targetDivisions, bool modIndex) {
        const size_t nblocks = (size_t)targetDivisions;
        if (resident) {  // boundaries by binary search over the column-major tuples, then the index localisation in one pass
            hnh_tuple_key key{};
            key.kind = HNH_KEY_COL_DIV;
            key.div = blockWidth;
            std::vector<int64_t> starts(nblocks + 1, 0);
            world->check(world->be->hnh_tuples_bucket_starts(world->ctx, dptr(), (int64_t)n_resident, &key, (int64_t)nblocks, starts.data(),
                                                             HNH_STREAM_COMPUTE), "hnh_tuples_bucket_starts");
            if ((size_t)starts[nblocks] != n_resident) hnh::fatal("Error, more block columns than expected!");
            blockStarts.assign(starts.begin(), starts.end());
            if (modIndex) localize(0, (uint64_t)blockWidth);
            return;
        }
        blockStarts.assign(nblocks + 1, coords.size());
        size_t next = 0;  // next block whose start is still unknown
        for (uint64_t i = 0; i < coords.size(); i++) {
            const uint64_t id = coords[i].c / (uint64_t)blockWidth;
            if (id >= nblocks) hnh::fatal("Error, more block columns than expected!");
            while (next <= id) blockStarts[next++] = i;
            if (modIndex) coords[i].c %= (uint64_t)blockWidth;
        }
    }

    // Piecewise relabelling of the column indices (hnh_tuples_remap_cols): segment seg = (c / div) * n_sub + (c % div) / sub_div
    // moves to dest[seg], offsets inside a segment are kept; a negative dest marks a segment that must be empty.
    void remapColumns(int64_t div, int64_t sub_div, int64_t n_sub, const std::vector<int64_t>& dest) {
        if (resident) {
            world->check(world->be->hnh_tuples_remap_cols(world->ctx, dptr(), (int64_t)n_resident, div, sub_div, n_sub, dest.data(),
                                                          (int64_t)dest.size(), HNH_STREAM_COMPUTE),
                         "hnh_tuples_remap_cols");
            return;
        }
        bool bad = false;
#pragma omp parallel for reduction(|| : bad)
        for (size_t e = 0; e < coords.size(); e++) {
            const uint64_t in = coords[e].c % (uint64_t)div, seg = (coords[e].c / (uint64_t)div) * (uint64_t)n_sub + in / (uint64_t)sub_div;
            if (seg >= dest.size() || dest[seg] < 0) { bad = true; continue; }
            coords[e].c = (uint64_t)dest[seg] + in % (uint64_t)sub_div;
        }
        if (bad) hnh::fatal("Error, a nonzero lies in a column segment that has no destination!");
    }

    // restores the column-major order divideIntoBlockCols() expects (after remapColumns)
    void sortColumnMajor(uint64_t ncols) {
        if (resident) {
            hnh_tuple_key ckey{};
            ckey.kind = HNH_KEY_COL_ROW;
            int col_bits = 1;
            while (col_bits < 32 && ((uint64_t)1 << col_bits) < ncols) col_bits++;
            world->check(world->be->hnh_tuples_sort(world->ctx, dptr(), (int64_t)n_resident, &ckey, 32 + col_bits, HNH_STREAM_COMPUTE),
                         "hnh_tuples_sort (column-major)");
            return;
        }
        __gnu_parallel::sort(coords.begin(), coords.end(), column_major);
    }

    // Two unequal block columns: block 0 = columns [0, width0), block 1 = [width0, width0 * parts) with its column indices
    // made block-local.  (The 1.5D dense-shift schedule under the mesh fetch: the rank's own block column, and all fetched
    // block columns as one block whose columns index the landing buffer.)
    void divideIntoLocalAndRemote(int64_t width0, int parts) {
        divideIntoBlockCols((int)width0, parts, false);  // parts uniform block columns of width0 ...
        const uint64_t cut = blockStarts[1], total = blockStarts[(size_t)parts];
        blockStarts = {0, cut, total};                   // ... of which all but the first are merged
        const size_t n1 = (size_t)(total - cut);
        if (n1 == 0) return;
        if (resident) {
            std::vector<int64_t> dest((size_t)parts);
            for (int b = 0; b < parts; b++) dest[(size_t)b] = (b == 0) ? -1 : (int64_t)(b - 1) * width0;
            world->check(world->be->hnh_tuples_remap_cols(world->ctx, dptr() + cut, (int64_t)n1, width0, width0, 1, dest.data(), parts,
                                                          HNH_STREAM_COMPUTE),
                         "hnh_tuples_remap_cols");
        } else {
#pragma omp parallel for
            for (size_t e = (size_t)cut; e < (size_t)total; e++) coords[e].c -= (uint64_t)width0;
        }
    }

    void monolithBlockColumn() {
        blockStarts.clear();
        blockStarts.push_back(0);
        blockStarts.push_back(num_tuples());
    }

    // values <-> per-block storage (SpmatLocal.hpp:571-605); device-to-device on the compute stream
    void setCSRValues(const hnh::VectorXd& values) {
        // the reference copies without looking (SpmatLocal.hpp:571-579); a vector made by the wrong like_S*_values would
        // read past its end, so the length is checked here
        if (!blockStarts.empty() && (uint64_t)values.size() < blockStarts.back()) hnh::fatal("Error, sparse value vector has the wrong length!");
        for (size_t i = 0; i + 1 < blockStarts.size(); i++)
            if (csr_blocks[i] != nullptr)
                world->copy(csr_blocks[i]->getActive()->values, values.data() + blockStarts[i],
                            sizeof(double) * (blockStarts[i + 1] - blockStarts[i]), HNH_COPY_D2D, HNH_STREAM_COMPUTE);
    }

    hnh::VectorXd getCSRValues() {
        hnh::VectorXd values = hnh::VectorXd::Constant((int64_t)blockStarts.back(), 0.0);
        for (size_t i = 0; i + 1 < blockStarts.size(); i++)
            if (csr_blocks[i] != nullptr)
                world->copy(values.data() + blockStarts[i], csr_blocks[i]->getActive()->values,
                            sizeof(double) * (blockStarts[i + 1] - blockStarts[i]), HNH_COPY_D2D, HNH_STREAM_COMPUTE);
        return values;
    }

    void setValuesConstant(double cval) {
        for (size_t i = 0; i + 1 < blockStarts.size(); i++)
            if (csr_blocks[i] != nullptr)
                world->check(world->be->hnh_fill_f64(world->ctx, csr_blocks[i]->getActive()->values,
                                                     (int64_t)(blockStarts[i + 1] - blockStarts[i]), cval, HNH_STREAM_COMPUTE),
                             "hnh_fill_f64");
    }

    // ---- borrowed value arrays (CSRLocal::spmm_values / sddmm_dst): the slice of a caller's vector that belongs to block i stands in
    // for the block's value array.  The narrow row kernels (R = 8 / 16 / 32) walk 128-byte lines of the value arrays, so at those
    // widths a slice is lent only when it starts on a line; blocks that cannot borrow fall back to the copy / the Hadamard pass.
    // block-level counts since construction: SpMM value arrays lent / copied, SDDMM results written in place / by a Hadamard pass
    int64_t borrow_stats[4] = {0, 0, 0, 0};
    // mode: Distributed_Sparse::borrow_mode (-1 never, 1 always, 0 by this rule)
    static bool lendable(const void* p, int64_t R, int mode) { return mode != 0 ? mode > 0 : (R > 32 || reinterpret_cast<uintptr_t>(p) % 128 == 0); }

    // setCSRValues for an SpMM whose kernel borrows_value_arrays(): stationary blocks read `values` in place
    void lendCSRValues(const hnh::VectorXd& values, int64_t R, int mode = 0) {
        if (!blockStarts.empty() && (uint64_t)values.size() < blockStarts.back()) hnh::fatal("Error, sparse value vector has the wrong length!");
        for (size_t i = 0; i + 1 < blockStarts.size(); i++) {
            CSRLocal* blk = csr_blocks[i];
            if (blk == nullptr) continue;
            const double* src = values.data() + blockStarts[i];
            if (!blk->shifting && lendable(src, R, mode)) {
                blk->spmm_values = src;
                borrow_stats[0]++;
                continue;
            }
            borrow_stats[1]++;
            world->copy(blk->getActive()->values, src, sizeof(double) * (blockStarts[i + 1] - blockStarts[i]), HNH_COPY_D2D, HNH_STREAM_COMPUTE);
        }
    }

    // for an SDDMM that visits every nonzero once (first visits store): stationary blocks write svalues .* dots straight into `out`;
    // hadamardWithCSRValues() then skips them
    void lendSddmmTargets(const hnh::VectorXd& svalues, hnh::VectorXd& out, int64_t R, int mode = 0) {
        if (!blockStarts.empty() && ((uint64_t)svalues.size() < blockStarts.back() || (uint64_t)out.size() < blockStarts.back()))
            hnh::fatal("Error, sparse value vector has the wrong length!");
        if (svalues.data() == out.data()) return;  // in-place call: the kernel's scale operand must not alias its destination
        for (size_t i = 0; i + 1 < blockStarts.size(); i++) {
            CSRLocal* blk = csr_blocks[i];
            if (blk == nullptr || blk->shifting) continue;
            double* dst = out.data() + blockStarts[i];
            const double* scale = svalues.data() + blockStarts[i];
            if (lendable(dst, R, mode) && lendable(scale, R, mode)) {
                blk->sddmm_dst = dst;
                blk->sddmm_scale = scale;
                borrow_stats[2]++;
            }
        }
    }

    void reclaimValueArrays() {
        for (CSRLocal* blk : csr_blocks)
            if (blk != nullptr) {
                blk->spmm_values = nullptr;
                blk->sddmm_dst = nullptr;
                blk->sddmm_scale = nullptr;
            }
    }

    // out[e] = svalues[e] * (block values)[e] — the Hadamard step that ends every SDDMM
    // (`SValues.cwiseProduct(choice->getCSRValues())`, 15D_dense_shift.hpp:366) without the temporary.  Blocks whose SDDMM already
    // wrote the product (lendSddmmTargets) are skipped.
    void hadamardWithCSRValues(const hnh::VectorXd& svalues, hnh::VectorXd& out, int64_t out_offset = 0) {
        if (!blockStarts.empty() && ((uint64_t)svalues.size() < out_offset + blockStarts.back() || (uint64_t)out.size() < out_offset + blockStarts.back()))
            hnh::fatal("Error, sparse value vector has the wrong length!");
        for (size_t i = 0; i + 1 < blockStarts.size(); i++)
            if (csr_blocks[i] != nullptr && csr_blocks[i]->sddmm_dst == nullptr && blockStarts[i + 1] > blockStarts[i]) {
                borrow_stats[3]++;
                world->check(world->be->hnh_hadamard_f64(world->ctx, out.data() + out_offset + blockStarts[i],
                                                         svalues.data() + out_offset + blockStarts[i],
                                                         csr_blocks[i]->getActive()->values,
                                                         (int64_t)(blockStarts[i + 1] - blockStarts[i]), HNH_STREAM_COMPUTE),
                             "hnh_hadamard_f64");
            }
    }
};
