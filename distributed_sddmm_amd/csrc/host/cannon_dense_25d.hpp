// 2.5D Cannon schedule replicating a DENSE operand — same class and behaviour as the reference's
// Sparse25D_Cannon_Dense (25D_cannon_dense.hpp): grid s x s x c with s = sqrt(p/c); one dense operand is
// all-gathered along the fiber, the other moves along grid columns, the sparse block moves along grid rows
// (Cannon); the sparse blocks are skewed once in the constructor (:138-145) and the moving dense operand
// is skewed / un-skewed by initial_shift / de_shift (:169-211).  R is split over the s ranks of a grid row.
//
// Differences from the reference that matter for correctness: every shift names its source explicitly
// (the reference receives from MPI_ANY_SOURCE with a reused tag and can hang) and
// the constructor skew completes all of its transfers before the block is used (the reference forgets to
// wait for the COO row indices in `both` mode, Appendix C #1 — our blocks ship rowStart instead).
#pragma once
#include <cmath>

#include "distributed_sparse.hpp"

class Block_Cyclic25D : public NonzeroDistribution {
public:
    int sqrtpc, c;
    std::shared_ptr<FlexibleGrid> grid;
    Block_Cyclic25D(int M, int N, int sqrtpc, int c, std::shared_ptr<FlexibleGrid>& grid) {
        world = grid->world;
        this->sqrtpc = sqrtpc;
        this->c = c;
        this->grid = grid;
        rows_in_block = divideAndRoundUp(M, sqrtpc * c) * c;
        cols_in_block = divideAndRoundUp(N, sqrtpc * c);
    }
    int blockOwner(int row_block, int col_block) override { return grid->get_global_rank(row_block, col_block / c, col_block % c); }
};

class Sparse25D_Cannon_Dense : public Distributed_Sparse {
public:
    int sqrtpc;
    std::vector<int> nnz_in_row_axis, nnz_in_row_axis_tpose;
    int sparse_shift;
    DenseMatrix accumulation_buffer;
    DenseMatrix ring_spare;
    DenseMatrix dense_spare[2];  // landing buffers of the read-only (SDDMM) dense ring
    bool acc_halves = true;      // SpMM: the accumulator's shift in two row halves, each under the other half's kernel (HNH_ACC_HALVES=0: off)

    Sparse25D_Cannon_Dense(SpmatLocal* S_input, int R, int c, KernelImplementation* k) : Distributed_Sparse(k) {
        this->c = c;
        if (const char* h = std::getenv("HNH_ACC_HALVES")) acc_halves = std::atoi(h) != 0;
        if (c < 1 || p % c != 0) hnh::fatal("Error, for 2.5D algorithm, p / c must be a perfect square!");
        sqrtpc = (int)std::lround(std::sqrt((double)(p / c)));
        if (sqrtpc * sqrtpc * c != p) hnh::fatal("Error, for 2.5D algorithm, p / c must be a perfect square!");

        algorithm_name = "2.5D Cannon's Algorithm Replicating Dense Matrices";
        proc_grid_names = {"# Rows", "# Cols", "# Layers"};
        perf_counter_keys = {"Dense Cyclic Shift Time", "Sparse Cyclic Shift Time", "Dense Fiber Communication Time",
                             "Computation Time", "Setup Shift Time"};

        grid.reset(new FlexibleGrid(sqrtpc, sqrtpc, c, 3));
        A_R_split_world = grid->row_world;
        B_R_split_world = grid->row_world;
        r_split = true;

        this->M = S_input->M;
        this->N = S_input->N;
        localArows = divideAndRoundUp((int)this->M, sqrtpc * c);
        localBrows = divideAndRoundUp((int)this->N, sqrtpc * c);
        setRValue(R);

        Block_Cyclic25D nonzero_dist((int)M, (int)N, sqrtpc, c, grid);
        Block_Cyclic25D transpose_dist((int)N, (int)M, sqrtpc, c, grid);
        S.reset(S_input->redistribute_nonzeros(&nonzero_dist, false, false));
        ST.reset(S_input->redistribute_nonzeros(&transpose_dist, true, false));

        nnz_in_row_axis.resize(sqrtpc);
        nnz_in_row_axis_tpose.resize(sqrtpc);
        int my_nnz = (int)S->num_tuples(), my_nnz_tpose = (int)ST->num_tuples();
        world->host_allgather_comm(grid->row_world, &my_nnz, nnz_in_row_axis.data(), sizeof(int));
        world->host_allgather_comm(grid->row_world, &my_nnz_tpose, nnz_in_row_axis_tpose.data(), sizeof(int));
        const int max_nnz = *std::max_element(nnz_in_row_axis.begin(), nnz_in_row_axis.end());
        const int max_nnz_tpose = *std::max_element(nnz_in_row_axis_tpose.begin(), nnz_in_row_axis_tpose.end());

        const uint64_t ar = (uint64_t)localArows * c, br = (uint64_t)localBrows * c;
        S->localize(ar, (uint64_t)localBrows);
        ST->localize(br, (uint64_t)localArows);
        S->own_all_coordinates();
        ST->own_all_coordinates();
        S->monolithBlockColumn();
        ST->monolithBlockColumn();
        S->initializeCSRBlocks(localArows * c, localBrows, max_nnz, true);
        S->release_tuples();
        ST->initializeCSRBlocks(localBrows * c, localArows, max_nnz_tpose, true);
        ST->release_tuples();

        publish_ring_max_row(S.get(), grid->row_world);
        publish_ring_max_row(ST.get(), grid->row_world);
        if (std::getenv("HNH_SHIP_INDICES") == nullptr) {  // default: the ring's sparsity structure stays resident, values travel
            S->csr_blocks[0]->replicate_ring_indices(grid->row_world, nnz_in_row_axis);
            ST->csr_blocks[0]->replicate_ring_indices(grid->row_world, nnz_in_row_axis_tpose);
        }
        // Skew the sparse blocks along grid rows in preparation for repeated Cannon passes (:138-145)
        const int src = pMod(grid->rankInRow + grid->rankInCol, sqrtpc);
        const int dst = pMod(grid->rankInRow - grid->rankInCol, sqrtpc);
        sparse_shift = src;
        S->csr_blocks[0]->shiftCSR(src, dst, grid->row_world, nnz_in_row_axis[src], 0, both, HNH_STREAM_COMPUTE, src);
        S->blockStarts[1] = S->csr_blocks[0]->num_coords;
        ST->csr_blocks[0]->shiftCSR(src, dst, grid->row_world, nnz_in_row_axis_tpose[src], 0, both, HNH_STREAM_COMPUTE, src);
        ST->blockStarts[1] = ST->csr_blocks[0]->num_coords;
        world->sync(HNH_STREAM_COMPUTE);
        check_initialized();
    }

    void setRValue(int R) override {
        this->R = R;
        localAcols = R / sqrtpc;
        localBcols = R / sqrtpc;
        if (localAcols * sqrtpc != R) hnh::fatal("Error, R must be divisible by sqrt(p) / c!");
        aSubmatrices.clear();
        bSubmatrices.clear();
        aSubmatrices.emplace_back(localArows * (grid->k + c * grid->i), localAcols * grid->j, localArows, localAcols);
        bSubmatrices.emplace_back(localBrows * (grid->k + c * grid->i), localBcols * grid->j, localBrows, localBcols);
    }

    // Cannon skew of the moving dense operand along its grid column (:169-190) ...
    void initial_shift(DenseMatrix* localA, DenseMatrix* localB, KernelMode mode) override {
        auto t = phase_begin("Setup Shift Time");
        DenseMatrix* m = (mode == k_sddmmA || mode == k_spmmA) ? localA : localB;
        if (m != nullptr) skew(m, -grid->rankInRow);
        phase_end(t);
    }
    // ... and its inverse (:192-211)
    void de_shift(DenseMatrix* localA, DenseMatrix* localB, KernelMode mode) override {
        auto t = phase_begin("Setup Shift Time");
        DenseMatrix* m = (mode == k_sddmmA || mode == k_spmmA) ? localA : localB;
        if (m != nullptr) skew(m, +grid->rankInRow);
        phase_end(t);
    }

    VectorXd like_S_values(double value) override { return VectorXd::Constant((int64_t)ST->blockStarts[1], value); }
    VectorXd like_ST_values(double value) override { return VectorXd::Constant((int64_t)S->blockStarts[1], value); }

    // the SpMM's accumulator starts at home holding nothing: step 0's kernel on every rank may store its rows (two-half ring only)
    bool spmm_stores_output() const override { return sqrtpc > 1 && acc_halves && kernel->handles_row_parts() && kernel->stores_fresh_output(); }

    void algorithm(DenseMatrix& localA, DenseMatrix& localB, VectorXd& SValues, VectorXd* sddmm_result_ptr, KernelMode mode,
                   bool initial_replicate) override {
        SpmatLocal* choice;
        DenseMatrix *Arole, *Brole;
        const std::vector<int>* nnz_in_axis;
        if (mode == k_spmmA || mode == k_sddmmA) {  // both kernels run on transposed blocks with swapped roles (:235-241)
            choice = ST.get(); Arole = &localB; Brole = &localA; nnz_in_axis = &nnz_in_row_axis_tpose;
        } else {
            choice = S.get(); Arole = &localA; Brole = &localB; nnz_in_axis = &nnz_in_row_axis;
        }
        if ((uint64_t)SValues.size() != choice->blockStarts[1]) hnh::fatal("Error, sparse value vector has the wrong length!");
        const bool is_sddmm = (mode == k_sddmmA || mode == k_sddmmB);
        // SpMM through spmmA / spmmB / fusedSpMM: the accumulator (Brole) has not been zeroed when this schedule said it stores (see
        // spmm_stores_output()); a path below that cannot store zeroes it first
        bool acc_unset = !is_sddmm && take_output_unset(*Brole);

        // (SDDMM: the travelling block's first visit — step 0, at home — may store instead of add, see CSRLocal::values_fresh)
        const bool fresh = is_sddmm && kernel->overwrites_fresh_values();
        {
            auto t = phase_begin("Computation Time");
            if (is_sddmm && !fresh) choice->setValuesConstant(0.0);
            else if (!is_sddmm) choice->setCSRValues(SValues);
            phase_end(t);
        }
        if (initial_replicate && c > 1) {
            auto t = phase_begin("Dense Fiber Communication Time");
            if (accumulation_buffer.rows() != Arole->rows() * c || accumulation_buffer.cols() != Arole->cols())
                accumulation_buffer = DenseMatrix(Arole->rows() * c, Arole->cols());
            world->allgather(grid->fiber_world, Arole->data(), accumulation_buffer.data(), (size_t)Arole->size() * sizeof(double),
                             HNH_STREAM_COMPUTE);
            phase_end(t);
        }
        DenseMatrix& stationary = (c > 1) ? accumulation_buffer : *Arole;
        CSRLocal* blk = choice->csr_blocks[0];
        const int s = sqrtpc;
        const int ddst = pMod(grid->rankInCol + 1, s), dsrc = pMod(grid->rankInCol - 1, s);
        const int ssrc = pMod(grid->rankInRow - 1, s), sdst = pMod(grid->rankInRow + 1, s);
        const KernelMode kmode = (mode == k_spmmA) ? k_spmmB : mode;
        if (s > 1) order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);

        // Each step moves two things: the dense operand down its grid column and the sparse block along its grid row.
        // Whatever the kernel only READS can travel while the kernel runs; whatever it WRITES has to wait for it:
        //   SDDMM writes the sparse values  -> dense shift overlaps the kernel (triple buffered, caller's matrix
        //                                      untouched, s-1 shifts), sparse shift follows it;
        //   SpMM  writes the dense operand  -> sparse shift overlaps the kernel (double buffered), dense shift follows it.
        if (is_sddmm) {
            if (s > 1)
                for (auto& sp : dense_spare)
                    if (sp.rows() != Brole->rows() || sp.cols() != Brole->cols()) sp = DenseMatrix(Brole->rows(), Brole->cols());
            const size_t bytes = (size_t)Brole->size() * sizeof(double);
            DenseMatrix* cur = Brole;
            for (int i = 0; i < s; i++) {
                auto t = phase_begin("Computation Time");
                if (i > 0) world->event_wait(event(1 + (i - 1) % 2), HNH_STREAM_COMPUTE);  // both shifts of step i-1 landed
                if (choice->csr_blocks[0] != nullptr) choice->csr_blocks[0]->values_fresh = fresh && i == 0;
                kernel->triple_function(kmode, *choice, stationary, *cur, 0, localAcols * grid->j);
                if (choice->csr_blocks[0] != nullptr) choice->csr_blocks[0]->values_fresh = false;
                phase_end(t);
                if (s > 1) {
                    t = phase_begin("Dense Cyclic Shift Time");
                    world->event_record(event(3 + i % 2), HNH_STREAM_COMPUTE);
                    if (i < s - 1) {
                        DenseMatrix* target = &dense_spare[i % 2];
                        if (i >= 2) world->event_wait(event(3 + (i - 1) % 2), HNH_STREAM_COMM);  // kernel i-1 last read `target`
                        world->sendrecv(grid->col_world, cur->data(), bytes, ddst, target->data(), bytes, dsrc, HNH_STREAM_COMM);
                        cur = target;
                    }
                    phase_end(t);
                    t = phase_begin("Sparse Cyclic Shift Time");
                    world->event_wait(event(3 + i % 2), HNH_STREAM_COMM);  // kernel i wrote the travelling values
                    blk->shiftCSR(ssrc, sdst, grid->row_world, (*nnz_in_axis)[pMod(sparse_shift - i - 1, s)], 72, coo, HNH_STREAM_COMM,
                                  pMod(sparse_shift - i - 1, s));
                    choice->blockStarts[1] = blk->num_coords;
                    world->event_record(event(1 + i % 2), HNH_STREAM_COMM);
                    phase_end(t);
                }
            }
            if (s > 1) world->event_wait(event(1 + (s - 1) % 2), HNH_STREAM_COMPUTE);  // the sparse block is home again
        } else if (s > 1 && acc_halves && kernel->handles_row_parts() && blk != nullptr && blk->supports_row_parts()) {
            const bool store0 = acc_unset && kernel->stores_fresh_output();
            if (acc_unset && !store0) Brole->setZero();
            acc_unset = false;
            // SpMM: the moving dense operand is the ACCUMULATOR, so its shift has to follow the kernel that wrote it
            // (25D_cannon_dense.hpp:274-302: kernel -> two shifts -> barrier).  In two row halves (CSRLocal::row_part) one half travels
            // under the other half's kernel:
            //     compute:  K(i,H0)  K(i,H1)            K(i+1,H0) ...        comm:  sparse(i)  D(i,H0)  D(i,H1)  sparse(i+1) ...
            // The sparse block is only read: its shift starts as soon as kernel i-1 has released the passive buffer, as before.
            hnh::BufferPair bBuf(Brole, &ring_spare);
            const int64_t h = Brole->rows() / 2, cols = Brole->cols();
            const size_t bytes0 = (size_t)h * (size_t)cols * sizeof(double), bytes1 = (size_t)(Brole->rows() - h) * (size_t)cols * sizeof(double);
            order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);  // earlier users of the spare are done
            for (int i = 0; i < s; i++) {
                DenseMatrix* act = bBuf.getActive();
                DenseMatrix* pas = bBuf.getPassive();
                auto t = phase_begin("Computation Time");
                if (i > 0) world->event_wait(event(1 + (i - 1) % 2), HNH_STREAM_COMPUTE);  // the sparse block of step i has landed
                for (int part = 0; part < 2; part++) {
                    if (i > 0) world->event_wait(event(14 + part), HNH_STREAM_COMPUTE);    // this half of the accumulator has landed
                    blk->select_row_part(part);
                    blk->out_fresh = store0 && i == 0;  // the accumulator's first kernel: its rows are stored, nobody zeroed them
                    kernel->triple_function(kmode, *choice, stationary, *act, 0, localAcols * grid->j);
                    blk->out_fresh = false;
                    blk->select_row_part(-1);
                    world->event_record(event(10 + 2 * (i % 2) + part), HNH_STREAM_COMPUTE);
                }
                phase_end(t);
                t = phase_begin("Sparse Cyclic Shift Time");
                if (i >= 1) world->event_wait(event(10 + 2 * ((i - 1) % 2) + 1), HNH_STREAM_COMM);  // kernel i-1 released the passive sparse buffer
                blk->shiftCSR(ssrc, sdst, grid->row_world, (*nnz_in_axis)[pMod(sparse_shift - i - 1, s)], 72, csr, HNH_STREAM_COMM,
                              pMod(sparse_shift - i - 1, s));
                choice->blockStarts[1] = blk->num_coords;
                world->event_record(event(1 + i % 2), HNH_STREAM_COMM);
                phase_end(t);
                t = phase_begin("Dense Cyclic Shift Time");
                for (int part = 0; part < 2; part++) {
                    world->event_wait(event(10 + 2 * (i % 2) + part), HNH_STREAM_COMM);  // kernel (i, part) wrote this half
                    const size_t off = part == 0 ? 0 : (size_t)h * (size_t)cols, bytes = part == 0 ? bytes0 : bytes1;
                    world->sendrecv(grid->col_world, act->data() + off, bytes, ddst, pas->data() + off, bytes, dsrc, HNH_STREAM_COMM);
                    world->event_record(event(14 + part), HNH_STREAM_COMM);
                }
                bBuf.swapActive();
                phase_end(t);
            }
            world->event_wait(event(1 + (s - 1) % 2), HNH_STREAM_COMPUTE);
            world->event_wait(event(14), HNH_STREAM_COMPUTE);
            world->event_wait(event(15), HNH_STREAM_COMPUTE);
            bBuf.sync_active();
        } else {
            if (acc_unset) Brole->setZero();
            acc_unset = false;
            hnh::BufferPair bBuf(Brole, &ring_spare);
            for (int i = 0; i < s; i++) {
                auto t = phase_begin("Computation Time");
                if (i > 0) world->event_wait(event(1 + (i - 1) % 2), HNH_STREAM_COMPUTE);
                kernel->triple_function(kmode, *choice, stationary, *bBuf.getActive(), 0, localAcols * grid->j);
                phase_end(t);
                if (s > 1) {
                    t = phase_begin("Sparse Cyclic Shift Time");
                    world->event_record(event(3 + i % 2), HNH_STREAM_COMPUTE);
                    if (i >= 1) world->event_wait(event(3 + (i - 1) % 2), HNH_STREAM_COMM);  // kernel i-1 released the passive sparse buffer
                    blk->shiftCSR(ssrc, sdst, grid->row_world, (*nnz_in_axis)[pMod(sparse_shift - i - 1, s)], 72, csr, HNH_STREAM_COMM,
                                  pMod(sparse_shift - i - 1, s));
                    choice->blockStarts[1] = blk->num_coords;
                    phase_end(t);
                    t = phase_begin("Dense Cyclic Shift Time");
                    world->event_wait(event(3 + i % 2), HNH_STREAM_COMM);  // kernel i wrote the moving dense operand
                    shiftDenseMatrix(bBuf, grid->col_world, ddst, dsrc, HNH_STREAM_COMM);
                    world->event_record(event(1 + i % 2), HNH_STREAM_COMM);
                    phase_end(t);
                }
            }
            if (s > 1) world->event_wait(event(1 + (s - 1) % 2), HNH_STREAM_COMPUTE);
            bBuf.sync_active();
        }

        auto t = phase_begin("Computation Time");
        if (is_sddmm) choice->hadamardWithCSRValues(SValues, *sddmm_result_ptr);
        phase_end(t);
    }

private:
    // move `m` by `offset` positions along the grid column (stream-ordered on the compute stream)
    void skew(DenseMatrix* m, int offset) {
        if (sqrtpc == 1) return;
        hnh::BufferPair buf(m, &ring_spare);
        shiftDenseMatrix(buf, grid->col_world, pMod(grid->rankInCol + offset, sqrtpc), pMod(grid->rankInCol - offset, sqrtpc),
                         HNH_STREAM_COMPUTE);
        buf.sync_active();
    }
};
