// 2.5D Cannon schedule replicating the SPARSE matrix — same class and behaviour as the reference's
// Sparse25D_Cannon_Sparse (25D_cannon_sparse.hpp): grid s x s x c, s = sqrt(p/c).  S is distributed over
// the s x s floor and broadcast to all c layers (:47-54); each layer owns 1/c of the values
// (shard_across_layers, :109-110).  Both dense operands move (A along grid rows, B along grid columns) and
// R is split over s*c (:139-145).  SpMM first all-gathers the value shards over the fiber (:224-233);
// SDDMM ends with a reduce-scatter of the partial dot products over the fiber (:294-300).
//
// Ring steps: SDDMM only reads both moving operands, so both ring transfers of step i are issued on the
// communication stream as soon as step i's kernel has been enqueued and overlap with it (double
// buffered); in SpMM the A-role operand is the accumulator, so the transfers follow the kernel.
#pragma once
#include <cmath>

#include "distributed_sparse.hpp"

class Floor2D : public NonzeroDistribution {
public:
    std::shared_ptr<FlexibleGrid> grid;
    Floor2D(int M, int N, int sqrtpc, int c, std::shared_ptr<FlexibleGrid>& grid) {
        (void)c;
        this->grid = grid;
        world = grid->world;
        rows_in_block = divideAndRoundUp(M, sqrtpc);
        cols_in_block = divideAndRoundUp(N, sqrtpc);
    }
    int blockOwner(int row_block, int col_block) override { return grid->get_global_rank(row_block, col_block, 0); }
};

class Sparse25D_Cannon_Sparse : public Distributed_Sparse {
public:
    int sqrtpc;
    int nnz, nnz_tpose;
    DenseMatrix spareA[2], spareB[2];
    VectorXd value_buffer;  // the reference's per-call `accumulation_buffer` (length = all nonzeros of the block)

    void broadcastCoordinatesFromFloor(std::unique_ptr<SpmatLocal>& spmat) {
        spmat->broadcast_tuples(grid->fiber_world, 0);
    }

    Sparse25D_Cannon_Sparse(SpmatLocal* S_input, int R, int c, KernelImplementation* k) : Distributed_Sparse(k) {
        this->c = c;
        if (c < 1 || p % c != 0) hnh::fatal("Error, for 2.5D algorithm, p / c must be a perfect square!");
        sqrtpc = (int)std::lround(std::sqrt((double)(p / c)));
        if (sqrtpc * sqrtpc * c != p) hnh::fatal("Error, for 2.5D algorithm, p / c must be a perfect square!");

        algorithm_name = "2.5D Cannon's Algorithm Replicating Sparse Matrix";
        proc_grid_names = {"# Rows", "# Cols", "# Layers"};
        perf_counter_keys = {"Dense Cyclic Shift Time", "Sparse Fiber Communication Time", "Computation Time", "Setup Shift Time"};

        grid.reset(new FlexibleGrid(sqrtpc, sqrtpc, c, 3));
        A_R_split_world = grid->colfiber_slice;
        B_R_split_world = grid->colfiber_slice;
        r_split = true;

        this->M = S_input->M;
        this->N = S_input->N;
        localArows = divideAndRoundUp((int)this->M, sqrtpc);
        localBrows = divideAndRoundUp((int)this->N, sqrtpc);
        setRValue(R);

        // nonzeros land on the bottom face of the cuboid, then are broadcast up the fibers
        Floor2D nonzero_dist((int)M, (int)N, sqrtpc, c, grid);
        Floor2D transpose_dist((int)N, (int)M, sqrtpc, c, grid);
        S.reset(S_input->redistribute_nonzeros(&nonzero_dist, false, false));
        ST.reset(S_input->redistribute_nonzeros(&transpose_dist, true, false));
        broadcastCoordinatesFromFloor(S);
        broadcastCoordinatesFromFloor(ST);

        // each layer is responsible for a contiguous 1/c of the values
        S->shard_across_layers(c, grid->k);
        ST->shard_across_layers(c, grid->k);

        S->localize((uint64_t)localArows, (uint64_t)localBrows);
        ST->localize((uint64_t)localBrows, (uint64_t)localArows);
        S->monolithBlockColumn();
        ST->monolithBlockColumn();
        S->initializeCSRBlocks(localArows, localBrows, -1, false);
        nnz = (int)S->num_tuples();
        S->release_tuples();
        ST->initializeCSRBlocks(localBrows, localArows, -1, false);
        nnz_tpose = (int)ST->num_tuples();
        ST->release_tuples();
        check_initialized();
    }

    void setRValue(int R) override {
        this->R = R;
        localAcols = R / (sqrtpc * c);
        localBcols = R / (sqrtpc * c);
        if (localAcols * sqrtpc * c != R) hnh::fatal("Error, R must be divisible by sqrt(pc)!");
        const int shift = pMod(grid->j + grid->i, sqrtpc);  // column slices are stored already skewed
        aSubmatrices.clear();
        bSubmatrices.clear();
        aSubmatrices.emplace_back(localArows * grid->i, localAcols * c * shift + grid->k * localAcols, localArows, localAcols);
        bSubmatrices.emplace_back(localBrows * grid->i, localBcols * c * shift + grid->k * localBcols, localBrows, localBcols);
    }

    // exchange with the grid-transposed rank (j, i, k); self-inverse (:157-186)
    void initial_shift(DenseMatrix* localA, DenseMatrix* localB, KernelMode mode) override {
        auto t = phase_begin("Setup Shift Time");
        DenseMatrix* m = (mode == k_sddmmA || mode == k_spmmA) ? localB : localA;
        if (m != nullptr && sqrtpc > 1) {
            const int partner = grid->get_global_rank(grid->j, grid->i, grid->k);
            hnh::Comm wc = world->world_comm();
            hnh::BufferPair buf(m, &spareA[0]);
            shiftDenseMatrix(buf, wc, partner, partner, HNH_STREAM_COMPUTE);
            buf.sync_active();
        }
        phase_end(t);
    }
    void de_shift(DenseMatrix* localA, DenseMatrix* localB, KernelMode mode) override { initial_shift(localA, localB, mode); }

    void algorithm(DenseMatrix& localA, DenseMatrix& localB, VectorXd& SValues, VectorXd* sddmm_result_ptr, KernelMode mode,
                   bool initial_replicate) override {
        (void)initial_replicate;
        SpmatLocal* choice;
        DenseMatrix *Arole, *Brole;
        int nnz_selection;
        if (mode == k_spmmA || mode == k_sddmmA) {
            choice = S.get(); Arole = &localA; Brole = &localB; nnz_selection = nnz;
        } else {
            choice = ST.get(); Arole = &localB; Brole = &localA; nnz_selection = nnz_tpose;
        }
        if (SValues.size() != choice->owned_coords_end - choice->owned_coords_start)
            hnh::fatal("Error, sparse value vector has the wrong length!");
        const bool is_sddmm = (mode == k_sddmmA || mode == k_sddmmB);
        if (value_buffer.size() != nnz_selection) value_buffer = VectorXd(nnz_selection);

        if (!is_sddmm) {
            if (c > 1) {
                auto t = phase_begin("Sparse Fiber Communication Time");
                world->allgatherv_f64(grid->fiber_world, SValues.data(), (size_t)SValues.size(), value_buffer.data(),
                                      choice->layer_coords_sizes, choice->layer_coords_start, HNH_STREAM_COMPUTE);
                if (kernel->borrows_value_arrays()) choice->lendCSRValues(value_buffer, Arole->cols(), borrow_mode);  // (the stationary block reads it in place)
                else choice->setCSRValues(value_buffer);
                phase_end(t);
            } else {
                auto t = phase_begin("Computation Time");
                if (kernel->borrows_value_arrays()) choice->lendCSRValues(SValues, Arole->cols(), borrow_mode);
                else choice->setCSRValues(SValues);
                phase_end(t);
            }
        } else if (!kernel->overwrites_fresh_values()) {  // (otherwise the stationary block's first visit stores instead of adding)
            auto t = phase_begin("Computation Time");
            choice->setValuesConstant(0.0);
            phase_end(t);
        }

        const KernelMode temp = (mode == k_sddmmB) ? k_sddmmA : (mode == k_spmmB ? k_spmmA : mode);
        const int s = sqrtpc;
        const int rdst = pMod(grid->rankInRow + 1, s), rsrc = pMod(grid->rankInRow - 1, s);
        const int cdst = pMod(grid->rankInCol + 1, s), csrc = pMod(grid->rankInCol - 1, s);

        if (is_sddmm) {
            // both operands read-only: triple buffering, s-1 overlapped double shifts, caller buffers untouched
            if (s > 1) {
                for (int t = 0; t < 2; t++) {
                    ensure(spareA[t], Arole->rows(), Arole->cols());
                    ensure(spareB[t], Brole->rows(), Brole->cols());
                }
                order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);
            }
            DenseMatrix *curA = Arole, *curB = Brole;
            const size_t abytes = (size_t)Arole->size() * sizeof(double), bbytes = (size_t)Brole->size() * sizeof(double);
            for (int i = 0; i < s; i++) {
                auto t = phase_begin("Computation Time");
                if (i > 0) world->event_wait(event(1 + (i - 1) % 2), HNH_STREAM_COMPUTE);
                if (choice->csr_blocks[0] != nullptr) choice->csr_blocks[0]->values_fresh = kernel->overwrites_fresh_values() && i == 0;
                kernel->triple_function(temp, *choice, *curA, *curB, 0, pMod(grid->i + grid->j + i, s) * localAcols);
                if (choice->csr_blocks[0] != nullptr) choice->csr_blocks[0]->values_fresh = false;
                phase_end(t);
                if (i < s - 1) {
                    t = phase_begin("Dense Cyclic Shift Time");
                    world->event_record(event(3 + i % 2), HNH_STREAM_COMPUTE);
                    if (i >= 2) world->event_wait(event(3 + (i - 1) % 2), HNH_STREAM_COMM);
                    DenseMatrix *ta = &spareA[i % 2], *tb = &spareB[i % 2];
                    world->group_begin();  // row ring and column ring: different peers, different links, one group
                    world->sendrecv(grid->row_world, curA->data(), abytes, rdst, ta->data(), abytes, rsrc, HNH_STREAM_COMM);
                    world->sendrecv(grid->col_world, curB->data(), bbytes, cdst, tb->data(), bbytes, csrc, HNH_STREAM_COMM);
                    world->group_end();
                    world->event_record(event(1 + i % 2), HNH_STREAM_COMM);
                    curA = ta;
                    curB = tb;
                    phase_end(t);
                }
            }
        } else {
            // The A-role operand accumulates the SpMM output: s shifts, each after its kernel.  The B-role operand is only
            // read: its shift for step i+1 is issued while kernel i runs (triple buffered, s-1 shifts, caller's matrix
            // untouched), so only A's transfer sits between two kernels.
            hnh::BufferPair aBuf(Arole, &spareA[0]);
            if (s > 1) {
                for (int t = 0; t < 2; t++) ensure(spareB[t], Brole->rows(), Brole->cols());
                order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);
            }
            DenseMatrix* curB = Brole;
            const size_t bbytes = (size_t)Brole->size() * sizeof(double);
            for (int i = 0; i < s; i++) {
                auto t = phase_begin("Computation Time");
                if (i > 0) world->event_wait(event(1 + (i - 1) % 2), HNH_STREAM_COMPUTE);  // both shifts of step i-1 landed
                kernel->triple_function(temp, *choice, *aBuf.getActive(), *curB, 0, pMod(grid->i + grid->j + i, s) * localAcols);
                phase_end(t);
                if (s > 1) {
                    t = phase_begin("Dense Cyclic Shift Time");
                    world->event_record(event(3 + i % 2), HNH_STREAM_COMPUTE);
                    if (i < s - 1) {
                        DenseMatrix* tb = &spareB[i % 2];
                        if (i >= 2) world->event_wait(event(3 + (i - 1) % 2), HNH_STREAM_COMM);  // kernel i-1 last read `tb`
                        world->sendrecv(grid->col_world, curB->data(), bbytes, cdst, tb->data(), bbytes, csrc, HNH_STREAM_COMM);
                        curB = tb;
                    }
                    world->event_wait(event(3 + i % 2), HNH_STREAM_COMM);  // kernel i wrote the moving accumulator
                    shiftDenseMatrix(aBuf, grid->row_world, rdst, rsrc, HNH_STREAM_COMM);
                    world->event_record(event(1 + i % 2), HNH_STREAM_COMM);
                    phase_end(t);
                }
            }
            if (s > 1) world->event_wait(event(1 + (s - 1) % 2), HNH_STREAM_COMPUTE);
            auto t = phase_begin("Computation Time");
            aBuf.sync_active();
            phase_end(t);
        }

        if (is_sddmm) {
            if (c > 1) {
                auto t = phase_begin("Sparse Fiber Communication Time");
                CSRLocal* blk = choice->csr_blocks[0];
                // partial dot products of all nonzeros -> this layer's shard, summed over the fiber
                if (blk != nullptr)
                    world->reduce_scatter_v_f64(grid->fiber_world, blk->getActive()->values, sddmm_result_ptr->data(),
                                                choice->layer_coords_sizes, HNH_STREAM_COMPUTE);
                phase_end(t);
                t = phase_begin("Computation Time");
                if (SValues.size())
                    world->check(world->be->hnh_hadamard_f64(world->ctx, sddmm_result_ptr->data(), SValues.data(),
                                                             sddmm_result_ptr->data(), SValues.size(), HNH_STREAM_COMPUTE),
                                 "hnh_hadamard_f64");
                phase_end(t);
            } else {
                auto t = phase_begin("Computation Time");
                choice->hadamardWithCSRValues(SValues, *sddmm_result_ptr);
                phase_end(t);
            }
        }
        choice->reclaimValueArrays();
    }

private:
    static void ensure(DenseMatrix& m, int64_t rows, int64_t cols) {
        if (m.rows() != rows || m.cols() != cols) m = DenseMatrix(rows, cols);
    }
};
