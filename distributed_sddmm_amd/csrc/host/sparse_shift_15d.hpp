// 1.5D sparse-shifting schedule — same class and behaviour as the reference's Sparse15D_Sparse_Shift
// (15D_sparse_shift.hpp): grid (p/c) x c; the dense operands stay put and are split along R over the ring
// of p/c ranks (each rank holds ALL p/c row slabs of its layer but only R*c/p columns, :142-157); the
// sparse block (values + indices) travels around the ring and SDDMM partial dot products accumulate in
// the travelling `values` (:228-269).
//
// MI355X notes.  The reference copies a row slab into `tmp`, runs the kernel, and copies it back
// (:233,236,248); a slab of a row-major matrix is contiguous, so the kernels here work in place on a
// view.  For SpMM the travelling block is read-only, so step i's kernel overlaps the block's send/recv
// into the passive buffer; for SDDMM the kernel writes `values`, so the shift follows the kernel.
#pragma once
#include "distributed_sparse.hpp"

class ShardedBlockRow : public NonzeroDistribution {
public:
    int p, c;
    std::shared_ptr<FlexibleGrid> grid;
    ShardedBlockRow(int M, int N, int p, int c, std::shared_ptr<FlexibleGrid>& grid) {
        world = grid->world;
        this->p = p;
        this->c = c;
        this->grid = grid;
        rows_in_block = divideAndRoundUp(M, p);
        cols_in_block = N;
    }
    int blockOwner(int row_block, int col_block) override {
        (void)col_block;
        return grid->get_global_rank(row_block / c, row_block % c, 0);
    }
};

class Sparse15D_Sparse_Shift : public Distributed_Sparse {
public:
    DenseMatrix accumulation_buffer;
    int blockAwidth, blockBwidth;
    std::vector<int> nnz_in_row_axis, nnz_in_row_axis_tpose;

    Sparse15D_Sparse_Shift(SpmatLocal* S_input, int R, int c, KernelImplementation* k) : Distributed_Sparse(k) {
        this->c = c;
        if (c < 1 || p % c != 0) hnh::fatal("Error, for 1.5D algorithm, must have c divide num_procs!");
        algorithm_name = "1.5D Sparse Shifting Dense Replicating Algorithm";
        proc_grid_names = {"# Rows", "# Layers"};
        perf_counter_keys = {"Replication Time", "Cyclic Shift Time", "Computation Time"};

        grid.reset(new FlexibleGrid(p / c, c, 1, 1));
        r_split = true;
        A_R_split_world = grid->col_world;
        B_R_split_world = grid->col_world;

        this->M = S_input->M;
        this->N = S_input->N;
        ShardedBlockRow standard_dist((int)M, (int)N, p, c, grid);
        ShardedBlockRow transpose_dist((int)N, (int)M, p, c, grid);
        S.reset(S_input->redistribute_nonzeros(&standard_dist, false, false));
        ST.reset(S->redistribute_nonzeros(&transpose_dist, true, false));

        blockAwidth = divideAndRoundUp((int)this->M, p);
        blockBwidth = divideAndRoundUp((int)this->N, p);
        localArows = blockAwidth * p / c;
        localBrows = blockBwidth * p / c;
        setRValue(R);

        S->localize((uint64_t)blockAwidth, 0);
        ST->localize((uint64_t)blockBwidth, 0);

        const int n = p / c;
        nnz_in_row_axis.resize(n);
        nnz_in_row_axis_tpose.resize(n);
        int my_nnz = (int)S->num_tuples(), my_nnz_tpose = (int)ST->num_tuples();
        world->host_allgather_comm(grid->col_world, &my_nnz, nnz_in_row_axis.data(), sizeof(int));
        world->host_allgather_comm(grid->col_world, &my_nnz_tpose, nnz_in_row_axis_tpose.data(), sizeof(int));
        const int max_nnz = *std::max_element(nnz_in_row_axis.begin(), nnz_in_row_axis.end());
        const int max_nnz_tpose = *std::max_element(nnz_in_row_axis_tpose.begin(), nnz_in_row_axis_tpose.end());

        S->own_all_coordinates();
        ST->own_all_coordinates();
        S->monolithBlockColumn();
        ST->monolithBlockColumn();

        // Column count of a block = number of rows of the OTHER dense operand on this rank.  (The reference
        // passes localArows / localBrows the other way round, which is only right for square S —
        // 15D_sparse_shift.hpp:132,134, SURVEY Appendix C #7.)
        S->initializeCSRBlocks(blockAwidth, localBrows * c, max_nnz, false);
        S->release_tuples();
        ST->initializeCSRBlocks(blockBwidth, localArows * c, max_nnz_tpose, false);
        ST->release_tuples();
        publish_ring_max_row(S.get(), grid->col_world);
        publish_ring_max_row(ST.get(), grid->col_world);
        if (std::getenv("HNH_SHIP_INDICES") == nullptr) {  // default: the ring's sparsity structure stays resident, values travel
            S->csr_blocks[0]->replicate_ring_indices(grid->col_world, nnz_in_row_axis);
            ST->csr_blocks[0]->replicate_ring_indices(grid->col_world, nnz_in_row_axis_tpose);
        }
        check_initialized();
    }

    void setRValue(int R) override {
        this->R = R;
        localAcols = R * c / p;
        localBcols = R * c / p;
        if (localAcols * p / c != R) hnh::fatal("Error, R must be divisible by p / c!");
        aSubmatrices.clear();
        bSubmatrices.clear();
        for (int i = 0; i < p / c; i++) {
            aSubmatrices.emplace_back(blockAwidth * (grid->j + c * i), localAcols * grid->i, blockAwidth, localAcols);
            bSubmatrices.emplace_back(blockBwidth * (grid->j + c * i), localBcols * grid->i, blockBwidth, localBcols);
        }
    }

    void initial_shift(DenseMatrix*, DenseMatrix*, KernelMode) override {}  // empty on purpose
    void de_shift(DenseMatrix*, DenseMatrix*, KernelMode) override {}       // empty on purpose

    bool spmm_stores_output() const override { return kernel->stores_fresh_output(); }

    void algorithm(DenseMatrix& localA, DenseMatrix& localB, VectorXd& SValues, VectorXd* sddmm_result_ptr, KernelMode mode,
                   bool initial_replicate) override {
        DenseMatrix *Arole, *Brole;
        SpmatLocal* choice;
        int arBwidth, brBwidth;
        const std::vector<int>* nnz_in_axis;
        if (mode == k_spmmA || mode == k_sddmmA) {
            Arole = &localA; Brole = &localB; choice = S.get();
            arBwidth = blockAwidth; brBwidth = blockBwidth; nnz_in_axis = &nnz_in_row_axis;
        } else {
            Arole = &localB; Brole = &localA; choice = ST.get();
            arBwidth = blockBwidth; brBwidth = blockAwidth; nnz_in_axis = &nnz_in_row_axis_tpose;
        }
        const bool is_sddmm = (mode == k_sddmmA || mode == k_sddmmB);
        const int n = p / c;
        const int64_t cols = Brole->cols();

        // Replicate the dense operand that the sparse block's columns index: p/c slab-wise all-gathers over
        // the layer communicator (15D_sparse_shift.hpp:203-214).
        if (initial_replicate && c > 1) {  // (c == 1: nothing is replicated, and no counter events are spent on an empty phase — small calls are host bound)
            auto t = phase_begin("Replication Time");
            if (accumulation_buffer.rows() != Brole->rows() * c || accumulation_buffer.cols() != cols)
                accumulation_buffer = DenseMatrix(Brole->rows() * c, cols);
            const size_t slab = (size_t)brBwidth * cols;
            for (int i = 0; i < n; i++)
                world->allgather(grid->row_world, Brole->data() + slab * i, accumulation_buffer.data() + slab * c * i,
                                 slab * sizeof(double), HNH_STREAM_COMPUTE);
            phase_end(t);
        }

        // SDDMM: the travelling block accumulates partial dot products (R is split over the ring); its FIRST visit — step 0, at home —
        // may store instead of add when the kernel honours CSRLocal::values_fresh, and then nobody has to zero the values first
        const bool fresh = is_sddmm && kernel->overwrites_fresh_values();
        // SpMM: every slab of the output is produced exactly once below (`tmp *= 0.0` in the reference): a kernel that stores fresh output
        // rows needs no zero fill at all (the wrapper skipped its own, spmm_stores_output()); otherwise one fill, the wrapper's or this one
        const bool zeroed = take_output_zeroed(*Arole), unset = take_output_unset(*Arole);
        const bool store_out = !is_sddmm && unset && kernel->stores_fresh_output();
        if (!(is_sddmm && fresh)) {  // (a storing first visit needs no preparation at all)
            auto t = phase_begin("Computation Time");
            if (is_sddmm) choice->setValuesConstant(0.0);
            else {
                choice->setCSRValues(SValues);
                if (!zeroed && !store_out) Arole->setZero();
            }
            phase_end(t);
        }

        DenseMatrix& gathered = (c > 1) ? accumulation_buffer : *Brole;
        CSRLocal* blk = choice->csr_blocks[0];
        const int src = pMod(grid->i - 1, n), dst = pMod(grid->i + 1, n);
        if (n > 1) order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);

        // One event per phase boundary (phase_end_mark / phase_begin_at): "kernel i done" ends step i's computation phase, orders the
        // shift behind it and begins step i + 1's computation phase; "shift i landed" ends its shift phase, orders kernel i + 1 and begins
        // the next shift phase — 2 records per step instead of 6 (the reference's counters: distributed_sparse.h:212-223 around the
        // same regions, 15D_sparse_shift.hpp:226-262).
        void* kernel_done[2] = {nullptr, nullptr};  // by step parity
        void* landed = nullptr;
        for (int i = 0; i < n; i++) {
            auto t = phase_begin_at("Computation Time", i > 0 ? kernel_done[(i - 1) % 2] : nullptr);
            const int block_id = pMod(grid->i - i, n);
            DenseMatrix slab = DenseMatrix::view(Arole->data() + (size_t)block_id * arBwidth * Arole->cols(), arBwidth, Arole->cols());
            if (i > 0) world->event_wait(landed, HNH_STREAM_COMPUTE);  // shift i-1 landed
            blk->values_fresh = fresh && i == 0;
            blk->out_fresh = store_out;
            kernel->triple_function(mode == k_spmmB ? k_spmmA : mode, *choice, slab, gathered, 0, 0);
            blk->values_fresh = blk->out_fresh = false;
            if (n == 1) {
                phase_end(t);
                break;
            }
            kernel_done[i % 2] = phase_end_mark(t, 3 + i % 2);

            t = phase_begin_at("Cyclic Shift Time", landed);
            // SDDMM writes the travelling values: ship after this step's kernel.  SpMM only reads the
            // block: ship concurrently with this step's kernel, but not before the previous kernel has
            // released the passive buffer.
            if (is_sddmm) world->event_wait(kernel_done[i % 2], HNH_STREAM_COMM);
            else if (i >= 1) world->event_wait(kernel_done[(i - 1) % 2], HNH_STREAM_COMM);
            blk->shiftCSR(src, dst, grid->col_world, (*nnz_in_axis)[pMod(grid->i - i - 1, n)], 72, is_sddmm ? coo : csr,
                          HNH_STREAM_COMM, pMod(grid->i - i - 1, n));
            choice->blockStarts[1] = blk->num_coords;
            landed = phase_end_mark(t, 1 + i % 2);
        }
        if (n > 1) world->event_wait(landed, HNH_STREAM_COMPUTE);  // block is home again

        if (is_sddmm) {
            auto t = phase_begin("Computation Time");
            choice->hadamardWithCSRValues(SValues, *sddmm_result_ptr);
            phase_end(t);
        }
    }
};
