// 1.5D dense-shifting schedule — same class and behaviour as the reference's Sparse15D_Dense_Shift
// (15D_dense_shift.hpp): grid (p/c) x c, S stationary in block rows replicated over the c layers, one
// dense operand all-gathered over the layer ("row") communicator, the other cyclically shifted around the
// ring of p/c ranks ("col" communicator).  fusionApproach 1 = replication reuse, 2 = local kernel fusion.
//
// Ring step on MI355X.  The reference does  kernel -> MPI_Sendrecv -> MPI_Barrier  per step
// (:343-356).  Here the moving operand is triple-buffered when it is read-only (SDDMM, and both kernels
// of approach 2): step i's kernel reads buffer `cur` on the compute stream while the communication
// stream already sends `cur` to ring rank i+1 and receives step i+1's operand into the next spare buffer
// (one RCCL send/recv group over one xGMI link per direction).  The caller's matrix is never overwritten,
// so only n-1 shifts are needed instead of n and nothing is copied back.  When the moving buffer is the
// SpMM accumulator (approach 1, :331-337) a step's shift has to wait for that step's kernel, so the ring
// degenerates to kernel -> shift -> kernel, as in the reference, with n shifts and a storage swap at the end.
//
// Relay ring vs mesh fetch.  A neighbour ring forwards every block n-1 times over ONE xGMI link per GPU
// (store and forward), although the 8 GPUs of a node form a full mesh with 7 links each.  Because a
// read-only moving operand never changes, step t can equally well take the block straight from its owner,
// ring rank (me - t): `ring_mode == kMeshFetch` issues all n-1 owner->consumer transfers at the start of
// the pass as ONE RCCL group (n-1 explicit-peer send/recv pairs over n-1 different links, each block
// crossing exactly one link) and runs the same kernel sequence on the same blocks.  The visiting order,
// the block decomposition and the arithmetic are those of the reference's cyclic shift; only the route
// differs.  HNH_RING_MODE=relay|mesh selects it (default mesh); the read-write ring always relays.
#pragma once
#include "distributed_sparse.hpp"

class ShardedBlockCyclicColumn : public NonzeroDistribution {
public:
    int p, c;
    std::shared_ptr<FlexibleGrid> grid;
    ShardedBlockCyclicColumn(int M, int N, int p, int c, std::shared_ptr<FlexibleGrid>& grid) {
        world = grid->world;
        this->p = p;
        this->c = c;
        this->grid = grid;
        rows_in_block = divideAndRoundUp(M, p) * c;
        cols_in_block = divideAndRoundUp(N, p);
    }
    int blockOwner(int row_block, int col_block) override { return grid->get_global_rank(row_block, col_block % c, 0); }
};

class Sparse15D_Dense_Shift : public Distributed_Sparse {
public:
    int fusionApproach;
    DenseMatrix accumulation_buffer;  // replicated stationary operand (approach 1) / replicated output (approach 2)
    DenseMatrix broadcast_buffer;     // approach-2 fused: replicated SDDMM row operand when c > 1
    DenseMatrix ring_spare[2];        // persistent spare buffers of the moving operand
    std::vector<DenseMatrix> mesh_spare;  // mesh fetch: one landing buffer per remote ring position
    enum RingMode { kRelay, kMeshFetch };
    RingMode ring_mode;
    // Column chunks (approach 2): every block column of S is cut into `chunks` column ranges = row ranges of the visiting
    // dense block; sub-block (b, q) is csr_blocks[b * chunks + q].  Two uses:
    //  * several ranks: the fetch is issued chunk by chunk and the kernels of chunk q run while chunk q+1 is still on the
    //    links (default 4 chunks);
    //  * one rank per ring (p == c, e.g. a single GPU), only when forced with HNH_MESH_CHUNKS: one launch per chunk.  This is
    //    how the Infinity-Cache panel effect was found (16.8 -> 14.8 ms at config 2 with 2 chunks,
    //    profiles/r01_panel_probe_same_box.log); by default a ring of one keeps whole blocks and the kernel library cuts
    //    the same panels itself from the sorted CSR rows (hnh_kernels.h, `cols` hint), for every schedule.
    // HNH_MESH_CHUNKS overrides the count (1 = whole blocks).
    int chunks = 1;
    int chunkA = 0, chunkB = 0;  // rows per chunk of a visiting A / B block

    // hold_moving_operand(): the remote blocks of this matrix stay valid in mesh_spare / ring_spare[0] between calls
    const double* held_ptr = nullptr;
    bool held_in_mesh = false, held_in_ring = false;  // which landing buffers currently hold its remote blocks

    void hold_moving_operand(const DenseMatrix* m) override {
        if (std::getenv("HNH_NO_HOLD") != nullptr) return;  // A/B switch for measurements
        held_ptr = m ? m->data() : nullptr;
        held_in_mesh = held_in_ring = false;
    }
    void release_moving_operand() override {
        held_ptr = nullptr;
        held_in_mesh = held_in_ring = false;
    }

    Sparse15D_Dense_Shift(SpmatLocal* S_input, int R, int c, int fusionApproach, KernelImplementation* k) : Distributed_Sparse(k) {
        this->fusionApproach = fusionApproach;
        this->c = c;
        ring_mode = kMeshFetch;
        if (const char* m = std::getenv("HNH_RING_MODE")) {
            if (std::string(m) == "relay") ring_mode = kRelay;
            else if (std::string(m) != "mesh") hnh::fatal("Error, HNH_RING_MODE must be relay or mesh!");
        }
        if (c < 1 || p % c != 0) hnh::fatal("Error, for 1.5D algorithm, must have c divide num_procs!");
        if (fusionApproach != 1 && fusionApproach != 2) hnh::fatal("Error, fusion approach must be 1 or 2!");

        algorithm_name = "1.5D Block Row Replicated S Striped AB Cyclic Shift";
        proc_grid_names = {"# Rows", "# Layers"};
        perf_counter_keys = {"Replication Time", "Cyclic Shift Time", "Computation Time"};

        grid.reset(new FlexibleGrid(p / c, c, 1, 1));
        r_split = false;
        this->M = S_input->M;
        this->N = S_input->N;

        ShardedBlockCyclicColumn standard_dist((int)M, (int)N, p, c, grid);
        ShardedBlockCyclicColumn transpose_dist((int)N, (int)M, p, c, grid);

        // private copies of the nonzeros in the layout this schedule wants; the caller's matrix is untouched
        S.reset(S_input->redistribute_nonzeros(&standard_dist, false, false));
        ST.reset(S->redistribute_nonzeros(&transpose_dist, true, false));

        localArows = divideAndRoundUp((int)this->M, p);
        localBrows = divideAndRoundUp((int)this->N, p);
        setRValue(R);

        if (fusionApproach == 2) {
            if (p / c > 1) chunks = 4;  // a ring of one keeps whole blocks: the kernel library cuts its own cache panels
            if (const char* q = std::getenv("HNH_MESH_CHUNKS")) chunks = std::atoi(q);
            if (chunks < 1 || chunks > 8) hnh::fatal("Error, HNH_MESH_CHUNKS must be between 1 and 8!");
        }
        chunkA = divideAndRoundUp(localArows, chunks);
        chunkB = divideAndRoundUp(localBrows, chunks);

        const uint64_t arows = (uint64_t)localArows * c, brows = (uint64_t)localBrows * c;
        S->localize(arows, 0);
        S->divideIntoBlockCols(localBrows, p, true, chunks, chunkB);
        ST->localize(brows, 0);
        ST->divideIntoBlockCols(localArows, p, true, chunks, chunkA);

        S->own_all_coordinates();
        ST->own_all_coordinates();

        const bool local_tpose = (fusionApproach == 1);
        S->initializeCSRBlocks(localArows * c, chunkB, -1, local_tpose);
        S->release_tuples();
        ST->initializeCSRBlocks(localBrows * c, chunkA, -1, local_tpose);
        ST->release_tuples();
        check_initialized();
    }

    void setRValue(int R) override {
        this->R = R;
        localAcols = R;
        localBcols = R;
        aSubmatrices.clear();
        bSubmatrices.clear();
        aSubmatrices.emplace_back(localArows * (c * grid->i + grid->j), 0, localArows, localAcols);
        bSubmatrices.emplace_back(localBrows * (c * grid->i + grid->j), 0, localBrows, localBcols);
    }

    void initial_shift(DenseMatrix*, DenseMatrix*, KernelMode) override {}  // empty on purpose
    void de_shift(DenseMatrix*, DenseMatrix*, KernelMode) override {}       // empty on purpose

    VectorXd like_S_values(double value) override {
        SpmatLocal* s = (fusionApproach == 1) ? ST.get() : S.get();
        return VectorXd::Constant(s->owned_coords_end - s->owned_coords_start, value);
    }
    VectorXd like_ST_values(double value) override {
        SpmatLocal* s = (fusionApproach == 1) ? S.get() : ST.get();
        return VectorXd::Constant(s->owned_coords_end - s->owned_coords_start, value);
    }

    // Block visited at ring step i (15D_dense_shift.hpp:326)
    int block_at(int i) const { return pMod((grid->rankInCol - i) * c + grid->rankInRow, p); }

    // ---- local kernel fusion (approach 2): SDDMM and SpMM on every visiting block in ONE pass of shifts
    void fusedSpMM(DenseMatrix& localA, DenseMatrix& localB, VectorXd& Svalues, VectorXd& sddmm_buffer, MatMode mode) override {
        if (fusionApproach == 1) {
            Distributed_Sparse::fusedSpMM(localA, localB, Svalues, sddmm_buffer, mode);
            return;
        }
        // NB: like the reference (15D_dense_shift.hpp:189,250-251) this path neither multiplies by Svalues
        // nor fills sddmm_buffer; it equals the generic path when Svalues == 1 (true in every app/benchmark).
        fused_pass(mode == Amat ? localA : localB, mode == Amat ? localB : localA, mode == Amat ? S.get() : ST.get(), nullptr, 0u, nullptr);
    }

    bool fusedSpMM_out(DenseMatrix& localA, DenseMatrix& localB, MatMode mode, DenseMatrix& Out, bool leaky,
                       const hnh_fused_extras& extras) override {
        if (fusionApproach != 2) return false;
        DenseMatrix& Xin = (mode == Amat) ? localA : localB;
        if (Out.rows() != Xin.rows() || Out.cols() != Xin.cols() || Out.data() == Xin.data())
            hnh::fatal("Error, fusedSpMM_out needs a separate output of the input's shape!");
        fused_pass(Xin, mode == Amat ? localB : localA, mode == Amat ? S.get() : ST.get(), &Out, leaky ? HNH_FUSED_LEAKY_RELU : 0u, &extras);
        return true;
    }

private:
    // One pass of shifts with the fused kernel on every visiting block.  target == nullptr: the result replaces
    // Xin (the reference's in-place fusedSpMM); otherwise it is written to *target and Xin survives.
    void fused_pass(DenseMatrix& Xin, DenseMatrix& moving, SpmatLocal* choice, DenseMatrix* target, unsigned act_flag,
                    const hnh_fused_extras* extras) {
        DenseMatrix* Arole = &Xin;
        DenseMatrix* Brole = &moving;
        const int n = p / c;
        const bool epilogue = KernelImplementation::wants_epilogue(extras);
        if (epilogue && target == nullptr) hnh::fatal("Error, a row epilogue needs the input rows: use fusedSpMM_out!");
        hnh_fused_extras act_only = {extras ? extras->leaky_alpha : 0.0, 0.0, nullptr};
        const hnh_fused_extras* act = act_flag ? &act_only : nullptr;
        // c == 1: every output row is finished by this rank's own launches, so the last of them also runs the
        // epilogue; c > 1: the partial outputs are reduce-scattered first
        const hnh_fused_extras* last = (c == 1 && epilogue) ? extras : act;

        // with c == 1 and a target the kernels accumulate straight into it
        DenseMatrix* accum = (c == 1 && target != nullptr) ? target : &accumulation_buffer;
        if (accum == &accumulation_buffer) ensure(accumulation_buffer, Arole->rows() * c, R);
        DenseMatrix* rowOperand = Arole;
        if (c > 1) {
            auto t = phase_begin("Replication Time");
            ensure(broadcast_buffer, Arole->rows() * c, R);
            world->allgather(grid->row_world, Arole->data(), broadcast_buffer.data(), (size_t)Arole->size() * sizeof(double),
                             HNH_STREAM_COMPUTE);
            rowOperand = &broadcast_buffer;
            phase_end(t);
        }

        const unsigned base = HNH_FUSED_VALUES_OVERWRITE | act_flag;
        const int cw = (choice == S.get()) ? chunkB : chunkA;  // rows per chunk of the visiting blocks
        bool out_fresh = true;
        // the fused kernel on the sub-blocks (ring step i, chunk q) of `steps` x `qs`, all in ONE launch
        auto launch = [&](const std::vector<std::pair<int, DenseMatrix*>>& steps, int q_begin, int q_end, const hnh_fused_extras* ex) {
            std::vector<DenseMatrix> views;
            std::vector<DenseMatrix*> Ys;
            std::vector<int> ids;
            views.reserve(steps.size() * (size_t)(q_end - q_begin));
            bool any = false;
            for (auto& st : steps)
                for (int q = q_begin; q < q_end; q++) {
                    const int id = block_at(st.first) * chunks + q;
                    views.push_back(chunk_view(*st.second, q, cw));
                    Ys.push_back(&views.back());
                    ids.push_back(id);
                    any = any || choice->csr_blocks[id] != nullptr;
                }
            if (!any && ex == act) return;  // nothing to multiply and no epilogue to run
            kernel->fused_multi_local(*choice, *rowOperand, Ys, *accum, ids, base | (out_fresh ? HNH_FUSED_OUT_OVERWRITE : 0u), ex);
            out_fresh = false;
        };

        if (ring_mode == kMeshFetch && n > 1) {
            // the remote blocks arrive chunk by chunk over all links at once: local block while chunk 0 flies, then
            // chunk q of ALL remote blocks in one launch while chunk q+1 is still on the links
            std::vector<DenseMatrix*> fetched = mesh_fetch_chunked(Brole, n, cw);
            auto t = phase_begin("Computation Time");
            launch({{0, Brole}}, 0, chunks, act);
            std::vector<std::pair<int, DenseMatrix*>> remote;
            for (int i = 1; i < n; i++) remote.push_back({i, fetched[i - 1]});
            for (int q = 0; q < chunks; q++) {
                world->event_wait(event(8 + q), HNH_STREAM_COMPUTE);  // chunk q of every remote block has landed
                launch(remote, q, q + 1, q == chunks - 1 ? last : act);
            }
            phase_end(t);
        } else {
            ring_readonly(Brole, n, [&](int i, DenseMatrix& cur) {
                auto t = phase_begin("Computation Time");
                const hnh_fused_extras* ex = (i == n - 1) ? last : act;
                if (chunks == 1 || n == 1) {
                    // the plain row kernel, one launch per block — or, on a ring of one, per column chunk: the panel of
                    // `cur` that a launch gathers from then fits the Infinity Cache
                    for (int q = 0; q < chunks; q++) {
                        const int block_id = block_at(i) * chunks + q;
                        const hnh_fused_extras* exq = (q == chunks - 1) ? ex : act;
                        if (choice->csr_blocks[block_id] == nullptr && exq == act) continue;
                        DenseMatrix part = chunk_view(cur, q, cw);
                        kernel->fused_local(*choice, *rowOperand, part, *accum, block_id, base | (out_fresh ? HNH_FUSED_OUT_OVERWRITE : 0u), exq);
                        out_fresh = false;
                    }
                } else {
                    launch({{i, &cur}}, 0, chunks, ex);
                }
                phase_end(t);
            });
        }
        if (out_fresh) accum->setZero();  // no block on this rank had a nonzero

        if (c > 1) {
            auto t = phase_begin("Replication Time");
            DenseMatrix* dest = target ? target : Arole;
            world->reduce_scatter_f64(grid->row_world, accumulation_buffer.data(), dest->data(), (size_t)Arole->rows() * R, HNH_STREAM_COMPUTE);
            phase_end(t);
            t = phase_begin("Computation Time");
            if (epilogue) KernelImplementation::row_epilogue(world, *Arole, *dest, extras);
            phase_end(t);
        } else if (target == nullptr) {
            auto t = phase_begin("Computation Time");
            if (Arole->owns_storage()) Arole->swap(accumulation_buffer);  // `*Arole = accumulation_buffer` without the copy
            else *Arole = accumulation_buffer;
            phase_end(t);
        }
    }

public:
    // SDDMM, SpMM with A as the output, or SpMM with B as the output (15D_dense_shift.hpp:276-384)
    void algorithm(DenseMatrix& localA, DenseMatrix& localB, VectorXd& SValues, VectorXd* sddmm_result_ptr, KernelMode mode,
                   bool initial_replicate) override {
        DenseMatrix *Arole, *Brole;
        SpmatLocal* choice;
        const bool invert = (fusionApproach == 1);
        if ((mode == k_spmmA || mode == k_sddmmA) == invert) {
            Arole = &localB;
            Brole = &localA;
            choice = ST.get();
        } else {
            Arole = &localA;
            Brole = &localB;
            choice = S.get();
        }
        const bool is_sddmm = (mode == k_sddmmA || mode == k_sddmmB);
        const int n = p / c;

        if (initial_replicate && c > 1) {
            auto t = phase_begin("Replication Time");
            ensure(accumulation_buffer, Arole->rows() * c, R);
            world->allgather(grid->row_world, Arole->data(), accumulation_buffer.data(), (size_t)Arole->size() * sizeof(double),
                             HNH_STREAM_COMPUTE);
            phase_end(t);
        }

        {
            auto t = phase_begin("Computation Time");
            if (is_sddmm) choice->setValuesConstant(0.0);
            else choice->setCSRValues(SValues);
            phase_end(t);
        }

        KernelMode mode_temp = mode;
        if (fusionApproach == 1 && mode == k_spmmA) mode_temp = k_spmmB;
        if (fusionApproach == 2 && mode == k_spmmB) mode_temp = k_spmmA;
        DenseMatrix& stationary = (c > 1) ? accumulation_buffer : *Arole;

        const int cw = (choice == S.get()) ? chunkB : chunkA;
        auto step = [&](int i, DenseMatrix& cur) {
            auto t = phase_begin("Computation Time");
            if (chunks == 1) {
                kernel->triple_function(mode_temp, *choice, stationary, cur, block_at(i), 0);
            } else {  // approach 2 keeps S in column chunks of each block: same kernels on each chunk's rows of `cur`
                for (int q = 0; q < chunks; q++) {
                    DenseMatrix part = chunk_view(cur, q, cw);
                    kernel->triple_function(mode_temp, *choice, stationary, part, block_at(i) * chunks + q, 0);
                }
            }
            phase_end(t);
        };

        // the moving operand is written only when it is the SpMM accumulator (approach 1)
        const bool moving_readonly = is_sddmm || fusionApproach == 2;
        if (moving_readonly && chunks > 1 && ring_mode == kMeshFetch && n > 1) {
            // chunk-pipelined mesh fetch, as in the fused pass: local block while chunk 0 flies, then chunk q of every
            // remote block while chunk q+1 is still on the links (same kernels on the same sub-blocks; only the order of
            // the steps differs from the reference's block-by-block walk)
            std::vector<DenseMatrix*> fetched = mesh_fetch_chunked(Brole, n, cw);
            auto t = phase_begin("Computation Time");
            auto one = [&](int i, DenseMatrix& blk, int q) {
                DenseMatrix part = chunk_view(blk, q, cw);
                kernel->triple_function(mode_temp, *choice, stationary, part, block_at(i) * chunks + q, 0);
            };
            for (int q = 0; q < chunks; q++) one(0, *Brole, q);
            for (int q = 0; q < chunks; q++) {
                world->event_wait(event(8 + q), HNH_STREAM_COMPUTE);  // chunk q of every remote block has landed
                for (int i = 1; i < n; i++) one(i, *fetched[i - 1], q);
            }
            phase_end(t);
        } else if (moving_readonly) {
            ring_readonly(Brole, n, step);
        } else {
            ring_readwrite(Brole, n, step);
        }

        if (is_sddmm) {
            auto t = phase_begin("Computation Time");
            choice->hadamardWithCSRValues(SValues, *sddmm_result_ptr);  // result = SValues .* block values
            phase_end(t);
        }

        if (fusionApproach == 2 && !is_sddmm && c > 1) {
            auto t = phase_begin("Replication Time");
            world->reduce_scatter_f64(grid->row_world, accumulation_buffer.data(), Arole->data(), (size_t)Arole->rows() * R,
                                      HNH_STREAM_COMPUTE);
            phase_end(t);
        }
    }

private:
    static void ensure(DenseMatrix& m, int64_t rows, int64_t cols) {
        if (m.rows() != rows || m.cols() != cols) m = DenseMatrix(rows, cols);
    }

    // n kernel steps over a READ-ONLY moving operand, n-1 overlapped shifts, caller's buffer untouched.
    template <typename Step>
    void ring_readonly(DenseMatrix* start, int n, Step&& step) {
        if (ring_mode == kMeshFetch && n > 2) {  // with two ranks the relay ring already is a single direct transfer
            mesh_readonly(start, n, step);
            return;
        }
        if (n > 1) {
            for (auto& s : ring_spare) ensure(s, start->rows(), start->cols());
            order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);  // inputs (and earlier readers of the spares) are done
            if (!(n == 2 && held_ptr == start->data())) held_in_ring = false;  // the spares are about to hold other blocks
        }
        const int dst = pMod(grid->rankInCol + 1, n), src = pMod(grid->rankInCol - 1, n);
        const size_t bytes = (size_t)start->size() * sizeof(double);
        DenseMatrix* cur = start;
        for (int i = 0; i < n; i++) {
            if (i > 0) world->event_wait(event(1 + (i - 1) % 2), HNH_STREAM_COMPUTE);  // shift i-1 landed
            step(i, *cur);
            if (i < n - 1) {
                auto t = phase_begin("Cyclic Shift Time");
                world->event_record(event(3 + i % 2), HNH_STREAM_COMPUTE);                 // kernel i enqueued
                DenseMatrix* target = &ring_spare[i % 2];
                if (i >= 2) world->event_wait(event(3 + (i - 1) % 2), HNH_STREAM_COMM);    // kernel i-1 last read `target`
                const bool held = (n == 2 && held_ptr == start->data());  // a ring of two: the one remote block can stay
                if (!(held && held_in_ring))
                    world->sendrecv(grid->col_world, cur->data(), bytes, dst, target->data(), bytes, src, HNH_STREAM_COMM);
                if (held) held_in_ring = true;
                world->event_record(event(1 + i % 2), HNH_STREAM_COMM);
                cur = target;
                phase_end(t);
            }
        }
    }

    // rows [q * cw, (q + 1) * cw) of a visiting block (clipped; possibly empty)
    static DenseMatrix chunk_view(DenseMatrix& block, int q, int cw) {
        const int64_t r0 = std::min<int64_t>((int64_t)q * cw, block.rows());
        const int64_t r1 = std::min<int64_t>(r0 + cw, block.rows());
        return DenseMatrix::view(block.data() + r0 * block.cols(), r1 - r0, block.cols());
    }

    // mesh_fetch_all in `chunks` row ranges: group q moves rows [q*cw, (q+1)*cw) of every remote block (all n-1 links
    // busy in every group) and event(8 + q) is recorded behind it, so consumers can start on chunk q while q+1 flies.
    std::vector<DenseMatrix*> mesh_fetch_chunked(DenseMatrix* start, int n, int cw) {
        if ((int)mesh_spare.size() < n - 1) mesh_spare.resize(n - 1);
        for (int k = 0; k < n - 1; k++) ensure(mesh_spare[k], start->rows(), start->cols());
        auto t = phase_begin("Cyclic Shift Time");
        order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);  // my block is final; earlier readers of the landing buffers are done
        const bool held = (held_ptr == start->data());
        if (!held) held_in_mesh = false;  // another operand lands in the buffers: a held one has to be fetched again
        for (int q = 0; q < chunks; q++) {
            const int64_t r0 = std::min<int64_t>((int64_t)q * cw, start->rows()), r1 = std::min<int64_t>(r0 + cw, start->rows());
            const size_t off = (size_t)r0 * start->cols(), bytes = (size_t)(r1 - r0) * start->cols() * sizeof(double);
            if (bytes > 0 && !(held && held_in_mesh)) {  // a held operand's blocks are still in the landing buffers
                world->group_begin();
                for (int k = 1; k < n; k++)  // my block is what ring rank me+k needs at ITS step k; I need the block of me-k at mine
                    world->sendrecv(grid->col_world, start->data() + off, bytes, pMod(grid->rankInCol + k, n), mesh_spare[k - 1].data() + off,
                                    bytes, pMod(grid->rankInCol - k, n), HNH_STREAM_COMM);
                world->group_end();
            }
            world->event_record(event(8 + q), HNH_STREAM_COMM);
        }
        if (held) held_in_mesh = true;
        phase_end(t);
        std::vector<DenseMatrix*> out;
        for (int k = 0; k < n - 1; k++) out.push_back(&mesh_spare[k]);
        return out;
    }

    // Issues all n-1 owner->consumer transfers of a read-only moving operand as one group on the communication
    // stream and records event(1) behind them; returns the landing buffers in visiting order (step 1 .. n-1).
    std::vector<DenseMatrix*> mesh_fetch_all(DenseMatrix* start, int n) {
        if ((int)mesh_spare.size() < n - 1) mesh_spare.resize(n - 1);
        for (int k = 0; k < n - 1; k++) ensure(mesh_spare[k], start->rows(), start->cols());
        const size_t bytes = (size_t)start->size() * sizeof(double);
        auto t = phase_begin("Cyclic Shift Time");
        order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);  // my block is final; earlier readers of the landing buffers are done
        const bool held = (held_ptr == start->data());
        if (!held) held_in_mesh = false;  // another operand lands in the buffers: a held one has to be fetched again
        if (!(held && held_in_mesh)) {  // a held operand's blocks are still in the landing buffers from the previous call
            world->group_begin();
            for (int k = 1; k < n; k++)  // my block is what ring rank me+k needs at ITS step k; I need the block of me-k at mine
                world->sendrecv(grid->col_world, start->data(), bytes, pMod(grid->rankInCol + k, n), mesh_spare[k - 1].data(), bytes,
                                pMod(grid->rankInCol - k, n), HNH_STREAM_COMM);
            world->group_end();
        }
        if (held) held_in_mesh = true;
        world->event_record(event(1), HNH_STREAM_COMM);
        phase_end(t);
        std::vector<DenseMatrix*> out;
        for (int k = 0; k < n - 1; k++) out.push_back(&mesh_spare[k]);
        return out;
    }

    // Same n kernel steps on the same blocks, but every block comes straight from its owner: the transfers
    // overlap with step 0's kernel.
    template <typename Step>
    void mesh_readonly(DenseMatrix* start, int n, Step&& step) {
        std::vector<DenseMatrix*> fetched = mesh_fetch_all(start, n);
        for (int i = 0; i < n; i++) {
            if (i == 1) world->event_wait(event(1), HNH_STREAM_COMPUTE);  // the remote blocks have landed
            step(i, i == 0 ? *start : *fetched[i - 1]);
        }
    }

    // n kernel steps that WRITE the moving operand, n shifts, result handed back to the caller's matrix.
    template <typename Step>
    void ring_readwrite(DenseMatrix* start, int n, Step&& step) {
        hnh::BufferPair bBuf(start, &ring_spare[0]);
        if (n > 1) held_in_ring = false;  // ring_spare[0] is overwritten
        const int dst = pMod(grid->rankInCol + 1, n), src = pMod(grid->rankInCol - 1, n);
        for (int i = 0; i < n; i++) {
            step(i, *bBuf.getActive());
            if (n > 1) {
                auto t = phase_begin("Cyclic Shift Time");
                order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);
                shiftDenseMatrix(bBuf, grid->col_world, dst, src, HNH_STREAM_COMM);
                order(HNH_STREAM_COMM, HNH_STREAM_COMPUTE, 5);
                phase_end(t);
            }
        }
        auto t = phase_begin("Computation Time");
        bBuf.sync_active();
        phase_end(t);
    }
};
