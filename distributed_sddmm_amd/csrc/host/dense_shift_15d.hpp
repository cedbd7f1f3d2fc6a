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
    DenseMatrix ring_spare[2];        // relay ring: persistent spare buffers of the moving operand
    enum RingMode { kRelay, kMeshFetch };
    RingMode ring_mode;
    // the read-WRITE ring (SpMM accumulator of replication reuse) in two row halves, one half's shift under the other half's kernel
    // (ring_readwrite_halves); HNH_ACC_HALVES=0: the reference's kernel -> shift sequence
    bool acc_halves = true;

    // Merged layout (approach 2 under the mesh fetch, more than one rank per ring).  The rank's S is kept as TWO blocks:
    //   csr_blocks[0]  the block column it owns (visited at step 0, gathers from the caller's own dense block), and
    //   csr_blocks[1]  ALL block columns visited at steps 1 .. n-1 as one CSR block whose column index is the row of the
    //                  LANDING BUFFER the fetched dense rows arrive in.
    // The landing buffer is chunk-major: the visiting blocks are cut into `windows` row chunks [cut[q], cut[q+1]) (cutA / cutB for
    // visiting A / B blocks), and chunk q of the blocks of steps 1 .. n-1 lies at rows [(n-1) cut[q], (n-1) cut[q+1]).  The
    // chunks are TAPERED — first and last half as high as the ones between, (1, 2, .., 2, 1) / (2Q - 2) of a block — because what
    // cannot overlap is the wait for the first chunk and the kernel of the last one, whichever of fetch and kernels is slower.  Fetch
    // group q moves chunk q of every remote block (all n-1 links busy) and one WINDOWED row pass over csr_blocks[1]
    // (hnh_csr_window: the columns of chunk q are a contiguous piece of every CSR row) runs as soon as it has landed, while
    // chunk q+1 is still on the links.  The reference walks the same nonzeros block by block (15D_dense_shift.hpp:199-227);
    // here a row's nonzeros of all fetched blocks are consecutive, so the gather batches of the row kernel stay full however
    // many ranks and chunks there are.  HNH_MESH_CHUNKS = Q symmetric chunks (1 = whole blocks), HNH_MESH_TAPER = any heights;
    // default (1, 2, 2, 2, 1, 1).
    // ADAPTIVE WINDOWS (round 5).  Every windowed pass re-reads the row operand and read-modify-writes the output rows (3 dense rows per
    // sparse row and window: counter traffic 1.17 x the byte model at the default six chunks against 1.05 x for one pass over all
    // fetched blocks, profiles/r05_rank_share_p8_counter_traffic.log) — the price of starting before everything has landed.  It is
    // only worth paying for chunks that have NOT landed: the host therefore decides each pass's window RANGE as late as it can — when
    // pass k - 2 has completed, i.e. while pass k - 1 runs and with one pass always queued behind it — and lets the pass cover every
    // chunk whose arrival event has completed by then, not just the next one.  Links slower than the kernels: the passes follow the
    // chunks one by one, as before; links faster than the kernels (or blocks that are already resident): the later chunks are taken in
    // one pass.  The order of the nonzeros within a row, and with it every result bit, is the same for any grouping of consecutive
    // windows.  The host blocks inside the call for this (the reference's calls are synchronous too); HNH_WINDOW_MERGE=0: one pass per
    // chunk whatever has landed (round 2-4 behaviour), HNH_WINDOW_MERGE_CAP=n: at most n chunks per pass.
    bool merge_windows = true;
    int merge_cap = 1 << 20;
    bool merged = false;
    // ROW-MERGED layout (round 6: approach 1 — replication reuse — under the mesh fetch).  The visiting blocks of approach 1 are
    // stored transposed (15D_dense_shift.hpp:141: rows of a block = rows of the MOVING operand, columns = rows of the stationary one),
    // so the same relabelling of the moving operand's rows to "own block | landing-buffer row" makes all fetched blocks ONE transposed
    // CSR block whose ROWS are the rows of the chunk-major landing buffer: chunk q of every fetched block is the contiguous row range
    // [(n-1) cut[q], (n-1) cut[q+1]).
    //   SDDMM (moving operand read-only): mesh fetch as in approach 2, then own block + one pass per ROW RANGE of landed chunks
    //   (CSRLocal::select_row_range) — rows, not column windows, so nothing is re-streamed and taking several landed chunks in one
    //   pass costs nothing (the reference: one kernel per visiting block behind each shift, :343-356);
    //   SpMM (moving operand = the accumulator): MESH REDUCE-SCATTER instead of the ring — the rank computes its partial result for every
    //   block into a chunk-major STAGING buffer, one pass per chunk, each followed by ONE group of n-1 explicit-peer transfers that
    //   carries chunk q of every partial block straight to its owner (n-1 links at once; the ring moves the same bytes over one link,
    //   block after block) while the next chunk's pass runs; the own block's pass accumulates into the caller's matrix while the last
    //   group flies; the owner adds the n-1 partial blocks it received in ring-step order, chunk by chunk as they land
    //   (hnh_sum_chunked_blocks_f64) — a fixed order, so results are bit-identical run to run.
    // HNH_FUSION1_MESH=0 (or HNH_RING_MODE=relay, or a kernel plugin without handles_row_ranges()): the block-by-block ring of rounds 1-5.
    bool row_merged = false;
    DenseMatrix staging[2];      // [slot]: partial results for the n-1 other owners, chunk-major like the landing buffer
    int windows = 1;
    std::vector<int> taper;  // chunk heights in units of a "fine" chunk; empty = the symmetric default (1, 2, .., 2, 1)
    std::vector<int64_t> cutA, cutB;  // chunk boundaries (windows + 1 entries, rows of a visiting A / B block)
    int fineA = 0, fineB = 0;         // granularity of the boundaries: a chunk is w_q "fine" chunks of this many rows (w_q = its weight, 1 .. 64)
    DenseMatrix landing[2];      // [0]: visiting B blocks (gathered through S), [1]: visiting A blocks (through ST)

    // hold_moving_operand(): the remote blocks of this matrix stay valid in a landing buffer / ring_spare[0] between calls
    const double* held_ptr = nullptr;
    int held_slot = -1;         // landing buffer that currently holds its remote blocks, or -1
    bool held_in_ring = false;  // ... or ring_spare[0] (relay ring of two)

    void hold_moving_operand(const DenseMatrix* m) override {
        if (std::getenv("HNH_NO_HOLD") != nullptr) return;  // A/B switch for measurements
        held_ptr = m ? m->data() : nullptr;
        held_slot = -1;
        held_in_ring = false;
    }
    void release_moving_operand() override {
        held_ptr = nullptr;
        held_slot = -1;
        held_in_ring = false;
    }
    bool force_windows = false, force_one_chunk_per_pass = false;
    void walk_windows_when_held(int mode) override {
        force_windows = mode != 0;
        force_one_chunk_per_pass = mode == 2;
    }

    Sparse15D_Dense_Shift(SpmatLocal* S_input, int R, int c, int fusionApproach, KernelImplementation* k) : Distributed_Sparse(k) {
        this->fusionApproach = fusionApproach;
        this->c = c;
        ring_mode = kMeshFetch;
        if (const char* m = std::getenv("HNH_RING_MODE")) {
            if (std::string(m) == "relay") ring_mode = kRelay;
            else if (std::string(m) != "mesh") hnh::fatal("Error, HNH_RING_MODE must be relay or mesh!");
        }
        if (const char* h = std::getenv("HNH_ACC_HALVES")) acc_halves = std::atoi(h) != 0;
        if (const char* mw = std::getenv("HNH_WINDOW_MERGE")) merge_windows = std::atoi(mw) != 0;
        if (const char* mc = std::getenv("HNH_WINDOW_MERGE_CAP")) merge_cap = std::max(1, std::atoi(mc));
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

        merged = (fusionApproach == 2 && ring_mode == kMeshFetch && p / c > 1);
        const bool f1_mesh = std::getenv("HNH_FUSION1_MESH") == nullptr || std::atoi(std::getenv("HNH_FUSION1_MESH")) != 0;
        row_merged = (fusionApproach == 1 && ring_mode == kMeshFetch && p / c > 1 && f1_mesh && k->handles_row_ranges());
        const bool chunked = merged || row_merged;
        // default: six chunks of heights (1, 2, 2, 2, 1, 1) / 9 — against paced transfers of 40 .. 100 GB/s per link it beats the
        // symmetric four (1, 2, 2, 1) / 6 of round 2 by 1 .. 6 % (a smaller last chunk: a fetch-bound call ends one last-chunk kernel
        // after the fetch) and loses 4 % to it only when the links are so fast that the kernels bind
        windows = 1;
        if (chunked) {
            // (row-merged: three chunks (1, 2, 1) / 4 — a staging pass and its group of transfers per chunk; measured against six on 8
            // logical ranks and paced links: 58 vs 75 ms per pair at 60 GB/s, 60 vs 75 at 100, and 4.95 vs 5.14 ms for one rank alone,
            // profiles/r06_job3_overlap_accumulator_p8_chunk_shapes.log, r06_job2_fusion1_rank_share.log)
            if (row_merged) taper = {1, 2, 1};
            else taper = {1, 2, 2, 2, 1, 1};
            windows = (int)taper.size();
        }
        if (const char* q = std::getenv("HNH_MESH_CHUNKS")) {  // Q chunks of the symmetric shape (1, 2, .., 2, 1)
            const int v = std::atoi(q);
            if (v < 1 || v > 8) hnh::fatal("Error, HNH_MESH_CHUNKS must be between 1 and 8!");
            if (chunked) {
                windows = v;
                taper.clear();
            }
        }
        // HNH_MESH_TAPER=w0,w1,..: chunk heights proportional to these weights instead of the symmetric default — e.g. 6,5,4,3,2,1
        // ends on a small chunk (a fetch-bound call ends one LAST-chunk kernel after the fetch); overrides HNH_MESH_CHUNKS
        if (const char* t = std::getenv("HNH_MESH_TAPER")) {
            std::vector<int> w;
            for (const char* p2 = t; *p2;) {
                char* e2 = nullptr;
                const long v = std::strtol(p2, &e2, 10);
                if (e2 == p2 || v < 1 || v > 64) hnh::fatal("Error, HNH_MESH_TAPER must be a comma list of weights between 1 and 64!");
                w.push_back((int)v);
                p2 = (*e2 == ',') ? e2 + 1 : e2;
                if (*e2 != ',' && *e2 != 0) hnh::fatal("Error, HNH_MESH_TAPER must be a comma list of weights between 1 and 64!");
            }
            if (w.empty() || w.size() > 12) hnh::fatal("Error, HNH_MESH_TAPER takes 1 to 12 weights!");
            if (chunked) {
                taper = w;
                windows = (int)w.size();
            }
        }
        cutA = chunk_cuts(localArows, &fineA);
        cutB = chunk_cuts(localBrows, &fineB);

        const uint64_t arows = (uint64_t)localArows * c, brows = (uint64_t)localBrows * c;
        S->localize(arows, 0);
        ST->localize(brows, 0);
        if (chunked) {
            lay_out_merged(S.get(), localBrows, cutB, fineB);
            lay_out_merged(ST.get(), localArows, cutA, fineA);
        } else {
            S->divideIntoBlockCols(localBrows, p, true);
            ST->divideIntoBlockCols(localArows, p, true);
        }

        S->own_all_coordinates();
        ST->own_all_coordinates();

        const bool local_tpose = (fusionApproach == 1);
        if (chunked) {
            const int n = p / c;
            const std::vector<int64_t> wS = {localBrows, (int64_t)(n - 1) * localBrows}, wST = {localArows, (int64_t)(n - 1) * localArows};
            S->initializeCSRBlocks(localArows * c, 0, -1, local_tpose, &wS);
            ST->initializeCSRBlocks(localBrows * c, 0, -1, local_tpose, &wST);
            if (merged) {  // (row-merged: the chunks are row ranges of the transposed block, no window boundaries)
                set_chunk_windows(S.get(), cutB);
                set_chunk_windows(ST.get(), cutA);
            }
        } else {
            S->initializeCSRBlocks(localArows * c, localBrows, -1, local_tpose);
            ST->initializeCSRBlocks(localBrows * c, localArows, -1, local_tpose);
        }
        S->release_tuples();
        ST->release_tuples();
        check_initialized();
    }

    void setRValue(int R) override {
        if (R != this->R) {  // buffers of the old width are about to be replaced: nothing stays held
            held_slot = -1;
            held_in_ring = false;
        }
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
    // ---- merged layout helpers
    // visiting step of global block column b on this rank (block_at(k) == b), or -1 when the rank never visits it
    int step_of_block(int b) const {
        if (pMod(b - grid->rankInRow, c) != 0) return -1;
        return pMod(grid->rankInCol - (b - grid->rankInRow) / c, p / c);
    }
    // Chunk boundaries of a visiting block of `br` rows: chunk q is w_q fine chunks of *fine rows, w = the taper weights
    // (HNH_MESH_TAPER: 1 .. 64 each, up to 12 of them) or the symmetric default (1, 2, .., 2, 1); a single chunk when windows == 1.
    std::vector<int64_t> chunk_cuts(int br, int* fine) const {
        std::vector<int64_t> cut((size_t)windows + 1, br);
        cut[0] = 0;
        if (windows == 1) {
            *fine = std::max(br, 1);
            return cut;
        }
        // weights of the chunks in fine chunks: HNH_MESH_TAPER, or first and last = one, the others = two
        std::vector<int> w = taper;
        if (w.empty()) {
            w.assign((size_t)windows, 2);
            w.front() = w.back() = 1;
        }
        int total = 0;
        for (int x : w) total += x;
        *fine = std::max(1, divideAndRoundUp(br, total));
        int64_t prefix = 0;
        for (int q = 1; q < windows; q++) {
            prefix += w[(size_t)q - 1];
            cut[(size_t)q] = std::min<int64_t>(prefix * *fine, br);
        }
        return cut;
    }
    // first landing-buffer row of chunk q of the block fetched at step k >= 1
    int64_t landing_row(int k, int q, const std::vector<int64_t>& cut) const {
        return (int64_t)(p / c - 1) * cut[(size_t)q] + (int64_t)(k - 1) * (cut[(size_t)q + 1] - cut[(size_t)q]);
    }
    // Relabels the (row-localised, column-major) tuples' columns from global to "own block | landing-buffer row" and splits
    // them into the two blocks of the merged layout.
    void lay_out_merged(SpmatLocal* s, int br, const std::vector<int64_t>& cut, int fine) {
        const int n = p / c;
        const int nfine = divideAndRoundUp(br, fine);  // fine chunks that hold rows (<= the sum of the weights, at most 12 x 64)
        std::vector<int64_t> dest((size_t)p * nfine, -1);
        for (int b = 0; b < p; b++) {
            const int k = step_of_block(b);
            if (k < 0) continue;  // the distribution sends no nonzero of such a block column here
            for (int f = 0; f < nfine; f++) {
                const int64_t r0 = (int64_t)f * fine;
                int q = 0;
                while (q + 1 < windows && cut[(size_t)q + 1] <= r0) q++;  // the chunk this fine chunk belongs to
                dest[(size_t)b * nfine + f] = (k == 0) ? r0 : (int64_t)br + landing_row(k, q, cut) + (r0 - cut[(size_t)q]);
            }
        }
        s->remapColumns(br, fine, nfine, dest);
        s->sortColumnMajor((uint64_t)n * (uint64_t)br);
        s->divideIntoLocalAndRemote(br, n);
    }
    void set_chunk_windows(SpmatLocal* s, const std::vector<int64_t>& cut) {
        CSRLocal* remote = s->csr_blocks[1];
        if (remote == nullptr || windows == 1) return;
        std::vector<int32_t> bounds;
        for (int q = 1; q < windows; q++) bounds.push_back((int32_t)((int64_t)(p / c - 1) * cut[(size_t)q]));
        remote->set_windows(bounds);
    }

    // Issues the fetch of every remote block of a READ-ONLY moving operand into its landing buffer, chunk by chunk: group q
    // moves chunk q of all n-1 blocks (explicit-peer pairs over n-1 different links) and event(8 + q) is recorded behind
    // it.  Returns true when nothing had to move because a held operand's blocks are still there.
    bool fetch_into_landing(DenseMatrix* start, int slot, int br, const std::vector<int64_t>& cut) {
        const int n = p / c;
        if (ensure(landing[slot], (int64_t)(n - 1) * br, R) && held_slot == slot) held_slot = -1;  // a fresh buffer holds nobody's blocks
        auto t = phase_begin("Cyclic Shift Time");
        order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);  // my block is final; earlier readers of the landing buffer are done
        const bool held = (held_ptr == start->data());
        if (!held && held_slot == slot) held_slot = -1;  // another operand lands here: a held one has to be fetched again
        const bool resident = held && held_slot == slot;
#ifdef HNH_MEASUREMENT_AIDS
        // (libhnh_host_aids.so only, tools/overlap_probe.py) HNH_PACE_LINK_GBPS=<rate>: a held operand's chunks are already in the
        // landing buffer, but the communication stream is held for as long as each chunk would take to cross ONE xGMI link at that
        // rate (chunk q of the n-1 blocks travels over n-1 links at once), without copying anything: the event protocol is then
        // timed against transfers of a known duration with the rank's kernels alone on the GPU.  HNH_PACE_COPY=<workgroups per
        // link>: the stand-in also moves the bytes (chunk q of the OWN block into every peer's place in the landing buffer — only
        // meaningful when all blocks hold the same contents, i.e. on the single-process transports).
        const double pace_gbps = std::getenv("HNH_PACE_LINK_GBPS") ? std::atof(std::getenv("HNH_PACE_LINK_GBPS")) : 0.0;
        const int copy_wgs = std::getenv("HNH_PACE_COPY") ? std::atoi(std::getenv("HNH_PACE_COPY")) : 0;
        if (copy_wgs > 0 && std::string(world->kind()) != "thread-loopback" && std::string(world->kind()) != "single")
            hnh::fatal("Error, HNH_PACE_COPY is a single-process measurement aid!");
#endif
        for (int q = 0; q < windows; q++) {
            const int64_t r0 = cut[(size_t)q], w = cut[(size_t)q + 1] - r0;
            const size_t bytes = (size_t)w * (size_t)R * sizeof(double);
#ifdef HNH_MEASUREMENT_AIDS
            if (bytes > 0 && resident && pace_gbps > 0.0) {
                const double us = (double)bytes / (pace_gbps * 1e3);
                if (copy_wgs > 0)
                    world->check(world->be->hnh_stream_paced_copy(world->ctx, HNH_STREAM_COMM, landing[slot].data() + landing_row(1, q, cut) * R,
                                                                  start->data() + r0 * R, bytes, n - 1, us, copy_wgs),
                                 "hnh_stream_paced_copy");
                else
                    world->delay_us(us, HNH_STREAM_COMM);
            }
#endif
            if (bytes > 0 && !resident) {
                world->group_begin();
                for (int k = 1; k < n; k++)  // my block is what ring rank me+k needs at ITS step k; I need the block of me-k at mine
                    world->sendrecv(grid->col_world, start->data() + r0 * R, bytes, pMod(grid->rankInCol + k, n),
                                    landing[slot].data() + landing_row(k, q, cut) * R, bytes, pMod(grid->rankInCol - k, n), HNH_STREAM_COMM);
                world->group_end();
            }
            world->event_record(event(8 + q), HNH_STREAM_COMM);
        }
        if (held) held_slot = slot;
        phase_end(t);
        return resident;
    }

    // The kernels of one pass in the merged layout: `one(block, Y, window, is_last)` is called for the own block (while the
    // first chunk flies) and then for the fetched blocks — window by window as the chunks land, or in one piece when the
    // data is already there or the kernel plugin does not understand windows.
    template <typename One>
    void walk_merged(SpmatLocal* choice, DenseMatrix* Brole, One&& one) {
        const int slot = (choice == S.get()) ? 0 : 1;
        const int br = slot == 0 ? localBrows : localArows;
        const bool resident = fetch_into_landing(Brole, slot, br, slot == 0 ? cutB : cutA);
        auto t = phase_begin("Computation Time");
        CSRLocal* remote = choice->csr_blocks[1];
        one(0, *Brole, -1, -1, remote == nullptr);
        if (remote != nullptr) {
            // walk_windows_when_held(): walk the windows although a held operand's blocks are already there, which lets one rank's
            // kernel sequence be timed without its peers (bench.py's rank-share entries, tools/rank_share_probe.py)
            const bool by_window = (row_merged ? windows > 1 : (kernel->handles_windows() && remote->n_windows > 1)) && (!resident || force_windows);
            if (!by_window) {
                world->event_wait(event(8 + windows - 1), HNH_STREAM_COMPUTE);  // every chunk has landed
                one(1, landing[slot], -1, -1, true);
            } else if (!merge_windows || !(row_merged || kernel->handles_window_ranges()) || (resident && force_one_chunk_per_pass)) {
                for (int q = 0; q < windows; q++) {
                    world->event_wait(event(8 + q), HNH_STREAM_COMPUTE);  // chunk q of every remote block has landed
                    one(1, landing[slot], q, q + 1, q == windows - 1);
                }
            } else {
                // adaptive windows (see above): pass k covers the chunks [q, L) that have landed when the host gets to decide it
                world->event_record(event(24), HNH_STREAM_COMPUTE);  // pass 0 (the own block) is enqueued
                int k = 1;
                for (int q = 0; q < windows; k++) {
                    if (k >= 2) world->event_sync(event(24 + (k - 2) % 3));  // pass k - 2 is done: pass k - 1 runs, this one queues behind it
                    int L = q + 1;
                    while (L < windows && L - q < merge_cap && world->event_done(event(8 + L))) L++;
                    world->event_wait(event(8 + L - 1), HNH_STREAM_COMPUTE);  // chunks q .. L - 1 of every remote block have landed
                    one(1, landing[slot], q, L, L == windows);
                    world->event_record(event(24 + k % 3), HNH_STREAM_COMPUTE);
                    q = L;
                }
            }
        } else {
            world->event_wait(event(8 + windows - 1), HNH_STREAM_COMPUTE);  // keep the streams ordered for the next call
        }
        phase_end(t);
    }

    // One pass of shifts with the fused kernel on every visiting block.  target == nullptr: the result replaces
    // Xin (the reference's in-place fusedSpMM); otherwise it is written to *target and Xin survives.
    void fused_pass(DenseMatrix& Xin, DenseMatrix& moving, SpmatLocal* choice, DenseMatrix* target, unsigned act_flag,
                    const hnh_fused_extras* extras) {
        DenseMatrix* Arole = &Xin;
        DenseMatrix* Brole = &moving;
        const int n = p / c;
        const bool epilogue = KernelImplementation::wants_epilogue(extras);
        if (epilogue && target == nullptr) hnh::fatal("Error, a row epilogue needs the input rows: use fusedSpMM_out!");
        hnh_fused_extras act_only = {extras ? extras->leaky_alpha : 0.0, 0.0, nullptr, nullptr, nullptr, 0};
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
        bool out_fresh = true;
        // the fused kernel on one block (or one window of it); `ex` = what the call applies besides the multiplication
        auto fused_on = [&](int block_id, DenseMatrix& Y, int window, int window_end, const hnh_fused_extras* ex) {
            CSRLocal* blk = choice->csr_blocks[block_id];
            if (blk == nullptr && ex == act) return;  // nothing to multiply and no epilogue to run
            if (blk != nullptr) {
                blk->window = window;
                blk->window_end = window_end;
            }
            kernel->fused_local(*choice, *rowOperand, Y, *accum, block_id, base | (out_fresh ? HNH_FUSED_OUT_OVERWRITE : 0u), ex);
            if (blk != nullptr) blk->window = blk->window_end = -1;
            out_fresh = false;
        };

        if (merged) {
            walk_merged(choice, Brole, [&](int block_id, DenseMatrix& Y, int window, int window_end, bool is_last) {
                fused_on(block_id, Y, window, window_end, is_last ? last : act);
            });
        } else {
            ring_readonly(Brole, n, [&](int i, DenseMatrix& cur) {
                auto t = phase_begin("Computation Time");
                fused_on(block_at(i), cur, -1, -1, (i == n - 1) ? last : act);
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

        // SDDMM: every block (window) is visited exactly once per call, so a kernel that honours CSRLocal::values_fresh stores its
        // results and the zero fill of the reference (distributed_sparse.h:280) is not needed; other plugins get zeroed values
        const bool fresh = is_sddmm && kernel->overwrites_fresh_values();
        // ... and a kernel that borrows_value_arrays() reads SValues in place (SpMM: no setCSRValues copy) and writes SValues .* dots
        // straight into the result (SDDMM: no closing Hadamard pass): the blocks of this schedule never move
        const bool borrow = kernel->borrows_value_arrays();
        {
            auto t = phase_begin("Computation Time");
            if (is_sddmm && !fresh) choice->setValuesConstant(0.0);
            else if (is_sddmm && borrow) choice->lendSddmmTargets(SValues, *sddmm_result_ptr, Arole->cols(), borrow_mode);
            else if (!is_sddmm && borrow) choice->lendCSRValues(SValues, Arole->cols(), borrow_mode);
            else if (!is_sddmm) choice->setCSRValues(SValues);
            phase_end(t);
        }

        KernelMode mode_temp = mode;
        if (fusionApproach == 1 && mode == k_spmmA) mode_temp = k_spmmB;
        if (fusionApproach == 2 && mode == k_spmmB) mode_temp = k_spmmA;
        DenseMatrix& stationary = (c > 1) ? accumulation_buffer : *Arole;

        // the moving operand is written only when it is the SpMM accumulator (approach 1)
        const bool moving_readonly = is_sddmm || fusionApproach == 2;
        auto step = [&](int i, DenseMatrix& cur) {
            auto t = phase_begin("Computation Time");
            CSRLocal* blk = choice->csr_blocks[block_at(i)];
            if (blk != nullptr) blk->values_fresh = fresh;
            kernel->triple_function(mode_temp, *choice, stationary, cur, block_at(i), 0);
            if (blk != nullptr) blk->values_fresh = false;
            phase_end(t);
        };
        if (merged) {
            // same kernels on the same nonzeros as the reference's block-by-block walk; the fetched blocks' share runs
            // window by window as the chunks land
            walk_merged(choice, Brole, [&](int block_id, DenseMatrix& Y, int window, int window_end, bool) {
                CSRLocal* blk = choice->csr_blocks[block_id];
                if (blk == nullptr) return;
                blk->window = window;
                blk->window_end = window_end;
                blk->values_fresh = fresh;
                kernel->triple_function(mode_temp, *choice, stationary, Y, block_id, 0);
                blk->values_fresh = false;
                blk->window = blk->window_end = -1;
            });
        } else if (row_merged && is_sddmm) {
            // row-merged layout: the chunks [window, window_end) of the fetched blocks are ONE row range of the transposed block
            const std::vector<int64_t>& cut = (choice == S.get()) ? cutB : cutA;
            walk_merged(choice, Brole, [&](int block_id, DenseMatrix& Y, int window, int window_end, bool) {
                CSRLocal* blk = choice->csr_blocks[block_id];
                if (blk == nullptr) return;
                if (window >= 0) blk->select_row_range((int64_t)(n - 1) * cut[(size_t)window], (int64_t)(n - 1) * cut[(size_t)window_end]);
                blk->values_fresh = fresh;
                kernel->triple_function(mode_temp, *choice, stationary, Y, block_id, 0);
                blk->values_fresh = false;
                blk->select_row_range(-1, -1);
            });
        } else if (row_merged) {
            spmm_mesh_reduce_scatter(choice, Brole, stationary, mode_temp);
        } else if (moving_readonly) {
            ring_readonly(Brole, n, step);
        } else if (n > 1 && acc_halves && kernel->handles_row_parts()) {
            ring_readwrite_halves(Brole, n, choice, step);
        } else {
            ring_readwrite(Brole, n, step);
        }

        if (is_sddmm) {
            auto t = phase_begin("Computation Time");
            choice->hadamardWithCSRValues(SValues, *sddmm_result_ptr);  // result = SValues .* block values (blocks that did not borrow)
            phase_end(t);
        }
        choice->reclaimValueArrays();

        if (fusionApproach == 2 && !is_sddmm && c > 1) {
            auto t = phase_begin("Replication Time");
            world->reduce_scatter_f64(grid->row_world, accumulation_buffer.data(), Arole->data(), (size_t)Arole->rows() * R,
                                      HNH_STREAM_COMPUTE);
            phase_end(t);
        }
    }

private:
    static bool ensure(DenseMatrix& m, int64_t rows, int64_t cols) {  // true when the buffer was (re)allocated: its contents are gone
        if (m.rows() == rows && m.cols() == cols) return false;
        m = DenseMatrix(rows, cols);
        return true;
    }

    // SpMM of replication reuse as a MESH REDUCE-SCATTER (row-merged layout, see the class comment).  `home` is the caller's block of the
    // output (the accumulator's home: the SpMM adds to what is there, sparse_kernels.cpp:95-121 with beta = 1), `stationary` the gathered
    // operand.  Events: 28 + q  = the staging pass of chunk q is done (compute -> comm), 8 + q = chunk q of every partial block has
    // landed (comm -> compute; the numbers the fetch uses: a landing buffer is either a fetch's or an exchange's at any one time).
    void spmm_mesh_reduce_scatter(SpmatLocal* choice, DenseMatrix* home, DenseMatrix& stationary, KernelMode mode_temp) {
        const int n = p / c;
        const int slot = (choice == S.get()) ? 0 : 1;
        const int br = slot == 0 ? localBrows : localArows;
        const std::vector<int64_t>& cut = slot == 0 ? cutB : cutA;
        ensure(staging[slot], (int64_t)(n - 1) * br, R);
        ensure(landing[slot], (int64_t)(n - 1) * br, R);
        if (held_slot == slot) held_slot = -1;  // the landing buffer is about to hold partial results, not a held operand's blocks
        CSRLocal* remote = choice->csr_blocks[1];
        CSRLocal* own = choice->csr_blocks[0];
        const bool store = kernel->stores_fresh_output();
        {
            // the staging buffer's last readers (the previous call's sends) and the landing buffer's (its adds) are behind these
            auto t = phase_begin("Cyclic Shift Time");
            order(HNH_STREAM_COMM, HNH_STREAM_COMPUTE, 5);
            order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);
            phase_end(t);
        }
        if (remote == nullptr || !store) {
            auto t = phase_begin("Computation Time");
            staging[slot].setZero();  // no nonzero in any other rank's block (the partial results are zero), or a kernel that adds
            phase_end(t);
        }
        for (int q = 0; q < windows; q++) {
            const int64_t r0 = (int64_t)(n - 1) * cut[(size_t)q], r1 = (int64_t)(n - 1) * cut[(size_t)q + 1];
            if (remote != nullptr && r1 > r0) {
                auto t = phase_begin("Computation Time");
                remote->select_row_range(r0, r1);
                remote->out_fresh = store;
                kernel->triple_function(mode_temp, *choice, stationary, staging[slot], 1, 0);
                remote->out_fresh = false;
                remote->select_row_range(-1, -1);
                phase_end(t);
            }
            auto t = phase_begin("Cyclic Shift Time");
            world->event_record(event(28 + q), HNH_STREAM_COMPUTE);
            world->event_wait(event(28 + q), HNH_STREAM_COMM);
            const int64_t w = cut[(size_t)q + 1] - cut[(size_t)q];
            const size_t bytes = (size_t)w * (size_t)R * sizeof(double);
            if (bytes > 0) {
                world->group_begin();
                for (int k = 1; k < n; k++)  // my partial result for the block I visit at step k goes to its owner, ring rank me - k; mine arrive from me + k
                    world->sendrecv(grid->col_world, staging[slot].data() + landing_row(k, q, cut) * R, bytes, pMod(grid->rankInCol - k, n),
                                    landing[slot].data() + landing_row(k, q, cut) * R, bytes, pMod(grid->rankInCol + k, n), HNH_STREAM_COMM);
                world->group_end();
            }
            world->event_record(event(8 + q), HNH_STREAM_COMM);
            phase_end(t);
        }
        {
            auto t = phase_begin("Computation Time");  // the own block's share goes straight into the caller's rows, under the last transfers
            if (own != nullptr) kernel->triple_function(mode_temp, *choice, stationary, *home, 0, 0);
            phase_end(t);
        }
        // the n-1 partial blocks are added in ring-step order, chunk by chunk as they land; consecutive chunks that have landed together
        // go in one launch (the host decides as late as it can, like the windowed passes of the fetch)
        auto t = phase_begin("Computation Time");
        for (int q = 0; q < windows;) {
            int L = q + 1;
            if (merge_windows)
                while (L < windows && L - q < merge_cap && world->event_done(event(8 + L))) L++;
            world->event_wait(event(8 + L - 1), HNH_STREAM_COMPUTE);
            world->check(world->be->hnh_sum_chunked_blocks_f64(world->ctx, home->data(), landing[slot].data(), n - 1, windows, cut.data(), q, L, R,
                                                               HNH_STREAM_COMPUTE),
                         "hnh_sum_chunked_blocks_f64");
            q = L;
        }
        phase_end(t);
    }

    // n kernel steps over a READ-ONLY moving operand on the neighbour ring: n-1 overlapped shifts, caller's buffer untouched.
    template <typename Step>
    void ring_readonly(DenseMatrix* start, int n, Step&& step) {
        if (n > 1) {
            for (auto& s : ring_spare)
                if (ensure(s, start->rows(), start->cols())) held_in_ring = false;  // a fresh spare holds nobody's block
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

    // The same ring with the accumulator cut into two row halves H0, H1 (CSRLocal::row_part: the visiting block's SpMM runs per half):
    //     compute:  K(i,H0)  K(i,H1)            K(i+1,H0)  K(i+1,H1) ...
    //     comm:              S(i,H0)  S(i,H1)              S(i+1,H0) ...
    // S(i,H0) waits for K(i,H0) only and runs under K(i,H1); K(i+1,H0) waits for S(i,H0) only and runs under S(i,H1): per step
    // max(kernel, shift) instead of their sum (the reference's kernel -> MPI_Sendrecv -> barrier, 15D_dense_shift.hpp:343-356).
    // Same rows, same arithmetic per row, same n shifts; a half is sent only after its kernel and overwritten only after its send.
    template <typename Step>
    void ring_readwrite_halves(DenseMatrix* start, int n, SpmatLocal* choice, Step&& step) {
        hnh::BufferPair bBuf(start, &ring_spare[0]);
        held_in_ring = false;  // ring_spare[0] is overwritten
        const int dst = pMod(grid->rankInCol + 1, n), src = pMod(grid->rankInCol - 1, n);
        const int64_t h = start->rows() / 2, cols = start->cols();
        const size_t bytes0 = (size_t)h * (size_t)cols * sizeof(double), bytes1 = (size_t)(start->rows() - h) * (size_t)cols * sizeof(double);
        order(HNH_STREAM_COMPUTE, HNH_STREAM_COMM, 0);  // earlier users of the spare are done
        for (int i = 0; i < n; i++) {
            CSRLocal* blk = choice->csr_blocks[block_at(i)];
            DenseMatrix* act = bBuf.getActive();
            DenseMatrix* pas = bBuf.getPassive();
            for (int part = 0; part < 2; part++) {
                if (i > 0) world->event_wait(event(12 + part), HNH_STREAM_COMPUTE);  // this half of the arriving accumulator has landed
                if (blk != nullptr) blk->select_row_part(part);
                step(i, *act);
                if (blk != nullptr) blk->select_row_part(-1);
                world->event_record(event(10 + part), HNH_STREAM_COMPUTE);
            }
            auto t = phase_begin("Cyclic Shift Time");
            for (int part = 0; part < 2; part++) {
                world->event_wait(event(10 + part), HNH_STREAM_COMM);
                const size_t off = part == 0 ? 0 : (size_t)h * (size_t)cols, bytes = part == 0 ? bytes0 : bytes1;
                world->sendrecv(grid->col_world, act->data() + off, bytes, dst, pas->data() + off, bytes, src, HNH_STREAM_COMM);
                world->event_record(event(12 + part), HNH_STREAM_COMM);
            }
            bBuf.swapActive();
            phase_end(t);
        }
        world->event_wait(event(12), HNH_STREAM_COMPUTE);
        world->event_wait(event(13), HNH_STREAM_COMPUTE);
        auto t = phase_begin("Computation Time");
        bBuf.sync_active();
        phase_end(t);
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
