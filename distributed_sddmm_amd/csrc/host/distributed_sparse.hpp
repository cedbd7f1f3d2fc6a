// Operator base class — same public surface as the reference's distributed_sparse.h:32-388:
//   fields  proc_rank, p, c, algorithm_name, proc_grid_names, perf_counter_keys, call_count, total_time,
//           M, N, R, localA{rows,cols}, localB{rows,cols}, aSubmatrices, bSubmatrices, S, ST, grid, kernel,
//           r_split, A_R_split_world, B_R_split_world, verbose
//   methods setRValue, like_{A,B}_matrix, like_{S,ST}_values, initial_shift, de_shift, sddmmA/B, spmmA/B,
//           fusedSpMM, algorithm, reset_performance_timers, stop_clock_and_add, json_perf_statistics,
//           json_algorithm_info, print_*, dummyInitialize, shiftDenseMatrix, check_initialized.
// Global meaning (row a9 of SURVEY §8): sddmm[e] = Sval[e] * <A[i_e,:], B[j_e,:]>;  spmmA: A = S B;
// spmmB: B = S^T A;  fusedSpMM(Amat): A <- (Sval .* (A B^T)|_S) B.
//
// What differs, deliberately:
//   * dense operands and value vectors are device resident (dense.hpp); MPI communicators are hnh::Comm;
//   * the per-step MPI_Barrier(MPI_COMM_WORLD) (e.g. 15D_dense_shift.hpp:355) is gone: ordering between the
//     local kernel (compute stream) and the ring shift (communication stream) is expressed with HIP events,
//     which is what lets the two overlap;
//   * every buffer the reference allocates per call (BufferPair::extra, accumulation_buffer, getCSRValues
//     temporaries) is allocated once and reused;
//   * JSON is produced as text (same keys as the reference's nlohmann objects).
#pragma once
#include "json.hpp"
#include <algorithm>
#include <cassert>
#include <iomanip>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "common.hpp"
#include "dense.hpp"
#include "flexible_grid.hpp"
#include "sparse_kernels.hpp"
#include "spmat_local.hpp"

class DenseSubmatrix {
public:
    int topRow, leftCol, rowCount, colCount;
    DenseSubmatrix(int tR, int lC, int rC, int cC) : topRow(tR), leftCol(lC), rowCount(rC), colCount(cC) {}
};

class Distributed_Sparse {
public:
    int proc_rank;  // global process rank
    int p, c;       // total # of processes, replication factor

    std::string algorithm_name;
    std::vector<std::string> proc_grid_names;

    // performance counting (same key names as the reference so result JSON stays drop-in)
    std::vector<std::string> perf_counter_keys;
    std::map<std::string, int> call_count;
    std::map<std::string, double> total_time;

    int64_t M, N, R;
    int localArows, localAcols, localBrows, localBcols;

    std::vector<DenseSubmatrix> aSubmatrices;
    std::vector<DenseSubmatrix> bSubmatrices;

    std::unique_ptr<SpmatLocal> S;
    std::unique_ptr<SpmatLocal> ST;
    std::shared_ptr<FlexibleGrid> grid;

    int superclass_constructor_sentinel;
    KernelImplementation* kernel;  // stored, not owned (distributed_sparse.h:65,84)

    bool r_split;
    hnh::Comm A_R_split_world, B_R_split_world;
    bool verbose;
    std::string debug_msg;
    hnh::World* world;
    // Borrowed value arrays (SpmatLocal::lendCSRValues / lendSddmmTargets) for stationary blocks, read at construction from
    // HNH_BORROW: unset = where it pays (SpmatLocal::lendable), "off" = the reference's copy / Hadamard passes, "force" = always
    int borrow_mode = 0;

    Distributed_Sparse(KernelImplementation* k) {
        if (const char* b = std::getenv("HNH_BORROW")) borrow_mode = std::string(b) == "off" ? -1 : (std::string(b) == "force" ? 1 : 0);
        world = hnh::current_world();
        proc_rank = world->rank;
        p = world->size;
        verbose = false;
        kernel = k;
        algorithm_name = "";
        M = N = R = -1;
        localArows = localAcols = localBrows = localBcols = -1;
        c = -1;
        r_split = false;
        superclass_constructor_sentinel = 3;
    }

    virtual ~Distributed_Sparse() {
        if (world) {
            world->sync_all_nothrow();  // a destructor must not throw (the C ABI runs with throw-on-error)
            for (void* e : events_) world->be->hnh_event_destroy(world->ctx, e);
            for (void* e : spare_events_) world->be->hnh_event_destroy(world->ctx, e);
            for (auto& sp : spans_) {
                if (sp.own0) world->be->hnh_event_destroy(world->ctx, sp.e0);
                if (sp.own1) world->be->hnh_event_destroy(world->ctx, sp.e1);
            }
        }
    }

    virtual void setRValue(int R) = 0;

    void check_initialized() {
        bool ok = algorithm_name != "" && proc_grid_names.size() > 0 && perf_counter_keys.size() > 0 && M != -1 && N != -1 &&
                  R != -1 && localAcols != -1 && localBcols != -1 && localArows != -1 && localBrows != -1 && c >= 1 &&
                  superclass_constructor_sentinel == 3 && aSubmatrices.size() > 0 && bSubmatrices.size() > 0 && S &&
                  ST && S->initialized && ST->initialized && S->coordinate_ownership_initialized &&
                  ST->coordinate_ownership_initialized && S->blockStarts.size() > 0 && ST->blockStarts.size() > 0 &&
                  S->csr_initialized && ST->csr_initialized;
        if (!ok) hnh::fatal("Error, distributed sparse operator was not completely initialized by its subclass!");
    }

    // ---- reporting (distributed_sparse.h:131-179, 245-261)
    hnh::json json_algorithm_info() {
        std::vector<uint64_t> mine = {(uint64_t)(S->owned_coords_end - S->owned_coords_start),
                                      (uint64_t)(ST->owned_coords_end - ST->owned_coords_start)};
        std::vector<uint64_t> all(2 * (size_t)p);
        world->host_allgather(mine.data(), all.data(), 2 * sizeof(uint64_t));
        hnh::json jobj;  // the reference's keys, in its order (distributed_sparse.h:132-141)
        jobj["alg_name"] = algorithm_name;
        jobj["m"] = (uint64_t)M;
        jobj["n"] = (uint64_t)N;
        jobj["nnz"] = (uint64_t)S->dist_nnz;
        jobj["r"] = R;
        jobj["adjacency_mode"] = grid->adjacency;
        jobj["p"] = p;
        jobj["c"] = c;
        hnh::json dim_interpretations = hnh::json::array(), dim_values = hnh::json::array();
        for (size_t i = 0; i < proc_grid_names.size(); i++) {
            dim_interpretations.push_back(proc_grid_names[i]);
            dim_values.push_back(grid->dim_list[i]);
        }
        jobj["dim_interpretations"] = dim_interpretations;
        jobj["dim_values"] = dim_values;
        hnh::json nnz = hnh::json::array(), nnz_tpose = hnh::json::array();
        for (int r = 0; r < p; r++) {
            nnz.push_back(all[2 * (size_t)r]);
            nnz_tpose.push_back(all[2 * (size_t)r + 1]);
        }
        jobj["nnz_procs"] = nnz;
        jobj["nnz_tpose_procs"] = nnz_tpose;
        jobj["transport"] = world->kind();  // (two keys the reference does not have)
        jobj["backend"] = world->be->name;
        return jobj;
    }
    void print_algorithm_info() { std::cout << json_algorithm_info().dump(4) << std::endl; }
    void setVerbose(bool value) { verbose = value; }

    virtual VectorXd like_S_values(double value) { return VectorXd::Constant(S->owned_coords_end - S->owned_coords_start, value); }
    virtual VectorXd like_ST_values(double value) { return VectorXd::Constant(ST->owned_coords_end - ST->owned_coords_start, value); }
    DenseMatrix like_A_matrix(double value) { return DenseMatrix::Constant(localArows, localAcols, value); }
    DenseMatrix like_B_matrix(double value) { return DenseMatrix::Constant(localBrows, localBcols, value); }

    void reset_performance_timers() {
        resolve_spans(spans_.size());
        for (auto& key : perf_counter_keys) {
            call_count[key] = 0;
            total_time[key] = 0.0;
        }
    }

    // The reference's wall-clock counter (distributed_sparse.h:212-223), kept for callers that time host-side work.
    void stop_clock_and_add(my_timer_t& start, const std::string& counter_name) {
        if (std::find(perf_counter_keys.begin(), perf_counter_keys.end(), counter_name) == perf_counter_keys.end())
            hnh::fatal("Error, performance counter " + counter_name + " not registered.");
        if (world->timing_sync) world->sync_all();
        call_count[counter_name]++;
        total_time[counter_name] += stop_clock_get_elapsed(start);
    }

    // Device-timed phases.  The schedules' work is asynchronous — a host clock around an enqueue sees microseconds — so a
    // phase is bracketed by a pair of HIP events on the stream it runs on ("... Shift Time" of the ring loops: the
    // communication stream; everything else, incl. the collectives of the replication phases: the compute stream).  The pairs
    // are resolved (elapsed time added to the reference's counter of the same name) when the statistics are read, without
    // draining a stream in the middle of a call, so the compute/communication overlap being measured is not disturbed.
    // A span covers what the stream did between its two events, waits on the other stream included — the device-side
    // analogue of the reference's counters, whose "Cyclic Shift Time" also contains the wait for the neighbour.
    struct PhaseClock {
        void* e0 = nullptr;
        int stream = HNH_STREAM_COMPUTE;
        int key = -1;
        bool off = false;
        bool own0 = true;  // the span recycles its start event (false: it belongs to the span that ended there)
        my_timer_t host;
    };
    // HNH_PERF_COUNTERS=0: the phases are counted but not timed (their event pairs are two thirds of the host's work in a call of
    // config 1's size); the three counters then read 0
    static bool perf_counters_on() {
        static const bool on = [] {
            const char* v = std::getenv("HNH_PERF_COUNTERS");
            return v == nullptr || std::atoi(v) != 0;
        }();
        return on;
    }
    static int phase_stream(const std::string& key) {
        return (key.find("Cyclic Shift") != std::string::npos) ? HNH_STREAM_COMM : HNH_STREAM_COMPUTE;
    }
    PhaseClock phase_begin(const char* counter_name) {
        PhaseClock t;
        auto it = std::find(perf_counter_keys.begin(), perf_counter_keys.end(), counter_name);
        if (it == perf_counter_keys.end()) hnh::fatal(std::string("Error, performance counter ") + counter_name + " not registered.");
        t.key = (int)(it - perf_counter_keys.begin());
        if (!perf_counters_on()) {
            t.off = true;
            return t;
        }
        t.host = start_clock();
        if (world->timing_sync) return t;  // reference-like attribution: wall clock around drained streams
        t.stream = phase_stream(*it);
        t.e0 = take_event();
        world->event_record(t.e0, t.stream);
        return t;
    }
    void phase_end(PhaseClock& t) {
        const std::string& key = perf_counter_keys[(size_t)t.key];
        call_count[key]++;
        if (t.off) return;
        if (t.e0 == nullptr) {
            world->sync_all();
            total_time[key] += stop_clock_get_elapsed(t.host);
            return;
        }
        void* e1 = take_event();
        world->event_record(e1, t.stream);
        spans_.push_back({t.e0, e1, t.key, t.own0, true});
        t.e0 = nullptr;
        if (spans_.size() > 2048) resolve_spans(spans_.size() / 2);  // old spans finished long ago: no stall to speak of
    }
    // SHARED BOUNDARY EVENTS (round 6: calls of config 1's size are host bound, and 38 hipEventRecord of 2.5 us were half of a fused
    // call's host time).  A loop whose phases follow one another on a stream needs ONE event per boundary, not three: the end of a
    // phase IS the ordering event the other stream waits for, and IS the start of the next phase on that stream.
    //   phase_end_mark(t, slot)   ends the phase like phase_end() and returns the event recorded at its end, for the caller to use as
    //                             its ordering event (timed: the span's end event; counters off or reference-like attribution: the
    //                             pool's event(slot), recorded here) — valid for waits enqueued before the caller's next end_mark with
    //                             the same slot;
    //   phase_begin_at(name, e)   begins a phase at `e`, an event the caller has just had recorded on the phase's stream with nothing
    //                             enqueued on that stream since (nullptr: phase_begin).
    void* phase_end_mark(PhaseClock& t, size_t slot) {
        const std::string& key = perf_counter_keys[(size_t)t.key];
        const int stream = phase_stream(key);
        if (t.off || t.e0 == nullptr) {
            phase_end(t);
            void* e = event(slot);
            world->event_record(e, stream);
            return e;
        }
        phase_end(t);
        return spans_.back().e1;
    }
    PhaseClock phase_begin_at(const char* counter_name, void* at) {
        if (at == nullptr || !perf_counters_on() || world->timing_sync) return phase_begin(counter_name);
        PhaseClock t;
        auto it = std::find(perf_counter_keys.begin(), perf_counter_keys.end(), counter_name);
        if (it == perf_counter_keys.end()) hnh::fatal(std::string("Error, performance counter ") + counter_name + " not registered.");
        t.key = (int)(it - perf_counter_keys.begin());
        t.host = start_clock();
        t.stream = phase_stream(*it);
        t.e0 = at;
        t.own0 = false;
        // the span that ended at `at` hands the event's recycling over to this one (it is resolved first)
        for (size_t k = spans_.size(); k-- > 0 && k + 8 > spans_.size();)
            if (spans_[k].e1 == at && spans_[k].own1) {
                spans_[k].own1 = false;
                t.own0 = true;
                break;
            }
        if (!t.own0) return phase_begin(counter_name);  // (not an event of a live span: nothing to share)
        return t;
    }

    hnh::json json_perf_statistics() {  // mean over ranks, as distributed_sparse.h:245-261
        resolve_spans(spans_.size());
        std::vector<double> vals;
        for (auto& key : perf_counter_keys) vals.push_back(total_time[key]);
        world->host_allreduce_sum(vals.data(), vals.size());
        hnh::json j_obj = hnh::json::object();
        for (size_t i = 0; i < perf_counter_keys.size(); i++) j_obj[perf_counter_keys[i]] = vals[i] / p;
        return j_obj;
    }

    void print_performance_statistics() {
        if (proc_rank == 0) {
            std::cout << std::endl << "================================" << std::endl << "==== Performance Statistics ====" << std::endl
                      << "================================" << std::endl;
        }
        hnh::json info = json_algorithm_info(), stats = json_perf_statistics();
        if (proc_rank == 0) std::cout << info.dump(4) << std::endl << stats.dump(4) << std::endl << "=================================" << std::endl;
    }

    // If the input buffers need to be shifted / transposed
    virtual void initial_shift(DenseMatrix* localA, DenseMatrix* localB, KernelMode op) = 0;
    virtual void de_shift(DenseMatrix* localA, DenseMatrix* localB, KernelMode op) = 0;

    // ---- the five convenience operations (distributed_sparse.h:274-312)
    void spmmA(DenseMatrix& localA, DenseMatrix& localB, VectorXd& SValues) {
        prepare_spmm_output(localA);
        algorithm(localA, localB, SValues, nullptr, k_spmmA, true);
    }
    void spmmB(DenseMatrix& localA, DenseMatrix& localB, VectorXd& SValues) {
        prepare_spmm_output(localB);
        algorithm(localA, localB, SValues, nullptr, k_spmmB, true);
    }
    void sddmmA(DenseMatrix& localA, DenseMatrix& localB, VectorXd& SValues, VectorXd& sddmm_result) {
        algorithm(localA, localB, SValues, &sddmm_result, k_sddmmA, true);
    }
    void sddmmB(DenseMatrix& localA, DenseMatrix& localB, VectorXd& SValues, VectorXd& sddmm_result) {
        algorithm(localA, localB, SValues, &sddmm_result, k_sddmmB, true);
    }

    // FusedMM by replication reuse: the SpMM skips the replication the SDDMM already did.  The user is
    // responsible for any initial and final shifts (fairness in benchmarking).
    virtual void fusedSpMM(DenseMatrix& localA, DenseMatrix& localB, VectorXd& Svalues, VectorXd& sddmm_buffer, MatMode mode) {
        if (mode == Amat) {
            algorithm(localA, localB, Svalues, &sddmm_buffer, k_sddmmA, true);
            prepare_spmm_output(localA);
            algorithm(localA, localB, sddmm_buffer, nullptr, k_spmmA, false);
        } else if (mode == Bmat) {
            algorithm(localA, localB, Svalues, &sddmm_buffer, k_sddmmB, true);
            prepare_spmm_output(localB);
            algorithm(localA, localB, sddmm_buffer, nullptr, k_spmmB, false);
        }
    }

    // Hint (an addition): between hold_moving_operand(&m) and release_moving_operand() the CONTENTS of dense matrix m do
    // not change (the fixed factor of an ALS half-step: 1 + cg_max_iter fused calls, als_conjugate_gradients.cpp:38-141).
    // A schedule that fetches m's blocks from other ranks on every call may then keep what it fetched.  Default: ignored.
    virtual void hold_moving_operand(const DenseMatrix* m) { (void)m; }
    virtual void release_moving_operand() {}
    // Measurement entry point (an addition): a held operand's fetched blocks are resident, so a call normally runs ONE pass over them.
    // mode 1: the call walks the chunk windows as a fetching call does — own block, then windowed passes over whatever has "landed"
    // (everything, here: the adaptive windows take it in one pass); mode 2: one windowed pass per chunk, the sequence of a call whose
    // chunks arrive one by one — which is what lets ONE rank's kernel sequence of a p-rank job be timed alone on a GPU (bench.py's
    // "rank share" entries, tools/rank_share_probe.py).  Results are identical either way.  Default: ignored.
    virtual void walk_windows_when_held(int mode) { (void)mode; }

    // Out-of-place fusedSpMM with the applications' surrounding work folded in (an addition; hnh_fused_extras in
    // hnh_kernels.h).  With X = localA for Amat (localB for Bmat) and Y the other operand:
    //     w_e        = <X[i_e,:], Y[j_e,:]>,  LeakyReLU'd when `leaky`                (gat.hpp:96-99)
    //     Out[i,:]   = sum_e w_e Y[j_e,:]  +  extras.x_scale * X[i,:]                 (als_conjugate_gradients.cpp:282,295)
    //     rowdot[i]  = <X[i,:], Out[i,:]>   when extras.rowdot != nullptr             (als_conjugate_gradients.cpp:93)
    // X and Y are left untouched (the in-place fusedSpMM forces callers to copy X first).  Svalues are taken as 1
    // and sddmm_buffer is not filled, like the local-kernel-fusion fusedSpMM this extends (15D_dense_shift.hpp:189).
    // Returns false — having done nothing — when the schedule has no such single pass; callers then compose the
    // public calls as the reference does.
    virtual bool fusedSpMM_out(DenseMatrix& localA, DenseMatrix& localB, MatMode mode, DenseMatrix& Out, bool leaky,
                               const hnh_fused_extras& extras) {
        (void)localA; (void)localB; (void)mode; (void)Out; (void)leaky; (void)extras;
        return false;
    }

    virtual void algorithm(DenseMatrix& localA, DenseMatrix& localB, VectorXd& SValues, VectorXd* sddmm_result_ptr, KernelMode mode,
                           bool initial_replicate) = 0;

    // Deterministic test fill: value(row, col) = row * R + col in GLOBAL coordinates, so fingerprints do
    // not depend on the distribution (distributed_sparse.h:322-346).
    void dummyInitialize(DenseMatrix& loc, MatMode mode) {
        std::vector<DenseSubmatrix>& subs = (mode == Amat) ? aSubmatrices : bSubmatrices;
        std::vector<double> host((size_t)loc.size());
        double* ptr = host.data();
        for (auto& s : subs)
            for (int i = 0; i < s.rowCount; i++)
                for (int j = 0; j < s.colCount; j++) *ptr++ = (double)(s.topRow + i) * (double)R + s.leftCol + j;
        loc.copy_from_host(host.data());
    }

    // ---- dense cyclic shift (distributed_sparse.h:351-361): active buffer -> `send_dst`, the buffer of
    // `recv_src` -> passive buffer, swap.  Explicit source (the reference receives from MPI_ANY_SOURCE and
    // races).  Stream-ordered on `stream`.
    void shiftDenseMatrix(hnh::BufferPair& buf, const hnh::Comm& comm, int send_dst, int recv_src, int stream = HNH_STREAM_COMM) {
        const size_t bytes = (size_t)buf.getActive()->size() * sizeof(double);
        world->sendrecv(comm, buf.getActive()->data(), bytes, send_dst, buf.getPassive()->data(), bytes, recv_src, stream);
        buf.swapActive();
    }

    // The reference's four-argument form (distributed_sparse.h:351): every caller shifts by the same distance on
    // every rank, so the source is the rank symmetric to `send_dst`; `tag` is unused (no wildcard matching here).
    void shiftDenseMatrix(hnh::BufferPair& buf, const hnh::Comm& comm, int send_dst, int tag) {
        (void)tag;
        shiftDenseMatrix(buf, comm, send_dst, pMod(2 * comm.me - send_dst, comm.size()), HNH_STREAM_COMPUTE);
    }

    // Rank-by-rank dump of grid position, local tuples (while they still exist) and the local dense operands
    // (distributed_sparse.h:363-387); debugging aid, downloads the matrices.
    void print_nonzero_distribution(DenseMatrix& localA, DenseMatrix& localB) {
        for (int i = 0; i < p; i++) {
            if (proc_rank == i) {
                std::cout << "==================================" << std::endl << "Process " << i << ":" << std::endl
                          << "Rank in Row: " << grid->rankInRow << std::endl << "Rank in Column: " << grid->rankInCol << std::endl
                          << "Rank in Fiber: " << grid->rankInFiber << std::endl;
                for (auto& t : S->coords) std::cout << t.string_rep() << std::endl;
                for (int which = 0; which < 2; which++) {
                    DenseMatrix& m = which == 0 ? localA : localB;
                    std::cout << "==================" << std::endl << (which == 0 ? "A matrix: " : "B matrix: ") << std::endl;
                    std::vector<double> h = m.to_host();
                    for (int64_t r = 0; r < m.rows(); r++) {
                        for (int64_t c2 = 0; c2 < m.cols(); c2++) std::cout << h[(size_t)(r * m.cols() + c2)] << (c2 + 1 < m.cols() ? " " : "");
                        std::cout << std::endl;
                    }
                }
                std::cout << "==================================" << std::endl;
            }
            world->barrier();
        }
    }

protected:
    // Travelling blocks: every rank of the ring must know the longest row of ANY block that will visit it
    // (the kernels' long-row hint); one int all-gathered over the ring at construction.
    void publish_ring_max_row(SpmatLocal* s, const hnh::Comm& ring) {
        CSRLocal* blk = s->csr_blocks[0];
        int mine = blk ? blk->max_row_nnz : 0;
        std::vector<int> all(ring.size(), 0);
        world->host_allgather_comm(ring, &mine, all.data(), sizeof(int));
        if (blk) blk->ring_max_row_nnz = *std::max_element(all.begin(), all.end());
    }

    // The output of an SpMM.  spmmA / spmmB / the generic fusedSpMM zero it before they call algorithm() (distributed_sparse.h:284,293,303,
    // 307), and some schedules zero it again at their start (15D_sparse_shift.hpp:216 `tmp *= 0`).  Here the wrapper ASKS the schedule:
    //   spmm_stores_output() (a schedule whose SpMM stores every output row on first touch — its kernel says stores_fresh_output()):
    //       nothing is filled; algorithm() is told that the matrix holds nothing yet (take_output_unset) and either stores, or — on a
    //       path that cannot — zeroes it itself;
    //   otherwise: the wrapper zeroes it and says so (take_output_zeroed), so that the schedule does not fill it a second time.
    // One or two memsets per call, which a call of config 1's size notices (and 0.1 ms of config 4's rank).  algorithm() called directly:
    // neither is set and everything is as in the reference.
    virtual bool spmm_stores_output() const { return false; }
    const double* output_zeroed_ = nullptr;
    const double* output_unset_ = nullptr;
    void prepare_spmm_output(DenseMatrix& out) {
        output_zeroed_ = output_unset_ = nullptr;
        if (spmm_stores_output()) {
            output_unset_ = out.data();
        } else {
            out.setZero();
            output_zeroed_ = out.data();
        }
    }
    bool take_output_zeroed(const DenseMatrix& m) {
        const bool yes = output_zeroed_ != nullptr && output_zeroed_ == m.data();
        output_zeroed_ = nullptr;
        return yes;
    }
    bool take_output_unset(const DenseMatrix& m) {
        const bool yes = output_unset_ != nullptr && output_unset_ == m.data();
        output_unset_ = nullptr;
        return yes;
    }

    // device-timed phases waiting to be resolved, and recycled events
    struct Span {
        void* e0;
        void* e1;
        int key;
        bool own0, own1;  // which of the two this span hands back to the pool (a shared boundary event belongs to the later span)
    };
    std::vector<Span> spans_;
    std::vector<void*> spare_events_;
    void* take_event() {
        if (spare_events_.empty()) return world->event_create();
        void* e = spare_events_.back();
        spare_events_.pop_back();
        return e;
    }
    void resolve_spans(size_t count) {  // the oldest `count` spans: wait for their end events, add the elapsed device time
        count = std::min(count, spans_.size());
        for (size_t i = 0; i < count; i++) {
            float ms = 0.f;
            world->check(world->be->hnh_event_sync(world->ctx, spans_[i].e1), "hnh_event_sync");
            world->check(world->be->hnh_event_elapsed_ms(world->ctx, spans_[i].e0, spans_[i].e1, &ms), "hnh_event_elapsed_ms");
            total_time[perf_counter_keys[(size_t)spans_[i].key]] += (double)ms * 1e-3;
            if (spans_[i].own0) spare_events_.push_back(spans_[i].e0);
            if (spans_[i].own1) spare_events_.push_back(spans_[i].e1);
        }
        spans_.erase(spans_.begin(), spans_.begin() + (long)count);
    }

    // small pool of events for compute/communication hand-offs
    std::vector<void*> events_;
    void* event(size_t i) {
        while (events_.size() <= i) events_.push_back(world->event_create());
        return events_[i];
    }
    // make `waiter` stream wait for everything enqueued so far on `signaller`
    void order(int signaller, int waiter, size_t slot) {
        void* e = event(slot);
        world->event_record(e, signaller);
        world->event_wait(e, waiter);
    }
};
