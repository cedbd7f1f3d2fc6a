// The reference's documented extension point (README.md:17-18, sparse_kernels.h:15-79): a user-supplied
// KernelImplementation.  This plugin implements ONLY the two pure virtuals of the reference's interface — here by
// delegating to StandardKernel and counting the calls — and is handed to every schedule constructor exactly as in the
// reference.  The program checks that each schedule produces the same SDDMM / SpMM / fusedSpMM results through the plugin
// as through StandardKernel (the local-kernel-fusion schedule then runs on the default fused_local(), i.e. the
// reference's own sddmm_local + spmm_local pair per visiting block), and that error conventions hold.
//
//   custom_kernel <kernel library path or ""> <logM> <edgeFactor> <R>
#include <cmath>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "cannon_dense_25d.hpp"
#include "cannon_sparse_25d.hpp"
#include "dense_shift_15d.hpp"
#include "sparse_shift_15d.hpp"

using namespace std;

class CountingKernel : public KernelImplementation {
public:
    StandardKernel inner;
    size_t sddmm_calls = 0, spmm_calls = 0;
    size_t sddmm_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, int block, int offset) override {
        sddmm_calls++;
        return inner.sddmm_local(S, A, B, block, offset);
    }
    size_t spmm_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, MatMode mode, int block) override {
        spmm_calls++;
        return inner.spmm_local(S, A, B, mode, block);
    }
};

// A plugin written against round 2-4's window contract: it honours CSRLocal::window — ONE column range of a block per pass — and has
// never heard of window ranges (round 5's adaptive windows: [window, window_end)), so it does not override handles_window_ranges().
// The schedule must therefore never hand it more than one window at a time; it checks that on every call and computes the window
// it is told about by reading `window` alone (by narrowing the selection to it before it delegates).
class OneWindowKernel : public KernelImplementation {
public:
    StandardKernel inner;
    size_t windowed_calls = 0, range_violations = 0;
    bool handles_windows() const override { return true; }
    void look(SpmatLocal& S, int block) {
        CSRLocal* blk = S.csr_blocks[block];
        if (blk == nullptr || blk->window < 0) return;
        windowed_calls++;
        if (blk->window_end > blk->window + 1) range_violations++;
        blk->window_end = blk->window + 1;  // "reads window alone": whatever else was selected is not computed by this plugin
    }
    size_t sddmm_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, int block, int offset) override {
        look(S, block);
        return inner.sddmm_local(S, A, B, block, offset);
    }
    size_t spmm_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, MatMode mode, int block) override {
        look(S, block);
        return inner.spmm_local(S, A, B, mode, block);
    }
    size_t fused_local(SpmatLocal& S, DenseMatrix& A, DenseMatrix& B, DenseMatrix& Out, int block, unsigned flags,
                       const hnh_fused_extras* extras = nullptr) override {
        look(S, block);
        return inner.fused_local(S, A, B, Out, block, flags, extras);
    }
};

static Distributed_Sparse* make(const string& alg, SpmatLocal* S, int R, KernelImplementation* k) {
    if (alg == "15d_fusion1") return new Sparse15D_Dense_Shift(S, R, 1, 1, k);
    if (alg == "15d_fusion2") return new Sparse15D_Dense_Shift(S, R, 1, 2, k);
    if (alg == "15d_sparse") return new Sparse15D_Sparse_Shift(S, R, 1, k);
    if (alg == "25d_dense_replicate") return new Sparse25D_Cannon_Dense(S, R, 1, k);
    return new Sparse25D_Cannon_Sparse(S, R, 1, k);
}

static vector<vector<double>> run(const string& alg, SpmatLocal* S, int R, KernelImplementation* k) {
    unique_ptr<Distributed_Sparse> d(make(alg, S, R, k));
    DenseMatrix A = d->like_A_matrix(0.0), B = d->like_B_matrix(0.0);
    VectorXd Sv = d->like_S_values(1.0), res = d->like_S_values(0.0);
    vector<vector<double>> out;
    auto refill = [&] { d->dummyInitialize(A, Amat); d->dummyInitialize(B, Bmat); A *= 1e-3; B *= 1e-3; };
    refill(); d->initial_shift(&A, &B, k_sddmmA); d->sddmmA(A, B, Sv, res); d->de_shift(&A, &B, k_sddmmA);
    out.push_back(res.to_host());
    refill(); d->initial_shift(&A, &B, k_spmmA); d->spmmA(A, B, Sv); d->de_shift(&A, &B, k_spmmA);
    out.push_back(A.to_host());
    refill(); d->initial_shift(&A, &B, k_sddmmA); d->fusedSpMM(A, B, Sv, res, Amat); d->de_shift(&A, &B, k_sddmmA);
    out.push_back(A.to_host());
    return out;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        cerr << "usage: custom_kernel <kernel library path or \"\"> logM edgeFactor R" << endl;
        return 2;
    }
    hnh::Backend* be = hnh::load_backend(string(argv[1]).empty() ? nullptr : argv[1]);
    hnh::SingleWorld world(be, 0);
    hnh::set_current_world(&world);
    const int logM = atoi(argv[2]), ef = atoi(argv[3]), R = atoi(argv[4]);
    int bad = 0;
    {
        SpmatLocal S;
        S.loadTuples(false, logM, ef, "");
        for (const string alg : {"15d_fusion1", "15d_fusion2", "15d_sparse", "25d_dense_replicate", "25d_sparse_replicate"}) {
            StandardKernel standard;
            CountingKernel plugin;
            auto want = run(alg, &S, R, &standard);
            auto got = run(alg, &S, R, &plugin);
            double worst = 0.0;
            for (size_t k = 0; k < want.size(); k++) {
                double scale = 0.0, err = 0.0;
                if (want[k].size() != got[k].size()) { bad++; continue; }
                for (size_t e = 0; e < want[k].size(); e++) {
                    scale = max(scale, fabs(want[k][e]));
                    err = max(err, fabs(want[k][e] - got[k][e]));
                }
                worst = max(worst, scale > 0 ? err / scale : err);
            }
            const bool ok = worst <= 1e-11 && plugin.sddmm_calls > 0 && plugin.spmm_calls > 0;
            cout << alg << ": plugin vs StandardKernel rel err " << worst << ", sddmm_local calls " << plugin.sddmm_calls
                 << ", spmm_local calls " << plugin.spmm_calls << (ok ? " ok" : " MISMATCH") << endl;
            if (!ok) bad++;
        }
    }
    world.sync_all();
    hnh::set_current_world(nullptr);
    // ---- the window contract on 4 logical ranks (the mesh fetch of the 1.5D dense shift walks the fetched blocks window by window):
    // StandardKernel takes window ranges, the one-window plugin must get one window per pass and the same results
    {
        const int p = 4;
        auto group = hnh::make_thread_group(p);
        vector<vector<vector<double>>> want(p), got(p);
        vector<size_t> calls(p, 0), violations(p, 0);
        vector<thread> threads;
        for (int r = 0; r < p; r++)
            threads.emplace_back([&, r] {
                hnh::ThreadWorld w(group, r, be, 0);
                hnh::set_current_world(&w);
                {
                    SpmatLocal S;
                    S.loadTuples(false, logM, ef, "");
                    StandardKernel standard;
                    OneWindowKernel plugin;
                    want[r] = run("15d_fusion2", &S, R, &standard);
                    got[r] = run("15d_fusion2", &S, R, &plugin);
                    calls[r] = plugin.windowed_calls;
                    violations[r] = plugin.range_violations;
                }
                w.sync_all();
                w.barrier();
                hnh::set_current_world(nullptr);
            });
        for (auto& t : threads) t.join();
        double worst = 0.0;
        size_t ncalls = 0, nviol = 0;
        for (int r = 0; r < p; r++) {
            ncalls += calls[r];
            nviol += violations[r];
            for (size_t k = 0; k < want[r].size(); k++)
                for (size_t e = 0; e < want[r][k].size(); e++) {
                    const double scale = max(1e-300, fabs(want[r][k][e]));
                    worst = max(worst, fabs(want[r][k][e] - got[r][k][e]) / max(scale, 1e-6));
                }
        }
        const bool ok = worst <= 1e-11 && ncalls > 0 && nviol == 0;
        cout << "window contract (4 ranks, 15d_fusion2): one-window plugin vs StandardKernel rel err " << worst << ", windowed calls " << ncalls
             << ", passes handed more than one window " << nviol << (ok ? " ok" : " MISMATCH") << endl;
        if (!ok) bad++;
    }
    cout << (bad ? "custom kernel plugin: FAILED" : "custom kernel plugin: all schedules ok") << endl;
    return bad ? 1 : 0;
}
