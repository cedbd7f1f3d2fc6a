// ALS by batched conjugate gradients on the normal equations — same classes and algorithm as the
// reference's als_conjugate_gradients.{h,cpp}:  ALS_CG (abstract: cg_optimizer, run_cg, allreduceVector) and
// Distributed_ALS (computeRHS, computeQueries, computeResidual, initializeEmbeddings).
//
// One CG iteration = one fusedSpMM (computeQueries, .cpp:265-301) + three row-wise dot products and three
// row-scaled updates (.cpp:82-139).  All of it stays on the GPU, and the dense passes around the fused call are
// folded together.  Schedules that keep whole rows on a rank (no R split): the ENTIRE rest of the iteration — `+ lambda p`,
// <p, Mp>, alpha, x += alpha p, r -= alpha Mp, <r, r>, p = r + beta p — runs in the fused kernel's row epilogue on the row
// that is in registers anyway (hnh_cg_update; where the schedule has no single fused pass, as one row-wise launch after it):
// beside the fused call an iteration reads x, r and writes x, r, p, instead of the reference's 17 matrix-sized passes.
// R-split schedules need the reference's two Allreduces between those steps (.cpp:95-97,122-124; an RCCL all-reduce on the
// schedule's R-split communicator): there `+ lambda p` and <p, Mp> ride in the epilogue and x += alpha p, r -= alpha Mp,
// <r, r> are one launch (hnh_cg_step_f64) — 9 passes.
//
// Differences, deliberate: the reference initialises embeddings and the artificial ground truth with
// Eigen's setRandom() on LOCAL buffers (.cpp:143-146), which makes results depend on the distribution; here
// every random fill is a hash of the GLOBAL (row, column), so the factorisation is the same for every
// schedule and rank count (and can be compared with the reference driven through public members).
#pragma once
#include <cstdlib>
#include "distributed_sparse.hpp"
#include "er_generator.hpp"

class ALS_CG {
public:
    Distributed_Sparse* d_ops = nullptr;
    DenseMatrix A, B;
    hnh::Comm A_R_split_world, B_R_split_world;
    int proc_rank = 0;
    double application_communication_time = 0.0;

    virtual void computeRHS(MatMode matrix_to_optimize, DenseMatrix& rhs) = 0;
    virtual void computeQueries(DenseMatrix& A, DenseMatrix& B, MatMode matrix_to_optimize, DenseMatrix& result) = 0;
    // computeQueries plus dot[i] = <x[i,:], result[i,:]> with x the operand being optimised (the pair the CG loop
    // always issues together, als_conjugate_gradients.cpp:89-93); subclasses may produce both in one pass
    virtual void computeQueriesDot(DenseMatrix& A, DenseMatrix& B, MatMode matrix_to_optimize, DenseMatrix& result, VectorXd& dot) {
        computeQueries(A, B, matrix_to_optimize, result);
        DenseMatrix& x = (matrix_to_optimize == Amat) ? A : B;
        hnh::World* w = d_ops->world;
        w->check(w->be->hnh_rowdot_f64(w->ctx, x.data(), result.data(), dot.data(), x.rows(), (int)x.cols(), HNH_STREAM_COMPUTE), "hnh_rowdot_f64");
    }
    // computeQueries plus everything else one CG iteration does to its rows (hnh_cg_update: x, r, p, rsold updated in place;
    // cg.p is the operand being optimised in this call).  Only for schedules without an R split.
    virtual void computeQueriesCG(DenseMatrix& A, DenseMatrix& B, MatMode matrix_to_optimize, DenseMatrix& result, const hnh_cg_update& cg) {
        computeQueries(A, B, matrix_to_optimize, result);
        DenseMatrix& x = (matrix_to_optimize == Amat) ? A : B;
        hnh::World* w = d_ops->world;
        hnh_fused_extras ex = {0.0, 0.0, nullptr, &cg, nullptr, 0};
        w->check(w->be->hnh_row_epilogue_x(w->ctx, result.data(), x.data(), &ex, x.rows(), (int)x.cols(), HNH_STREAM_COMPUTE), "hnh_row_epilogue_x");
    }
    virtual double computeResidual() = 0;
    virtual void initializeEmbeddings() = 0;
    virtual ~ALS_CG() {}

    // false: keep the CG updates as separate launches also where they could ride in the fused call (measurement / tests)
    bool fold_cg_updates = std::getenv("HNH_ALS_UNFOLDED") == nullptr;

    void allreduceVector(VectorXd& vec, const hnh::Comm& comm) {
        auto t = start_clock();
        d_ops->world->allreduce_f64(comm, vec.data(), (size_t)vec.size(), HNH_STREAM_COMPUTE);
        if (d_ops->world->timing_sync) d_ops->world->sync_all();
        application_communication_time += stop_clock_get_elapsed(t);
    }

    // batched CG, one independent system per row (als_conjugate_gradients.cpp:38-141)
    void cg_optimizer(MatMode matrix_to_optimize, int cg_max_iter) {
        const double nan_avoidance_constant = 1e-8;
        hnh::World* w = d_ops->world;
        hnh::Backend* be = w->be;
        const hnh::Comm& reduction_world = (matrix_to_optimize == Amat) ? A_R_split_world : B_R_split_world;
        DenseMatrix& X = (matrix_to_optimize == Amat) ? A : B;
        const int64_t nrows = X.rows();
        const int ncols = (int)A.cols();

        // the other factor is fixed for this whole half-step (rhs + 1 + cg_max_iter fused calls): schedules that fetch
        // its blocks from other ranks may keep them
        struct Hold {
            Distributed_Sparse* d;
            Hold(Distributed_Sparse* d_in, const DenseMatrix* m) : d(d_in) { d->hold_moving_operand(m); }
            ~Hold() { d->release_moving_operand(); }
        } hold(d_ops, matrix_to_optimize == Amat ? &B : &A);

        DenseMatrix rhs(nrows, ncols), Mx(nrows, ncols), Mp(nrows, ncols);
        rhs.setZero();
        computeRHS(matrix_to_optimize, rhs);
        computeQueries(A, B, matrix_to_optimize, Mx);

        auto rowdot = [&](DenseMatrix& x, DenseMatrix& y, VectorXd& out) {
            w->check(be->hnh_rowdot_f64(w->ctx, x.data(), y.data(), out.data(), nrows, ncols, HNH_STREAM_COMPUTE), "hnh_rowdot_f64");
        };
        auto row_update = [&](DenseMatrix& y, const double* yv, double ya, DenseMatrix& x, const double* xv, double xa) {
            w->check(be->hnh_row_scale_add_f64(w->ctx, y.data(), yv, ya, x.data(), xv, xa, nrows, ncols, HNH_STREAM_COMPUTE),
                     "hnh_row_scale_add_f64");
        };

        DenseMatrix r = rhs;                       // r = rhs - Mx
        row_update(r, nullptr, 1.0, Mx, nullptr, -1.0);
        DenseMatrix p = r;
        VectorXd rsold(nrows), alpha(nrows), coeffs(nrows), bdot(nrows), rsnew(nrows);
        rowdot(r, r, rsold);
        if (d_ops->r_split) allreduceVector(rsold, reduction_world);

        const bool folded = fold_cg_updates && !d_ops->r_split;
        for (int cg_iter = 0; cg_iter < cg_max_iter; cg_iter++) {
            if (folded) {  // every row's dot products are complete on this rank: the whole iteration is one call
                const hnh_cg_update cg = {X.data(), r.data(), p.data(), rsold.data(), nan_avoidance_constant};
                if (matrix_to_optimize == Amat) computeQueriesCG(p, B, Amat, Mp, cg);
                else computeQueriesCG(A, p, Bmat, Mp, cg);
                continue;
            }
            if (matrix_to_optimize == Amat) computeQueriesDot(p, B, Amat, Mp, bdot);  // Mp and bdot = <p, Mp> row-wise
            else computeQueriesDot(A, p, Bmat, Mp, bdot);
            if (d_ops->r_split) allreduceVector(bdot, reduction_world);

            // the reference adds the constant to BOTH vectors in place (.cpp:99-100); rsold keeps it
            w->check(be->hnh_vec_add_scalar_f64(w->ctx, bdot.data(), nan_avoidance_constant, nrows, HNH_STREAM_COMPUTE), "vec_add");
            w->check(be->hnh_vec_add_scalar_f64(w->ctx, rsold.data(), nan_avoidance_constant, nrows, HNH_STREAM_COMPUTE), "vec_add");
            w->check(be->hnh_vec_div_f64(w->ctx, alpha.data(), rsold.data(), bdot.data(), nrows, HNH_STREAM_COMPUTE), "vec_div");

            // X += alpha .* p;  r -= alpha .* Mp;  rsnew = <r, r> row-wise  (.cpp:117-127, one pass)
            w->check(be->hnh_cg_step_f64(w->ctx, X.data(), r.data(), p.data(), Mp.data(), alpha.data(), rsnew.data(), nrows, ncols,
                                         HNH_STREAM_COMPUTE), "hnh_cg_step_f64");
            if (d_ops->r_split) allreduceVector(rsnew, reduction_world);

            w->check(be->hnh_vec_div_f64(w->ctx, coeffs.data(), rsnew.data(), rsold.data(), nrows, HNH_STREAM_COMPUTE), "vec_div");
            row_update(p, coeffs.data(), 1.0, r, nullptr, 1.0);   // p = r + coeffs .* p
            std::swap(rsold, rsnew);                              // rsold = rsnew
        }
    }

    void run_cg(int n_alternating_steps) {  // als_conjugate_gradients.cpp:235-263
        initializeEmbeddings();
        if (proc_rank == 0) std::cout << "Embeddings initialized +" << std::endl;
        for (int i = 0; i < n_alternating_steps; i++) {
            cg_optimizer(Amat, 10);
            cg_optimizer(Bmat, 10);
            if (proc_rank == 0 && i < n_alternating_steps - 1) std::cout << "Completed step " << i << std::endl;
        }
    }
};

class Distributed_ALS : public ALS_CG {
public:
    VectorXd ground_truth, ground_truth_transpose;
    uint64_t seed = 2022;
    VectorXd ones_S, scratch_S, ones_ST, scratch_ST;  // persistent arguments of computeQueries

    // artificial_groundtruth: ground truth = SDDMM of two hashed random factor matrices with S = 1
    // (als_conjugate_gradients.cpp:157-184); otherwise the caller sets ground_truth{,_transpose}.
    Distributed_ALS(Distributed_Sparse* d_ops, bool artificial_groundtruth, uint64_t seed_in = 2022) {
        seed = seed_in;
        this->d_ops = d_ops;
        proc_rank = d_ops->proc_rank;
        A_R_split_world = d_ops->A_R_split_world;
        B_R_split_world = d_ops->B_R_split_world;
        if (artificial_groundtruth) {
            DenseMatrix Agt = d_ops->like_A_matrix(0.0), Bgt = d_ops->like_B_matrix(0.0);
            hashed_fill(Agt, Amat, seed + 1, 1.0 / ((double)d_ops->R * (double)d_ops->M * (double)d_ops->R));
            hashed_fill(Bgt, Bmat, seed + 2, 1.0 / ((double)d_ops->R * (double)d_ops->N * (double)d_ops->R));
            VectorXd ones = d_ops->like_S_values(1.0);
            ground_truth = d_ops->like_S_values(0.0);
            d_ops->initial_shift(&Agt, &Bgt, k_sddmmA);
            d_ops->sddmmA(Agt, Bgt, ones, ground_truth);
            d_ops->de_shift(&Agt, &Bgt, k_sddmmA);
            ones = d_ops->like_ST_values(1.0);
            ground_truth_transpose = d_ops->like_ST_values(0.0);
            d_ops->initial_shift(&Agt, &Bgt, k_sddmmB);
            d_ops->sddmmB(Agt, Bgt, ones, ground_truth_transpose);
            d_ops->de_shift(&Agt, &Bgt, k_sddmmB);
        }
    }

    void computeRHS(MatMode matrix_to_optimize, DenseMatrix& rhs) override {  // .cpp:186-199
        if (matrix_to_optimize == Amat) {
            d_ops->initial_shift(&rhs, &B, k_spmmA);
            d_ops->spmmA(rhs, B, ground_truth);
            d_ops->de_shift(&rhs, &B, k_spmmA);
        } else {
            d_ops->initial_shift(&A, &rhs, k_spmmB);
            d_ops->spmmB(A, rhs, ground_truth_transpose);
            d_ops->de_shift(&A, &rhs, k_spmmB);
        }
    }

    double computeResidual() override {  // .cpp:201-219
        VectorXd ones = d_ops->like_S_values(1.0), sddmm_result = d_ops->like_S_values(0.0);
        d_ops->initial_shift(&A, &B, k_sddmmA);
        d_ops->sddmmA(A, B, ones, sddmm_result);
        d_ops->de_shift(&A, &B, k_sddmmA);
        std::vector<double> x = sddmm_result.to_host(), g = ground_truth.to_host();
        double sqnorm = 0.0;
        for (size_t e = 0; e < x.size(); e++) sqnorm += (x[e] - g[e]) * (x[e] - g[e]);
        return std::sqrt(d_ops->world->host_allreduce_sum(sqnorm));
    }

    void initializeEmbeddings() override {  // .cpp:221-233: random/R, then A *= 1.4, B /= 1.3
        A = d_ops->like_A_matrix(0.0);
        B = d_ops->like_B_matrix(0.0);
        hashed_fill(A, Amat, seed + 3, 1.4 / (double)d_ops->R);
        hashed_fill(B, Bmat, seed + 4, 1.0 / (1.3 * (double)d_ops->R));
    }

    // result = (S .* (X Y^T)|_S) Y + lambda X  with S == 1 (.cpp:265-301)
    void computeQueries(DenseMatrix& A_in, DenseMatrix& B_in, MatMode matrix_to_optimize, DenseMatrix& result) override {
        queries(A_in, B_in, matrix_to_optimize, result, nullptr);
    }
    void computeQueriesDot(DenseMatrix& A_in, DenseMatrix& B_in, MatMode matrix_to_optimize, DenseMatrix& result, VectorXd& dot) override {
        queries(A_in, B_in, matrix_to_optimize, result, &dot);
    }
    void computeQueriesCG(DenseMatrix& A_in, DenseMatrix& B_in, MatMode matrix_to_optimize, DenseMatrix& result, const hnh_cg_update& cg) override {
        queries(A_in, B_in, matrix_to_optimize, result, nullptr, &cg);
    }

    void queries(DenseMatrix& A_in, DenseMatrix& B_in, MatMode matrix_to_optimize, DenseMatrix& result, VectorXd* dot,
                 const hnh_cg_update* cg = nullptr) {
        const double lambda = 1e-13;
        hnh::World* w = d_ops->world;
        DenseMatrix& x = (matrix_to_optimize == Amat) ? A_in : B_in;
        // Schedules with a single fused pass take the + lambda x and the row-wise <x, result> in the same launch and
        // write `result` directly (no `result = x` copy); the 1.5D dense schedule's shifts are empty (.cpp:280,284).
        if (result.rows() != x.rows() || result.cols() != x.cols()) result = DenseMatrix(x.rows(), x.cols());
        hnh_fused_extras ex = {0.0, lambda, dot ? dot->data() : nullptr, cg, nullptr, 0};
        if (d_ops->fusedSpMM_out(A_in, B_in, matrix_to_optimize, result, false, ex)) return;

        // the all-ones S values and the SDDMM scratch vector are the same for every call: keep them
        // (the reference re-creates both per call, .cpp:276-277,289-290)
        if (matrix_to_optimize == Amat) {
            if (ones_S.size() == 0) { ones_S = d_ops->like_S_values(1.0); scratch_S = d_ops->like_S_values(0.0); }
            VectorXd &ones = ones_S, &sddmm_result = scratch_S;
            result = A_in;
            d_ops->initial_shift(&result, &B_in, k_sddmmA);
            d_ops->fusedSpMM(result, B_in, ones, sddmm_result, Amat);
            d_ops->de_shift(&result, &B_in, k_sddmmA);
        } else {
            if (ones_ST.size() == 0) { ones_ST = d_ops->like_ST_values(1.0); scratch_ST = d_ops->like_ST_values(0.0); }
            VectorXd &ones = ones_ST, &sddmm_result = scratch_ST;
            result = B_in;
            d_ops->initial_shift(&A_in, &result, k_sddmmB);
            d_ops->fusedSpMM(A_in, result, ones, sddmm_result, Bmat);
            d_ops->de_shift(&A_in, &result, k_sddmmB);
        }
        w->check(w->be->hnh_row_epilogue_x(w->ctx, result.data(), x.data(), &ex, result.rows(), (int)result.cols(), HNH_STREAM_COMPUTE),
                 "hnh_row_epilogue_x");
    }

    // uniform(-1, 1) * scale keyed by the GLOBAL (row, col) through the operator's submatrix descriptors; generated
    // on the device (hnh_fill_hashed_f64), one launch per submatrix
    void hashed_fill(DenseMatrix& loc, MatMode mode, uint64_t fill_seed, double scale) {
        std::vector<DenseSubmatrix>& subs = (mode == Amat) ? d_ops->aSubmatrices : d_ops->bSubmatrices;
        hnh::World* w = d_ops->world;
        double* ptr = loc.data();
        for (auto& s : subs) {
            w->check(w->be->hnh_fill_hashed_f64(w->ctx, ptr, s.rowCount, s.colCount, s.topRow, s.leftCol, d_ops->R, fill_seed, scale,
                                                HNH_STREAM_COMPUTE),
                     "hnh_fill_hashed_f64");
            ptr += (size_t)s.rowCount * s.colCount;
        }
    }
};
