// TEST INFRASTRUCTURE ONLY — never linked into the product, never used as a compute path.
//
// Oracle driver: runs the UNMODIFIED reference implementation (PASSIONLab/distributed_sddmm, sources
// compiled where they lie under /root/reference; see oracle/Makefile) on inputs that this repo
// supplies, and dumps results keyed by GLOBAL coordinates so they can be compared with the HIP path.
//
// It only calls the reference's public operator API (distributed_sparse.h:274-320):
//   like_{A,B}_matrix, like_{S,ST}_values, initial_shift, de_shift, sddmmA/B, spmmA/B, fusedSpMM,
// constructed exactly as benchmark_dist.cpp:45-82 does.  Inputs are injected directly into
// SpmatLocal::{coords,M,N,dist_nnz,initialized} (CombBLAS generation/I-O is replaced).
//
// Because the reference frees its coordinates in the constructors (15D_dense_shift.hpp:122,124) and
// returns SDDMM values in per-block CSR order, the (row, col) key of every local value slot is
// recovered through the public API with a "coordinate probe" SDDMM:  A[i,:] = (i, 1, 0...),
// B[j,:] = (N, j, 0...)  =>  <A[i,:], B[j,:]> = i*N + j  exactly (integers < 2^53).
//
// Modes:
//   ref_driver dump  <case.bin> <alg> <c> <outprefix>     element-wise results, all ops
//   ref_driver fp    <case.bin> <alg> <c>                 scratch.cpp:26-76 style fingerprints
//   ref_driver bench <case.bin> <alg> <c> <fused> <trials> benchmark_dist.cpp:102-162 timing loop
//   ref_driver sweep <case.bin> <alg> <c> <fused> <trials> <reps> <t1,t2,...>   the same timing loop at several OpenMP/MKL thread
//        counts behind ONE set-up: per count one warm-up call, then <reps> batches of <trials> timed calls (every batch reported)
//   ref_driver als   <case.bin> <alg> <c> <outprefix> <steps> <cg_iters>   ALS-CG (als_conjugate_gradients.cpp):
//        ground truth = the input S values (keyed by coordinate), embeddings initialised from the case's A, B,
//        `steps` x { cg_optimizer(Amat, cg_iters); cg_optimizer(Bmat, cg_iters) } (= run_cg, :235-263), dumps A, B
//   ref_driver gat   <case.bin> <alg> <c> <outprefix> <alpha> <in:fph:heads>[,<in:fph:heads>...]   GAT forward pass
//        (gat.hpp:106-112) with input features = the case's A, weights W(layer, head)[k][j] = hashed uniform / K and
//        leaky_relu_alpha = <alpha> set through public members (the reference leaves both unset)
// <alg> in {15d_fusion1, 15d_fusion2, 15d_sparse, 25d_dense_replicate, 25d_sparse_replicate}.
#include "15D_dense_shift.hpp"
#include "15D_sparse_shift.hpp"
#include "25D_cannon_dense.hpp"
#include "25D_cannon_sparse.hpp"
#include "gat.hpp"

#include <omp.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Case {
    int64_t M = 0, N = 0, nnz = 0, R = 0;
    std::vector<int64_t> r, c;
    std::vector<double> v, A, B;
    std::unordered_map<int64_t, double> sval;  // key i*N+j -> input S value
};

void die(const std::string& msg) {
    std::fprintf(stderr, "ref_driver: %s\n", msg.c_str());
    MPI_Abort(MPI_COMM_WORLD, 2);
}

template <typename T>
void read_vec(FILE* f, std::vector<T>& out, size_t n) {
    out.resize(n);
    if (n && std::fread(out.data(), sizeof(T), n, f) != n) die("short read");
}

Case load_case(const char* path, bool with_dense) {
    Case cs;
    FILE* f = std::fopen(path, "rb");
    if (!f) die(std::string("cannot open ") + path);
    char magic[8];
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "HNHCASE1", 8) != 0) die("bad magic");
    int64_t hdr[4];
    if (std::fread(hdr, 8, 4, f) != 4) die("bad header");
    cs.M = hdr[0]; cs.N = hdr[1]; cs.nnz = hdr[2]; cs.R = hdr[3];
    read_vec(f, cs.r, cs.nnz);
    read_vec(f, cs.c, cs.nnz);
    read_vec(f, cs.v, cs.nnz);
    if (with_dense) {
        read_vec(f, cs.A, (size_t)cs.M * cs.R);
        read_vec(f, cs.B, (size_t)cs.N * cs.R);
        cs.sval.reserve(cs.nnz * 2);
        for (int64_t e = 0; e < cs.nnz; e++) cs.sval[cs.r[e] * cs.N + cs.c[e]] = cs.v[e];
    }
    std::fclose(f);
    return cs;
}

// every rank keeps an arbitrary (strided) slice, as SpmatLocal::loadTuples would leave it
void inject(SpmatLocal& S, const Case& cs, int rank, int p) {
    S.M = cs.M; S.N = cs.N; S.dist_nnz = cs.nnz; S.initialized = true;
    for (int64_t e = rank; e < cs.nnz; e += p)
        S.coords.push_back({(uint64_t)cs.r[e], (uint64_t)cs.c[e], cs.v[e]});
}

Distributed_Sparse* make_alg(const std::string& name, SpmatLocal* S, int R, int c, KernelImplementation* k) {
    if (name == "15d_fusion1") return new Sparse15D_Dense_Shift(S, R, c, 1, k);
    if (name == "15d_fusion2") return new Sparse15D_Dense_Shift(S, R, c, 2, k);
    if (name == "15d_sparse") return new Sparse15D_Sparse_Shift(S, R, c, k);
    if (name == "25d_dense_replicate") return new Sparse25D_Cannon_Dense(S, R, c, k);
    if (name == "25d_sparse_replicate") return new Sparse25D_Cannon_Sparse(S, R, c, k);
    die("unknown algorithm " + name);
    return nullptr;
}

// fill a local dense buffer from a global row-major matrix through the operator's own submatrix
// descriptors (same walk as dummyInitialize, distributed_sparse.h:322-346); rows past the end -> 0
void fill_local(Distributed_Sparse* d, DenseMatrix& loc, MatMode mode, const double* global,
                int64_t grows, int64_t R) {
    std::vector<DenseSubmatrix>& subs = (mode == Amat) ? d->aSubmatrices : d->bSubmatrices;
    double* ptr = loc.data();
    for (auto& s : subs)
        for (int i = 0; i < s.rowCount; i++)
            for (int j = 0; j < s.colCount; j++) {
                int64_t gr = s.topRow + i, gc = s.leftCol + j;
                *ptr++ = (gr < grows && gc < R) ? global[gr * R + gc] : 0.0;
            }
}

enum ProbeSide { kProbeA, kProbeB };
void fill_probe(Distributed_Sparse* d, DenseMatrix& loc, MatMode mode, int64_t N) {
    std::vector<DenseSubmatrix>& subs = (mode == Amat) ? d->aSubmatrices : d->bSubmatrices;
    double* ptr = loc.data();
    for (auto& s : subs)
        for (int i = 0; i < s.rowCount; i++)
            for (int j = 0; j < s.colCount; j++) {
                int64_t gr = s.topRow + i, gc = s.leftCol + j;
                double x = 0.0;
                if (mode == Amat) x = (gc == 0) ? (double)gr : (gc == 1 ? 1.0 : 0.0);
                else x = (gc == 0) ? (double)N : (gc == 1 ? (double)gr : 0.0);
                *ptr++ = x;
            }
}

void dump(const std::string& prefix, int rank, const char* name, const void* p, size_t bytes) {
    std::string path = prefix + ".r" + std::to_string(rank) + "." + name;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) die("cannot write " + path);
    if (bytes) std::fwrite(p, 1, bytes, f);
    std::fclose(f);
}

void dump_subs(const std::string& prefix, int rank, const char* name, std::vector<DenseSubmatrix>& subs) {
    std::vector<int64_t> flat;
    for (auto& s : subs) { flat.push_back(s.topRow); flat.push_back(s.leftCol); flat.push_back(s.rowCount); flat.push_back(s.colCount); }
    dump(prefix, rank, name, flat.data(), flat.size() * 8);
}

std::vector<int64_t> keys_of(const VectorXd& probe) {
    std::vector<int64_t> k(probe.size());
    for (long e = 0; e < probe.size(); e++) k[e] = (int64_t)std::llround(probe[e]);
    return k;
}

VectorXd svals_for(const Case& cs, const std::vector<int64_t>& keys) {
    VectorXd s = VectorXd::Constant((long)keys.size(), 0.0);
    for (size_t e = 0; e < keys.size(); e++) {
        auto it = cs.sval.find(keys[e]);
        if (it == cs.sval.end()) die("probe produced a key that is not an input nonzero");
        s[e] = it->second;
    }
    return s;
}

void run_dump(const Case& cs, Distributed_Sparse* d, const std::string& prefix) {
    const int rank = d->proc_rank;
    DenseMatrix A = d->like_A_matrix(0.0), B = d->like_B_matrix(0.0);

    // ---- coordinate probes (S ordering and ST ordering) ----
    VectorXd onesS = d->like_S_values(1.0), onesST = d->like_ST_values(1.0);
    VectorXd probeS = d->like_S_values(0.0), probeST = d->like_ST_values(0.0);
    fill_probe(d, A, Amat, cs.N); fill_probe(d, B, Bmat, cs.N);
    d->initial_shift(&A, &B, k_sddmmA); MPI_Barrier(MPI_COMM_WORLD);
    d->sddmmA(A, B, onesS, probeS);
    fill_probe(d, A, Amat, cs.N); fill_probe(d, B, Bmat, cs.N);
    d->initial_shift(&A, &B, k_sddmmB); MPI_Barrier(MPI_COMM_WORLD);
    d->sddmmB(A, B, onesST, probeST);
    std::vector<int64_t> keysS = keys_of(probeS), keysST = keys_of(probeST);
    VectorXd S = svals_for(cs, keysS), ST = svals_for(cs, keysST);

    int64_t dims[8] = {d->localArows, d->localAcols, d->localBrows, d->localBcols, d->p, d->c, cs.M, cs.N};
    dump(prefix, rank, "dims.i64", dims, sizeof(dims));
    dump_subs(prefix, rank, "subA.i64", d->aSubmatrices);
    dump_subs(prefix, rank, "subB.i64", d->bSubmatrices);
    dump(prefix, rank, "keysS.i64", keysS.data(), keysS.size() * 8);
    dump(prefix, rank, "keysST.i64", keysST.data(), keysST.size() * 8);

    auto refill = [&]() {
        fill_local(d, A, Amat, cs.A.data(), cs.M, cs.R);
        fill_local(d, B, Bmat, cs.B.data(), cs.N, cs.R);
    };

    // sddmmA: result[e] = S[e] * <A[i,:], B[j,:]>
    { VectorXd res = d->like_S_values(0.0); refill();
      d->initial_shift(&A, &B, k_sddmmA); MPI_Barrier(MPI_COMM_WORLD);
      d->sddmmA(A, B, S, res); d->de_shift(&A, &B, k_sddmmA);
      dump(prefix, rank, "sddmmA.f64", res.data(), res.size() * 8); }
    // sddmmB
    { VectorXd res = d->like_ST_values(0.0); refill();
      d->initial_shift(&A, &B, k_sddmmB); MPI_Barrier(MPI_COMM_WORLD);
      d->sddmmB(A, B, ST, res); d->de_shift(&A, &B, k_sddmmB);
      dump(prefix, rank, "sddmmB.f64", res.data(), res.size() * 8); }
    // spmmA: A = S * B
    { refill();
      d->initial_shift(&A, &B, k_spmmA); MPI_Barrier(MPI_COMM_WORLD);
      d->spmmA(A, B, S); d->de_shift(&A, &B, k_spmmA);
      dump(prefix, rank, "spmmA.f64", A.data(), A.size() * 8); }
    // spmmB: B = S^T * A
    { refill();
      d->initial_shift(&A, &B, k_spmmB); MPI_Barrier(MPI_COMM_WORLD);
      d->spmmB(A, B, ST); d->de_shift(&A, &B, k_spmmB);
      dump(prefix, rank, "spmmB.f64", B.data(), B.size() * 8); }
    // fusedSpMM(Amat), called as als_conjugate_gradients.cpp:283-286 does
    { VectorXd buf = d->like_S_values(0.0); refill();
      d->initial_shift(&A, &B, k_sddmmA); MPI_Barrier(MPI_COMM_WORLD);
      d->fusedSpMM(A, B, S, buf, Amat); d->de_shift(&A, &B, k_sddmmA);
      dump(prefix, rank, "fusedA.f64", A.data(), A.size() * 8);
      dump(prefix, rank, "fusedA_buf.f64", buf.data(), buf.size() * 8); }
    // fusedSpMM(Bmat), als_conjugate_gradients.cpp:293-295
    { VectorXd buf = d->like_ST_values(0.0); refill();
      d->initial_shift(&A, &B, k_sddmmB); MPI_Barrier(MPI_COMM_WORLD);
      d->fusedSpMM(A, B, ST, buf, Bmat); d->de_shift(&A, &B, k_sddmmB);
      dump(prefix, rank, "fusedB.f64", B.data(), B.size() * 8);
      dump(prefix, rank, "fusedB_buf.f64", buf.data(), buf.size() * 8); }
}

// scratch.cpp:26-76 fingerprints (dummyInitialize fill, S = 1)
void run_fp(Distributed_Sparse* d, const std::string& name) {
    DenseMatrix A = d->like_A_matrix(0.0), B = d->like_B_matrix(0.0);
    VectorXd S = d->like_S_values(1.0), ST = d->like_ST_values(1.0), res = d->like_S_values(0.0);
    d->dummyInitialize(A, Amat); d->dummyInitialize(B, Bmat);
    d->initial_shift(&A, &B, k_sddmmA); MPI_Barrier(MPI_COMM_WORLD);
    d->sddmmA(A, B, S, res);
    double f1 = res.squaredNorm(); MPI_Allreduce(MPI_IN_PLACE, &f1, 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    d->dummyInitialize(A, Amat); d->dummyInitialize(B, Bmat);
    d->initial_shift(&A, &B, k_spmmA); MPI_Barrier(MPI_COMM_WORLD);
    d->spmmA(A, B, S);
    double f2 = A.squaredNorm(); MPI_Allreduce(MPI_IN_PLACE, &f2, 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    d->dummyInitialize(A, Amat); d->dummyInitialize(B, Bmat);
    d->initial_shift(&A, &B, k_spmmB); MPI_Barrier(MPI_COMM_WORLD);
    d->spmmB(A, B, ST);
    double f3 = B.squaredNorm(); MPI_Allreduce(MPI_IN_PLACE, &f3, 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    if (d->proc_rank == 0)
        std::printf("{\"alg\": \"%s\", \"p\": %d, \"c\": %d, \"sddmm\": %.17e, \"spmmA\": %.17e, \"spmmB\": %.17e}\n",
                    name.c_str(), d->p, d->c, f1, f2, f3);
}

// benchmark_dist.cpp:102-162: A = B = 0.001, S = 1, `trials` fused or unfused SDDMM+SpMM calls
void run_bench(const Case& cs, Distributed_Sparse* d, const std::string& name, bool fused, int trials) {
    DenseMatrix A = d->like_A_matrix(0.001), B = d->like_B_matrix(0.001);
    VectorXd S = d->like_S_values(1.0), res = d->like_S_values(0.0);
    // one untimed warm-up call (the reference has none; first-touch / MKL inspector costs are excluded)
    if (fused) d->fusedSpMM(A, B, S, res, Amat); else { d->sddmmA(A, B, S, res); d->spmmA(A, B, S); }
    MPI_Barrier(MPI_COMM_WORLD);
    d->reset_performance_timers();
    my_timer_t t = start_clock();
    for (int it = 0; it < trials; it++) {
        if (fused) d->fusedSpMM(A, B, S, res, Amat);
        else { d->sddmmA(A, B, S, res); d->spmmA(A, B, S); }
    }
    MPI_Barrier(MPI_COMM_WORLD);
    double elapsed = stop_clock_get_elapsed(t);
    json stats = d->json_perf_statistics();
    if (d->proc_rank == 0) {
        double gflops = 2.0 * (double)cs.nnz * 2.0 * (double)d->R * trials / elapsed / 1e9;  // benchmark_dist.cpp:147-149
        json j;
        j["alg_name"] = name; j["fused"] = fused; j["num_trials"] = trials; j["elapsed"] = elapsed;
        j["overall_throughput"] = gflops; j["nnz_R_per_s"] = gflops * 1e9 / 4.0;
        j["nnz"] = cs.nnz; j["r"] = d->R; j["p"] = d->p; j["c"] = d->c; j["perf_stats"] = stats;
        j["omp_threads"] = omp_get_max_threads();
        std::printf("%s\n", j.dump().c_str());
    }
}

extern "C" void MKL_Set_Num_Threads(int);  // mkl_service.h: mkl_set_num_threads

// The timing loop of run_bench at each of `counts` threads, one set-up for all of them (bench.py's cpu_baseline leg: the reference's
// set-up takes several times longer than its timed calls, so a sweep of separate runs is mostly set-up).
void run_sweep(const Case& cs, Distributed_Sparse* d, const std::string& name, bool fused, int trials, int reps, const std::vector<int>& counts) {
    DenseMatrix A = d->like_A_matrix(0.001), B = d->like_B_matrix(0.001);
    VectorXd S = d->like_S_values(1.0), res = d->like_S_values(0.0);
    auto call = [&]() { if (fused) d->fusedSpMM(A, B, S, res, Amat); else { d->sddmmA(A, B, S, res); d->spmmA(A, B, S); } };
    json points = json::array();
    for (int t : counts) {
        omp_set_num_threads(t);
        MKL_Set_Num_Threads(t);
        call();  // untimed warm-up at this thread count
        json batches = json::array();
        double best = -1.0, best_comp = 0.0;
        for (int rep = 0; rep < reps; rep++) {
            MPI_Barrier(MPI_COMM_WORLD);
            d->reset_performance_timers();
            my_timer_t tm = start_clock();
            for (int it = 0; it < trials; it++) call();
            MPI_Barrier(MPI_COMM_WORLD);
            double elapsed = stop_clock_get_elapsed(tm);
            json stats = d->json_perf_statistics();
            batches.push_back(elapsed);
            if (best < 0.0 || elapsed < best) {
                best = elapsed;
                best_comp = stats.contains("Computation Time") ? stats["Computation Time"].get<double>() : 0.0;
            }
        }
        json pt;
        pt["threads"] = t; pt["elapsed"] = best; pt["batches"] = batches; pt["computation_time"] = best_comp;
        pt["nnz_R_per_s"] = (double)cs.nnz * (double)d->R * trials / best;
        points.push_back(pt);
    }
    if (d->proc_rank == 0) {
        json j;
        j["alg_name"] = name; j["fused"] = fused; j["num_trials"] = trials; j["reps"] = reps; j["nnz"] = cs.nnz; j["r"] = d->R;
        j["p"] = d->p; j["c"] = d->c; j["points"] = points;
        std::printf("%s\n", j.dump().c_str());
    }
}

// ALS-CG on the reference's own Distributed_ALS (als_conjugate_gradients.cpp:148-301).  The random
// initialisations (:143-146, :225-233) are replaced by the case's keyed fills through public members.
void run_als(const Case& cs, Distributed_Sparse* d, const std::string& prefix, int steps, int cg_iters) {
    const int rank = d->proc_rank;
    DenseMatrix A = d->like_A_matrix(0.0), B = d->like_B_matrix(0.0);
    VectorXd onesS = d->like_S_values(1.0), onesST = d->like_ST_values(1.0);
    VectorXd probeS = d->like_S_values(0.0), probeST = d->like_ST_values(0.0);
    fill_probe(d, A, Amat, cs.N); fill_probe(d, B, Bmat, cs.N);
    d->initial_shift(&A, &B, k_sddmmA); MPI_Barrier(MPI_COMM_WORLD);
    d->sddmmA(A, B, onesS, probeS);
    fill_probe(d, A, Amat, cs.N); fill_probe(d, B, Bmat, cs.N);
    d->initial_shift(&A, &B, k_sddmmB); MPI_Barrier(MPI_COMM_WORLD);
    d->sddmmB(A, B, onesST, probeST);

    Distributed_ALS als(d, false);
    als.ground_truth = svals_for(cs, keys_of(probeS));
    als.ground_truth_transpose = svals_for(cs, keys_of(probeST));
    als.A = d->like_A_matrix(0.0);
    als.B = d->like_B_matrix(0.0);
    fill_local(d, als.A, Amat, cs.A.data(), cs.M, cs.R);
    fill_local(d, als.B, Bmat, cs.B.data(), cs.N, cs.R);
    std::vector<double> residuals;
    residuals.push_back(als.computeResidual());
    for (int s = 0; s < steps; s++) {
        als.cg_optimizer(Amat, cg_iters);
        als.cg_optimizer(Bmat, cg_iters);
        residuals.push_back(als.computeResidual());
    }
    int64_t dims[8] = {d->localArows, d->localAcols, d->localBrows, d->localBcols, d->p, d->c, cs.M, cs.N};
    dump(prefix, rank, "dims.i64", dims, sizeof(dims));
    dump_subs(prefix, rank, "subA.i64", d->aSubmatrices);
    dump_subs(prefix, rank, "subB.i64", d->bSubmatrices);
    dump(prefix, rank, "alsA.f64", als.A.data(), als.A.size() * 8);
    dump(prefix, rank, "alsB.f64", als.B.data(), als.B.size() * 8);
    dump(prefix, rank, "residuals.f64", residuals.data(), residuals.size() * 8);
}

double hashed_uniform(uint64_t key, uint64_t seed) {  // twin of oracle/oracle.py:hashed_uniform
    auto mix = [](uint64_t x) {
        uint64_t z = x + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    const uint64_t h = mix(seed * 0xD1342543DE82EF95ull + key * 0x9E3779B97F4A7C15ull);
    return (double)(h >> 11) * 0x1.0p-52 - 1.0;
}

// GAT forward pass on the reference's own gat.hpp.  Only meaningful for schedules that do not split R
// (the reference multiplies LOCAL column slices by W without reducing, gat.hpp:88).
void run_gat(const Case& cs, Distributed_Sparse* d, const std::string& prefix, double alpha, const std::string& spec) {
    const int rank = d->proc_rank;
    std::vector<GATLayer> layers;
    size_t pos = 0;
    while (pos < spec.size()) {
        size_t end = spec.find(',', pos);
        if (end == std::string::npos) end = spec.size();
        int in = 0, fph = 0, heads = 0;
        if (std::sscanf(spec.substr(pos, end - pos).c_str(), "%d:%d:%d", &in, &fph, &heads) != 3) die("bad layer spec");
        layers.emplace_back(in, fph, heads);
        pos = end + 1;
    }
    GAT gnn(layers, d);
    gnn.leaky_relu_alpha = alpha;
    d->setRValue(gnn.layers[0].input_features);
    fill_local(d, gnn.buffers[0], Bmat, cs.A.data(), cs.N, cs.R);
    for (size_t l = 0; l < gnn.layers.size(); l++)
        for (int h = 0; h < gnn.layers[l].num_heads; h++) {
            DenseMatrix& W = gnn.layers[l].wMats[h];
            for (long k = 0; k < W.rows(); k++)
                for (long j = 0; j < W.cols(); j++)
                    W(k, j) = hashed_uniform((((uint64_t)l * 64 + (uint64_t)h) * 65536 + (uint64_t)k) * 65536 + (uint64_t)j, 31) / (double)W.rows();
        }
    gnn.forwardPass();
    const GATLayer& last = gnn.layers.back();
    d->setRValue(last.features_per_head * last.num_heads);
    int64_t dims[8] = {d->localArows, d->localAcols, d->localBrows, d->localBcols, d->p, d->c, cs.M, cs.N};
    dump(prefix, rank, "dims.i64", dims, sizeof(dims));
    dump_subs(prefix, rank, "subA.i64", d->aSubmatrices);
    dump(prefix, rank, "gat.f64", gnn.buffers.back().data(), gnn.buffers.back().size() * 8);
}

}  // namespace

int main(int argc, char** argv) {
    MPI_Init(&argc, &argv);
    initialize_mpi_datatypes();
    int rank, p;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &p);
    if (argc < 5) die("usage: ref_driver dump|fp|bench <case.bin> <alg> <c> [...]");
    std::string mode = argv[1], alg = argv[3];
    int c = std::atoi(argv[4]);
    Case cs = load_case(argv[2], mode == "dump" || mode == "als" || mode == "gat");
    {
        SpmatLocal S;
        inject(S, cs, rank, p);
        StandardKernel kernel;
        Distributed_Sparse* d = make_alg(alg, &S, (int)cs.R, c, &kernel);
        if (mode == "dump") {
            if (argc < 6) die("dump needs <outprefix>");
            run_dump(cs, d, argv[5]);
        } else if (mode == "als") {
            if (argc < 8) die("als needs <outprefix> <steps> <cg_iters>");
            run_als(cs, d, argv[5], std::atoi(argv[6]), std::atoi(argv[7]));
        } else if (mode == "gat") {
            if (argc < 8) die("gat needs <outprefix> <alpha> <layers>");
            run_gat(cs, d, argv[5], std::atof(argv[6]), argv[7]);
        } else if (mode == "fp") {
            run_fp(d, alg);
        } else if (mode == "bench") {
            bool fused = argc > 5 ? std::atoi(argv[5]) != 0 : true;
            int trials = argc > 6 ? std::atoi(argv[6]) : 5;
            run_bench(cs, d, alg, fused, trials);
        } else if (mode == "sweep") {
            if (argc < 9) die("sweep needs <fused> <trials> <reps> <t1,t2,...>");
            std::vector<int> counts;
            for (char* tok = std::strtok(argv[8], ","); tok; tok = std::strtok(nullptr, ",")) counts.push_back(std::max(1, std::atoi(tok)));
            run_sweep(cs, d, alg, std::atoi(argv[5]) != 0, std::atoi(argv[6]), std::max(1, std::atoi(argv[7])), counts);
        } else {
            die("unknown mode " + mode);
        }
        MPI_Barrier(MPI_COMM_WORLD);
        delete d;  // frees the grid's communicators before MPI_Finalize (FlexibleGrid.hpp:96-103)
    }
    MPI_Finalize();
    return 0;
}
