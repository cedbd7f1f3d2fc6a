// The reference's only correctness check — scratch.cpp:26-76 `verify_operation` — against THIS repository's class headers, call
// for call: fill A and B with the distribution-independent pattern value(row, col) = row * R + col (dummyInitialize,
// distributed_sparse.h:322-346), S = 1, run sddmmA / spmmA / spmmB (each after initial_shift) and print the globally summed
// squared norms.  The three numbers must be the same for every algorithm, every process count and every c — and the same as
// the reference prints for the same matrix, which is how a user of the reference checks a build of this library against it.
//
//     verify <matrix.mtx> <15d_fusion1|15d_fusion2|15d_sparse|25d_dense_replicate|25d_sparse_replicate|all> <R> <c>
//     verify er:<logM>:<edgeFactor> <algorithm|all> <R> <c>
//
// (scratch.cpp takes `file R c` and has the algorithm edited into its main(), scratch.cpp:95-122.)
#include "bench_common.hpp"

static void verify_operation(Distributed_Sparse* d_ops) {  // scratch.cpp:26-76
    hnh::World* world = hnh::current_world();
    DenseMatrix A = d_ops->like_A_matrix(0.0);
    DenseMatrix B = d_ops->like_B_matrix(0.0);
    VectorXd S = d_ops->like_S_values(1.0);
    VectorXd ST = d_ops->like_ST_values(1.0);
    VectorXd result = d_ops->like_S_values(0.0);

    d_ops->dummyInitialize(A, Amat);
    d_ops->dummyInitialize(B, Bmat);
    d_ops->initial_shift(&A, &B, k_sddmmA);
    d_ops->sddmmA(A, B, S, result);
    const double sddmm_fingerprint = world->host_allreduce_sum(result.squaredNorm());

    d_ops->dummyInitialize(A, Amat);
    d_ops->dummyInitialize(B, Bmat);
    d_ops->initial_shift(&A, &B, k_spmmA);
    d_ops->spmmA(A, B, S);
    const double spmmA_fingerprint = world->host_allreduce_sum(A.squaredNorm());

    d_ops->dummyInitialize(A, Amat);
    d_ops->dummyInitialize(B, Bmat);
    d_ops->initial_shift(&A, &B, k_spmmB);
    d_ops->spmmB(A, B, ST);
    const double spmmB_fingerprint = world->host_allreduce_sum(B.squaredNorm());

    if (world->rank == 0) {
        cout << setprecision(17);
        cout << "SDDMM Fingerprint: " << sddmm_fingerprint << endl;
        cout << "SpMMA Fingerprint: " << spmmA_fingerprint << endl;
        cout << "SpMMB Fingerprint: " << spmmB_fingerprint << endl;
    }
}

int main(int argc, char** argv) {
    if (argc < 5) {
        cerr << "usage: verify <matrix.mtx | er:logM:edgeFactor> <algorithm | all> R c" << endl;
        return 2;
    }
    hnh::World* world = make_world();
    hnh::set_current_world(world);
    const string source(argv[1]), which(argv[2]);
    const int R = atoi(argv[3]), c = atoi(argv[4]);
    {
        SpmatLocal S;
        if (source.rfind("er:", 0) == 0) {
            int logM = 0, edgeFactor = 0;
            if (sscanf(source.c_str(), "er:%d:%d", &logM, &edgeFactor) != 2) hnh::fatal("Error, expected er:<logM>:<edgeFactor>!");
            S.loadTuples(false, logM, edgeFactor, "");
        } else {
            S.loadTuples(true, -1, -1, source);
        }
        StandardKernel local_ops;
        const vector<string> all = {"15d_fusion1", "15d_fusion2", "15d_sparse", "25d_dense_replicate", "25d_sparse_replicate"};
        for (const string& name : (which == "all" ? all : vector<string>{which})) {
            unique_ptr<Distributed_Sparse> d_ops;
            if (name == "15d_fusion1") d_ops.reset(new Sparse15D_Dense_Shift(&S, R, c, 1, &local_ops));
            else if (name == "15d_fusion2") d_ops.reset(new Sparse15D_Dense_Shift(&S, R, c, 2, &local_ops));
            else if (name == "15d_sparse") d_ops.reset(new Sparse15D_Sparse_Shift(&S, R, c, &local_ops));
            else if (name == "25d_dense_replicate") d_ops.reset(new Sparse25D_Cannon_Dense(&S, R, c, &local_ops));
            else if (name == "25d_sparse_replicate") d_ops.reset(new Sparse25D_Cannon_Sparse(&S, R, c, &local_ops));
            else hnh::fatal("Error, unknown algorithm " + name);
            if (world->rank == 0) cout << "== " << name << " (R = " << R << ", c = " << c << ", " << world->size << " rank(s))" << endl;
            verify_operation(d_ops.get());
        }
    }
    world->sync_all();
    hnh::set_current_world(nullptr);
    delete world;
    return 0;
}
