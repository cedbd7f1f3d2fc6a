// The fingerprint check a user of the reference knows from scratch.cpp:26-76, as a TABLE of the three operations: for each of
// sddmmA / spmmA / spmmB fill A and B with the distribution-independent pattern value(row, col) = row * R + col (dummyInitialize,
// distributed_sparse.h:322-346), S = 1, apply the schedule's initial_shift, run the operation and sum the squared norm of its result
// over the ranks.  The three numbers do not depend on the algorithm, the process count or c, and equal what the reference prints
// for the same matrix.  Beyond the reference: with `all` the program ITSELF compares the algorithms' fingerprints and exits 1 when
// two of them differ by more than 1e-11 relative.
//
//     verify <matrix.mtx> <15d_fusion1|15d_fusion2|15d_sparse|25d_dense_replicate|25d_sparse_replicate|all> <R> <c>
//     verify er:<logM>:<edgeFactor> <algorithm|all> <R> <c>
//
// (scratch.cpp takes `file R c` and has the algorithm edited into its main(), scratch.cpp:95-122.)
#include "bench_common.hpp"

#include <array>
#include <cmath>
#include <functional>

namespace {

struct Operands {
    DenseMatrix A, B;
    VectorXd S, ST, result;
};

// one row per operation: label, the kernel mode its initial_shift takes, the call, and which object carries the result
struct Operation {
    const char* label;
    KernelMode mode;
    std::function<void(Distributed_Sparse&, Operands&)> run;
    std::function<double(const Operands&)> squared_norm;
};

const std::array<Operation, 3> kOperations = {{
    {"SDDMM", k_sddmmA, [](Distributed_Sparse& d, Operands& o) { d.sddmmA(o.A, o.B, o.S, o.result); },
     [](const Operands& o) { return o.result.squaredNorm(); }},
    {"SpMMA", k_spmmA, [](Distributed_Sparse& d, Operands& o) { d.spmmA(o.A, o.B, o.S); }, [](const Operands& o) { return o.A.squaredNorm(); }},
    {"SpMMB", k_spmmB, [](Distributed_Sparse& d, Operands& o) { d.spmmB(o.A, o.B, o.ST); }, [](const Operands& o) { return o.B.squaredNorm(); }},
}};

std::array<double, 3> fingerprints(Distributed_Sparse& d_ops) {
    hnh::World* world = hnh::current_world();
    Operands o{d_ops.like_A_matrix(0.0), d_ops.like_B_matrix(0.0), d_ops.like_S_values(1.0), d_ops.like_ST_values(1.0), d_ops.like_S_values(0.0)};
    std::array<double, 3> out{};
    for (size_t k = 0; k < kOperations.size(); k++) {
        const Operation& op = kOperations[k];
        d_ops.dummyInitialize(o.A, Amat);
        d_ops.dummyInitialize(o.B, Bmat);
        d_ops.initial_shift(&o.A, &o.B, op.mode);
        op.run(d_ops, o);
        out[k] = world->host_allreduce_sum(op.squared_norm(o));
        if (world->rank == 0) cout << setprecision(17) << op.label << " Fingerprint: " << out[k] << endl;
    }
    return out;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) {
        cerr << "usage: verify <matrix.mtx | er:logM:edgeFactor> <algorithm | all> R c" << endl;
        return 2;
    }
    hnh::World* world = make_world();
    hnh::set_current_world(world);
    const string source(argv[1]), which(argv[2]);
    const int R = atoi(argv[3]), c = atoi(argv[4]);
    bool agree = true;
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
        vector<std::array<double, 3>> seen;
        for (const string& name : (which == "all" ? all : vector<string>{which})) {
            unique_ptr<Distributed_Sparse> d_ops;
            if (name == "15d_fusion1") d_ops.reset(new Sparse15D_Dense_Shift(&S, R, c, 1, &local_ops));
            else if (name == "15d_fusion2") d_ops.reset(new Sparse15D_Dense_Shift(&S, R, c, 2, &local_ops));
            else if (name == "15d_sparse") d_ops.reset(new Sparse15D_Sparse_Shift(&S, R, c, &local_ops));
            else if (name == "25d_dense_replicate") d_ops.reset(new Sparse25D_Cannon_Dense(&S, R, c, &local_ops));
            else if (name == "25d_sparse_replicate") d_ops.reset(new Sparse25D_Cannon_Sparse(&S, R, c, &local_ops));
            else hnh::fatal("Error, unknown algorithm " + name);
            if (world->rank == 0) cout << "== " << name << " (R = " << R << ", c = " << c << ", " << world->size << " rank(s))" << endl;
            seen.push_back(fingerprints(*d_ops));
        }
        for (size_t a = 1; a < seen.size(); a++)
            for (size_t k = 0; k < 3; k++)
                if (std::fabs(seen[a][k] - seen[0][k]) > 1e-11 * std::fabs(seen[0][k])) agree = false;
        if (world->rank == 0 && seen.size() > 1)
            cout << (agree ? "all algorithms agree to 1e-11" : "MISMATCH: the algorithms' fingerprints differ by more than 1e-11") << endl;
    }
    world->sync_all();
    hnh::set_current_world(nullptr);
    delete world;
    return agree ? 0 : 1;
}
