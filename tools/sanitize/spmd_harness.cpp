// sanitizer harness: p logical ranks (threads, loopback transport) run each schedule's ops over the CPU test double
#include <iostream>
#include <memory>
#include <thread>
#include <vector>
#include "cannon_dense_25d.hpp"
#include "cannon_sparse_25d.hpp"
#include "dense_shift_15d.hpp"
#include "sparse_shift_15d.hpp"
#include "als_conjugate_gradients.hpp"
#include "gat.hpp"
using namespace std;
static Distributed_Sparse* make(const string& alg, SpmatLocal* S, int R, int c, KernelImplementation* k) {
    if (alg == "15d_fusion1") return new Sparse15D_Dense_Shift(S, R, c, 1, k);
    if (alg == "15d_fusion2") return new Sparse15D_Dense_Shift(S, R, c, 2, k);
    if (alg == "15d_sparse") return new Sparse15D_Sparse_Shift(S, R, c, k);
    if (alg == "25d_dense_replicate") return new Sparse25D_Cannon_Dense(S, R, c, k);
    return new Sparse25D_Cannon_Sparse(S, R, c, k);
}
int main(int argc, char** argv) {
    hnh::Backend* be = hnh::load_backend(argv[1]);
    const int p = atoi(argv[2]), c = atoi(argv[3]);
    const string alg = argv[4];
    auto group = hnh::make_thread_group(p);
    vector<thread> ts;
    vector<double> sums(p, 0.0);
    for (int r = 0; r < p; r++)
        ts.emplace_back([&, r] {
            hnh::ThreadWorld w(group, r, be, 0);
            hnh::set_current_world(&w);
            {
                SpmatLocal S;
                S.loadTuples(false, 8, 6, "");
                StandardKernel k;
                unique_ptr<Distributed_Sparse> d(make(alg, &S, 16, c, &k));
                DenseMatrix A = d->like_A_matrix(0.001), B = d->like_B_matrix(0.002);
                VectorXd Sv = d->like_S_values(1.0), res = d->like_S_values(0.0);
                VectorXd STv = d->like_ST_values(1.0), resT = d->like_ST_values(0.0);
                for (int it = 0; it < 2; it++) {
                    d->initial_shift(&A, &B, k_sddmmA); d->fusedSpMM(A, B, Sv, res, Amat); d->de_shift(&A, &B, k_sddmmA);
                    d->initial_shift(&A, &B, k_sddmmB); d->sddmmB(A, B, STv, resT); d->de_shift(&A, &B, k_sddmmB);
                    d->initial_shift(&A, &B, k_spmmB); d->spmmB(A, B, STv); d->de_shift(&A, &B, k_spmmB);
                }
                for (int it = 0; it < 2; it++) {
                    d->initial_shift(&A, &B, k_sddmmA); d->sddmmA(A, B, Sv, res); d->de_shift(&A, &B, k_sddmmA);
                    d->initial_shift(&A, &B, k_spmmA); d->spmmA(A, B, Sv); d->de_shift(&A, &B, k_spmmA);
                    d->initial_shift(&A, &B, k_sddmmB); d->fusedSpMM(A, B, STv, resT, Bmat); d->de_shift(&A, &B, k_sddmmB);
                }
                if (alg == "15d_fusion1" || alg == "15d_fusion2") {  // (the schedules that do not split R: gat.hpp's precondition)
                    vector<GATLayer> layers = {GATLayer(16, 8, 2), GATLayer(16, 8, 3)};
                    GAT gat(layers, d.get());
                    for (auto& l : gat.layers)
                        for (auto& W : l.wMats) W = DenseMatrix::Constant(W.rows(), W.cols(), 0.01);
                    gat.buffers[0] = DenseMatrix::Constant(gat.buffers[0].rows(), gat.buffers[0].cols(), 0.5);
                    gat.forwardPass();
                    gat.forwardPass();
                    w.sync_all();
                    d->setRValue(16);
                }
                Distributed_ALS als(d.get(), true);
                als.initializeEmbeddings();
                als.cg_optimizer(Amat, 2);
                als.cg_optimizer(Bmat, 2);
                sums[r] = als.computeResidual();
                w.sync_all();
            }
            hnh::set_current_world(nullptr);
        });
    for (auto& t : ts) t.join();
    cout << alg << " p=" << p << " c=" << c << " residual " << sums[0] << endl;
    return 0;
}
