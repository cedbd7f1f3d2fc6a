// Shared by the two drop-in drivers (bench_er.cpp, bench_file.cpp): the process bootstrap and the reference's
// benchmark_algorithm() (benchmark_dist.cpp:26-167) re-written against THIS repository's class headers.
//
// One process per GPU.  Without a launcher a driver runs on GPU 0 (p = 1).  Multi-GPU: start N processes with
// RANK / WORLD_SIZE / LOCAL_RANK set (torchrun, mpiexec -env, a shell loop) and either
//   HNH_ID_FILE=<path on a shared filesystem>          RCCL: rank 0 writes the unique id there, the others read it, or
//   HNH_TRANSPORT=ipc HNH_IPC_SESSION=<name>           the ipc-pull transport of one node (receivers copy out of their peers'
//                                                      mapped buffers; the ranks meet in a shared-memory session of that name);
//                                                      HNH_DEVICE=<ordinal> overrides LOCAL_RANK as the device (processes may share one).
// (hnh::world_from_environment(), world.hpp.  The reference uses MPI_Init; MPI is not needed here — and the reference's own mains
// compile unchanged against include/compat, whose MPI_Init is this bootstrap: tests/test_reference_mains_cpu.py.)
#pragma once
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <thread>

#include "als_conjugate_gradients.hpp"
#include "gat.hpp"
#include "cannon_dense_25d.hpp"
#include "cannon_sparse_25d.hpp"
#include "dense_shift_15d.hpp"
#include "sparse_shift_15d.hpp"

using namespace std;
using json = hnh::json;  // (the reference: `using json = nlohmann::json;`, benchmark_dist.cpp:22)

// the process bootstrap lives in the host library (world.hpp); the reference's own mains reach it through MPI_Init of include/compat/mpi.h
inline hnh::World* make_world() { return hnh::world_from_environment(); }

// benchmark_dist.cpp:26-167
inline void benchmark_algorithm(SpmatLocal* spmat, string algorithm_name, string output_file, bool fused, int R, int c, string app) {
    hnh::World* world = hnh::current_world();
    const int rank = world->rank;
    StandardKernel local_ops;
    Distributed_Sparse* d_ops = nullptr;
    if (algorithm_name == "15d_fusion1") d_ops = new Sparse15D_Dense_Shift(spmat, R, c, 1, &local_ops);
    else if (algorithm_name == "15d_sparse") d_ops = new Sparse15D_Sparse_Shift(spmat, R, c, &local_ops);
    else if (algorithm_name == "15d_fusion2") d_ops = new Sparse15D_Dense_Shift(spmat, R, c, 2, &local_ops);
    else if (algorithm_name == "25d_dense_replicate") d_ops = new Sparse25D_Cannon_Dense(spmat, R, c, &local_ops);
    else if (algorithm_name == "25d_sparse_replicate") d_ops = new Sparse25D_Cannon_Sparse(spmat, R, c, &local_ops);
    else hnh::fatal("Error, unknown algorithm " + algorithm_name);

    unique_ptr<Distributed_ALS> d_als;
    unique_ptr<GAT> gnn;
    vector<GATLayer> layers;
    if (app == "gat") {  // benchmark_dist.cpp:88-94: input features, features per head, heads
        layers.emplace_back(256, 256, 4);
        layers.emplace_back(1024, 256, 4);
        layers.emplace_back(1024, 256, 6);
        gnn.reset(new GAT(layers, d_ops));
    } else if (app == "als") {
        d_als.reset(new Distributed_ALS(d_ops, true));
    } else if (app != "vanilla") {
        hnh::fatal("Error, app must be vanilla, als or gat");
    }

    DenseMatrix A = d_ops->like_A_matrix(0.001), B = d_ops->like_B_matrix(0.001);
    VectorXd S = d_ops->like_S_values(1.0), sddmm_result = d_ops->like_S_values(0.0);
    if (rank == 0) cout << "Starting benchmark " << app << endl;

    // one untimed call: first-touch / lazily created communicators stay out of the measurement
    if (app == "vanilla") d_ops->fusedSpMM(A, B, S, sddmm_result, Amat);
    world->sync_all();
    world->barrier();

    d_ops->reset_performance_timers();
    my_timer_t t = start_clock();
    int num_trials = 0;
    double application_communication_time = 0.0;
    do {
        num_trials++;
        if (app == "vanilla") {
            if (fused) d_ops->fusedSpMM(A, B, S, sddmm_result, Amat);
            else {
                d_ops->sddmmA(A, B, S, sddmm_result);
                d_ops->spmmA(A, B, S);
            }
        } else if (app == "gat") {
            gnn->forwardPass();
        } else {
            d_als->application_communication_time = 0.0;
            d_als->run_cg(1);
            application_communication_time = d_als->application_communication_time;
        }
    } while (num_trials < 5);
    world->sync_all();  // GPU work is asynchronous: drain before stopping the clock
    world->barrier();
    const double elapsed = stop_clock_get_elapsed(t);
    const double ops = 2.0 * (double)spmat->dist_nnz * 2.0 * R * num_trials;  // benchmark_dist.cpp:147
    const double throughput = ops / elapsed / 1e9;
    json j_obj;  // the reference's record (benchmark_dist.cpp:144-162)
    j_obj["elapsed"] = elapsed;
    j_obj["overall_throughput"] = throughput;
    j_obj["fused"] = fused;
    j_obj["num_trials"] = num_trials;
    j_obj["alg_name"] = algorithm_name;
    j_obj["alg_info"] = d_ops->json_algorithm_info();
    j_obj["application_communication_time"] = application_communication_time;
    j_obj["perf_stats"] = d_ops->json_perf_statistics();
    if (rank == 0) {
        ofstream fout(output_file, ios_base::app);
        fout << j_obj.dump(4) << "," << endl;
        cout << algorithm_name << ": " << elapsed << " s for " << num_trials << " trials, " << throughput << " GFLOP/s = "
             << throughput * 1e9 / 4.0 << " nnz*R/s" << endl;
    }
    d_als.reset();
    gnn.reset();
    delete d_ops;
}

