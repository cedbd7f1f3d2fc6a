// Source-level drop-in demonstration: the reference's file benchmark driver (bench_file.cpp:19-105) re-written against
// THIS repository's class headers.  Same positional arguments:
//
//     bench_file <matrix.mtx> <15d|25d|15d_fusion1|15d_fusion2|15d_sparse|25d_dense_replicate|25d_sparse_replicate>
//                <R> <c> <outfile> <vanilla|als|gat> [fused|unfused]
//
// "15d" and "25d" select what the reference's main() has enabled (15d_sparse / 25d_dense_replicate, unfused,
// bench_file.cpp:32-97); a schedule may also be named directly.  The matrix is a MatrixMarket coordinate file (general or
// symmetric; pattern, integer or real); duplicate coordinates keep their maximum like the reference's
// ParallelReadMM(..., maximum<double>()) (SpmatLocal.hpp:485-498) — parsed on the host cores, ordered and merged on the GPU.
// HNH_PERMUTE_SEED=<seed> applies the seeded random vertex relabelling (the reference ships random_permute.cpp:42-50 as a
// separate pre-processing program for load balance on real graphs).
// Process bootstrap and benchmark_algorithm(): bench_common.hpp.
#include "bench_common.hpp"

int main(int argc, char** argv) {
    if (argc < 7) {
        cerr << "usage: bench_file matrix.mtx algorithm R c outfile vanilla|als|gat [fused|unfused]" << endl;
        return 2;
    }
    hnh::World* world = make_world();
    hnh::set_current_world(world);
    const string fname(argv[1]), algorithm_name(argv[2]), output_file(argv[5]), app(argv[6]);
    const int R = atoi(argv[3]), c = atoi(argv[4]);
    const bool fused = argc > 7 ? string(argv[7]) != "unfused" : false;
    {
        SpmatLocal S;
        S.loadTuples(true, -1, -1, fname);
        if (algorithm_name == "15d") benchmark_algorithm(&S, "15d_sparse", output_file, false, R, c, app);             // bench_file.cpp:41-46
        else if (algorithm_name == "25d") benchmark_algorithm(&S, "25d_dense_replicate", output_file, false, R, c, app);  // bench_file.cpp:89-94
        else benchmark_algorithm(&S, algorithm_name, output_file, fused, R, c, app);
    }
    world->sync_all();
    hnh::set_current_world(nullptr);
    delete world;
    return 0;
}
