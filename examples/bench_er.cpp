// Source-level drop-in demonstration: the reference's ER benchmark driver (bench_erdos_renyi.cpp:19-120 +
// benchmark_dist.cpp:26-167) re-written against THIS repository's class headers.  Same positional arguments,
// same algorithm names, same JSON keys appended to the output file:
//
//     bench_er <logM> <edgeFactor> <15d|25d|15d_fusion1|15d_fusion2|15d_sparse|25d_dense_replicate|25d_sparse_replicate>
//              <R> <c> <outfile> [fused|unfused] [vanilla|als|gat]
//
// Process bootstrap and benchmark_algorithm(): bench_common.hpp.
#include "bench_common.hpp"

int main(int argc, char** argv) {
    if (argc < 7) {
        cerr << "usage: bench_er logM edgeFactor algorithm R c outfile [fused|unfused] [vanilla|als|gat]" << endl;
        return 2;
    }
    hnh::World* world = make_world();
    hnh::set_current_world(world);
    const int logM = atoi(argv[1]), edgeFactor = atoi(argv[2]), R = atoi(argv[4]), c = atoi(argv[5]);
    const string algorithm_name(argv[3]), output_file(argv[6]);
    const bool fused = argc > 7 ? string(argv[7]) != "unfused" : true;
    const string app = argc > 8 ? argv[8] : "vanilla";
    {
        SpmatLocal S;
        S.loadTuples(false, logM, edgeFactor, "");
        if (algorithm_name == "15d") {  // bench_erdos_renyi.cpp:51-66
            benchmark_algorithm(&S, "15d_fusion1", output_file, true, R, c, app);
            benchmark_algorithm(&S, "15d_fusion2", output_file, true, R, c, app);
        } else if (algorithm_name == "25d") {  // bench_erdos_renyi.cpp:92-107
            benchmark_algorithm(&S, "25d_sparse_replicate", output_file, false, R, c, app);
            benchmark_algorithm(&S, "25d_dense_replicate", output_file, true, R, c, app);
        } else {
            benchmark_algorithm(&S, algorithm_name, output_file, fused, R, c, app);
        }
    }
    world->sync_all();
    hnh::set_current_world(nullptr);
    delete world;
    return 0;
}
