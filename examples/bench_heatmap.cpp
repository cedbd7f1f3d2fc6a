// The reference's embedding-width sweep (bench_heatmap.cpp:19-109) against THIS repository's class headers: one Erdos-Renyi
// matrix, every width of {64, 128, ..., 448} (bench_heatmap.cpp:33), the algorithms of one family per width, one JSON record per
// run appended to the output file.
//
//     bench_heatmap <logM> <edgeFactor> <15d|25d> <c> <outfile> [R,R,...]
//
// "15d" = 15d_fusion1, 15d_fusion2, 15d_sparse (all fused); "25d" = 25d_sparse_replicate (unfused), 25d_dense_replicate (fused).
// The optional last argument replaces the reference's width list (tests use a short one).
#include "bench_common.hpp"

int main(int argc, char** argv) {
    if (argc < 6) {
        cerr << "usage: bench_heatmap logM edgeFactor 15d|25d c outfile [R,R,...]" << endl;
        return 2;
    }
    hnh::World* world = make_world();
    hnh::set_current_world(world);
    const int logM = atoi(argv[1]), edgeFactor = atoi(argv[2]), c = atoi(argv[4]);
    const string family(argv[3]), output_file(argv[5]), app = "vanilla";
    vector<int> rvalues = {64, 128, 192, 256, 320, 384, 448};
    if (argc > 6) {
        rvalues.clear();
        stringstream ss(argv[6]);
        for (string tok; getline(ss, tok, ',');)
            if (atoi(tok.c_str()) > 0) rvalues.push_back(atoi(tok.c_str()));
    }
    if (family != "15d" && family != "25d") hnh::fatal("Error, the algorithm family is 15d or 25d!");
    {
        SpmatLocal S;
        S.loadTuples(false, logM, edgeFactor, "");
        for (int R : rvalues) {
            if (family == "15d") {  // bench_heatmap.cpp:38-77
                benchmark_algorithm(&S, "15d_fusion1", output_file, true, R, c, app);
                benchmark_algorithm(&S, "15d_fusion2", output_file, true, R, c, app);
                benchmark_algorithm(&S, "15d_sparse", output_file, true, R, c, app);
            } else {  // bench_heatmap.cpp:79-100
                benchmark_algorithm(&S, "25d_sparse_replicate", output_file, false, R, c, app);
                benchmark_algorithm(&S, "25d_dense_replicate", output_file, true, R, c, app);
            }
        }
    }
    world->sync_all();
    hnh::set_current_world(nullptr);
    delete world;
    return 0;
}
