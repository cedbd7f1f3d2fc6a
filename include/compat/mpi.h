// Stand-in for the few MPI calls the reference's APPLICATION code makes on MPI_COMM_WORLD (benchmark_dist.cpp:35-36,142;
// bench_erdos_renyi.cpp): the process group of this engine is hnh::World (one process per GPU, RCCL / ipc-pull underneath), and
// `MPI_COMM_WORLD` is the thread's current world (hnh::current_world(), world.hpp).  Code that really links MPI does not
// include this header (its own <mpi.h> comes first on the include path).
#pragma once
#include "../../distributed_sddmm_amd/csrc/host/world.hpp"
typedef int MPI_Comm;
#define MPI_COMM_WORLD 0
#define MPI_SUCCESS 0
inline int MPI_Comm_rank(MPI_Comm, int* rank) {
    *rank = hnh::current_world()->rank;
    return MPI_SUCCESS;
}
inline int MPI_Comm_size(MPI_Comm, int* size) {
    *size = hnh::current_world()->size;
    return MPI_SUCCESS;
}
// the reference stops its clock after a barrier (benchmark_dist.cpp:142): GPU work is asynchronous, so drain the streams first
inline int MPI_Barrier(MPI_Comm) {
    hnh::current_world()->sync_all();
    hnh::current_world()->barrier();
    return MPI_SUCCESS;
}
