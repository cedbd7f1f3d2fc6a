// Stand-in for the few MPI calls the reference's APPLICATION code makes on MPI_COMM_WORLD (benchmark_dist.cpp:35-36,142; the mains
// of bench_erdos_renyi.cpp, bench_file.cpp, bench_heatmap.cpp, scratch.cpp): the process group of this engine is hnh::World (one
// process per GPU, RCCL / ipc-pull underneath), and `MPI_COMM_WORLD` is the thread's current world (hnh::current_world(),
// world.hpp).  With it the reference's four mains compile UNCHANGED (tests/test_reference_mains_cpu.py).  Code that really links
// MPI does not include this header (its own <mpi.h> comes first on the include path).
#pragma once
#include <chrono>
#include <cstring>
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

// MPI_Init = the process bootstrap (world.hpp: world_from_environment — RANK / WORLD_SIZE / LOCAL_RANK from the launcher, RCCL through
// HNH_ID_FILE or ipc-pull through HNH_TRANSPORT=ipc HNH_IPC_SESSION=...).  A world the embedder made current beforehand is kept.
inline int MPI_Init(int*, char***) {
    if (hnh::current_world_or_null() == nullptr) hnh::process_world();
    return MPI_SUCCESS;
}
// The reference's mains still own device-backed objects when they call MPI_Finalize (their SpmatLocal lives until main returns):
// drain and meet the other ranks here, the process world itself goes at exit (hnh::process_world()).
inline int MPI_Finalize() {
    if (hnh::World* w = hnh::current_world_or_null()) {
        w->sync_all();
        w->barrier();
    }
    return MPI_SUCCESS;
}
inline double MPI_Wtime() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// host-side reductions of the reference's checks (scratch.cpp:46-68: MPI_Allreduce(MPI_IN_PLACE, &x, 1, MPI_DOUBLE, MPI_SUM, ...))
typedef int MPI_Datatype;
typedef int MPI_Op;
#define MPI_DOUBLE 1
#define MPI_SUM 1
#define MPI_IN_PLACE ((void*)-1)
inline int MPI_Allreduce(const void* sendbuf, void* recvbuf, int count, MPI_Datatype type, MPI_Op op, MPI_Comm) {
    if (type != MPI_DOUBLE || op != MPI_SUM) hnh::fatal("Error, the MPI stand-in reduces MPI_DOUBLE with MPI_SUM only");
    double* out = static_cast<double*>(recvbuf);
    if (sendbuf != MPI_IN_PLACE && sendbuf != recvbuf) std::memcpy(out, sendbuf, (size_t)count * sizeof(double));
    hnh::current_world()->host_allreduce_sum(out, (size_t)count);
    return MPI_SUCCESS;
}
