// RCCL (xGMI) transport entry points of include/hnh_kernels.h.
//
// The reference moves data with MPI: MPI_Sendrecv with MPI_ANY_SOURCE for the dense ring
// (distributed_sparse.h:351-361), 2-4 Isend/Irecv pairs for the sparse ring (SpmatLocal.hpp:200-259),
// Allgather / Reduce_scatter for replication (15D_dense_shift.hpp:194-195,240-242).  Here a ring step is
// ONE ncclGroup holding an explicit-peer send and recv, enqueued on the context's communication stream so
// that it overlaps the local kernel running on the compute stream; ordering comes from HIP events, not
// from world barriers.  xGMI is a full mesh of point-to-point links: a neighbour shift uses one link per
// direction, all-gather / reduce-scatter can use all seven.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstring>
#include "hnh_ctx.hpp"

namespace {
int check_nccl(hnh_ctx* ctx, ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return HNH_OK;
    return hnh::fail(ctx, HNH_ERR_DEVICE, std::string(what) + ": " + ncclGetErrorString(r));
}
}  // namespace

#define HNH_TRY_NCCL(ctx, expr)                          \
    do {                                                 \
        int _st = check_nccl((ctx), (expr), #expr);      \
        if (_st != HNH_OK) return _st;                   \
    } while (0)

static_assert(sizeof(ncclUniqueId) == HNH_UNIQUE_ID_BYTES, "unique id size");

extern "C" {

int hnh_comm_unique_id(void* id_host) {
    if (!id_host) return HNH_ERR_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return HNH_ERR_DEVICE;
    std::memcpy(id_host, &id, sizeof(id));
    return HNH_OK;
}

int hnh_comm_init(hnh_ctx* ctx, int nranks, int rank, const void* id_host, void** comm) {
    if (!ctx || !id_host || !comm || nranks <= 0 || rank < 0 || rank >= nranks) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id_host, sizeof(id));
    ncclComm_t c;
    HNH_TRY_NCCL(ctx, ncclCommInitRank(&c, nranks, id, rank));
    *comm = (void*)c;
    return HNH_OK;
}

int hnh_comm_split(hnh_ctx* ctx, void* comm, int color, int key, void** newcomm) {
    if (!ctx || !comm || !newcomm) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    ncclComm_t c;
    HNH_TRY_NCCL(ctx, ncclCommSplit((ncclComm_t)comm, color, key, &c, nullptr));
    *newcomm = (void*)c;
    return HNH_OK;
}

int hnh_comm_destroy(hnh_ctx* ctx, void* comm) {
    if (!ctx) return HNH_ERR_INVALID;
    if (!comm) return HNH_OK;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return check_nccl(ctx, ncclCommDestroy((ncclComm_t)comm), "ncclCommDestroy");
}

int hnh_comm_identity(hnh_ctx* ctx, void* comm, int* nranks, int* rank, int* device) {
    if (!ctx || !comm || !nranks || !rank || !device) return HNH_ERR_INVALID;
    HNH_TRY_NCCL(ctx, ncclCommCount((ncclComm_t)comm, nranks));
    HNH_TRY_NCCL(ctx, ncclCommUserRank((ncclComm_t)comm, rank));
    HNH_TRY_NCCL(ctx, ncclCommCuDevice((ncclComm_t)comm, device));
    return HNH_OK;
}

int hnh_comm_sendrecv(hnh_ctx* ctx, void* comm, const void* sendbuf, size_t sendbytes, int dst, void* recvbuf,
                      size_t recvbytes, int src, int stream) {
    HNH_ENTER(ctx, stream);
    if (!comm) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_comm_sendrecv: null communicator");
    hipStream_t st = ctx->streams[stream];
    HNH_TRY_NCCL(ctx, ncclGroupStart());
    // a call that fails inside the group still closes it: an open group would swallow every later call on this thread
    ncclResult_t r = ncclSuccess;
    const char* what = "ncclSend";
    if (sendbytes) r = ncclSend(sendbuf, sendbytes, ncclInt8, dst, (ncclComm_t)comm, st);
    if (r == ncclSuccess && recvbytes) {
        what = "ncclRecv";
        r = ncclRecv(recvbuf, recvbytes, ncclInt8, src, (ncclComm_t)comm, st);
    }
    const ncclResult_t rend = ncclGroupEnd();
    if (r != ncclSuccess) return check_nccl(ctx, r, what);
    return check_nccl(ctx, rend, "ncclGroupEnd");
}

int hnh_comm_group_begin(hnh_ctx* ctx) {
    if (!ctx) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return check_nccl(ctx, ncclGroupStart(), "ncclGroupStart");
}

int hnh_comm_group_end(hnh_ctx* ctx) {
    if (!ctx) return HNH_ERR_INVALID;
    HNH_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return check_nccl(ctx, ncclGroupEnd(), "ncclGroupEnd");
}

int hnh_comm_allgather(hnh_ctx* ctx, void* comm, const void* sendbuf, void* recvbuf, size_t bytes_per_rank, int stream) {
    HNH_ENTER(ctx, stream);
    if (!comm) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_comm_allgather: null communicator");
    return check_nccl(ctx, ncclAllGather(sendbuf, recvbuf, bytes_per_rank, ncclInt8, (ncclComm_t)comm, ctx->streams[stream]),
                      "ncclAllGather");
}

int hnh_comm_reduce_scatter_f64(hnh_ctx* ctx, void* comm, const double* sendbuf, double* recvbuf, size_t count_per_rank,
                                int stream) {
    HNH_ENTER(ctx, stream);
    if (!comm) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_comm_reduce_scatter_f64: null communicator");
    return check_nccl(ctx, ncclReduceScatter(sendbuf, recvbuf, count_per_rank, ncclDouble, ncclSum, (ncclComm_t)comm,
                                             ctx->streams[stream]),
                      "ncclReduceScatter");
}

int hnh_comm_allreduce_f64(hnh_ctx* ctx, void* comm, const double* sendbuf, double* recvbuf, size_t count, int stream) {
    HNH_ENTER(ctx, stream);
    if (!comm) return hnh::fail(ctx, HNH_ERR_INVALID, "hnh_comm_allreduce_f64: null communicator");
    return check_nccl(ctx, ncclAllReduce(sendbuf, recvbuf, count, ncclDouble, ncclSum, (ncclComm_t)comm, ctx->streams[stream]),
                      "ncclAllReduce");
}

}  // extern "C"
