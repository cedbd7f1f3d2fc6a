// Process-group abstraction that replaces MPI in the reference.
//
// One `World` per rank.  A rank is either a process that owns a GPU (RcclWorld: RCCL over xGMI, one
// process per GPU, the production path), a host thread sharing one GPU with its peers (ThreadWorld:
// "loopback", device-to-device copies on the communication stream — lets every shift schedule be
// verified on the single GPU a test box has), or a process whose transport is supplied through
// callbacks (CallbackWorld: used with torch.distributed/gloo in the CPU tests).
//
// Device-buffer operations are STREAM-ORDERED like RCCL: they are enqueued on the selected stream of the
// rank's hnh_ctx; when the work previously enqueued on that stream has completed, `recvbuf` holds the
// data and `sendbuf` may be reused.  Host operations block.
//
// `hnh::current_world()` is the thread-local analogue of MPI_COMM_WORLD: the reference's classes call
// MPI_Comm_rank(MPI_COMM_WORLD, ...) in their constructors (distributed_sparse.h:81-82); ours read the
// current world instead, which keeps the constructor signatures identical.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>
#include "backend.hpp"
#include "hnh_dist.h"
#include "common.hpp"

namespace hnh {

// An ordered subset of world ranks (MPI_Comm analogue).  `me` is this rank's index, -1 if not a member.
struct Comm {
    std::vector<int> ranks;
    int me = -1;
    int color = 0, key = 0;  // what it was split with (MPI_Comm_split arguments)
    bool is_world = false;
    int size() const { return (int)ranks.size(); }
    int rank() const { return me; }
};

class World {
public:
    int rank = 0, size = 1;
    Backend* be = nullptr;
    hnh_ctx* ctx = nullptr;
    int device = 0;
    bool timing_sync = false;  // when true perf counters synchronise the streams first (reference-like attribution)
    // running hash over the world-wide (color, key) tables of every split() so far, and their number: identical on all
    // ranks iff they created their communicators in the same order (checked by bench.py's multi-GPU preflight)
    uint64_t split_signature = 1469598103934665603ULL;
    int split_count = 0;

    World();
    virtual ~World();
    World(const World&) = delete;
    World& operator=(const World&) = delete;
    virtual const char* kind() const = 0;
    // Called by hnh::fatal() on the failing thread's current world before it exits or throws: a transport whose peers would
    // otherwise wait for this rank until their time limit tells them now (IpcWorld: the session's `failed` word).
    virtual void note_failure() noexcept {}

    // Measurement entry point (an addition): SOLO REPLAY.  While on, this rank runs its own side of every collective call ALONE: each
    // message it would receive is replaced by a device-to-device copy of the message it would send (same bytes written, same stream,
    // same events), host collectives return its own contribution for every rank, barriers return at once.  The results of such calls
    // are meaningless; their kernel sequence, launch pattern, event protocol and the HBM side of the exchange are those of the rank
    // in a real p-rank call — with the GPU to itself (bench.py's "rank share" entries).  Loopback transport only.
    virtual void set_solo(bool on);
    bool solo() const { return solo_; }

    // Where this rank runs: pid, device ordinal, the GPU's PCI bus id and (RCCL) the communicator's own view; identities() gathers
    // every rank's record (collective).  include/hnh_dist.h: hnh_rank_identity.
    virtual hnh_rank_identity identity();
    std::vector<hnh_rank_identity> identities();

    // ---- communicators
    Comm world_comm();
    virtual Comm split(int color, int key);  // MPI_Comm_split semantics (FlexibleGrid.hpp:80-88)
    virtual void free_comm(Comm& c);

    // ---- device buffers, stream-ordered.  dst / src are indices INTO `comm`.
    virtual void sendrecv(const Comm& comm, const void* sendbuf, size_t sendbytes, int dst, void* recvbuf,
                          size_t recvbytes, int src, int stream) = 0;
    // Transfers enqueued between group_begin() and group_end() may progress concurrently (one RCCL group:
    // send/recv pairs towards different peers use different xGMI links).  No-ops for the other transports.
    virtual void group_begin() {}
    virtual void group_end() {}
    virtual void allgather(const Comm& comm, const void* sendbuf, void* recvbuf, size_t bytes_per_rank, int stream);
    virtual void reduce_scatter_f64(const Comm& comm, const double* sendbuf, double* recvbuf, size_t count, int stream);
    // in-place sum over the members of `comm` (MPI_Allreduce(MPI_IN_PLACE), als_conjugate_gradients.cpp:31-36)
    virtual void allreduce_f64(const Comm& comm, double* buf, size_t count, int stream);
    // variable counts (25D_cannon_sparse.hpp:224-233,294-300); counts / displs in elements, per comm index
    virtual void allgatherv_f64(const Comm& comm, const double* sendbuf, size_t sendcount, double* recvbuf,
                                const std::vector<int>& counts, const std::vector<int>& displs, int stream);
    virtual void reduce_scatter_v_f64(const Comm& comm, const double* sendbuf, double* recvbuf,
                                      const std::vector<int>& counts, int stream);

    // ---- host data, blocking, collective over the whole world unless a Comm is given
    virtual void barrier() = 0;
    virtual void host_allgather(const void* send, void* recv, size_t bytes_per_rank) = 0;
    virtual void host_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl,
                                void* recv, const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) = 0;
    virtual void host_bcast(const Comm& comm, int root, void* buf, size_t bytes);
    // The same two exchanges for DEVICE buffers (setup pipeline: tuples never visit the host): pairwise rounds of
    // sendrecv over the world communicator on `stream`, drained before returning.
    void device_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl, void* recv,
                          const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl, int stream);
    void device_bcast(const Comm& comm, int root, void* buf, size_t bytes, int stream);
    void host_allgather_comm(const Comm& comm, const void* send, void* recv, size_t bytes_per_rank);
    double host_allreduce_sum(double v);
    void host_allreduce_sum(double* v, size_t n);

    // ---- conveniences over the backend (status -> fatal)
    void check(int status, const char* what) const;
    void* dmalloc(size_t bytes);
    void dfree(void* p);
    void copy(void* dst, const void* src, size_t bytes, int kind, int stream);
    void memset0(void* dst, size_t bytes, int stream);
    void sync(int stream);
    void sync_all();
    void sync_all_nothrow() noexcept;  // for destructors: reports a device error instead of throwing
    void* event_create();
    void event_destroy(void* e);
    void event_record(void* e, int stream);
    void event_wait(void* e, int stream);
    void event_sync(void* e);   // the host waits
    bool event_done(void* e);   // the host asks (never blocks)
#ifdef HNH_MEASUREMENT_AIDS
    void delay_us(double us, int stream);  // holds the stream without memory traffic (paced stand-in of a transfer)
#endif
    // scratch that persists across calls (grown on demand), one per slot
    void* scratch(int slot, size_t bytes);

protected:
    bool solo_ = false;
    // a device block is about to be given back to the driver (transports that exported it forget the handle)
    virtual void on_device_release(void* p) { (void)p; }
    void host_alltoallv_staged(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl, void* recv,
                               const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl);
    void init_device(Backend* backend, int device_ordinal);
    void destroy_device();
    std::vector<std::pair<void*, size_t>> scratch_;

    // Caching allocator.  hipMalloc / hipFree synchronise the device, and the reference's call pattern creates
    // and drops temporaries on every call (like_S_values in computeQueries, als_conjugate_gradients.cpp:276-277).
    // Freed blocks are parked with one event per stream; a block handed out again makes each stream wait for
    // the other stream's last use, so no host synchronisation is needed.  Bounded by HNH_POOL_LIMIT_GB (default 16).
    struct Parked {
        void* ptr;
        void* ev[2];
    };
    std::multimap<size_t, Parked> pool_;
    std::unordered_map<void*, size_t> live_;
    size_t pooled_bytes_ = 0, pool_limit_ = (size_t)16 << 30;
    void drain_pool();
};

World* current_world();
World* current_world_or_null();
void set_current_world(World* w);

// Process bootstrap of a stand-alone driver — what MPI_Init is to the reference's mains (bench_erdos_renyi.cpp:20, bench_file.cpp:20,
// scratch.cpp:79).  One process per GPU.  Rank, world size and local rank come from whichever launcher started the processes:
// RANK / WORLD_SIZE / LOCAL_RANK (torchrun, a shell loop), PMI_RANK / PMI_SIZE / MPI_LOCALRANKID (MPICH's mpiexec — the reference's own
// launch line `mpiexec -n 8 ./bench_erdos_renyi ...` works as typed; the launcher only starts processes, no MPI library is linked) or
// OMPI_COMM_WORLD_* (Open MPI's).
//   one rank                                             SingleWorld on device HNH_DEVICE (default: the local rank)
//   several, default or HNH_TRANSPORT=rccl               RcclWorld: rank 0 hands the RCCL unique id to the others through HNH_ID_FILE (a path
//                                                        every rank sees) or, without it, a file in /dev/shm named after the launch
//   several, HNH_TRANSPORT=ipc                           IpcWorld (ipc-pull): the ranks of one node meet in a shared-memory session —
//                                                        HNH_IPC_SESSION, or a name derived from the launch
//   HNH_DEVICE=<ordinal>                                 overrides the local rank as the device (processes may share one)
// "Named after the launch" = HNH_JOB_TOKEN, or the parent process id (the launcher's agent / proxy / shell, common to the ranks of a
// one-node launch) plus MASTER_PORT when set.  Loads the kernel library that sits next to this one (exits loudly if it or the GPU is
// missing).  The caller owns the world.
World* world_from_environment();
// The same, once per process and owned by the library: created on the first call, made the calling thread's current world, and torn
// down by an exit handler — after main() has returned, because the reference's mains destroy their SpmatLocal and operators AFTER
// MPI_Finalize (bench_erdos_renyi.cpp:119-120, scratch.cpp:148) and those objects hand their device memory back through the world.
// include/compat/mpi.h maps MPI_Init onto this.
World* process_world();

// ---- p = 1
class SingleWorld : public World {
public:
    SingleWorld(Backend* backend, int device_ordinal);
    ~SingleWorld() override;
    const char* kind() const override { return "single"; }
    void sendrecv(const Comm&, const void*, size_t, int, void*, size_t, int, int) override;
    void barrier() override {}
    void host_allgather(const void* send, void* recv, size_t bytes) override;
    void host_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl, void* recv,
                        const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) override;
};

// ---- N logical ranks = N host threads on one device ("loopback" transport)
struct ThreadGroup;
std::shared_ptr<ThreadGroup> make_thread_group(int nranks);

class ThreadWorld : public World {
public:
    ThreadWorld(std::shared_ptr<ThreadGroup> group, int rank_in_group, Backend* backend, int device_ordinal);
    ~ThreadWorld() override;
    const char* kind() const override { return "thread-loopback"; }
    void set_solo(bool on) override { solo_ = on; }
#ifdef HNH_MEASUREMENT_AIDS
    // (measurement build only) a group's paced messages travel together: see sendrecv
    void group_begin() override;
    void group_end() override;
    bool in_group_ = false;
    double group_us_ = 0.0;
    int group_stream_ = 0;
#endif
    void sendrecv(const Comm& comm, const void* sendbuf, size_t sendbytes, int dst, void* recvbuf, size_t recvbytes,
                  int src, int stream) override;
    void barrier() override;
    void host_allgather(const void* send, void* recv, size_t bytes) override;
    void host_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl, void* recv,
                        const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) override;

private:
    std::shared_ptr<ThreadGroup> g_;
    std::vector<const void*> publish(const void* mine);  // all ranks' pointers, valid until release()
    void release();
};

// ---- one process per GPU, RCCL over xGMI
class RcclWorld : public World {
public:
    RcclWorld(int rank, int nranks, Backend* backend, int device_ordinal, const void* unique_id);
    ~RcclWorld() override;
    const char* kind() const override { return "rccl"; }
    hnh_rank_identity identity() override;
    void group_begin() override;
    void group_end() override;
    void sendrecv(const Comm& comm, const void* sendbuf, size_t sendbytes, int dst, void* recvbuf, size_t recvbytes,
                  int src, int stream) override;
    void allgather(const Comm& comm, const void* sendbuf, void* recvbuf, size_t bytes_per_rank, int stream) override;
    void reduce_scatter_f64(const Comm& comm, const double* sendbuf, double* recvbuf, size_t count, int stream) override;
    void allreduce_f64(const Comm& comm, double* buf, size_t count, int stream) override;
    void barrier() override;
    void host_allgather(const void* send, void* recv, size_t bytes) override;
    void host_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl, void* recv,
                        const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) override;

private:
    void* comm_ = nullptr;  // world communicator (the only RCCL communicator: see world.cpp)
};

// ---- one process per GPU of ONE node, no RCCL: the receiver PULLS out of the sender's mapped buffer (include/hnh_kernels.h, "ipc").
// Control plane = a POSIX shared-memory segment all ranks attach to (`session` names it; the launcher hands every rank the same
// string): barrier, host all-gather, one mailbox per ordered pair of ranks carrying (memory handle, offset, bytes) of each
// message, and the flag words the ranks' STREAMS order themselves with.  A send/recv pair is
//     sender  : [write ready(me->dst) = s]  ...  [wait done(me->dst) >= s]
//     receiver: [wait ready(src->me) >= s]  [copy peer -> local]  [write done(src->me) = s]
// all enqueued on the stream the call names; the host only exchanges the message descriptors (posted at enqueue time, so a
// rank waits for its peer's HOST to reach the matching call, never for its device).  A group (group_begin .. group_end) issues
// all its ready flags first and pulls all its sources together: copy engines on forked streams, or one gather-copy kernel
// (HNH_IPC_PULL=engine|kernel).  Works between processes that share ONE GPU as well, which is how the tests run it.
struct IpcShared;
class IpcWorld : public World {
public:
    IpcWorld(int rank, int nranks, Backend* backend, int device_ordinal, const std::string& session);
    ~IpcWorld() override;
    const char* kind() const override { return "ipc-pull"; }
    void note_failure() noexcept override;
    void group_begin() override;
    void group_end() override;
    void sendrecv(const Comm& comm, const void* sendbuf, size_t sendbytes, int dst, void* recvbuf, size_t recvbytes, int src,
                  int stream) override;
    void barrier() override;
    void host_allgather(const void* send, void* recv, size_t bytes) override;
    void host_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl, void* recv,
                        const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) override;

protected:
    void on_device_release(void* p) override;

private:
    struct Op {
        int dst, src;  // world ranks
        const void* sendbuf;
        size_t sendbytes;
        void* recvbuf;
        size_t recvbytes;
        int stream;
    };
    struct Exported {
        size_t bytes;
        unsigned char handle[HNH_IPC_HANDLE_BYTES];
    };
    IpcShared* sh_ = nullptr;
    void* flags_dev_ = nullptr;
    std::string shm_name_;
    int group_depth_ = 0;
    std::vector<Op> pending_;
    std::vector<uint64_t> msg_out_, msg_in_;       // mailbox sequence per peer
    std::vector<uint64_t> flag_out_[2], flag_in_[2];  // ready / done sequence per stream and peer
    std::map<uintptr_t, Exported> exported_;       // by base address: allocations of this world whose handle is known
    std::map<std::string, void*> opened_;          // by handle bytes: peers' allocations mapped into this process
    int pull_mode_ = HNH_IPC_PULL_ENGINE, pull_wgs_ = 16;
    double wait_limit_s_ = 300.0;

    void flush();
    void* flag(int kind, int stream, int from, int to) const;
    const Exported& export_of(const void* ptr, uint64_t* offset, Exported* scratch);
    void* open_peer(const unsigned char* handle, uint64_t alloc_bytes);
    void make_room(size_t incoming);
    size_t max_opened_ = 512;  // HNH_IPC_MAX_OPENED (tests force evictions with a small table)
    template <typename Pred>
    void wait_host(Pred&& done, const char* what);
};

// ---- transport supplied by the embedding program (torch.distributed / gloo in tests, MPI, ...).
// All buffers handed to the callbacks are pointers in the backend's memory space.
// (struct hnh_comm_callbacks is declared in include/hnh_dist.h)
class CallbackWorld : public World {
public:
    CallbackWorld(int rank, int nranks, Backend* backend, int device_ordinal, const hnh_comm_callbacks& cb);
    ~CallbackWorld() override;
    const char* kind() const override { return "callback"; }
    void sendrecv(const Comm& comm, const void* sendbuf, size_t sendbytes, int dst, void* recvbuf, size_t recvbytes,
                  int src, int stream) override;
    void barrier() override;
    void host_allgather(const void* send, void* recv, size_t bytes) override;
    void host_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl, void* recv,
                        const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) override;

private:
    hnh_comm_callbacks cb_;
};

}  // namespace hnh
