#include "world.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <mutex>
#include <set>
#include <thread>

namespace hnh {

// ------------------------------------------------------------------------------------------------ errors
namespace {
bool g_throw = false;
}
void set_throw_on_error(bool on) { g_throw = on; }

namespace {
// worlds that exist: a thread's current-world pointer may outlive its world (destroyed on another thread, or a stack object gone out
// of scope), and fatal() must not call into one that is gone
std::mutex g_live_mu;
std::set<const World*> g_live_worlds;
}  // namespace

void fatal(const std::string& msg) {
    std::cout << msg << std::endl;  // the reference reports configuration errors on cout
    if (World* w = current_world_or_null()) {  // peers that wait for this rank learn now, not at their time limit
        std::lock_guard<std::mutex> lk(g_live_mu);
        if (g_live_worlds.count(w)) w->note_failure();
    }
    if (g_throw) throw Error(msg);
    std::exit(1);
}

// ------------------------------------------------------------------------------------------------ backend
namespace {
std::mutex g_backend_mu;
std::map<std::string, Backend*> g_backends;
Backend* g_default = nullptr;

std::string dir_of_this_library() {
    Dl_info info;
    if (dladdr((void*)&dir_of_this_library, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t slash = p.find_last_of('/');
        if (slash != std::string::npos) return p.substr(0, slash);
    }
    return ".";
}
}  // namespace

// Whether the process that left a rendezvous record (the RCCL unique id file, the ipc session's segment) is still there: a record whose
// writer is gone is a leftover of an earlier launch.  The test is by process id, so ranks that share /dev/shm or the id file's directory but
// run in SEPARATE PID namespaces (one container per rank) see each other's ids as dead: HNH_PID_LIVENESS=0 turns the test off there (records
// are then told apart by the launch's name alone — HNH_JOB_TOKEN / MASTER_PORT — as before round 5).
bool pid_is_live(int64_t pid) {
    static const bool check = [] {
        const char* v = std::getenv("HNH_PID_LIVENESS");
        return v == nullptr || std::atoi(v) != 0;
    }();
    if (!check) return true;
    return kill((pid_t)pid, 0) == 0 || errno == EPERM;
}

std::string default_backend_path() { return dir_of_this_library() + "/libhnh_kernels.so"; }

Backend* load_backend(const char* path_c) {
    std::lock_guard<std::mutex> lk(g_backend_mu);
    std::string path = (path_c && *path_c) ? std::string(path_c) : default_backend_path();
    auto it = g_backends.find(path);
    if (it != g_backends.end()) {
        g_default = it->second;
        return it->second;
    }
    void* dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dl) fatal(std::string("Error, cannot load the kernel library ") + path + ": " + dlerror() +
                   " (there is no CPU fallback; build it with __graft_entry__.build())");
    Backend* b = new Backend();
    b->dl = dl;
    b->path = path;
#define HNH_BIND(sym)                                                                  \
    b->sym = (decltype(b->sym))dlsym(dl, #sym);                                        \
    if (!b->sym) fatal(std::string("Error, kernel library ") + path + " lacks symbol " #sym);
    HNH_BIND(hnh_backend_name)
    HNH_BIND(hnh_ctx_create) HNH_BIND(hnh_ctx_destroy) HNH_BIND(hnh_last_error) HNH_BIND(hnh_ctx_stream) HNH_BIND(hnh_ctx_device_identity)
    HNH_BIND(hnh_malloc) HNH_BIND(hnh_free) HNH_BIND(hnh_memcpy) HNH_BIND(hnh_memset) HNH_BIND(hnh_stream_sync)
    HNH_BIND(hnh_event_create) HNH_BIND(hnh_event_destroy) HNH_BIND(hnh_event_record) HNH_BIND(hnh_event_wait)
    HNH_BIND(hnh_event_sync) HNH_BIND(hnh_event_query) HNH_BIND(hnh_event_elapsed_ms)
    HNH_BIND(hnh_sddmm_coo) HNH_BIND(hnh_sddmm_csr) HNH_BIND(hnh_spmm_csr) HNH_BIND(hnh_fused_sddmm_spmm_csr)
    HNH_BIND(hnh_sddmm_csr_ex) HNH_BIND(hnh_spmm_csr_ex) HNH_BIND(hnh_fused_sddmm_spmm_csr_ex) HNH_BIND(hnh_csr_max_row_nnz)
    HNH_BIND(hnh_fused_sddmm_spmm_csr_x) HNH_BIND(hnh_row_epilogue_f64) HNH_BIND(hnh_row_epilogue_x) HNH_BIND(hnh_cg_step_f64)
    HNH_BIND(hnh_tuples_sort) HNH_BIND(hnh_tuples_bucket_starts) HNH_BIND(hnh_tuples_transform) HNH_BIND(hnh_tuples_to_csr)
    HNH_BIND(hnh_csr_window_bounds) HNH_BIND(hnh_sddmm_csr_w) HNH_BIND(hnh_spmm_csr_w) HNH_BIND(hnh_fused_sddmm_spmm_csr_w) HNH_BIND(hnh_tuples_remap_cols) HNH_BIND(hnh_tuples_dedup_max) HNH_BIND(hnh_tuples_take_strided)
    HNH_BIND(hnh_panel_count) HNH_BIND(hnh_generate_er_keys) HNH_BIND(hnh_generate_rmat_keys) HNH_BIND(hnh_tuples_from_keys) HNH_BIND(hnh_tuples_relabel)
    HNH_BIND(hnh_fill_f64) HNH_BIND(hnh_hadamard_f64) HNH_BIND(hnh_axpy_f64) HNH_BIND(hnh_expand_rowptr) HNH_BIND(hnh_sum_chunked_blocks_f64)
    HNH_BIND(hnh_rowdot_f64) HNH_BIND(hnh_row_scale_add_f64) HNH_BIND(hnh_vec_add_scalar_f64) HNH_BIND(hnh_vec_div_f64) HNH_BIND(hnh_fill_hashed_f64)
    HNH_BIND(hnh_gemm_f64) HNH_BIND(hnh_leaky_relu_f64) HNH_BIND(hnh_relu_store_cols_f64)
    HNH_BIND(hnh_comm_unique_id) HNH_BIND(hnh_comm_init) HNH_BIND(hnh_comm_split) HNH_BIND(hnh_comm_destroy) HNH_BIND(hnh_comm_identity)
    HNH_BIND(hnh_comm_sendrecv) HNH_BIND(hnh_comm_group_begin) HNH_BIND(hnh_comm_group_end) HNH_BIND(hnh_comm_allgather) HNH_BIND(hnh_comm_reduce_scatter_f64)
    HNH_BIND(hnh_comm_allreduce_f64)
    HNH_BIND(hnh_ipc_export) HNH_BIND(hnh_ipc_open) HNH_BIND(hnh_ipc_close) HNH_BIND(hnh_ipc_pull) HNH_BIND(hnh_ipc_flags_register) HNH_BIND(hnh_ipc_flags_unregister)
    HNH_BIND(hnh_stream_write_flag) HNH_BIND(hnh_stream_wait_flag)
    HNH_BIND(hnh_csr_plan_create) HNH_BIND(hnh_csr_plan_destroy) HNH_BIND(hnh_sddmm_csr_p) HNH_BIND(hnh_sddmm_csr_ps) HNH_BIND(hnh_spmm_csr_p) HNH_BIND(hnh_spmm_csr_pf) HNH_BIND(hnh_fused_sddmm_spmm_csr_p)
#ifdef HNH_MEASUREMENT_AIDS
    HNH_BIND(hnh_stream_delay_us) HNH_BIND(hnh_stream_paced_copy) HNH_BIND(hnh_stream_pace_begin) HNH_BIND(hnh_stream_pace_end)
#endif
#undef HNH_BIND
    b->name = b->hnh_backend_name();
    g_backends[path] = b;
    g_default = b;
    return b;
}

Backend* default_backend() {
    {
        std::lock_guard<std::mutex> lk(g_backend_mu);
        if (g_default) return g_default;
    }
    return load_backend(nullptr);
}

// ------------------------------------------------------------------------------------------------ World base
namespace {
thread_local World* t_world = nullptr;
}
World* current_world() {
    if (!t_world) fatal("Error, no current world: create one and call hnh::set_current_world() first");
    return t_world;
}
World* current_world_or_null() { return t_world; }
void set_current_world(World* w) { t_world = w; }

World::World() {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live_worlds.insert(this);
}
World::~World() {
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live_worlds.erase(this);
    }
    if (t_world == this) t_world = nullptr;
}

void World::init_device(Backend* backend, int device_ordinal) {
    be = backend ? backend : default_backend();
    device = device_ordinal;
    if (const char* lim = std::getenv("HNH_POOL_LIMIT_GB")) pool_limit_ = (size_t)std::strtoull(lim, nullptr, 10) << 30;
    int st = be->hnh_ctx_create(device_ordinal, &ctx);
    if (st != HNH_OK || !ctx)
        fatal("Error, cannot create a device context on device " + std::to_string(device_ordinal) + " with backend " +
              be->name + " (status " + std::to_string(st) + "): no GPU? There is no CPU fallback.");
}

void World::destroy_device() {
    if (ctx) {
        for (auto& s : scratch_)
            if (s.first) dfree(s.first);
        scratch_.clear();
        drain_pool();
        be->hnh_ctx_destroy(ctx);
        ctx = nullptr;
    }
}

void World::check(int status, const char* what) const {
    if (status != HNH_OK)
        fatal(std::string("Error in ") + what + " (status " + std::to_string(status) + "): " + be->hnh_last_error(ctx));
}

void* World::dmalloc(size_t bytes) {
    bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    auto it = pool_.find(bytes);
    if (it != pool_.end()) {
        Parked e = it->second;
        pool_.erase(it);
        pooled_bytes_ -= bytes;
        // the new owner may touch the block on either stream: order it after the old owner's last use on the other one
        event_wait(e.ev[HNH_STREAM_COMPUTE], HNH_STREAM_COMM);
        event_wait(e.ev[HNH_STREAM_COMM], HNH_STREAM_COMPUTE);
        event_destroy(e.ev[0]);
        event_destroy(e.ev[1]);
        live_[e.ptr] = bytes;
        return e.ptr;
    }
    void* p = nullptr;
    int st = be->hnh_malloc(ctx, bytes, &p);
    if (st == HNH_ERR_NOMEM && !pool_.empty()) {  // give parked memory back and retry once
        drain_pool();
        st = be->hnh_malloc(ctx, bytes, &p);
    }
    check(st, "hnh_malloc");
    live_[p] = bytes;
    return p;
}

void World::dfree(void* p) {
    if (!p) return;
    auto it = live_.find(p);
    if (it == live_.end()) fatal("Error, freeing a device pointer this world did not allocate");
    const size_t bytes = it->second;
    live_.erase(it);
    if (pooled_bytes_ + bytes <= pool_limit_) {
        Parked e{p, {event_create(), event_create()}};
        event_record(e.ev[HNH_STREAM_COMPUTE], HNH_STREAM_COMPUTE);
        event_record(e.ev[HNH_STREAM_COMM], HNH_STREAM_COMM);
        pool_.insert({bytes, e});
        pooled_bytes_ += bytes;
        return;
    }
    sync_all();
    on_device_release(p);
    check(be->hnh_free(ctx, p), "hnh_free");
}

void World::drain_pool() {
    sync_all_nothrow();
    for (auto& kv : pool_) {
        be->hnh_event_destroy(ctx, kv.second.ev[0]);
        be->hnh_event_destroy(ctx, kv.second.ev[1]);
        on_device_release(kv.second.ptr);
        be->hnh_free(ctx, kv.second.ptr);
    }
    pool_.clear();
    pooled_bytes_ = 0;
}
void World::copy(void* dst, const void* src, size_t bytes, int kind, int stream) {
    if (bytes) check(be->hnh_memcpy(ctx, dst, src, bytes, kind, stream), "hnh_memcpy");
}
void World::memset0(void* dst, size_t bytes, int stream) {
    if (bytes) check(be->hnh_memset(ctx, dst, 0, bytes, stream), "hnh_memset");
}
void World::sync(int stream) { check(be->hnh_stream_sync(ctx, stream), "hnh_stream_sync"); }
void World::sync_all() {
    sync(HNH_STREAM_COMPUTE);
    sync(HNH_STREAM_COMM);
    sync(HNH_STREAM_AUX);
}
// Teardown variant for destructors (implicitly noexcept): a device error at this point is reported, never thrown.
void World::sync_all_nothrow() noexcept {
    if (!ctx) return;
    for (int st : {HNH_STREAM_COMPUTE, HNH_STREAM_COMM, HNH_STREAM_AUX}) {
        const int rc = be->hnh_stream_sync(ctx, st);
        if (rc != HNH_OK) std::cerr << "hnh: device error during teardown (stream " << st << ", status " << rc << "): " << be->hnh_last_error(ctx) << std::endl;
    }
}
void* World::event_create() {
    void* e = nullptr;
    check(be->hnh_event_create(ctx, &e), "hnh_event_create");
    return e;
}
void World::event_destroy(void* e) {
    if (e) check(be->hnh_event_destroy(ctx, e), "hnh_event_destroy");
}
void World::event_record(void* e, int stream) { check(be->hnh_event_record(ctx, e, stream), "hnh_event_record"); }
void World::event_wait(void* e, int stream) { check(be->hnh_event_wait(ctx, e, stream), "hnh_event_wait"); }
#ifdef HNH_MEASUREMENT_AIDS
void World::delay_us(double us, int stream) { check(be->hnh_stream_delay_us(ctx, stream, us), "hnh_stream_delay_us"); }
#endif

void* World::scratch(int slot, size_t bytes) {
    if ((int)scratch_.size() <= slot) scratch_.resize(slot + 1, {nullptr, 0});
    auto& s = scratch_[slot];
    if (s.second < bytes) {
        if (s.first) dfree(s.first);
        s.first = dmalloc(bytes);
        s.second = bytes;
    }
    return s.first;
}
void World::event_sync(void* e) { check(be->hnh_event_sync(ctx, e), "hnh_event_sync"); }
bool World::event_done(void* e) {
    int done = 0;
    check(be->hnh_event_query(ctx, e, &done), "hnh_event_query");
    return done != 0;
}

void World::set_solo(bool on) {
    if (on) fatal(std::string("Error, solo replay is a measurement mode of the loopback transport, not of ") + kind());
}

hnh_rank_identity World::identity() {
    hnh_rank_identity id;
    std::memset(&id, 0, sizeof id);
    id.rank = rank;
    id.pid = (int32_t)getpid();
    id.comm_count = id.comm_rank = id.comm_device = -1;
    int ordinal = device;
    if (be->hnh_ctx_device_identity(ctx, &ordinal, id.pci_bus_id, (int)sizeof id.pci_bus_id) != HNH_OK)
        std::snprintf(id.pci_bus_id, sizeof id.pci_bus_id, "unknown:%d", device);  // (a report, never a reason to fail)
    id.device_ordinal = ordinal;
    return id;
}
std::vector<hnh_rank_identity> World::identities() {
    const hnh_rank_identity mine = identity();
    std::vector<hnh_rank_identity> all((size_t)size);
    host_allgather(&mine, all.data(), sizeof mine);
    return all;
}

Comm World::world_comm() {
    Comm c;
    c.ranks.resize(size);
    for (int i = 0; i < size; i++) c.ranks[i] = i;
    c.me = rank;
    c.is_world = true;
    return c;
}

Comm World::split(int color, int key) {
    // collective: gather (color, key) of every rank, keep my colour, order by (key, world rank)
    std::vector<int> mine = {color, key}, all(2 * (size_t)size);
    host_allgather(mine.data(), all.data(), 2 * sizeof(int));
    std::vector<std::pair<int, int>> members;  // (key, world rank)
    for (int r = 0; r < size; r++)
        if (all[2 * r] == color) members.push_back({all[2 * r + 1], r});
    std::sort(members.begin(), members.end());
    // every rank folds the WORLD-WIDE (color, key) table of this split into a running signature: ranks that created their
    // communicators in a different order end up with different signatures (bench.py's preflight compares them)
    for (int v : all) split_signature = (split_signature ^ (uint64_t)(uint32_t)v) * 1099511628211ULL;
    split_count++;
    Comm c;
    c.color = color;
    c.key = key;
    for (size_t i = 0; i < members.size(); i++) {
        c.ranks.push_back(members[i].second);
        if (members[i].second == rank) c.me = (int)i;
    }
    return c;
}

void World::free_comm(Comm& c) { (void)c; }

// Default collectives: (n - 1) rounds of pairwise exchange; round k pairs me -> me + k, me - k -> me.
void World::allgather(const Comm& comm, const void* sendbuf, void* recvbuf, size_t bytes, int stream) {
    const int n = comm.size(), me = comm.me;
    char* out = static_cast<char*>(recvbuf);
    if (out + (size_t)me * bytes != sendbuf) copy(out + (size_t)me * bytes, sendbuf, bytes, HNH_COPY_D2D, stream);
    if (n == 1) return;
    group_begin();  // RCCL: the n - 1 explicit-peer pairs of a group progress concurrently, one xGMI link each
    for (int k = 1; k < n; k++) {
        const int dst = (me + k) % n, src = (me - k + n) % n;
        sendrecv(comm, out + (size_t)me * bytes, bytes, dst, out + (size_t)src * bytes, bytes, src, stream);
    }
    group_end();
}

void World::allgatherv_f64(const Comm& comm, const double* sendbuf, size_t sendcount, double* recvbuf,
                           const std::vector<int>& counts, const std::vector<int>& displs, int stream) {
    const int n = comm.size(), me = comm.me;
    if ((size_t)counts[me] != sendcount) fatal("Error, allgatherv: send count does not match counts[me]");
    if (recvbuf + displs[me] != sendbuf) copy(recvbuf + displs[me], sendbuf, sendcount * sizeof(double), HNH_COPY_D2D, stream);
    if (n == 1) return;
    group_begin();
    for (int k = 1; k < n; k++) {
        const int dst = (me + k) % n, src = (me - k + n) % n;
        sendrecv(comm, recvbuf + displs[me], sendcount * sizeof(double), dst, recvbuf + displs[src],
                 (size_t)counts[src] * sizeof(double), src, stream);
    }
    group_end();
}

void World::reduce_scatter_v_f64(const Comm& comm, const double* sendbuf, double* recvbuf, const std::vector<int>& counts,
                                 int stream) {
    const int n = comm.size(), me = comm.me;
    std::vector<size_t> displ(n + 1, 0);
    for (int i = 0; i < n; i++) displ[i + 1] = displ[i] + (size_t)counts[i];
    const size_t mycount = (size_t)counts[me];
    copy(recvbuf, sendbuf + displ[me], mycount * sizeof(double), HNH_COPY_D2D, stream);
    if (n == 1) return;
    // every member's contribution to my shard lands in its own buffer (one group: n - 1 links at once), then the
    // shards are added in a fixed order, so the result does not depend on arrival order
    double* tmp = static_cast<double*>(scratch(0, std::max<size_t>(mycount, 1) * (size_t)(n - 1) * sizeof(double)));
    group_begin();
    for (int k = 1; k < n; k++) {
        const int dst = (me + k) % n, src = (me - k + n) % n;
        sendrecv(comm, sendbuf + displ[dst], (size_t)counts[dst] * sizeof(double), dst, tmp + (size_t)(k - 1) * mycount,
                 mycount * sizeof(double), src, stream);
    }
    group_end();
    for (int k = 1; k < n; k++)
        if (mycount) check(be->hnh_axpy_f64(ctx, recvbuf, tmp + (size_t)(k - 1) * mycount, 1.0, (int64_t)mycount, stream), "hnh_axpy_f64");
}

void World::reduce_scatter_f64(const Comm& comm, const double* sendbuf, double* recvbuf, size_t count, int stream) {
    std::vector<int> counts(comm.size(), (int)count);
    reduce_scatter_v_f64(comm, sendbuf, recvbuf, counts, stream);
}

void World::allreduce_f64(const Comm& comm, double* buf, size_t count, int stream) {
    const int n = comm.size();
    if (n == 1 || count == 0) return;
    // gather every member's vector, then add them up in member order (deterministic)
    double* all = static_cast<double*>(scratch(2, (size_t)n * count * sizeof(double)));
    allgather(comm, buf, all, count * sizeof(double), stream);
    memset0(buf, count * sizeof(double), stream);
    for (int i = 0; i < n; i++) check(be->hnh_axpy_f64(ctx, buf, all + (size_t)i * count, 1.0, (int64_t)count, stream), "hnh_axpy_f64");
}

void World::host_allgather_comm(const Comm& comm, const void* send, void* recv, size_t bytes) {
    // tiny payloads only (nnz counts): gather over the world, keep the members
    std::vector<char> all((size_t)size * bytes);
    host_allgather(send, all.data(), bytes);
    for (int i = 0; i < comm.size(); i++)
        std::memcpy(static_cast<char*>(recv) + (size_t)i * bytes, all.data() + (size_t)comm.ranks[i] * bytes, bytes);
}

void World::host_bcast(const Comm& comm, int root, void* buf, size_t bytes) {
    // generic: route through alltoallv (the root sends a copy to every other member)
    std::vector<size_t> sb(size, 0), sd(size, 0), rb(size, 0), rd(size, 0);
    const bool is_root = (comm.me == root);
    if (is_root) {
        for (int i = 0; i < comm.size(); i++)
            if (i != root) sb[comm.ranks[i]] = bytes;  // all read the same source bytes (displ 0)
    } else {
        rb[comm.ranks[root]] = bytes;
    }
    std::vector<char> tmp(is_root ? 0 : bytes);
    host_alltoallv(buf, sb, sd, is_root ? nullptr : tmp.data(), rb, rd);
    if (!is_root && bytes) std::memcpy(buf, tmp.data(), bytes);
}

void World::device_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl, void* recv,
                             const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl, int stream) {
    Comm w = world_comm();
    const char* s = static_cast<const char*>(send);
    char* r = static_cast<char*>(recv);
    for (int k = 0; k < size; k++) {  // round k: send to me + k, receive from me - k
        const int dst = (rank + k) % size, src = (rank - k + size) % size;
        if (k == 0) {
            if (sendbytes[rank] != recvbytes[rank]) fatal("Error, self send/recv size mismatch");
            if (sendbytes[rank]) copy(r + recvdispl[rank], s + senddispl[rank], sendbytes[rank], HNH_COPY_D2D, stream);
        } else if (sendbytes[dst] || recvbytes[src]) {
            sendrecv(w, s + senddispl[dst], sendbytes[dst], dst, r + recvdispl[src], recvbytes[src], src, stream);
        }
    }
    sync(stream);
}

void World::device_bcast(const Comm& comm, int root, void* buf, size_t bytes, int stream) {
    std::vector<size_t> sb(size, 0), sd(size, 0), rb(size, 0), rd(size, 0);
    const bool is_root = (comm.me == root);
    if (is_root) {
        for (int i = 0; i < comm.size(); i++)
            if (i != root) sb[comm.ranks[i]] = bytes;  // every member reads the same source bytes (displ 0)
    } else {
        rb[comm.ranks[root]] = bytes;
    }
    device_alltoallv(buf, sb, sd, buf, rb, rd, stream);
}

double World::host_allreduce_sum(double v) {
    host_allreduce_sum(&v, 1);
    return v;
}

void World::host_allreduce_sum(double* v, size_t n) {
    std::vector<double> all((size_t)size * n);
    host_allgather(v, all.data(), n * sizeof(double));
    for (size_t j = 0; j < n; j++) {
        double s = 0.0;
        for (int r = 0; r < size; r++) s += all[(size_t)r * n + j];
        v[j] = s;
    }
}

// ------------------------------------------------------------------------------------------------ bootstrap from the environment
namespace {
int env_int(const char* k, int dflt) {
    const char* v = std::getenv(k);
    return (v && *v) ? std::atoi(v) : dflt;
}

// Who am I, says the launcher: torchrun / a shell loop (RANK, WORLD_SIZE, LOCAL_RANK), MPICH's mpiexec (PMI_RANK, PMI_SIZE,
// MPI_LOCALRANKID) or Open MPI's (OMPI_COMM_WORLD_*) — the launchers are only used to START the processes, no MPI library is linked.
// `token` is a name all ranks of THIS launch share and other launches do not: HNH_JOB_TOKEN, or the launcher's process id (every
// rank of a one-node launch is a child of the same agent / proxy / shell) plus the rendezvous port when there is one.
struct Launch {
    int rank = 0, size = 1, local = 0;
    std::string token;
};
Launch launch_from_environment() {
    static const char* const kinds[][3] = {{"RANK", "WORLD_SIZE", "LOCAL_RANK"},
                                           {"PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID"},
                                           {"OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK"}};
    Launch l;
    for (const auto& k : kinds) {
        const char* n = std::getenv(k[1]);
        if (n == nullptr || !*n) continue;
        l.size = std::atoi(n);
        l.rank = env_int(k[0], 0);
        l.local = env_int(k[2], l.rank);
        break;
    }
    if (const char* t = std::getenv("HNH_JOB_TOKEN")) l.token = t;
    if (l.token.empty()) {
        l.token = "p" + std::to_string((long)getppid());
        if (const char* port = std::getenv("MASTER_PORT")) l.token += std::string("_") + port;
    }
    return l;
}
}  // namespace

World* world_from_environment() {
    const Launch l = launch_from_environment();
    const int rank = l.rank, n = l.size, local = l.local;
    if (n < 1 || rank < 0 || rank >= n) fatal("Error, the launcher's environment (RANK / WORLD_SIZE, PMI_RANK / PMI_SIZE, ...) does not describe a rank of a world");
    // (before the HIP runtime starts) streams that share a hardware queue serialise: leave room beyond the default 4 queues so
    // that the compute and the communication stream never have to share one with each other or with RCCL's
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);  // hosts that only support dmabuf IPC (RCCL and mapped peer memory across processes)
    Backend* be = load_backend(nullptr);  // the kernel library next to this one; exits if missing
    const int device = env_int("HNH_DEVICE", local);
    if (n == 1) return new SingleWorld(be, device);
    const char* transport = std::getenv("HNH_TRANSPORT");
    if (transport && std::string(transport) == "ipc") {
        const char* session = std::getenv("HNH_IPC_SESSION");
        return new IpcWorld(rank, n, be, device, (session && *session) ? std::string(session) : "auto_" + l.token);
    }
    if (transport && *transport && std::string(transport) != "rccl") fatal(std::string("Error, unknown HNH_TRANSPORT ") + transport + " (rccl or ipc)");
    // RCCL: rank 0 hands the unique id to the others through a file — HNH_ID_FILE (any path all ranks see), or one in /dev/shm named
    // after the launch, which rank 0 removes once the communicator exists (creating it is collective: by then every rank has read it)
    const char* idfile_env = std::getenv("HNH_ID_FILE");
    const bool own_file = (idfile_env == nullptr || !*idfile_env);
    const std::string idfile = own_file ? "/dev/shm/hnh_rccl_id_" + l.token : std::string(idfile_env);
    // The record is the id followed by rank 0's pid: a file left behind by an earlier launch (a run killed before rank 0 removed it, a
    // user-named file that was never removed; launches from one shell share the automatic name) names a process that is gone, and is
    // waited out instead of being joined — ncclCommInitRank on a stale id hangs.
    char id[HNH_UNIQUE_ID_BYTES];
    if (rank == 0) {
        if (be->hnh_comm_unique_id(id) != HNH_OK) fatal("Error, cannot create an RCCL unique id (HNH_TRANSPORT=ipc selects the ipc-pull transport)");
        std::remove(idfile.c_str());
        const std::string tmp = idfile + ".tmp";
        const int64_t me = (int64_t)getpid();
        {
            std::ofstream f(tmp, std::ios::binary);
            f.write(id, sizeof(id));
            f.write(reinterpret_cast<const char*>(&me), sizeof(me));
        }
        if (std::rename(tmp.c_str(), idfile.c_str()) != 0) fatal("Error, cannot write the RCCL unique id file " + idfile);
    } else {
        for (int tries = 0;; tries++) {
            std::ifstream f(idfile, std::ios::binary);
            int64_t writer = 0;
            if (f && f.read(id, sizeof(id)) && f.read(reinterpret_cast<char*>(&writer), sizeof(writer)) && writer > 0 && pid_is_live(writer))
                break;
            if (tries > 6000) fatal("Error, timed out waiting for the RCCL unique id file " + idfile + " (of a live rank 0)");
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    World* w = new RcclWorld(rank, n, be, device, id);
    if (rank == 0) std::remove(idfile.c_str());  // creating the communicator is collective: every rank has read the record
    return w;
}

namespace {
World* g_process_world = nullptr;
void destroy_process_world() {
    World* w = g_process_world;
    g_process_world = nullptr;
    if (w == nullptr) return;
    w->sync_all_nothrow();
    if (current_world_or_null() == w) set_current_world(nullptr);
    delete w;
}
}  // namespace

World* process_world() {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (g_process_world == nullptr) {
        g_process_world = world_from_environment();
        std::atexit(destroy_process_world);  // registered after the kernel library (and with it the HIP runtime) was loaded: runs before their handlers
    }
    set_current_world(g_process_world);
    return g_process_world;
}

// ------------------------------------------------------------------------------------------------ SingleWorld
SingleWorld::SingleWorld(Backend* backend, int device_ordinal) {
    rank = 0;
    size = 1;
    init_device(backend, device_ordinal);
}
SingleWorld::~SingleWorld() { destroy_device(); }

void SingleWorld::sendrecv(const Comm&, const void* sendbuf, size_t sendbytes, int, void* recvbuf, size_t recvbytes, int,
                           int stream) {
    if (sendbytes != recvbytes) fatal("Error, self send/recv size mismatch");
    if (sendbuf != recvbuf) copy(recvbuf, sendbuf, sendbytes, HNH_COPY_D2D, stream);
}
void SingleWorld::host_allgather(const void* send, void* recv, size_t bytes) { std::memcpy(recv, send, bytes); }
void SingleWorld::host_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl,
                                 void* recv, const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) {
    if (sendbytes[0] != recvbytes[0]) fatal("Error, alltoallv size mismatch");
    if (sendbytes[0]) std::memcpy(static_cast<char*>(recv) + recvdispl[0], static_cast<const char*>(send) + senddispl[0], sendbytes[0]);
}

// ------------------------------------------------------------------------------------------------ ThreadWorld
struct ThreadGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    // barrier
    int arrived = 0;
    uint64_t generation = 0;
    // A rank that waits for a peer longer than this gives up with an error instead of hanging for ever (a peer that failed
    // — configuration error, device error — never arrives).  HNH_THREAD_WAIT_S, default 300 s.
    double wait_limit_s() const {
        const char* v = std::getenv("HNH_THREAD_WAIT_S");
        return v ? std::atof(v) : 300.0;
    }
    template <typename Pred>
    void wait_or_fail(std::unique_lock<std::mutex>& lk, Pred&& done) {
        if (!cv.wait_for(lk, std::chrono::duration<double>(wait_limit_s()), done))
            fatal("Error, a peer rank of the thread group did not arrive (it probably failed)");
    }
    // pointer publication
    std::vector<const void*> slots;
    // point-to-point mailboxes, index src * n + dst
    struct Msg {
        const void* ptr = nullptr;
        size_t bytes = 0;
        void* ready = nullptr;  // sender's event: data valid
        void* done = nullptr;   // receiver's event: copy finished
        bool completed = false;
    };
    std::vector<std::deque<Msg*>> box;
};

std::shared_ptr<ThreadGroup> make_thread_group(int nranks) {
    auto g = std::make_shared<ThreadGroup>();
    g->n = nranks;
    g->slots.assign(nranks, nullptr);
    g->box.resize((size_t)nranks * nranks);
    return g;
}

ThreadWorld::ThreadWorld(std::shared_ptr<ThreadGroup> group, int rank_in_group, Backend* backend, int device_ordinal)
    : g_(std::move(group)) {
    rank = rank_in_group;
    size = g_->n;
    init_device(backend, device_ordinal);
}
ThreadWorld::~ThreadWorld() { destroy_device(); }

void ThreadWorld::barrier() {
    if (solo_) return;
    std::unique_lock<std::mutex> lk(g_->mu);
    const uint64_t gen = g_->generation;
    if (++g_->arrived == g_->n) {
        g_->arrived = 0;
        g_->generation++;
        g_->cv.notify_all();
    } else {
        if (!g_->cv.wait_for(lk, std::chrono::duration<double>(g_->wait_limit_s()), [&] { return g_->generation != gen; })) {
            g_->arrived--;  // this rank leaves the barrier again
            fatal("Error, a peer rank of the thread group did not arrive (it probably failed)");
        }
    }
}

std::vector<const void*> ThreadWorld::publish(const void* mine) {
    {
        std::lock_guard<std::mutex> lk(g_->mu);
        g_->slots[rank] = mine;
    }
    barrier();
    return g_->slots;  // copy; peers' data stays valid until release()
}
void ThreadWorld::release() { barrier(); }

void ThreadWorld::host_allgather(const void* send, void* recv, size_t bytes) {
    if (solo_) {
        for (int r = 0; r < size; r++) std::memcpy(static_cast<char*>(recv) + (size_t)r * bytes, send, bytes);
        return;
    }
    auto all = publish(send);
    for (int r = 0; r < size; r++) std::memcpy(static_cast<char*>(recv) + (size_t)r * bytes, all[r], bytes);
    release();
}

void ThreadWorld::host_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl,
                                 void* recv, const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) {
    if (solo_) {
        for (int r = 0; r < size; r++)
            if (const size_t nb = std::min(sendbytes[r], recvbytes[r]))
                std::memcpy(static_cast<char*>(recv) + recvdispl[r], static_cast<const char*>(send) + senddispl[r], nb);
        return;
    }
    struct Desc { const void* base; const size_t* bytes; const size_t* displ; } mine{send, sendbytes.data(), senddispl.data()};
    auto all = publish(&mine);
    for (int r = 0; r < size; r++) {
        const Desc* d = static_cast<const Desc*>(all[r]);
        if (d->bytes[rank] != recvbytes[r]) fatal("Error, alltoallv size mismatch between ranks");
        if (recvbytes[r])
            std::memcpy(static_cast<char*>(recv) + recvdispl[r], static_cast<const char*>(d->base) + d->displ[rank], recvbytes[r]);
    }
    release();
}

#ifdef HNH_MEASUREMENT_AIDS
void ThreadWorld::group_begin() {
    in_group_ = true;
    group_us_ = 0.0;
}
void ThreadWorld::group_end() {
    if (group_us_ > 0.0) check(be->hnh_stream_pace_end(ctx, group_stream_, group_us_), "hnh_stream_pace_end");
    in_group_ = false;
    group_us_ = 0.0;
}
#endif

void ThreadWorld::sendrecv(const Comm& comm, const void* sendbuf, size_t sendbytes, int dst_idx, void* recvbuf, size_t recvbytes,
                           int src_idx, int stream) {
    if (solo_) {  // the message this rank would receive = a copy of the one it would send
        const size_t nb = std::min(sendbytes, recvbytes);
        if (nb && sendbuf != recvbuf) copy(recvbuf, sendbuf, nb, HNH_COPY_D2D, stream);
        return;
    }
    const int n = g_->n;
    ThreadGroup::Msg* out = nullptr;
    if (sendbytes) {
        const int dst = comm.ranks[dst_idx];
        out = new ThreadGroup::Msg();
        out->ptr = sendbuf;
        out->bytes = sendbytes;
        out->ready = event_create();
        event_record(out->ready, stream);  // everything enqueued so far on `stream` produced sendbuf
        std::lock_guard<std::mutex> lk(g_->mu);
        g_->box[(size_t)rank * n + dst].push_back(out);
        g_->cv.notify_all();
    }
    if (recvbytes) {
        const int src = comm.ranks[src_idx];
        ThreadGroup::Msg* in = nullptr;
        {
            std::unique_lock<std::mutex> lk(g_->mu);
            auto& q = g_->box[(size_t)src * n + rank];
            g_->wait_or_fail(lk, [&] { return !q.empty(); });
            in = q.front();
            q.pop_front();
        }
        if (in->bytes != recvbytes) fatal("Error, send/recv size mismatch between ranks");
        event_wait(in->ready, stream);
#ifdef HNH_MEASUREMENT_AIDS
        // (libhnh_host_aids.so only) HNH_PACE_LINK_GBPS=<rate>: the message takes at least as long as it would need to cross ONE
        // xGMI link at that rate — the loopback copy included (stamp the clock, copy, hold the stream until the modelled time has
        // passed since the stamp): a transfer of known duration for overlap measurements on a single GPU
        // (tools/overlap_probe_accumulator.py)
        double pace_us = 0.0;
        if (const char* pace = std::getenv("HNH_PACE_LINK_GBPS")) {
            const double gbps = std::atof(pace);
            if (gbps > 0.0) pace_us = (double)recvbytes / (gbps * 1e3);
        }
        // The messages of ONE group (group_begin .. group_end) go to different peers over different links and travel together: the group
        // as a whole takes as long as its longest message, not their sum (clock stamped before the first copy, stream held at group_end).
        if (pace_us > 0.0 && in_group_) {
            if (group_us_ == 0.0) check(be->hnh_stream_pace_begin(ctx, stream), "hnh_stream_pace_begin");
            group_us_ = std::max(group_us_, pace_us);
            group_stream_ = stream;
            pace_us = 0.0;
        }
        if (pace_us > 0.0) check(be->hnh_stream_pace_begin(ctx, stream), "hnh_stream_pace_begin");
#endif
        copy(recvbuf, in->ptr, recvbytes, HNH_COPY_D2D, stream);
#ifdef HNH_MEASUREMENT_AIDS
        if (pace_us > 0.0) check(be->hnh_stream_pace_end(ctx, stream, pace_us), "hnh_stream_pace_end");
#endif
        void* done = event_create();
        event_record(done, stream);
        {
            std::lock_guard<std::mutex> lk(g_->mu);
            in->done = done;
            in->completed = true;
            g_->cv.notify_all();
        }
    }
    if (out) {
        {
            std::unique_lock<std::mutex> lk(g_->mu);
            g_->wait_or_fail(lk, [&] { return out->completed; });
        }
        event_wait(out->done, stream);  // do not let later work on `stream` overwrite sendbuf before the peer's copy ran
        event_destroy(out->done);
        event_destroy(out->ready);
        delete out;
    }
}

// ------------------------------------------------------------------------------------------------ RcclWorld
RcclWorld::RcclWorld(int r, int nranks, Backend* backend, int device_ordinal, const void* unique_id) {
    rank = r;
    size = nranks;
    init_device(backend, device_ordinal);
    check(be->hnh_comm_init(ctx, nranks, r, unique_id, &comm_), "hnh_comm_init");
}
hnh_rank_identity RcclWorld::identity() {
    hnh_rank_identity id = World::identity();
    // (a report, never a reason to fail: identities() is collective, and a rank that threw here would leave the others waiting)
    int n = -1, r = -1, d = -1;
    if (be->hnh_comm_identity(ctx, comm_, &n, &r, &d) != HNH_OK) n = r = d = -2;
    id.comm_count = n;
    id.comm_rank = r;
    id.comm_device = d;
    return id;
}
RcclWorld::~RcclWorld() {
    if (ctx) {
        sync_all_nothrow();
        if (comm_) be->hnh_comm_destroy(ctx, comm_);
    }
    destroy_device();
}

// Sub-communicators (rows / columns / fibers of the process grid) are NOT RCCL communicators: their collectives run as
// one group of explicit-peer send/recv pairs on the world communicator (World::allgather etc.).  On the xGMI full mesh
// that is the natural algorithm — every pair has its own link — and it needs no ncclCommSplit, whose preconditions
// (no operation in flight on the parent, identical creation order on all ranks) a lazily created communicator could
// violate (round-1 advisor finding).  Native RCCL collectives are used where the communicator IS the world.
void RcclWorld::group_begin() { check(be->hnh_comm_group_begin(ctx), "hnh_comm_group_begin"); }
void RcclWorld::group_end() { check(be->hnh_comm_group_end(ctx), "hnh_comm_group_end"); }

void RcclWorld::sendrecv(const Comm& comm, const void* sendbuf, size_t sendbytes, int dst, void* recvbuf, size_t recvbytes,
                         int src, int stream) {
    if (comm.ranks[dst] == rank && comm.ranks[src] == rank) {  // ring of one: plain copy, no RCCL call
        if (sendbytes != recvbytes) fatal("Error, self send/recv size mismatch");
        if (sendbuf != recvbuf) copy(recvbuf, sendbuf, sendbytes, HNH_COPY_D2D, stream);
        return;
    }
    // point-to-point always goes through the world communicator with world ranks: explicit peers
    check(be->hnh_comm_sendrecv(ctx, comm_, sendbuf, sendbytes, comm.ranks[dst], recvbuf, recvbytes, comm.ranks[src], stream),
          "hnh_comm_sendrecv");
}
void RcclWorld::allgather(const Comm& comm, const void* sendbuf, void* recvbuf, size_t bytes, int stream) {
    if (comm.size() == 1) {
        if (sendbuf != recvbuf) copy(recvbuf, sendbuf, bytes, HNH_COPY_D2D, stream);
        return;
    }
    if (!comm.is_world) return World::allgather(comm, sendbuf, recvbuf, bytes, stream);
    check(be->hnh_comm_allgather(ctx, comm_, sendbuf, recvbuf, bytes, stream), "hnh_comm_allgather");
}
void RcclWorld::reduce_scatter_f64(const Comm& comm, const double* sendbuf, double* recvbuf, size_t count, int stream) {
    if (comm.size() == 1) {
        copy(recvbuf, sendbuf, count * sizeof(double), HNH_COPY_D2D, stream);
        return;
    }
    if (!comm.is_world) return World::reduce_scatter_f64(comm, sendbuf, recvbuf, count, stream);
    check(be->hnh_comm_reduce_scatter_f64(ctx, comm_, sendbuf, recvbuf, count, stream), "hnh_comm_reduce_scatter_f64");
}

void RcclWorld::allreduce_f64(const Comm& comm, double* buf, size_t count, int stream) {
    if (comm.size() == 1 || count == 0) return;
    if (!comm.is_world) return World::allreduce_f64(comm, buf, count, stream);
    check(be->hnh_comm_allreduce_f64(ctx, comm_, buf, buf, count, stream), "hnh_comm_allreduce_f64");
}

void RcclWorld::barrier() {
    double* d = static_cast<double*>(scratch(1, sizeof(double)));
    check(be->hnh_comm_allreduce_f64(ctx, comm_, d, d, 1, HNH_STREAM_COMM), "hnh_comm_allreduce_f64");
    sync(HNH_STREAM_COMM);
}

void RcclWorld::host_allgather(const void* send, void* recv, size_t bytes) {
    // host data is staged through device memory (setup path only)
    char* d = static_cast<char*>(scratch(1, std::max<size_t>((size_t)size * bytes, 16)));
    copy(d + (size_t)rank * bytes, send, bytes, HNH_COPY_H2D, HNH_STREAM_COMM);
    check(be->hnh_comm_allgather(ctx, comm_, d + (size_t)rank * bytes, d, bytes, HNH_STREAM_COMM), "hnh_comm_allgather");
    copy(recv, d, (size_t)size * bytes, HNH_COPY_D2H, HNH_STREAM_COMM);
    sync(HNH_STREAM_COMM);
}

void RcclWorld::host_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl,
                               void* recv, const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) {
    host_alltoallv_staged(send, sendbytes, senddispl, recv, recvbytes, recvdispl);
}

// Host data staged through device memory and the device transport (set-up path only).
void World::host_alltoallv_staged(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl,
                                  void* recv, const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) {
    size_t stot = 0, rtot = 0;
    for (int r = 0; r < size; r++) {
        stot = std::max(stot, senddispl[r] + sendbytes[r]);
        rtot = std::max(rtot, recvdispl[r] + recvbytes[r]);
    }
    char* ds = static_cast<char*>(dmalloc(std::max<size_t>(stot, 16)));
    char* dr = static_cast<char*>(dmalloc(std::max<size_t>(rtot, 16)));
    copy(ds, send, stot, HNH_COPY_H2D, HNH_STREAM_COMM);
    Comm w = world_comm();
    for (int k = 0; k < size; k++) {  // round k: send to me + k, receive from me - k
        const int dst = (rank + k) % size, src = (rank - k + size) % size;
        if (k == 0) {
            copy(dr + recvdispl[rank], ds + senddispl[rank], sendbytes[rank], HNH_COPY_D2D, HNH_STREAM_COMM);
        } else {
            sendrecv(w, ds + senddispl[dst], sendbytes[dst], dst, dr + recvdispl[src], recvbytes[src], src, HNH_STREAM_COMM);
        }
    }
    if (rtot) copy(recv, dr, rtot, HNH_COPY_D2H, HNH_STREAM_COMM);
    sync(HNH_STREAM_COMM);
    dfree(ds);
    dfree(dr);
}

// ------------------------------------------------------------------------------------------------ IpcWorld
namespace {
constexpr int kIpcMaxRanks = 16;
constexpr int kIpcSlots = 32;                   // descriptors in flight per ordered pair before the sender's HOST waits
constexpr size_t kIpcPublishBytes = 64 * 1024;  // host all-gather staging per rank and round
constexpr uint64_t kIpcMagic = 0x686e685f69706331ULL;
}  // namespace

struct IpcShared {
    std::atomic<uint64_t> magic;
    int nranks;
    int64_t creator_pid;  // rank 0's: a segment whose creator is gone is a leftover of an earlier launch with the same session name
    std::atomic<int> attached, detached, failed;
    std::atomic<uint64_t> bar_arrived, bar_generation;
    struct Msg {
        std::atomic<uint64_t> seq;
        uint64_t offset, bytes, alloc_bytes;
        int32_t stream, pad;
        unsigned char handle[HNH_IPC_HANDLE_BYTES];
    };
    Msg box[kIpcMaxRanks * kIpcMaxRanks][kIpcSlots];       // [from * max + to]
    std::atomic<uint64_t> consumed[kIpcMaxRanks * kIpcMaxRanks];
    // device-visible part (registered with the runtime): flag words, one 64-byte line each: [ready | done][stream][from * max + to]
    alignas(4096) uint64_t flags[2][2][kIpcMaxRanks * kIpcMaxRanks][8];
    alignas(4096) char publish[kIpcMaxRanks][kIpcPublishBytes];
};

template <typename Pred>
void IpcWorld::wait_host(Pred&& done, const char* what) {
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    for (uint64_t spins = 0;; spins++) {
        if (done()) return;
        if (spins < 2000) continue;
        if ((spins & 63) == 0) {
            if (sh_->failed.load(std::memory_order_relaxed)) fatal(std::string("Error, a peer rank of the ipc world gave up while this one waited for ") + what);
            if (std::chrono::duration<double>(clock::now() - t0).count() > wait_limit_s_) {
                sh_->failed.store(1);
                if (rank == 0) shm_unlink(shm_name_.c_str());  // (no-op once every rank has attached)
                fatal(std::string("Error, ipc world: rank ") + std::to_string(rank) + " waited more than " + std::to_string((int)wait_limit_s_) +
                      " s for " + what + " (a peer probably failed)");
            }
        }
        usleep(spins < 20000 ? 1 : 50);
    }
}

IpcWorld::IpcWorld(int r, int nranks, Backend* backend, int device_ordinal, const std::string& session) {
    rank = r;
    size = nranks;
    if (nranks < 1 || nranks > kIpcMaxRanks || r < 0 || r >= nranks) fatal("Error, the ipc world takes 1 to 16 ranks of one node");
    if (session.empty() || session.size() > 200 || session.find('/') != std::string::npos) fatal("Error, the ipc world needs a session name (no '/')");
    if (const char* w = std::getenv("HNH_IPC_WAIT_S")) wait_limit_s_ = std::atof(w);
    if (const char* m = std::getenv("HNH_IPC_PULL")) {
        if (std::string(m) == "kernel") pull_mode_ = HNH_IPC_PULL_KERNEL;
        else if (std::string(m) != "engine") fatal("Error, HNH_IPC_PULL must be engine or kernel!");
    }
    if (const char* w = std::getenv("HNH_IPC_PULL_WGS")) pull_wgs_ = std::max(1, std::min(256, std::atoi(w)));
    if (const char* w = std::getenv("HNH_IPC_MAX_OPENED")) max_opened_ = (size_t)std::max(1, std::atoi(w));
    init_device(backend, device_ordinal);
    shm_name_ = "/hnh_ipc_" + session;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        int fd = -1;
        if (rank == 0) {
            shm_unlink(shm_name_.c_str());
            fd = shm_open(shm_name_.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)sizeof(IpcShared)) != 0) fatal("Error, ipc world: cannot create the shared-memory segment " + shm_name_);
        } else {
            for (;;) {
                fd = shm_open(shm_name_.c_str(), O_RDWR, 0600);
                struct stat st;
                if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(IpcShared)) break;
                if (fd >= 0) close(fd);
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_limit_s_)
                    fatal("Error, ipc world: rank 0 never created the shared-memory segment " + shm_name_);
                usleep(1000);
            }
        }
        void* m = mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) fatal("Error, ipc world: cannot map the shared-memory segment");
        sh_ = static_cast<IpcShared*>(m);
        if (rank == 0) {
            sh_->nranks = nranks;  // (a fresh segment is zero-filled)
            sh_->creator_pid = (int64_t)getpid();
            sh_->magic.store(kIpcMagic, std::memory_order_release);
            break;
        }
        // A leftover of an earlier launch under the same session name (killed before rank 0 removed the name) can be opened before
        // rank 0 replaces it: its creator is gone, or it already counts a full world.  Drop the mapping and look again.
        bool stale = false;
        const auto t1 = std::chrono::steady_clock::now();
        while (sh_->magic.load(std::memory_order_acquire) != kIpcMagic && !stale) {
            stale = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() > 5.0;  // a creator that never finished
            usleep(200);
        }
        if (!stale) {
            const pid_t creator = (pid_t)sh_->creator_pid;
            stale = creator <= 0 || !pid_is_live((int64_t)creator) || sh_->attached.load() >= sh_->nranks || sh_->failed.load() != 0;
        }
        if (!stale) {
            if (sh_->nranks != nranks) fatal("Error, ipc world: the ranks disagree about the world size");
            break;
        }
        munmap(m, sizeof(IpcShared));
        sh_ = nullptr;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_limit_s_)
            fatal("Error, ipc world: only a stale shared-memory segment " + shm_name_ + " was found (rank 0 never replaced it)");
        usleep(2000);
    }
    sh_->attached.fetch_add(1);
    wait_host([&] { return sh_->attached.load() >= nranks; }, "every rank to attach");
    if (rank == 0) shm_unlink(shm_name_.c_str());  // everybody holds a mapping: the name is no longer needed and cannot leak
    check(be->hnh_ipc_flags_register(ctx, (void*)sh_->flags, sizeof(sh_->flags), &flags_dev_), "hnh_ipc_flags_register");
    msg_out_.assign(nranks, 0);
    msg_in_.assign(nranks, 0);
    for (int s = 0; s < 2; s++) {
        flag_out_[s].assign(nranks, 0);
        flag_in_[s].assign(nranks, 0);
    }
    barrier();
}

IpcWorld::~IpcWorld() {
    if (ctx) {
        sync_all_nothrow();
        if (sh_) {
            // nobody unmaps a peer's memory while that peer may still be reading it
            sh_->detached.fetch_add(1);
            const auto t0 = std::chrono::steady_clock::now();
            while (sh_->detached.load() < size && !sh_->failed.load() &&
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < std::min(wait_limit_s_, 60.0))
                usleep(200);
        }
        for (auto& kv : opened_) be->hnh_ipc_close(ctx, kv.second);
        opened_.clear();
        if (sh_) be->hnh_ipc_flags_unregister(ctx, (void*)sh_->flags);
    }
    destroy_device();
    if (sh_) munmap((void*)sh_, sizeof(IpcShared));
}

void IpcWorld::note_failure() noexcept {
    if (sh_) sh_->failed.store(1);
}

void IpcWorld::barrier() {
    const uint64_t gen = sh_->bar_generation.load(std::memory_order_acquire);
    if (sh_->bar_arrived.fetch_add(1) + 1 == (uint64_t)size) {
        sh_->bar_arrived.store(0);
        sh_->bar_generation.fetch_add(1, std::memory_order_release);
    } else {
        wait_host([&] { return sh_->bar_generation.load(std::memory_order_acquire) != gen; }, "the barrier");
    }
}

void IpcWorld::host_allgather(const void* send, void* recv, size_t bytes) {
    for (size_t off = 0; off < bytes || off == 0; off += kIpcPublishBytes) {
        const size_t chunk = std::min(kIpcPublishBytes, bytes - off);
        std::memcpy(sh_->publish[rank], static_cast<const char*>(send) + off, chunk);
        barrier();
        for (int r = 0; r < size; r++) std::memcpy(static_cast<char*>(recv) + (size_t)r * bytes + off, sh_->publish[r], chunk);
        barrier();
        if (bytes == 0) break;
    }
}

void IpcWorld::host_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl,
                              void* recv, const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) {
    host_alltoallv_staged(send, sendbytes, senddispl, recv, recvbytes, recvdispl);
}

void* IpcWorld::flag(int kind, int stream, int from, int to) const {
    const size_t word = ((((size_t)kind * 2 + (size_t)stream) * (kIpcMaxRanks * kIpcMaxRanks)) + (size_t)from * kIpcMaxRanks + (size_t)to) * 8;
    return static_cast<uint64_t*>(flags_dev_) + word;
}

void IpcWorld::on_device_release(void* p) { exported_.erase((uintptr_t)p); }

const IpcWorld::Exported& IpcWorld::export_of(const void* ptr, uint64_t* offset, Exported* scratch) {
    const uintptr_t a = (uintptr_t)ptr;
    auto it = exported_.upper_bound(a);
    if (it != exported_.begin()) {
        --it;
        if (a < it->first + it->second.bytes) {
            *offset = (uint64_t)(a - it->first);
            return it->second;
        }
    }
    uint64_t alloc = 0;
    check(be->hnh_ipc_export(ctx, ptr, scratch->handle, offset, &alloc), "hnh_ipc_export");
    scratch->bytes = (size_t)alloc;
    void* base = (void*)(a - (uintptr_t)*offset);
    // only blocks this world allocated are remembered: it hears when they go back to the driver (on_device_release)
    if (live_.count(base)) return exported_[(uintptr_t)base] = *scratch;
    return *scratch;
}

void* IpcWorld::open_peer(const unsigned char* handle, uint64_t alloc_bytes) {
    std::string key(reinterpret_cast<const char*>(handle), HNH_IPC_HANDLE_BYTES);
    auto it = opened_.find(key);
    if (it != opened_.end()) return it->second;
    // never evicts: a flush() in progress holds `base + offset` of this group's earlier receives (make_room() runs before the group)
    void* base = nullptr;
    check(be->hnh_ipc_open(ctx, handle, alloc_bytes, &base), "hnh_ipc_open");
    opened_[key] = base;
    return base;
}

// Handles of blocks the peers have long freed pile up.  They are dropped — all of them, behind a drained device — only BETWEEN
// groups: before a flush() that could take the table past its cap even if every one of its receives names a new allocation.
void IpcWorld::make_room(size_t incoming) {
    if (opened_.size() + incoming <= max_opened_) return;
    sync_all();  // nothing of the coming group is enqueued yet: this waits for earlier groups only
    for (auto& kv : opened_) check(be->hnh_ipc_close(ctx, kv.second), "hnh_ipc_close");
    opened_.clear();
}

void IpcWorld::group_begin() { group_depth_++; }
void IpcWorld::group_end() {
    if (group_depth_ <= 0) fatal("Error, group_end without group_begin");
    if (--group_depth_ == 0) flush();
}

void IpcWorld::sendrecv(const Comm& comm, const void* sendbuf, size_t sendbytes, int dst, void* recvbuf, size_t recvbytes, int src,
                        int stream) {
    if (stream != HNH_STREAM_COMPUTE && stream != HNH_STREAM_COMM) fatal("Error, bad stream selector");
    if (!pending_.empty() && pending_.back().stream != stream) flush();  // a group stays on one stream
    pending_.push_back({comm.ranks[dst], comm.ranks[src], sendbuf, sendbytes, recvbuf, recvbytes, stream});
    if (group_depth_ == 0) flush();
}

void IpcWorld::flush() {
    if (pending_.empty()) return;
    std::vector<Op> ops;
    ops.swap(pending_);
    const int stream = ops[0].stream;
    const int M = kIpcMaxRanks;
    size_t incoming = 0;
    for (const Op& o : ops) incoming += (o.src != rank && o.recvbytes) ? 1 : 0;
    make_room(incoming);
    // 1. sends: raise "ready" behind everything enqueued so far, then post where the bytes are
    std::vector<uint64_t> sent(ops.size(), 0), received(ops.size(), 0);
    for (size_t i = 0; i < ops.size(); i++) {
        const Op& o = ops[i];
        if ((o.dst == rank) != (o.src == rank)) fatal("Error, ipc world: a send/recv pair with exactly one end on this rank");
        if (o.dst == rank) {
            if (o.sendbytes != o.recvbytes) fatal("Error, self send/recv size mismatch");
            continue;
        }
        if (!o.sendbytes) continue;
        Exported scratch;
        uint64_t offset = 0;
        const Exported& ex = export_of(o.sendbuf, &offset, &scratch);
        sent[i] = ++flag_out_[stream][o.dst];
        check(be->hnh_stream_write_flag(ctx, stream, flag(0, stream, rank, o.dst), sent[i]), "hnh_stream_write_flag");
        const uint64_t m = ++msg_out_[o.dst];
        const size_t pair = (size_t)rank * M + o.dst;
        if (m > kIpcSlots) wait_host([&] { return sh_->consumed[pair].load(std::memory_order_acquire) + kIpcSlots >= m; }, "a free descriptor slot");
        IpcShared::Msg& slot = sh_->box[pair][m % kIpcSlots];
        slot.offset = offset;
        slot.bytes = o.sendbytes;
        slot.alloc_bytes = ex.bytes;
        slot.stream = stream;
        std::memcpy(slot.handle, ex.handle, HNH_IPC_HANDLE_BYTES);
        slot.seq.store(m, std::memory_order_release);
    }
    // 2. receives: learn where the peer's bytes are, wait (on the stream) until they are final
    std::vector<void*> dsts;
    std::vector<const void*> srcs;
    std::vector<size_t> sizes;
    for (size_t i = 0; i < ops.size(); i++) {
        const Op& o = ops[i];
        if (o.src == rank || !o.recvbytes) continue;
        const uint64_t m = ++msg_in_[o.src];
        const size_t pair = (size_t)o.src * M + rank;
        IpcShared::Msg& slot = sh_->box[pair][m % kIpcSlots];
        wait_host([&] { return slot.seq.load(std::memory_order_acquire) == m; }, "a peer's message descriptor");
        const uint64_t offset = slot.offset, bytes = slot.bytes, alloc = slot.alloc_bytes;
        const int peer_stream = slot.stream;
        unsigned char handle[HNH_IPC_HANDLE_BYTES];
        std::memcpy(handle, slot.handle, HNH_IPC_HANDLE_BYTES);
        sh_->consumed[pair].store(m, std::memory_order_release);
        if (bytes != o.recvbytes) fatal("Error, send/recv size mismatch between ranks");
        if (peer_stream != stream) fatal("Error, ipc world: the two ends of a message name different streams");
        const char* base = static_cast<const char*>(open_peer(handle, alloc));
        received[i] = ++flag_in_[stream][o.src];
        check(be->hnh_stream_wait_flag(ctx, stream, flag(0, stream, o.src, rank), received[i]), "hnh_stream_wait_flag");
        dsts.push_back(o.recvbuf);
        srcs.push_back(base + offset);
        sizes.push_back(o.recvbytes);
    }
    // 3. the copies: self pairs on the stream, the peers' bytes pulled together
    for (const Op& o : ops)
        if (o.dst == rank && o.sendbytes && o.sendbuf != o.recvbuf) copy(o.recvbuf, o.sendbuf, o.sendbytes, HNH_COPY_D2D, stream);
    // the copy engines take their sources from streams forked off the communication stream; on the compute stream (replication
    // collectives) one gather-copy launch does the same without a second set of streams
    const int mode = (stream == HNH_STREAM_COMM) ? pull_mode_ : HNH_IPC_PULL_KERNEL;
    for (size_t lo = 0; lo < dsts.size(); lo += HNH_IPC_MAX_PULL) {
        const int cnt = (int)std::min<size_t>(HNH_IPC_MAX_PULL, dsts.size() - lo);
        check(be->hnh_ipc_pull(ctx, stream, cnt, dsts.data() + lo, srcs.data() + lo, sizes.data() + lo, mode, pull_wgs_), "hnh_ipc_pull");
    }
    // 4. tell the senders their buffers are free again; my own buffers are free once my receivers said so
    for (size_t i = 0; i < ops.size(); i++)
        if (received[i]) check(be->hnh_stream_write_flag(ctx, stream, flag(1, stream, ops[i].src, rank), received[i]), "hnh_stream_write_flag");
    for (size_t i = 0; i < ops.size(); i++)
        if (sent[i]) check(be->hnh_stream_wait_flag(ctx, stream, flag(1, stream, rank, ops[i].dst), sent[i]), "hnh_stream_wait_flag");
}

// ------------------------------------------------------------------------------------------------ CallbackWorld
CallbackWorld::CallbackWorld(int r, int nranks, Backend* backend, int device_ordinal, const hnh_comm_callbacks& cb) : cb_(cb) {
    rank = r;
    size = nranks;
    if (!cb_.sendrecv || !cb_.barrier || !cb_.allgather) fatal("Error, callback transport needs sendrecv, barrier and allgather");
    init_device(backend, device_ordinal);
}
CallbackWorld::~CallbackWorld() { destroy_device(); }

void CallbackWorld::sendrecv(const Comm& comm, const void* sendbuf, size_t sendbytes, int dst, void* recvbuf, size_t recvbytes,
                             int src, int stream) {
    if (comm.ranks[dst] == rank && comm.ranks[src] == rank) {  // ring of one: plain copy
        if (sendbytes != recvbytes) fatal("Error, self send/recv size mismatch");
        if (sendbuf != recvbuf) copy(recvbuf, sendbuf, sendbytes, HNH_COPY_D2D, stream);
        return;
    }
    sync(stream);  // the callback moves data now: everything that produced sendbuf must have finished
    if (cb_.sendrecv(cb_.user, sendbuf, sendbytes, comm.ranks[dst], recvbuf, recvbytes, comm.ranks[src]) != 0)
        fatal("Error, callback sendrecv failed");
}
void CallbackWorld::barrier() {
    if (cb_.barrier(cb_.user) != 0) fatal("Error, callback barrier failed");
}
void CallbackWorld::host_allgather(const void* send, void* recv, size_t bytes) {
    if (cb_.allgather(cb_.user, send, recv, bytes) != 0) fatal("Error, callback allgather failed");
}
void CallbackWorld::host_alltoallv(const void* send, const std::vector<size_t>& sendbytes, const std::vector<size_t>& senddispl,
                                   void* recv, const std::vector<size_t>& recvbytes, const std::vector<size_t>& recvdispl) {
    // host buffers: pairwise rounds straight through the callback (it is given host pointers here)
    for (int k = 0; k < size; k++) {
        const int dst = (rank + k) % size, src = (rank - k + size) % size;
        const char* s = static_cast<const char*>(send) + senddispl[dst];
        char* r = static_cast<char*>(recv) + recvdispl[src];
        if (k == 0) {
            if (sendbytes[rank]) std::memcpy(r, s, sendbytes[rank]);
        } else if (cb_.sendrecv(cb_.user, s, sendbytes[dst], dst, r, recvbytes[src], src) != 0) {
            fatal("Error, callback sendrecv failed");
        }
    }
}

}  // namespace hnh
