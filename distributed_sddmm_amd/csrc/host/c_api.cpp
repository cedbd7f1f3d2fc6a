// C ABI of the host layer (include/hnh_dist.h): thin handle wrappers over the C++ classes.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <memory>
#include <omp.h>
#include <string>

#include "als_conjugate_gradients.hpp"
#include "cannon_dense_25d.hpp"
#include "cannon_sparse_25d.hpp"
#include "dense_shift_15d.hpp"
#include "er_generator.hpp"
#include "gat.hpp"
#include "hnh_dist.h"
#include "sparse_shift_15d.hpp"

struct hnh_thread_group {
    std::shared_ptr<hnh::ThreadGroup> g;
};
struct hnh_world {
    std::unique_ptr<hnh::World> w;
};
struct hnh_spmat {
    hnh::World* w;
    std::unique_ptr<SpmatLocal> s;
};
struct hnh_dense {
    hnh::World* w;
    DenseMatrix m;
};
struct hnh_vec {
    hnh::World* w;
    VectorXd v;
};
struct hnh_dist {
    hnh::World* w;
    std::unique_ptr<StandardKernel> kernel;
    std::unique_ptr<Distributed_Sparse> d;
};

struct hnh_als {
    hnh::World* w;
    std::unique_ptr<Distributed_ALS> a;
};

struct hnh_gat {
    hnh::World* w;
    std::unique_ptr<GAT> g;
};

namespace {
thread_local std::string t_error;

template <typename F>
int guarded(hnh::World* w, F&& f) {
    try {
        hnh::set_throw_on_error(true);
        if (w) hnh::set_current_world(w);
        f();
        return HNH_OK;
    } catch (const hnh::Error& e) {
        t_error = e.what();
        return HNH_ERR_INVALID;
    } catch (const std::bad_alloc&) {
        t_error = "out of host memory";
        return HNH_ERR_NOMEM;
    } catch (const std::exception& e) {
        t_error = e.what();
        return HNH_ERR_DEVICE;
    }
}
struct ErKeys {
    uint64_t n;
    std::vector<uint64_t> keys;
};
}  // namespace

extern "C" {

const char* hnh_host_last_error(void) { return t_error.c_str(); }

int hnh_backend_load(const char* path) {
    return guarded(nullptr, [&] { hnh::load_backend(path); });
}
const char* hnh_host_backend_name(void) {
    static thread_local std::string name;
    if (guarded(nullptr, [&] { name = hnh::default_backend()->name; }) != HNH_OK) return "";
    return name.c_str();
}

// ------------------------------------------------------------------ worlds
int hnh_world_create_single(int device, hnh_world** out) {
    return guarded(nullptr, [&] {
        auto* h = new hnh_world();
        h->w.reset(new hnh::SingleWorld(hnh::default_backend(), device));
        *out = h;
    });
}
int hnh_thread_group_create(int nranks, hnh_thread_group** out) {
    return guarded(nullptr, [&] {
        if (nranks < 1) hnh::fatal("Error, a thread group needs at least one rank");
        auto* h = new hnh_thread_group();
        h->g = hnh::make_thread_group(nranks);
        *out = h;
    });
}
int hnh_thread_group_destroy(hnh_thread_group* g) {
    delete g;
    return HNH_OK;
}
int hnh_world_create_thread(hnh_thread_group* g, int rank, int device, hnh_world** out) {
    return guarded(nullptr, [&] {
        auto* h = new hnh_world();
        h->w.reset(new hnh::ThreadWorld(g->g, rank, hnh::default_backend(), device));
        *out = h;
    });
}
int hnh_rccl_unique_id(void* id) {
    return guarded(nullptr, [&] {
        if (hnh::default_backend()->hnh_comm_unique_id(id) != HNH_OK) hnh::fatal("Error, cannot create an RCCL unique id");
    });
}
int hnh_world_create_rccl(int rank, int nranks, int device, const void* id, hnh_world** out) {
    return guarded(nullptr, [&] {
        auto* h = new hnh_world();
        h->w.reset(new hnh::RcclWorld(rank, nranks, hnh::default_backend(), device, id));
        *out = h;
    });
}
int hnh_world_create_ipc(int rank, int nranks, int device, const char* session, hnh_world** out) {
    return guarded(nullptr, [&] {
        auto* h = new hnh_world();
        h->w.reset(new hnh::IpcWorld(rank, nranks, hnh::default_backend(), device, session ? session : ""));
        *out = h;
    });
}
int hnh_world_create_callback(int rank, int nranks, int device, const hnh_comm_callbacks* cb, hnh_world** out) {
    return guarded(nullptr, [&] {
        auto* h = new hnh_world();
        h->w.reset(new hnh::CallbackWorld(rank, nranks, hnh::default_backend(), device, *cb));
        *out = h;
    });
}
int hnh_world_destroy(hnh_world* w) {
    return guarded(nullptr, [&] {
        if (w && hnh::current_world_or_null() == w->w.get()) hnh::set_current_world(nullptr);
        delete w;
    });
}
int hnh_world_rank(hnh_world* w) { return w->w->rank; }
int hnh_world_size(hnh_world* w) { return w->w->size; }
int hnh_world_barrier(hnh_world* w) {
    return guarded(w->w.get(), [&] { w->w->barrier(); });
}
int hnh_world_sync(hnh_world* w) {
    return guarded(w->w.get(), [&] { w->w->sync_all(); });
}
int hnh_world_set_solo(hnh_world* w, int on) {
    return guarded(w->w.get(), [&] { w->w->set_solo(on != 0); });
}
int hnh_world_set_timing_sync(hnh_world* w, int on) {
    w->w->timing_sync = on != 0;
    return HNH_OK;
}
void* hnh_world_stream(hnh_world* w, int stream) { return w->w->be->hnh_ctx_stream(w->w->ctx, stream); }
hnh_ctx* hnh_world_ctx(hnh_world* w) { return w->w->ctx; }

int hnh_world_grid_probe(hnh_world* w, int nr, int nc, int nh, int adjacency, int* out9, int* ok) {
    return guarded(w->w.get(), [&] {
        FlexibleGrid g(nr, nc, nh, adjacency);
        const int vals[9] = {g.i, g.j, g.k, g.rankInRow, g.rankInCol, g.rankInFiber, g.row_world.size(), g.col_world.size(),
                             g.fiber_world.size()};
        std::memcpy(out9, vals, sizeof(vals));
        *ok = g.self_test(false) ? 1 : 0;
    });
}

int hnh_world_identities(hnh_world* wh, hnh_rank_identity* out) {
    if (!wh || !out) return HNH_ERR_INVALID;
    return guarded(wh->w.get(), [&] {
        const std::vector<hnh_rank_identity> all = wh->w->identities();
        std::copy(all.begin(), all.end(), out);
    });
}

int hnh_world_split_signature(hnh_world* w, uint64_t* signature, int* count) {
    if (!w || !signature || !count) return HNH_ERR_INVALID;
    *signature = w->w->split_signature;
    *count = w->w->split_count;
    return HNH_OK;
}

// Transport self-test: one communication primitive on small device buffers with known contents, verified on the host.
// Every rank of the world calls it with the same arguments.  *max_err = largest deviation from the expected values.
int hnh_world_preflight(hnh_world* wh, int what, int64_t count, double* max_err) {
    return guarded(wh->w.get(), [&] {
        hnh::World* w = wh->w.get();
        const int p = w->size, me = w->rank;
        if (count < 1) hnh::fatal("Error, preflight needs a positive element count!");
        const size_t n = (size_t)count;
        auto val = [](int rank, size_t i, int salt) { return (double)(rank + 1) * 1000.0 + (double)(i % 997) + 0.25 * salt; };
        auto upload = [&](double* d, const std::vector<double>& h) {
            w->copy(d, h.data(), h.size() * sizeof(double), HNH_COPY_H2D, HNH_STREAM_COMM);
        };
        auto download = [&](const double* d, size_t m) {
            std::vector<double> h(m);
            w->copy(h.data(), d, m * sizeof(double), HNH_COPY_D2H, HNH_STREAM_COMM);
            w->sync(HNH_STREAM_COMM);
            return h;
        };
        double err = 0.0;
        auto expect = [&](double got, double want) { err = std::max(err, std::abs(got - want)); };
        hnh::Comm world = w->world_comm();
        // a sub-communicator like the schedules' layer communicator: pairs (c = 2) when p is even, else everybody
        const int c = (p % 2 == 0) ? 2 : 1;
        hnh::Comm layer = w->split(me / c, me % c);
        hnh::Comm ring = w->split(me % c, me / c);
        switch (what) {
            case HNH_PREFLIGHT_RING: {  // relay step: send to ring rank +1, receive from -1 (distributed_sparse.h:351-361)
                hnh::DeviceArray s(w, n * 8), r(w, n * 8);
                std::vector<double> h(n);
                for (size_t i = 0; i < n; i++) h[i] = val(me, i, 1);
                upload((double*)s.ptr(), h);
                const int rn = ring.size();
                w->sendrecv(ring, s.ptr(), n * 8, (ring.me + 1) % rn, r.ptr(), n * 8, (ring.me - 1 + rn) % rn, HNH_STREAM_COMM);
                auto got = download((double*)r.ptr(), n);
                const int src = ring.ranks[(ring.me - 1 + rn) % rn];
                for (size_t i = 0; i < n; i++) expect(got[i], val(src, i, 1));
                break;
            }
            case HNH_PREFLIGHT_MESH: {  // all n - 1 owner -> consumer transfers as ONE group (mesh fetch)
                const int rn = ring.size();
                hnh::DeviceArray s(w, n * 8), r(w, n * 8 * (size_t)std::max(rn - 1, 1));
                std::vector<double> h(n);
                for (size_t i = 0; i < n; i++) h[i] = val(me, i, 2);
                upload((double*)s.ptr(), h);
                w->group_begin();
                for (int k = 1; k < rn; k++)
                    w->sendrecv(ring, s.ptr(), n * 8, (ring.me + k) % rn, (double*)r.ptr() + (size_t)(k - 1) * n, n * 8,
                                (ring.me - k + rn) % rn, HNH_STREAM_COMM);
                w->group_end();
                auto got = download((double*)r.ptr(), n * (size_t)std::max(rn - 1, 1));
                for (int k = 1; k < rn; k++) {
                    const int src = ring.ranks[(ring.me - k + rn) % rn];
                    for (size_t i = 0; i < n; i++) expect(got[(size_t)(k - 1) * n + i], val(src, i, 2));
                }
                break;
            }
            case HNH_PREFLIGHT_ALLGATHER:
            case HNH_PREFLIGHT_ALLGATHER_WORLD: {
                hnh::Comm& cm = (what == HNH_PREFLIGHT_ALLGATHER) ? layer : world;
                const int m = cm.size();
                hnh::DeviceArray s(w, n * 8), r(w, n * 8 * (size_t)m);
                std::vector<double> h(n);
                for (size_t i = 0; i < n; i++) h[i] = val(me, i, 3);
                upload((double*)s.ptr(), h);
                w->allgather(cm, s.ptr(), r.ptr(), n * 8, HNH_STREAM_COMM);
                auto got = download((double*)r.ptr(), n * (size_t)m);
                for (int j = 0; j < m; j++)
                    for (size_t i = 0; i < n; i++) expect(got[(size_t)j * n + i], val(cm.ranks[j], i, 3));
                break;
            }
            case HNH_PREFLIGHT_REDUCE_SCATTER:
            case HNH_PREFLIGHT_REDUCE_SCATTER_WORLD: {
                hnh::Comm& cm = (what == HNH_PREFLIGHT_REDUCE_SCATTER) ? layer : world;
                const int m = cm.size();
                hnh::DeviceArray s(w, n * 8 * (size_t)m), r(w, n * 8);
                std::vector<double> h(n * (size_t)m);
                for (int j = 0; j < m; j++)
                    for (size_t i = 0; i < n; i++) h[(size_t)j * n + i] = val(me, i, 4 + j);
                upload((double*)s.ptr(), h);
                w->reduce_scatter_f64(cm, (const double*)s.ptr(), (double*)r.ptr(), n, HNH_STREAM_COMM);
                auto got = download((double*)r.ptr(), n);
                for (size_t i = 0; i < n; i++) {
                    double want = 0.0;
                    for (int j = 0; j < m; j++) want += val(cm.ranks[j], i, 4 + cm.me);
                    expect(got[i], want);
                }
                break;
            }
            case HNH_PREFLIGHT_ALLREDUCE: {
                hnh::DeviceArray b(w, n * 8);
                std::vector<double> h(n);
                for (size_t i = 0; i < n; i++) h[i] = val(me, i, 5);
                upload((double*)b.ptr(), h);
                w->allreduce_f64(layer, (double*)b.ptr(), n, HNH_STREAM_COMM);
                auto got = download((double*)b.ptr(), n);
                for (size_t i = 0; i < n; i++) {
                    double want = 0.0;
                    for (int j = 0; j < layer.size(); j++) want += val(layer.ranks[j], i, 5);
                    expect(got[i], want);
                }
                break;
            }
            case HNH_PREFLIGHT_VARIABLE: {  // allgatherv + reduce_scatter_v with ragged counts (25D_cannon_sparse.hpp:224-233,294-300)
                hnh::Comm& cm = ring;
                const int m = cm.size();
                std::vector<int> counts(m), displs(m);
                int total = 0;
                for (int j = 0; j < m; j++) {
                    counts[j] = (int)std::min<size_t>(n, 1 + (size_t)(j * 37 + 11) % n);
                    displs[j] = total;
                    total += counts[j];
                }
                hnh::DeviceArray s(w, (size_t)counts[cm.me] * 8), r(w, (size_t)total * 8), r2(w, (size_t)counts[cm.me] * 8);
                std::vector<double> h((size_t)counts[cm.me]);
                for (size_t i = 0; i < h.size(); i++) h[i] = val(me, i, 6);
                upload((double*)s.ptr(), h);
                w->allgatherv_f64(cm, (const double*)s.ptr(), h.size(), (double*)r.ptr(), counts, displs, HNH_STREAM_COMM);
                auto got = download((double*)r.ptr(), (size_t)total);
                for (int j = 0; j < m; j++)
                    for (int i = 0; i < counts[j]; i++) expect(got[(size_t)displs[j] + i], val(cm.ranks[j], (size_t)i, 6));
                // every member contributes the gathered vector scaled by (its comm index + 1)
                std::vector<double> contrib((size_t)total);
                for (int t = 0; t < total; t++) contrib[t] = got[t] * (cm.me + 1);
                upload((double*)r.ptr(), contrib);
                w->reduce_scatter_v_f64(cm, (const double*)r.ptr(), (double*)r2.ptr(), counts, HNH_STREAM_COMM);
                auto red = download((double*)r2.ptr(), (size_t)counts[cm.me]);
                const double scale = 0.5 * m * (m + 1);
                for (int i = 0; i < counts[cm.me]; i++) expect(red[i] / scale, val(me, (size_t)i, 6));
                break;
            }
            case HNH_PREFLIGHT_ALLTOALLV: {  // the setup pipeline's device exchange (SpmatLocal.hpp:389-462)
                std::vector<size_t> sb(p), sd(p), rb(p), rd(p);
                size_t st = 0, rt = 0;
                for (int j = 0; j < p; j++) {
                    sb[j] = 8 * (1 + (size_t)(me * 7 + j * 3) % n);
                    sd[j] = st; st += sb[j];
                    rb[j] = 8 * (1 + (size_t)(j * 7 + me * 3) % n);
                    rd[j] = rt; rt += rb[j];
                }
                hnh::DeviceArray s(w, st), r(w, rt);
                std::vector<double> h(st / 8);
                for (int j = 0; j < p; j++)
                    for (size_t i = 0; i < sb[j] / 8; i++) h[sd[j] / 8 + i] = val(me, i, 7 + j);
                upload((double*)s.ptr(), h);
                w->sync(HNH_STREAM_COMM);
                w->device_alltoallv(s.ptr(), sb, sd, r.ptr(), rb, rd, HNH_STREAM_COMM);
                auto got = download((double*)r.ptr(), rt / 8);
                for (int j = 0; j < p; j++)
                    for (size_t i = 0; i < rb[j] / 8; i++) expect(got[rd[j] / 8 + i], val(j, i, 7 + me));
                break;
            }
            default:
                hnh::fatal("Error, unknown preflight test!");
        }
        w->sync_all();
        *max_err = err;
    });
}

// ------------------------------------------------------------------ sparse input
int hnh_spmat_create(hnh_world* w, int64_t M, int64_t N, int64_t dist_nnz, int64_t local_nnz, const int64_t* rows,
                     const int64_t* cols, const double* values, hnh_spmat** out) {
    return guarded(w->w.get(), [&] {
        auto* h = new hnh_spmat();
        h->w = w->w.get();
        h->s.reset(new SpmatLocal());
        h->s->M = (uint64_t)M;
        h->s->N = (uint64_t)N;
        h->s->dist_nnz = (uint64_t)dist_nnz;
        h->s->coords.resize((size_t)local_nnz);
        for (int64_t e = 0; e < local_nnz; e++) {
            if (rows[e] < 0 || rows[e] >= M || cols[e] < 0 || cols[e] >= N) hnh::fatal("Error, tuple outside the matrix!");
            h->s->coords[e] = {(uint64_t)rows[e], (uint64_t)cols[e], values ? values[e] : 1.0};
        }
        h->s->initialized = true;
        *out = h;
    });
}
int hnh_spmat_load_tuples(hnh_world* w, int read_from_file, int logM, int nnz_per_row, const char* filename, hnh_spmat** out) {
    return guarded(w->w.get(), [&] {
        auto* h = new hnh_spmat();
        h->w = w->w.get();
        h->s.reset(new SpmatLocal());
        h->s->loadTuples(read_from_file != 0, logM, nnz_per_row, filename ? filename : "");
        *out = h;
    });
}
int hnh_spmat_info(hnh_spmat* s, int64_t out4[4]) {
    out4[0] = (int64_t)s->s->M;
    out4[1] = (int64_t)s->s->N;
    out4[2] = (int64_t)s->s->dist_nnz;
    out4[3] = (int64_t)s->s->num_tuples();
    return HNH_OK;
}
int hnh_spmat_permute(hnh_spmat* s, uint64_t seed) {
    return guarded(s->w, [&] { s->s->permuteVertices(seed); });
}
int hnh_spmat_destroy(hnh_spmat* s) {
    return guarded(s ? s->w : nullptr, [&] { delete s; });
}
int hnh_er_generate(uint64_t m, uint64_t n, uint64_t draws, uint64_t seed, void** handle, int64_t* count) {
    return guarded(nullptr, [&] {
        auto* k = new ErKeys{n, hnh::erdos_renyi_keys(m, n, draws, seed)};
        *count = (int64_t)k->keys.size();
        *handle = k;
    });
}
int hnh_rmat_generate(int logm, uint64_t edges, double a, double b, double c, uint64_t seed, int scramble, void** handle,
                      int64_t* count) {
    return guarded(nullptr, [&] {
        if (logm < 1 || logm > 31 || a < 0 || b < 0 || c < 0 || a + b + c > 1.0) hnh::fatal("Error, bad R-MAT parameters");
        auto* k = new ErKeys{1ull << logm, hnh::rmat_keys(logm, edges, a, b, c, seed, scramble != 0)};
        *count = (int64_t)k->keys.size();
        *handle = k;
    });
}
int hnh_er_fetch(void* handle, int64_t* rows, int64_t* cols) {
    ErKeys* k = static_cast<ErKeys*>(handle);
    const uint64_t n = k->n;
#pragma omp parallel for
    for (size_t e = 0; e < k->keys.size(); e++) {
        rows[e] = (int64_t)(k->keys[e] / n);
        cols[e] = (int64_t)(k->keys[e] % n);
    }
    delete k;
    return HNH_OK;
}
// A MatrixMarket coordinate file of the given entries (1-based on disk), written by all host cores: every thread formats its slice
// of the lines, the slices go out in order.  Tests and benchmarks of the input side (SpmatLocal::loadTuples(readFromFile = true)).
int hnh_write_matrix_market(const char* path, int64_t M, int64_t N, int64_t n, const int64_t* rows, const int64_t* cols, const double* values,
                            int symmetric) {
    return guarded(nullptr, [&] {
        if (!path || M < 1 || N < 1 || n < 0 || (n && (!rows || !cols))) hnh::fatal("Error, bad arguments for the MatrixMarket writer");
        FILE* f = std::fopen(path, "wb");
        if (!f) hnh::fatal(std::string("Error, cannot create ") + path);
        std::fprintf(f, "%%%%MatrixMarket matrix coordinate real %s\n%% written by hnh_write_matrix_market\n%lld %lld %lld\n", symmetric ? "symmetric" : "general",
                     (long long)M, (long long)N, (long long)n);
        // slices of 2^18 entries (at most ~45 bytes each: 12 MB of text per slice), as many side by side as there are threads, written in
        // order.  Nothing may leave the parallel region as an exception (std::terminate): an allocation failure or an entry outside the
        // matrix sets a flag, the slice stops, and the error is reported after the region.
        const int64_t slice = 1 << 18;
        const int64_t nslices = (n + slice - 1) / slice;
        const int64_t batch = std::max(1, omp_get_max_threads());
        bool ok = true;
        int failure = 0;  // 1 = out of memory, 2 = an entry outside the matrix
        for (int64_t s0 = 0; s0 < nslices && ok && !failure; s0 += batch) {
            const int64_t s1 = std::min(nslices, s0 + batch);
            std::vector<std::string> text((size_t)(s1 - s0));
#pragma omp parallel for schedule(dynamic, 1)
            for (int64_t s = s0; s < s1; s++) {
                try {
                    std::string& t = text[(size_t)(s - s0)];
                    t.reserve((size_t)slice * 24);
                    char line[96];
                    for (int64_t e = s * slice; e < std::min(n, (s + 1) * slice); e++) {
                        if (rows[e] < 0 || rows[e] >= M || cols[e] < 0 || cols[e] >= N) {
#pragma omp atomic write
                            failure = 2;
                            break;
                        }
                        int len;
                        if (values) len = std::snprintf(line, sizeof(line), "%lld %lld %.17g\n", (long long)rows[e] + 1, (long long)cols[e] + 1, values[e]);
                        else len = std::snprintf(line, sizeof(line), "%lld %lld 1\n", (long long)rows[e] + 1, (long long)cols[e] + 1);
                        t.append(line, (size_t)len);
                    }
                } catch (...) {  // (std::bad_alloc of reserve / append)
#pragma omp atomic write
                    failure = 1;
                }
            }
            if (!failure)
                for (const std::string& t : text) ok = ok && std::fwrite(t.data(), 1, t.size(), f) == t.size();
        }
        const bool closed = std::fclose(f) == 0;
        if (failure == 1) throw std::bad_alloc();  // -> HNH_ERR_NOMEM through guarded()
        if (failure == 2) hnh::fatal(std::string("Error, an entry of the MatrixMarket file ") + path + " lies outside the matrix");
        if (!closed || !ok) hnh::fatal(std::string("Error, writing ") + path + " failed");
    });
}

// ------------------------------------------------------------------ operator
int hnh_dist_create(hnh_world* w, const char* alg_c, hnh_spmat* s, int R, int c, hnh_dist** out) {
    return guarded(w->w.get(), [&] {
        const std::string alg(alg_c ? alg_c : "");
        std::unique_ptr<hnh_dist> h(new hnh_dist());
        h->w = w->w.get();
        h->kernel.reset(new StandardKernel());
        // name -> class exactly as benchmark_dist.cpp:45-82
        if (alg == "15d_fusion1") h->d.reset(new Sparse15D_Dense_Shift(s->s.get(), R, c, 1, h->kernel.get()));
        else if (alg == "15d_fusion2") h->d.reset(new Sparse15D_Dense_Shift(s->s.get(), R, c, 2, h->kernel.get()));
        else if (alg == "15d_sparse") h->d.reset(new Sparse15D_Sparse_Shift(s->s.get(), R, c, h->kernel.get()));
        else if (alg == "25d_dense_replicate") h->d.reset(new Sparse25D_Cannon_Dense(s->s.get(), R, c, h->kernel.get()));
        else if (alg == "25d_sparse_replicate") h->d.reset(new Sparse25D_Cannon_Sparse(s->s.get(), R, c, h->kernel.get()));
        else hnh::fatal("Error, unknown algorithm name " + alg);
        *out = h.release();
    });
}
int hnh_dist_destroy(hnh_dist* d) {
    return guarded(d ? d->w : nullptr, [&] { delete d; });
}
int hnh_dist_info(hnh_dist* d, int64_t o[16]) {
    return guarded(d->w, [&] {
        Distributed_Sparse* x = d->d.get();
        o[0] = x->M; o[1] = x->N; o[2] = x->R; o[3] = x->p; o[4] = x->c;
        o[5] = x->localArows; o[6] = x->localAcols; o[7] = x->localBrows; o[8] = x->localBcols;
        o[9] = x->like_S_values(0.0).size();
        o[10] = x->like_ST_values(0.0).size();
        o[11] = x->r_split ? 1 : 0;
        o[12] = (int64_t)x->S->dist_nnz;
        o[13] = x->proc_rank;
        o[14] = (int64_t)x->aSubmatrices.size();
        o[15] = (int64_t)x->bSubmatrices.size();
    });
}
int hnh_dist_submatrices(hnh_dist* d, int matmode, int64_t* out, int capacity) {
    return guarded(d->w, [&] {
        auto& subs = matmode == HNH_AMAT ? d->d->aSubmatrices : d->d->bSubmatrices;
        if ((int)subs.size() > capacity) hnh::fatal("Error, submatrix buffer too small");
        for (size_t i = 0; i < subs.size(); i++) {
            out[4 * i] = subs[i].topRow; out[4 * i + 1] = subs[i].leftCol;
            out[4 * i + 2] = subs[i].rowCount; out[4 * i + 3] = subs[i].colCount;
        }
    });
}
int hnh_dist_set_r(hnh_dist* d, int R) {
    return guarded(d->w, [&] { d->d->setRValue(R); });
}
int hnh_dist_json(hnh_dist* d, int which, char* buf, size_t capacity) {
    return guarded(d->w, [&] {
        std::string s = (which == 0 ? d->d->json_algorithm_info() : d->d->json_perf_statistics()).dump();
        if (s.size() + 1 > capacity) hnh::fatal("Error, JSON buffer too small");
        std::memcpy(buf, s.c_str(), s.size() + 1);
    });
}
int hnh_dist_reset_timers(hnh_dist* d) {
    return guarded(d->w, [&] { d->d->reset_performance_timers(); });
}
int hnh_dist_kernel_profile(hnh_dist* d, int enable, double* total_ms, int64_t* launches) {
    {
        const int st = guarded(d->w, [&] { d->kernel->resolve_profile(); });  // the pairs recorded so far are read here, not inside the calls
        if (st != HNH_OK) return st;
    }
    if (total_ms) *total_ms = d->kernel->kernel_ms;
    if (launches) *launches = d->kernel->kernel_launches;
    if (enable >= 0) {
        d->kernel->profile = enable != 0;
        d->kernel->kernel_ms = 0.0;
        d->kernel->kernel_launches = 0;
    }
    return HNH_OK;
}

int hnh_dist_borrow_stats(hnh_dist* d, int64_t out4[4]) {
    if (!d || !out4) return HNH_ERR_INVALID;
    for (int k = 0; k < 4; k++) out4[k] = d->d->S->borrow_stats[k] + d->d->ST->borrow_stats[k];
    return HNH_OK;
}

// ------------------------------------------------------------------ dense / vectors
int hnh_dense_create(hnh_world* w, int64_t rows, int64_t cols, double fill, hnh_dense** out) {
    return guarded(w->w.get(), [&] {
        auto* h = new hnh_dense{w->w.get(), DenseMatrix::Constant(rows, cols, fill)};
        *out = h;
    });
}
int hnh_dense_wrap(hnh_world* w, void* ptr, int64_t rows, int64_t cols, hnh_dense** out) {
    return guarded(w->w.get(), [&] {
        auto* h = new hnh_dense{w->w.get(), DenseMatrix::view(static_cast<double*>(ptr), rows, cols)};
        *out = h;
    });
}
int hnh_dense_like(hnh_dist* d, int matmode, double fill, hnh_dense** out) {
    return guarded(d->w, [&] {
        auto* h = new hnh_dense{d->w, matmode == HNH_AMAT ? d->d->like_A_matrix(fill) : d->d->like_B_matrix(fill)};
        *out = h;
    });
}
int hnh_dense_shape(hnh_dense* m, int64_t o[2]) {
    o[0] = m->m.rows();
    o[1] = m->m.cols();
    return HNH_OK;
}
void* hnh_dense_data(hnh_dense* m) { return m->m.data(); }
int hnh_dense_upload(hnh_dense* m, const double* host) {
    return guarded(m->w, [&] { m->m.copy_from_host(host); });
}
int hnh_dense_download(hnh_dense* m, double* host) {
    return guarded(m->w, [&] { m->m.copy_to_host(host); });
}
int hnh_dense_fill(hnh_dense* m, double v) {
    return guarded(m->w, [&] { m->m.setConstant(v); });
}
int hnh_dense_copy(hnh_dense* dst, hnh_dense* src) {
    return guarded(dst->w, [&] { dst->m = src->m; });
}
int hnh_dense_destroy(hnh_dense* m) {
    return guarded(m ? m->w : nullptr, [&] { delete m; });
}
int hnh_dense_dummy_initialize(hnh_dist* d, hnh_dense* m, int matmode) {
    return guarded(d->w, [&] { d->d->dummyInitialize(m->m, matmode == HNH_AMAT ? Amat : Bmat); });
}
int hnh_vec_create(hnh_world* w, int64_t n, double fill, hnh_vec** out) {
    return guarded(w->w.get(), [&] {
        auto* h = new hnh_vec{w->w.get(), VectorXd::Constant(n, fill)};
        *out = h;
    });
}
int hnh_vec_like(hnh_dist* d, int which, double fill, hnh_vec** out) {
    return guarded(d->w, [&] {
        auto* h = new hnh_vec{d->w, which == 0 ? d->d->like_S_values(fill) : d->d->like_ST_values(fill)};
        *out = h;
    });
}
int64_t hnh_vec_size(hnh_vec* v) { return v->v.size(); }
void* hnh_vec_data(hnh_vec* v) { return v->v.data(); }
int hnh_vec_upload(hnh_vec* v, const double* host) {
    return guarded(v->w, [&] { if (v->v.size()) v->v.copy_from_host(host); });
}
int hnh_vec_download(hnh_vec* v, double* host) {
    return guarded(v->w, [&] { if (v->v.size()) v->v.copy_to_host(host); });
}
int hnh_vec_fill(hnh_vec* v, double value) {
    return guarded(v->w, [&] { v->v.setConstant(value); });
}
int hnh_vec_destroy(hnh_vec* v) {
    return guarded(v ? v->w : nullptr, [&] { delete v; });
}

// ------------------------------------------------------------------ operations
static KernelMode kmode(int m) {
    switch (m) {
        case HNH_K_SDDMM_A: return k_sddmmA;
        case HNH_K_SPMM_A: return k_spmmA;
        case HNH_K_SPMM_B: return k_spmmB;
        case HNH_K_SDDMM_B: return k_sddmmB;
    }
    hnh::fatal("Error, bad kernel mode");
}
int hnh_dist_initial_shift(hnh_dist* d, hnh_dense* A, hnh_dense* B, int mode) {
    return guarded(d->w, [&] { d->d->initial_shift(A ? &A->m : nullptr, B ? &B->m : nullptr, kmode(mode)); });
}
int hnh_dist_de_shift(hnh_dist* d, hnh_dense* A, hnh_dense* B, int mode) {
    return guarded(d->w, [&] { d->d->de_shift(A ? &A->m : nullptr, B ? &B->m : nullptr, kmode(mode)); });
}
int hnh_dist_sddmmA(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S, hnh_vec* result) {
    return guarded(d->w, [&] { d->d->sddmmA(A->m, B->m, S->v, result->v); });
}
int hnh_dist_sddmmB(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S, hnh_vec* result) {
    return guarded(d->w, [&] { d->d->sddmmB(A->m, B->m, S->v, result->v); });
}
int hnh_dist_spmmA(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S) {
    return guarded(d->w, [&] { d->d->spmmA(A->m, B->m, S->v); });
}
int hnh_dist_spmmB(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S) {
    return guarded(d->w, [&] { d->d->spmmB(A->m, B->m, S->v); });
}
int hnh_dist_fusedSpMM(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S, hnh_vec* buf, int matmode) {
    return guarded(d->w, [&] { d->d->fusedSpMM(A->m, B->m, S->v, buf->v, matmode == HNH_AMAT ? Amat : Bmat); });
}
int hnh_dist_hold_moving_operand(hnh_dist* d, hnh_dense* m) {
    return guarded(d->w, [&] {
        if (m) d->d->hold_moving_operand(&m->m);
        else d->d->release_moving_operand();
    });
}
int hnh_dist_walk_windows_when_held(hnh_dist* d, int on) {
    return guarded(d->w, [&] { d->d->walk_windows_when_held(on); });
}
int hnh_dist_fusedSpMM_out(hnh_dist* d, hnh_dense* A, hnh_dense* B, int matmode, hnh_dense* Out, int leaky, double leaky_alpha,
                           double x_scale, hnh_vec* rowdot, int* supported) {
    return guarded(d->w, [&] {
        hnh_fused_extras ex = {leaky_alpha, x_scale, rowdot ? rowdot->v.data() : nullptr, nullptr, nullptr, 0};
        const bool ok = d->d->fusedSpMM_out(A->m, B->m, matmode == HNH_AMAT ? Amat : Bmat, Out->m, leaky != 0, ex);
        if (supported) *supported = ok ? 1 : 0;
    });
}
int hnh_dist_algorithm(hnh_dist* d, hnh_dense* A, hnh_dense* B, hnh_vec* S, hnh_vec* result, int mode, int initial_replicate) {
    return guarded(d->w, [&] { d->d->algorithm(A->m, B->m, S->v, result ? &result->v : nullptr, kmode(mode), initial_replicate != 0); });
}

// ------------------------------------------------------------------ ALS-CG
int hnh_als_create(hnh_dist* d, int artificial_groundtruth, uint64_t seed, hnh_als** out) {
    return guarded(d->w, [&] {
        std::unique_ptr<hnh_als> h(new hnh_als());
        h->w = d->w;
        h->a.reset(new Distributed_ALS(d->d.get(), artificial_groundtruth != 0, seed));
        *out = h.release();
    });
}
int hnh_als_destroy(hnh_als* a) {
    return guarded(a ? a->w : nullptr, [&] { delete a; });
}
int hnh_als_set_ground_truth(hnh_als* a, hnh_vec* gs, hnh_vec* gst) {
    return guarded(a->w, [&] {
        VectorXd s = a->a->d_ops->like_S_values(0.0), st = a->a->d_ops->like_ST_values(0.0);
        if (gs->v.size() != s.size() || gst->v.size() != st.size()) hnh::fatal("Error, ground truth vectors have the wrong length!");
        a->w->copy(s.data(), gs->v.data(), (size_t)s.size() * sizeof(double), HNH_COPY_D2D, HNH_STREAM_COMPUTE);
        a->w->copy(st.data(), gst->v.data(), (size_t)st.size() * sizeof(double), HNH_COPY_D2D, HNH_STREAM_COMPUTE);
        a->a->ground_truth = std::move(s);
        a->a->ground_truth_transpose = std::move(st);
    });
}
int hnh_als_initialize_embeddings(hnh_als* a) {
    return guarded(a->w, [&] { a->a->initializeEmbeddings(); });
}
int hnh_als_set_embeddings(hnh_als* a, hnh_dense* A, hnh_dense* B) {
    return guarded(a->w, [&] {
        a->a->A = A->m;
        a->a->B = B->m;
    });
}
int hnh_als_get_embeddings(hnh_als* a, hnh_dense* A, hnh_dense* B) {
    return guarded(a->w, [&] {
        if (A) A->m = a->a->A;
        if (B) B->m = a->a->B;
    });
}
int hnh_als_cg_optimizer(hnh_als* a, int matmode, int cg_max_iter) {
    return guarded(a->w, [&] { a->a->cg_optimizer(matmode == HNH_AMAT ? Amat : Bmat, cg_max_iter); });
}
int hnh_als_run_cg(hnh_als* a, int steps) {
    return guarded(a->w, [&] { a->a->run_cg(steps); });
}
int hnh_als_compute_residual(hnh_als* a, double* out) {
    return guarded(a->w, [&] { *out = a->a->computeResidual(); });
}

// ------------------------------------------------------------------ GAT
int hnh_gat_create(hnh_dist* d, int nlayers, const int* spec3, double alpha, hnh_gat** out) {
    return guarded(d->w, [&] {
        std::vector<GATLayer> layers;
        for (int l = 0; l < nlayers; l++) layers.emplace_back(spec3[3 * l], spec3[3 * l + 1], spec3[3 * l + 2]);
        std::unique_ptr<hnh_gat> h(new hnh_gat());
        h->w = d->w;
        h->g.reset(new GAT(layers, d->d.get()));
        h->g->leaky_relu_alpha = alpha;
        *out = h.release();
    });
}
int hnh_gat_destroy(hnh_gat* g) {
    return guarded(g ? g->w : nullptr, [&] { delete g; });
}
int hnh_gat_weight_shape(hnh_gat* g, int layer, int head, int64_t o[2]) {
    return guarded(g->w, [&] {
        DenseMatrix& W = g->g->layers.at(layer).wMats.at(head);
        o[0] = W.rows();
        o[1] = W.cols();
    });
}
int hnh_gat_set_weight(hnh_gat* g, int layer, int head, const double* host) {
    return guarded(g->w, [&] { g->g->layers.at(layer).wMats.at(head).copy_from_host(host); });
}
int hnh_gat_set_input(hnh_gat* g, hnh_dense* X) {
    return guarded(g->w, [&] {
        DenseMatrix& b = g->g->buffers.at(0);
        if (b.rows() != X->m.rows() || b.cols() != X->m.cols()) hnh::fatal("Error, GAT input has the wrong shape!");
        b = X->m;
    });
}
int hnh_gat_get_output(hnh_gat* g, hnh_dense* out) {
    return guarded(g->w, [&] { out->m = g->g->buffers.back(); });
}
int hnh_gat_buffer_shape(hnh_gat* g, int index, int64_t o[2]) {
    return guarded(g->w, [&] {
        DenseMatrix& b = g->g->buffers.at(index);
        o[0] = b.rows();
        o[1] = b.cols();
    });
}
int hnh_gat_forward(hnh_gat* g) {
    return guarded(g->w, [&] { g->g->forwardPass(); });
}

}  // extern "C"
