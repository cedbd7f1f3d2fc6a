/*
 * TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's local kernels in plain C, exported
 * through the SAME C ABI as the product's HIP library (include/hnh_kernels.h) so that tests can
 *   (1) compare it with the golden vectors produced by the reference itself (tests/test_oracle_golden.py)
 *   (2) stand in for the GPU when the HOST logic (redistribution, block layout, shift schedules,
 *       transports) is exercised on a machine without one (pytest -m "not gpu").
 * "Device" pointers are host pointers here and every call runs on the spot (streams and events do not schedule anything) — but what
 * the host layer ENQUEUES is watched: hnh_stream_order.h keeps a vector clock per stream and the bytes every call reads and writes,
 * and reports conflicting accesses that no event / synchronisation orders (HNH_ORDER_CHECK=1; on for every test process).
 * The product never loads this library: hnh_backend_load() is given its path explicitly by tests only,
 * and bench.py / smoke() assert that the active backend is "hip-gfx950".
 *
 * Restated reference code (file:line under /root/reference):
 *   hnh_sddmm_coo / hnh_sddmm_csr : StandardKernel::sddmm_local loop, sparse_kernels.cpp:44-55
 *   hnh_spmm_csr                  : mkl_sparse_d_mm(NON_TRANSPOSE, alpha=1, CSR, ROW_MAJOR, beta=1),
 *                                   sparse_kernels.cpp:95-107 (Intel MKL 2021.4, third-party: the
 *                                   documented contract C = alpha*A*B + beta*C is what is restated)
 *   hnh_fused_sddmm_spmm_csr      : the two calls of 15D_dense_shift.hpp:203-217 back to back
 *   hnh_fill_f64                  : SpmatLocal::setValuesConstant, SpmatLocal.hpp:595-605
 *   hnh_hadamard_f64              : SValues.cwiseProduct(getCSRValues()), 15D_dense_shift.hpp:366
 */
#define _GNU_SOURCE /* clock_gettime, process_vm_readv */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/uio.h>
#include <time.h>
#include <unistd.h>

#include "hnh_kernels.h"
#include "hnh_measurement_aids.h"

#include "hnh_stream_order.h" /* the happens-before checker of the stream protocol (HNH_ORDER_CHECK=1); does not touch the arithmetic */

struct hnh_ctx {
    int device;
    char err[256];
    int hb_slot; /* timelines of this context in the checker, or -1 */
    char* flags_base; size_t flags_bytes; /* hnh_ipc_flags_register: this rank's mapping of the session's flag words */
    long hb_blocks; /* blocks this context has allocated: a context that never allocates (a test that hands host arrays to the kernels
                       directly) is not held to "operands are device memory" */
};

static int fail(hnh_ctx* c, int code, const char* msg) {
    if (c) { strncpy(c->err, msg, sizeof(c->err) - 1); c->err[sizeof(c->err) - 1] = 0; }
    return code;
}

const char* hnh_backend_name(void) { return "oracle-cpu-test-double"; }

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
int hnh_ctx_create(int device, hnh_ctx** out) {
    if (!out) return HNH_ERR_INVALID;
    hnh_ctx* c = (hnh_ctx*)calloc(1, sizeof(hnh_ctx));
    if (!c) return HNH_ERR_NOMEM;
    c->device = device;
    c->hb_slot = -1;
    if (hb_enabled()) {
        pthread_mutex_lock(&g_mu);
        c->hb_slot = hb_slot_take();
        pthread_mutex_unlock(&g_mu);
    }
    *out = c;
    return HNH_OK;
}
int hnh_ctx_destroy(hnh_ctx* c) {
    if (c && c->hb_slot >= 0) {  /* (the slot's clock components keep counting for its next owner: old records stay "earlier") */
        pthread_mutex_lock(&g_mu);
        hb_slot_used[c->hb_slot] = 0;
        pthread_mutex_unlock(&g_mu);
    }
    free(c);
    return HNH_OK;
}
const char* hnh_last_error(hnh_ctx* c) { return c ? c->err : "null context"; }
void* hnh_ctx_stream(hnh_ctx* c, int s) { (void)c; (void)s; return NULL; }
/* the double's "devices" are the host: one bus id per ordinal ($HNH_ORACLE_PCI_BUS_ID replaces it: tests of ranks that share a device) */
int hnh_ctx_device_identity(hnh_ctx* c, int* ordinal, char* pci_bus_id, int len) {
    if (!c || !ordinal || !pci_bus_id || len < 16) return HNH_ERR_INVALID;
    *ordinal = c->device;
    const char* forced = getenv("HNH_ORACLE_PCI_BUS_ID");
    if (forced && *forced) snprintf(pci_bus_id, (size_t)len, "%s", forced);
    else snprintf(pci_bus_id, (size_t)len, "cpu0:%02x:00.0", c->device & 0xff);
    return HNH_OK;
}
/* "device" blocks are remembered (base, size) so that the ipc double below can say which block a pointer lies in */
typedef struct block { char* base; size_t bytes; struct block* next; hb_rec* recs; int nrec, cap; } block;
static block* g_blocks = NULL;

/* ---- the checker's hooks (hnh_stream_order.h): HB_OP opens a call on a stream, HB_R / HB_W declare the bytes it reads / writes */
static int hb_op_begin(hnh_ctx* c, int stream, const char* name) {
    if (!hb_enabled() || !c || c->hb_slot < 0 || stream < 0 || stream >= HB_HOST) return 0;
    if (hb_cur.depth++ > 0) return 1;  /* a call made by another call of the ABI belongs to the outer one */
    pthread_mutex_lock(&g_mu);
    hb_cur.t = hb_tick(c->hb_slot, stream);
    pthread_mutex_unlock(&g_mu);
    hb_cur.name = name;
    hb_cur.strict = c->hb_blocks > 0;
    return 1;
}
static void hb_op_end(int* opened) {
    if (*opened) hb_cur.depth--;
}
#define HB_OP(c, stream, name) int hb_scope_ __attribute__((cleanup(hb_op_end))) = hb_op_begin((c), (stream), (name)); (void)hb_scope_
/* `where`: 1 = an operand that must be device memory, 0 = the host side of a copy (must NOT be device memory) */
static void hb_access_at(const void* p, size_t bytes, int write, int where) {
    if (hb_cur.depth <= 0 || !p || !bytes) return;
    pthread_mutex_lock(&g_mu);
    block* b = g_blocks;
    for (; b; b = b->next)
        if ((const char*)p >= b->base && (const char*)p < b->base + b->bytes) break;
    if (b && where) {
        uint64_t lo = (uint64_t)((const char*)p - b->base), hi = lo + bytes;
        if (hi > b->bytes) {  /* the ranges are exact (index streams are scanned): this call runs past the end of its operand */
            hb_note_misuse(hb_cur.name, write, "runs past the end of its block", p, bytes);
            hi = b->bytes;
        }
        hb_touch(&b->recs, &b->nrec, &b->cap, b->base, lo, hi, write, hb_cur.t, hb_cur.name);
    } else if (b && !where) {
        hb_note_misuse(hb_cur.name, write, "takes device memory where the copy kind says host memory", p, bytes);
    } else if (!b && where && hb_cur.strict) {  /* a host pointer works here and faults on the GPU */
        hb_note_misuse(hb_cur.name, write, "is given a pointer that is not device memory (no block of hnh_malloc holds it)", p, bytes);
    }
    pthread_mutex_unlock(&g_mu);
}
#define HB_R(p, bytes) hb_access_at((p), (size_t)(bytes), 0, 1)
#define HB_W(p, bytes) hb_access_at((p), (size_t)(bytes), 1, 1)
#define HB_HOST_SIDE(p, bytes) hb_access_at((p), (size_t)(bytes), 0, 0)
/* Device-to-host copies are asynchronous on the GPU (hipMemcpyAsync): the host may read the destination only behind a synchronisation
 * that covers the copy.  Under the checker the double holds the bytes back — the destination is filled with 0xFF at once and gets the
 * data when the host thread of some context synchronises past the copy (stream / event synchronise, free) — so host code that reads a
 * result too early reads NaNs here instead of, by luck of timing, the right numbers. */
typedef struct hb_pending { void* dst; void* snap; size_t bytes; int t; uint32_t clk; struct hb_pending* next; } hb_pending;
static hb_pending *hb_pend_head = NULL, **hb_pend_tail = &hb_pend_head;
static void hb_deliver(int host_t) {  /* under g_mu: everything the host timeline is now behind */
    for (hb_pending** q = &hb_pend_head; *q;) {
        hb_pending* e = *q;
        if (hb_vc[host_t][e->t] >= e->clk) {
            memcpy(e->dst, e->snap, e->bytes);
            *q = e->next;
            free(e->snap);
            free(e);
        } else {
            q = &e->next;
        }
    }
    hb_pend_tail = &hb_pend_head;
    while (*hb_pend_tail) hb_pend_tail = &(*hb_pend_tail)->next;
}
/* the accesses of a row pass over [beg[r], end[r]) of every row (NULL = the whole rows): index streams, the nonzeros' values
 * (`values_write`: -1 = not touched), and the rows of the gathered operand the column indices in range address */
static void hb_row_pass(int64_t rows, const int32_t* rowptr, const int32_t* col_idx, const int32_t* beg, const int32_t* end, const double* values,
                        int values_write, const double* svalues, const double* gathered, int R) {
    if (hb_cur.depth <= 0 || rows <= 0) return;
    int64_t lo = -1, hi = -1;
    int32_t cmin = 0, cmax = 0;
    for (int64_t r = 0; r < rows; r++) {
        const int32_t b = beg ? beg[r] : rowptr[r], e = end ? end[r] : rowptr[r + 1];
        if (b >= e) continue;
        if (lo < 0) { lo = b; cmin = cmax = col_idx[b]; }
        if (b < lo) lo = b;
        if (e > hi) hi = e;
        for (int32_t i = b; i < e; i++) {
            if (col_idx[i] < cmin) cmin = col_idx[i];
            if (col_idx[i] > cmax) cmax = col_idx[i];
        }
    }
    HB_R(rowptr, (size_t)(rows + 1) * sizeof(int32_t));
    if (beg) HB_R(beg, (size_t)rows * sizeof(int32_t));
    if (end) HB_R(end, (size_t)rows * sizeof(int32_t));
    if (lo < 0) return;
    HB_R(col_idx + lo, (size_t)(hi - lo) * sizeof(int32_t));
    if (values && values_write >= 0) hb_access_at(values + lo, (size_t)(hi - lo) * sizeof(double), values_write, 1);
    if (svalues) HB_R(svalues + lo, (size_t)(hi - lo) * sizeof(double));
    if (gathered) HB_R(gathered + (int64_t)cmin * R, (size_t)(cmax - cmin + 1) * (size_t)R * sizeof(double));
}
int hnh_malloc(hnh_ctx* c, size_t bytes, void** out) {
    if (!out) return HNH_ERR_INVALID;
    *out = NULL;  /* 256-byte aligned like hipMalloc: the host layer's alignment rules (SpmatLocal::lendable) then decide as on the device */
    if (posix_memalign(out, 256, bytes ? bytes : 16) != 0 || !*out) return fail(c, HNH_ERR_NOMEM, "malloc failed");
    /* hipMalloc does not clear memory: neither does the double — fresh blocks are filled with 0xFF (NaN as doubles, -1 as indices), so host
     * code that relies on a zeroed allocation fails here as it would, eventually, on the device (HNH_ORACLE_NO_POISON=1 turns it off) */
    if (!getenv("HNH_ORACLE_NO_POISON")) memset(*out, 0xFF, bytes ? bytes : 16);
    block* b = (block*)malloc(sizeof(block));
    if (b) {
        b->base = (char*)*out;
        b->bytes = bytes ? bytes : 16;
        b->recs = NULL;
        b->nrec = b->cap = 0;
        if (c) c->hb_blocks++;
        pthread_mutex_lock(&g_mu);
        b->next = g_blocks;
        g_blocks = b;
        pthread_mutex_unlock(&g_mu);
    }
    return HNH_OK;
}
int hnh_free(hnh_ctx* c, void* p) {
    pthread_mutex_lock(&g_mu);
    if (c && c->hb_slot >= 0)  /* hipFree synchronises the device: the host thread is behind every stream of every context */
    {
        for (int t = 0; t < HB_T; t++) hb_join(hb_vc[c->hb_slot * HB_LANES + HB_HOST], hb_vc[t]);
        hb_deliver(c->hb_slot * HB_LANES + HB_HOST);
    }
    for (block** q = &g_blocks; *q; q = &(*q)->next)
        if ((*q)->base == (char*)p) {
            block* d = *q;
            *q = d->next;
            if (!getenv("HNH_ORACLE_NO_POISON")) memset(d->base, 0xFF, d->bytes);  /* whoever still reads this block reads NaNs / -1 */
            free(d->recs);
            free(d);
            break;
        }
    pthread_mutex_unlock(&g_mu);
    free(p);
    return HNH_OK;
}
int hnh_memcpy(hnh_ctx* c, void* dst, const void* src, size_t bytes, int kind, int stream) {
    HB_OP(c, stream, "hnh_memcpy");
    if (kind == HNH_COPY_H2D) HB_HOST_SIDE(src, bytes); else HB_R(src, bytes);
    if (kind == HNH_COPY_D2H) HB_HOST_SIDE(dst, bytes); else HB_W(dst, bytes);
    if (bytes && kind == HNH_COPY_D2H && hb_cur.depth == 1) {  /* held back until the host synchronises past this copy (above) */
        hb_pending* e = (hb_pending*)malloc(sizeof(hb_pending));
        void* snap = malloc(bytes);
        if (e && snap) {
            memcpy(snap, src, bytes);
            memset(dst, 0xFF, bytes);
            pthread_mutex_lock(&g_mu);
            *e = (hb_pending){dst, snap, bytes, hb_cur.t, hb_vc[hb_cur.t][hb_cur.t], NULL};
            *hb_pend_tail = e;
            hb_pend_tail = &e->next;
            pthread_mutex_unlock(&g_mu);
            return HNH_OK;
        }
        free(e);
        free(snap);
    }
    if (bytes) memmove(dst, src, bytes);
    return HNH_OK;
}
int hnh_memset(hnh_ctx* c, void* dst, int byte, size_t bytes, int stream) {
    HB_OP(c, stream, "hnh_memset");
    HB_W(dst, bytes);
    if (bytes) memset(dst, byte, bytes);
    return HNH_OK;
}
/* calls the ABI documents as synchronous (they hand a count or a bound back to the host) end with this */
static int hb_synchronous(hnh_ctx* c, int s) {
    if (c && c->hb_slot >= 0 && s >= 0 && s < HB_HOST) {
        pthread_mutex_lock(&g_mu);
        hb_join(hb_vc[c->hb_slot * HB_LANES + HB_HOST], hb_vc[c->hb_slot * HB_LANES + s]);
        hb_deliver(c->hb_slot * HB_LANES + HB_HOST);
        pthread_mutex_unlock(&g_mu);
    }
    return HNH_OK;
}
int hnh_stream_sync(hnh_ctx* c, int s) {
    if (c && c->hb_slot >= 0 && s >= 0 && s < HB_HOST) {  /* the host thread is behind everything the stream was given */
        pthread_mutex_lock(&g_mu);
        hb_join(hb_vc[c->hb_slot * HB_LANES + HB_HOST], hb_vc[c->hb_slot * HB_LANES + s]);
        hb_deliver(c->hb_slot * HB_LANES + HB_HOST);
        pthread_mutex_unlock(&g_mu);
    }
    return HNH_OK;
}
/* the test double runs everything synchronously, so an event is just the host time at which it was recorded */
static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
int hnh_event_create(hnh_ctx* c, void** e) { (void)c; *e = calloc(1, sizeof(hb_event)); return *e ? HNH_OK : HNH_ERR_NOMEM; }
int hnh_event_destroy(hnh_ctx* c, void* e) {
    (void)c;
    if (e) free(((hb_event*)e)->clk);
    free(e);
    return HNH_OK;
}
int hnh_event_record(hnh_ctx* c, void* e, int s) {
    hb_event* ev = (hb_event*)e;
    ev->ms = now_ms();
    if (c && c->hb_slot >= 0 && s >= 0 && s < HB_HOST) {  /* the event stands for everything the stream was given so far */
        pthread_mutex_lock(&g_mu);
        const int t = hb_tick(c->hb_slot, s);
        if (!ev->clk) ev->clk = (uint32_t*)malloc(sizeof(uint32_t) * HB_T);
        if (ev->clk) memcpy(ev->clk, hb_vc[t], sizeof(uint32_t) * HB_T);
        pthread_mutex_unlock(&g_mu);
    }
    return HNH_OK;
}
int hnh_event_wait(hnh_ctx* c, void* e, int s) {
    hb_event* ev = (hb_event*)e;
    /* (the checker's own tests: HNH_ORDER_CHECK_DROP_WAITS ignores every wait — the protocol must then be reported as racy —
     * and hnh_oracle_order_drop_wait(k) ignores the k-th wait from now: single-fault injection, which waits are load-bearing) */
    if (c && c->hb_slot >= 0 && s >= 0 && s < HB_HOST && ev && ev->clk && !getenv("HNH_ORDER_CHECK_DROP_WAITS") &&
        __atomic_add_fetch(&hb_wait_count, 1, __ATOMIC_RELAXED) != hb_wait_drop) {  /* what the stream is given from now on runs behind the event */
        pthread_mutex_lock(&g_mu);
        hb_join(hb_vc[hb_tick(c->hb_slot, s)], ev->clk);
        pthread_mutex_unlock(&g_mu);
    }
    return HNH_OK;
}
int hnh_event_sync(hnh_ctx* c, void* e) {
    hb_event* ev = (hb_event*)e;
    if (c && c->hb_slot >= 0 && ev && ev->clk) {
        pthread_mutex_lock(&g_mu);
        hb_join(hb_vc[c->hb_slot * HB_LANES + HB_HOST], ev->clk);
        hb_deliver(c->hb_slot * HB_LANES + HB_HOST);
        pthread_mutex_unlock(&g_mu);
    }
    return HNH_OK;
}

/* The double runs every call on the spot, so an event that was recorded has completed; a query that says so tells the host what a
 * synchronise would (same edge).  HNH_ORACLE_EVENTS_PENDING=k: every k-th query (counted over the process) answers "not yet" — for
 * tests of callers that act on the answer (the adaptive chunk windows of the 1.5D dense shift take fewer chunks per pass then). */
int hnh_event_query(hnh_ctx* c, void* e, int* done) {
    if (!e || !done) return HNH_ERR_INVALID;
    static long asked = 0;
    const char* k = getenv("HNH_ORACLE_EVENTS_PENDING");
    const long every = k ? atol(k) : 0;
    if (every > 0 && __atomic_add_fetch(&asked, 1, __ATOMIC_RELAXED) % every == 0) {
        *done = 0;
        return HNH_OK;
    }
    *done = 1;
    return hnh_event_sync(c, e);
}

/* ---- the checker's own entry points (tests only) */
void hnh_oracle_order_enable(int on) {
    pthread_mutex_lock(&g_mu);
    if (on && hb_on <= 0) {
        static int registered = 0;
        if (!registered && (registered = 1)) atexit(hb_at_exit);
    }
    hb_on = on ? 1 : 0;
    pthread_mutex_unlock(&g_mu);
}
long hnh_oracle_order_races(void) { return hb_races; }
/* restarts the count of event waits and ignores the k-th one from now on (k <= 0: none); returns the waits counted before */
long hnh_oracle_order_drop_wait(long k) {
    const long seen = hb_wait_count;
    hb_wait_count = 0;
    hb_wait_drop = k > 0 ? k : -1;
    return seen;
}
long hnh_oracle_order_accesses(void) { return hb_checked; }
/* copies the reports so far (NUL terminated) and forgets them and the count */
long hnh_oracle_order_report(char* buf, size_t capacity) {
    pthread_mutex_lock(&g_mu);
    const long n = hb_races;
    if (buf && capacity) {
        const size_t k = hb_report_len < capacity - 1 ? hb_report_len : capacity - 1;
        memcpy(buf, hb_report, k);
        buf[k] = 0;
    }
    hb_races = 0;
    hb_report_len = 0;
    hb_report[0] = 0;
    pthread_mutex_unlock(&g_mu);
    return n;
}
int hnh_stream_paced_copy(hnh_ctx* c, int stream, void* dst, const void* src, size_t bytes, int n, double us, int wgs) {
    (void)us; (void)wgs;
    HB_OP(c, stream, "hnh_stream_paced_copy");
    HB_R(src, bytes);
    HB_W(dst, (size_t)n * bytes);
    for (int k = 0; k < n; k++) memcpy((char*)dst + (size_t)k * bytes, src, bytes);
    return HNH_OK;
}
int hnh_stream_pace_begin(hnh_ctx* c, int stream) { (void)c; (void)stream; return HNH_OK; }
int hnh_stream_pace_end(hnh_ctx* c, int stream, double us) { (void)c; (void)stream; (void)us; return HNH_OK; }
int hnh_stream_delay_us(hnh_ctx* c, int stream, double us) { (void)c; (void)stream; (void)us; return HNH_OK; }  /* synchronous double: nothing to pace */
int hnh_event_elapsed_ms(hnh_ctx* c, void* a, void* b, float* ms) { (void)c; *ms = (float)(((hb_event*)b)->ms - ((hb_event*)a)->ms); return HNH_OK; }

/* sparse_kernels.cpp:44-55 */
int hnh_sddmm_coo(hnh_ctx* c, int64_t nnz, const int32_t* row_idx, const int32_t* col_idx, double* values, const double* X,
                  const double* Y, int R, int stream) {
    if (nnz < 0 || R <= 0) return fail(c, HNH_ERR_INVALID, "bad size");
    HB_OP(c, stream, "hnh_sddmm_coo");
    if (hb_cur.depth > 0 && nnz > 0) {
        int32_t rmin = row_idx[0], rmax = row_idx[0], cmin = col_idx[0], cmax = col_idx[0];
        for (int64_t i = 1; i < nnz; i++) {
            if (row_idx[i] < rmin) rmin = row_idx[i];
            if (row_idx[i] > rmax) rmax = row_idx[i];
            if (col_idx[i] < cmin) cmin = col_idx[i];
            if (col_idx[i] > cmax) cmax = col_idx[i];
        }
        HB_R(row_idx, nnz * sizeof(int32_t));
        HB_R(col_idx, nnz * sizeof(int32_t));
        HB_W(values, nnz * sizeof(double));
        HB_R(X + (int64_t)rmin * R, (size_t)(rmax - rmin + 1) * R * sizeof(double));
        HB_R(Y + (int64_t)cmin * R, (size_t)(cmax - cmin + 1) * R * sizeof(double));
    }
    for (int64_t i = 0; i < nnz; i++) {
        const double* Arow = X + (int64_t)R * row_idx[i];
        const double* Brow = Y + (int64_t)R * col_idx[i];
        double value = 0.0;
        for (int k = 0; k < R; k++) value += Arow[k] * Brow[k];
        values[i] += value;
    }
    return HNH_OK;
}

int hnh_sddmm_csr(hnh_ctx* c, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values, const double* X,
                  const double* Y, int R, int stream) {
    if (rows < 0 || R <= 0) return fail(c, HNH_ERR_INVALID, "bad size");
    HB_OP(c, stream, "hnh_sddmm_csr");
    hb_row_pass(rows, rowptr, col_idx, NULL, NULL, values, 1, NULL, Y, R);
    HB_R(X, (size_t)rows * R * sizeof(double));
    for (int64_t r = 0; r < rows; r++)
        for (int32_t i = rowptr[r]; i < rowptr[r + 1]; i++) {
            const double* Arow = X + (int64_t)R * r;
            const double* Brow = Y + (int64_t)R * col_idx[i];
            double value = 0.0;
            for (int k = 0; k < R; k++) value += Arow[k] * Brow[k];
            values[i] += value;
        }
    return HNH_OK;
}

int hnh_sddmm_csr_ex(hnh_ctx* c, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values, const double* X,
                     const double* Y, int R, int64_t nnz, int max_row_nnz, int64_t cols, int stream) {
    (void)nnz; (void)max_row_nnz; (void)cols;  /* hints only */
    return hnh_sddmm_csr(c, rows, rowptr, col_idx, values, X, Y, R, stream);
}

/* C = 1.0 * S * X + 1.0 * C, row-major, ld = R (sparse_kernels.cpp:95-107) */
static void spmm_csr_loop(int64_t rows, const int32_t* rowptr, const int32_t* col_idx, const double* values, const double* X, double* Out, int R) {
    for (int64_t r = 0; r < rows; r++)
        for (int32_t i = rowptr[r]; i < rowptr[r + 1]; i++) {
            const double v = values[i];
            const double* Xrow = X + (int64_t)R * col_idx[i];
            double* Crow = Out + (int64_t)R * r;
            for (int k = 0; k < R; k++) Crow[k] += v * Xrow[k];
        }
}
int hnh_spmm_csr(hnh_ctx* c, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, const double* values, const double* X,
                 double* Out, int R, int stream) {
    if (rows < 0 || R <= 0) return fail(c, HNH_ERR_INVALID, "bad size");
    if (X == Out) return fail(c, HNH_ERR_INVALID, "X and Out alias");
    HB_OP(c, stream, "hnh_spmm_csr");
    hb_row_pass(rows, rowptr, col_idx, NULL, NULL, values, 0, NULL, X, R);
    HB_W(Out, (size_t)rows * R * sizeof(double));
    spmm_csr_loop(rows, rowptr, col_idx, values, X, Out, R);
    return HNH_OK;
}

int hnh_spmm_csr_ex(hnh_ctx* c, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, const double* values, const double* X,
                    double* Out, int R, int64_t nnz, int max_row_nnz, int64_t cols, int stream) {
    (void)nnz; (void)max_row_nnz; (void)cols;
    return hnh_spmm_csr(c, rows, rowptr, col_idx, values, X, Out, R, stream);
}

/* 15D_dense_shift.hpp:203-217: sddmm on the block, then spmm with the block's (updated) values */
int hnh_fused_sddmm_spmm_csr(hnh_ctx* c, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                             const double* svalues, const double* X, const double* Y, double* Out, int R, unsigned flags,
                             int stream) {
    if (rows < 0 || R <= 0) return fail(c, HNH_ERR_INVALID, "bad size");
    if (rows == 0) return HNH_OK;
    HB_OP(c, stream, "hnh_fused_sddmm_spmm_csr");
    if (svalues) HB_R(svalues, (size_t)rowptr[rows] * sizeof(double));
    const int32_t nnz = rowptr[rows];
    if (flags & HNH_FUSED_VALUES_OVERWRITE) memset(values, 0, sizeof(double) * (size_t)nnz);
    if (flags & HNH_FUSED_OUT_OVERWRITE) memset(Out, 0, sizeof(double) * (size_t)rows * (size_t)R);
    int rc = hnh_sddmm_csr(c, rows, rowptr, col_idx, values, X, Y, R, stream);
    if (rc != HNH_OK) return rc;
    if (!svalues) return hnh_spmm_csr(c, rows, rowptr, col_idx, values, Y, Out, R, stream);
    double* w = (double*)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    if (!w) return fail(c, HNH_ERR_NOMEM, "malloc failed");
    for (int32_t i = 0; i < nnz; i++) w[i] = values[i] * svalues[i];
    if (Y == Out) { free(w); return fail(c, HNH_ERR_INVALID, "X and Out alias"); }
    HB_W(Out, (size_t)rows * R * sizeof(double));  /* (the SDDMM half above declared the index streams, the values and the rows of Y) */
    spmm_csr_loop(rows, rowptr, col_idx, w, Y, Out, R);  /* w is the double's own scratch, not an operand */
    free(w);
    return HNH_OK;
}

int hnh_fused_sddmm_spmm_csr_ex(hnh_ctx* c, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                                const double* svalues, const double* X, const double* Y, double* Out, int R, unsigned flags,
                                int64_t nnz, int max_row_nnz, int64_t cols, int stream) {
    (void)nnz; (void)max_row_nnz; (void)cols;
    return hnh_fused_sddmm_spmm_csr(c, rows, rowptr, col_idx, values, svalues, X, Y, Out, R, flags, stream);
}

/* als_conjugate_gradients.cpp:282,295 (+ lambda * X) and :93 (batch_dot_product(p, Mp)) on a finished output */
static void row_epilogue_loop(double* Out, const double* X, double x_scale, double* rowdot, int64_t rows, int R) {
    for (int64_t i = 0; i < rows; i++) {
        double s = 0.0;
        for (int j = 0; j < R; j++) {
            Out[i * R + j] += x_scale * X[i * R + j];
            s += X[i * R + j] * Out[i * R + j];
        }
        if (rowdot) rowdot[i] = s;
    }
}
int hnh_row_epilogue_f64(hnh_ctx* c, double* Out, const double* X, double x_scale, double* rowdot, int64_t rows, int R, int stream) {
    if (rows < 0 || R <= 0) return fail(c, HNH_ERR_INVALID, "bad size");
    HB_OP(c, stream, "hnh_row_epilogue_f64");
    HB_W(Out, (size_t)rows * R * sizeof(double));
    HB_R(X, (size_t)rows * R * sizeof(double));
    if (rowdot) HB_W(rowdot, (size_t)rows * sizeof(double));
    row_epilogue_loop(Out, X, x_scale, rowdot, rows, R);
    return HNH_OK;
}

/* the same with the whole extras record: + the remaining updates of one batched-CG iteration, written as the reference's
 * sequence of whole-matrix statements (als_conjugate_gradients.cpp:91-139) */
int hnh_row_epilogue_x(hnh_ctx* c, double* Out, const double* X, const hnh_fused_extras* ex, int64_t rows, int R, int stream) {
    if (!ex) return fail(c, HNH_ERR_INVALID, "null extras");
    if (rows < 0 || R <= 0) return fail(c, HNH_ERR_INVALID, "bad size");
    const hnh_cg_update* cg = ex->cg;
    if (cg) {
        if (!cg->x || !cg->r || !cg->p || !cg->rsold) return fail(c, HNH_ERR_INVALID, "hnh_cg_update with a null pointer");
        if (cg->p != X) return fail(c, HNH_ERR_INVALID, "hnh_cg_update.p must be the row operand X");
        if (cg->x == cg->r || cg->x == cg->p || cg->r == cg->p || cg->x == Out || cg->r == Out || cg->p == Out)
            return fail(c, HNH_ERR_INVALID, "hnh_cg_update operands alias");
    }
    if (ex->relu_dst) {
        if (cg) return fail(c, HNH_ERR_INVALID, "relu_dst and cg exclude each other");
        if (ex->relu_ld <= 0) return fail(c, HNH_ERR_INVALID, "relu_dst needs its row pitch");
        if (ex->relu_dst == Out || ex->relu_dst == X) return fail(c, HNH_ERR_INVALID, "relu_dst aliases an operand");
    }
    if (rows == 0) return HNH_OK;
    HB_OP(c, stream, "hnh_row_epilogue_x");
    HB_W(Out, (size_t)rows * R * sizeof(double));
    HB_R(X, (size_t)rows * R * sizeof(double));
    if (ex->rowdot) HB_W(ex->rowdot, (size_t)rows * sizeof(double));
    if (cg) {
        HB_W(cg->x, (size_t)rows * R * sizeof(double));
        HB_W(cg->r, (size_t)rows * R * sizeof(double));
        HB_W(cg->p, (size_t)rows * R * sizeof(double));
        HB_W(cg->rsold, (size_t)rows * sizeof(double));
    }
    if (ex->relu_dst)
        for (int64_t i = 0; i < rows && hb_cur.depth > 0; i++) HB_W(ex->relu_dst + i * ex->relu_ld, (size_t)R * sizeof(double));
    double* bdot = cg ? (double*)malloc(sizeof(double) * (size_t)rows) : NULL;
    if (cg && !bdot) return fail(c, HNH_ERR_NOMEM, "malloc failed");
    int rc = HNH_OK;
    if (ex->x_scale != 0.0 || ex->rowdot || cg) {
        row_epilogue_loop(Out, X, ex->x_scale, cg ? bdot : ex->rowdot, rows, R);  /* (bdot is the double's own scratch; the operands are declared above) */
        if (cg && ex->rowdot) memcpy(ex->rowdot, bdot, sizeof(double) * (size_t)rows);
    }
    if (rc == HNH_OK && cg) {
        for (int64_t i = 0; i < rows; i++) {
            bdot[i] += cg->eps;                         /* :99  */
            const double rs = cg->rsold[i] + cg->eps;   /* :100 */
            const double alpha = rs / bdot[i];          /* :102 */
            double rsnew = 0.0;
            for (int j = 0; j < R; j++) {
                cg->x[i * R + j] += alpha * cg->p[i * R + j];   /* :112-117 */
                cg->r[i * R + j] -= alpha * Out[i * R + j];     /* :118 */
                rsnew += cg->r[i * R + j] * cg->r[i * R + j];   /* :120 */
            }
            const double coeff = rsnew / rs;            /* :136 */
            for (int j = 0; j < R; j++) cg->p[i * R + j] = cg->r[i * R + j] + coeff * cg->p[i * R + j]; /* :137 */
            cg->rsold[i] = rsnew;                       /* :138 */
        }
    }
    if (rc == HNH_OK && ex->relu_dst) { /* gat.hpp:101 */
        for (int64_t i = 0; i < rows; i++)
            for (int j = 0; j < R; j++) ex->relu_dst[i * ex->relu_ld + j] = Out[i * R + j] > 0.0 ? Out[i * R + j] : 0.0;
    }
    free(bdot);
    return rc;
}

/* gat.hpp:96-99: SDDMM, LeakyReLU on the values, SpMM with them; then the row epilogue */
int hnh_fused_sddmm_spmm_csr_x(hnh_ctx* c, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                               const double* svalues, const double* X, const double* Y, double* Out, int R, unsigned flags,
                               int64_t nnz_hint, int max_row_nnz, int64_t cols, const hnh_fused_extras* ex, int stream) {
    (void)nnz_hint; (void)max_row_nnz; (void)cols;
    if (rows < 0 || R <= 0) return fail(c, HNH_ERR_INVALID, "bad size");
    if ((flags & HNH_FUSED_LEAKY_RELU) && !ex) return fail(c, HNH_ERR_INVALID, "HNH_FUSED_LEAKY_RELU needs extras");
    if (rows == 0) return HNH_OK;
    HB_OP(c, stream, "hnh_fused_sddmm_spmm_csr_x");
    if (svalues) HB_R(svalues, (size_t)rowptr[rows] * sizeof(double));
    int rc;
    if (flags & HNH_FUSED_LEAKY_RELU) {
        const int32_t nnz = rowptr[rows];
        if (flags & HNH_FUSED_VALUES_OVERWRITE) memset(values, 0, sizeof(double) * (size_t)nnz);
        if (flags & HNH_FUSED_OUT_OVERWRITE) memset(Out, 0, sizeof(double) * (size_t)rows * (size_t)R);
        rc = hnh_sddmm_csr(c, rows, rowptr, col_idx, values, X, Y, R, stream);
        if (rc != HNH_OK) return rc;
        for (int32_t i = 0; i < nnz; i++) {
            const double v = values[i] * (svalues ? svalues[i] : 1.0);
            values[i] = v > 0.0 ? v : ex->leaky_alpha * v;
        }
        rc = hnh_spmm_csr(c, rows, rowptr, col_idx, values, Y, Out, R, stream);
    } else {
        rc = hnh_fused_sddmm_spmm_csr(c, rows, rowptr, col_idx, values, svalues, X, Y, Out, R, flags, stream);
    }
    if (rc != HNH_OK) return rc;
    if (ex && (ex->x_scale != 0.0 || ex->rowdot || ex->cg || ex->relu_dst)) return hnh_row_epilogue_x(c, Out, X, ex, rows, R, stream);
    return HNH_OK;
}

/* ---- row windows (hnh_csr_window): the same loops over [beg[r], end[r]) of every row */
int hnh_csr_window_bounds(hnh_ctx* c, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, int nbounds, const int32_t* bounds,
                          int32_t* split, int stream) {
    if (rows < 0 || nbounds < 0 || nbounds > 15) return fail(c, HNH_ERR_INVALID, "bad size");
    HB_OP(c, stream, "hnh_csr_window_bounds");
    hb_row_pass(rows, rowptr, col_idx, NULL, NULL, NULL, -1, NULL, NULL, 0);
    HB_W(split, (size_t)nbounds * (size_t)rows * sizeof(int32_t));
    for (int b = 1; b < nbounds; b++)
        if (bounds[b] < bounds[b - 1]) return fail(c, HNH_ERR_INVALID, "bounds must not decrease");
    for (int b = 0; b < nbounds; b++)
        for (int64_t r = 0; r < rows; r++) {
            int32_t e = rowptr[r];
            while (e < rowptr[r + 1] && col_idx[e] < bounds[b]) e++;
            split[(int64_t)b * rows + r] = e;
        }
    return HNH_OK;
}

int hnh_sddmm_csr_w(hnh_ctx* c, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values, const double* X,
                    const double* Y, int R, int64_t nnz, int max_row_nnz, const hnh_csr_window* w, int stream) {
    (void)nnz; (void)max_row_nnz;
    if (rows < 0 || R <= 0 || !w) return fail(c, HNH_ERR_INVALID, "bad argument");
    HB_OP(c, stream, "hnh_sddmm_csr_w");
    hb_row_pass(rows, rowptr, col_idx, w->beg, w->end, values, 1, NULL, Y, R);
    HB_R(X, (size_t)rows * R * sizeof(double));
    for (int64_t r = 0; r < rows; r++) {
        const int32_t b = w->beg ? w->beg[r] : rowptr[r], e = w->end ? w->end[r] : rowptr[r + 1];
        for (int32_t i = b; i < e; i++) {
            const double* Arow = X + (int64_t)R * r;
            const double* Brow = Y + (int64_t)R * col_idx[i];
            double value = 0.0;
            for (int k = 0; k < R; k++) value += Arow[k] * Brow[k];
            values[i] += value;
        }
    }
    return HNH_OK;
}

int hnh_spmm_csr_w(hnh_ctx* c, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, const double* values, const double* X,
                   double* Out, int R, int64_t nnz, int max_row_nnz, const hnh_csr_window* w, int stream) {
    (void)nnz; (void)max_row_nnz;
    if (rows < 0 || R <= 0 || !w) return fail(c, HNH_ERR_INVALID, "bad argument");
    if (X == Out) return fail(c, HNH_ERR_INVALID, "X and Out alias");
    HB_OP(c, stream, "hnh_spmm_csr_w");
    hb_row_pass(rows, rowptr, col_idx, w->beg, w->end, values, 0, NULL, X, R);
    HB_W(Out, (size_t)rows * R * sizeof(double));
    for (int64_t r = 0; r < rows; r++) {
        const int32_t b = w->beg ? w->beg[r] : rowptr[r], e = w->end ? w->end[r] : rowptr[r + 1];
        for (int32_t i = b; i < e; i++) {
            const double v = values[i];
            const double* Xrow = X + (int64_t)R * col_idx[i];
            double* Crow = Out + (int64_t)R * r;
            for (int k = 0; k < R; k++) Crow[k] += v * Xrow[k];
        }
    }
    return HNH_OK;
}

int hnh_fused_sddmm_spmm_csr_w(hnh_ctx* c, int64_t rows, const int32_t* rowptr, const int32_t* col_idx, double* values,
                               const double* svalues, const double* X, const double* Y, double* Out, int R, unsigned flags,
                               int64_t nnz, int max_row_nnz, const hnh_fused_extras* ex, const hnh_csr_window* w, int stream) {
    if (rows < 0 || R <= 0 || !w) return fail(c, HNH_ERR_INVALID, "bad argument");
    if ((flags & HNH_FUSED_LEAKY_RELU) && !ex) return fail(c, HNH_ERR_INVALID, "HNH_FUSED_LEAKY_RELU needs extras");
    const int epilogue = ex && (ex->x_scale != 0.0 || ex->rowdot || ex->cg || ex->relu_dst);
    if (epilogue && !w->last) return fail(c, HNH_ERR_INVALID, "a row epilogue belongs to the last window");
    if (rows == 0) return HNH_OK;
    HB_OP(c, stream, "hnh_fused_sddmm_spmm_csr_w");
    hb_row_pass(rows, rowptr, col_idx, w->beg, w->end, values, 1, svalues, Y, R);
    HB_R(X, (size_t)rows * R * sizeof(double));
    HB_W(Out, (size_t)rows * R * sizeof(double));
    if (flags & HNH_FUSED_OUT_OVERWRITE) memset(Out, 0, sizeof(double) * (size_t)rows * (size_t)R);
    for (int64_t r = 0; r < rows; r++) {
        const int32_t b = w->beg ? w->beg[r] : rowptr[r], e = w->end ? w->end[r] : rowptr[r + 1];
        if (flags & HNH_FUSED_VALUES_OVERWRITE)
            for (int32_t i = b; i < e; i++) values[i] = 0.0;
    }
    int rc = hnh_sddmm_csr_w(c, rows, rowptr, col_idx, values, X, Y, R, nnz, max_row_nnz, w, stream);
    if (rc != HNH_OK) return rc;
    for (int64_t r = 0; r < rows; r++) {  /* the weights of the SpMM half: values, times svalues, activated (and kept) with LEAKY_RELU */
        const int32_t b = w->beg ? w->beg[r] : rowptr[r], e = w->end ? w->end[r] : rowptr[r + 1];
        for (int32_t i = b; i < e; i++) {
            double v = values[i] * (svalues ? svalues[i] : 1.0);
            if (flags & HNH_FUSED_LEAKY_RELU) {
                v = v > 0.0 ? v : ex->leaky_alpha * v;
                values[i] = v;
            }
            const double* Yrow = Y + (int64_t)R * col_idx[i];
            double* Crow = Out + (int64_t)R * r;
            for (int k = 0; k < R; k++) Crow[k] += v * Yrow[k];
        }
    }
    if (epilogue) return hnh_row_epilogue_x(c, Out, X, ex, rows, R, stream);
    return HNH_OK;
}

/* als_conjugate_gradients.cpp:117-127 */
int hnh_cg_step_f64(hnh_ctx* c, double* X, double* Rm, const double* P, const double* MP, const double* alpha, double* rsnew,
                    int64_t rows, int R, int stream) {
    HB_OP(c, stream, "hnh_cg_step_f64");
    HB_W(X, (size_t)rows * R * sizeof(double));
    HB_W(Rm, (size_t)rows * R * sizeof(double));
    HB_R(P, (size_t)rows * R * sizeof(double));
    HB_R(MP, (size_t)rows * R * sizeof(double));
    HB_R(alpha, (size_t)rows * sizeof(double));
    HB_W(rsnew, (size_t)rows * sizeof(double));
    for (int64_t i = 0; i < rows; i++) {
        double s = 0.0;
        for (int j = 0; j < R; j++) {
            X[i * R + j] += alpha[i] * P[i * R + j];
            Rm[i * R + j] -= alpha[i] * MP[i * R + j];
            s += Rm[i * R + j] * Rm[i * R + j];
        }
        rsnew[i] = s;
    }
    return HNH_OK;
}
int hnh_panel_count(hnh_ctx* c, int64_t rows, int64_t nnz, int64_t cols, int R, int max_row_nnz) {
    (void)c; (void)rows; (void)nnz; (void)cols; (void)R; (void)max_row_nnz;
    return 1;
}
int hnh_csr_max_row_nnz(hnh_ctx* c, int64_t rows, const int32_t* rowptr, int* out_host, int stream) {
    HB_OP(c, stream, "hnh_csr_max_row_nnz");
    HB_R(rowptr, (size_t)(rows + 1) * sizeof(int32_t));
    int m = 0;
    for (int64_t r = 0; r < rows; r++) if (rowptr[r + 1] - rowptr[r] > m) m = rowptr[r + 1] - rowptr[r];
    *out_host = m;
    return hb_synchronous(c, stream);
}

int hnh_fill_f64(hnh_ctx* c, double* dst, int64_t n, double v, int stream) {
    HB_OP(c, stream, "hnh_fill_f64");
    HB_W(dst, (size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; i++) dst[i] = v;
    return HNH_OK;
}
int hnh_hadamard_f64(hnh_ctx* c, double* out, const double* a, const double* b, int64_t n, int stream) {
    HB_OP(c, stream, "hnh_hadamard_f64");
    HB_R(a, (size_t)n * sizeof(double));
    HB_R(b, (size_t)n * sizeof(double));
    HB_W(out, (size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; i++) out[i] = a[i] * b[i];
    return HNH_OK;
}
int hnh_axpy_f64(hnh_ctx* c, double* y, const double* x, double alpha, int64_t n, int stream) {
    HB_OP(c, stream, "hnh_axpy_f64");
    HB_R(x, (size_t)n * sizeof(double));
    HB_W(y, (size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; i++) y[i] += alpha * x[i];
    return HNH_OK;
}
int hnh_expand_rowptr(hnh_ctx* c, int64_t rows, const int32_t* rowptr, int32_t* row_idx, int stream) {
    HB_OP(c, stream, "hnh_expand_rowptr");
    HB_R(rowptr, (size_t)(rows + 1) * sizeof(int32_t));
    if (rows > 0) HB_W(row_idx + rowptr[0], (size_t)(rowptr[rows] - rowptr[0]) * sizeof(int32_t));
    for (int64_t r = 0; r < rows; r++)
        for (int32_t i = rowptr[r]; i < rowptr[r + 1]; i++) row_idx[i] = (int32_t)r;
    return HNH_OK;
}

int hnh_sum_chunked_blocks_f64(hnh_ctx* c, double* dst, const double* src, int nblocks, int nchunks, const int64_t* cuts, int q0, int q1, int R,
                               int stream) {
    if (nblocks < 0 || nchunks < 1 || nchunks > HNH_MAX_CHUNKS || !cuts || q0 < 0 || q1 > nchunks || q0 > q1 || R <= 0)
        return fail(c, HNH_ERR_INVALID, "hnh_sum_chunked_blocks_f64: bad argument");
    if (cuts[q1] == cuts[q0] || nblocks == 0) return HNH_OK;
    if (!dst || !src) return fail(c, HNH_ERR_INVALID, "null pointer");
    HB_OP(c, stream, "hnh_sum_chunked_blocks_f64");
    HB_W(dst + cuts[q0] * R, (size_t)(cuts[q1] - cuts[q0]) * R * sizeof(double));
    HB_R(src + (int64_t)nblocks * cuts[q0] * R, (size_t)nblocks * (size_t)(cuts[q1] - cuts[q0]) * R * sizeof(double));
    for (int q = q0; q < q1; q++) {
        const int64_t w = cuts[q + 1] - cuts[q];
        for (int64_t i = 0; i < w; i++)
            for (int k = 0; k < R; k++) {
                double acc = dst[(cuts[q] + i) * R + k];
                for (int b = 0; b < nblocks; b++) acc += src[((int64_t)nblocks * cuts[q] + (int64_t)b * w + i) * R + k];
                dst[(cuts[q] + i) * R + k] = acc;
            }
    }
    return HNH_OK;
}

/* als_conjugate_gradients.cpp:9-11 */
int hnh_rowdot_f64(hnh_ctx* c, const double* A, const double* B, double* out, int64_t rows, int R, int stream) {
    HB_OP(c, stream, "hnh_rowdot_f64");
    HB_R(A, (size_t)rows * R * sizeof(double));
    HB_R(B, (size_t)rows * R * sizeof(double));
    HB_W(out, (size_t)rows * sizeof(double));
    for (int64_t i = 0; i < rows; i++) {
        double s = 0.0;
        for (int j = 0; j < R; j++) s += A[i * R + j] * B[i * R + j];
        out[i] = s;
    }
    return HNH_OK;
}
/* scale_matrix_rows + add (als_conjugate_gradients.cpp:13-29,117-123,137) */
int hnh_row_scale_add_f64(hnh_ctx* c, double* Y, const double* yv, double ya, const double* X, const double* xv, double xa,
                          int64_t rows, int R, int stream) {
    HB_OP(c, stream, "hnh_row_scale_add_f64");
    HB_W(Y, (size_t)rows * R * sizeof(double));
    HB_R(X, (size_t)rows * R * sizeof(double));
    if (yv) HB_R(yv, (size_t)rows * sizeof(double));
    if (xv) HB_R(xv, (size_t)rows * sizeof(double));
    for (int64_t i = 0; i < rows; i++) {
        const double fy = ya * (yv ? yv[i] : 1.0), fx = xa * (xv ? xv[i] : 1.0);
        for (int j = 0; j < R; j++) Y[i * R + j] = fy * Y[i * R + j] + fx * X[i * R + j];
    }
    return HNH_OK;
}
int hnh_vec_add_scalar_f64(hnh_ctx* c, double* v, double s, int64_t n, int stream) {
    HB_OP(c, stream, "hnh_vec_add_scalar_f64");
    HB_W(v, (size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; i++) v[i] += s;
    return HNH_OK;
}
int hnh_fill_hashed_f64(hnh_ctx* c, double* dst, int64_t rows, int64_t cols, int64_t top_row, int64_t left_col, int64_t rg,
                        uint64_t seed, double scale, int stream) {
    HB_OP(c, stream, "hnh_fill_hashed_f64");
    HB_W(dst, (size_t)rows * (size_t)cols * sizeof(double));
    for (int64_t i = 0; i < rows; i++)
        for (int64_t j = 0; j < cols; j++) {
            uint64_t z = seed * 0xD1342543DE82EF95ull + (uint64_t)((top_row + i) * rg + left_col + j) * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z = z ^ (z >> 31);
            dst[i * cols + j] = ((double)(z >> 11) * 0x1.0p-52 - 1.0) * scale;
        }
    return HNH_OK;
}
int hnh_vec_div_f64(hnh_ctx* c, double* out, const double* num, const double* den, int64_t n, int stream) {
    HB_OP(c, stream, "hnh_vec_div_f64");
    HB_R(num, (size_t)n * sizeof(double));
    HB_R(den, (size_t)n * sizeof(double));
    HB_W(out, (size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; i++) out[i] = num[i] / den[i];
    return HNH_OK;
}

/* gat.hpp:88 (Eigen dense product), :96-97, :103 */
int hnh_gemm_f64(hnh_ctx* c, int64_t M, int64_t N, int64_t K, const double* A, const double* B, double* C, int stream) {
    HB_OP(c, stream, "hnh_gemm_f64");
    HB_R(A, (size_t)M * K * sizeof(double));
    HB_R(B, (size_t)K * N * sizeof(double));
    HB_W(C, (size_t)M * N * sizeof(double));
    for (int64_t i = 0; i < M; i++) {
        for (int64_t j = 0; j < N; j++) C[i * N + j] = 0.0;
        for (int64_t k = 0; k < K; k++) {
            const double a = A[i * K + k];
            for (int64_t j = 0; j < N; j++) C[i * N + j] += a * B[k * N + j];
        }
    }
    return HNH_OK;
}
int hnh_leaky_relu_f64(hnh_ctx* c, double* v, double alpha, int64_t n, int stream) {
    HB_OP(c, stream, "hnh_leaky_relu_f64");
    HB_W(v, (size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; i++) v[i] = (v[i] > 0.0 ? v[i] : 0.0) + (v[i] < 0.0 ? v[i] : 0.0) * alpha;
    return HNH_OK;
}
int hnh_relu_store_cols_f64(hnh_ctx* c, double* dst, int64_t ld, int64_t col0, const double* src, int64_t rows, int64_t cols, int stream) {
    HB_OP(c, stream, "hnh_relu_store_cols_f64");
    HB_R(src, (size_t)rows * (size_t)cols * sizeof(double));
    for (int64_t i = 0; i < rows && hb_cur.depth > 0; i++) HB_W(dst + i * ld + col0, (size_t)cols * sizeof(double));
    for (int64_t i = 0; i < rows; i++)
        for (int64_t j = 0; j < cols; j++) dst[i * ld + col0 + j] = src[i * cols + j] > 0.0 ? src[i * cols + j] : 0.0;
    return HNH_OK;
}

/* ---- setup primitives: the host code they stand for (SpmatLocal.hpp:45-52,404-420,454,541-563,117-147), literally ---- */
static uint64_t tuple_key(const hnh_tuple* t, const hnh_tuple_key* k) {
    switch (k->kind) {
        case HNH_KEY_ROW_COL: return (t->r << 32) | (t->c & 0xffffffffull);
        case HNH_KEY_COL_ROW: return (t->c << 32) | (t->r & 0xffffffffull);
        case HNH_KEY_OWNER: {
            const uint64_t rb = (k->transpose ? t->c : t->r) / (uint64_t)k->rows_in_block;
            const uint64_t cb = (k->transpose ? t->r : t->c) / (uint64_t)k->cols_in_block;
            return (uint64_t)(uint32_t)k->owner_table[rb * (uint64_t)k->n_col_blocks + cb];
        }
        default: return t->c / (uint64_t)k->div;
    }
}
static int key_ok(hnh_ctx* c, const hnh_tuple_key* k) {
    if (!k) return fail(c, HNH_ERR_INVALID, "null key");
    if (k->kind == HNH_KEY_OWNER && (k->rows_in_block <= 0 || k->cols_in_block <= 0 || k->n_col_blocks <= 0 || !k->owner_table))
        return fail(c, HNH_ERR_INVALID, "incomplete owner key");
    if (k->kind == HNH_KEY_COL_DIV && k->div <= 0) return fail(c, HNH_ERR_INVALID, "column divisor must be positive");
    if (k->kind < HNH_KEY_ROW_COL || k->kind > HNH_KEY_COL_DIV) return fail(c, HNH_ERR_INVALID, "unknown key kind");
    return HNH_OK;
}
typedef struct { uint64_t key; int64_t idx; } keyed_t;
static int keyed_cmp(const void* a, const void* b) {
    const keyed_t *x = (const keyed_t*)a, *y = (const keyed_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0); /* stable */
}
int hnh_tuples_sort(hnh_ctx* c, hnh_tuple* t, int64_t n, const hnh_tuple_key* k, int key_bits, int stream) {
    (void)key_bits;
    if (n < 0) return fail(c, HNH_ERR_INVALID, "negative size");
    HB_OP(c, stream, "hnh_tuples_sort");
    HB_W(t, (size_t)n * sizeof(hnh_tuple));
    int rc = key_ok(c, k);
    if (rc != HNH_OK) return rc;
    if (n <= 1) return HNH_OK;
    keyed_t* ks = (keyed_t*)malloc(sizeof(keyed_t) * (size_t)n);
    hnh_tuple* cp = (hnh_tuple*)malloc(sizeof(hnh_tuple) * (size_t)n);
    if (!ks || !cp) { free(ks); free(cp); return fail(c, HNH_ERR_NOMEM, "malloc failed"); }
    for (int64_t i = 0; i < n; i++) {
        if ((k->kind == HNH_KEY_ROW_COL || k->kind == HNH_KEY_COL_ROW) && ((t[i].r >> 32) || (t[i].c >> 32))) {
            free(ks); free(cp);
            return fail(c, HNH_ERR_UNSUPPORTED, "a row or column index does not fit 32 bits");
        }
        ks[i].key = tuple_key(&t[i], k);
        ks[i].idx = i;
    }
    qsort(ks, (size_t)n, sizeof(keyed_t), keyed_cmp);
    memcpy(cp, t, sizeof(hnh_tuple) * (size_t)n);
    for (int64_t i = 0; i < n; i++) t[i] = cp[ks[i].idx];
    free(ks); free(cp);
    return HNH_OK;
}
int hnh_tuples_bucket_starts(hnh_ctx* c, const hnh_tuple* t, int64_t n, const hnh_tuple_key* k, int64_t nbuckets, int64_t* starts,
                             int stream) {
    if (n < 0 || nbuckets < 0 || !starts) return fail(c, HNH_ERR_INVALID, "bad argument");
    HB_OP(c, stream, "hnh_tuples_bucket_starts");
    HB_R(t, (size_t)n * sizeof(hnh_tuple));  /* (`starts` is host memory by contract: starts_host) */
    int rc = key_ok(c, k);
    if (rc != HNH_OK) return rc;
    int64_t i = 0;
    for (int64_t b = 0; b <= nbuckets; b++) {
        while (i < n && tuple_key(&t[i], k) < (uint64_t)b) i++;
        starts[b] = i;
    }
    return hb_synchronous(c, stream);
}
int hnh_tuples_transform(hnh_ctx* c, hnh_tuple* t, int64_t n, int swap_rc, uint64_t rmod, uint64_t cmod, int stream) {
    HB_OP(c, stream, "hnh_tuples_transform");
    HB_W(t, (size_t)n * sizeof(hnh_tuple));
    for (int64_t i = 0; i < n; i++) {
        if (swap_rc) { const uint64_t x = t[i].r; t[i].r = t[i].c; t[i].c = x; }
        if (rmod) t[i].r %= rmod;
        if (cmod) t[i].c %= cmod;
    }
    return HNH_OK;
}
/* SpmatLocal.hpp:485-498: ParallelReadMM(..., maximum<double>()) keeps the largest value of duplicate coordinates */
int hnh_tuples_dedup_max(hnh_ctx* c, hnh_tuple* t, int64_t n, int64_t* n_unique, int stream) {
    if (n < 0 || !n_unique) return fail(c, HNH_ERR_INVALID, "bad argument");
    HB_OP(c, stream, "hnh_tuples_dedup_max");
    HB_W(t, (size_t)n * sizeof(hnh_tuple));
    int64_t out = 0;
    for (int64_t i = 0; i < n; i++) {
        if (out > 0 && t[out - 1].r == t[i].r && t[out - 1].c == t[i].c) {
            if (t[i].value > t[out - 1].value) t[out - 1].value = t[i].value;
        } else {
            t[out++] = t[i];
        }
    }
    *n_unique = out;
    return hb_synchronous(c, stream);
}
int hnh_tuples_take_strided(hnh_ctx* c, const hnh_tuple* src, int64_t first, int64_t stride, hnh_tuple* out, int64_t n_out, int stream) {
    if (n_out < 0 || first < 0 || stride <= 0) return fail(c, HNH_ERR_INVALID, "bad argument");
    HB_OP(c, stream, "hnh_tuples_take_strided");
    if (n_out > 0) HB_R(src + first, (size_t)((n_out - 1) * stride + 1) * sizeof(hnh_tuple));
    HB_W(out, (size_t)n_out * sizeof(hnh_tuple));
    for (int64_t i = 0; i < n_out; i++) out[i] = src[first + i * stride];
    return HNH_OK;
}

int hnh_tuples_remap_cols(hnh_ctx* c, hnh_tuple* t, int64_t n, int64_t div, int64_t sub_div, int64_t n_sub, const int64_t* dest,
                          int64_t ndest, int stream) {
    if (n < 0 || div <= 0 || sub_div <= 0 || n_sub <= 0 || ndest <= 0 || !dest) return fail(c, HNH_ERR_INVALID, "bad argument");
    HB_OP(c, stream, "hnh_tuples_remap_cols");
    HB_W(t, (size_t)n * sizeof(hnh_tuple));
    for (int64_t i = 0; i < n; i++) {
        const uint64_t col = t[i].c, in = col % (uint64_t)div;
        const uint64_t seg = (col / (uint64_t)div) * (uint64_t)n_sub + in / (uint64_t)sub_div;
        if ((int64_t)seg >= ndest || dest[seg] < 0) return fail(c, HNH_ERR_INVALID, "a tuple lies in a segment that has no destination");
        t[i].c = (uint64_t)dest[seg] + in % (uint64_t)sub_div;
    }
    return hb_synchronous(c, stream);
}

int hnh_tuples_to_csr(hnh_ctx* c, const hnh_tuple* t, int64_t n, int64_t rows, int64_t cols, int32_t* rowptr, int32_t* col_idx,
                      double* values, int* max_row, int stream) {
    if (n < 0 || rows < 0 || cols < 0 || !rowptr) return fail(c, HNH_ERR_INVALID, "bad argument");
    HB_OP(c, stream, "hnh_tuples_to_csr");
    HB_R(t, (size_t)n * sizeof(hnh_tuple));
    HB_W(rowptr, (size_t)(rows + 1) * sizeof(int32_t));
    HB_W(col_idx, (size_t)n * sizeof(int32_t));
    HB_W(values, (size_t)n * sizeof(double));
    for (int64_t r = 0; r <= rows; r++) rowptr[r] = 0;
    for (int64_t i = 0; i < n; i++) {
        if ((int64_t)t[i].r >= rows || (int64_t)t[i].c >= cols) return fail(c, HNH_ERR_INVALID, "nonzero outside its block");
        if (i > 0 && ((t[i].r < t[i - 1].r) || (t[i].r == t[i - 1].r && t[i].c < t[i - 1].c)))
            return fail(c, HNH_ERR_INVALID, "tuples are not in (row, col) order");
        rowptr[t[i].r + 1]++;
        col_idx[i] = (int32_t)t[i].c;
        values[i] = t[i].value;
    }
    int m = 0;
    for (int64_t r = 0; r < rows; r++) {
        if (rowptr[r + 1] > m) m = rowptr[r + 1];
        rowptr[r + 1] += rowptr[r];
    }
    if (max_row) *max_row = m;
    return hb_synchronous(c, stream);
}

static uint64_t splitmix64_c(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int u64_cmp(const void* a, const void* b) {
    const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
int hnh_generate_er_keys(hnh_ctx* c, uint64_t m, uint64_t n, uint64_t draws, uint64_t seed, uint64_t* keys, int64_t* n_unique, int stream) {
    if (!n_unique || m == 0 || n == 0) return fail(c, HNH_ERR_INVALID, "bad argument");
    HB_OP(c, stream, "hnh_generate_er_keys");
    HB_W(keys, (size_t)draws * sizeof(uint64_t));
    const uint64_t G = 0x9E3779B97F4A7C15ull;
    for (uint64_t k = 0; k < draws; k++) {
        const uint64_t base = seed + (2 * k) * G;
        keys[k] = (splitmix64_c(base) % m) * n + splitmix64_c(base + G) % n;
    }
    qsort(keys, (size_t)draws, sizeof(uint64_t), u64_cmp);
    uint64_t out = 0;
    for (uint64_t k = 0; k < draws; k++)
        if (out == 0 || keys[out - 1] != keys[k]) keys[out++] = keys[k];
    *n_unique = (int64_t)out;
    return hb_synchronous(c, stream);
}
/* er_generator.cpp: rmat_keys, restated */
int hnh_generate_rmat_keys(hnh_ctx* c, int logm, uint64_t edges, double a, double b, double cc, uint64_t seed, int scramble, uint64_t* keys,
                           int64_t* n_unique, int stream) {
    if (!n_unique || logm < 1 || logm > 31 || a < 0.0 || b < 0.0 || cc < 0.0 || a + b + cc > 1.0) return fail(c, HNH_ERR_INVALID, "bad argument");
    HB_OP(c, stream, "hnh_generate_rmat_keys");
    HB_W(keys, (size_t)edges * sizeof(uint64_t));
    const uint64_t G = 0x9E3779B97F4A7C15ull, n = 1ull << logm, mask = n - 1;
    const double ab = a + b, abc = a + b + cc;
    for (uint64_t k = 0; k < edges; k++) {
        uint64_t r = 0, col = 0;
        for (int l = 0; l < logm; l++) {
            const double u = (double)(splitmix64_c(seed + (k * (uint64_t)logm + (uint64_t)l) * G) >> 11) * 0x1.0p-53;
            r = (r << 1) | ((u >= ab) ? 1u : 0u);
            col = (col << 1) | (((u >= a && u < ab) || (u >= abc)) ? 1u : 0u);
        }
        if (scramble) {
            r = (r * 0x9E3779B1ull + 0x7F4A7C15ull) & mask;
            col = (col * 0x9E3779B1ull + 0x7F4A7C15ull) & mask;
        }
        keys[k] = r * n + col;
    }
    qsort(keys, (size_t)edges, sizeof(uint64_t), u64_cmp);
    uint64_t out = 0;
    for (uint64_t k = 0; k < edges; k++)
        if (out == 0 || keys[out - 1] != keys[k]) keys[out++] = keys[k];
    *n_unique = (int64_t)out;
    return hb_synchronous(c, stream);
}
int hnh_tuples_from_keys(hnh_ctx* c, const uint64_t* keys, uint64_t ncols, int64_t first, int64_t stride, double value, hnh_tuple* out,
                         int64_t n_out, int stream) {
    if (n_out < 0 || first < 0 || stride <= 0 || ncols == 0) return fail(c, HNH_ERR_INVALID, "bad argument");
    HB_OP(c, stream, "hnh_tuples_from_keys");
    if (n_out > 0) HB_R(keys + first, (size_t)((n_out - 1) * stride + 1) * sizeof(uint64_t));
    HB_W(out, (size_t)n_out * sizeof(hnh_tuple));
    for (int64_t i = 0; i < n_out; i++) {
        const uint64_t key = keys[first + i * stride];
        out[i].r = key / ncols; out[i].c = key % ncols; out[i].value = value;
    }
    return HNH_OK;
}
int hnh_tuples_relabel(hnh_ctx* c, hnh_tuple* t, int64_t n, const uint64_t* row_label, const uint64_t* col_label, int stream) {
    HB_OP(c, stream, "hnh_tuples_relabel");
    HB_W(t, (size_t)n * sizeof(hnh_tuple));
    for (int64_t i = 0; i < n; i++) { t[i].r = row_label[t[i].r]; t[i].c = col_label[t[i].c]; }
    return HNH_OK;
}

/* ---- the RCCL section of the ABI, EMULATED between ranks that are threads of this process or processes of this host
 * (tests/test_rccl_emulation_cpu.py).  What is emulated is the calling contract the product's default transport (RcclWorld) depends
 * on, as NCCL / RCCL document it:
 *   - a communicator is formed by all n ranks calling init with the same unique id (collective);
 *   - point-to-point operations of one ncclGroupStart/End are issued together; a send and a receive match in the ORDER they were issued
 *     for their (source, destination) pair on the communicator, and their sizes must agree;
 *   - collectives are called by every rank of the communicator in the same order;
 *   - an operation that never finds its partner hangs on the GPU — here it fails after HNH_ORACLE_COMM_WAIT_S (default 120 s).
 * The communicator's state lives in a shared-memory segment named after the unique id (descriptor rings per ordered pair, one meeting
 * point for collectives); bytes move with memcpy inside a process and process_vm_readv between processes.  Inside one process the
 * stream-order checker sees a send as a read of its buffer on the sender's stream, a receive as a write on the receiver's stream that
 * runs behind the matching send, a collective as running behind every rank's contribution. */
#include <fcntl.h>
#include <stdatomic.h>
#include <sys/stat.h>
#define EMU_MAX_RANKS 32
#define EMU_RING 64
#define EMU_MAGIC 0x686e68656d7532ULL /* "hnhemu2" */
typedef struct emu_desc { _Atomic unsigned state; /* 0 free, 1 posted, 2 claimed, 3 consumed */ long pid; unsigned long long addr, bytes; unsigned long token; } emu_desc;
typedef struct emu_pair { _Atomic unsigned long posted, claimed; emu_desc ring[EMU_RING]; } emu_pair;
typedef struct emu_coll {
    _Atomic unsigned long gen;
    _Atomic int arrived, leaving;
    long pid[EMU_MAX_RANKS];
    unsigned long long addr[EMU_MAX_RANKS];
    unsigned long token[EMU_MAX_RANKS];
} emu_coll;
typedef struct emu_shared {
    _Atomic unsigned long long magic;
    _Atomic int n, joined;
    emu_coll coll;
    emu_pair pair[EMU_MAX_RANKS * EMU_MAX_RANKS];
} emu_shared;
typedef struct emu_comm { emu_shared* g; int rank, n; } emu_comm;
typedef struct emu_op { emu_comm* comm; hnh_ctx* ctx; int is_send, peer, stream; void* buf; size_t bytes; emu_desc* desc; } emu_op;
static __thread int emu_depth = 0, emu_nops = 0;
static __thread emu_op emu_ops[1024];
/* clocks of this process's sends / contributions, by token (the checker's edges exist inside one process only) */
#define EMU_TOKENS 4096
static uint32_t (*emu_clk)[HB_T] = NULL;
static unsigned long emu_next_token = 0;

static double emu_wait_limit_ms(void) {
    const char* v = getenv("HNH_ORACLE_COMM_WAIT_S");
    return (v && *v ? atof(v) : 120.0) * 1e3;
}
/* polls until cond; evaluates to 0 when the time limit passed first */
#define EMU_POLL(cond)                                                       \
    ({                                                                       \
        const double t0_ = now_ms(), lim_ = emu_wait_limit_ms();             \
        int ok_ = 1;                                                         \
        for (unsigned spins_ = 0; !(cond); spins_++) {                       \
            if (spins_ > 200) { struct timespec ts_ = {0, 20000}; nanosleep(&ts_, NULL); } \
            if ((spins_ & 255) == 255 && now_ms() - t0_ > lim_) { ok_ = (cond) ? 1 : 0; break; } \
        }                                                                    \
        ok_;                                                                 \
    })
static unsigned long emu_store_clock(void) {  /* the current call's clock, kept for receivers of this process */
    if (hb_cur.depth <= 0) return 0;
    pthread_mutex_lock(&g_mu);
    if (!emu_clk) emu_clk = calloc(EMU_TOKENS, sizeof *emu_clk);
    const unsigned long token = ++emu_next_token;
    if (emu_clk) memcpy(emu_clk[token % EMU_TOKENS], hb_vc[hb_cur.t], sizeof emu_clk[0]);
    pthread_mutex_unlock(&g_mu);
    return token;
}
static void emu_join_clock(long pid, unsigned long token) {
    if (hb_cur.depth <= 0 || !token || pid != (long)getpid() || !emu_clk) return;
    pthread_mutex_lock(&g_mu);
    hb_join(hb_vc[hb_cur.t], emu_clk[token % EMU_TOKENS]);
    pthread_mutex_unlock(&g_mu);
}
static int emu_fetch(void* dst, long pid, unsigned long long addr, size_t bytes) {  /* a peer's bytes */
    if (pid == (long)getpid()) { memcpy(dst, (const void*)(uintptr_t)addr, bytes); return 1; }
    for (size_t done = 0; done < bytes;) {
        struct iovec l = {(char*)dst + done, bytes - done}, r = {(void*)(uintptr_t)(addr + done), bytes - done};
        const ssize_t k = process_vm_readv((pid_t)pid, &l, 1, &r, 1, 0);
        if (k <= 0) return 0;
        done += (size_t)k;
    }
    return 1;
}

int hnh_comm_unique_id(void* id) {
    if (!id) return HNH_ERR_INVALID;
    static _Atomic unsigned long counter = 0;
    unsigned char* b = (unsigned char*)id;
    memset(b, 0, HNH_UNIQUE_ID_BYTES);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    const unsigned long long stamp[4] = {EMU_MAGIC, (unsigned long long)getpid(), ++counter, (unsigned long long)ts.tv_sec * 1000000000ULL + (unsigned long long)ts.tv_nsec};
    memcpy(b, stamp, sizeof stamp);
    return HNH_OK;
}
int hnh_comm_init(hnh_ctx* c, int n, int r, const void* id, void** comm) {
    if (!id || !comm || n <= 0 || r < 0 || r >= n) return fail(c, HNH_ERR_INVALID, "hnh_comm_init: bad argument");
    unsigned long long stamp[4];
    memcpy(stamp, id, sizeof stamp);
    if (stamp[0] != EMU_MAGIC)
        return fail(c, HNH_ERR_UNSUPPORTED, "RCCL transport is not available in the CPU test double (its emulation joins ranks of this host by an id from hnh_comm_unique_id)");
    if (n > EMU_MAX_RANKS) return fail(c, HNH_ERR_UNSUPPORTED, "the RCCL emulation takes up to 32 ranks");
    char name[96];
    snprintf(name, sizeof name, "/hnh_emu_%llx_%llx_%llx", stamp[1], stamp[2], stamp[3]);
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    const int creator = fd >= 0;
    if (creator) {
        if (ftruncate(fd, (off_t)sizeof(emu_shared)) != 0) { close(fd); shm_unlink(name); return fail(c, HNH_ERR_NOMEM, "hnh_comm_init: cannot size the communicator's segment"); }
    } else {
        struct stat st;
        const int ok = EMU_POLL(({ fd = shm_open(name, O_RDWR, 0600); int good = fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(emu_shared); if (!good && fd >= 0) { close(fd); fd = -1; } good; }));
        if (!ok) return fail(c, HNH_ERR_DEVICE, "hnh_comm_init: the communicator's segment never appeared");
    }
    emu_shared* g = (emu_shared*)mmap(NULL, sizeof(emu_shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (g == MAP_FAILED) return fail(c, HNH_ERR_NOMEM, "hnh_comm_init: cannot map the communicator's segment");
    if (creator) {  /* (a fresh segment is zero-filled: every ring empty) */
        atomic_store(&g->n, n);
        atomic_store(&g->magic, EMU_MAGIC);
    } else if (!EMU_POLL(atomic_load(&g->magic) == EMU_MAGIC)) {
        munmap(g, sizeof(emu_shared));
        return fail(c, HNH_ERR_DEVICE, "hnh_comm_init: the communicator's segment was never initialised");
    }
    if (atomic_load(&g->n) != n) { munmap(g, sizeof(emu_shared)); return fail(c, HNH_ERR_INVALID, "hnh_comm_init: the ranks disagree about the communicator's size"); }
    atomic_fetch_add(&g->joined, 1);
    const int all = EMU_POLL(atomic_load(&g->joined) >= n);  /* ncclCommInitRank is collective */
    if (creator) shm_unlink(name);  /* everybody who will ever join has it open (or never will) */
    if (!all) { munmap(g, sizeof(emu_shared)); return fail(c, HNH_ERR_DEVICE, "hnh_comm_init: the other ranks never joined the communicator"); }
    emu_comm* ec = (emu_comm*)calloc(1, sizeof(emu_comm));
    if (!ec) { munmap(g, sizeof(emu_shared)); return fail(c, HNH_ERR_NOMEM, "hnh_comm_init: malloc failed"); }
    ec->g = g;
    ec->rank = r;
    ec->n = n;
    *comm = ec;
    return HNH_OK;
}
int hnh_comm_split(hnh_ctx* c, void* comm, int color, int key, void** nc) {
    (void)comm; (void)color; (void)key; (void)nc;
    return fail(c, HNH_ERR_UNSUPPORTED, "hnh_comm_split is not emulated (the product does not split RCCL communicators)");
}
int hnh_comm_destroy(hnh_ctx* c, void* comm) {
    (void)c;
    emu_comm* ec = (emu_comm*)comm;
    if (!ec) return HNH_OK;
    munmap(ec->g, sizeof(emu_shared));
    free(ec);
    return HNH_OK;
}
int hnh_comm_identity(hnh_ctx* c, void* comm, int* nranks, int* rank, int* device) {
    emu_comm* ec = (emu_comm*)comm;
    if (!c || !ec || !nranks || !rank || !device) return HNH_ERR_INVALID;
    *nranks = ec->n;
    *rank = ec->rank;
    *device = c->device;
    return HNH_OK;
}
static void emu_declare_send(emu_op* o) {
    HB_OP(o->ctx, o->stream, "ncclSend");
    HB_R(o->buf, o->bytes);
    o->desc->token = emu_store_clock();
}
static void emu_declare_recv(emu_op* o, const emu_desc* d) {
    HB_OP(o->ctx, o->stream, "ncclRecv");
    emu_join_clock(d->pid, d->token);  /* behind the matching send */
    HB_W(o->buf, o->bytes);
}
/* issues the operations collected since the outermost group began */
static int emu_issue(hnh_ctx* c) {
    int rc = HNH_OK;
    const int nops = emu_nops;
    emu_nops = 0;
    for (int i = 0; i < nops && rc == HNH_OK; i++) {  /* 1. every send is posted: the next descriptor of its pair's ring */
        emu_op* o = &emu_ops[i];
        if (!o->is_send) continue;
        emu_pair* pr = &o->comm->g->pair[o->comm->rank * EMU_MAX_RANKS + o->peer];
        emu_desc* d = &pr->ring[atomic_load(&pr->posted) % EMU_RING];
        if (!EMU_POLL(atomic_load(&d->state) == 0)) { rc = fail(c, HNH_ERR_DEVICE, "ncclSend: too many sends of this pair were never received (this hangs on the GPU)"); break; }
        o->desc = d;
        d->pid = (long)getpid();
        d->addr = (unsigned long long)(uintptr_t)o->buf;
        d->bytes = o->bytes;
        emu_declare_send(o);
        atomic_store(&d->state, 1);
        atomic_fetch_add(&pr->posted, 1);
    }
    for (int i = 0; i < nops && rc == HNH_OK; i++) {  /* 2. every receive takes the oldest unclaimed send of its pair */
        emu_op* o = &emu_ops[i];
        if (o->is_send) continue;
        emu_pair* pr = &o->comm->g->pair[o->peer * EMU_MAX_RANKS + o->comm->rank];
        const unsigned long idx = atomic_load(&pr->claimed);
        emu_desc* d = &pr->ring[idx % EMU_RING];
        if (!EMU_POLL(atomic_load(&pr->posted) > idx && atomic_load(&d->state) == 1)) {
            rc = fail(c, HNH_ERR_DEVICE, "ncclRecv: the peer never issued the matching ncclSend (this hangs on the GPU)");
            break;
        }
        atomic_store(&d->state, 2);
        atomic_fetch_add(&pr->claimed, 1);
        if (d->bytes != o->bytes) rc = fail(c, HNH_ERR_INVALID, "ncclSend / ncclRecv sizes of a matching pair differ (undefined on the GPU)");
        else {
            emu_declare_recv(o, d);
            if (!emu_fetch(o->buf, d->pid, d->addr, o->bytes)) rc = fail(c, HNH_ERR_DEVICE, "ncclRecv: cannot read the peer's memory (process_vm_readv: ptrace permission?)");
        }
        atomic_store(&d->state, 3);  /* consumed: the sender may go on */
    }
    for (int i = 0; i < nops; i++) {  /* 3. a send is complete when its bytes were taken; its descriptor is free again */
        emu_op* o = &emu_ops[i];
        if (!o->is_send || !o->desc) continue;
        if (rc == HNH_OK && !EMU_POLL(atomic_load(&o->desc->state) == 3))
            rc = fail(c, HNH_ERR_DEVICE, "ncclSend: the peer never issued the matching ncclRecv (this hangs on the GPU)");
        if (atomic_load(&o->desc->state) == 3) atomic_store(&o->desc->state, 0);
    }
    return rc;
}
int hnh_comm_sendrecv(hnh_ctx* c, void* comm, const void* s, size_t sb, int dst, void* r, size_t rb, int src, int st) {
    emu_comm* ec = (emu_comm*)comm;
    if (!ec) return fail(c, HNH_ERR_INVALID, "hnh_comm_sendrecv: null communicator");
    if (dst < 0 || dst >= ec->n || src < 0 || src >= ec->n) return fail(c, HNH_ERR_INVALID, "hnh_comm_sendrecv: peer out of range");
    if (emu_nops + 2 > (int)(sizeof emu_ops / sizeof emu_ops[0])) return fail(c, HNH_ERR_UNSUPPORTED, "too many operations in one group for the emulation");
    if (sb) emu_ops[emu_nops++] = (emu_op){ec, c, 1, dst, st, (void*)s, sb, NULL};
    if (rb) emu_ops[emu_nops++] = (emu_op){ec, c, 0, src, st, r, rb, NULL};
    return emu_depth > 0 ? HNH_OK : emu_issue(c);
}
int hnh_comm_group_begin(hnh_ctx* c) { (void)c; emu_depth++; return HNH_OK; }
int hnh_comm_group_end(hnh_ctx* c) {
    if (emu_depth <= 0) return fail(c, HNH_ERR_INVALID, "ncclGroupEnd without ncclGroupStart");
    return --emu_depth > 0 ? HNH_OK : emu_issue(c);
}
/* collectives: everybody contributes, everybody reads everybody's contribution, nobody leaves before everybody has read */
static int emu_collective(hnh_ctx* c, emu_comm* ec, const void* send, size_t send_bytes, void* recv, size_t recv_bytes, int stream, const char* name,
                          void (*combine)(int n, int me, void* const* contrib, void* out, size_t unit), size_t unit) {
    emu_coll* k = &ec->g->coll;
    const int n = ec->n;
    if (emu_depth > 0) return fail(c, HNH_ERR_UNSUPPORTED, "collectives inside a group are not emulated");
    if (!EMU_POLL(atomic_load(&k->leaving) == 0)) return fail(c, HNH_ERR_DEVICE, "a previous collective never completed");
    const unsigned long gen = atomic_load(&k->gen);
    {
        HB_OP(c, stream, name);
        HB_R(send, send_bytes);
        k->token[ec->rank] = emu_store_clock();
    }
    k->pid[ec->rank] = (long)getpid();
    k->addr[ec->rank] = (unsigned long long)(uintptr_t)send;
    if (atomic_fetch_add(&k->arrived, 1) + 1 == n) {
        atomic_store(&k->arrived, 0);
        atomic_store(&k->leaving, n);
        atomic_fetch_add(&k->gen, 1);
    }
    if (!EMU_POLL(atomic_load(&k->gen) != gen)) return fail(c, HNH_ERR_DEVICE, "a collective was not called by every rank of the communicator (this hangs on the GPU)");
    /* everybody's contribution, fetched whole (peers in other processes: a copy); nothing is written before everybody has read */
    int rc = HNH_OK;
    void* contrib[EMU_MAX_RANKS];
    for (int r = 0; r < n; r++) {
        contrib[r] = malloc(send_bytes ? send_bytes : 1);
        if (!contrib[r] || !emu_fetch(contrib[r], k->pid[r], k->addr[r], send_bytes)) rc = fail(c, HNH_ERR_DEVICE, "a collective cannot read a peer's contribution");
    }
    void* tmp = malloc(recv_bytes ? recv_bytes : 1);
    if (!tmp) rc = fail(c, HNH_ERR_NOMEM, "malloc failed");
    if (rc == HNH_OK) combine(n, ec->rank, contrib, tmp, unit);
    {
        HB_OP(c, stream, name);
        for (int r = 0; r < n; r++) emu_join_clock(k->pid[r], k->token[r]);
        HB_W(recv, recv_bytes);
    }
    for (int r = 0; r < n; r++) free(contrib[r]);
    atomic_fetch_sub(&k->leaving, 1);
    const int ok = EMU_POLL(atomic_load(&k->leaving) == 0 || atomic_load(&k->gen) != gen + 1);
    if (rc == HNH_OK && tmp) memcpy(recv, tmp, recv_bytes);
    free(tmp);
    if (rc != HNH_OK) return rc;
    return ok ? HNH_OK : fail(c, HNH_ERR_DEVICE, "a collective did not complete on every rank");
}
static void emu_allgather(int n, int me, void* const* contrib, void* out, size_t unit) {
    (void)me;
    for (int r = 0; r < n; r++) memcpy((char*)out + (size_t)r * unit, contrib[r], unit);
}
static void emu_reduce_scatter(int n, int me, void* const* contrib, void* out, size_t unit) {  /* unit = doubles per rank; rank order: deterministic */
    double* o = (double*)out;
    for (size_t i = 0; i < unit; i++) o[i] = 0.0;
    for (int r = 0; r < n; r++)
        for (size_t i = 0; i < unit; i++) o[i] += ((const double*)contrib[r])[(size_t)me * unit + i];
}
static void emu_allreduce(int n, int me, void* const* contrib, void* out, size_t unit) {
    (void)me;
    double* o = (double*)out;
    for (size_t i = 0; i < unit; i++) o[i] = 0.0;
    for (int r = 0; r < n; r++)
        for (size_t i = 0; i < unit; i++) o[i] += ((const double*)contrib[r])[i];
}
int hnh_comm_allgather(hnh_ctx* c, void* comm, const void* s, void* r, size_t b, int st) {
    emu_comm* ec = (emu_comm*)comm;
    if (!ec) return fail(c, HNH_ERR_INVALID, "hnh_comm_allgather: null communicator");
    return emu_collective(c, ec, s, b, r, b * (size_t)ec->n, st, "ncclAllGather", emu_allgather, b);
}
int hnh_comm_reduce_scatter_f64(hnh_ctx* c, void* comm, const double* s, double* r, size_t n, int st) {
    emu_comm* ec = (emu_comm*)comm;
    if (!ec) return fail(c, HNH_ERR_INVALID, "hnh_comm_reduce_scatter_f64: null communicator");
    return emu_collective(c, ec, s, n * (size_t)ec->n * sizeof(double), r, n * sizeof(double), st, "ncclReduceScatter", emu_reduce_scatter, n);
}
int hnh_comm_allreduce_f64(hnh_ctx* c, void* comm, const double* s, double* r, size_t n, int st) {
    emu_comm* ec = (emu_comm*)comm;
    if (!ec) return fail(c, HNH_ERR_INVALID, "hnh_comm_allreduce_f64: null communicator");
    return emu_collective(c, ec, s, n * sizeof(double), r, n * sizeof(double), st, "ncclAllReduce", emu_allreduce, n);
}

/* ---- the "ipc" section of the ABI between PROCESSES of this host: a handle is (pid, address); an opened block is a reserved,
 * inaccessible address range standing for the peer's block; a pull reads the peer's memory with process_vm_readv; flag words
 * live in the processes' common shared memory and, the double being synchronous, are stored / awaited on the spot. */
typedef struct { long pid; unsigned long long base, bytes; } ipc_handle;
typedef struct opened { char* local; size_t bytes; long pid; unsigned long long remote; struct opened* next; } opened;
static opened* g_opened = NULL;
int hnh_ipc_export(hnh_ctx* c, const void* ptr, void* handle_host, uint64_t* offset, uint64_t* alloc_bytes) {
    if (!ptr || !handle_host || !offset || !alloc_bytes) return HNH_ERR_INVALID;
    ipc_handle h = {(long)getpid(), 0, 0};
    pthread_mutex_lock(&g_mu);
    for (block* b = g_blocks; b; b = b->next)
        if ((const char*)ptr >= b->base && (const char*)ptr < b->base + b->bytes) { h.base = (unsigned long long)(uintptr_t)b->base; h.bytes = b->bytes; break; }
    pthread_mutex_unlock(&g_mu);
    if (!h.bytes) return fail(c, HNH_ERR_INVALID, "hnh_ipc_export: not a block of hnh_malloc");
    memset(handle_host, 0, HNH_IPC_HANDLE_BYTES);
    memcpy(handle_host, &h, sizeof h);
    *offset = (uint64_t)((uintptr_t)ptr - (uintptr_t)h.base);
    *alloc_bytes = h.bytes;
    return HNH_OK;
}
int hnh_ipc_open(hnh_ctx* c, const void* handle_host, uint64_t alloc_bytes, void** base) {
    if (!handle_host || !base) return HNH_ERR_INVALID;
    ipc_handle h;
    memcpy(&h, handle_host, sizeof h);
    if (h.bytes != alloc_bytes || !h.bytes) return fail(c, HNH_ERR_INVALID, "hnh_ipc_open: bad handle");
    void* r = mmap(NULL, (size_t)h.bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    opened* o = (opened*)malloc(sizeof(opened));
    if (r == MAP_FAILED || !o) return fail(c, HNH_ERR_NOMEM, "hnh_ipc_open: cannot reserve the address range");
    o->local = (char*)r; o->bytes = (size_t)h.bytes; o->pid = h.pid; o->remote = h.base;
    pthread_mutex_lock(&g_mu);
    o->next = g_opened;
    g_opened = o;
    pthread_mutex_unlock(&g_mu);
    *base = r;
    return HNH_OK;
}
int hnh_ipc_close(hnh_ctx* c, void* base) {
    (void)c;
    pthread_mutex_lock(&g_mu);
    for (opened** q = &g_opened; *q; q = &(*q)->next)
        if ((*q)->local == (char*)base) { opened* d = *q; *q = d->next; munmap(d->local, d->bytes); free(d); break; }
    pthread_mutex_unlock(&g_mu);
    return HNH_OK;
}
int hnh_ipc_pull(hnh_ctx* c, int stream, int n, void* const* dst, const void* const* src, const size_t* bytes, int mode, int wgs) {
    (void)mode; (void)wgs;
    HB_OP(c, stream, "hnh_ipc_pull");  /* (the sources live in other processes: only this side of the transfer is seen) */
    for (int i = 0; i < n; i++) {
        if (!bytes[i]) continue;
        HB_W(dst[i], bytes[i]);
        long pid = 0;
        unsigned long long remote = 0;
        pthread_mutex_lock(&g_mu);
        for (opened* o = g_opened; o; o = o->next)
            if ((const char*)src[i] >= o->local && (const char*)src[i] + bytes[i] <= o->local + o->bytes) { pid = o->pid; remote = o->remote + (unsigned long long)((const char*)src[i] - o->local); break; }
        pthread_mutex_unlock(&g_mu);
        if (!pid) return fail(c, HNH_ERR_INVALID, "hnh_ipc_pull: source is not inside an opened block");
        if (pid == (long)getpid()) HB_R((const void*)(uintptr_t)remote, bytes[i]);  /* ranks that are threads of this process: the checker sees both ends */
        size_t done = 0;
        while (done < bytes[i]) {
            struct iovec l = {(char*)dst[i] + done, bytes[i] - done}, r = {(void*)(uintptr_t)(remote + done), bytes[i] - done};
            ssize_t k = process_vm_readv((pid_t)pid, &l, 1, &r, 1, 0);
            if (k <= 0) return fail(c, HNH_ERR_DEVICE, "hnh_ipc_pull: process_vm_readv failed (ptrace permission?)");
            done += (size_t)k;
        }
    }
    return HNH_OK;
}
int hnh_ipc_flags_register(hnh_ctx* c, void* host_shm, size_t bytes, void** device_view) {
    if (c) { c->flags_base = (char*)host_shm; c->flags_bytes = bytes; }
    *device_view = host_shm;
    return HNH_OK;
}
int hnh_ipc_flags_unregister(hnh_ctx* c, void* host_shm) { (void)c; (void)host_shm; return HNH_OK; }
/* Flag words order streams of different ranks like events do: a wait for value v runs behind the write that raised the word to v (values
 * only grow).  The checker keeps the clocks of the recent writes of THIS process, so ranks that are threads of one process (the ipc-pull
 * transport run that way by tests/test_stream_order_cpu.py) have their whole protocol checked; writes of other processes leave no edge. */
typedef struct hb_flag_write { void* f; uint64_t v; unsigned long seq; uint32_t clk[HB_T]; } hb_flag_write;
#define HB_FLAG_WRITES 1024
static hb_flag_write* hb_flag_ring = NULL;
static unsigned long hb_flag_next = 0;
/* every rank maps the session's segment at its own address: a flag word is named by its offset in the registered region */
static void* hb_flag_name(hnh_ctx* c, void* f) {
    if (c && c->flags_base && (char*)f >= c->flags_base && (char*)f < c->flags_base + c->flags_bytes) return (void*)(uintptr_t)((char*)f - c->flags_base + 1);
    return f;
}
int hnh_stream_write_flag(hnh_ctx* c, int stream, void* f, uint64_t v) {
    if (c && c->hb_slot >= 0 && stream >= 0 && stream < HB_HOST) {
        pthread_mutex_lock(&g_mu);
        if (!hb_flag_ring) hb_flag_ring = (hb_flag_write*)calloc(HB_FLAG_WRITES, sizeof(hb_flag_write));
        if (hb_flag_ring) {
            hb_flag_write* w = &hb_flag_ring[hb_flag_next % HB_FLAG_WRITES];
            w->f = hb_flag_name(c, f);
            w->v = v;
            w->seq = ++hb_flag_next;
            memcpy(w->clk, hb_vc[hb_tick(c->hb_slot, stream)], sizeof(w->clk));
        }
        pthread_mutex_unlock(&g_mu);
    }
    __atomic_store_n((uint64_t*)f, v, __ATOMIC_RELEASE);
    return HNH_OK;
}
int hnh_stream_wait_flag(hnh_ctx* c, int stream, void* f, uint64_t v) {
    const double t0 = now_ms();
    for (unsigned spins = 0; __atomic_load_n((uint64_t*)f, __ATOMIC_ACQUIRE) < v; spins++) {
        if (spins > 1000) { struct timespec ts = {0, 50000}; nanosleep(&ts, NULL); }
        if ((spins & 1023) == 0 && now_ms() - t0 > 120000.0) return fail(c, HNH_ERR_DEVICE, "hnh_stream_wait_flag: the peer never raised the flag");
    }
    if (c && c->hb_slot >= 0 && stream >= 0 && stream < HB_HOST && !getenv("HNH_ORDER_CHECK_DROP_WAITS")) {
        pthread_mutex_lock(&g_mu);
        const hb_flag_write* best = NULL;  /* the write that raised the word to (at least) v: the smallest recorded value >= v */
        void* name = hb_flag_name(c, f);
        for (int i = 0; hb_flag_ring && i < HB_FLAG_WRITES; i++)
            if (hb_flag_ring[i].f == name && hb_flag_ring[i].v >= v &&
                (!best || hb_flag_ring[i].v < best->v || (hb_flag_ring[i].v == best->v && hb_flag_ring[i].seq > best->seq)))
                best = &hb_flag_ring[i];
        const int t = hb_tick(c->hb_slot, stream);
        if (best) hb_join(hb_vc[t], best->clk);
        pthread_mutex_unlock(&g_mu);
    }
    return HNH_OK;
}

/* ---- block descriptors: the double has no structure-only work to cache, a plan is an empty token; the _p entry points are the
 * _x / _w entry points on the descriptor's fields */
struct hnh_csr_plan { int unused; };
int hnh_csr_plan_create(hnh_ctx* c, hnh_csr_plan** out) {
    if (!out) return HNH_ERR_INVALID;
    *out = (hnh_csr_plan*)calloc(1, sizeof(hnh_csr_plan));
    return *out ? HNH_OK : fail(c, HNH_ERR_NOMEM, "malloc failed");
}
int hnh_csr_plan_destroy(hnh_ctx* c, hnh_csr_plan* p) { (void)c; free(p); return HNH_OK; }
static const hnh_csr_window whole_block = {NULL, NULL, 1};
int hnh_sddmm_csr_p(hnh_ctx* c, const hnh_csr_block* b, double* values, const double* X, const double* Y, int R, unsigned flags,
                    const hnh_csr_window* w, int stream) {
    if (!b) return fail(c, HNH_ERR_INVALID, "null block");
    HB_OP(c, stream, "hnh_sddmm_csr_p");
    if (flags & HNH_FUSED_VALUES_OVERWRITE) { /* "known to be zero": the double does not rely on it */
        const hnh_csr_window* ww = w ? w : &whole_block;
        for (int64_t r = 0; r < b->rows; r++) {
            const int32_t lo = ww->beg ? ww->beg[r] : b->rowptr[r], hi = ww->end ? ww->end[r] : b->rowptr[r + 1];
            for (int32_t i = lo; i < hi; i++) values[i] = 0.0;
        }
    }
    return hnh_sddmm_csr_w(c, b->rows, b->rowptr, b->col_idx, values, X, Y, R, b->nnz, b->max_row_nnz, w ? w : &whole_block, stream);
}
/* dst[e] (+)= scale[e] * <X[i_e,:], Y[j_e,:]>: the SDDMM with its closing Hadamard folded in (sparse_kernels.cpp:44-55 followed by
 * 15D_dense_shift.hpp:366).  With HNH_FUSED_VALUES_OVERWRITE the destination is written without being read. */
int hnh_sddmm_csr_ps(hnh_ctx* c, const hnh_csr_block* b, double* dst, const double* scale, const double* X, const double* Y, int R,
                     unsigned flags, const hnh_csr_window* w, int stream) {
    if (!scale) return hnh_sddmm_csr_p(c, b, dst, X, Y, R, flags, w, stream);
    if (!b) return fail(c, HNH_ERR_INVALID, "null block");
    if (flags & ~HNH_FUSED_VALUES_OVERWRITE) return fail(c, HNH_ERR_INVALID, "unknown flag");
    if (b->rows < 0 || R <= 0) return fail(c, HNH_ERR_INVALID, "bad argument");
    if (scale == dst) return fail(c, HNH_ERR_INVALID, "scale aliases dst");
    const hnh_csr_window* ww = w ? w : &whole_block;
    HB_OP(c, stream, "hnh_sddmm_csr_ps");
    hb_row_pass(b->rows, b->rowptr, b->col_idx, ww->beg, ww->end, dst, 1, scale, Y, R);
    HB_R(X, (size_t)b->rows * R * sizeof(double));
    for (int64_t r = 0; r < b->rows; r++) {
        const int32_t lo = ww->beg ? ww->beg[r] : b->rowptr[r], hi = ww->end ? ww->end[r] : b->rowptr[r + 1];
        for (int32_t i = lo; i < hi; i++) {
            const double* Arow = X + (int64_t)R * r;
            const double* Brow = Y + (int64_t)R * b->col_idx[i];
            double value = 0.0;
            for (int k = 0; k < R; k++) value += Arow[k] * Brow[k];
            if (flags & HNH_FUSED_VALUES_OVERWRITE) dst[i] = scale[i] * value;
            else dst[i] += scale[i] * value;
        }
    }
    return HNH_OK;
}
int hnh_spmm_csr_p(hnh_ctx* c, const hnh_csr_block* b, const double* values, const double* X, double* Out, int R, const hnh_csr_window* w, int stream) {
    if (!b) return fail(c, HNH_ERR_INVALID, "null block");
    return hnh_spmm_csr_w(c, b->rows, b->rowptr, b->col_idx, values, X, Out, R, b->nnz, b->max_row_nnz, w ? w : &whole_block, stream);
}
int hnh_spmm_csr_pf(hnh_ctx* c, const hnh_csr_block* b, const double* values, const double* X, double* Out, int R, unsigned flags,
                    const hnh_csr_window* w, int stream) {
    if (!b) return fail(c, HNH_ERR_INVALID, "null block");
    if (flags & ~HNH_FUSED_OUT_OVERWRITE) return fail(c, HNH_ERR_INVALID, "hnh_spmm_csr_pf: unknown flag");
    if ((flags & HNH_FUSED_OUT_OVERWRITE) && w) return fail(c, HNH_ERR_INVALID, "hnh_spmm_csr_pf: a window of a block cannot overwrite its rows");
    if ((flags & HNH_FUSED_OUT_OVERWRITE) && b->rows > 0) {  /* the stored rows start from nothing: what was there is never read */
        if (!Out) return fail(c, HNH_ERR_INVALID, "null pointer");
        HB_OP(c, stream, "hnh_spmm_csr_pf (store)");
        HB_W(Out, (size_t)b->rows * R * sizeof(double));
        memset(Out, 0, sizeof(double) * (size_t)b->rows * (size_t)R);
    }
    return hnh_spmm_csr_w(c, b->rows, b->rowptr, b->col_idx, values, X, Out, R, b->nnz, b->max_row_nnz, w ? w : &whole_block, stream);
}
int hnh_fused_sddmm_spmm_csr_p(hnh_ctx* c, const hnh_csr_block* b, double* values, const double* svalues, const double* X, const double* Y,
                               double* Out, int R, unsigned flags, const hnh_fused_extras* ex, const hnh_csr_window* w, int stream) {
    if (!b) return fail(c, HNH_ERR_INVALID, "null block");
    return hnh_fused_sddmm_spmm_csr_w(c, b->rows, b->rowptr, b->col_idx, values, svalues, X, Y, Out, R, flags, b->nnz, b->max_row_nnz, ex,
                                      w ? w : &whole_block, stream);
}
