/*
 * TEST INFRASTRUCTURE ONLY — a happens-before checker for the host layer's STREAM PROTOCOL, built into the CPU test double.
 *
 * The test double runs every call synchronously, so a missing event wait between the compute, communication and auxiliary
 * streams (or between the streams of two logical ranks) can never show up as a wrong number on CPU.  With HNH_ORDER_CHECK=1
 * (or hnh_oracle_order_enable(1)) the double therefore also keeps, for what the host layer ENQUEUES,
 *   - one vector clock per timeline: the three streams of every context plus the host thread that drives it;
 *     the edges are exactly HIP's: program order on a stream, enqueue order host -> stream, hnh_event_record / _wait / _sync,
 *     hnh_stream_sync and hnh_free (a device-wide synchronisation).  Host-side meetings of the threads that drive several
 *     contexts (the loopback transport's barrier) are deliberately NOT edges: the schedules order device work with events
 *     only — the reference's per-step world barrier is gone — and the checker holds them to that;
 *   - for every block of "device" memory the byte ranges each call read and wrote, with the call's timeline and clock value
 *     (gathered operands: the rows the block's column indices actually address; windows: their own nonzeros; copies: the bytes).
 * Two accesses to overlapping bytes, at least one a write, on different timelines, neither ordered before the other, are a
 * RACE of the stream protocol — what would be a data race on a GPU, where the streams really run concurrently — and are
 * reported with both calls' names (hnh_oracle_order_report).  The shift schedules, the mesh fetch with its landing-buffer
 * windows, the two-half accumulator rings, the GAT pipeline on the auxiliary stream and the caching allocator's recycling
 * are all checked this way in tests/test_stream_order_cpu.py.  Nothing here changes what the double computes.  With the same clocks the
 * double also delivers device-to-host copies only when the host synchronises past them (hb_deliver in hnh_oracle_backend.c).
 */
#ifndef HNH_STREAM_ORDER_H
#define HNH_STREAM_ORDER_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define HB_CTX_SLOTS 32
#define HB_LANES 4                       /* per context: stream 0 (compute), 1 (communication), 2 (auxiliary), 3 = the host thread */
#define HB_T (HB_CTX_SLOTS * HB_LANES)   /* timelines */
#define HB_HOST 3
#define HB_MAX_RECS 2048                 /* access records kept per block (older ones are dropped: fewer reports, never false ones) */
#define HB_REPORT_BYTES 16384

typedef struct hb_rec {
    uint64_t lo, hi;   /* byte offsets in the block */
    uint32_t clk;      /* value of the accessing timeline's own clock component */
    uint16_t t;        /* timeline */
    uint8_t write;
    const char* op;    /* static string: the ABI call */
} hb_rec;

typedef struct hb_event {
    double ms;         /* (first member: hnh_event_elapsed_ms reads it) */
    uint32_t* clk;     /* HB_T entries, the recording stream's clock at the record; NULL = never recorded under the checker */
} hb_event;

static int hb_on = -1;                    /* -1 = read HNH_ORDER_CHECK at first use */
static uint32_t hb_vc[HB_T][HB_T];
static unsigned char hb_slot_used[HB_CTX_SLOTS];
static long hb_races = 0, hb_checked = 0;
static long hb_wait_count = 0, hb_wait_drop = -1;   /* single-fault injection of the checker's own tests */
static char hb_report[HB_REPORT_BYTES];
static size_t hb_report_len = 0;
static __thread struct { int depth, t, strict; const char* name; } hb_cur;

/* a process that ends with unreported races says so and fails (a pytest session drains the report first: tests/conftest.py) */
static void hb_at_exit(void) {
    if (hb_races > 0) {
        fprintf(stderr, "[stream-order checker] %ld races of the stream protocol in process %d:\n%s", hb_races, (int)getpid(), hb_report);
        fflush(stderr);
        _exit(86);
    }
}
static int hb_enabled(void) {
    if (hb_on < 0) {
        const char* v = getenv("HNH_ORDER_CHECK");
        hb_on = (v && *v && strcmp(v, "0") != 0) ? 1 : 0;
        if (hb_on) atexit(hb_at_exit);
    }
    return hb_on;
}
static void hb_join(uint32_t* into, const uint32_t* from) {
    for (int i = 0; i < HB_T; i++)
        if (from[i] > into[i]) into[i] = from[i];
}
/* (all of the following run under g_mu, the double's one lock) */
static int hb_slot_take(void) {
    for (int s = 0; s < HB_CTX_SLOTS; s++)
        if (!hb_slot_used[s]) { hb_slot_used[s] = 1; return s; }
    return -1;  /* more live contexts than slots: this one is not checked */
}
/* a call is being enqueued on `lane` of context slot `slot`: after everything its host thread did, next in the stream's order */
static int hb_tick(int slot, int lane) {
    const int t = slot * HB_LANES + lane;
    if (lane != HB_HOST) hb_join(hb_vc[t], hb_vc[slot * HB_LANES + HB_HOST]);
    hb_vc[t][t]++;
    return t;
}
static void hb_note_race(const hb_rec* r, int t, int write, const char* op, const void* base, uint64_t lo, uint64_t hi) {
    hb_races++;
    if (hb_report_len + 400 < HB_REPORT_BYTES)
        hb_report_len += (size_t)snprintf(hb_report + hb_report_len, HB_REPORT_BYTES - hb_report_len,
                                          "RACE %s (%s, context %d %s) vs %s (%s, context %d %s): block %p bytes [%llu, %llu) and [%llu, %llu)\n",
                                          op, write ? "write" : "read", t / HB_LANES, (const char*[]){"compute", "comm", "aux", "host"}[t % HB_LANES],
                                          r->op, r->write ? "write" : "read", r->t / HB_LANES, (const char*[]){"compute", "comm", "aux", "host"}[r->t % HB_LANES],
                                          base, (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)r->lo, (unsigned long long)r->hi);
    if (getenv("HNH_ORDER_CHECK_ABORT")) { fputs(hb_report, stderr); abort(); }
}
/* a call that misuses memory in a way the CPU forgives and the GPU does not (counted with the races) */
static void hb_note_misuse(const char* op, int write, const char* what, const void* p, size_t bytes) {
    hb_races++;
    if (hb_report_len + 300 < HB_REPORT_BYTES)
        hb_report_len += (size_t)snprintf(hb_report + hb_report_len, HB_REPORT_BYTES - hb_report_len, "MISUSE %s (%s of %zu bytes at %p) %s\n", op,
                                          write ? "write" : "read", bytes, p, what);
    if (getenv("HNH_ORDER_CHECK_ABORT")) { fputs(hb_report, stderr); abort(); }
}
/* one access of the current call to bytes [lo, hi) of a block whose records are (*recs)[0 .. *n) */
static void hb_touch(hb_rec** recs, int* n, int* cap, const void* base, uint64_t lo, uint64_t hi, int write, int t, const char* op) {
    const uint32_t* mine = hb_vc[t];
    int keep = 0;
    hb_checked++;
    for (int i = 0; i < *n; i++) {
        hb_rec* r = &(*recs)[i];
        const int overlap = r->lo < hi && lo < r->hi;
        const int before = (r->t == t) || (mine[r->t] >= r->clk);   /* r happens before this access */
        if (overlap && (r->write || write) && !before) hb_note_race(r, t, write, op, base, lo, hi);
        /* records this access supersedes: covered by it, ordered before it, and no weaker than it (a write covers all, a read covers reads) */
        if (before && r->lo >= lo && r->hi <= hi && (write || !r->write)) continue;
        (*recs)[keep++] = *r;
    }
    *n = keep;
    if (*n >= HB_MAX_RECS) { memmove(*recs, *recs + HB_MAX_RECS / 4, sizeof(hb_rec) * (size_t)(*n - HB_MAX_RECS / 4)); *n -= HB_MAX_RECS / 4; }
    if (*n >= *cap) {
        const int ncap = *cap ? *cap * 2 : 16;
        hb_rec* p = (hb_rec*)realloc(*recs, sizeof(hb_rec) * (size_t)ncap);
        if (!p) return;
        *recs = p; *cap = ncap;
    }
    (*recs)[*n] = (hb_rec){lo, hi, mine[t], (uint16_t)t, (uint8_t)write, op};
    (*n)++;
}
#endif
