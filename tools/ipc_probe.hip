// Probe of the cross-process primitives the IPC transport is built on — run on ONE GPU (two processes share it).
//   hipcc --offload-arch=gfx950 -O2 tools/ipc_probe.hip -o tools/ipc_probe -lrt && tools/ipc_probe
// Answers, per mechanism, "does it work between two processes" and "what does a round trip cost":
//   1. hipIpcGetMemHandle / hipIpcOpenMemHandle of a hipMalloc block (interior pointers via base + offset), read by
//      hipMemcpyAsync and by a kernel;
//   2. flag words in a POSIX shared-memory segment registered with hipHostRegister, written / awaited by
//      hipStreamWriteValue64 / hipStreamWaitValue64 (command processor, no CUs);
//   3. the same flags written / awaited by one-lane kernels (system-scope atomics);
//   4. interprocess events (hipEventInterprocess + hipIpcGetEventHandle).
// The parent forks BEFORE any HIP call; both processes then use device 0.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct Shared {
    std::atomic<int> stage[2];
    hipIpcMemHandle_t mem[2];
    hipIpcEventHandle_t ev[2];
    size_t offset[2];
    int result[2][16];
    double timing[2][16];
    alignas(128) unsigned long long flags[256];
};

#define CK(x)                                                                                              \
    do {                                                                                                   \
        hipError_t e_ = (x);                                                                               \
        if (e_ != hipSuccess) {                                                                            \
            std::fprintf(stderr, "[rank %d] %s -> %s (line %d)\n", g_rank, #x, hipGetErrorString(e_), __LINE__); \
            return 1;                                                                                      \
        }                                                                                                  \
    } while (0)

static int g_rank = 0;

__global__ void fill_kernel(double* p, size_t n, double v) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = v + (double)i;
}
__global__ void sum_kernel(const double* p, size_t n, double* out) {
    double s = 0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    atomicAdd(out, s);
}
__global__ void flag_write_kernel(unsigned long long* f, unsigned long long v) {
    __threadfence_system();
    __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void flag_wait_kernel(const unsigned long long* f, unsigned long long v) {
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) __builtin_amdgcn_s_sleep(8);
}

static void sync_stage(Shared* sh, int me, int value) {
    sh->stage[me].store(value);
    while (sh->stage[1 - me].load() < value) usleep(50);
}
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int body(Shared* sh, int me) {
    g_rank = me;
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t n = 1 << 20, off_elems = 4096;  // an interior pointer: base + 32 KiB
    double* mine = nullptr;
    CK(hipMalloc(&mine, (n + off_elems) * sizeof(double)));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mine + off_elems, n, me == 0 ? 1000.0 : 2000.0);
    CK(hipStreamSynchronize(st));
    // ---- 1. memory handles
    void* base = nullptr;
    size_t range = 0;
    CK(hipMemGetAddressRange((hipDeviceptr_t*)&base, &range, (hipDeviceptr_t)(mine + off_elems)));
    sh->offset[me] = (size_t)((char*)(mine + off_elems) - (char*)base);
    CK(hipIpcGetMemHandle(&sh->mem[me], base));
    sync_stage(sh, me, 1);
    void* peer_base = nullptr;
    hipError_t eo = hipIpcOpenMemHandle(&peer_base, sh->mem[1 - me], hipIpcMemLazyEnablePeerAccess);
    sh->result[me][0] = (int)eo;
    if (eo != hipSuccess) {
        std::fprintf(stderr, "[rank %d] hipIpcOpenMemHandle -> %s\n", me, hipGetErrorString(eo));
        sync_stage(sh, me, 99);
        return 1;
    }
    const double* peer = (const double*)((char*)peer_base + sh->offset[1 - me]);
    double *landing = nullptr, *acc = nullptr;
    CK(hipMalloc(&landing, n * sizeof(double)));
    CK(hipMalloc(&acc, sizeof(double)));
    CK(hipMemsetAsync(acc, 0, sizeof(double), st));
    CK(hipMemcpyAsync(landing, peer, n * sizeof(double), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, landing, n, acc);
    double got = 0;
    CK(hipMemcpyAsync(&got, acc, sizeof(double), hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    const double want = (double)n * (me == 0 ? 2000.0 : 1000.0) + (double)n * (double)(n - 1) / 2.0;
    sh->result[me][1] = (got == want);
    CK(hipMemsetAsync(acc, 0, sizeof(double), st));
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, peer, n, acc);  // a kernel reading the peer's memory directly
    CK(hipMemcpyAsync(&got, acc, sizeof(double), hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    sh->result[me][2] = (got == want);
    {  // copy rate peer -> local through the mapping (same device here)
        const int reps = 20;
        CK(hipMemcpyAsync(landing, peer, n * sizeof(double), hipMemcpyDeviceToDevice, st));
        CK(hipStreamSynchronize(st));
        const double t0 = now_s();
        for (int i = 0; i < reps; i++) CK(hipMemcpyAsync(landing, peer, n * sizeof(double), hipMemcpyDeviceToDevice, st));
        CK(hipStreamSynchronize(st));
        sh->timing[me][0] = (double)reps * n * 8 / (now_s() - t0) / 1e9;
    }
    sync_stage(sh, me, 2);

    // ---- 2. flags in registered shared memory, stream memory operations
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
    sh->result[me][3] = can_wait;
    hipError_t er = hipHostRegister((void*)sh->flags, sizeof(sh->flags), hipHostRegisterMapped);
    sh->result[me][4] = (int)er;
    unsigned long long* dflags = nullptr;
    if (er == hipSuccess) er = hipHostGetDevicePointer((void**)&dflags, (void*)sh->flags, 0);
    sh->result[me][5] = (int)er;
    if (er != hipSuccess) {
        std::fprintf(stderr, "[rank %d] hipHostRegister / GetDevicePointer of shm -> %s\n", me, hipGetErrorString(er));
        sync_stage(sh, me, 99);
        return 1;
    }
    // ping-pong: rank 0 writes flag[0] = i, rank 1 waits for it and writes flag[1] = i, rank 0 waits for that
    const int rounds = 200;
    for (int mech = 0; mech < 2; mech++) {
        unsigned long long* mine_f = dflags + 16 * (2 * mech + me);
        unsigned long long* peer_f = dflags + 16 * (2 * mech + 1 - me);
        sync_stage(sh, me, 10 + 2 * mech);
        hipError_t ew = hipSuccess;
        const double t0 = now_s();
        for (int i = 1; i <= rounds && ew == hipSuccess; i++) {
            if (mech == 0) {
                if (me == 0) {
                    ew = hipStreamWriteValue64(st, mine_f, (uint64_t)i, 0);
                    if (ew == hipSuccess) ew = hipStreamWaitValue64(st, peer_f, (uint64_t)i, hipStreamWaitValueGte, ~0ull);
                } else {
                    ew = hipStreamWaitValue64(st, peer_f, (uint64_t)i, hipStreamWaitValueGte, ~0ull);
                    if (ew == hipSuccess) ew = hipStreamWriteValue64(st, mine_f, (uint64_t)i, 0);
                }
            } else {
                if (me == 0) {
                    hipLaunchKernelGGL(flag_write_kernel, dim3(1), dim3(1), 0, st, mine_f, (unsigned long long)i);
                    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(1), 0, st, peer_f, (unsigned long long)i);
                } else {
                    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(1), 0, st, peer_f, (unsigned long long)i);
                    hipLaunchKernelGGL(flag_write_kernel, dim3(1), dim3(1), 0, st, mine_f, (unsigned long long)i);
                }
                ew = hipGetLastError();
            }
        }
        if (ew == hipSuccess) ew = hipStreamSynchronize(st);
        sh->result[me][6 + mech] = (int)ew;
        sh->timing[me][1 + mech] = (now_s() - t0) / rounds * 1e6;
        if (ew != hipSuccess) std::fprintf(stderr, "[rank %d] flag mechanism %d -> %s\n", me, mech, hipGetErrorString(ew));
        sync_stage(sh, me, 11 + 2 * mech);
    }
    // ---- 3. a flag ORDERS data: rank 0 refills its block then raises the flag; rank 1 waits for the flag on its stream, then copies
    for (int mech = 0; mech < 2; mech++) {
        unsigned long long* f = dflags + 16 * (4 + mech);
        sync_stage(sh, me, 20 + 2 * mech);
        int ok = 1;
        for (int i = 1; i <= 20; i++) {
            if (me == 0) {
                hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mine + off_elems, n, 1000.0 * i);
                if (mech == 0) CK(hipStreamWriteValue64(st, f, (uint64_t)i, 0));
                else hipLaunchKernelGGL(flag_write_kernel, dim3(1), dim3(1), 0, st, f, (unsigned long long)i);
                // wait for the reader's acknowledgement before the next refill
                if (mech == 0) CK(hipStreamWaitValue64(st, f + 8, (uint64_t)i, hipStreamWaitValueGte, ~0ull));
                else hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(1), 0, st, f + 8, (unsigned long long)i);
            } else {
                if (mech == 0) CK(hipStreamWaitValue64(st, f, (uint64_t)i, hipStreamWaitValueGte, ~0ull));
                else hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(1), 0, st, f, (unsigned long long)i);
                CK(hipMemcpyAsync(landing, peer, n * sizeof(double), hipMemcpyDeviceToDevice, st));
                if (mech == 0) CK(hipStreamWriteValue64(st, f + 8, (uint64_t)i, 0));
                else hipLaunchKernelGGL(flag_write_kernel, dim3(1), dim3(1), 0, st, f + 8, (unsigned long long)i);
                CK(hipMemsetAsync(acc, 0, sizeof(double), st));
                hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, landing, n, acc);
                CK(hipMemcpyAsync(&got, acc, sizeof(double), hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                const double w = (double)n * 1000.0 * i + (double)n * (double)(n - 1) / 2.0;
                if (got != w) ok = 0;
            }
        }
        CK(hipStreamSynchronize(st));
        sh->result[me][8 + mech] = ok;
        sync_stage(sh, me, 21 + 2 * mech);
    }
    // ---- 4. interprocess events
    hipEvent_t ev;
    hipError_t ee = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess);
    if (ee == hipSuccess) ee = hipIpcGetEventHandle(&sh->ev[me], ev);
    sh->result[me][10] = (int)ee;
    sync_stage(sh, me, 30);
    hipEvent_t pev;
    hipError_t eop = hipErrorUnknown;
    if (sh->result[0][10] == 0 && sh->result[1][10] == 0) {
        eop = hipIpcOpenEventHandle(&pev, sh->ev[1 - me]);
        sh->result[me][11] = (int)eop;
        if (eop == hipSuccess) {
            if (me == 0) {
                hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mine + off_elems, n, 7000.0);
                CK(hipEventRecord(ev, st));
            }
            sync_stage(sh, me, 31);
            if (me == 1) {
                hipError_t ew = hipStreamWaitEvent(st, pev, 0);
                sh->result[me][12] = (int)ew;
                CK(hipMemcpyAsync(landing, peer, n * sizeof(double), hipMemcpyDeviceToDevice, st));
                CK(hipMemsetAsync(acc, 0, sizeof(double), st));
                hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, landing, n, acc);
                CK(hipMemcpyAsync(&got, acc, sizeof(double), hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                sh->result[me][13] = (got == (double)n * 7000.0 + (double)n * (double)(n - 1) / 2.0);
            }
        }
    } else {
        sh->result[me][11] = -1;
    }
    sync_stage(sh, me, 40);
    CK(hipIpcCloseMemHandle(peer_base));
    sync_stage(sh, me, 41);
    CK(hipHostUnregister((void*)sh->flags));
    CK(hipFree(mine));
    CK(hipFree(landing));
    CK(hipFree(acc));
    return 0;
}

int main() {
    char name[64];
    std::snprintf(name, sizeof name, "/hnh_ipc_probe_%d", (int)getpid());
    int fd = shm_open(name, O_CREAT | O_RDWR | O_EXCL, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) { std::perror("shm"); return 2; }
    Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    shm_unlink(name);
    std::memset((void*)sh, 0, sizeof(Shared));
    for (int r = 0; r < 2; r++)
        for (int k = 0; k < 16; k++) sh->result[r][k] = -99;
    pid_t child = fork();  // before any HIP call
    if (child == 0) {
        alarm(120);
        _exit(body(sh, 1));
    }
    alarm(120);
    const int rc0 = body(sh, 0);
    int status = 0;
    waitpid(child, &status, 0);
    const char* names[] = {"hipIpcOpenMemHandle status", "memcpy from the peer's block correct", "kernel read of the peer's block correct",
                           "hipDeviceAttributeCanUseStreamWaitValue", "hipHostRegister(shm) status", "hipHostGetDevicePointer status",
                           "ping-pong, stream memory ops: status", "ping-pong, flag kernels: status", "flag orders data, stream memory ops",
                           "flag orders data, flag kernels", "interprocess event create+handle status", "hipIpcOpenEventHandle status",
                           "hipStreamWaitEvent(opened) status", "event orders data"};
    for (int k = 0; k < 14; k++) std::printf("%-48s rank0 %4d   rank1 %4d\n", names[k], sh->result[0][k], sh->result[1][k]);
    std::printf("copy through the mapping: %.0f / %.0f GB/s;  round trip: stream memory ops %.1f us, flag kernels %.1f us\n", sh->timing[0][0],
                sh->timing[1][0], sh->timing[0][1], sh->timing[0][2]);
    std::printf("exit codes: rank0 %d, rank1 %d\n", rc0, WIFEXITED(status) ? WEXITSTATUS(status) : -1);
    return rc0 || !WIFEXITED(status) || WEXITSTATUS(status);
}
