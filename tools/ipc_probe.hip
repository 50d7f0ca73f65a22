// ipc_probe.hip -- which cross-process mechanisms does this box give two ranks that share records without RCCL?
//   (1) hipExtMallocWithFlags(uncached) + hipIpcGetMemHandle / hipIpcOpenMemHandle between two processes;
//   (2) a kernel of process B storing (system scope) into A's buffer while a kernel of A polls it: ping-pong latency;
//   (3) hipStreamWriteValue64 by B / hipStreamWaitValue64 by A on the same memory (stream-level hand-over).
// Two processes are forked BEFORE any HIP call; they talk over a socketpair.  Every spin is bounded (wall_clock64).
//   hipcc --offload-arch=gfx950 -O3 -o ipc_probe ipc_probe.hip && ./ipc_probe
#include <hip/hip_runtime.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] %s: %s\n", g_rank, #x, hipGetErrorString(e_)); fflush(stdout); _exit(2); } } while (0)
static int g_rank = 0;

typedef unsigned long long u64;
__device__ inline u64 ld_sys(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void st_sys(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// rounds of: (me == 0) write mine[i], wait theirs[i]  /  (me == 1) wait theirs[i], write mine[i]
__global__ void k_pingpong(u64* mine, const u64* theirs, int me, int rounds, u64 base, u64 budget_ticks, u64* out) {
    if (threadIdx.x != 0) return;
    const u64 t0 = wall_clock64();
    int done = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (me == 0) st_sys(mine, base + i);
        bool ok = false;
        while (wall_clock64() - t0 < budget_ticks) { if (ld_sys(theirs) >= base + i) { ok = true; break; } }
        if (!ok) break;
        if (me == 1) st_sys(mine, base + i);
        done = i;
    }
    out[0] = done; out[1] = wall_clock64() - t0;
}
__global__ void k_store(u64* p, u64 v) { if (threadIdx.x == 0) st_sys(p, v); }
__global__ void k_read(const u64* p, u64* out) { if (threadIdx.x == 0) out[0] = ld_sys(p); }

static void xsend(int fd, const void* p, size_t n) { if (write(fd, p, n) != (ssize_t)n) { perror("write"); _exit(3); } }
static void xrecv(int fd, void* p, size_t n) { size_t got = 0; while (got < n) { ssize_t r = read(fd, (char*)p + got, n - got); if (r <= 0) { perror("read"); _exit(3); } got += r; } }
static void barrier(int fd) { char c = 'x'; xsend(fd, &c, 1); xrecv(fd, &c, 1); }

int main(int argc, char** argv) {
    const int flags_kind = argc > 1 ? atoi(argv[1]) : 3;       // 3 = uncached, 1 = fine-grained, 0 = plain hipMalloc
    int sv[2];
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { perror("socketpair"); return 1; }
    const pid_t pid = fork();
    g_rank = pid == 0 ? 1 : 0;
    const int fd = sv[g_rank];
    setvbuf(stdout, nullptr, _IOLBF, 0);

    CK(hipSetDevice(0));
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    if (g_rank == 0) printf("alloc kind %d; hipDeviceAttributeCanUseStreamWaitValue = %d\n", flags_kind, can);
    u64* mine = nullptr;
    if (flags_kind == 0) CK(hipMalloc((void**)&mine, 4096));
    else CK(hipExtMallocWithFlags((void**)&mine, 4096, flags_kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
    CK(hipMemset(mine, 0, 4096));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t hm, ht;
    CK(hipIpcGetMemHandle(&hm, mine));
    xsend(fd, &hm, sizeof hm); xrecv(fd, &ht, sizeof ht);
    u64* theirs = nullptr;
    CK(hipIpcOpenMemHandle((void**)&theirs, ht, hipIpcMemLazyEnablePeerAccess));
    printf("[%d] ipc open ok: mine %p theirs %p\n", g_rank, (void*)mine, (void*)theirs);
    u64* out = nullptr;
    CK(hipHostMalloc((void**)&out, 64));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    barrier(fd);

    // (2a) one remote store, seen by a later kernel of the owner
    if (g_rank == 1) { hipLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, st, theirs + 8, 777ull); CK(hipStreamSynchronize(st)); }
    barrier(fd);
    if (g_rank == 0) { hipLaunchKernelGGL(k_read, dim3(1), dim3(64), 0, st, mine + 8, out); CK(hipStreamSynchronize(st)); printf("[0] remote store then read: %llu (expect 777)\n", out[0]); }
    barrier(fd);

    // (2b) ping-pong between two running kernels: rank r writes word r of the OTHER rank's buffer, polls word (1-r) of its own
    {
        const int rounds = 2000;
        const u64 budget = 300000000ull;                       // 3 s of the 100 MHz wall clock
        barrier(fd);
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(64), 0, st, theirs + g_rank, mine + (1 - g_rank), g_rank, rounds, 1000ull, budget, out);
        CK(hipStreamSynchronize(st));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("[%d] ping-pong (polling kernels, peer-written local memory): %llu/%d rounds, %.3f us per round trip (device clock), host %.1f ms\n",
               g_rank, out[0], rounds, out[0] ? out[1] / 100.0 / out[0] : 0.0, ms);
    }
    barrier(fd);

    // (3) stream-level: B writes a value with hipStreamWriteValue64 into A's buffer; A's stream waits for it, then runs a kernel
    if (can) {
        if (g_rank == 0) {
            hipError_t e = hipStreamWaitValue64(st, mine + 16, 5ull, hipStreamWaitValueGte, ~0ull);
            printf("[0] hipStreamWaitValue64 on own ipc-exported memory: %s\n", hipGetErrorString(e));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_read, dim3(1), dim3(64), 0, st, mine + 16, out);
                barrier(fd);                                   // now let B write
                const auto t0 = std::chrono::steady_clock::now();
                hipError_t q = hipErrorNotReady;
                while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 5.0) { q = hipStreamQuery(st); if (q != hipErrorNotReady) break; }
                printf("[0] after the remote write: stream %s, kernel read %llu (expect 5)\n", hipGetErrorString(q), out[0]);
                if (q == hipErrorNotReady) { printf("[0] wait never released: releasing it locally\n"); hipLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, nullptr, mine + 16, 5ull); hipDeviceSynchronize(); }
            } else barrier(fd);
            (void)hipGetLastError();
        } else {
            barrier(fd);
            hipError_t e = hipStreamWriteValue64(st, theirs + 16, 5ull, 0);
            printf("[1] hipStreamWriteValue64 into the peer's memory: %s\n", hipGetErrorString(e));
            if (e != hipSuccess) { (void)hipGetLastError(); hipLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, st, theirs + 16, 5ull); }
            CK(hipStreamSynchronize(st));
        }
        barrier(fd);
        // latency of write-value -> wait-value -> kernel, 200 rounds, A side timed
        const int rounds = 200;
        if (g_rank == 0) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 1; i <= rounds; ++i) {
                if (hipStreamWriteValue64(st, theirs + 24, 100ull + i, 0) != hipSuccess) { (void)hipGetLastError(); break; }
                if (hipStreamWaitValue64(st, mine + 24, 100ull + i, hipStreamWaitValueGte, ~0ull) != hipSuccess) { (void)hipGetLastError(); break; }
            }
            hipError_t q = hipErrorNotReady;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 5.0) { q = hipStreamQuery(st); if (q != hipErrorNotReady) break; }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("[0] stream-value ping-pong: %s, %.2f us per round trip\n", hipGetErrorString(q), us / rounds);
            if (q == hipErrorNotReady) { hipLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, nullptr, mine + 24, 100000ull); hipDeviceSynchronize(); }
        } else {
            for (int i = 1; i <= rounds; ++i) {
                if (hipStreamWaitValue64(st, mine + 24, 100ull + i, hipStreamWaitValueGte, ~0ull) != hipSuccess) { (void)hipGetLastError(); break; }
                if (hipStreamWriteValue64(st, theirs + 24, 100ull + i, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            }
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t q = hipErrorNotReady;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 6.0) { q = hipStreamQuery(st); if (q != hipErrorNotReady) break; }
            if (q == hipErrorNotReady) { hipLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, nullptr, mine + 24, 100000ull); hipDeviceSynchronize(); }
        }
    }
    barrier(fd);
    CK(hipIpcCloseMemHandle(theirs));
    barrier(fd);
    CK(hipFree(mine));
    if (g_rank == 0) { int status = 0; waitpid(pid, &status, 0); printf("done (child status %d)\n", status); }
    return 0;
}
