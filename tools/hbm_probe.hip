// hbm_probe.hip -- what does this memory system give (a) a properly unrolled 16 B/lane flat copy / multi-stream sweep far
// beyond the 256 MiB Infinity Cache, and (b) the CG kernels' access shape: independent waves marching DOWN a 1 KiB-wide
// column strip of a row-major array with a 131-262 KB row pitch (N = 16384 / 32768)?  Round 1 measured a 1-load-in-flight
// copy (4.9 TB/s) and called it the ceiling; the guide's figure is 6.0-6.3 TB/s.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_probe hbm_probe.hip && ./hbm_probe [N]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Ptrs { const v2* in[4]; v2* out[2]; };

// ---- flat sweeps: R read streams, W write streams, U groups of loads in flight per lane -------------------------
template <int R, int W, int U, bool NT>
__global__ __launch_bounds__(256) void flat(long long n, Ptrs P) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += U * stride) {
        v2 v[U][R];
#pragma unroll
        for (int k = 0; k < U; ++k) { const long long i = i0 + k * stride;
#pragma unroll
            for (int s = 0; s < R; ++s) if (i < n) v[k][s] = NT ? __builtin_nontemporal_load(&P.in[s][i]) : P.in[s][i]; }
#pragma unroll
        for (int k = 0; k < U; ++k) { const long long i = i0 + k * stride;
            v2 t = v[k][0];
#pragma unroll
            for (int s = 1; s < R; ++s) t += v[k][s];
#pragma unroll
            for (int s = 0; s < W; ++s) if (i < n) { if (NT) __builtin_nontemporal_store(t, &P.out[s][i]); else P.out[s][i] = t; }
            if (W == 0 && t.x == 1.2345e300) P.out[0][0] = t;
        }
    }
}
// each workgroup owns a contiguous segment (block-contiguous instead of grid-strided)
template <int R, int W, int U>
__global__ __launch_bounds__(256) void flat_seg(long long n, Ptrs P) {
    const long long per = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const long long b = (long long)blockIdx.x * per, e = std::min(n, b + per);
    for (long long i0 = b + threadIdx.x; i0 < e; i0 += U * 256) {
        v2 v[U][R];
#pragma unroll
        for (int k = 0; k < U; ++k) { const long long i = i0 + k * 256;
#pragma unroll
            for (int s = 0; s < R; ++s) if (i < e) v[k][s] = P.in[s][i]; }
#pragma unroll
        for (int k = 0; k < U; ++k) { const long long i = i0 + k * 256;
            v2 t = v[k][0];
#pragma unroll
            for (int s = 1; s < R; ++s) t += v[k][s];
#pragma unroll
            for (int s = 0; s < W; ++s) if (i < e) P.out[s][i] = t;
            if (W == 0 && t.x == 1.2345e300) P.out[0][0] = t;
        }
    }
}

// ---- the CG kernels' shape: one wave marches `ty` rows of a (64*16*VW)-byte strip ------------------------------
// items enumerated chunk-major (strip fastest), one item per wave (grid = nitems / 4 workgroups), D rows in flight.
// SYNC: the 4 waves of a workgroup hit a barrier every D rows (does lock-step marching matter?)
template <int R, int W, int VW, int D, bool SYNC>
__global__ __launch_bounds__(256) void strips(Ptrs P, long long pitch /*v2 per row*/, int ns, int ty, int rows, int nitems) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int item = blockIdx.x * 4 + wave; item < nitems; item += gridDim.x * 4) {
        const int chunk = item / ns, strip = item - chunk * ns;
        const int ya = chunk * ty, yb = std::min(rows, ya + ty);
        long long off = (long long)ya * pitch + (long long)strip * (64 * VW) + lane;
        for (int y = ya; y < yb; y += D) {
            v2 v[D][VW][R];
#pragma unroll
            for (int k = 0; k < D; ++k)
#pragma unroll
                for (int w = 0; w < VW; ++w)
#pragma unroll
                    for (int s = 0; s < R; ++s) if (y + k < yb) v[k][w][s] = P.in[s][off + k * pitch + w * 64];
#pragma unroll
            for (int k = 0; k < D; ++k)
#pragma unroll
                for (int w = 0; w < VW; ++w) {
                    v2 t = v[k][w][0];
#pragma unroll
                    for (int s = 1; s < R; ++s) t += v[k][w][s];
#pragma unroll
                    for (int s = 0; s < W; ++s) if (y + k < yb) P.out[s][off + k * pitch + w * 64] = t;
                    if (W == 0 && t.x == 1.2345e300) P.out[0][0] = t;
                }
            off += D * pitch;
            if (SYNC) __syncthreads();
        }
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16384;
    const long long pitch_el = ((N + 1 + 31) / 32) * 32;          // doubles per row, as in the library (Pu)
    const int rows = 3 * N / 4;                                    // ~ the L-shape's unknown count as a rectangle
    const long long n_el = pitch_el * rows, n = n_el / 2, pitch = pitch_el / 2;
    printf("N=%d pitch=%lld B rows=%d bytes/stream=%.2f GB\n", N, pitch_el * 8, rows, n_el * 8 / 1e9);
    double* v[6];
    for (auto& p : v) { CK(hipMalloc(&p, n_el * 8 + (1 << 20))); CK(hipMemset(p, 0, n_el * 8)); }
    Ptrs P{};
    for (int s = 0; s < 4; ++s) P.in[s] = (const v2*)v[s];
    P.out[0] = (v2*)v[4]; P.out[1] = (v2*)v[5];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, int R, int W, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
        }
        CK(hipGetLastError());
        std::sort(ts.begin(), ts.end());
        printf("%-58s median %8.3f ms  min %8.3f ms  -> %7.1f GB/s\n", name, ts[2], ts[0], (R + W) * (double)n_el * 8 / (ts[2] * 1e-3) / 1e9);
        fflush(stdout);
    };
    char nm[128];
#define FLAT(R, W, U, NT, G) do { snprintf(nm, 128, "flat %dR%dW U=%d nt=%d grid=%d", R, W, U, NT, G); \
        timeit(nm, R, W, [&] { hipLaunchKernelGGL((flat<R, W, U, NT>), dim3(G), dim3(256), 0, 0, n, P); }); } while (0)
#define SEG(R, W, U, G) do { snprintf(nm, 128, "flat-seg %dR%dW U=%d grid=%d", R, W, U, G); \
        timeit(nm, R, W, [&] { hipLaunchKernelGGL((flat_seg<R, W, U>), dim3(G), dim3(256), 0, 0, n, P); }); } while (0)
    FLAT(1, 1, 1, false, 2048); FLAT(1, 1, 4, false, 2048); FLAT(1, 1, 8, false, 2048); FLAT(1, 1, 4, false, 1024); FLAT(1, 1, 4, false, 4096);
    FLAT(1, 1, 4, true, 2048); FLAT(1, 1, 8, false, 512);
    FLAT(1, 0, 8, false, 2048); FLAT(2, 0, 4, false, 2048); FLAT(4, 0, 2, false, 2048);
    FLAT(2, 1, 1, false, 2048); FLAT(2, 1, 2, false, 2048); FLAT(2, 1, 4, false, 2048); FLAT(2, 1, 4, false, 512); FLAT(2, 1, 4, false, 1024); FLAT(2, 1, 4, true, 2048);
    FLAT(4, 2, 1, false, 2048); FLAT(4, 2, 2, false, 2048); FLAT(4, 2, 2, false, 512); FLAT(4, 2, 2, true, 2048);
    SEG(1, 1, 4, 2048); SEG(2, 1, 4, 2048); SEG(2, 1, 4, 512); SEG(4, 2, 2, 2048);

    // strips: ns strips per row; item height so that one round of `waves` waves covers the array
#define STRIPS(R, W, VW, D, SYNC, WAVES) do { \
        const int ns = (int)((pitch_el * 8 + 1024 * VW - 1) / (1024 * VW)); \
        int ty = (int)(((long long)ns * rows + WAVES - 1) / WAVES); \
        int nchunks = (rows + ty - 1) / ty; const int nitems = ns * nchunks; \
        snprintf(nm, 128, "strips %dR%dW strip=%dB D=%d sync=%d waves=%d ty=%d items=%d", R, W, 1024 * VW, D, SYNC, WAVES, ty, nitems); \
        timeit(nm, R, W, [&] { hipLaunchKernelGGL((strips<R, W, VW, D, SYNC>), dim3((nitems + 3) / 4), dim3(256), 0, 0, P, pitch, ns, ty, rows, nitems); }); } while (0)
    STRIPS(2, 1, 1, 2, false, 2048); STRIPS(2, 1, 1, 4, false, 2048); STRIPS(2, 1, 1, 8, false, 2048);
    STRIPS(2, 1, 1, 4, false, 4096); STRIPS(2, 1, 1, 4, false, 8192); STRIPS(2, 1, 1, 2, false, 8192);
    STRIPS(2, 1, 1, 4, true, 2048); STRIPS(2, 1, 1, 4, true, 4096);
    STRIPS(2, 1, 2, 2, false, 2048); STRIPS(2, 1, 2, 4, false, 2048); STRIPS(2, 1, 2, 2, false, 1024); STRIPS(2, 1, 2, 4, false, 1024);
    STRIPS(2, 1, 4, 2, false, 1024); STRIPS(2, 1, 4, 1, false, 2048); STRIPS(2, 1, 4, 2, false, 512);
    STRIPS(4, 2, 1, 2, false, 2048); STRIPS(4, 2, 1, 4, false, 2048); STRIPS(4, 2, 2, 2, false, 2048); STRIPS(4, 2, 2, 2, false, 1024);
    STRIPS(4, 2, 1, 2, true, 2048);
    // many short items, chunk-major: concurrently running waves stay inside a band of ~ (waves / ns) * ty rows
    {
        const int ns = (int)((pitch_el * 8 + 1023) / 1024);
        for (int ty : {16, 32, 64, 128}) {
            const int nchunks = (rows + ty - 1) / ty, nitems = ns * nchunks;
            snprintf(nm, 128, "strips 2R1W strip=1024B D=4 rounds ty=%d items=%d grid=512", ty, nitems);
            timeit(nm, 2, 1, [&] { hipLaunchKernelGGL((strips<2, 1, 1, 4, false>), dim3(512), dim3(256), 0, 0, P, pitch, ns, ty, rows, nitems); });
            snprintf(nm, 128, "strips 2R1W strip=1024B D=4 rounds ty=%d items=%d grid=all", ty, nitems);
            timeit(nm, 2, 1, [&] { hipLaunchKernelGGL((strips<2, 1, 1, 4, false>), dim3((nitems + 3) / 4), dim3(256), 0, 0, P, pitch, ns, ty, rows, nitems); });
        }
    }
    return 0;
}
