// copy_ceiling.hip -- what does this box's memory system give the simplest streaming kernels, and where does the guide's
// "6.29 TB/s float4 copy" (MI355X_MICROARCH.md:36) sit?  One float4 per thread (a grid of millions of workgroups, the usual
// micro-benchmark shape) and a few per thread, plain and nontemporal, read-only / write-only / copy, for stream sizes from inside
// the 256 MiB Infinity Cache to far beyond it; hipMemcpyDtoD for comparison.  Rates count bytes read + bytes written.
//   hipcc --offload-arch=gfx950 -O3 -o copy_ceiling copy_ceiling.hip && ./copy_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// E elements per thread, block-contiguous: thread t of block b handles b*256*E + k*256 + t
template <int E, bool NT, int MODE /*0 copy, 1 read, 2 write*/>
__global__ __launch_bounds__(256) void k_shot(const f4* __restrict__ in, f4* __restrict__ out, long long n) {
    const long long base = (long long)blockIdx.x * (256 * E) + threadIdx.x;
    f4 v[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const long long i = base + k * 256;
        if (MODE != 2) { if (i < n) v[k] = NT ? __builtin_nontemporal_load(in + i) : in[i]; }
        else v[k] = f4{1.f, 2.f, 3.f, 4.f};
    }
    if (MODE == 1) {
        f4 t = v[0];
#pragma unroll
        for (int k = 1; k < E; ++k) t += v[k];
        if (t.x == 1.2345e30f) out[0] = t;
        return;
    }
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const long long i = base + k * 256;
        if (i < n) { if (NT) __builtin_nontemporal_store(v[k], out + i); else out[i] = v[k]; }
    }
}
// the CG launches' traffic mixes as flat one-shot sweeps: R read streams summed into W write streams (stencil launch 2R1W, 6-word
// update 4R2W), E elements per thread
template <int R, int W, int E>
__global__ __launch_bounds__(256) void k_mix(const f4* __restrict__ a, f4* __restrict__ b, long long n) {
    const long long base = (long long)blockIdx.x * (256 * E) + threadIdx.x;
    f4 v[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const long long i = base + k * 256;
        v[k] = f4{0.f, 0.f, 0.f, 0.f};
        if (i < n) {
#pragma unroll
            for (int s = 0; s < R; ++s) v[k] += a[(long long)s * n + i];
        }
    }
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const long long i = base + k * 256;
        if (i < n) {
#pragma unroll
            for (int s = 0; s < W; ++s) b[(long long)s * n + i] = v[k];
        }
    }
}
// persistent grid-stride copy, U loads in flight per lane
template <int U>
__global__ __launch_bounds__(256) void k_stride(const f4* __restrict__ in, f4* __restrict__ out, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += U * stride) {
        f4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) if (i0 + k * stride < n) v[k] = in[i0 + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) if (i0 + k * stride < n) out[i0 + k * stride] = v[k];
    }
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("%s  CUs %d  core clock %d MHz  memory clock %d MHz  bus %d bit\n", pr.name, pr.multiProcessorCount, pr.clockRate / 1000, pr.memoryClockRate / 1000, pr.memoryBusWidth);
    const long long maxb = 4LL << 30;
    f4 *a, *b;
    CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb));
    CK(hipMemset(a, 1, maxb)); CK(hipMemset(b, 0, maxb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double bytes, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int rep = 0; rep < 9; ++rep) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms); }
        CK(hipGetLastError());
        std::sort(ts.begin(), ts.end());
        printf("  %-46s median %8.3f ms  best %8.3f ms  -> %7.1f GB/s (best %7.1f)\n", name, ts[4], ts[0], bytes / (ts[4] * 1e-3) / 1e9, bytes / (ts[0] * 1e-3) / 1e9);
        fflush(stdout);
    };
    for (long long mb : {64LL, 256LL, 1024LL, 4096LL}) {
        const long long bytes = mb << 20, n = bytes / 16;
        printf("stream size %lld MiB (copy touches 2 x that)\n", mb);
#define SHOT(E, NT, MODE, label) do { const int g = (int)((n + 256 * E - 1) / (256 * E)); \
        timeit(label, (MODE == 0 ? 2.0 : 1.0) * bytes, [&] { hipLaunchKernelGGL((k_shot<E, NT, MODE>), dim3(g), dim3(256), 0, 0, a, b, n); }); } while (0)
        SHOT(1, false, 0, "copy  1 float4/thread"); SHOT(2, false, 0, "copy  2 float4/thread"); SHOT(4, false, 0, "copy  4 float4/thread"); SHOT(8, false, 0, "copy  8 float4/thread");
        SHOT(1, true, 0, "copy  1 float4/thread nontemporal"); SHOT(4, true, 0, "copy  4 float4/thread nontemporal");
        SHOT(1, false, 1, "read  1 float4/thread"); SHOT(4, false, 1, "read  4 float4/thread"); SHOT(4, true, 1, "read  4 float4/thread nontemporal");
        SHOT(1, false, 2, "write 1 float4/thread"); SHOT(4, false, 2, "write 4 float4/thread"); SHOT(4, true, 2, "write 4 float4/thread nontemporal");
        timeit("copy  grid-stride 2048 blocks, 1 in flight", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_stride<1>), dim3(2048), dim3(256), 0, 0, a, b, n); });
        timeit("copy  grid-stride 2048 blocks, 4 in flight", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_stride<4>), dim3(2048), dim3(256), 0, 0, a, b, n); });
        timeit("hipMemcpyDtoDAsync", 2.0 * bytes, [&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
        if (mb <= 1024) {                                          // R streams of this size live in a, W in b (4 GiB each)
            const long long nn = n;
#define MIX(R, W, E, label) do { const int g = (int)((nn + 256 * E - 1) / (256 * E)); \
            timeit(label, (double)(R + W) * bytes, [&] { hipLaunchKernelGGL((k_mix<R, W, E>), dim3(g), dim3(256), 0, 0, a, b, nn); }); } while (0)
            MIX(2, 1, 1, "2R1W (stencil launch's mix) 1 float4/thread"); MIX(2, 1, 2, "2R1W 2 float4/thread");
            MIX(4, 2, 1, "4R2W (6-word update's mix) 1 float4/thread"); MIX(4, 2, 2, "4R2W 2 float4/thread");
            MIX(2, 1, 4, "2R1W 4 float4/thread"); MIX(3, 1, 1, "3R1W 1 float4/thread");
        }
    }
    return 0;
}
