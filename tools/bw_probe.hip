// bw_probe.hip -- on-box ceilings for the update kernel's access pattern (4 read + 2 write fp64 streams of
// 12.7 M elements) and variants of it.  hipcc --offload-arch=gfx950 -O3 -o bw_probe bw_probe.hip && ./bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>   // 0 plain, 1 nt on x (load+store) and ap load, 2 nt everywhere
__global__ __launch_bounds__(256) void upd(long long n, v2* x, v2* r, const v2* p, const v2* q, double alpha, double* part) {
    const long long stride = (long long)gridDim.x * 256;
    double s = 0;
    auto body = [&](long long i) {
        v2 x0, pv, r0, qv;
        if (MODE >= 1) { x0 = __builtin_nontemporal_load(&x[i]); qv = __builtin_nontemporal_load(&q[i]); } else { x0 = x[i]; qv = q[i]; }
        if (MODE == 2) { pv = __builtin_nontemporal_load(&p[i]); r0 = __builtin_nontemporal_load(&r[i]); } else { pv = p[i]; r0 = r[i]; }
        v2 xn = x0 + alpha * pv, rn = r0 - alpha * qv;
        s += rn.x * rn.x + rn.y * rn.y;
        if (MODE >= 1) __builtin_nontemporal_store(xn, &x[i]); else x[i] = xn;
        if (MODE == 2) __builtin_nontemporal_store(rn, &r[i]); else r[i] = rn;
    };
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n; i += 2 * stride) { body(i); body(i + stride); }
    if (i < n) body(i);
    if (s == 12345.678) part[blockIdx.x] = s;
}
template <int U>
__global__ __launch_bounds__(256) void upd_unroll(long long n, v2* x, v2* r, const v2* p, const v2* q, double alpha, double* part) {
    const long long stride = (long long)gridDim.x * 256;
    double s = 0;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += U * stride) {
        v2 x0[U], pv[U], r0[U], qv[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { long long i = i0 + k * stride; if (i < n) { x0[k] = x[i]; pv[k] = p[i]; r0[k] = r[i]; qv[k] = q[i]; } }
#pragma unroll
        for (int k = 0; k < U; ++k) { long long i = i0 + k * stride; if (i < n) { v2 xn = x0[k] + alpha * pv[k], rn = r0[k] - alpha * qv[k]; s += rn.x * rn.x + rn.y * rn.y; x[i] = xn; r[i] = rn; } }
    }
    if (s == 12345.678) part[blockIdx.x] = s;
}
// the 3-stream light update (r -= alpha*q, sum r^2), 4 groups in flight, optional reverse sweep
template <bool REV>
__global__ __launch_bounds__(256) void upd_light(long long n, v2* __restrict__ r, const v2* __restrict__ q, double alpha, double* part) {
    const long long stride = (long long)gridDim.x * 256;
    double s = 0;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        v2 r0[4], qv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { long long j = i + k * stride; if (REV) j = n - 1 - j; r0[k] = r[j]; qv[k] = q[j]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) { long long j = i + k * stride; if (REV) j = n - 1 - j; v2 rn = r0[k] - alpha * qv[k]; s += rn.x * rn.x + rn.y * rn.y; r[j] = rn; }
    }
    for (; i < n; i += stride) { long long j = REV ? n - 1 - i : i; v2 rn = r[j] - alpha * q[j]; s += rn.x * rn.x + rn.y * rn.y; r[j] = rn; }
    if (s == 12345.678) part[blockIdx.x] = s;
}
// each block owns a contiguous segment
__global__ __launch_bounds__(256) void upd_seg(long long n, v2* x, v2* r, const v2* p, const v2* q, double alpha, double* part) {
    const long long per = (n + gridDim.x - 1) / gridDim.x, b = (long long)blockIdx.x * per, e = b + per < n ? b + per : n;
    double s = 0;
    for (long long i = b + threadIdx.x; i < e; i += 512) {
        const long long j = i + 256;
        v2 x0 = x[i], pv = p[i], r0 = r[i], qv = q[i], x1, p1, r1, q1;
        const bool two = j < e;
        if (two) { x1 = x[j]; p1 = p[j]; r1 = r[j]; q1 = q[j]; }
        v2 xn = x0 + alpha * pv, rn = r0 - alpha * qv; s += rn.x * rn.x + rn.y * rn.y; x[i] = xn; r[i] = rn;
        if (two) { v2 xm = x1 + alpha * p1, rm = r1 - alpha * q1; s += rm.x * rm.x + rm.y * rm.y; x[j] = xm; r[j] = rm; }
    }
    if (s == 12345.678) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void rd4(long long n, const v2* a, const v2* b, const v2* c, const v2* d, double* part) {
    const long long stride = (long long)gridDim.x * 256; double s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { v2 t = a[i] + b[i] + c[i] + d[i]; s += t.x + t.y; }
    if (s == 12345.678) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void wr2(long long n, v2* a, v2* b, double v) {
    const long long stride = (long long)gridDim.x * 256; v2 t = {v, v};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { a[i] = t; b[i] = t; }
}
__global__ __launch_bounds__(256) void cp1(long long n, const v2* a, v2* b) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) b[i] = a[i];
}

int main() {
    const long long N = 12709856, n = N / 2;
    double* v[6]; double* part;
    for (auto& p : v) { CK(hipMalloc(&p, N * 8)); CK(hipMemset(p, 0, N * 8)); }
    CK(hipMalloc(&part, 8192 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double bytes, auto launch) {
        for (int w = 0; w < 5; ++w) launch();
        std::vector<float> ts;
        for (int rep = 0; rep < 7; ++rep) {
            hipEventRecord(e0); for (int k = 0; k < 20; ++k) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms / 20);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-34s median %8.2f us  min %8.2f us  -> %7.1f GB/s (median)\n", name, ts[3] * 1e3, ts[0] * 1e3, bytes / (ts[3] * 1e-3) / 1e9);
    };
    v2 *x = (v2*)v[0], *r = (v2*)v[1], *p = (v2*)v[2], *q = (v2*)v[3];
    const double B6 = 6.0 * N * 8;
    for (int g : {256, 512, 1024, 2048}) {
        char nm[64]; snprintf(nm, 64, "update plain grid=%d", g);
        timeit(nm, B6, [&] { hipLaunchKernelGGL(upd<0>, dim3(g), dim3(256), 0, 0, n, x, r, p, q, 1e-3, part); });
    }
    timeit("update nt(x,ap) grid=512", B6, [&] { hipLaunchKernelGGL(upd<1>, dim3(512), dim3(256), 0, 0, n, x, r, p, q, 1e-3, part); });
    timeit("update nt(all) grid=512", B6, [&] { hipLaunchKernelGGL(upd<2>, dim3(512), dim3(256), 0, 0, n, x, r, p, q, 1e-3, part); });
    timeit("update unroll4 grid=512", B6, [&] { hipLaunchKernelGGL(upd_unroll<4>, dim3(512), dim3(256), 0, 0, n, x, r, p, q, 1e-3, part); });
    timeit("update unroll4 grid=256", B6, [&] { hipLaunchKernelGGL(upd_unroll<4>, dim3(256), dim3(256), 0, 0, n, x, r, p, q, 1e-3, part); });
    timeit("update unroll1 grid=2048", B6, [&] { hipLaunchKernelGGL(upd_unroll<1>, dim3(2048), dim3(256), 0, 0, n, x, r, p, q, 1e-3, part); });
    timeit("update unroll1 grid=4096", B6, [&] { hipLaunchKernelGGL(upd_unroll<1>, dim3(4096), dim3(256), 0, 0, n, x, r, p, q, 1e-3, part); });
    timeit("update segmented grid=512", B6, [&] { hipLaunchKernelGGL(upd_seg, dim3(512), dim3(256), 0, 0, n, x, r, p, q, 1e-3, part); });
    timeit("update segmented grid=2048", B6, [&] { hipLaunchKernelGGL(upd_seg, dim3(2048), dim3(256), 0, 0, n, x, r, p, q, 1e-3, part); });
    for (int g : {256, 512, 1024}) {
        char nm[64]; snprintf(nm, 64, "light 2R1W fwd grid=%d", g);
        timeit(nm, 3.0 * N * 8, [&] { hipLaunchKernelGGL(upd_light<false>, dim3(g), dim3(256), 0, 0, n, r, q, 1e-3, part); });
        snprintf(nm, 64, "light 2R1W rev grid=%d", g);
        timeit(nm, 3.0 * N * 8, [&] { hipLaunchKernelGGL(upd_light<true>, dim3(g), dim3(256), 0, 0, n, r, q, 1e-3, part); });
    }
    timeit("read 4 streams grid=1024", 4.0 * N * 8, [&] { hipLaunchKernelGGL(rd4, dim3(1024), dim3(256), 0, 0, n, x, r, p, q, part); });
    timeit("write 2 streams grid=1024", 2.0 * N * 8, [&] { hipLaunchKernelGGL(wr2, dim3(1024), dim3(256), 0, 0, n, x, r, 0.0); });
    timeit("copy 1->1 grid=1024", 2.0 * N * 8, [&] { hipLaunchKernelGGL(cp1, dim3(1024), dim3(256), 0, 0, n, (const v2*)v[4], (v2*)v[5]); });
    timeit("copy 1->1 grid=2048", 2.0 * N * 8, [&] { hipLaunchKernelGGL(cp1, dim3(2048), dim3(256), 0, 0, n, (const v2*)v[4], (v2*)v[5]); });
    // same six streams, bases staggered by odd multiples of 4 KiB + 256 B (do the streams collide on HBM channels?)
    {
        double* w[4]; const long long padB = 1 << 20;
        for (auto& q2 : w) { CK(hipMalloc(&q2, N * 8 + padB)); CK(hipMemset(q2, 0, N * 8 + padB)); }
        for (int stagger : {0, 256, 4352, 69888}) {
            v2* xs = (v2*)((char*)w[0] + 0 * stagger); v2* rs = (v2*)((char*)w[1] + 1 * stagger);
            v2* ps = (v2*)((char*)w[2] + 2 * stagger); v2* qs = (v2*)((char*)w[3] + 3 * stagger);
            char nm[64]; snprintf(nm, 64, "update plain g=512 stagger=%d", stagger);
            timeit(nm, B6, [&] { hipLaunchKernelGGL(upd<0>, dim3(512), dim3(256), 0, 0, n, xs, rs, ps, qs, 1e-3, part); });
        }
    }
    // big copy (well past the 256 MiB Infinity Cache): 2 x 813 MB
    double *A, *Bb; const long long M = 8 * N; CK(hipMalloc(&A, M * 8)); CK(hipMalloc(&Bb, M * 8)); CK(hipMemset(A, 0, M * 8));
    timeit("copy 813MB->813MB grid=2048", 2.0 * M * 8, [&] { hipLaunchKernelGGL(cp1, dim3(2048), dim3(256), 0, 0, M / 2, (const v2*)A, (v2*)Bb); });
    return 0;
}
