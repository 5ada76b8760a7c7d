// Infinity-cache (MALL) probe: what does an in-place pass cost when its data is already in the
// 256 MB memory-side cache, and what is the floor of "two in-place passes per chunk" — the
// schedule of the z / y FFT passes (cg_fft.hip) — as a pure access pattern without any
// transform?  build: hipcc --offload-arch=gfx950 -O3 -o tools/mall_probe tools/mall_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one workgroup per 4 KiB * U
template <int U>
__global__ __launch_bounds__(256) void k_rmw(d2 *__restrict__ buf, size_t n) {
    size_t base = (size_t)blockIdx.x * 256 * U;
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = buf[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u].x += 1.0; buf[base + u * 256 + threadIdx.x] = v[u]; }
}
template <int U>
__global__ __launch_bounds__(256) void k_read(const d2 *__restrict__ src, double *out, size_t n) {
    size_t base = (size_t)blockIdx.x * 256 * U;
    double acc = 0;
#pragma unroll
    for (int u = 0; u < U; u++) { d2 v = src[base + u * 256 + threadIdx.x]; acc += v.x + v.y; }
    if (acc == 1.2345) out[0] = acc;
}
int main() {
    const size_t total = (size_t)8704 << 20;  // ~ the 1024^3 mesh
    d2 *a; double *o; CK(hipMalloc(&a, total)); CK(hipMalloc(&o, 8)); CK(hipMemset(a, 0, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr int U = 4;
    // 1. repeated passes over a window of S bytes (steady state: the window is cache-resident
    //    when it fits)
    for (size_t mb : {32, 64, 128, 192, 224, 256, 288, 320, 512, 2048, 8704}) {
        size_t bytes = mb << 20, n = bytes / 16;
        int reps = (int)(total / bytes) * 2; if (reps < 2) reps = 2;
        unsigned g = (unsigned)(n / (256 * U));
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_rmw<U>), dim3(g), dim3(256), 0, 0, a, n);
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_rmw<U>), dim3(g), dim3(256), 0, 0, a, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double rmw = 2.0 * bytes * reps / ms / 1e9;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_read<U>), dim3(g), dim3(256), 0, 0, a, o, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("window %5zu MB: in-place %.2f TB/s   read-only %.2f TB/s\n", mb, rmw, (double)bytes * reps / ms / 1e9);
    }
    // 2. the chunked two-pass schedule over the whole buffer: pass A then pass B per chunk
    for (size_t mb : {64, 128, 192, 256, 272, 320, 8704}) {
        size_t chunk = mb << 20; if (chunk > total) chunk = total;
        size_t nchunks = total / chunk;
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 3; rep++)
            for (size_t c = 0; c < nchunks; c++) {
                d2 *p = a + c * (chunk / 16); size_t n = chunk / 16; unsigned g = (unsigned)(n / (256 * U));
                hipLaunchKernelGGL((k_rmw<U>), dim3(g), dim3(256), 0, 0, p, n);
                hipLaunchKernelGGL((k_rmw<U>), dim3(g), dim3(256), 0, 0, p, n);
            }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("two in-place passes per chunk of %5zu MB over %zu MB: %.2f ms per sweep (%zu launches)\n", mb,
               nchunks * chunk >> 20, ms / 3, 2 * nchunks);
    }
    // 3. three passes per chunk (z, y and something else riding along)
    for (size_t mb : {128, 192, 256}) {
        size_t chunk = mb << 20; size_t nchunks = total / chunk;
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 3; rep++)
            for (size_t c = 0; c < nchunks; c++) {
                d2 *p = a + c * (chunk / 16); size_t n = chunk / 16; unsigned g = (unsigned)(n / (256 * U));
                for (int k = 0; k < 3; k++) hipLaunchKernelGGL((k_rmw<U>), dim3(g), dim3(256), 0, 0, p, n);
            }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("three in-place passes per chunk of %5zu MB: %.2f ms per sweep\n", mb, ms / 3);
    }
    return 0;
}
