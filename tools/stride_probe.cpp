// Microbenchmark for the x pass of the FFT: every workgroup reads (and rewrites) 1024
// segments of 128 bytes that lie `plane` bytes apart — the access pattern of one x-pass
// tile — for two plane strides: N*pad*8 (a large power of two times an odd number) and
// (N+1)*pad*8.  Tells whether the stride (channel/bank aliasing) explains why an x pass
// takes 4.05 ms where a y pass takes 3.5 ms.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void k(double2 *base, long plane16, long row16, int nkb, long ntiles) {
    // tile t: j = t / nkb, kb = t % nkb; element m (0..1023) at base + m*plane16 + j*row16 + kb*8 + w
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        long j = t / nkb, kb = t - j * nkb;
        double2 *p = base + j * row16 + kb * 8 + (threadIdx.x & 7);
        int ml = threadIdx.x >> 3;  // 0..63
        double2 v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = p[(long)(ml + 64 * r) * plane16];
#pragma unroll
        for (int r = 0; r < 16; r++) { v[r].x += 1.0; p[(long)(ml + 64 * r) * plane16] = v[r]; }
    }
}
int main() {
    const long N = 1024, pad = 1040, cp = pad / 2;
    for (int variant = 0; variant < 3; variant++) {
        long ny = N + (variant == 1 ? 1 : (variant == 2 ? 8 : 0));
        long plane16 = ny * cp, row16 = cp;
        size_t bytes = (size_t)N * ny * pad * 8;
        double2 *d;
        if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(d, 0, bytes);
        int nkb = 65;
        long ntiles = N * nkb;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, plane16, row16, nkb, ntiles);
        hipEventRecord(e0);
        for (int it = 0; it < 5; it++) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, plane16, row16, nkb, ntiles);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        double gb = 2.0 * N * N * 520 * 16 / 1e9;
        printf("rows per plane %ld (plane stride %ld B): %.3f ms  %.1f GB/s\n", ny, plane16 * 16, ms, gb / ms * 1e3 / 1e0 / 1e3 * 1e3 / 1e3);
        hipFree(d);
    }
    return 0;
}
