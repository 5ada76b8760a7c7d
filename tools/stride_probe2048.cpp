// Microbenchmark for the strided FFT passes at 2048^3: every workgroup reads and rewrites the
// 2048 row segments of one tile — W complex numbers (W*16 bytes) per row, rows `stride` apart
// — for the x pass (stride = one mesh layer, 33.8 MB) and the y pass (stride = one row,
// 16.5 KB), with 4 pencils per tile (64-byte segments, what fits the LDS today) and 8 (full
// 128-byte lines).  Pure access pattern, no transform: what the memory system gives each
// variant.   hipcc --offload-arch=gfx950 -O3 tools/stride_probe2048.cpp -o tools/stride_probe2048
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int W, int NT>
__global__ __launch_bounds__(NT) void k(double2 *base, long es16, long os16, int nkb, long ntiles) {
    constexpr int PER = 2048 * W / NT, MSTEP = NT / W;
    const int wl = threadIdx.x % W, ml = threadIdx.x / W;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        long o = t / nkb, kb = t - o * nkb;
        double2 *p = base + o * os16 + kb * W + wl;
        double2 v[PER];
#pragma unroll
        for (int r = 0; r < PER; r++) v[r] = p[(long)(ml + MSTEP * r) * es16];
#pragma unroll
        for (int r = 0; r < PER; r++) { v[r].x += 1.0; p[(long)(ml + MSTEP * r) * es16] = v[r]; }
    }
}
template <int W, int NT>
static void run(const char *name, double2 *d, long es16, long os16, long nouter) {
    const int nkb = (1025 + W - 1) / W;
    const long ntiles = nouter * nkb;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<W, NT>), dim3(256), dim3(NT), 0, 0, d, es16, os16, nkb, ntiles);
    hipEventRecord(e0);
    for (int it = 0; it < 2; it++)
        hipLaunchKernelGGL((k<W, NT>), dim3(256), dim3(NT), 0, 0, d, es16, os16, nkb, ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 2;
    double gb = 2.0 * nouter * 2048.0 * nkb * W * 16 / 1e9;
    printf("%-44s %8.2f ms  %7.1f GB/s\n", name, ms, gb / ms);
}
int main() {
    const long N = 2048, pad = 2064, cp = pad / 2, ny = N + 1;
    size_t bytes = (size_t)N * ny * pad * 8;
    double2 *d;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, bytes);
    // x pass: element stride = layer, outer = row j
    run<4, 512>("x pass, 4 pencils (64 B), 512 lanes", d, ny * cp, cp, N);
    run<8, 512>("x pass, 8 pencils (128 B), 512 lanes x 32", d, ny * cp, cp, N);
    run<8, 1024>("x pass, 8 pencils (128 B), 1024 lanes x 16", d, ny * cp, cp, N);
    // y pass: element stride = row, outer = layer i
    run<4, 512>("y pass, 4 pencils (64 B), 512 lanes", d, cp, ny * cp, N);
    run<8, 512>("y pass, 8 pencils (128 B), 512 lanes x 32", d, cp, ny * cp, N);
    run<8, 1024>("y pass, 8 pencils (128 B), 1024 lanes x 16", d, cp, ny * cp, N);
    hipFree(d);
    return 0;
}
