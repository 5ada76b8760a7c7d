// Out-of-place against in-place: does a pass that reads buffer A and writes buffer B run faster
// than one that overwrites A?  (The FFT passes of cg_fft.hip are in place.)  Also the strided
// access of the y / x passes: rows of 16 B * 8 = 128 B gathered with a stride of one plane row.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/copy_probe tools/copy_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, int NT>
__global__ __launch_bounds__(256) void k_pass(const d2 *src, d2 *dst) {
    size_t base = (size_t)blockIdx.x * 256 * U;
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++)
        v[u] = NT & 1 ? __builtin_nontemporal_load(&src[base + u * 256 + threadIdx.x]) : src[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; u++) {
        v[u].x += 1.0;
        if (NT & 2) __builtin_nontemporal_store(v[u], &dst[base + u * 256 + threadIdx.x]);
        else dst[base + u * 256 + threadIdx.x] = v[u];
    }
}
// persistent form: a fixed grid, each workgroup walks chunks
template <int U>
__global__ __launch_bounds__(256) void k_pass_persist(const d2 *src, d2 *dst, size_t nchunks) {
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        size_t base = c * 256 * U;
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = src[base + u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < U; u++) { v[u].x += 1.0; dst[base + u * 256 + threadIdx.x] = v[u]; }
    }
}
template <int U, int NT>
static double run(const d2 *s, d2 *d, size_t bytes, hipEvent_t e0, hipEvent_t e1) {
    unsigned g = (unsigned)(bytes / 16 / (256 * U));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_pass<U, NT>), dim3(g), dim3(256), 0, 0, s, d);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 6; i++) hipLaunchKernelGGL((k_pass<U, NT>), dim3(g), dim3(256), 0, 0, s, d);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 2.0 * bytes * 6 / ms / 1e9;
}
int main() {
    const size_t total = (size_t)8704 << 20;
    d2 *a, *b; CK(hipMalloc(&a, total)); CK(hipMalloc(&b, total)); CK(hipMemset(a, 0, total)); CK(hipMemset(b, 0, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("U=4  in-place      plain %.2f  nt-load %.2f  nt-store %.2f  nt-both %.2f TB/s\n",
           run<4, 0>(a, a, total, e0, e1), run<4, 1>(a, a, total, e0, e1), run<4, 2>(a, a, total, e0, e1), run<4, 3>(a, a, total, e0, e1));
    printf("U=4  out-of-place  plain %.2f  nt-load %.2f  nt-store %.2f  nt-both %.2f TB/s\n",
           run<4, 0>(a, b, total, e0, e1), run<4, 1>(a, b, total, e0, e1), run<4, 2>(a, b, total, e0, e1), run<4, 3>(a, b, total, e0, e1));
    printf("U=8  in-place      plain %.2f  nt-both %.2f   out-of-place plain %.2f  nt-both %.2f TB/s\n",
           run<8, 0>(a, a, total, e0, e1), run<8, 3>(a, a, total, e0, e1), run<8, 0>(a, b, total, e0, e1), run<8, 3>(a, b, total, e0, e1));
    printf("U=16 in-place      plain %.2f  nt-both %.2f   out-of-place plain %.2f  nt-both %.2f TB/s\n",
           run<16, 0>(a, a, total, e0, e1), run<16, 3>(a, a, total, e0, e1), run<16, 0>(a, b, total, e0, e1), run<16, 3>(a, b, total, e0, e1));
    printf("U=1  in-place      plain %.2f  out-of-place plain %.2f  nt-both %.2f TB/s\n",
           run<1, 0>(a, a, total, e0, e1), run<1, 0>(a, b, total, e0, e1), run<1, 3>(a, b, total, e0, e1));
    for (unsigned g : {256u * 4, 256u * 8, 256u * 16, 256u * 32}) {
        size_t nch = total / 16 / (256 * 4);
        for (int pass = 0; pass < 2; pass++) {
            d2 *dst = pass ? b : a;
            hipLaunchKernelGGL((k_pass_persist<4>), dim3(g), dim3(256), 0, 0, a, dst, nch);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 6; i++) hipLaunchKernelGGL((k_pass_persist<4>), dim3(g), dim3(256), 0, 0, a, dst, nch);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("persistent grid %5u %s: %.2f TB/s\n", g, pass ? "out-of-place" : "in-place", 2.0 * total * 6 / ms / 1e9);
        }
    }
    // chunked two-pass schedules (the z / y FFT passes): (a) both in place on the chunk, (b) pass 1
    // writes a chunk-sized scratch that stays cache-resident, pass 2 reads it and writes the chunk
    {
        d2 *scr = b;
        for (size_t mb : {64, 96, 128, 192, 256}) {
            size_t chunk = mb << 20, nchunks = total / chunk, n = chunk / 16;
            unsigned g = (unsigned)(n / (256 * 4));
            float ms;
            CK(hipEventRecord(e0));
            for (int rep = 0; rep < 3; rep++)
                for (size_t c = 0; c < nchunks; c++) {
                    d2 *pc = a + c * n;
                    hipLaunchKernelGGL((k_pass<4, 0>), dim3(g), dim3(256), 0, 0, pc, pc);
                    hipLaunchKernelGGL((k_pass<4, 0>), dim3(g), dim3(256), 0, 0, pc, pc);
                }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            float t_in = ms / 3;
            CK(hipEventRecord(e0));
            for (int rep = 0; rep < 3; rep++)
                for (size_t c = 0; c < nchunks; c++) {
                    d2 *pc = a + c * n;
                    hipLaunchKernelGGL((k_pass<4, 0>), dim3(g), dim3(256), 0, 0, pc, scr);
                    hipLaunchKernelGGL((k_pass<4, 0>), dim3(g), dim3(256), 0, 0, scr, pc);
                }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            float t_scr = ms / 3;
            CK(hipEventRecord(e0));
            for (int rep = 0; rep < 3; rep++)
                for (size_t c = 0; c < nchunks; c++) {
                    d2 *pc = a + c * n;
                    hipLaunchKernelGGL((k_pass<4, 1>), dim3(g), dim3(256), 0, 0, pc, scr);
                    hipLaunchKernelGGL((k_pass<4, 2>), dim3(g), dim3(256), 0, 0, scr, pc);
                }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            float t_scr_nt = ms / 3;
            unsigned g1 = (unsigned)(n / 256);
            CK(hipEventRecord(e0));
            for (int rep = 0; rep < 3; rep++)
                for (size_t c = 0; c < nchunks; c++) {
                    d2 *pc = a + c * n;
                    hipLaunchKernelGGL((k_pass<1, 0>), dim3(g1), dim3(256), 0, 0, pc, scr);
                    hipLaunchKernelGGL((k_pass<1, 0>), dim3(g1), dim3(256), 0, 0, scr, pc);
                }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            float t_scr_u1 = ms / 3;
            printf("two passes per chunk of %3zu MB: in place %.2f ms   via scratch %.2f   via scratch nt %.2f   via scratch U=1 %.2f ms per sweep\n",
                   mb, t_in, t_scr, t_scr_nt, t_scr_u1);
        }
    }
    return 0;
}
