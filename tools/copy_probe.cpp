// HBM copy-rate probe: how fast can a hand-written kernel copy 8 GB on this part, and does the
// form of the kernel matter?  (torch's elementwise copy: 4.8 TB/s read+write.)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/copy_probe tools/copy_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const d2 *__restrict__ src, d2 *__restrict__ dst, size_t n) {
    // grid-stride over blocks of 256*U elements
    for (size_t base = (size_t)blockIdx.x * 256 * U; base < n; base += (size_t)gridDim.x * 256 * U) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t i = base + u * 256 + threadIdx.x;
            v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t i = base + u * 256 + threadIdx.x;
            if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u];
        }
    }
}
template <int U>
__global__ __launch_bounds__(256) void k_read(const d2 *__restrict__ src, double *out, size_t n) {
    double acc = 0;
    for (size_t base = (size_t)blockIdx.x * 256 * U; base < n; base += (size_t)gridDim.x * 256 * U) {
#pragma unroll
        for (int u = 0; u < U; u++) { d2 v = src[base + u * 256 + threadIdx.x]; acc += v.x + v.y; }
    }
    if (acc == 1.2345) out[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void k_rmw(d2 *__restrict__ buf, size_t n) {
    for (size_t base = (size_t)blockIdx.x * 256 * U; base < n; base += (size_t)gridDim.x * 256 * U) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = buf[base + u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < U; u++) { v[u].x += 1.0; buf[base + u * 256 + threadIdx.x] = v[u]; }
    }
}
template <class F> float timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    d2 *a, *b; double *o; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 8));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    int grids[] = {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 64, 0};
    for (int gi = 0; gi < 6; gi++) {
        size_t g = grids[gi] ? grids[gi] : n / (256 * 4);
        float t1 = timeit([&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
        float t2 = timeit([&] { hipLaunchKernelGGL((k_copy<8, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
        float t3 = timeit([&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
        float t4 = timeit([&] { hipLaunchKernelGGL((k_read<8>), dim3(g), dim3(256), 0, 0, a, o, n); });
        float t5 = timeit([&] { hipLaunchKernelGGL((k_rmw<4>), dim3(g), dim3(256), 0, 0, a, n); });
        printf("grid %8zu: copy U4 %.2f TB/s  U8 %.2f  NT %.2f | read %.2f TB/s | rmw %.2f TB/s\n", g,
               2 * bytes / t1 / 1e9, 2 * bytes / t2 / 1e9, 2 * bytes / t3 / 1e9, bytes / t4 / 1e9, 2 * bytes / t5 / 1e9);
    }
    float tm = timeit([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
    printf("hipMemcpy D2D: %.2f TB/s (read+write)\n", 2 * bytes / tm / 1e9);
    return 0;
}
