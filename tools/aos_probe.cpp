// Does the AoS particle layout (double[3N], xyzxyz..., 8-byte accesses 24 bytes apart across
// the lanes of a wavefront) cost bandwidth against SoA (three arrays, 8-byte accesses
// contiguous across lanes)?  Streams 2^28 particles' pos + mom through both.
//   hipcc --offload-arch=gfx950 -O3 tools/aos_probe.cpp -o tools/aos_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_aos(const double *__restrict__ a, const double *__restrict__ b,
                                             double *__restrict__ c, double *__restrict__ d, long n) {
    long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    double x = a[3 * p], y = a[3 * p + 1], z = a[3 * p + 2];
    double u = b[3 * p], v = b[3 * p + 1], w = b[3 * p + 2];
    c[3 * p] = x + u, c[3 * p + 1] = y + v, c[3 * p + 2] = z + w;
    d[3 * p] = u, d[3 * p + 1] = v, d[3 * p + 2] = w;
}
__global__ __launch_bounds__(256) void k_soa(const double *__restrict__ a, const double *__restrict__ b,
                                             double *__restrict__ c, double *__restrict__ d, long n) {
    long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    double x = a[p], y = a[p + n], z = a[p + 2 * n];
    double u = b[p], v = b[p + n], w = b[p + 2 * n];
    c[p] = x + u, c[p + n] = y + v, c[p + 2 * n] = z + w;
    d[p] = u, d[p + n] = v, d[p + 2 * n] = w;
}
int main() {
    const long n = 1L << 28;
    double *a, *b, *c, *d;
    for (double **q : {&a, &b, &c, &d}) { hipMalloc(q, 24 * n); hipMemset(*q, 0, 24 * n); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        auto run = [&]() {
            if (which == 0) hipLaunchKernelGGL(k_aos, dim3(n / 256), dim3(256), 0, 0, a, b, c, d, n);
            else hipLaunchKernelGGL(k_soa, dim3(n / 256), dim3(256), 0, 0, a, b, c, d, n);
        };
        run(); hipEventRecord(e0);
        for (int i = 0; i < 5; i++) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%s: %.3f ms  %.1f GB/s\n", which ? "SoA" : "AoS", ms, 96.0 * n / ms / 1e6);
    }
    return 0;
}
