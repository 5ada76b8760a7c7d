// tools/fft_probe.cpp — times rocFFT plan variants for the 1024^3 FP64 real
// transform to decide how the Poisson solve should be decomposed.
//   hipcc --offload-arch=gfx950 -O2 tools/fft_probe.cpp -o /tmp/fft_probe -lrocfft
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("FAIL %s = %d line %d\n", #x, (int)e_, __LINE__); exit(1);} } while (0)

static double time_plan(rocfft_plan plan, void* in, void* out, int reps) {
    size_t wb = 0; CK(rocfft_plan_get_work_buffer_size(plan, &wb));
    void* work = nullptr; rocfft_execution_info info; CK(rocfft_execution_info_create(&info));
    if (wb) { CK(hipMalloc(&work, wb)); CK(rocfft_execution_info_set_work_buffer(info, work, wb)); }
    void* ib[1] = {in}; void* ob[1] = {out};
    CK(rocfft_execute(plan, ib, out ? ob : nullptr, info)); CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; r++) CK(rocfft_execute(plan, ib, out ? ob : nullptr, info));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (work) CK(hipFree(work));
    rocfft_execution_info_destroy(info);
    printf("   work buffer %.2f GB\n", wb / 1e9);
    return ms / reps;
}

int main(int argc, char** argv) {
    size_t N = argc > 1 ? atoi(argv[1]) : 1024;
    CK(rocfft_setup());
    for (size_t P : {N + 2, N + 16}) {
        size_t nreal = N * N * P;
        double* buf; CK(hipMalloc(&buf, nreal * 8)); CK(hipMemset(buf, 0, nreal * 8));
        size_t off[1] = {0};
        printf("=== pitch P = N+%zu\n", P - N);
        {   // A: 3-D in-place R2C
            size_t len[3] = {N, N, N}, rs[3] = {1, P, P * N}, cs[3] = {1, P / 2, P / 2 * N};
            rocfft_plan_description d; CK(rocfft_plan_description_create(&d));
            CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, off, off, 3, rs, P * N * N, 3, cs, P / 2 * N * N));
            rocfft_plan p; auto st = rocfft_plan_create(&p, rocfft_placement_inplace, rocfft_transform_type_real_forward, rocfft_precision_double, 3, len, 1, d);
            if (st == rocfft_status_success) { printf("A 3D in-place R2C: %.3f ms\n", time_plan(p, buf, nullptr, 5)); rocfft_plan_destroy(p); }
            else printf("A 3D in-place R2C: plan failed %d\n", (int)st);
        }
        {   // B: batched 1-D R2C along z, in place (N*N rows)
            size_t len[1] = {N}, rs[1] = {1}, cs[1] = {1};
            rocfft_plan_description d; CK(rocfft_plan_description_create(&d));
            CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, off, off, 1, rs, P, 1, cs, P / 2));
            rocfft_plan p; auto st = rocfft_plan_create(&p, rocfft_placement_inplace, rocfft_transform_type_real_forward, rocfft_precision_double, 1, len, N * N, d);
            if (st == rocfft_status_success) { printf("B 1D z R2C batched in-place: %.3f ms\n", time_plan(p, buf, nullptr, 5)); rocfft_plan_destroy(p); }
            else printf("B plan failed %d\n", (int)st);
        }
        {   // C: 1-D C2C along y (stride P/2), batch over kk (fast, dist 1) for one i; 2-D batching: lengths {N}, howmany = P/2 per i plane -> use 2D trick: dims = {y} with batch = (P/2) and loop over i? use rank-1 with batch N*(P/2) not expressible; do per-i plan and N launches
            size_t len[1] = {N}, s[1] = {P / 2};
            rocfft_plan_description d; CK(rocfft_plan_description_create(&d));
            CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, off, off, 1, s, 1, 1, s, 1));
            rocfft_plan p; auto st = rocfft_plan_create(&p, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 1, len, N / 2 + 1, d);
            if (st == rocfft_status_success) {
                size_t wb = 0; rocfft_plan_get_work_buffer_size(p, &wb);
                rocfft_execution_info info; CK(rocfft_execution_info_create(&info));
                void* work = nullptr; if (wb) { CK(hipMalloc(&work, wb)); CK(rocfft_execution_info_set_work_buffer(info, work, wb)); }
                hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(a));
                    for (size_t i = 0; i < N; i++) { void* ib[1] = {(char*)buf + i * (P / 2) * N * 16}; CK(rocfft_execute(p, ib, nullptr, info)); }
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                }
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                printf("C 1D y C2C strided, N launches (batch over kk): %.3f ms  (work %.3f GB)\n", ms, wb / 1e9);
                rocfft_plan_destroy(p);
            } else printf("C plan failed %d\n", (int)st);
        }
        {   // D: 1-D C2C along x (stride P/2*N), batch over (j,kk) contiguous: N*(P/2) transforms with dist 1
            size_t len[1] = {N}, s[1] = {P / 2 * N};
            rocfft_plan_description d; CK(rocfft_plan_description_create(&d));
            CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, off, off, 1, s, 1, 1, s, 1));
            rocfft_plan p; auto st = rocfft_plan_create(&p, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 1, len, N * (P / 2), d);
            if (st == rocfft_status_success) { printf("D 1D x C2C strided batched in-place: %.3f ms\n", time_plan(p, buf, nullptr, 5)); rocfft_plan_destroy(p); }
            else printf("D plan failed %d\n", (int)st);
        }
        {   // E: 2-D C2C over (y, x) strided, batch over kk (dist 1): lengths {N(y), N(x)}
            size_t len[2] = {N, N}, s[2] = {P / 2, P / 2 * N};
            rocfft_plan_description d; CK(rocfft_plan_description_create(&d));
            CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, off, off, 2, s, 1, 2, s, 1));
            rocfft_plan p; auto st = rocfft_plan_create(&p, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 2, len, N / 2 + 1, d);
            if (st == rocfft_status_success) { printf("E 2D (y,x) C2C strided batched over kk in-place: %.3f ms\n", time_plan(p, buf, nullptr, 5)); rocfft_plan_destroy(p); }
            else printf("E plan failed %d\n", (int)st);
        }
        CK(hipFree(buf));
    }
    return 0;
}
