// cg_rungs.hip — per-particle kernels of the adaptive rung time stepping (A16).
//   Component.nullify_Δ / apply_Δmom (only active rungs)  species.py:3717-3741, 2253-2266
//   Component.convert_Δmom_to_acc                          species.py:2290-2325
//   Component.get_rung / assign_rungs                      species.py:2341-2363, 2422-2445
//   Component.flag_rung_jumps                              species.py:2463-2513
//   Component.apply_rung_jumps                             species.py:2526-2549
// All are streaming, one lane per particle; rung indices are the reference's
// `signed char` arrays (int8), jumped indices carry +N_rungs (down) / +2 N_rungs (up).
#include "cg_internal.h"

#define CG_LAUNCH_CHECK()                                                                     \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) {                                                               \
            cg_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                           \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

typedef signed char i8;

static inline unsigned nblocks(i64 n) { return (unsigned)((n + 255) / 256); }

// op 0: Δmom = 0 for active particles; op 1: mom += Δmom for active particles
__global__ void k_dmom_active(double *__restrict__ mom, double *__restrict__ dmom,
                              const i8 *__restrict__ rung, i64 n, int lowest_active, int op) {
    i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (rung && rung[p] < lowest_active) return;
    if (op == 0) {
        dmom[3 * p] = 0;
        dmom[3 * p + 1] = 0;
        dmom[3 * p + 2] = 0;
    } else {
        mom[3 * p] += dmom[3 * p];
        mom[3 * p + 1] += dmom[3 * p + 1];
        mom[3 * p + 2] += dmom[3 * p + 2];
    }
}

// Δmom *= conversion_factors[rung(_jumped)] for active particles (species.py:2311-2325)
__global__ void k_dmom_to_acc(double *__restrict__ dmom, const i8 *__restrict__ rung,
                              const i8 *__restrict__ rung_jumped, i64 n, int lowest_active,
                              const double *__restrict__ conv, int any_jumps) {
    i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int r = rung[p];
    if (r < lowest_active) return;
    double f = conv[any_jumps ? rung_jumped[p] : r];
    dmom[3 * p] *= f;
    dmom[3 * p + 1] *= f;
    dmom[3 * p + 2] *= f;
}

// get_rung, species.py:2341-2363
__device__ __forceinline__ int get_rung(const double *__restrict__ dmom, i64 p, int current,
                                        double rung_factor, int N_rungs) {
    double ax = dmom[3 * p], ay = dmom[3 * p + 1], az = dmom[3 * p + 2];
    double acc2 = ax * ax + ay * ay + az * az;
    if (acc2 == 0) return current;
    double f = rung_factor + 0.25 * log2(acc2);
    if (f < 0) return 0;
    if (f > N_rungs - 1) return N_rungs - 1;
    return 1 + (int)(i8)f;
}

__global__ void k_assign_rungs(const double *__restrict__ dmom, i8 *__restrict__ rung,
                               i8 *__restrict__ rung_jumped, i64 n, double rung_factor,
                               int N_rungs) {
    i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int r = get_rung(dmom, p, rung[p], rung_factor, N_rungs);
    rung[p] = (i8)r;
    rung_jumped[p] = (i8)r;  // no jump
}

// flag_rung_jumps, species.py:2476-2512; *any_out != 0 iff a jump was flagged
__global__ void k_flag_rung_jumps(const double *__restrict__ dmom, const i8 *__restrict__ rung,
                                  i8 *__restrict__ rung_jumped, i64 n, int lowest_active,
                                  const double *__restrict__ integrals, double rf_up,
                                  double rf_down, int N_rungs, int *__restrict__ any_out) {
    i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int r = rung[p];
    if (lowest_active > 0 && r < lowest_active) return;
    if (integrals[r] == 0) return;
    int ought = get_rung(dmom, p, r, rf_up, N_rungs);
    if (ought > r) {
        rung_jumped[p] = (i8)(r + 2 * N_rungs);
        *any_out = 1;
        return;
    }
    int down = r + N_rungs;
    if (integrals[down] == -1) return;
    ought = get_rung(dmom, p, r, rf_down, N_rungs);
    if (ought < r) {
        rung_jumped[p] = (i8)down;
        *any_out = 1;
    }
}

// apply_rung_jumps, species.py:2536-2547
__global__ void k_apply_rung_jumps(i8 *__restrict__ rung, i8 *__restrict__ rung_jumped, i64 n,
                                   int N_rungs) {
    i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int j = rung_jumped[p];
    if (j < N_rungs) return;
    int r = rung[p] + (2 * (j >= 2 * N_rungs) - 1);
    rung[p] = (i8)r;
    rung_jumped[p] = (i8)r;
}

// set_rungs_N, species.py:2560-2587: how many particles sit on each rung.  Every thread takes 16
// consecutive particles; per wave one ballot per rung and member, the counts gathered in LDS, one
// atomic per rung and workgroup.
__global__ __launch_bounds__(256) void k_rung_populations(const i8 *__restrict__ rung, i64 n,
                                                          int N_rungs,
                                                          unsigned long long *__restrict__ counts) {
    __shared__ unsigned s_cnt[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const i64 first = ((i64)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    unsigned mine[16];
#pragma unroll
    for (int k = 0; k < 16; k++) mine[k] = first + k < n ? (unsigned)(unsigned char)rung[first + k] : 255u;
    for (int r = 0; r < N_rungs && r < 64; r++) {
        unsigned c = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) c += (unsigned)__popcll(__ballot(mine[k] == (unsigned)r));
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt[r], c);
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)N_rungs && threadIdx.x < 64 && s_cnt[threadIdx.x])
        atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}
int cgk_rung_populations(cg_ctx *c, const i8 *rung, i64 n, int N_rungs, long long *counts) {
    if (hipMemsetAsync(counts, 0, sizeof(long long) * (size_t)N_rungs, c->stream) != hipSuccess) {
        cg_set_error("cg_rung_populations: hipMemsetAsync failed");
        return 1;
    }
    if (n == 0) return 0;
    const i64 per = 256 * 16;
    hipLaunchKernelGGL(k_rung_populations, dim3((unsigned)((n + per - 1) / per)), dim3(256), 0,
                       c->stream, rung, n, N_rungs, (unsigned long long *)counts);
    CG_LAUNCH_CHECK();
    return 0;
}

int cgk_dmom_active(cg_ctx *c, double *mom, double *dmom, const i8 *rung, i64 n, int lowest_active,
                    int op) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_dmom_active, dim3(nblocks(n)), dim3(256), 0, c->stream, mom, dmom, rung, n,
                       lowest_active, op);
    CG_LAUNCH_CHECK();
    return 0;
}
int cgk_dmom_to_acc(cg_ctx *c, double *dmom, const i8 *rung, const i8 *rung_jumped, i64 n,
                    int lowest_active, const double *conv, int any_jumps) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_dmom_to_acc, dim3(nblocks(n)), dim3(256), 0, c->stream, dmom, rung,
                       rung_jumped, n, lowest_active, conv, any_jumps);
    CG_LAUNCH_CHECK();
    return 0;
}
int cgk_assign_rungs(cg_ctx *c, const double *dmom, i8 *rung, i8 *rung_jumped, i64 n,
                     double rung_factor, int N_rungs) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_assign_rungs, dim3(nblocks(n)), dim3(256), 0, c->stream, dmom, rung,
                       rung_jumped, n, rung_factor, N_rungs);
    CG_LAUNCH_CHECK();
    return 0;
}
int cgk_flag_rung_jumps(cg_ctx *c, const double *dmom, const i8 *rung, i8 *rung_jumped, i64 n,
                        int lowest_active, const double *integrals, double rf_up, double rf_down,
                        int N_rungs, int *any_out) {
    CG_HIP(hipMemsetAsync(any_out, 0, sizeof(int), c->stream));
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_flag_rung_jumps, dim3(nblocks(n)), dim3(256), 0, c->stream, dmom, rung,
                       rung_jumped, n, lowest_active, integrals, rf_up, rf_down, N_rungs, any_out);
    CG_LAUNCH_CHECK();
    return 0;
}
int cgk_apply_rung_jumps(cg_ctx *c, i8 *rung, i8 *rung_jumped, i64 n, int N_rungs) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_apply_rung_jumps, dim3(nblocks(n)), dim3(256), 0, c->stream, rung,
                       rung_jumped, n, N_rungs);
    CG_LAUNCH_CHECK();
    return 0;
}
