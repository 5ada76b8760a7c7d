// cg_rungs.hip — per-particle kernels of the adaptive rung time stepping (A16).
//   Component.nullify_Δ / apply_Δmom (only active rungs)  species.py:3717-3741, 2253-2266
//   Component.convert_Δmom_to_acc                          species.py:2290-2325
//   Component.get_rung / assign_rungs                      species.py:2341-2363, 2422-2445
//   Component.flag_rung_jumps                              species.py:2463-2513
//   Component.apply_rung_jumps                             species.py:2526-2549
// All are streaming, one lane per particle; rung indices are the reference's
// `signed char` arrays (int8), jumped indices carry +N_rungs (down) / +2 N_rungs (up).
#include "cg_internal.h"
#include "cg_substep.h"

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

// get_rung (species.py:2341-2363): cg_substep.h
#define get_rung cg_get_rung

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

// ---------------------------------------------------------------------------
// A sub-step of driftkick_short (main.py:1347-1624) in two passes over the particles instead of
// seven.  Every kernel above is one lane per particle and touches only that particle's rows, so
// running them back to back per particle gives the same values:
//   begin: Component.drift (species.py:2179-2199) -> flag_rung_jumps -> nullify_Δ('mom')
//   end:   apply_Δmom -> convert_Δmom_to_acc -> apply_rung_jumps -> set_rungs_N
// The rung tables (3 N_rungs - 1 doubles) travel as kernel arguments: no upload, no wait.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_substep_begin(SubstepBegin B) {
    __shared__ unsigned s_cnt[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    int r = 255;
    if (p < B.n) {
        double x, y, z;
        r = cg_substep_begin_particle(B, p, x, y, z);
    }
    if (!B.partial) return;
    cg_count_rungs(r, B.N_rungs, s_cnt);
    __syncthreads();
    if (threadIdx.x < (unsigned)B.N_rungs)
        B.partial[(i64)B.N_rungs * blockIdx.x + threadIdx.x] = s_cnt[threadIdx.x];
}
// counts[r] = sum over the workgroups' partial[N_rungs * w + r]: a workgroup per rung, fixed order
__global__ __launch_bounds__(1024) void k_substep_populations(const unsigned *__restrict__ partial,
                                                              i64 nwg, int N_rungs,
                                                              unsigned long long *__restrict__ counts) {
    __shared__ unsigned long long red[1024];
    const int r = blockIdx.x;
    unsigned long long s = 0;
    for (i64 w = threadIdx.x; w < nwg; w += 1024) s += partial[(i64)N_rungs * w + r];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 512; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[r] = red[0];
}
int cgk_substep_populations(cg_ctx *c, i64 nwg, int N_rungs, long long *counts) {
    hipLaunchKernelGGL(k_substep_populations, dim3(N_rungs), dim3(1024), 0, c->stream, c->sub_partial,
                       nwg, N_rungs, (unsigned long long *)counts);
    CG_LAUNCH_CHECK();
    return 0;
}
// room for the per-workgroup populations of a pass over n particles in workgroups of `per`
int cgk_substep_partial(cg_ctx *c, i64 n, i64 per, int N_rungs, i64 *nwg_out) {
    const i64 nwg = (n + per - 1) / per;
    const size_t need = sizeof(unsigned) * (size_t)N_rungs * (size_t)(nwg + 1);
    if (need > c->sub_partial_bytes) {
        CG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->sub_partial);
        c->sub_partial = nullptr;
        c->sub_partial_bytes = 0;
        CG_HIP(hipMalloc((void **)&c->sub_partial, need));
        c->sub_partial_bytes = need;
    }
    *nwg_out = nwg;
    return 0;
}

// (a fixed number of workgroups, every thread a strided share: the populations cost 8 atomics
// per workgroup on 8 addresses, which one workgroup per 256 particles would turn into 65,000
// per address at 256^3)
constexpr int kSubstepEndBlocks = 2048;
template <bool APPLY>
__global__ __launch_bounds__(256) void k_substep_end(double *__restrict__ mom,
                                                     double *__restrict__ dmom,
                                                     i8 *__restrict__ rung,
                                                     i8 *__restrict__ rung_jumped, i64 n,
                                                     int lowest_active, RungTable conv,
                                                     int N_rungs,
                                                     unsigned long long *__restrict__ counts) {
    __shared__ unsigned s_cnt[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 p0 = (i64)blockIdx.x * blockDim.x; p0 < n; p0 += stride) {
        const i64 p = p0 + threadIdx.x;
        int r = 255;
        if (p < n) {
            r = rung[p];
            const int j = rung_jumped[p];
            if (APPLY && r >= lowest_active) {
                const double f = conv.v[j];  // species.py:2311-2325 (no jump: j == r)
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const double dm = dmom[3 * p + d];
                    mom[3 * p + d] += dm;      // apply_Δmom, species.py:2253-2266
                    dmom[3 * p + d] = dm * f;  // convert_Δmom_to_acc
                }
            }
            if (j >= N_rungs) {  // apply_rung_jumps, species.py:2536-2547
                r += 2 * (j >= 2 * N_rungs) - 1;
                rung[p] = (i8)r;
                rung_jumped[p] = (i8)r;
            }
        }
        // set_rungs_N, species.py:2560-2587 (null: the sub-step's first pass has counted)
        if (counts) cg_count_rungs(r, N_rungs, s_cnt);
    }
    __syncthreads();
    if (counts && threadIdx.x < (unsigned)N_rungs && threadIdx.x < 64 && s_cnt[threadIdx.x])
        atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// defer: nothing is launched — the cell list the sub-step's sweep asks for next (cg_shortrange_cells
// [_rungs] on the same positions) runs the pass on every particle it bins, cgk_substep_flush
// (any other use of the particles) launches it by itself
int cgk_substep_begin(cg_ctx *c, double *pos, const double *mom, double *dmom, const i8 *rung,
                      i8 *rung_jumped, i64 n, int do_drift, double dt_over_mass, int do_flag,
                      int lowest_active, const double *integrals_1, double rf_up, double rf_down,
                      int N_rungs, int *any_out, long long *counts_after, int defer) {
    if (cgk_substep_flush(c)) return 1;
    if (do_flag) CG_HIP(hipMemsetAsync(any_out, 0, sizeof(int), c->stream));
    if (n == 0 || (!do_drift && !do_flag)) return 0;
    SubstepBegin &B = *c->sub_begin;
    B = SubstepBegin{pos, mom, dmom, rung, rung_jumped, n, do_drift, do_flag, dt_over_mass,
                     c->p.boxsize, lowest_active, RungTable{}, rf_up, rf_down, N_rungs, any_out,
                     nullptr};
    if (do_flag)
        for (int i = 0; i < 3 * N_rungs - 1; i++) B.integrals.v[i] = integrals_1[i];
    c->sub_counts = do_flag ? counts_after : nullptr;
    c->sub_pending = true;
    return defer ? 0 : cgk_substep_flush(c);
}
int cgk_substep_flush(cg_ctx *c) {
    if (!c->sub_pending) return 0;
    c->sub_pending = false;
    SubstepBegin &B = *c->sub_begin;
    i64 nwg = 0;
    if (c->sub_counts) {
        if (cgk_substep_partial(c, B.n, 256, B.N_rungs, &nwg)) return 1;
        B.partial = c->sub_partial;
    }
    hipLaunchKernelGGL(k_substep_begin, dim3(nblocks(B.n)), dim3(256), 0, c->stream, B);
    CG_LAUNCH_CHECK();
    if (c->sub_counts && cgk_substep_populations(c, nwg, B.N_rungs, c->sub_counts)) return 1;
    return 0;
}

int cgk_substep_end(cg_ctx *c, double *mom, double *dmom, i8 *rung, i8 *rung_jumped, i64 n,
                    int do_apply, int lowest_active, const double *conversion_factors, int N_rungs,
                    long long *counts) {
    if (counts) CG_HIP(hipMemsetAsync(counts, 0, sizeof(long long) * (size_t)N_rungs, c->stream));
    if (n == 0) return 0;
    RungTable T{};
    if (do_apply)
        for (int i = 0; i < 3 * N_rungs - 1; i++) T.v[i] = conversion_factors[i];
    const i64 want = (n + 255) / 256;
    const unsigned blocks = (unsigned)(want < kSubstepEndBlocks ? want : kSubstepEndBlocks);
    hipLaunchKernelGGL(do_apply ? k_substep_end<true> : k_substep_end<false>, dim3(blocks),
                       dim3(256), 0, c->stream, mom, dmom, rung, rung_jumped, n, lowest_active, T,
                       N_rungs, (unsigned long long *)counts);
    CG_LAUNCH_CHECK();
    return 0;
}
