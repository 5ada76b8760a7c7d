// cg_shortrange.hip — P3M short-range tile sweep (A13-A15) on gfx950.
//
//   Tiling.sort                 species.py:707-823      particle -> tile
//   particle_particle           interactions.py:1563-1791  tile neighbours, periodic offset,
//                                                         x_ji = xi - xj
//   gravity_pairwise_shortrange gravity.py:263-354      r2 cut, r2-indexed table, Δmom
//
// Form: one-sided.  The reference visits every unordered pair once and updates
// both partners (Δmom_r += r*f, Δmom_s -= r*f); here every receiver particle
// sums over all its partners itself — twice the arithmetic, but no atomics, no
// write conflicts (the order of partners inside a tile follows the cell-list
// scatter, so sums are reproducible to rounding, not bit for bit).  The pair
// vector, r2 and the table index are evaluated with the reference's expression
// and operation order ((xi - xj) + offset; x*x + y*y + z*z; int(r2*scaling)):
// exact negation symmetry makes the two directions of a pair bit-consistent.
//
// Layout: a cell list over the short-range tiling (uint32 order[] + offset[]);
// one wavefront per receiver tile; supplier tiles are staged through LDS in
// chunks of 64 and broadcast to all lanes.
#include <hipcub/hipcub.hpp>

#include <cstdlib>

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

// Tiling.sort (species.py:775-780) with tiling location 0:
//   i = int((x - loc*(1 + 2 eps))*((1/tile_extent)*(1 - 2 eps)))
__device__ __forceinline__ unsigned sr_tile(const double *__restrict__ pos, i64 p, double inv,
                                            unsigned nt) {
    unsigned i = (unsigned)(i64)((pos[3 * p + 0] - 0.0) * inv);
    unsigned j = (unsigned)(i64)((pos[3 * p + 1] - 0.0) * inv);
    unsigned k = (unsigned)(i64)((pos[3 * p + 2] - 0.0) * inv);
    // a position exactly at boxsize cannot occur (drift wraps into [0, L)); clamp anyway
    i = i >= nt ? nt - 1 : i;
    j = j >= nt ? nt - 1 : j;
    k = k >= nt ? nt - 1 : k;
    return (i * nt + j) * nt + k;
}

// runs of equal keys inside a wavefront -> one atomic per run (device-scope atomics are
// memory-side on MI355X; particle memory is in mesh-tile order, so runs are long)
__device__ __forceinline__ void sr_wave_runs(unsigned key, int lane, int &run_start, int &run_len) {
    unsigned prev = __shfl_up(key, 1);
    bool head = (lane == 0) || (key != prev);
    unsigned long long mask = __ballot(head);
    unsigned long long below = mask & (~0ull >> (63 - lane));
    run_start = 63 - __clzll(below);
    unsigned long long above = (lane == 63) ? 0ull : (mask >> (lane + 1));
    int next = above ? (lane + 1 + (__ffsll((long long)above) - 1)) : 64;
    run_len = next - run_start;
}

__global__ __launch_bounds__(256) void k_sr_histogram(const double *__restrict__ pos, i64 n,
                                                      double inv, unsigned nt,
                                                      unsigned *__restrict__ count) {
    i64 stride = (i64)gridDim.x * blockDim.x;
    int lane = threadIdx.x & 63;
    for (i64 base = (i64)blockIdx.x * blockDim.x; base < n; base += stride) {
        i64 p = base + threadIdx.x;
        unsigned key = p < n ? sr_tile(pos, p, inv, nt) : 0xffffffffu;
        int rs, rl;
        sr_wave_runs(key, lane, rs, rl);
        if (lane == rs && p < n) atomicAdd(&count[key], (unsigned)rl);
    }
}
__global__ __launch_bounds__(256) void k_sr_scatter(const double *__restrict__ pos, i64 n,
                                                    double inv, unsigned nt,
                                                    const unsigned *__restrict__ offset,
                                                    unsigned *__restrict__ cursor,
                                                    unsigned *__restrict__ order) {
    i64 stride = (i64)gridDim.x * blockDim.x;
    int lane = threadIdx.x & 63;
    for (i64 base = (i64)blockIdx.x * blockDim.x; base < n; base += stride) {
        i64 p = base + threadIdx.x;
        unsigned key = p < n ? sr_tile(pos, p, inv, nt) : 0xffffffffu;
        int rs, rl;
        sr_wave_runs(key, lane, rs, rl);
        unsigned first = 0;
        if (lane == rs && p < n) first = offset[key] + atomicAdd(&cursor[key], (unsigned)rl);
        first = __shfl(first, rs);
        if (p < n) order[first + (lane - rs)] = (unsigned)p;
    }
}
int cgk_shortrange_build(cg_ctx *c, const double *pos, i64 n, i64 nt, double tile_extent,
                         unsigned *order, unsigned *offset) {
    const double eps = 2.220446049250313e-16;
    double inv = (1 / tile_extent) * (1 - 2 * eps);
    i64 ntiles = nt * nt * nt;
    if ((size_t)(8 * (ntiles + 1)) > c->sr_tmp_bytes) {
        CG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->sr_tmp);
        c->sr_tmp = nullptr;
        CG_HIP(hipMalloc(&c->sr_tmp, 8 * (ntiles + 1)));
        c->sr_tmp_bytes = 8 * (ntiles + 1);
    }
    unsigned *count = (unsigned *)c->sr_tmp, *cursor = count + (ntiles + 1);
    CG_HIP(hipMemsetAsync(c->sr_tmp, 0, 8 * (ntiles + 1), c->stream));
    i64 blocks = (n + 255) / 256;  // one workgroup per 256 particles (see cgk_sort)
    if (n > 0) {
        hipLaunchKernelGGL(k_sr_histogram, dim3((unsigned)blocks), dim3(256), 0, c->stream, pos, n,
                           inv, (unsigned)nt, count);
        CG_LAUNCH_CHECK();
    }
    size_t need = 0;
    CG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, count, offset, (int)(ntiles + 1),
                                            c->stream));
    if (need > c->scan_tmp_bytes) {
        CG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->scan_tmp);
        c->scan_tmp = nullptr;
        CG_HIP(hipMalloc(&c->scan_tmp, need));
        c->scan_tmp_bytes = need;
    }
    CG_HIP(hipcub::DeviceScan::ExclusiveSum(c->scan_tmp, need, count, offset, (int)(ntiles + 1),
                                            c->stream));
    if (n > 0) {
        hipLaunchKernelGGL(k_sr_scatter, dim3((unsigned)blocks), dim3(256), 0, c->stream, pos, n,
                           inv, (unsigned)nt, offset, cursor, order);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

struct SrParams {
    double boxsize, r2_index_scaling, r2_max, factor;
    int nt;
    int same;  // receiver and supplier arrays are the same component
    // adaptive rungs (null = every particle on rung 0 with `factor`): a receiver on an
    // active rung (rung >= lowest_active) is kicked with factors[rung_jumped]; inactive
    // receivers are skipped.  One-sided form of interactions.py:1688-1761 +
    // gravity.py:318-349: each particle's kick uses its OWN rung's integral.
    const double *factors;
    const signed char *rung, *rung_jumped;
    int lowest_active;
};

__global__ __launch_bounds__(64) void k_sr_sweep(
    const double *__restrict__ pos_r, const unsigned *__restrict__ order_r,
    const unsigned *__restrict__ off_r, double *__restrict__ dmom_r,
    const double *__restrict__ pos_s, const unsigned *__restrict__ order_s,
    const unsigned *__restrict__ off_s, const double *__restrict__ table, SrParams P) {
    // A tile holds ~20 particles with the default parameters, far fewer than the 64
    // lanes: the wavefront is split into S = 64/R groups of R lanes (R = the power of
    // two >= the receivers of a chunk); group s takes suppliers s, s+S, s+2S, ... of
    // every staged chunk and the S partial sums of a receiver are folded with shuffles.
    __shared__ double sx[64], sy[64], sz[64];
    __shared__ unsigned sidx[64];
    const int lane = threadIdx.x;
    const int nt = P.nt;
    const unsigned tr = blockIdx.x;
    const unsigned rbeg = off_r[tr], rend = off_r[tr + 1];
    if (rbeg == rend) return;
    const int ra = tr / (nt * nt), rb = (tr / nt) % nt, rc = tr % nt;
    for (unsigned base = rbeg; base < rend; base += 64) {
        const int nrec = (int)min(64u, rend - base);
        int R = 8;
        while (R < nrec) R <<= 1;
        const int S = 64 / R, sub = lane / R, rl = lane % R;
        bool active = rl < nrec;
        const unsigned pi = active ? order_r[base + rl] : 0u;
        double my_factor = P.factor;
        if (active && P.rung) {
            if (P.rung[pi] < P.lowest_active) active = false;
            else my_factor = P.factors[P.rung_jumped[pi]];
        }
        double xi = 0, yi = 0, zi = 0;
        if (active) {
            xi = pos_r[3 * (i64)pi];
            yi = pos_r[3 * (i64)pi + 1];
            zi = pos_r[3 * (i64)pi + 2];
        }
        double ax = 0, ay = 0, az = 0;
        for (int d = 0; d < 27; d++) {
            int sa = ra + d / 9 - 1, sb = rb + (d / 3) % 3 - 1, sc = rc + d % 3 - 1;
            // periodic offset from the tile separation (interactions.py:1615-1621)
            double ox = 0, oy = 0, oz = 0;
            if (sa < 0) { sa += nt; ox = P.boxsize; } else if (sa >= nt) { sa -= nt; ox = -P.boxsize; }
            if (sb < 0) { sb += nt; oy = P.boxsize; } else if (sb >= nt) { sb -= nt; oy = -P.boxsize; }
            if (sc < 0) { sc += nt; oz = P.boxsize; } else if (sc >= nt) { sc -= nt; oz = -P.boxsize; }
            const bool shifted = (ox != 0) | (oy != 0) | (oz != 0);
            const unsigned ts = (unsigned)((sa * nt + sb) * nt + sc);
            const unsigned sbeg = off_s[ts], send = off_s[ts + 1];
            for (unsigned cb = sbeg; cb < send; cb += 64) {
                __syncthreads();
                if (cb + lane < send) {
                    unsigned pj = order_s[cb + lane];
                    sidx[lane] = pj;
                    sx[lane] = pos_s[3 * (i64)pj];
                    sy[lane] = pos_s[3 * (i64)pj + 1];
                    sz[lane] = pos_s[3 * (i64)pj + 2];
                }
                __syncthreads();
                const int cnt = (int)min(64u, send - cb);
                if (active) {
                    // straight-line body (predicated, no early exits) so that the compiler
                    // can overlap the LDS reads and FP64 chains of several partners
#pragma unroll 4
                    for (int k = sub; k < cnt; k += S) {
                        double x_ji = xi - sx[k];               // interactions.py:1787-1789
                        double y_ji = yi - sy[k];
                        double z_ji = zi - sz[k];
                        if (shifted) {                          // gravity.py:299-302 (uniform)
                            x_ji += ox;
                            y_ji += oy;
                            z_ji += oz;
                        }
                        double r2 = x_ji * x_ji + y_ji * y_ji + z_ji * z_ji;  // gravity.py:306
                        bool hit = !(r2 > P.r2_max) && !(P.same && sidx[k] == pi);
                        if (hit) {
                            int idx = (int)(r2 * P.r2_index_scaling);         // gravity.py:316 (< 4096: int)
                            double total_factor = my_factor * table[idx];    // gravity.py:321
                            ax += x_ji * total_factor;
                            ay += y_ji * total_factor;
                            az += z_ji * total_factor;
                        }
                    }
                }
            }
        }
        // fold the S partial sums of each receiver (lanes rl, rl + R, rl + 2R, ...)
        for (int o = 32; o >= R; o >>= 1) {
            ax += __shfl_down(ax, o);
            ay += __shfl_down(ay, o);
            az += __shfl_down(az, o);
        }
        if (active && sub == 0) {
            dmom_r[3 * (i64)pi] += ax;
            dmom_r[3 * (i64)pi + 1] += ay;
            dmom_r[3 * (i64)pi + 2] += az;
        }
    }
}

// ---------------------------------------------------------------------------
// Column form of the sweep.  With the default tiling a tile holds ~20 particles,
// so one wavefront per tile leaves two thirds of the lanes idle.  Here a
// workgroup owns a z-COLUMN of tiles (ra, rb, *): the cell list is z-fastest, so
// the column's receivers are one contiguous run, cut into chunks of 64 lanes.
// A chunk spans a few tiles tc_lo..tc_hi; its lanes test every particle of the
// supplier tiles (sa, sb, tc_lo-1 .. tc_hi+1) of the 9 neighbouring columns.
// A lane takes part only for the three supplier tiles adjacent to its own tile
// (exactly the reference's pairs, each once), so a chunk spanning three tiles
// keeps ~60 % of the lanes busy instead of ~34 %.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_sr_sweep_columns(
    const double *__restrict__ pos_r, const unsigned *__restrict__ order_r,
    const unsigned *__restrict__ off_r, double *__restrict__ dmom_r,
    const double *__restrict__ pos_s, const unsigned *__restrict__ order_s,
    const unsigned *__restrict__ off_s, const double *__restrict__ table, SrParams P,
    double inv_extent) {
    __shared__ double sx[64], sy[64], sz[64];
    __shared__ unsigned sidx[64];
    const int lane = threadIdx.x;
    const int nt = P.nt;
    const int ra = blockIdx.x / nt, rb = blockIdx.x % nt;
    const unsigned col0 = (unsigned)((ra * nt + rb) * nt);
    const unsigned cbeg = off_r[col0], cend = off_r[col0 + nt];
    for (unsigned base = cbeg; base < cend; base += 64) {
        const bool active = base + lane < cend;
        const unsigned pi = active ? order_r[base + lane] : 0u;
        double xi = 0, yi = 0, zi = 0;
        int my_tc = 0;
        if (active) {
            xi = pos_r[3 * (i64)pi];
            yi = pos_r[3 * (i64)pi + 1];
            zi = pos_r[3 * (i64)pi + 2];
            my_tc = (int)(i64)((zi - 0.0) * inv_extent);  // Tiling.sort expression
            my_tc = my_tc >= nt ? nt - 1 : my_tc;
        }
        // tile range of this chunk (the list is sorted by tile: first / last active lane)
        const int last = (int)min(63u, cend - base - 1);
        const int tc_lo = __shfl(my_tc, 0), tc_hi = __shfl(my_tc, last);
        double ax = 0, ay = 0, az = 0;
        for (int d = 0; d < 9; d++) {
            int sa = ra + d / 3 - 1, sb = rb + d % 3 - 1;
            double ox = 0, oy = 0;
            if (sa < 0) { sa += nt; ox = P.boxsize; } else if (sa >= nt) { sa -= nt; ox = -P.boxsize; }
            if (sb < 0) { sb += nt; oy = P.boxsize; } else if (sb >= nt) { sb -= nt; oy = -P.boxsize; }
            for (int sc_raw = tc_lo - 1; sc_raw <= tc_hi + 1; sc_raw++) {
                int sc = sc_raw;
                double oz = 0;
                if (sc < 0) { sc += nt; oz = P.boxsize; } else if (sc >= nt) { sc -= nt; oz = -P.boxsize; }
                const bool shifted = (ox != 0) | (oy != 0) | (oz != 0);
                const unsigned ts = (unsigned)((sa * nt + sb) * nt + sc);
                const unsigned sbeg = off_s[ts], send = off_s[ts + 1];
                for (unsigned cb = sbeg; cb < send; cb += 64) {
                    __syncthreads();
                    if (cb + lane < send) {
                        unsigned pj = order_s[cb + lane];
                        sidx[lane] = pj;
                        sx[lane] = pos_s[3 * (i64)pj];
                        sy[lane] = pos_s[3 * (i64)pj + 1];
                        sz[lane] = pos_s[3 * (i64)pj + 2];
                    }
                    __syncthreads();
                    const int cnt = (int)min(64u, send - cb);
                    // a supplier tile two or more tiles away from a lane's own tile is not
                    // its neighbour: with a periodic shift its image could alias a true
                    // neighbour, so such lanes sit the tile out
                    int dz = sc_raw - my_tc;
                    if (active && dz >= -1 && dz <= 1) {
                        for (int k = 0; k < cnt; k++) {
                            if (P.same && sidx[k] == pi) continue;
                            double x_ji = xi - sx[k];
                            double y_ji = yi - sy[k];
                            double z_ji = zi - sz[k];
                            if (shifted) {
                                x_ji += ox;
                                y_ji += oy;
                                z_ji += oz;
                            }
                            double r2 = x_ji * x_ji + y_ji * y_ji + z_ji * z_ji;
                            if (r2 > P.r2_max) continue;
                            int idx = (int)(r2 * P.r2_index_scaling);
                            double total_factor = P.factor * table[idx];
                            ax += x_ji * total_factor;
                            ay += y_ji * total_factor;
                            az += z_ji * total_factor;
                        }
                    }
                }
            }
        }
        if (active) {
            dmom_r[3 * (i64)pi] += ax;
            dmom_r[3 * (i64)pi + 1] += ay;
            dmom_r[3 * (i64)pi + 2] += az;
        }
    }
}

int cgk_shortrange_sweep(cg_ctx *c, const double *pos_r, const unsigned *order_r,
                         const unsigned *off_r, double *dmom_r, const double *pos_s,
                         const unsigned *order_s, const unsigned *off_s, i64 nt, int same,
                         const double *table, double r2_index_scaling, double r2_max,
                         double factor, const double *factors, const signed char *rung,
                         const signed char *rung_jumped, int lowest_active) {
    SrParams P{c->p.boxsize, r2_index_scaling, r2_max, factor, (int)nt, same,
               factors,      rung,             rung_jumped, lowest_active};
    // Default: one wavefront per tile.  The column form (CONCEPT_GPU_SR=columns) keeps more
    // lanes busy but measured slower at 256^3 / 512^3 (26.7 vs 24.7 ms): kept for A/B work.
    const char *env = getenv("CONCEPT_GPU_SR");
    if (rung || !(env && std::string(env) == "columns")) {
        hipLaunchKernelGGL(k_sr_sweep, dim3((unsigned)(nt * nt * nt)), dim3(64), 0, c->stream,
                           pos_r, order_r, off_r, dmom_r, pos_s, order_s, off_s, table, P);
    } else {
        const double eps = 2.220446049250313e-16;
        double tile_extent = c->p.boxsize / (double)nt;  // species.py:607-609
        double inv = (1 / tile_extent) * (1 - 2 * eps);
        hipLaunchKernelGGL(k_sr_sweep_columns, dim3((unsigned)(nt * nt)), dim3(64), 0, c->stream,
                           pos_r, order_r, off_r, dmom_r, pos_s, order_s, off_s, table, P, inv);
    }
    CG_LAUNCH_CHECK();
    return 0;
}
