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

// The pair tests of one staged supplier chunk for this lane's receiver: branch-free (a miss adds
// x*0) and with a wave-uniform trip count, so that the loop unrolls into straight-line code with
// several table loads in flight.  SHIFTED (the supplier tile is a periodic image) and SELF (it is
// the receiver's own tile of the same component: skip i == j) are wave-uniform and resolved at
// compile time — as run-time conditions inside the loop they cost 9 selects and an LDS read per
// pair (rocprofv3: the sweep is VALU-bound, VALUBusy 75 %).  For a hit the arithmetic and its
// order are the reference's (gravity.py:299-349); r2 and the table index are bit-identical.
constexpr int kSrStage = 128;  // staged suppliers per chunk

template <bool SHIFTED, bool SELF>
__device__ __forceinline__ void sr_pairs(int sub, int S, int cnt, double xi, double yi, double zi,
                                         double ox, double oy, double oz, const double *sx,
                                         const double *sy, const double *sz, const unsigned *sidx,
                                         unsigned pi, double r2_max, double r2_index_scaling,
                                         double my_factor, const double *__restrict__ table,
                                         double &ax, double &ay, double &az) {
#pragma unroll 4
    for (int k0 = 0; k0 < cnt; k0 += S) {
        const int kk = k0 + sub;
        const int k = kk < kSrStage - 1 ? kk : kSrStage - 1;  // the staged arrays' last entry
        double x_ji = xi - sx[k];                        // interactions.py:1787-1789
        double y_ji = yi - sy[k];
        double z_ji = zi - sz[k];
        if (SHIFTED) {                                   // gravity.py:299-302
            x_ji += ox;
            y_ji += oy;
            z_ji += oz;
        }
        const double r2 = x_ji * x_ji + y_ji * y_ji + z_ji * z_ji;  // gravity.py:306
        bool hit = (kk < cnt) & !(r2 > r2_max);
        if (SELF) hit &= sidx[k] != pi;
        // (a fully branch-free form — table[hit ? idx : 0] loaded unconditionally, four loads in
        // flight — measured slower: 15.3 vs 14.1 ms, it costs 12 more registers and the misses'
        // loads; the compiler branches around the hit part per unrolled pair)
        double total_factor = 0.0;
        if (hit) {
            const unsigned idx = (unsigned)(int)(r2 * r2_index_scaling);  // gravity.py:316
            total_factor = my_factor * table[idx];                        // gravity.py:321
        }
        ax += x_ji * total_factor;
        ay += y_ji * total_factor;
        az += z_ji * total_factor;
    }
}

__global__ __launch_bounds__(64) void k_sr_sweep(
    const double *__restrict__ pos_r, const unsigned *__restrict__ order_r,
    const unsigned *__restrict__ off_r, double *__restrict__ dmom_r,
    const double *__restrict__ pos_s, const unsigned *__restrict__ order_s,
    const unsigned *__restrict__ off_s, const double *__restrict__ table, SrParams P) {
    // A tile holds ~20 particles with the default parameters, far fewer than the 64
    // lanes: the wavefront is split into S = 64/R groups of R lanes (R = the receivers of a
    // chunk); group s takes suppliers s, s+S, s+2S, ... of every staged chunk and the S
    // partial sums of a receiver are folded with shuffles.
    __shared__ double sx[kSrStage], sy[kSrStage], sz[kSrStage];
    __shared__ unsigned sidx[kSrStage];
    const int lane = threadIdx.x;
    const int nt = P.nt;
    const unsigned tr = blockIdx.x;
    const unsigned rbeg = off_r[tr], rend = off_r[tr + 1];
    if (rbeg == rend) return;
    const int ra = tr / (nt * nt), rb = (tr / nt) % nt, rc = tr % nt;
    for (unsigned base = rbeg; base < rend; base += 64) {
        const int nrec = (int)min(64u, rend - base);
        // R = nrec receivers x S = 64/nrec supplier groups: 22 receivers (the mean of the default
        // tiling) give 2 x 22 lanes, 21 give 3 x 21 (a power-of-two R left 17..32 receivers at
        // 2 groups)
        const int R = nrec;
        const int S = 64 / R, sub = lane / R, rl = lane - sub * R;
        bool active = sub < S;
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
        // The cell list is z-fastest: the three supplier tiles (sa, sb, rc-1 .. rc+1) of a
        // column are one contiguous run of it, staged and swept as one range (9 ranges of ~66
        // suppliers instead of 27 of ~22: a third of the staging round trips and barriers) unless
        // the column wraps around the box in z, where the three tiles carry different offsets.
        const bool zwrap = rc == 0 || rc == nt - 1;
        for (int d = 0; d < (zwrap ? 27 : 9); d++) {
            int sa, sb, sc0, sc1;
            double ox = 0, oy = 0, oz = 0;
            if (zwrap) {
                sa = ra + d / 9 - 1;
                sb = rb + (d / 3) % 3 - 1;
                sc0 = rc + d % 3 - 1;
                if (sc0 < 0) { sc0 += nt; oz = P.boxsize; } else if (sc0 >= nt) { sc0 -= nt; oz = -P.boxsize; }
                sc1 = sc0;
            } else {
                sa = ra + d / 3 - 1;
                sb = rb + d % 3 - 1;
                sc0 = rc - 1;
                sc1 = rc + 1;
            }
            // periodic offset from the tile separation (interactions.py:1615-1621)
            if (sa < 0) { sa += nt; ox = P.boxsize; } else if (sa >= nt) { sa -= nt; ox = -P.boxsize; }
            if (sb < 0) { sb += nt; oy = P.boxsize; } else if (sb >= nt) { sb -= nt; oy = -P.boxsize; }
            const bool shifted = (ox != 0) | (oy != 0) | (oz != 0);
            const unsigned ts0 = (unsigned)((sa * nt + sb) * nt + sc0);
            const unsigned ts1 = (unsigned)((sa * nt + sb) * nt + sc1);
            const unsigned sbeg = off_s[ts0], send = off_s[ts1 + 1];
            const bool self = P.same && ts0 <= tr && tr <= ts1;  // the range holds the receivers
            for (unsigned cb = sbeg; cb < send; cb += kSrStage) {
                __syncthreads();
#pragma unroll
                for (int h = 0; h < kSrStage / 64; h++) {
                    const int e = lane + 64 * h;
                    if (cb + e < send) {
                        unsigned pj = order_s[cb + e];
                        sidx[e] = pj;
                        sx[e] = pos_s[3 * (i64)pj];
                        sy[e] = pos_s[3 * (i64)pj + 1];
                        sz[e] = pos_s[3 * (i64)pj + 2];
                    } else {
                        // entries past the chunk are read (and masked) by sr_pairs: keep them
                        // finite, a masked pair contributes x*0
                        sx[e] = sy[e] = sz[e] = 0;
                    }
                }
                __syncthreads();
                const int cnt = (int)min((unsigned)kSrStage, send - cb);
                if (active) {
#define CG_SR_PAIRS(SH, SE)                                                                    \
    sr_pairs<SH, SE>(sub, S, cnt, xi, yi, zi, ox, oy, oz, sx, sy, sz, sidx, pi, P.r2_max,      \
                     P.r2_index_scaling, my_factor, table, ax, ay, az)
                    if (shifted) {
                        if (self) CG_SR_PAIRS(true, true);
                        else CG_SR_PAIRS(true, false);
                    } else {
                        if (self) CG_SR_PAIRS(false, true);
                        else CG_SR_PAIRS(false, false);
                    }
#undef CG_SR_PAIRS
                }
            }
        }
        // fold the S partial sums of each receiver (lanes rl, rl + R, rl + 2R, ...) into lane rl
        {
            double tx = ax, ty = ay, tz = az;
            for (int g = 1; g < S; g++) {  // S is wave-uniform
                const int src = (lane + g * R) & 63;
                tx += __shfl(ax, src);
                ty += __shfl(ay, src);
                tz += __shfl(az, src);
            }
            ax = tx;
            ay = ty;
            az = tz;
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
