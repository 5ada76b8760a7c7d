// cg_shortrange_mfma.hip — P3M short-range sweep with a matrix-core range pre-filter (round 4).
//
//   Tiling.sort                 species.py:707-823         particle -> tile (bit-exact)
//   particle_particle           interactions.py:1563-1791  tile neighbours, periodic offset
//   gravity_pairwise_shortrange gravity.py:263-354         r2 cut, r2-indexed table, Δmom
//   subtile refinement          species.py:4031-4142, interactions.py:1141-1278 (what the
//                               reference does about the pairs that are out of range)
//
// Why another sweep.  The cells sweep (cg_shortrange.hip) evaluates every candidate pair in
// FP64: 412 pair tests per particle of which 93 are inside the range, and in a 64-lane
// wavefront a miss costs what a hit costs (rocprofv3, round 3: 24.9 VALU lane-instructions per
// test against 12 for the bare test).  The reference attacks the misses with ever finer
// subtiles.  On CDNA4 the cheaper answer is to make the misses (almost) free: the squared
// distance of every (receiver, supplier) pair of a 16 x 16 block is one single-precision
// matrix product on the matrix cores,
//       D[i][j] = [-2x_j, -2y_j, -2z_j, |u_j|^2] . [x_i, y_i, z_i, 1]  +  (|u_i|^2 - r2_pre)
//               = |u_i - u_j|^2 - r2_pre                       (v_mfma_f32_16x16x4_f32)
// with coordinates u relative to a local origin in units of the tile extent, so that the sign
// bit of D says "possibly in range" — 256 pair tests per instruction on a pipe of its own,
// beside the vector ALU.  r2_pre sits above r2_max by more than single-precision rounding can
// move a distance (bound below), so the filter never drops a pair; the pairs it lets through
// are then evaluated in FP64 exactly as before — (xi - xj) + offset, x*x + y*y + z*z, the range
// test, int(r2*scaling): every contribution is bit-identical to the cells sweep's, only the
// order of the additions differs.  No MFMA result ever reaches the momenta.
//
// Layout of the work (density-adaptive by construction: the units are COUNTS, not volumes):
//  * particles sorted by tile, z fastest (cg_shortrange_tiles), positions copied in that order;
//  * a wavefront takes 16 consecutive receivers of a tile column — wherever the tile borders
//    fall: a dense tile is many such rows, a void is one row over many tiles — four lanes per
//    receiver, each lane four supplier rows of every 16-row block;
//  * a workgroup (8 wavefronts, 128 consecutive receivers) stages the suppliers of the 3 x 3
//    neighbouring columns slab by slab in z (a slab = the 9 tiles of one z), in windows of
//    kW rows: a wavefront's suppliers — slabs tz0-1 .. tz1+1 of its receivers' tiles — are
//    one contiguous range of the staged sequence;
//  * per 512 rows: the matrix products, four sign bits per product shifted into four mask
//    registers per lane; then the lanes walk their masks independently (count leading
//    zeros -> supplier row), one FP64 pair per trip.
// Periodic images: a supplier row of an image tile carries a 6-bit code of its offset
// (-L, 0, +L per dimension), added as the reference does, (xi - xj) + offset.
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

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kW = 640;               // supplier rows staged per window
constexpr int kSlack = 64;            // rows past a window a started group of 4 blocks may read
constexpr int kRows = kW + kSlack;
constexpr int kSlabs = 14;            // z slabs per piece table
constexpr int kWaves = 8;             // wavefronts per workgroup, 16 receivers each
constexpr int kChunk = 16 * kWaves;   // receivers per workgroup and chunk
constexpr int kBatch = 512;           // rows per mask batch: 4 words x 8 blocks x 16 rows
constexpr int kSplit = 4;             // workgroups per tile column (chunks dealt round robin)

// Tiling.sort (species.py:775-780) with tiling location 0
__device__ __forceinline__ unsigned srm_tile1(double x, double inv, unsigned nt) {
    unsigned t = (unsigned)(i64)((x - 0.0) * inv);
    return t >= nt ? nt - 1 : t;
}
__device__ __forceinline__ unsigned srm_tile(const double *__restrict__ pos, i64 p, double inv,
                                             unsigned nt) {
    const unsigned i = srm_tile1(pos[3 * p + 0], inv, nt);
    const unsigned j = srm_tile1(pos[3 * p + 1], inv, nt);
    const unsigned k = srm_tile1(pos[3 * p + 2], inv, nt);
    return (i * nt + j) * nt + k;
}

// runs of equal keys inside a wavefront -> one atomic per run
__device__ __forceinline__ void srm_wave_runs(unsigned key, int lane, int &run_start,
                                              int &run_len) {
    unsigned prev = __shfl_up(key, 1);
    bool head = (lane == 0) || (key != prev);
    unsigned long long mask = __ballot(head);
    unsigned long long below = mask & (~0ull >> (63 - lane));
    run_start = 63 - __clzll(below);
    unsigned long long above = (lane == 63) ? 0ull : (mask >> (lane + 1));
    int next = above ? (lane + 1 + (__ffsll((long long)above) - 1)) : 64;
    run_len = next - run_start;
}

constexpr unsigned kNoKey = 0xffffffffu;

// (rung != null: only the particles on rungs >= lowest_active are listed — the receivers of a
// sub-step, gravity.py:318-349 through the tiles' active rungs)
__global__ __launch_bounds__(256) void k_srm_histogram(const double *__restrict__ pos, i64 n,
                                                       double inv, unsigned nt,
                                                       const signed char *__restrict__ rung,
                                                       int lowest_active,
                                                       unsigned *__restrict__ count) {
    const int lane = threadIdx.x & 63;
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned key = kNoKey;
    if (p < n && !(rung && rung[p] < lowest_active)) key = srm_tile(pos, p, inv, nt);
    int rs, rl;
    srm_wave_runs(key, lane, rs, rl);
    if (lane == rs && key != kNoKey) atomicAdd(&count[key], (unsigned)rl);
}
__global__ __launch_bounds__(256) void k_srm_scatter(const double *__restrict__ pos, i64 n,
                                                     double inv, unsigned nt,
                                                     const signed char *__restrict__ rung,
                                                     int lowest_active,
                                                     const unsigned *__restrict__ offset,
                                                     unsigned *__restrict__ cursor,
                                                     unsigned *__restrict__ order,
                                                     double *__restrict__ pos_sorted) {
    const int lane = threadIdx.x & 63;
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned key = kNoKey;
    if (p < n && !(rung && rung[p] < lowest_active)) key = srm_tile(pos, p, inv, nt);
    int rs, rl;
    srm_wave_runs(key, lane, rs, rl);
    unsigned first = 0;
    if (lane == rs && key != kNoKey) first = offset[key] + atomicAdd(&cursor[key], (unsigned)rl);
    first = __shfl(first, rs);
    if (key != kNoKey) {
        const i64 q = (i64)first + (lane - rs);
        order[q] = (unsigned)p;
        pos_sorted[3 * q] = pos[3 * p];
        pos_sorted[3 * q + 1] = pos[3 * p + 1];
        pos_sorted[3 * q + 2] = pos[3 * p + 2];
    }
}

#ifdef SRM_PROBE_COUNT  // probe build: how many trips, candidates, hits, matrix products
__device__ unsigned long long srm_dbg[8];
#define SRM_COUNT(i, v) atomicAdd(&srm_dbg[i], (unsigned long long)(v))
#else
#define SRM_COUNT(i, v) ((void)0)
#endif

struct SrmParams {
    double boxsize, ext, inv_ext, inv_tile, r2_index_scaling, r2_max, factor;
    const double *factors;          // adaptive rungs: factors[rung_jumped[i]] per receiver
    const signed char *rung_jumped;
    int nt, tablesize;
};

// inclusive scan over the 64 lanes of a wave in DPP adds
__device__ __forceinline__ unsigned srm_wave_scan(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

#ifndef SRM_LDS_TABLE
#define SRM_LDS_TABLE 1
#endif
constexpr int kTableLds = 4096;   // entries of the short-range table kept in LDS
struct SrmShared {
    double sx[kRows], sy[kRows], sz[kRows];   // staged supplier positions (FP64, as stored)
    f32x4 fa[kRows];                          // (-2u_x, -2u_y, -2u_z, |u|^2): the A operand
    unsigned char simg[kRows];                // image code of the row: (ix, iy, iz) 2 bits each
    unsigned pbeg[128], ppre[130];            // piece table: first source row, prefix of the counts
    unsigned char pimg[128];
    double ltab[4];                           // image code -> offset: -L, 0, +L
#if SRM_LDS_TABLE
    double table[kTableLds];                  // the short-range table (gravity.py:373-424)
#endif
};

// The lanes' candidates: each lane walks its own masks — bit t of a word (from the top) is row
// 16 (t / 4) + 4 g + (t % 4) of the word's 128 rows — one exact FP64 pair per trip, in the
// reference's operation order (interactions.py:1787-1789, gravity.py:299-321).
template <bool FACE>
__device__ __forceinline__ void srm_candidates(unsigned m0, unsigned m1, unsigned m2, unsigned m3,
                                               int rowb, int bend, double xi, double yi, double zi,
                                               const SrmShared &S, double r2_max,
                                               double r2_index_scaling,
                                               const double *table, double &ax,
                                               double &ay, double &az) {
    for (;;) {
        if (m0 == 0) {  // this lane's word is used up: the next one moves down
            m0 = m1;
            m1 = m2;
            m2 = m3;
            m3 = 0;
            rowb += 128;
        }
        if (!__any((m0 | m1 | m2) != 0)) break;
        const bool have = m0 != 0;
#ifdef SRM_PROBE_COUNT
        if ((threadIdx.x & 63) == 0) SRM_COUNT(0, 1);
        SRM_COUNT(1, have ? 1 : 0);
#endif
        const int t = have ? __clz((int)m0) : 0;
        m0 = have ? (m0 ^ (0x80000000u >> t)) : 0u;
        const int row = rowb + ((t & ~3) << 2) + (t & 3);
        const bool ok = have && row < bend;   // (a started block may reach past the range)
        const int rr = ok ? row : 0;
        double x_ji = xi - S.sx[rr];          // interactions.py:1787-1789
        double y_ji = yi - S.sy[rr];
        double z_ji = zi - S.sz[rr];
        if (FACE) {                           // gravity.py:299-302
            const unsigned c = S.simg[rr];
            x_ji += S.ltab[c & 3];
            y_ji += S.ltab[(c >> 2) & 3];
            z_ji += S.ltab[(c >> 4) & 3];
        }
        const double r2 = x_ji * x_ji + y_ji * y_ji + z_ji * z_ji;  // gravity.py:306
        const bool hit = ok && r2 <= r2_max;                          // gravity.py:311
#ifdef SRM_PROBE_COUNT
        SRM_COUNT(2, hit ? 1 : 0);
        SRM_COUNT(4, ok ? 1 : 0);
#endif
        double tv = 0.0;
        if (hit) tv = table[(unsigned)(int)(r2 * r2_index_scaling)];  // gravity.py:316-321
        ax = __builtin_fma(x_ji, tv, ax);
        ay = __builtin_fma(y_ji, tv, ay);
        az = __builtin_fma(z_ji, tv, az);
    }
}

__global__ __launch_bounds__(64 * kWaves) void k_sr_sweep_mfma(
    const double *__restrict__ pos_r, const unsigned *__restrict__ order_r,
    const unsigned *__restrict__ off_r, double *__restrict__ dmom_r,
    const double *__restrict__ pos_s, const unsigned *__restrict__ off_s,
    const double *__restrict__ table, SrmParams P) {
    __shared__ SrmShared S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int nt = P.nt;
    const int ta = blockIdx.z, tb = blockIdx.y;
    const unsigned col = (unsigned)(ta * nt + tb) * (unsigned)nt;
    const unsigned qc0 = __builtin_amdgcn_readfirstlane(off_r[col]),
                   qc1 = __builtin_amdgcn_readfirstlane(off_r[col + nt]);
    if (qc0 == qc1) return;
    if (tid < 4) S.ltab[tid] = tid == 0 ? -P.boxsize : (tid == 2 ? P.boxsize : 0.0);
#if SRM_LDS_TABLE
    for (int e = tid; e < P.tablesize; e += 64 * kWaves) S.table[e] = table[e];
    const double *tbl = S.table;
#else
    const double *tbl = table;
#endif
    const bool xyface = ta == 0 || ta == nt - 1 || tb == 0 || tb == nt - 1;
    const double ox = (ta + 0.5) * P.ext, oy = (tb + 0.5) * P.ext;
    const float rc2u = (float)(P.r2_max * P.inv_ext * P.inv_ext);

    for (unsigned qa = qc0 + (unsigned)kChunk * blockIdx.x; qa < qc1;
         qa += (unsigned)kChunk * gridDim.x) {
        const unsigned qb = min(qa + (unsigned)kChunk, qc1);
        // tiles of the chunk's first and last receiver (the list is sorted by tile)
        const int TZ0 = __builtin_amdgcn_readfirstlane(
                      (int)srm_tile1(pos_r[3 * (i64)qa + 2], P.inv_tile, (unsigned)nt)),
                  TZ1 = __builtin_amdgcn_readfirstlane(
                      (int)srm_tile1(pos_r[3 * (i64)(qb - 1) + 2], P.inv_tile, (unsigned)nt));
        // this wave's 16 receivers, four lanes each
        const unsigned qw = qa + 16u * (unsigned)wave;
        const bool wvalid = qw < qb;
        const bool valid = qw + (unsigned)j < qb;
        const unsigned ql = valid ? qw + (unsigned)j : qa;  // (a finite stand-in)
        const double xi = pos_r[3 * (i64)ql], yi = pos_r[3 * (i64)ql + 1],
                     zi = pos_r[3 * (i64)ql + 2];
        int tz0 = TZ0, tz1 = TZ0;
        if (wvalid) {
            const int nv = (int)min(16u, qb - qw);
            const int tzl = (int)srm_tile1(zi, P.inv_tile, (unsigned)nt);
            tz0 = __builtin_amdgcn_readlane(tzl, 0);
            tz1 = __builtin_amdgcn_readlane(tzl, nv - 1);
        }
        const bool wface = xyface || tz0 - 1 < 0 || tz1 + 1 >= nt;
        double ax = 0, ay = 0, az = 0;

        for (int zg = TZ0 - 1; zg <= TZ1 + 1; zg += kSlabs) {
            const int ns = min(kSlabs, TZ1 + 2 - zg);  // slabs zg .. zg + ns - 1
            __syncthreads();  // everybody is done with the previous table and window
            if (wave == 0) {
                // piece p = slab * 9 + column: tile (ta + p/3 % 3 - 1, tb + p % 3 - 1, zg + slab)
                unsigned beg[2], cnt[2], img[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int p = 2 * lane + e;
                    beg[e] = cnt[e] = 0;
                    img[e] = 0x15;
                    if (p < 9 * ns) {
                        const int s = p / 9, c9 = p - 9 * s;
                        int gx = ta + c9 / 3 - 1, gy = tb + c9 % 3 - 1, gz = zg + s;
                        unsigned ix = 1, iy = 1, iz = 1;
                        // periodic offset from the tile separation (interactions.py:1615-1621)
                        if (gx < 0) { gx += nt; ix = 2; } else if (gx >= nt) { gx -= nt; ix = 0; }
                        if (gy < 0) { gy += nt; iy = 2; } else if (gy >= nt) { gy -= nt; iy = 0; }
                        if (gz < 0) { gz += nt; iz = 2; } else if (gz >= nt) { gz -= nt; iz = 0; }
                        const unsigned t = ((unsigned)gx * nt + (unsigned)gy) * nt + (unsigned)gz;
                        beg[e] = off_s[t];
                        cnt[e] = off_s[t + 1] - beg[e];
                        img[e] = ix | (iy << 2) | (iz << 4);
                    }
                }
                const unsigned incl = srm_wave_scan(cnt[0] + cnt[1]);
                const unsigned excl = incl - (cnt[0] + cnt[1]);
                S.pbeg[2 * lane] = beg[0];
                S.pbeg[2 * lane + 1] = beg[1];
                S.pimg[2 * lane] = (unsigned char)img[0];
                S.pimg[2 * lane + 1] = (unsigned char)img[1];
                S.ppre[2 * lane] = excl;
                S.ppre[2 * lane + 1] = excl + cnt[0];
                if (lane == 63) S.ppre[128] = incl;
            }
            __syncthreads();
            const unsigned total = S.ppre[9 * ns];
            // single-precision coordinates: origin at the centre of the column and of the slabs,
            // unit = tile extent.  |u| <= 1.5 across, uzmax along z.  Error of D against the
            // exact |u_i - u_j|^2: the coordinates' rounding moves a distance d ~ 1 by
            // 2 sqrt(3) 2^-24 |u|max, i.e. d^2 by ~7 eps |u|max; the two norms carry 3 eps n2max
            // each and the four fused multiply-adds of the product 2 eps n2max each: below
            // eps (14 n2max + 7 |u|max).  Twice that on top of r2_max.
            const double oz = (zg + 0.5 * ns) * P.ext;
            const float uzmax = 0.5f * (float)ns + 1.5f;
            const float n2max = 8.0f + uzmax * uzmax;
            const float r2pre = rc2u + 2.0f * 5.9604645e-08f * (14.0f * n2max + 7.0f * uzmax);
            const float ux = (float)((xi - ox) * P.inv_ext), uy = (float)((yi - oy) * P.inv_ext),
                        uz = (float)((zi - oz) * P.inv_ext);
            const float cval = valid ? (ux * ux + uy * uy + uz * uz) - r2pre : 1e30f;
            const f32x4 cvec = {cval, cval, cval, cval};
            const float bq = g == 0 ? ux : (g == 1 ? uy : (g == 2 ? uz : 1.0f));
            // this wave's suppliers: slabs tz0 - 1 .. tz1 + 1, rows [RA, RB) of the sequence
            int RA = 0, RB = 0;
            if (wvalid) {
                const int sa = max(tz0 - 1, zg), sb = min(tz1 + 1, zg + ns - 1);
                if (sa <= sb) {
                    RA = (int)S.ppre[9 * (sa - zg)];
                    RB = (int)S.ppre[9 * (sb - zg + 1)];
                }
            }
            for (unsigned r0 = 0; r0 < total;) {
                unsigned r1 = min(total, r0 + (unsigned)kW);
                if (r1 < total) {  // cut at a slab boundary when one falls into the window
                    unsigned best = 0;
                    for (int s = 1; s < ns; s++) {
                        const unsigned e = S.ppre[9 * s];
                        if (e > r0 && e <= r0 + (unsigned)kW) best = e;
                    }
                    if (best) r1 = best;
                }
                if (r0) __syncthreads();  // the previous window has been consumed
                const unsigned nw = r1 - r0;
                for (unsigned w = tid; w < nw + kSlack; w += 64 * kWaves) {
                    if (w < nw) {
                        const unsigned row = r0 + w;
                        int p = 0;  // the piece of this row: ppre[p] <= row < ppre[p + 1]
#pragma unroll
                        for (int step = 64; step; step >>= 1)
                            if (S.ppre[p + step] <= row) p += step;
                        const i64 src = (i64)S.pbeg[p] + (row - S.ppre[p]);
                        const unsigned c = S.pimg[p];
                        const double xs = pos_s[3 * src], ys = pos_s[3 * src + 1],
                                     zs = pos_s[3 * src + 2];
                        S.sx[w] = xs;
                        S.sy[w] = ys;
                        S.sz[w] = zs;
                        S.simg[w] = (unsigned char)c;
                        // x_ji = (xi - xj) + offset: the image sits at xj - offset
                        const float vx = (float)(((xs - S.ltab[c & 3]) - ox) * P.inv_ext),
                                    vy = (float)(((ys - S.ltab[(c >> 2) & 3]) - oy) * P.inv_ext),
                                    vz = (float)(((zs - S.ltab[(c >> 4) & 3]) - oz) * P.inv_ext);
                        const f32x4 a4 = {-2.0f * vx, -2.0f * vy, -2.0f * vz,
                                          vx * vx + vy * vy + vz * vz};
                        S.fa[w] = a4;
                    } else {  // slack: rows that never pass the filter
                        S.sx[w] = S.sy[w] = S.sz[w] = 0.0;
                        S.simg[w] = 0x15;
                        const f32x4 a4 = {0.0f, 0.0f, 0.0f, 1e30f};
                        S.fa[w] = a4;
                    }
                }
                __syncthreads();
                const int a = max(RA, (int)r0) - (int)r0, b = min(RB, (int)r1) - (int)r0;
                const float *fa1 = (const float *)S.fa;
#ifdef SRM_PROBE_NOMFMA
                const int b_ = a;  // probe build: staging only
#else
                const int b_ = b;
#endif
                for (int ba = a; ba < b_; ba += kBatch) {
                    const int nrows = min(kBatch, b - ba);
                    const int ngrp = (nrows + 63) >> 6;  // groups of 4 blocks of 16 rows
                    unsigned m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int w = 0; w < 4; w++) {
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            if (2 * w + h < ngrp) {  // wave-uniform
#ifdef SRM_PROBE_COUNT
                                if (lane == 0) SRM_COUNT(3, 4);
#endif
                                const int rowg = ba + 64 * (2 * w + h);
                                float av[4];
#pragma unroll
                                for (int k = 0; k < 4; k++) av[k] = fa1[(rowg + 16 * k + j) * 4 + g];
                                f32x4 d[4];
#pragma unroll
                                for (int k = 0; k < 4; k++)
                                    d[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], bq, cvec, 0, 0, 0);
#pragma unroll
                                for (int k = 0; k < 4; k++) {
#pragma unroll
                                    for (int r = 0; r < 4; r++)
                                        m[w] = __builtin_amdgcn_alignbit(m[w], __float_as_uint(d[k][r]), 31);
                                }
                            } else {
                                m[w] <<= 16;
                            }
                        }
                    }
#ifdef SRM_PROBE_NOCAND  // probe build: the matrix products and masks, no pair evaluated
                    ax += (double)(m[0] ^ m[1] ^ m[2] ^ m[3]) * 1e-300;
                    continue;
#endif
                    if (wface)
                        srm_candidates<true>(m[0], m[1], m[2], m[3], ba + 4 * g, b, xi, yi, zi, S,
                                             P.r2_max, P.r2_index_scaling, tbl, ax, ay, az);
                    else
                        srm_candidates<false>(m[0], m[1], m[2], m[3], ba + 4 * g, b, xi, yi, zi, S,
                                              P.r2_max, P.r2_index_scaling, tbl, ax, ay, az);
                }
                r0 = r1;
            }
        }
        // the four lanes of a receiver: one sum, in a fixed order
        ax += __shfl_xor(ax, 16);
        ay += __shfl_xor(ay, 16);
        az += __shfl_xor(az, 16);
        ax += __shfl_xor(ax, 32);
        ay += __shfl_xor(ay, 32);
        az += __shfl_xor(az, 32);
        if (valid && g == 0) {
            const i64 pi = (i64)order_r[ql];
            // gravity.py:321 (total_factor = factors[rung] * table[...])
            const double f = P.factors ? P.factors[P.rung_jumped[pi]] : P.factor;
            dmom_r[3 * pi] += ax * f;
            dmom_r[3 * pi + 1] += ay * f;
            dmom_r[3 * pi + 2] += az * f;
        }
    }
}

}  // namespace

#ifdef SRM_PROBE_COUNT
extern "C" int cg_srm_debug_counters(unsigned long long *out, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(srm_dbg), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {};
        hipMemcpyToSymbol(HIP_SYMBOL(srm_dbg), z, sizeof(z));
    }
    return 0;
}
#endif

int cgk_shortrange_tiles(cg_ctx *c, const double *pos, i64 n, i64 nt, double tile_extent,
                         const signed char *rung, int lowest_active, unsigned *order,
                         unsigned *offset, double *pos_sorted) {
    const double eps = 2.220446049250313e-16;
    const double inv = (1 / tile_extent) * (1 - 2 * eps);
    const i64 ntiles = nt * nt * nt;
    if ((size_t)(8 * (ntiles + 1)) > c->sr_tmp_bytes) {
        CG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->sr_tmp);
        c->sr_tmp = nullptr;
        c->sr_tmp_bytes = 0;
        CG_HIP(hipMalloc(&c->sr_tmp, 8 * (ntiles + 1)));
        c->sr_tmp_bytes = 8 * (ntiles + 1);
    }
    unsigned *count = (unsigned *)c->sr_tmp, *cursor = count + (ntiles + 1);
    CG_HIP(hipMemsetAsync(c->sr_tmp, 0, 8 * (ntiles + 1), c->stream));
    const i64 blocks = (n + 255) / 256;
    if (n > 0) {
        hipLaunchKernelGGL(k_srm_histogram, dim3((unsigned)blocks), dim3(256), 0, c->stream, pos, n,
                           inv, (unsigned)nt, rung, lowest_active, count);
        CG_LAUNCH_CHECK();
    }
    size_t need = 0;
    CG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, count, offset, (int)(ntiles + 1),
                                            c->stream));
    if (need > c->scan_tmp_bytes) {
        CG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->scan_tmp);
        c->scan_tmp = nullptr;
        c->scan_tmp_bytes = 0;
        CG_HIP(hipMalloc(&c->scan_tmp, need));
        c->scan_tmp_bytes = need;
    }
    CG_HIP(hipcub::DeviceScan::ExclusiveSum(c->scan_tmp, need, count, offset, (int)(ntiles + 1),
                                            c->stream));
    if (n > 0) {
        hipLaunchKernelGGL(k_srm_scatter, dim3((unsigned)blocks), dim3(256), 0, c->stream, pos, n,
                           inv, (unsigned)nt, rung, lowest_active, offset, cursor, order,
                           pos_sorted);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

int cgk_shortrange_sweep_tiles(cg_ctx *c, const double *pos_r_sorted, const unsigned *order_r,
                               const unsigned *off_r, double *dmom_r, const double *pos_s_sorted,
                               const unsigned *off_s, i64 nt, const double *table,
                               int64_t tablesize, double r2_index_scaling, double r2_max,
                               double factor, const double *factors,
                               const signed char *rung_jumped) {
    const double eps = 2.220446049250313e-16;
    const double ext = c->p.boxsize / (double)nt;  // species.py:607-609
    SrmParams P{c->p.boxsize, ext, 1.0 / ext, (1 / ext) * (1 - 2 * eps), r2_index_scaling, r2_max,
                factor, factors, rung_jumped, (int)nt, (int)tablesize};
#if SRM_LDS_TABLE
    CG_CHECK(tablesize <= kTableLds, "cg_shortrange_sweep_tiles: tables of up to %d entries", kTableLds);
#endif
    hipLaunchKernelGGL(k_sr_sweep_mfma, dim3(kSplit, (unsigned)nt, (unsigned)nt),
                       dim3(64 * kWaves), 0, c->stream, pos_r_sorted, order_r, off_r, dmom_r,
                       pos_s_sorted, off_s, table, P);
    CG_LAUNCH_CHECK();
    return 0;
}
