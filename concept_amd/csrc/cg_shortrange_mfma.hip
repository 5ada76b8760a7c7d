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
//       D[i][j] = [-2x_j, -2y_j, -2z_j, |v_j|^2] . [x_i, y_i, z_i, 1]  +  (|u_i|^2 - r2_pre)
//               = |u_i - v_j|^2 - r2_pre                       (v_mfma_f32_16x16x4_f32)
// so that the sign bit of D says "possibly in range" — 256 pair tests per instruction on a
// pipe of its own, beside the vector ALU.  r2_pre sits above r2_max by more than
// single-precision rounding can move a distance (bound below), so the filter never drops a
// pair; the pairs it lets through (measured: hits + 0.1 %) are then evaluated in FP64 exactly
// as before — (xi - xj) + offset, x*x + y*y + z*z, the range test, int(r2*scaling): every
// contribution is bit-identical to the cells sweep's, only the order of the additions differs.
// No MFMA result ever reaches the momenta.
//
// Coordinates of the products.  A supplier's row of the A operand, (-2v, |v|^2), is written
// once, when the list is built: v = position / tile extent - (tile x, tile y, first tile of
// its block of kZB tiles in z) — a local coordinate that does not depend on who asks.  The
// receiver side u is the receiver's position relative to that same corner, unwrapped across
// the box faces: a periodic image costs a different corner, not a different operand.
//
// Layout of the work (density-adaptive by construction: the units are COUNTS, not volumes):
//  * particles listed by tile, z fastest (cg_shortrange_tiles), positions copied in that order;
//  * a wavefront takes 16 consecutive receivers of a tile column — wherever the tile borders
//    fall: a dense tile is many such rows, a void is one row over many tiles — four lanes per
//    receiver, each lane four supplier rows of every 16-row block;
//  * a workgroup (8 wavefronts, 128 consecutive receivers) copies the suppliers its receivers
//    can reach — the 3 x 3 neighbouring columns over the tiles TZ0-1 .. TZ1+1, nine
//    contiguous runs of the list — into LDS once (FP64 positions only: they are what the
//    candidates read at random), then its wavefronts work on their own: no barrier until the
//    next 128 receivers;
//  * a wavefront walks the nine runs over its own tiles tz0-1 .. tz1+1: operand rows straight
//    from the list (coalesced 256-byte reads, L2), one product per 16 rows, four sign bits per
//    product shifted into mask words; then the lanes walk their masks independently (count
//    leading zeros -> block -> row), one FP64 pair per trip, software-pipelined over three
//    trips (positions, table entry, accumulation).
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

#ifndef SRM_LDS_TABLE
#define SRM_LDS_TABLE 0               // 1: the short-range table in LDS (32 KB per workgroup)
#endif
#ifndef SRM_WAVES
#define SRM_WAVES 8
#endif
#ifndef SRM_CAP
#define SRM_CAP (SRM_LDS_TABLE ? 1792 : 2016)
#endif
constexpr int kZB = 16;               // tiles per z block of the operand coordinates
constexpr int kCap = SRM_CAP;         // supplier rows a workgroup holds in LDS (< 2048)
constexpr int kSlabs = 32;            // z slabs (tiles along z) per piece table
constexpr int kPieces = 64;           // pieces: 9 columns x runs of slabs
constexpr int kWaves = SRM_WAVES;     // wavefronts per workgroup, 16 receivers each
constexpr int kChunk = 16 * kWaves;   // receivers per workgroup and chunk
constexpr int kWords = 6;             // mask words per batch: 6 x 8 blocks x 16 rows
constexpr int kSplit = 4;             // workgroups per tile column (chunks dealt round robin)
[[maybe_unused]] constexpr int kTableLds = 4096;  // entries of the short-range table kept in LDS
static_assert(kCap < 2048 && kWords * 8 <= 64 && 9 * 7 <= kPieces, "packing of the block table");

// Tiling.sort (species.py:775-780) with tiling location 0
__device__ __forceinline__ unsigned srm_tile1(double x, double inv, unsigned nt) {
    unsigned t = (unsigned)(i64)((x - 0.0) * inv);
    return t >= nt ? nt - 1 : t;
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

struct SrmKey {
    unsigned key, gx, gy, gz;
    double x, y, z;
};
// (rung != null: only the particles on rungs >= lowest_active are listed — the receivers of a
// sub-step, gravity.py:318-349 through the tiles' active rungs)
__device__ __forceinline__ SrmKey srm_key(const double *__restrict__ pos, i64 p, i64 n, double inv,
                                          unsigned nt, const signed char *__restrict__ rung,
                                          int lowest_active) {
    SrmKey k;
    k.key = kNoKey;
    k.gx = k.gy = k.gz = 0;
    k.x = k.y = k.z = 0;
    if (p < n && !(rung && rung[p] < lowest_active)) {
        k.x = pos[3 * p], k.y = pos[3 * p + 1], k.z = pos[3 * p + 2];
        k.gx = srm_tile1(k.x, inv, nt);
        k.gy = srm_tile1(k.y, inv, nt);
        k.gz = srm_tile1(k.z, inv, nt);
        k.key = (k.gx * nt + k.gy) * nt + k.gz;
    }
    return k;
}
__global__ __launch_bounds__(256) void k_srm_histogram(const double *__restrict__ pos, i64 n,
                                                       double inv, unsigned nt,
                                                       const signed char *__restrict__ rung,
                                                       int lowest_active,
                                                       unsigned *__restrict__ count) {
    const int lane = threadIdx.x & 63;
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const SrmKey k = srm_key(pos, p, n, inv, nt, rung, lowest_active);
    int rs, rl;
    srm_wave_runs(k.key, lane, rs, rl);
    if (lane == rs && k.key != kNoKey) atomicAdd(&count[k.key], (unsigned)rl);
}
__global__ __launch_bounds__(256) void k_srm_scatter(const double *__restrict__ pos, i64 n,
                                                     double inv, double inv_ext, unsigned nt,
                                                     const signed char *__restrict__ rung,
                                                     int lowest_active,
                                                     const unsigned *__restrict__ offset,
                                                     unsigned *__restrict__ cursor,
                                                     unsigned *__restrict__ order,
                                                     double *__restrict__ pos_sorted,
                                                     f32x4 *__restrict__ aop) {
    const int lane = threadIdx.x & 63;
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const SrmKey k = srm_key(pos, p, n, inv, nt, rung, lowest_active);
    int rs, rl;
    srm_wave_runs(k.key, lane, rs, rl);
    unsigned first = 0;
    if (lane == rs && k.key != kNoKey)
        first = offset[k.key] + atomicAdd(&cursor[k.key], (unsigned)rl);
    first = __shfl(first, rs);
    if (k.key != kNoKey) {
        const i64 q = (i64)first + (lane - rs);
        order[q] = (unsigned)p;
        pos_sorted[3 * q] = k.x;
        pos_sorted[3 * q + 1] = k.y;
        pos_sorted[3 * q + 2] = k.z;
        if (aop) {
            // the A operand of the range products: coordinates relative to the corner of the
            // particle's own tile (z: of its block of kZB tiles), unit = tile extent
            const float vx = (float)(k.x * inv_ext - (double)k.gx),
                        vy = (float)(k.y * inv_ext - (double)k.gy),
                        vz = (float)(k.z * inv_ext - (double)(k.gz - k.gz % kZB));
            const f32x4 a4 = {-2.0f * vx, -2.0f * vy, -2.0f * vz, vx * vx + vy * vy + vz * vz};
            aop[q] = a4;
        }
    }
}

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

// ---------------------------------------------------------------------------
// Density-adaptive order inside the tiles (the counterpart of the reference's automatic
// subtile refinement, species.py:4031-4142: subtiles fine enough to hold a handful of
// particles each).  A tile with many particles has its rows of the list re-ordered by
// sub-cell — 8^3 sub-cells along a Hilbert curve — so that
// 16 consecutive rows (a wavefront's receivers; a block of suppliers) are neighbours in space
// wherever the particles are many: the receivers of a wavefront then see the same suppliers
// (their lanes finish together), and a supplier block has a bounding box worth testing.
// Sparse tiles (fewer than kSubMin rows) stay as the tile sort left them.
// ---------------------------------------------------------------------------
constexpr int kSubMin = 48;      // rows from which a tile is re-ordered

__global__ __launch_bounds__(256) void k_srm_dense_tiles(const unsigned *__restrict__ offset,
                                                         unsigned ntiles,
                                                         unsigned *__restrict__ ndense,
                                                         unsigned *__restrict__ dense) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    if (offset[t + 1] - offset[t] >= (unsigned)kSubMin) dense[atomicAdd(ndense, 1u)] = t;
}

// sub-cell (a, b, c) of an 8 x 8 x 8 division of the tile -> its place on a Hilbert curve
// through the 512 sub-cells (consecutive places are face neighbours: a run of rows is a compact
// blob wherever it starts, which a Morton curve's jumps do not give).  Filled once by the host
// (srm_hilbert_table).
__device__ unsigned short srm_hilbert[512];
constexpr int kSubCells = 8;

__device__ __forceinline__ unsigned srm_subkey(double x, double y, double z, double inv_ext,
                                               unsigned gx, unsigned gy, unsigned gz) {
    const int a = min(kSubCells - 1, max(0, (int)((x * inv_ext - (double)gx) * kSubCells))),
              b = min(kSubCells - 1, max(0, (int)((y * inv_ext - (double)gy) * kSubCells))),
              c = min(kSubCells - 1, max(0, (int)((z * inv_ext - (double)gz) * kSubCells)));
    return srm_hilbert[(a * kSubCells + b) * kSubCells + c];
}

struct SrmRow {  // what travels with a row of the list
    double x, y, z;
    f32x4 a;
    unsigned order;
    unsigned pad;
};

__global__ __launch_bounds__(256) void k_srm_subsort(const unsigned *__restrict__ offset,
                                                     const unsigned *__restrict__ ndense,
                                                     const unsigned *__restrict__ dense,
                                                     double inv_ext, unsigned nt,
                                                     unsigned *__restrict__ order,
                                                     double *__restrict__ pos_sorted,
                                                     f32x4 *__restrict__ aop,
                                                     SrmRow *__restrict__ scratch) {
    __shared__ unsigned hist[512], base[512];
    if (blockIdx.x >= *ndense) return;
    const unsigned t = dense[blockIdx.x];
    const unsigned b = offset[t], e = offset[t + 1];
    const unsigned gz = t % nt, gy = (t / nt) % nt, gx = t / (nt * nt);
    for (int i = threadIdx.x; i < 512; i += 256) hist[i] = 0;
    __syncthreads();
    for (unsigned q = b + threadIdx.x; q < e; q += 256)
        atomicAdd(&hist[srm_subkey(pos_sorted[3 * (i64)q], pos_sorted[3 * (i64)q + 1],
                                   pos_sorted[3 * (i64)q + 2], inv_ext, gx, gy, gz)], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of the 512 counts: 8 per lane of one wave
        unsigned v[8], sum = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            v[i] = hist[8 * threadIdx.x + i];
            sum += v[i];
        }
        unsigned run = srm_wave_scan(sum) - sum;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            base[8 * threadIdx.x + i] = run;
            run += v[i];
        }
    }
    __syncthreads();
    for (unsigned q = b + threadIdx.x; q < e; q += 256) {
        SrmRow r;
        r.x = pos_sorted[3 * (i64)q], r.y = pos_sorted[3 * (i64)q + 1],
        r.z = pos_sorted[3 * (i64)q + 2];
        r.order = order[q];
        r.pad = 0;
        if (aop) r.a = aop[q];
        const unsigned at = atomicAdd(&base[srm_subkey(r.x, r.y, r.z, inv_ext, gx, gy, gz)],
                                      1u);
        scratch[(i64)b + at] = r;
    }
    __syncthreads();  // (the workgroup's own writes to global memory, read back by itself)
    __threadfence_block();
    for (unsigned q = b + threadIdx.x; q < e; q += 256) {
        const SrmRow r = scratch[q];
        pos_sorted[3 * (i64)q] = r.x;
        pos_sorted[3 * (i64)q + 1] = r.y;
        pos_sorted[3 * (i64)q + 2] = r.z;
        order[q] = r.order;
        if (aop) aop[q] = r.a;
    }
}

// Bounding box of every block of 16 consecutive rows of the list (global blocks: rows 16 b ..
// 16 b + 15), in units of the tile extent, kept behind the operand rows: the sweep does not
// form the products of a block none of whose rows can be in range of any of a wavefront's
// receivers — the minimum-distance test of the reference's subtile pairs
// (interactions.py:1236-1251) at the granularity the sub-cell order above makes worthwhile.
__global__ __launch_bounds__(256) void k_srm_bbox(const double *__restrict__ pos_sorted,
                                                  const unsigned *__restrict__ offset,
                                                  unsigned ntiles, double inv_ext,
                                                  f32x4 *__restrict__ bb) {
    const i64 n = offset[ntiles];
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 gb = q >> 4;
    if (16 * gb >= n) return;
    const i64 qq = q < n ? q : 16 * gb;  // (past the end: the block's first row again)
    float lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; d++) lo[d] = hi[d] = (float)(pos_sorted[3 * qq + d] * inv_ext);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], m));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], m));
        }
    }
    if ((threadIdx.x & 15) == 0) {
        bb[2 * gb] = f32x4{lo[0], lo[1], lo[2], 0.0f};
        bb[2 * gb + 1] = f32x4{hi[0], hi[1], hi[2], 0.0f};
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
    i64 n_s;                        // rows of the supplier list
    int nt, tablesize;
};

struct SrmShared {
    double sx[kCap], sy[kCap], sz[kCap];    // the suppliers' positions (FP64, as stored)
    unsigned char simg[kCap];               // image code of the row: (ix, iy, iz) 2 bits each
    unsigned tstart[9][kSlabs], tend[9][kSlabs];  // list rows of tile (column c, slab s)
    // pieces: column c9 over the slabs [ps0, ps1) — a run of the list without a box face or a
    // z-block border inside — in the order (c9, slab)
    unsigned pbeg[kPieces], ppre[kPieces + 1];
    short ps0[kPieces], ps1[kPieces], pgx[kPieces], pgy[kPieces], pzb[kPieces];
    unsigned char pimg[kPieces];
    int nseg;                               // pieces per column
    double ltab[4];                         // image code -> offset: -L, 0, +L
#if SRM_LDS_TABLE
    double table[kTableLds];                // the short-range table (gravity.py:373-424)
#endif
};

// The lanes' candidates.  Each lane walks its own masks, one exact FP64 pair per trip in the
// reference's operation order (interactions.py:1787-1789, gravity.py:299-321), as a software
// pipeline of four stages over four trips, so that no trip waits for what it asked for itself:
//   S1  next set bit of the masks -> block; ask lane `block` for the block's entry (bpermute)
//   S2  entry -> LDS row; ask for the supplier's position
//   S3  x_ji, r2, range test, table index; ask for the table entry
//   S4  the three multiply-adds
// The loop body is two trips with the two register sets swapped, so that nothing loaded is
// ever copied (a copy would wait for it).
struct SrmMasks {
    unsigned m[kWords];
    int wc;  // first block of word m[0]
};
struct SrmBit {   // S1 -> S2
    bool have;
    unsigned r;   // row of the block: 4 g + t % 4
    unsigned e;   // the block's entry: LDS row of its row 0 + 16 | first row << 12 | rows << 17
};
struct SrmPos {   // S2 -> S3
    bool ok;
    double sx, sy, sz;
    unsigned img;
};
struct SrmEv {    // S3 -> S4
    double x, y, z, t;
    bool hit;
};
[[maybe_unused]] __device__ __forceinline__ bool srm_left(const SrmMasks &M) {
    unsigned any = 0;
#pragma unroll
    for (int w = 0; w < kWords; w++) any |= M.m[w];
    return any != 0;
}
template <bool FACE>
__device__ __forceinline__ void srm_trip(SrmMasks &M, unsigned btab, int g, double xi, double yi,
                                         double zi, const SrmShared &S, double boxsize,
                                         double r2_max, double r2_index_scaling,
                                         const double *table, const SrmBit &bin, SrmBit &bout,
                                         const SrmPos &pin, SrmPos &pout, const SrmEv &ein,
                                         SrmEv &eout, double &ax, double &ay, double &az) {
    // S4 (a miss has read entry 0 of the table: it counts for nothing)
    {
        const double t = ein.hit ? ein.t : 0.0;
        ax = __builtin_fma(ein.x, t, ax);
        ay = __builtin_fma(ein.y, t, ay);
        az = __builtin_fma(ein.z, t, az);
    }
    // S3
    {
        double x_ji = xi - pin.sx;            // interactions.py:1787-1789
        double y_ji = yi - pin.sy;
        double z_ji = zi - pin.sz;
        if (FACE) {                           // gravity.py:299-302: + the image's offset,
            // -L, 0 or +L from the row's code (a product with -1, 0 or 1 is exact)
            x_ji += (double)((int)(pin.img & 3) - 1) * boxsize;
            y_ji += (double)((int)((pin.img >> 2) & 3) - 1) * boxsize;
            z_ji += (double)((int)((pin.img >> 4) & 3) - 1) * boxsize;
        }
        const double r2 = x_ji * x_ji + y_ji * y_ji + z_ji * z_ji;  // gravity.py:306
        const bool hit = pin.ok && r2 <= r2_max;                      // gravity.py:311
#ifdef SRM_PROBE_COUNT
        SRM_COUNT(1, pin.ok ? 1 : 0);
        SRM_COUNT(2, hit ? 1 : 0);
        if ((threadIdx.x & 63) == 0) SRM_COUNT(0, 1);
#endif
        const unsigned idx = hit ? (unsigned)(int)(r2 * r2_index_scaling) : 0u;  // gravity.py:316
        eout.x = x_ji, eout.y = y_ji, eout.z = z_ji;
        eout.t = table[idx];                                                     // gravity.py:321
        eout.hit = hit;
    }
    // S2
    {
        pout.ok = bin.have && bin.r - ((bin.e >> 12) & 31u) < (bin.e >> 17);
        const unsigned row = pout.ok ? (bin.e & 0xfffu) + bin.r - 16u : 0u;
        pout.sx = S.sx[row];
        pout.sy = S.sy[row];
        pout.sz = S.sz[row];
        pout.img = FACE ? S.simg[row] : 0x15u;
    }
    // S1
    {
        if (M.m[0] == 0) {  // this lane's word is used up: the next ones move down
#pragma unroll
            for (int w = 0; w + 1 < kWords; w++) M.m[w] = M.m[w + 1];
            M.m[kWords - 1] = 0;
            M.wc += 8;
        }
        bout.have = M.m[0] != 0;
        const int t = bout.have ? __clz((int)M.m[0]) : 0;
        M.m[0] = bout.have ? (M.m[0] ^ (0x80000000u >> t)) : 0u;
        // bit t of a word (from the top) = block t / 4 of the word, row 4 g + t % 4 of it
        const int blk = (M.wc + (t >> 2)) & 63;
        bout.e = (unsigned)__builtin_amdgcn_ds_bpermute(4 * blk, (int)btab);
        bout.r = 4u * (unsigned)g + (unsigned)(t & 3);
    }
}
template <bool FACE>
__device__ __forceinline__ void srm_candidates(SrmMasks M, unsigned btab, int g, double xi,
                                               double yi, double zi, const SrmShared &S,
                                               double boxsize, double r2_max,
                                               double r2_index_scaling, const double *table,
                                               double &ax, double &ay, double &az) {
    SrmBit b0 = {false, 0, 0}, b1 = {false, 0, 0};
    SrmPos p0 = {false, 0, 0, 0, 0x15u}, p1 = {false, 0, 0, 0, 0x15u};
    SrmEv e0 = {0, 0, 0, 0, false}, e1 = {0, 0, 0, 0, false};
    // How many trips the slowest lane needs: one per set bit, plus one for every empty word
    // with a set bit somewhere behind it (a trip that only moves the words down) — the
    // maximum over the wave, once, instead of a look at every lane's masks in every trip.
    int need = 0, behind = 0;
#pragma unroll
    for (int w = kWords - 1; w >= 0; w--) {
        need += __popc(M.m[w]) + (M.m[w] == 0 ? behind : 0);
        behind |= M.m[w] != 0 ? 1 : 0;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) need = max(need, __shfl_xor(need, d));
    const int trips = __builtin_amdgcn_readfirstlane(need) + 3;  // + what is under way at the end
    for (int trip = 0; trip < trips; trip += 2) {
        srm_trip<FACE>(M, btab, g, xi, yi, zi, S, boxsize, r2_max, r2_index_scaling, table, b0, b1,
                       p0, p1, e0, e1, ax, ay, az);
        srm_trip<FACE>(M, btab, g, xi, yi, zi, S, boxsize, r2_max, r2_index_scaling, table, b1, b0,
                       p1, p0, e1, e0, ax, ay, az);
    }
}

#ifndef SRM_WAVES_PER_EU
#define SRM_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(64 * kWaves)
__attribute__((amdgpu_waves_per_eu(SRM_WAVES_PER_EU, 8))) void k_sr_sweep_mfma(
    const double *__restrict__ pos_r, const unsigned *__restrict__ order_r,
    const unsigned *__restrict__ off_r, double *__restrict__ dmom_r,
    const double *__restrict__ pos_s, const unsigned *__restrict__ off_s,
    const f32x4 *__restrict__ aop_s, const double *__restrict__ table, SrmParams P) {
    __shared__ SrmShared S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int lane_off = 4 * j + g;  // this lane's element of a 16-row block of operand rows
    const int nt = P.nt;
    const int ta = blockIdx.z, tb = blockIdx.y;
    const unsigned col = (unsigned)(ta * nt + tb) * (unsigned)nt;
    const unsigned qc0 = __builtin_amdgcn_readfirstlane(off_r[col]),
                   qc1 = __builtin_amdgcn_readfirstlane(off_r[col + nt]);
    if (qc0 + (unsigned)kChunk * blockIdx.x >= qc1) return;
    if (tid < 4) S.ltab[tid] = tid == 0 ? -P.boxsize : (tid == 2 ? P.boxsize : 0.0);
#if SRM_LDS_TABLE
    for (int e = tid; e < P.tablesize; e += 64 * kWaves) S.table[e] = table[e];
    const double *tbl = S.table;
#else
    const double *tbl = table;
#endif
    const bool xyface = ta == 0 || ta == nt - 1 || tb == 0 || tb == nt - 1;
    const float *aop1 = (const float *)aop_s;
    const f32x4 *bbs = aop_s + (P.n_s + 16);   // the blocks' bounding boxes, behind the operand rows
    // (single-precision coordinates of up to nt tiles: 2^-24 nt each, twice, on a distance ~ 1)
    const float r2cull = (float)(P.r2_max * P.inv_ext * P.inv_ext) * 1.0001f + 4.0e-6f * (float)P.nt;
    // The filter's threshold.  |u - v| <= kZB + 2 along z and 3 across; error of D against the
    // exact |u - v|^2: the coordinates' rounding moves a distance d ~ 1 by 2 sqrt(3) 2^-24
    // |u|max, i.e. d^2 by ~7 eps |u|max; the two norms carry 3 eps n2max each and the four fused
    // multiply-adds of the product 2 eps n2max each: below eps (14 n2max + 7 |u|max).  Twice
    // that on top of r2_max.
    const float umax = (float)(kZB + 2), n2max = umax * umax + 8.0f;
    const float r2pre = (float)(P.r2_max * P.inv_ext * P.inv_ext) +
                        2.0f * 5.9604645e-08f * (14.0f * n2max + 7.0f * (umax + kSlabs + kZB));

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
        // the box of this wave's receivers, in tiles (wave-uniform)
        float rlo[3], rhi[3];
        {
            const double pr[3] = {xi, yi, zi};
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const float v = (float)(pr[d] * P.inv_ext);
                rlo[d] = valid ? v : 3.0e38f;
                rhi[d] = valid ? v : -3.0e38f;
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    rlo[d] = fminf(rlo[d], __shfl_xor(rlo[d], m));
                    rhi[d] = fmaxf(rhi[d], __shfl_xor(rhi[d], m));
                }
            }
        }
        // the receiver in units of the tile extent (FP64: the corners are subtracted exactly)
        // and in single precision relative to the column's corner at the group's first slab: a
        // piece's corner is a whole number of tiles from there (one rounding more, below the
        // margin of r2_pre)
        const float fx = (float)(xi * P.inv_ext - (double)ta), fy = (float)(yi * P.inv_ext - (double)tb);
        double ax = 0, ay = 0, az = 0;

        for (int zg = TZ0 - 1; zg <= TZ1 + 1; zg += kSlabs) {
            const int ns = min(kSlabs, TZ1 + 2 - zg);  // slabs zg .. zg + ns - 1
            const float fz = (float)(zi * P.inv_ext - (double)zg);
            __syncthreads();  // everybody is done with the previous tables and rows
            // list rows of every tile (column c9, slab s) of this group
            for (int e = tid; e < 9 * ns; e += 64 * kWaves) {
                const int c9 = e / ns, s = e - c9 * ns;
                int gx = ta + c9 / 3 - 1, gy = tb + c9 % 3 - 1, gz = zg + s;
                gx = gx < 0 ? gx + nt : (gx >= nt ? gx - nt : gx);
                gy = gy < 0 ? gy + nt : (gy >= nt ? gy - nt : gy);
                gz = gz < 0 ? gz + nt : (gz >= nt ? gz - nt : gz);
                const unsigned t = ((unsigned)gx * nt + (unsigned)gy) * nt + (unsigned)gz;
                S.tstart[c9][s] = off_s[t];
                S.tend[c9][s] = off_s[t + 1];
            }
            __syncthreads();
            if (wave == 0) {
                // a piece starts at slab 0 of the group, below and above the box (a periodic
                // image: another offset) and at the first tile of every z block (another
                // corner of the operand coordinates)
                unsigned flags = 0;
                for (int s = 0; s < ns; s++) {
                    const int z = zg + s, gz = z < 0 ? z + nt : (z >= nt ? z - nt : z);
                    if (s == 0 || z == 0 || z == nt || gz % kZB == 0) flags |= 1u << s;
                }
                const int nseg = __popc(flags);
                unsigned beg = 0, cnt = 0;
                if (lane < 9 * nseg) {
                    const int c9 = lane / nseg, k = lane - c9 * nseg;
                    unsigned f = flags;
                    for (int i = 0; i < k; i++) f &= f - 1;
                    const int s0 = __ffs((int)f) - 1;
                    f &= f - 1;
                    const int s1 = f ? __ffs((int)f) - 1 : ns;
                    const int gx = ta + c9 / 3 - 1, gy = tb + c9 % 3 - 1, z0 = zg + s0;
                    // periodic offset from the tile separation (interactions.py:1615-1621)
                    const unsigned ix = gx < 0 ? 2 : (gx >= nt ? 0 : 1),
                                   iy = gy < 0 ? 2 : (gy >= nt ? 0 : 1),
                                   iz = z0 < 0 ? 2 : (z0 >= nt ? 0 : 1);
                    const int gz0 = z0 < 0 ? z0 + nt : (z0 >= nt ? z0 - nt : z0);
                    beg = S.tstart[c9][s0];               // (the tiles of a piece follow each
                    cnt = S.tend[c9][s1 - 1] - beg;       // other in the list)
                    S.ps0[lane] = (short)s0;
                    S.ps1[lane] = (short)s1;
                    S.pgx[lane] = (short)gx;
                    S.pgy[lane] = (short)gy;
                    S.pzb[lane] = (short)(z0 - gz0 % kZB);  // unwrapped first tile of the z block
                    S.pimg[lane] = (unsigned char)(ix | (iy << 2) | (iz << 4));
                }
                const unsigned incl = srm_wave_scan(cnt);
                S.pbeg[lane] = beg;
                S.ppre[lane] = incl - cnt;
                if (lane == 63) S.ppre[64] = incl;
                if (lane == 0) S.nseg = nseg;
            }
            __syncthreads();
            const int nseg = S.nseg, np = 9 * nseg;
            const unsigned total = S.ppre[np];

            for (unsigned r0 = 0; r0 < total; r0 += kCap) {
                const unsigned r1 = min(total, r0 + (unsigned)kCap);
                if (r0) __syncthreads();  // the previous window has been consumed
                // copy rows [r0, r1) of the sequence: the column first, then the piece in it
                for (unsigned w = tid; w < r1 - r0; w += 64 * kWaves) {
                    const unsigned row = r0 + w;
                    int c9 = 0;
#pragma unroll
                    for (int c = 1; c < 9; c++) c9 += S.ppre[c * nseg] <= row ? 1 : 0;
                    int p = c9 * nseg;
                    for (int k = 1; k < nseg; k++) p += S.ppre[c9 * nseg + k] <= row ? 1 : 0;
                    const i64 src = (i64)S.pbeg[p] + (row - S.ppre[p]);
                    S.sx[w] = pos_s[3 * src];
                    S.sy[w] = pos_s[3 * src + 1];
                    S.sz[w] = pos_s[3 * src + 2];
                    S.simg[w] = S.pimg[p];
                }
                __syncthreads();
#ifdef SRM_PROBE_NOMFMA
                continue;  // probe build: the copies only
#endif
                if (!wvalid) continue;
                // ---- this wave: products over its own rows of every piece, then the pairs ----
                // (A) lane p: the wave's rows of piece p — slabs tz0 - 1 .. tz1 + 1 — as LDS rows
                // [pa, pb_) of this window, the list row of the first (pg0) and the blocks of
                // 16 list rows they lie in: GLOBAL blocks (rows 16 b .. 16 b + 15 of the list,
                // the ones that have a bounding box), of which the first and the last may reach
                // past the range
                int pa = 0, pb_ = 0, pnblk = 0;
                unsigned pg0 = 0;
                float pdx = 0, pdy = 0, pdz = 0;  // the piece's corner relative to (ta, tb, zg)
                float psx = 0, psy = 0, psz = 0;  // the periodic image, in tiles: -nt, 0, +nt
                if (lane < np) {
                    const int c9 = lane / nseg;
                    const int lo = max((int)S.ps0[lane], tz0 - 1 - zg),
                              hi = min((int)S.ps1[lane] - 1, tz1 + 1 - zg);
                    if (lo <= hi) {
                        const unsigned pb = S.pbeg[lane], pp = S.ppre[lane];
                        const i64 fa = (i64)pp + (S.tstart[c9][lo] - pb),
                                  fb = (i64)pp + (S.tend[c9][hi] - pb);
                        const int a = (int)(max(fa, (i64)r0) - (i64)r0),
                                  b = (int)(min(fb, (i64)r1) - (i64)r0);
                        if (b > a) {
                            pa = a, pb_ = b;
                            pg0 = pb + (unsigned)((i64)a + (i64)r0 - (i64)pp);
                            pnblk = (int)(((pg0 + (unsigned)(b - a) + 15u) >> 4) - (pg0 >> 4));
                        }
                    }
                    pdx = (float)((int)S.pgx[lane] - ta);
                    pdy = (float)((int)S.pgy[lane] - tb);
                    pdz = (float)((int)S.pzb[lane] - zg);
                    const unsigned im = S.pimg[lane];
                    psx = (float)(((int)(im & 3) - 1) * nt);
                    psy = (float)(((int)((im >> 2) & 3) - 1) * nt);
                    psz = (float)(((int)((im >> 4) & 3) - 1) * nt);
                }
                const unsigned pincl = srm_wave_scan((unsigned)pnblk);
                const int pstart = (int)pincl - pnblk;
                const int NB = __builtin_amdgcn_readlane((int)pincl, 63);  // the wave's blocks
                for (int B0 = 0; B0 < NB; B0 += 8 * kWords) {
                    // (B) lane b: block B0 + b of the wave's list of blocks -> its piece (first
                    // lane whose inclusive count exceeds it), LDS rows, list rows; then the
                    // blocks whose bounding box no receiver of the wave can reach are dropped
                    // and the others move up
                    const int fblk = B0 + lane;
                    const bool isblk = lane < 8 * kWords && fblk < NB;
                    int pid = 0;
#pragma unroll
                    for (int step = 32; step; step >>= 1) {
                        const int v = __builtin_amdgcn_ds_bpermute(4 * (pid + step - 1), (int)pincl);
                        pid += v <= fblk ? step : 0;
                    }
                    pid = min(pid, 63);
                    const int qa_ = __builtin_amdgcn_ds_bpermute(4 * pid, pa),
                              qb_ = __builtin_amdgcn_ds_bpermute(4 * pid, pb_),
                              qs_ = __builtin_amdgcn_ds_bpermute(4 * pid, pstart);
                    const unsigned qg_ = (unsigned)__builtin_amdgcn_ds_bpermute(4 * pid, (int)pg0);
                    // (every lane takes part in a bpermute: a lane that is switched off reads as 0)
                    const float sx_ = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * pid, __builtin_bit_cast(int, psx))),
                                sy_ = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * pid, __builtin_bit_cast(int, psy))),
                                sz_ = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * pid, __builtin_bit_cast(int, psz)));
                    unsigned btab = 0, grow = 0;
                    bool keep = false;
                    if (isblk) {
                        const int k = fblk - qs_, off = (int)(qg_ & 15u);
                        const int base = qa_ - off + 16 * k;        // LDS row of the block's row 0
                        const int first = k == 0 ? off : 0,
                                  last = min(16, qb_ - base);       // its rows inside the range
                        // entry: LDS row of row 0 (+ 16: it may lie before the window) |
                        // first row << 12 | number of rows << 17
                        btab = (unsigned)(base + 16) | ((unsigned)first << 12) |
                               ((unsigned)(last - first) << 17);
                        grow = (qg_ & ~15u) + 16u * (unsigned)k;
                        // minimum distance of the receivers' box to the block's (both in
                        // tiles; the receivers seen from the supplier's periodic image)
                        const f32x4 blo = bbs[2 * (i64)(grow >> 4)], bhi = bbs[2 * (i64)(grow >> 4) + 1];
                        const float gx_ = fmaxf(fmaxf(blo[0] - (rhi[0] + sx_), (rlo[0] + sx_) - bhi[0]), 0.0f),
                                    gy_ = fmaxf(fmaxf(blo[1] - (rhi[1] + sy_), (rlo[1] + sy_) - bhi[1]), 0.0f),
                                    gz_ = fmaxf(fmaxf(blo[2] - (rhi[2] + sz_), (rlo[2] + sz_) - bhi[2]), 0.0f);
                        keep = gx_ * gx_ + gy_ * gy_ + gz_ * gz_ <= r2cull;
#ifdef SRM_NOCULL
                        keep = true;
#endif
                    }
                    const unsigned long long kmask = __ballot(keep);
                    const int nb = __popcll(kmask);   // blocks of this batch (uniform)
                    SRM_COUNT(5, lane == 0 ? __popcll(__ballot(isblk)) - nb : 0);
                    if (nb == 0) continue;
                    {
                        const int rank = __popcll(kmask & ((1ull << lane) - 1ull));
                        const int to = 4 * (keep ? rank : 63);
                        btab = (unsigned)__builtin_amdgcn_ds_permute(to, (int)btab);
                        grow = (unsigned)__builtin_amdgcn_ds_permute(to, (int)grow);
                        pid = __builtin_amdgcn_ds_permute(to, pid);
                        if (lane >= nb) btab = 0, grow = 0;
                    }
                    // (C) the products, eight blocks (one mask word) at a time: operand rows
                    // straight from the list — 16 rows x 16 bytes, one coalesced read
                    SrmMasks M;
                    M.wc = 0;
                    int curpid = -1;
                    float bq = 0;
                    f32x4 cvec = {0, 0, 0, 0};
#pragma unroll
                    for (int w = 0; w < kWords; w++) {
                        M.m[w] = 0;
                        if (8 * w < nb) {  // wave-uniform
                            float av[8];
#pragma unroll
                            for (int i = 0; i < 8; i++) {
                                // (blocks past the batch: row 0 of the list, their bits are
                                // never looked at)
                                const unsigned srow =
                                    (unsigned)__builtin_amdgcn_readlane((int)grow, 8 * w + i);
                                // (a scalar base + one lane offset: no vector address arithmetic)
                                av[i] = (aop1 + 4 * (i64)srow)[lane_off];
                            }
                            unsigned cur = 0;
#pragma unroll
                            for (int i = 0; i < 8; i++) {
                                const int bp = __builtin_amdgcn_readlane(pid, 8 * w + i);
                                if (bp != curpid) {  // another piece: the receivers relative
                                    curpid = bp;     // to its corner
                                    const float ux = fx - __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pdx), bp)),
                                                uy = fy - __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pdy), bp)),
                                                uz = fz - __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pdz), bp));
                                    const float cval =
                                        valid ? (ux * ux + uy * uy + uz * uz) - r2pre : 1e30f;
                                    cvec = f32x4{cval, cval, cval, cval};
                                    bq = g == 0 ? ux : (g == 1 ? uy : (g == 2 ? uz : 1.0f));
                                }
                                const f32x4 d =
                                    __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bq, cvec, 0, 0, 0);
                                SRM_COUNT(3, lane == 0 ? 1 : 0);
#pragma unroll
                                for (int r = 0; r < 4; r++)
                                    cur = __builtin_amdgcn_alignbit(cur, __float_as_uint(d[r]), 31);
                            }
                            M.m[w] = cur;
                        }
                    }
                    // (D) the pairs
#ifdef SRM_PROBE_NOCAND
                    ax += (double)(M.m[0] ^ M.m[1] ^ M.m[2] ^ M.m[3] ^ btab) * 1e-300;
#else
                    if (wface)
                        srm_candidates<true>(M, btab, g, xi, yi, zi, S, P.boxsize, P.r2_max,
                                             P.r2_index_scaling, tbl, ax, ay, az);
                    else
                        srm_candidates<false>(M, btab, g, xi, yi, zi, S, P.boxsize, P.r2_max,
                                              P.r2_index_scaling, tbl, ax, ay, az);
#endif
                }
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

// Rows of the densely populated cells of a list (cells of `extent`, nper per dimension, z
// fastest: the tiles of cg_shortrange_tiles, the half-tiles of cg_shortrange_cells) re-ordered
// by sub-cell; aop (operand rows that travel with the positions) may be null.
// Hilbert index of every cell of the 8^3 grid (Skilling's transpose form: Gray decode of the
// axes, then the bits interleaved)
static void srm_hilbert_table(unsigned short *out) {
    const int bits = 3, M = 1 << (bits - 1);
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++)
            for (int z = 0; z < 8; z++) {
                int X[3] = {x, y, z};
                for (int Q = M; Q > 1; Q >>= 1) {
                    const int P = Q - 1;
                    for (int i = 0; i < 3; i++) {
                        if (X[i] & Q) {
                            X[0] ^= P;
                        } else {
                            const int t = (X[0] ^ X[i]) & P;
                            X[0] ^= t;
                            X[i] ^= t;
                        }
                    }
                }
                for (int i = 1; i < 3; i++) X[i] ^= X[i - 1];
                int t = 0;
                for (int Q = M; Q > 1; Q >>= 1)
                    if (X[2] & Q) t ^= Q - 1;
                for (int i = 0; i < 3; i++) X[i] ^= t;
                int h = 0;
                for (int b = bits - 1; b >= 0; b--)
                    for (int i = 0; i < 3; i++) h = (h << 1) | ((X[i] >> b) & 1);
                out[(x * 8 + y) * 8 + z] = (unsigned short)h;
            }
}

int cgk_shortrange_subsort(cg_ctx *c, const unsigned *offset, i64 n, i64 nper, double extent,
                           unsigned *order, double *pos_sorted, float *aop) {
    static bool table_done = false;
    if (!table_done) {
        unsigned short h[512];
        srm_hilbert_table(h);
        CG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(srm_hilbert), h, sizeof(h)));
        table_done = true;
    }
    static int subsort = -1;
    if (subsort < 0) {
        const char *env = getenv("CONCEPT_GPU_SR_SUBSORT");
        subsort = env ? atoi(env) : 1;
    }
    if (!subsort || n <= 0) return 0;
    const i64 ncells = nper * nper * nper;
    const i64 maxdense = n / kSubMin + 1;
    const size_t head = (size_t)((4 * (maxdense + 4) + 255) / 256 * 256);
    const size_t need2 = head + sizeof(SrmRow) * (size_t)n;
    if (need2 > c->sr_sub_bytes) {
        CG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->sr_sub_tmp);
        c->sr_sub_tmp = nullptr;
        c->sr_sub_bytes = 0;
        CG_HIP(hipMalloc(&c->sr_sub_tmp, need2));
        c->sr_sub_bytes = need2;
    }
    unsigned *ndense = (unsigned *)c->sr_sub_tmp, *dense = ndense + 4;
    SrmRow *scratch = (SrmRow *)((char *)c->sr_sub_tmp + head);
    CG_HIP(hipMemsetAsync(ndense, 0, 16, c->stream));
    hipLaunchKernelGGL(k_srm_dense_tiles, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0,
                       c->stream, offset, (unsigned)ncells, ndense, dense);
    CG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_srm_subsort, dim3((unsigned)maxdense), dim3(256), 0, c->stream, offset,
                       ndense, dense, 1 / extent, (unsigned)nper, order, pos_sorted, (f32x4 *)aop,
                       scratch);
    CG_LAUNCH_CHECK();
    return 0;
}

// phase 0: the whole list; 1: the tiles' offsets only (histogram + scan — what a caller needs to
// decide whether it wants the list); 2: the rest, after a call with phase 1 on the same arguments
int cgk_shortrange_tiles_phase(cg_ctx *c, int phase, const double *pos, i64 n, i64 nt,
                               double tile_extent, const signed char *rung, int lowest_active,
                               unsigned *order, unsigned *offset, double *pos_sorted, float *aop) {
    const double eps = 2.220446049250313e-16;
    const double inv = (1 / tile_extent) * (1 - 2 * eps);
    const i64 ntiles = nt * nt * nt;
    const i64 blocks = (n + 255) / 256;
    if (phase != 2) {
        if ((size_t)(8 * (ntiles + 1)) > c->sr_tmp_bytes) {
            CG_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(c->sr_tmp);
            c->sr_tmp = nullptr;
            c->sr_tmp_bytes = 0;
            CG_HIP(hipMalloc(&c->sr_tmp, 8 * (ntiles + 1)));
            c->sr_tmp_bytes = 8 * (ntiles + 1);
        }
        unsigned *count = (unsigned *)c->sr_tmp;
        CG_HIP(hipMemsetAsync(c->sr_tmp, 0, 8 * (ntiles + 1), c->stream));
        if (n > 0) {
            hipLaunchKernelGGL(k_srm_histogram, dim3((unsigned)blocks), dim3(256), 0, c->stream,
                               pos, n, inv, (unsigned)nt, rung, lowest_active, count);
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
        CG_HIP(hipcub::DeviceScan::ExclusiveSum(c->scan_tmp, need, count, offset,
                                                (int)(ntiles + 1), c->stream));
    }
    if (phase != 1 && n > 0) {
        unsigned *cursor = (unsigned *)c->sr_tmp + (ntiles + 1);
        hipLaunchKernelGGL(k_srm_scatter, dim3((unsigned)blocks), dim3(256), 0, c->stream, pos, n,
                           inv, 1 / tile_extent, (unsigned)nt, rung, lowest_active, offset, cursor,
                           order, pos_sorted, (f32x4 *)aop);
        CG_LAUNCH_CHECK();
        // the densely populated tiles: rows re-ordered by sub-cell (a filtered list has fewer
        // rows than n: the scratch is sized for n all the same)
        if (cgk_shortrange_subsort(c, offset, n, nt, tile_extent, order, pos_sorted, aop)) return 1;
        if (aop) {
            hipLaunchKernelGGL(k_srm_bbox, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                               c->stream, pos_sorted, offset, (unsigned)ntiles, 1 / tile_extent,
                               (f32x4 *)aop + (n + 16));
            CG_LAUNCH_CHECK();
        }
    }
    return 0;
}

int cgk_shortrange_tiles(cg_ctx *c, const double *pos, i64 n, i64 nt, double tile_extent,
                         const signed char *rung, int lowest_active, unsigned *order,
                         unsigned *offset, double *pos_sorted, float *aop) {
    return cgk_shortrange_tiles_phase(c, 0, pos, n, nt, tile_extent, rung, lowest_active, order,
                                      offset, pos_sorted, aop);
}

int cgk_shortrange_sweep_tiles(cg_ctx *c, const double *pos_r_sorted, const unsigned *order_r,
                               const unsigned *off_r, double *dmom_r, const double *pos_s_sorted,
                               const unsigned *off_s, const float *aop_s, i64 n_s, i64 nt,
                               const double *table, int64_t tablesize, double r2_index_scaling,
                               double r2_max, double factor, const double *factors,
                               const signed char *rung_jumped) {
    const double eps = 2.220446049250313e-16;
    const double ext = c->p.boxsize / (double)nt;  // species.py:607-609
    SrmParams P{c->p.boxsize, ext, 1.0 / ext, (1 / ext) * (1 - 2 * eps), r2_index_scaling, r2_max,
                factor, factors, rung_jumped, n_s, (int)nt, (int)tablesize};
#if SRM_LDS_TABLE
    CG_CHECK(tablesize <= kTableLds, "cg_shortrange_sweep_tiles: tables of up to %d entries",
             kTableLds);
#endif
    if (n_s <= 0) return 0;
    hipLaunchKernelGGL(k_sr_sweep_mfma, dim3(kSplit, (unsigned)nt, (unsigned)nt),
                       dim3(64 * kWaves), 0, c->stream, pos_r_sorted, order_r, off_r, dmom_r,
                       pos_s_sorted, off_s, (const f32x4 *)aop_s, table, P);
    CG_LAUNCH_CHECK();
    return 0;
}
