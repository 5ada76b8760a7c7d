// cg_shortrange_dense.hip — the densely populated tiles of the P3M short-range sweep (round 4).
//
//   particle_particle           interactions.py:1563-1791  tile neighbours, periodic offset
//   gravity_pairwise_shortrange gravity.py:263-354         r2 cut, r2-indexed table, Δmom
//   subtile refinement          species.py:4031-4142 (init_subtiling: subtiles of a handful of
//                               particles), interactions.py:145-329 (refinement chosen from the
//                               population), interactions.py:1236-1251 (subtile pairs further apart
//                               than the range are never visited)
//
// The cells sweep (cg_shortrange.hip) searches 5 x 5 x 6 half-tile cells around every receiver
// whatever the density: 4.2-4.4 pair tests per pair in range.  Where a tile holds a hundred
// particles or more, the reference refines its subtiles until each holds ~10 and skips the
// subtile pairs that are out of reach.  The counterpart here:
//  * a list by tile (below: cg_shortrange_tiles) whose rows, in such a tile, follow a Hilbert
//    curve through 8^3 sub-cells, so that ANY run of consecutive rows is a
//    compact blob whose size follows the density (a run of k rows at density rho fills a volume
//    ~ k / rho) — subtiles by count instead of by edge length;
//  * a wavefront takes 16 consecutive receivers (four lanes each) and the suppliers come as
//    "quads" of 4 consecutive rows: a quad whose bounding box is further than the range from
//    the bounding box of the 16 receivers is skipped (one lane per quad, 64 quads per look);
//  * the quads that are left are evaluated exactly as the cells sweep evaluates a row of
//    suppliers — 16 receivers x 4 suppliers per trip, (xi - xj) + offset, r2, the range test,
//    table[int(r2*scaling)], three multiply-adds in FP64 — every contribution bit-identical to the
//    cells sweep's, only the order of the additions differs.  All 64 lanes work in every trip:
//    no per-lane candidate lists.
// Measured on a Gaussian blob of the bench's clustered box (profiles/HISTORY_design_r1_r5.md §16b): 2.1-2.6 pair
// tests per pair in range in tiles of 256 particles and more, 3.2 at 128-256, 4.1 at 64-128
// (the cells sweep: 4.1-4.7), worse below — so cg_shortrange_sweep_cells hands the tiles above a
// population threshold to this kernel and keeps the others.
#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <cstring>

#include "cg_internal.h"
#include "cg_tiles.h"

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

// Measured (256^3 / 512^3 clustered, tools/sr_dense_time.py): windows of 960 rows, four quads per
// trip at 92 registers (5 waves per SIMD) 53.2 ms; 448 rows and two quads per trip at 64 registers
// (8 waves per SIMD) 48.0 ms; 256 rows 48.4, 320 rows 50.1, 8 wavefronts per workgroup 47.9-49.4,
// two 57.9.  The pair loop is 18 vector instructions per trip of 64 pair tests.
constexpr int kdNB = 2;                // quads per trip of a wavefront (their table loads in flight together)
constexpr int kdWaves = 4;             // wavefronts per workgroup, 16 receivers each
constexpr int kdChunk = 16 * kdWaves;  // receivers per work item
constexpr int kdCap = 448;             // supplier rows per LDS window (whole quads)
constexpr int kdQuads = kdCap / 4;
constexpr int kdPieces = 18;           // 9 tile columns x 2 (a column that wraps around in z)
static_assert(kdCap % 4 == 0, "whole quads");

struct SrdParams {
    double boxsize, ext, inv_ext, r2_index_scaling, r2_max, factor;
    const double *factors;          // adaptive rungs: factors[rung_jumped[i]] per receiver
    const signed char *rung_jumped;
    int nt;
    unsigned long long *stats;      // cg_shortrange_stats: [3] tests, [4] in range, [5] trips
};

struct SrdShared {
    // the suppliers' positions as stored (+ 4 rows far away: the stand-in of an absent quad)
    double sx[kdCap + 4], sy[kdCap + 4], sz[kdCap + 4];
    unsigned char simg[kdCap + 4];     // image code of the row: (ix, iy, iz) 2 bits each, 1 = none
    // bounding box of every quad: tile units relative to the receivers' tile, image applied
    float qlo[3][kdQuads], qhi[3][kdQuads];
    unsigned pbeg[kdPieces], ppre[kdPieces + 1];
    unsigned char pimg[kdPieces];
};

// inclusive scan over the 64 lanes of a wave in DPP adds
__device__ __forceinline__ unsigned srd_wave_scan(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// ---------------------------------------------------------------------------
// Lists by tile (Tiling.sort, species.py:775-780; z fastest): a counting sort — histogram,
// scan, scatter — with the positions copied in list order.
// ---------------------------------------------------------------------------
// Tiling.sort (species.py:775-780) with tiling location 0
__device__ __forceinline__ unsigned srd_tile1(double x, double inv, unsigned nt) {
    unsigned t = (unsigned)(i64)((x - 0.0) * inv);
    return t >= nt ? nt - 1 : t;
}

// runs of equal keys inside a wavefront -> one atomic per run (device-scope atomics execute
// memory-side on MI355X; particle memory is in mesh-tile order, so runs are long)
__device__ __forceinline__ void srd_wave_runs(unsigned key, int lane, int &run_start,
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

constexpr unsigned kdNoKey = 0xffffffffu;

struct SrdKey {
    unsigned key;
    double x, y, z;
};
// (rung != null: only the particles on rungs >= lowest_active are listed — the receivers of a
// sub-step, gravity.py:318-349 through the tiles' active rungs)
__device__ __forceinline__ SrdKey srd_key(const double *__restrict__ pos, i64 p, i64 n, double inv,
                                          unsigned nt, const signed char *__restrict__ rung,
                                          int lowest_active) {
    SrdKey k;
    k.key = kdNoKey;
    k.x = k.y = k.z = 0;
    if (p < n && !(rung && rung[p] < lowest_active)) {
        k.x = pos[3 * p], k.y = pos[3 * p + 1], k.z = pos[3 * p + 2];
        k.key = (srd_tile1(k.x, inv, nt) * nt + srd_tile1(k.y, inv, nt)) * nt +
                srd_tile1(k.z, inv, nt);
    }
    return k;
}
__global__ __launch_bounds__(256) void k_srd_histogram(const double *__restrict__ pos, i64 n,
                                                       double inv, unsigned nt,
                                                       const signed char *__restrict__ rung,
                                                       int lowest_active,
                                                       unsigned *__restrict__ count) {
    const int lane = threadIdx.x & 63;
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const SrdKey k = srd_key(pos, p, n, inv, nt, rung, lowest_active);
    int rs, rl;
    srd_wave_runs(k.key, lane, rs, rl);
    if (lane == rs && k.key != kdNoKey) atomicAdd(&count[k.key], (unsigned)rl);
}
__global__ __launch_bounds__(256) void k_srd_scatter(const double *__restrict__ pos, i64 n,
                                                     double inv, unsigned nt,
                                                     const signed char *__restrict__ rung,
                                                     int lowest_active,
                                                     const unsigned *__restrict__ offset,
                                                     unsigned *__restrict__ cursor,
                                                     unsigned *__restrict__ order,
                                                     double *__restrict__ pos_sorted) {
    const int lane = threadIdx.x & 63;
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const SrdKey k = srd_key(pos, p, n, inv, nt, rung, lowest_active);
    int rs, rl;
    srd_wave_runs(k.key, lane, rs, rl);
    unsigned first = 0;
    if (lane == rs && k.key != kdNoKey)
        first = offset[k.key] + atomicAdd(&cursor[k.key], (unsigned)rl);
    first = __shfl(first, rs);
    if (k.key != kdNoKey) {
        const i64 q = (i64)first + (lane - rs);
        order[q] = (unsigned)p;
        if (pos_sorted) {
            pos_sorted[3 * q] = k.x;
            pos_sorted[3 * q + 1] = k.y;
            pos_sorted[3 * q + 2] = k.z;
        }
    }
}

// ---------------------------------------------------------------------------
// Density-adaptive order inside the tiles (the counterpart of the reference's automatic
// subtile refinement, species.py:4031-4142: subtiles fine enough to hold a handful of
// particles each).  A tile with many particles has its rows of the list re-ordered by
// sub-cell — 8^3 sub-cells along a Hilbert curve (consecutive places are face neighbours: a run
// of rows is a compact blob wherever it starts, which a Morton curve's jumps do not give) — so
// that 16 consecutive rows (a wavefront's receivers) and 4 consecutive rows (a quad of
// suppliers) are neighbours in space wherever the particles are many.  Tiles below `min_rows`
// stay as the tile sort left them.
// ---------------------------------------------------------------------------
__device__ unsigned short srd_hilbert[512];   // filled per context by srd_hilbert_upload
constexpr int kdSubCells = 8;

__device__ __forceinline__ unsigned srd_subkey(double x, double y, double z, double inv_ext,
                                               unsigned gx, unsigned gy, unsigned gz) {
    const int a = min(kdSubCells - 1, max(0, (int)((x * inv_ext - (double)gx) * kdSubCells))),
              b = min(kdSubCells - 1, max(0, (int)((y * inv_ext - (double)gy) * kdSubCells))),
              c = min(kdSubCells - 1, max(0, (int)((z * inv_ext - (double)gz) * kdSubCells)));
    return srd_hilbert[(a * kdSubCells + b) * kdSubCells + c];
}

struct SrdRow {  // what travels with a row of the list
    double x, y, z;
    unsigned order, pad;
};

// the tiles to re-order, each with a segment of the scratch (as many rows as it holds)
__global__ __launch_bounds__(256) void k_srd_sub_tiles(const unsigned *__restrict__ offset,
                                                       unsigned ntiles, unsigned min_rows,
                                                       unsigned cap_tiles, unsigned cap_rows,
                                                       unsigned *__restrict__ counters,
                                                       unsigned *__restrict__ tiles,
                                                       unsigned *__restrict__ seg) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const unsigned cnt = offset[t + 1] - offset[t];
    if (cnt < min_rows) return;
    // (a caller's bound that turns out too small leaves a tile in the order of the tile sort:
    // the sweep's results do not depend on the order, only its culling does)
    const unsigned at = atomicAdd(&counters[1], cnt);
    if (at + cnt > cap_rows) return;
    const unsigned i = atomicAdd(&counters[0], 1u);
    if (i >= cap_tiles) return;
    tiles[i] = t;
    seg[i] = at;
}

__global__ __launch_bounds__(256) void k_srd_subsort(const unsigned *__restrict__ offset,
                                                     const unsigned *__restrict__ counters,
                                                     const unsigned *__restrict__ tiles,
                                                     const unsigned *__restrict__ seg,
                                                     unsigned cap_tiles, double inv_ext,
                                                     unsigned nt, unsigned *__restrict__ order,
                                                     double *__restrict__ pos_sorted,
                                                     SrdRow *__restrict__ scratch) {
    __shared__ unsigned hist[512], base[512];
    if (blockIdx.x >= min(counters[0], cap_tiles)) return;
    const unsigned t = tiles[blockIdx.x];
    const unsigned b = offset[t], e = offset[t + 1];
    SrdRow *const mine = scratch + seg[blockIdx.x];
    const unsigned gz = t % nt, gy = (t / nt) % nt, gx = t / (nt * nt);
    for (int i = threadIdx.x; i < 512; i += 256) hist[i] = 0;
    __syncthreads();
    for (unsigned q = b + threadIdx.x; q < e; q += 256)
        atomicAdd(&hist[srd_subkey(pos_sorted[3 * (i64)q], pos_sorted[3 * (i64)q + 1],
                                   pos_sorted[3 * (i64)q + 2], inv_ext, gx, gy, gz)], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of the 512 counts: 8 per lane of one wave
        unsigned v[8], sum = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            v[i] = hist[8 * threadIdx.x + i];
            sum += v[i];
        }
        unsigned run = srd_wave_scan(sum) - sum;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            base[8 * threadIdx.x + i] = run;
            run += v[i];
        }
    }
    __syncthreads();
    for (unsigned q = b + threadIdx.x; q < e; q += 256) {
        SrdRow r;
        r.x = pos_sorted[3 * (i64)q], r.y = pos_sorted[3 * (i64)q + 1],
        r.z = pos_sorted[3 * (i64)q + 2];
        r.order = order[q];
        r.pad = 0;
        mine[atomicAdd(&base[srd_subkey(r.x, r.y, r.z, inv_ext, gx, gy, gz)], 1u)] = r;
    }
    __syncthreads();  // (the workgroup's own writes to global memory, read back by itself)
    __threadfence_block();
    for (unsigned q = b + threadIdx.x; q < e; q += 256) {
        const SrdRow r = mine[q - b];
        pos_sorted[3 * (i64)q] = r.x;
        pos_sorted[3 * (i64)q + 1] = r.y;
        pos_sorted[3 * (i64)q + 2] = r.z;
        order[q] = r.order;
    }
}

// Hilbert index of every cell of the 8^3 grid (Skilling's transpose form: Gray decode of the
// axes, then the bits interleaved)
void srd_hilbert_table(unsigned short *out) {
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

// Population of every tile from the half-tile cell list (8 cells = 4 z-pairs): how many
// receivers sit in tiles of min_pop and more, and in how many tiles
__global__ __launch_bounds__(256) void k_srd_precheck(const unsigned *__restrict__ off_cells,
                                                      unsigned nt, unsigned min_pop,
                                                      unsigned *__restrict__ out) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned pop = 0;
    if (t < nt * nt * nt) {
        const unsigned tc = t % nt, tb = (t / nt) % nt, ta = t / (nt * nt), nc = 2 * nt;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const unsigned cell = ((2 * ta + a) * nc + (2 * tb + b)) * nc + 2 * tc;
                pop += off_cells[cell + 2] - off_cells[cell];
            }
    }
    const bool dense = pop >= min_pop;
    const unsigned long long m = __ballot(dense);
    unsigned sum = dense ? pop : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) sum += __shfl_xor(sum, d);
    // (sum of pop^2 over the dense tiles: what the cells sweep would spend there is 18.75 pair
    // tests per receiver and particle of the neighbourhood, ~ 18.75 pop^2 per tile)
    unsigned long long sq = dense ? (unsigned long long)pop * pop : 0ull;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) sq += __shfl_xor(sq, d);
    if ((threadIdx.x & 63) == 0 && m) {
        atomicAdd(&out[0], sum);
        atomicAdd(&out[1], (unsigned)__popcll(m));
        atomicAdd((unsigned long long *)(out + 4), sq);
    }
}

// The plan: one byte per tile for the cells sweep (1 = yours), one work item (tile << 32 |
// chunk) per 64 receivers of the other tiles
__global__ __launch_bounds__(256) void k_srd_plan(const unsigned *__restrict__ off_tiles,
                                                  unsigned ntiles, unsigned min_pop,
                                                  const unsigned char *__restrict__ active_in,
                                                  unsigned char *__restrict__ take,
                                                  unsigned long long *__restrict__ items,
                                                  unsigned *__restrict__ nitems) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const unsigned pop = off_tiles[t + 1] - off_tiles[t];
    const bool dense = pop >= min_pop;
    // (active_in: the tiles with a receiver on an active rung, k_sr_tile_activity)
    take[t] = dense ? 0 : (active_in ? active_in[t] : 1);
    if (dense) {
        const unsigned nch = (pop + kdChunk - 1) / kdChunk;
        const unsigned base = atomicAdd(nitems, nch);
        for (unsigned c = 0; c < nch; c++) items[base + c] = ((unsigned long long)t << 32) | c;
    }
}

// A sub-step that kicks the rungs >= lowest_active only: the rung of every row of the cell list
__global__ __launch_bounds__(256) void k_srd_gather_rung(const unsigned *__restrict__ order,
                                                         const signed char *__restrict__ rung,
                                                         unsigned n, signed char *__restrict__ out) {
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) out[q] = rung[order[q]];
}
// ... and what its dense tiles (min_pop ACTIVE receivers and more; off_active: the tile offsets of
// the list of active receivers) would cost the cells sweep: active receivers x particles of the
// tile, summed — out[0] active receivers in such tiles, out[1] such tiles, out[2..3] the sum
__global__ __launch_bounds__(256) void k_srd_gate(const unsigned *__restrict__ off_active,
                                                  const unsigned *__restrict__ off_cells,
                                                  unsigned nt, unsigned min_pop,
                                                  unsigned *__restrict__ out) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned act = 0, pop = 0;
    if (t < nt * nt * nt) {
        act = off_active[t + 1] - off_active[t];
        if (act >= min_pop) {
            const unsigned tc = t % nt, tb = (t / nt) % nt, ta = t / (nt * nt), nc = 2 * nt;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const unsigned cell = ((2 * ta + a) * nc + (2 * tb + b)) * nc + 2 * tc;
                    pop += off_cells[cell + 2] - off_cells[cell];
                }
        }
    }
    const bool dense = act >= min_pop;
    const unsigned long long m = __ballot(dense);
    unsigned sum = dense ? act : 0u;
    unsigned long long sq = dense ? (unsigned long long)act * pop : 0ull;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        sum += __shfl_xor(sum, d);
        sq += __shfl_xor(sq, d);
    }
    if ((threadIdx.x & 63) == 0 && m) {
        atomicAdd(&out[0], sum);
        atomicAdd(&out[1], (unsigned)__popcll(m));
        atomicAdd((unsigned long long *)(out + 2), sq);
    }
}

// NB quads of 4 supplier rows against the wave's 16 receivers: lane (j, g) pairs receiver j
// with row g of every quad.  Same arithmetic, in the same order, as sr_cell_batch
// (cg_shortrange.hip): the table loads of the hits are all issued before the first is used.
template <bool FACE, int NB, bool STATS>
__device__ __forceinline__ void srd_quads(const SrdShared &S, const int (&q)[NB], int g, double xi,
                                          double yi, double zi, double boxsize, double r2_max,
                                          double r2_index_scaling,
                                          const double *__restrict__ table, double &ax, double &ay,
                                          double &az, bool counted, unsigned *cnt) {
    double xj[NB], yj[NB], zj[NB], r2[NB], t[NB];
    bool hit[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int row = 4 * q[k] + g;
        xj[k] = xi - S.sx[row];                          // interactions.py:1787-1789
        yj[k] = yi - S.sy[row];
        zj[k] = zi - S.sz[row];
        if (FACE) {                                      // gravity.py:299-302: + the image's
            const unsigned im = S.simg[row];             // offset -L, 0 or +L (exact products)
            xj[k] += (double)((int)(im & 3) - 1) * boxsize;
            yj[k] += (double)((int)((im >> 2) & 3) - 1) * boxsize;
            zj[k] += (double)((int)((im >> 4) & 3) - 1) * boxsize;
        }
        r2[k] = sr_r2(xj[k], yj[k], zj[k]);  // gravity.py:306
        hit[k] = r2[k] <= r2_max;                        // gravity.py:311
        if (STATS) {
            // a row of padding (the rest of a last quad, the far quad) sits at 1e300
            const bool valid = counted && S.sx[4 * q[k] + g] < 1e299;
            cnt[0] += (unsigned)__popcll(__ballot(valid));
            cnt[1] += (unsigned)__popcll(__ballot(valid && hit[k]));
            cnt[2]++;
        }
    }
#pragma unroll
    for (int k = 0; k < NB; k++) {
        t[k] = 0.0;
        if (hit[k]) t[k] = table[(unsigned)(int)(r2[k] * r2_index_scaling)];  // gravity.py:316-321
    }
#pragma unroll
    for (int k = 0; k < NB; k++) {
        ax = __builtin_fma(xj[k], t[k], ax);
        ay = __builtin_fma(yj[k], t[k], ay);
        az = __builtin_fma(zj[k], t[k], az);
    }
}

// the quads of one look (bits of `m`: quad 64 look + bit), four at a time
template <bool FACE, bool STATS>
__device__ __forceinline__ void srd_look(const SrdShared &S, unsigned long long m, int q0, int g,
                                         double xi, double yi, double zi, double boxsize,
                                         double r2_max, double r2_index_scaling,
                                         const double *__restrict__ table, double &ax, double &ay,
                                         double &az, bool counted, unsigned *cnt) {
    while (m) {
        int q[kdNB];
#pragma unroll
        for (int k = 0; k < kdNB; k++) {
            q[k] = m ? q0 + (int)__builtin_ctzll(m) : kdQuads;  // (none left: the far quad)
            m &= m - 1;                                         // (0 stays 0)
        }
        srd_quads<FACE, kdNB, STATS>(S, q, g, xi, yi, zi, boxsize, r2_max, r2_index_scaling, table, ax, ay,
                              az, counted, cnt);
    }
}

constexpr unsigned kdGranule = 16;   // consecutive work items per XCD (below)
template <bool STATS>
__global__ __launch_bounds__(64 * kdWaves)
__attribute__((amdgpu_waves_per_eu(8, 8))) void k_sr_sweep_dense(
    const double *__restrict__ pos_r, const unsigned *__restrict__ order_rt,
    const unsigned *__restrict__ order_rc, const unsigned *__restrict__ off_r,
    double *__restrict__ dmom_r, const double *__restrict__ pos_s,
    const unsigned *__restrict__ off_s, const unsigned long long *__restrict__ items,
    const unsigned *__restrict__ nitems, const double *__restrict__ table, SrdParams P) {
    __shared__ SrdShared S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int nt = P.nt;
    const unsigned count = *nitems;
    // Block b runs on XCD b % 8.  The items of a tile follow each other on the list and stage the
    // same 27 tiles: granules of kdGranule consecutive items go to one XCD (one L2), the
    // granules to the XCDs in turn (profiles/r04_sr_xcd_walk_ab.txt: 46.5 against 47.8 ms).
    constexpr unsigned G = kdGranule;
    const unsigned vmax = (count + 8u * G - 1u) / (8u * G) * (8u * G);
    for (unsigned v = blockIdx.x; v < vmax; v += gridDim.x) {
    if (v != blockIdx.x) __syncthreads();  // the previous item's tables and rows are done with
    const unsigned it = ((v / 8u / G) * 8u + v % 8u) * G + (v / 8u) % G;
    if (it >= count) continue;
    const unsigned long long item = items[it];
    const unsigned t = (unsigned)(item >> 32), chunk = (unsigned)item;
    const int tc = (int)(t % (unsigned)nt), tb = (int)((t / (unsigned)nt) % (unsigned)nt),
              ta = (int)(t / (unsigned)(nt * nt));
    const unsigned q0 = off_r[t] + (unsigned)kdChunk * chunk, q1 = min(q0 + (unsigned)kdChunk, off_r[t + 1]);
    const bool face = ta == 0 || ta == nt - 1 || tb == 0 || tb == nt - 1 || tc == 0 || tc == nt - 1;
    // the pieces: column (ta + dx, tb + dy), tiles tc - 1 .. tc + 1 — one run of the list (z is
    // fastest), two where the column wraps around the box in z
    if (tid < kdPieces) {
        // (worked out per item: hoisted out of the item loop, the column's coordinates were the
        // two registers this kernel spilled at its 64)
        int c9 = tid >> 1;
        asm volatile("" : "+v"(c9));
        const int half = tid & 1;
        int gx = ta + c9 / 3 - 1, gy = tb + c9 % 3 - 1;
        // periodic offset from the tile separation (interactions.py:1615-1621): code 2 = +L
        unsigned ix = 1, iy = 1, iz = 1;
        if (gx < 0) { gx += nt; ix = 2; } else if (gx >= nt) { gx -= nt; ix = 0; }
        if (gy < 0) { gy += nt; iy = 2; } else if (gy >= nt) { gy -= nt; iy = 0; }
        int za = 0, zb = -1;  // tiles [za, zb] of the column
        if (tc == 0) {
            if (half == 0) { za = zb = nt - 1; iz = 2; } else { za = 0; zb = 1; }
        } else if (tc == nt - 1) {
            if (half == 0) { za = nt - 2; zb = nt - 1; } else { za = zb = 0; iz = 0; }
        } else if (half == 0) {
            za = tc - 1;
            zb = tc + 1;
        }
        const unsigned col = ((unsigned)gx * (unsigned)nt + (unsigned)gy) * (unsigned)nt;
        unsigned beg = 0, cnt = 0;
        if (zb >= za) {
            beg = off_s[col + (unsigned)za];
            cnt = off_s[col + (unsigned)zb + 1] - beg;
        }
        S.pbeg[tid] = beg;
        S.ppre[tid] = cnt;  // (counts for now)
        S.pimg[tid] = (unsigned char)(ix | (iy << 2) | (iz << 4));
    }
    if (tid < 4) {  // the far quad
        S.sx[kdCap + tid] = S.sy[kdCap + tid] = S.sz[kdCap + tid] = 1e300;
        S.simg[kdCap + tid] = 0x15;
    }
    // this wave's 16 receivers, four lanes each
    const unsigned qw = q0 + 16u * (unsigned)wave;
    const bool wvalid = qw < q1;
    const bool valid = qw + (unsigned)j < q1;
    const unsigned ql = valid ? qw + (unsigned)j : q0;  // (a finite stand-in)
    const double xi = pos_r[3 * (i64)ql], yi = pos_r[3 * (i64)ql + 1], zi = pos_r[3 * (i64)ql + 2];
    const double cx = (double)ta * P.ext, cy = (double)tb * P.ext, cz = (double)tc * P.ext;
    // the box of the wave's receivers, in tiles relative to the tile's corner (wave-uniform)
    float rlo[3], rhi[3];
    {
        const double pr[3] = {(xi - cx) * P.inv_ext, (yi - cy) * P.inv_ext, (zi - cz) * P.inv_ext};
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float v = (float)pr[d];
            rlo[d] = valid ? v : 3.0e38f;
            rhi[d] = valid ? v : -3.0e38f;
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                rlo[d] = fminf(rlo[d], __shfl_xor(rlo[d], m));
                rhi[d] = fmaxf(rhi[d], __shfl_xor(rhi[d], m));
            }
        }
    }
    // (single-precision coordinates of size <= 2 tiles: 2^-23 each, on a distance ~ 1)
    const float r2cull = (float)(P.r2_max * P.inv_ext * P.inv_ext) * 1.0001f + 1.0e-5f;
    __syncthreads();
    if (wave == 0) {  // counts -> exclusive prefix
        const unsigned cnt = lane < kdPieces ? S.ppre[lane] : 0u;
        const unsigned incl = srd_wave_scan(cnt);
        if (lane < kdPieces) S.ppre[lane] = incl - cnt;
        if (lane == kdPieces) S.ppre[kdPieces] = incl;
    }
    __syncthreads();
    const unsigned total = S.ppre[kdPieces];
    double ax = 0, ay = 0, az = 0;
    unsigned cnt[3] = {0, 0, 0};
    for (unsigned r0 = 0; r0 < total; r0 += kdCap) {
        const int nrows = (int)(min(total, r0 + (unsigned)kdCap) - r0);
        const int nq = (nrows + 3) >> 2;
        if (r0) __syncthreads();  // the previous window has been consumed
        for (int w = tid; w < 4 * nq; w += 64 * kdWaves) {
            if (w < nrows) {
                const unsigned row = r0 + (unsigned)w;
                int p = 0;  // the last piece that starts at or before the row
#pragma unroll
                for (int step = 16; step; step >>= 1)
                    if (p + step < kdPieces && S.ppre[p + step] <= row) p += step;
                const i64 src = (i64)S.pbeg[p] + (row - S.ppre[p]);
                S.sx[w] = pos_s[3 * src];
                S.sy[w] = pos_s[3 * src + 1];
                S.sz[w] = pos_s[3 * src + 2];
                S.simg[w] = S.pimg[p];
            } else {  // the rest of the last quad: out of everybody's range
                S.sx[w] = S.sy[w] = S.sz[w] = 1e300;
                S.simg[w] = 0x15;
            }
        }
        __syncthreads();
        for (int q = tid; q < nq; q += 64 * kdWaves) {
            float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int w = 4 * q + i;
                if (w < nrows) {
                    const unsigned im = S.simg[w];
                    // x_ji = (xi - xj) + offset: the supplier's image sits at xj - offset
                    const double v[3] = {
                        ((S.sx[w] - (double)((int)(im & 3) - 1) * P.boxsize) - cx) * P.inv_ext,
                        ((S.sy[w] - (double)((int)((im >> 2) & 3) - 1) * P.boxsize) - cy) * P.inv_ext,
                        ((S.sz[w] - (double)((int)((im >> 4) & 3) - 1) * P.boxsize) - cz) * P.inv_ext};
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        lo[d] = fminf(lo[d], (float)v[d]);
                        hi[d] = fmaxf(hi[d], (float)v[d]);
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < 3; d++) {
                S.qlo[d][q] = lo[d];
                S.qhi[d][q] = hi[d];
            }
        }
        __syncthreads();
        if (wvalid)
        for (int look = 0; 64 * look < nq; look++) {
            const int q = 64 * look + lane;
            bool keep = false;
            if (q < nq) {
                // minimum distance of the receivers' box to the quad's (interactions.py:1236-1251)
                float d2 = 0;
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const float gap = fmaxf(fmaxf(S.qlo[d][q] - rhi[d], rlo[d] - S.qhi[d][q]), 0.0f);
                    d2 = __builtin_fmaf(gap, gap, d2);
                }
                keep = d2 <= r2cull;
            }
            const unsigned long long m = __ballot(keep);
            if (face)
                srd_look<true, STATS>(S, m, 64 * look, g, xi, yi, zi, P.boxsize, P.r2_max,
                               P.r2_index_scaling, table, ax, ay, az, valid, cnt);
            else
                srd_look<false, STATS>(S, m, 64 * look, g, xi, yi, zi, P.boxsize, P.r2_max,
                                P.r2_index_scaling, table, ax, ay, az, valid, cnt);
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
        i64 pi = (i64)order_rt[ql];
        if (order_rc) pi = (i64)order_rc[pi];
        // gravity.py:321 (total_factor = factors[rung] * table[...])
        const double f = P.factors ? P.factors[P.rung_jumped[pi]] : P.factor;
        dmom_r[3 * pi] += ax * f;
        dmom_r[3 * pi + 1] += ay * f;
        dmom_r[3 * pi + 2] += az * f;
    }
    if (STATS && lane == 0) {
        atomicAdd(&P.stats[3], (unsigned long long)cnt[0]);
        atomicAdd(&P.stats[4], (unsigned long long)cnt[1]);
        atomicAdd(&P.stats[5], (unsigned long long)cnt[2]);
    }
    }
}

// grows a device buffer of the context (the stream is drained first: the old one may be in use)
int srd_reserve(cg_ctx *c, void **buf, size_t *have, size_t need) {
    if (need <= *have) return 0;
    CG_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    CG_HIP(hipMalloc(buf, need));
    *have = need;
    return 0;
}

}  // namespace

// CONCEPT_GPU_SR_DENSE_MIN: the population from which a tile is "dense" (64; 0 switches the
// dense tiles' sweep off; read at every call: a test may change it)
int cgk_shortrange_dense_min() {
    const char *mp = getenv("CONCEPT_GPU_SR_DENSE_MIN");
    const int min_pop = mp ? atoi(mp) : 64;
    return min_pop < 1 ? -1 : min_pop;
}

// phase 0: the whole list; 1: the tiles' offsets only (histogram + scan — what a caller needs to
// decide whether it wants the list); 2: the rest, after a call with phase 1 on the same arguments.
// sub_min > 0: the tiles of sub_min rows and more are re-ordered by sub-cell (their rows are
// sub_rows at most — the caller's count; a bound that is too small leaves tiles unordered).
int cgk_shortrange_tiles_phase(cg_ctx *c, int phase, const double *pos, i64 n, i64 nt,
                               double tile_extent, const signed char *rung, int lowest_active,
                               unsigned *order, unsigned *offset, double *pos_sorted,
                               i64 sub_min, i64 sub_rows) {
    const double eps = 2.220446049250313e-16;
    const double inv = (1 / tile_extent) * (1 - 2 * eps);
    const i64 ntiles = nt * nt * nt;
    const i64 blocks = (n + 255) / 256;
    if (phase != 2) {
        if (srd_reserve(c, &c->sr_tmp, &c->sr_tmp_bytes, (size_t)(8 * (ntiles + 1)))) return 1;
        unsigned *count = (unsigned *)c->sr_tmp;
        CG_HIP(hipMemsetAsync(c->sr_tmp, 0, 8 * (ntiles + 1), c->stream));
        if (n > 0) {
            hipLaunchKernelGGL(k_srd_histogram, dim3((unsigned)blocks), dim3(256), 0, c->stream,
                               pos, n, inv, (unsigned)nt, rung, lowest_active, count);
            CG_LAUNCH_CHECK();
        }
        size_t need = 0;
        CG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, count, offset, (int)(ntiles + 1),
                                                c->stream));
        if (srd_reserve(c, &c->scan_tmp, &c->scan_tmp_bytes, need)) return 1;
        CG_HIP(hipcub::DeviceScan::ExclusiveSum(c->scan_tmp, need, count, offset,
                                                (int)(ntiles + 1), c->stream));
    }
    if (phase != 1 && n > 0) {
        unsigned *cursor = (unsigned *)c->sr_tmp + (ntiles + 1);
        hipLaunchKernelGGL(k_srd_scatter, dim3((unsigned)blocks), dim3(256), 0, c->stream, pos, n,
                           inv, (unsigned)nt, rung, lowest_active, offset, cursor, order,
                           pos_sorted);
        CG_LAUNCH_CHECK();
        if (sub_min > 0 && sub_rows > 0 && pos_sorted) {
            if (!c->srd_hilbert_done) {  // (per context: the table lives on the context's device)
                unsigned short h[512];
                srd_hilbert_table(h);
                CG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(srd_hilbert), h, sizeof(h)));
                c->srd_hilbert_done = true;
            }
            const i64 cap_tiles = sub_rows / sub_min + 1;
            const size_t head = (size_t)((8 * (cap_tiles + 2) + 255) / 256 * 256);
            if (srd_reserve(c, &c->sr_sub_tmp, &c->sr_sub_bytes,
                            head + sizeof(SrdRow) * (size_t)sub_rows))
                return 1;
            unsigned *counters = (unsigned *)c->sr_sub_tmp, *tiles = counters + 2,
                     *seg = tiles + cap_tiles;
            SrdRow *scratch = (SrdRow *)((char *)c->sr_sub_tmp + head);
            CG_HIP(hipMemsetAsync(counters, 0, 8, c->stream));
            hipLaunchKernelGGL(k_srd_sub_tiles, dim3((unsigned)((ntiles + 255) / 256)), dim3(256),
                               0, c->stream, offset, (unsigned)ntiles, (unsigned)sub_min,
                               (unsigned)cap_tiles, (unsigned)sub_rows, counters, tiles, seg);
            CG_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_srd_subsort, dim3((unsigned)cap_tiles), dim3(256), 0, c->stream,
                               offset, counters, tiles, seg, (unsigned)cap_tiles,
                               1 / tile_extent, (unsigned)nt, order, pos_sorted, scratch);
            CG_LAUNCH_CHECK();
        }
    }
    return 0;
}

// cg_shortrange_tiles: the plain list by tile (no sub-cell order)
int cgk_shortrange_tiles(cg_ctx *c, const double *pos, i64 n, i64 nt, double tile_extent,
                         const signed char *rung, int lowest_active, unsigned *order,
                         unsigned *offset, double *pos_sorted) {
    return cgk_shortrange_tiles_phase(c, 0, pos, n, nt, tile_extent, rung, lowest_active, order,
                                      offset, pos_sorted, 0, 0);
}

// The look at a half-tile cell list, taken when the list is built (cg_shortrange_cells): how
// many of its particles sit in tiles of min_pop and more, in how many tiles, the sum of their
// squared populations, and the length of the list — copied to pinned memory behind an event, so
// that the sweeps over this list (one per sub-step with adaptive rungs) find the answer without
// a kernel, a copy or a wait of their own.  Slots are keyed by the offsets' address.
namespace {
constexpr int kdLooks = 4;
int srd_look_slot(cg_ctx *c, const unsigned *off, i64 nt, int min_pop) {
    for (int i = 0; i < kdLooks; i++)
        if (c->srd_look[i].off == off && c->srd_look[i].nt == nt &&
            c->srd_look[i].min_pop == min_pop)
            return i;
    return -1;
}
}  // namespace
int cgk_shortrange_dense_look(cg_ctx *c, const unsigned *off_cells, i64 nt) {
    for (int i = 0; i < kdLooks; i++)   // whatever was known about this address is stale now
        if (c->srd_look[i].off == off_cells) c->srd_look[i].off = nullptr;
    const int min_pop = cgk_shortrange_dense_min();
    if (min_pop < 0 || nt < 4) return 0;
    const i64 ntiles = nt * nt * nt, ncells = 8 * ntiles;
    if (!c->srd_host) CG_HIP(hipHostMalloc((void **)&c->srd_host, 256));
    if (srd_reserve(c, (void **)&c->srd_small, &c->srd_small_bytes, 256)) return 1;
    const int slot = c->srd_look_next;
    c->srd_look_next = (slot + 1) % kdLooks;
    cg_ctx::SrdLook &L = c->srd_look[slot];
    if (!L.ev) CG_HIP(hipEventCreateWithFlags(&L.ev, hipEventDisableTiming));
    // device words 16 + 8 slot: [0] particles in dense tiles, [1] dense tiles, [4..5] sum pop^2
    unsigned *dev = (unsigned *)c->srd_small + 16 + 8 * slot, *host = c->srd_host + 16 + 8 * slot;
    CG_HIP(hipMemsetAsync(dev, 0, 32, c->stream));
    hipLaunchKernelGGL(k_srd_precheck, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0,
                       c->stream, off_cells, (unsigned)nt, (unsigned)min_pop, dev);
    CG_LAUNCH_CHECK();
    CG_HIP(hipMemcpyAsync(host, dev, 32, hipMemcpyDeviceToHost, c->stream));
    CG_HIP(hipMemcpyAsync(host + 2, off_cells + ncells, 4, hipMemcpyDeviceToHost, c->stream));
    CG_HIP(hipEventRecord(L.ev, c->stream));
    L.off = off_cells;
    L.nt = nt;
    L.min_pop = min_pop;
    L.harvested = false;
    return 0;
}

// Are there tiles of min_pop receivers and more (the look taken when the receivers' cell list
// was built)?  If so: tile lists (Hilbert order inside the dense tiles) of the receivers and the
// suppliers, the byte per tile that keeps the cells sweep off the dense tiles (*take_out) and the
// launch of the dense tiles' sweep on a stream of its own (joined back into the context's stream
// by cgk_shortrange_dense_join).  *take_out stays null where the cells sweep does it all.
// rung, lowest_active > 0: a sub-step for the upper rungs — the receivers are the particles on
// rungs >= lowest_active (gravity.py:318-349), "dense" counts those, active_in is the cells
// sweep's byte per tile (tiles with an active receiver).
// Host waits: for the look's event (over when the cell list is; nothing on a list looked at
// before), and — only on a sub-step over a box that has dense tiles — once for the gate's counts.
int cgk_shortrange_dense(cg_ctx *c, const double *pos_r_sorted, const unsigned *order_r,
                         const unsigned *off_r, double *dmom_r, const double *pos_s_sorted,
                         const unsigned *off_s, i64 nt, const double *table,
                         double r2_index_scaling, double r2_max, double factor,
                         const double *factors, const signed char *rung,
                         const signed char *rung_jumped, int lowest_active,
                         const unsigned char *active_in, const unsigned char **take_out) {
    *take_out = nullptr;
    const int min_pop = cgk_shortrange_dense_min();
    if (min_pop < 0 || nt < 4) return 0;
    const bool partial = rung && lowest_active > 0;
    const bool by_threshold = getenv("CONCEPT_GPU_SR_DENSE_MIN") != nullptr;
    const i64 ntiles = nt * nt * nt;
    int lr = srd_look_slot(c, off_r, nt, min_pop);
    if (lr < 0) {  // (a list that did not come from cg_shortrange_cells of this context)
        if (cgk_shortrange_dense_look(c, off_r, nt)) return 1;
        lr = srd_look_slot(c, off_r, nt, min_pop);
    }
    // The looks that have completed since the last call: how many in a row found nothing dense.
    // While that has been so for a while (a box without clumps: every sub-step of a rung loop)
    // a look that is still on its way is not waited for — its sweep goes without the dense
    // tiles' form, which is a matter of speed alone, and the host keeps queueing (the wait had
    // the GPU idle once per sub-step until the sweep's launch arrived).  A look that finds
    // dense tiles ends the streak; the sweeps after it wait again.
    for (int i = 0; i < kdLooks; i++) {
        cg_ctx::SrdLook &K = c->srd_look[i];
        if (K.ev && !K.harvested && hipEventQuery(K.ev) == hipSuccess) {
            K.harvested = true;
            c->srd_quiet = c->srd_host[16 + 8 * i] == 0 ? c->srd_quiet + 1 : 0;
        }
    }
    if (!c->srd_look[lr].harvested) {
        if (c->srd_quiet >= 2 * kdLooks && !by_threshold) return 0;
        CG_HIP(hipEventSynchronize(c->srd_look[lr].ev));
        c->srd_look[lr].harvested = true;
        c->srd_quiet = c->srd_host[16 + 8 * lr] == 0 ? c->srd_quiet + 1 : 0;
    }
    const unsigned *hr = c->srd_host + 16 + 8 * lr;
    const i64 ndense = hr[0], tdense = hr[1], n_r = hr[2];
    unsigned long long sq;   // (with them: a look for the suppliers below may take this slot)
    memcpy(&sq, hr + 4, 8);
    // buffers nobody has used for a while go back (a run whose clumps dissolve, a test)
    if (ndense == 0) {
        if (++c->srd_idle >= 16 && c->srd_buf_bytes + c->sr_sub_bytes > ((size_t)256 << 20)) {
            CG_HIP(hipStreamSynchronize(c->stream));
            if (c->srd_stream) CG_HIP(hipStreamSynchronize(c->srd_stream));
            (void)hipFree(c->srd_buf);
            (void)hipFree(c->sr_sub_tmp);
            c->srd_buf = c->sr_sub_tmp = nullptr;
            c->srd_buf_bytes = c->sr_sub_bytes = 0;
        }
        return 0;
    }
    // (the list of a sub-step's receivers is a subset: the suppliers get a list of their own)
    const bool same = !partial && pos_r_sorted == pos_s_sorted && off_r == off_s;
    i64 n_s = n_r, sdense = ndense;
    if (off_s != off_r) {
        int ls = srd_look_slot(c, off_s, nt, min_pop);
        if (ls < 0) {
            if (cgk_shortrange_dense_look(c, off_s, nt)) return 1;
            ls = srd_look_slot(c, off_s, nt, min_pop);
        }
        CG_HIP(hipEventSynchronize(c->srd_look[ls].ev));
        c->srd_look[ls].harvested = true;
        n_s = c->srd_host[16 + 8 * ls + 2];
        sdense = c->srd_host[16 + 8 * ls];
    }
    // Is it worth the lists?  What the dense tiles cost the cells sweep (~18.75 pair tests per
    // receiver and particle of the tile at its 0.9e12 tests/s; this sweep needs about half)
    // against a list build (measured 0.07 ns per particle; 0.1 here).  A uniform box of 2^28
    // particles at 44 per tile has 0.3 % of its tiles above the threshold and 17 ms of lists to
    // pay: not worth it.  (With CONCEPT_GPU_SR_DENSE_MIN set the threshold alone decides.)
    const double cost = 5e-5 + 1e-10 * (double)(n_r + (same ? 0 : n_s));
    if (!by_threshold && 0.45 * 18.75 * (double)sq / 0.9e12 < 2 * cost) return 0;
    c->srd_idle = 0;
    // [2] items, [6] active receivers in tiles dense with them, [7] such tiles, [8..9] sum of
    // active x pop
    unsigned *dev = (unsigned *)c->srd_small;
    CG_HIP(hipMemsetAsync(dev, 0, 40, c->stream));
    // buffers: take | items | offsets r, s | order r, s | positions r, s
    const size_t a_take = 0, a_items = (size_t)((ntiles + 255) / 256 * 256),
                 a_offr = a_items + 8 * (size_t)(ndense / kdChunk + tdense + 1),
                 a_offs = a_offr + (size_t)((4 * (ntiles + 1) + 255) / 256 * 256),
                 a_ordr = a_offs + (same ? 0 : (size_t)((4 * (ntiles + 1) + 255) / 256 * 256)),
                 a_ords = a_ordr + (size_t)((4 * n_r + 255) / 256 * 256),
                 a_posr = a_ords + (same ? 0 : (size_t)((4 * n_s + 255) / 256 * 256)),
                 a_poss = a_posr + 24 * (size_t)n_r,
                 a_end = a_poss + (same ? 0 : 24 * (size_t)n_s);
    if (srd_reserve(c, &c->srd_buf, &c->srd_buf_bytes, a_end)) return 1;
    char *B = (char *)c->srd_buf;
    unsigned char *take = (unsigned char *)(B + a_take);
    unsigned long long *items = (unsigned long long *)(B + a_items);
    unsigned *offr = (unsigned *)(B + a_offr), *offs = same ? offr : (unsigned *)(B + a_offs);
    unsigned *ordr = (unsigned *)(B + a_ordr), *ords = same ? ordr : (unsigned *)(B + a_ords);
    double *posr = (double *)(B + a_posr), *poss = same ? posr : (double *)(B + a_poss);
    const double ext = c->p.boxsize / (double)nt;  // species.py:607-609
    if (partial) {
        // the active receivers' offsets first (a histogram: 0.07 ms at 256^3): do the tiles that
        // are dense WITH THEM hold enough of this sub-step's pair work?
        if (srd_reserve(c, &c->srd_rung, &c->srd_rung_bytes, (size_t)n_r)) return 1;
        signed char *rung_sorted = (signed char *)c->srd_rung;
        hipLaunchKernelGGL(k_srd_gather_rung, dim3((unsigned)((n_r + 255) / 256)), dim3(256), 0,
                           c->stream, order_r, rung, (unsigned)n_r, rung_sorted);
        CG_LAUNCH_CHECK();
        if (cgk_shortrange_tiles_phase(c, 1, pos_r_sorted, n_r, nt, ext, rung_sorted, lowest_active,
                                       ordr, offr, posr, 0, 0))
            return 1;
        hipLaunchKernelGGL(k_srd_gate, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0,
                           c->stream, offr, off_r, (unsigned)nt, (unsigned)min_pop, dev + 6);
        CG_LAUNCH_CHECK();
        CG_HIP(hipMemcpyAsync(c->srd_host + 6, dev + 6, 16, hipMemcpyDeviceToHost, c->stream));
        CG_HIP(hipStreamSynchronize(c->stream));
        memcpy(&sq, c->srd_host + 8, 8);
        if (c->srd_host[6] == 0) return 0;
        if (!by_threshold && 0.45 * 18.75 * (double)sq / 0.9e12 < 2 * cost) return 0;
        if (cgk_shortrange_tiles_phase(c, 2, pos_r_sorted, n_r, nt, ext, rung_sorted, lowest_active,
                                       ordr, offr, posr, min_pop, (i64)c->srd_host[6]))
            return 1;
    } else if (cgk_shortrange_tiles_phase(c, 0, pos_r_sorted, n_r, nt, ext, nullptr, 0, ordr, offr,
                                          posr, min_pop, ndense)) {
        return 1;
    }
    if (!same && cgk_shortrange_tiles_phase(c, 0, pos_s_sorted, n_s, nt, ext, nullptr, 0, ords,
                                            offs, poss, min_pop, sdense))
        return 1;
    hipLaunchKernelGGL(k_srd_plan, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, c->stream,
                       offr, (unsigned)ntiles, (unsigned)min_pop, partial ? active_in : nullptr,
                       take, items, dev + 2);
    CG_LAUNCH_CHECK();
    if (!c->srd_stream) {
        CG_HIP(hipStreamCreateWithFlags(&c->srd_stream, hipStreamNonBlocking));
        CG_HIP(hipEventCreateWithFlags(&c->srd_fork, hipEventDisableTiming));
        CG_HIP(hipEventCreateWithFlags(&c->srd_join, hipEventDisableTiming));
    }
    CG_HIP(hipEventRecord(c->srd_fork, c->stream));
    CG_HIP(hipStreamWaitEvent(c->srd_stream, c->srd_fork, 0));
    SrdParams P{c->p.boxsize, ext, 1.0 / ext, r2_index_scaling, r2_max, factor, factors,
                rung_jumped, (int)nt, c->sr_stats};
    // (as many workgroups as there can be items — counted on all the particles, an upper bound
    // for a subset; the ones past the count leave at once)
    const unsigned grid = (unsigned)(ndense / kdChunk + tdense) + 8u * kdGranule;
    hipLaunchKernelGGL(c->sr_stats ? k_sr_sweep_dense<true> : k_sr_sweep_dense<false>, dim3(grid),
                       dim3(64 * kdWaves), 0, c->srd_stream, posr,
                       ordr, order_r, offr, dmom_r, poss, offs, items, dev + 2, table, P);
    CG_LAUNCH_CHECK();
    CG_HIP(hipEventRecord(c->srd_join, c->srd_stream));
    *take_out = take;
    return 0;
}

int cgk_shortrange_dense_join(cg_ctx *c) {
    CG_HIP(hipStreamWaitEvent(c->stream, c->srd_join, 0));
    return 0;
}
