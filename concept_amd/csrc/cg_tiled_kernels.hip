// cg_tiled_kernels.hip — LDS-tiled CIC deposit (A1/A2) and fused finite
// difference + CIC gather + kick (A9/A10) for particles kept in mesh-tile
// order (cg_sort_particles).  One 512-lane workgroup per T^3-cell tile.
//
// Why tiles: on MI355X device-scope FP64 atomics execute memory-side, one
// fabric transaction per lane (measured at 2^28 particles / 1024^3: 2^31
// direct atomic adds = 77 ms; LDS tiles with atomics only on the tile faces
// = 28 ms).  The deposit therefore uses no global atomics at all: each
// workgroup accumulates everything its tile receives in LDS (ds_add_f64) and
// stores the tile once.
//
// Arithmetic per particle and per cell is the reference's (see
// cg_mesh_kernels.hip); only the order in which particles are added to a
// cell differs.  Compiled with -ffp-contract=off.
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

// 32-bit cell arithmetic: x is positive and < gridsize + 2*nghosts + 1 (positions are in
// [0, boxsize)), so the reference's truncation to Py_ssize_t equals truncation to int —
// one v_cvt_i32_f64 instead of the ~15-instruction double -> int64 sequence, and every
// index below (N <= 2048 and change) stays in 32-bit registers.
// two doubles at 8-byte alignment (the xy of an xyz record)
typedef double d2u8 __attribute__((ext_vector_type(2), aligned(8)));
struct Cic1 {
    int index;
    double w0, w1;
};
__device__ __forceinline__ Cic1 cic1(double pos, double off, double scale) {
    double x = (pos - off) * scale;
    Cic1 r;
    r.index = (int)x;
    double dist = x - (double)r.index;
    r.w0 = 1 - dist;
    r.w1 = dist;
    return r;
}
__device__ __forceinline__ int wrap(int a, int n) {
    a = a < 0 ? a + n : a;
    return a >= n ? a - n : a;
}

// blockIdx -> tile.  Consecutive workgroups land on different XCDs
// (block b -> XCD b % 8, observed); give each XCD a contiguous eighth of the
// tile list so that neighbouring tiles (shared halo rows) meet in one L2.
__device__ __forceinline__ unsigned tile_for_block(unsigned b, unsigned ntiles) {
    if (ntiles % 8u) return b;
    return (b % 8u) * (ntiles / 8u) + b / 8u;
}
// The same with the tiles of an XCD's eighth walked in G x G groups in (a, b): the G^2 tiles of a
// group at column position c are consecutive, then c advances — the workgroups in flight on an
// XCD then share their stencil halos through its L2 in all three dimensions instead of along c
// only.  nb, nc: tiles per dimension (b, c); needs a whole number of groups per XCD, else the
// plain order.  Measured (2^28 particles / 1024^3, fused pass inside bench.py, i.e. from regions
// with gaps): G = 1 (plain) 8.89 ms, 2: 8.83, 4: 8.74, 8: 8.85.  (The pull deposit does not
// gain from it.  Dealing the groups out to the XCDs in a 2-D pattern instead of a contiguous
// eighth of the box per XCD: no gain either — what a clustered box needed was the launch
// ORDER, cgk_tile_order below.)
__device__ __forceinline__ unsigned tile_for_block_grouped(unsigned b, unsigned ntiles,
                                                           unsigned nb, unsigned nc) {
    constexpr unsigned G = 4;
    const unsigned per = ntiles / 8u;          // tiles per XCD
    const unsigned nbg = nb / G;
    if (ntiles % 8u || per % (G * nb * nc) || nb % G) return tile_for_block(b, ntiles);
    const unsigned x = b % 8u, i = b / 8u;     // XCD, position in its walk
    const unsigned q = i % (G * G), col = i / (G * G);   // member of the group, group number
    const unsigned c = col % nc, g = col / nc; // column position, group in the (a/G, b/G) plane
    const unsigned gb = g % nbg, ga = g / nbg;
    const unsigned ta = G * ga + q / G, tb = G * gb + q % G;
    return x * per + (ta * nb + tb) * nc + c;
}

// ---------------------------------------------------------------------------
// heavy tiles first: the launch order when the tiles' populations are far from equal
//
// The kernels of this file give a tile to a workgroup, and a workgroup walks its tile's
// particles in batches of 512: its time follows the population.  With the plain walk every XCD
// owns an eighth of the box (a slab of a-layers), so on a clustered box (bench.py --dist
// clustered: 64 Gaussian clumps with 80 % of 2^28 particles, up to 79,000 in a tile against a
// mean of 1024) the XCDs' loads differ by the clumps each slab happens to hold, and a tile of
// 150 batches that starts late is a tail of its own.  Measured, fused pass, with workgroup ->
// tile tables made on the host (round 4): plain walk 10.8 ms; the heavy tiles of each XCD first,
// by falling population: 9.3; the heavy tiles of the whole box by falling population dealt out
// to the XCDs in turn (block b runs on XCD b % 8), then all others in the walk's order: 8.4 ms
// — the uniform box's time.  ALL tiles by falling population: 11.0 (the sparse tiles lose the
// L2 locality of their halos); heavy and sparse tiles merged at equal fractions of their work:
// 10.0.  Heavy = more than max(1536, 1.5 x mean) particles (thresholds of 1024 .. 3072: 8.39 ..
// 8.45 ms).
//
// A table read by EVERY workgroup costs the uniform box what it gains the clustered one (a
// dependent load in front of a workgroup's first loads, 341 rounds of workgroups: fused pass
// 8.45 -> 8.78 ms with a table that holds the plain walk).  So only the heavy tiles go
// through a list: the grid is `cap` blocks longer; block b < cap takes heavy[b] (or ends at
// once when the list is shorter), block cap + b' takes the tile of the plain walk at b' and
// ends when that tile's place on the list is in front of `cap` — a load that travels with the
// workgroup's first loads instead of in front of them.  How many blocks go in front is the
// launch's choice (a tile beyond them is simply left to the walk): the list's length of the
// last build the host has seen (written to pinned memory, read without waiting), a quarter
// more; none at all, and no loads either, when that was zero — the uniform box launches what
// it always did.
//
// cgk_tile_order makes the list and every tile's place on it on the device from the populations the kernel is about
// to read.  The order among tiles of about the same population is free, so the heavy tiles
// are counted into 61 classes of population (units of a third of the threshold) and take
// their places through one cursor per class: three launches of a few microseconds.  List and
// places of one build describe every tile exactly once whatever the populations are by the
// time they are used, so the gather-kick takes over what the deposit of the same tables made.
// ---------------------------------------------------------------------------
constexpr int ORD_B = 256;       // lanes per workgroup of the order kernels
constexpr int ORD_CLASSES = 64;  // counters: [0] particles, [1] tiles on the list, [2..65] tiles
                                 // per class, [66..129] cursor per class
__global__ __launch_bounds__(1024) void k_order_pops(const unsigned *__restrict__ start, const unsigned *__restrict__ count,
                             unsigned ntiles, bool vec, unsigned *__restrict__ pops,
                             unsigned *__restrict__ counters) {
    const unsigned tile = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned pop = 0;
    if (tile < ntiles) {
        if (count && vec) {  // (a table at a 16-byte boundary: two loads instead of eight)
            const uint4 lo = *(const uint4 *)(count + 8 * (size_t)tile),
                        hi = *(const uint4 *)(count + 8 * (size_t)tile + 4);
            pop = ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
        } else if (count) {
#pragma unroll
            for (int f = 0; f < 8; f++) pop += count[8 * (size_t)tile + f];
        } else {
            pop = start[8 * (size_t)tile + 8] - start[8 * (size_t)tile];
        }
        pops[tile] = pop;
    }
    // (sum over the workgroup, one device atomic each: they execute memory-side one after the
    // other — one per wave, 4096 of them on one word, was 45 of this kernel's 49 us)
    __shared__ unsigned wsum[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pop += __shfl_down(pop, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = pop;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned sum = 0;
        for (unsigned i = 0; i < blockDim.x / 64; i++) sum += wsum[i];
        if (sum) atomicAdd(&counters[0], sum);
    }
}
// class of population into pops[] (0: not heavy), heavy tiles per class
__global__ __launch_bounds__(ORD_B) void k_order_classes(unsigned *__restrict__ pops,
                                                         unsigned ntiles, unsigned floor_,
                                                         unsigned *__restrict__ counters) {
    __shared__ unsigned hist[ORD_CLASSES];
    if (threadIdx.x < ORD_CLASSES) hist[threadIdx.x] = 0;
    __syncthreads();
    const unsigned t = blockIdx.x * ORD_B + threadIdx.x;
    const unsigned mean15 = (unsigned)(((unsigned long long)counters[0] * 3ull) / (2ull * ntiles));
    const unsigned thr = mean15 > floor_ ? mean15 : floor_, unit = thr / 3u ? thr / 3u : 1u;
    if (t < ntiles) {
        const unsigned pop = pops[t];
        unsigned cls = 0;
        if (pop > thr) {
            cls = pop / unit;  // >= 3
            cls = cls < (unsigned)ORD_CLASSES ? cls : (unsigned)ORD_CLASSES - 1u;
            atomicAdd(&hist[cls], 1u);
        }
        pops[t] = cls;
    }
    __syncthreads();
    if (threadIdx.x < ORD_CLASSES && hist[threadIdx.x])
        atomicAdd(&counters[2 + threadIdx.x], hist[threadIdx.x]);
}
// the list by falling class, as far as it reaches; a tile's place on it (none: all ones)
__global__ __launch_bounds__(ORD_B) void k_order_place(const unsigned *__restrict__ cls_of,
                                                       unsigned ntiles, unsigned cap,
                                                       unsigned *__restrict__ counters,
                                                       unsigned *__restrict__ heavy,
                                                       unsigned *__restrict__ rank,
                                                       unsigned *__restrict__ host_n) {
    __shared__ unsigned base[ORD_CLASSES];
    if (threadIdx.x < ORD_CLASSES) {
        // first place of a class: the tiles of the classes above it
        const unsigned mine = counters[2 + threadIdx.x];
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_down(incl, o);
            if ((int)threadIdx.x + o < ORD_CLASSES) incl += u;
        }
        base[threadIdx.x] = incl - mine;
        if (blockIdx.x == 0 && threadIdx.x == 0) *host_n = counters[1] = incl < cap ? incl : cap;
    }
    __syncthreads();
    // places inside a class: counted per workgroup in LDS, one device atomic per workgroup and
    // class it holds (device atomics run memory-side, one after the other)
    __shared__ unsigned mine[ORD_CLASSES], first[ORD_CLASSES];
    if (threadIdx.x < ORD_CLASSES) mine[threadIdx.x] = 0;
    __syncthreads();
    const unsigned t = blockIdx.x * ORD_B + threadIdx.x;
    const unsigned cls = t < ntiles ? cls_of[t] : 0u;
    unsigned local = 0;
    if (cls) local = atomicAdd(&mine[cls], 1u);
    __syncthreads();
    if (threadIdx.x < ORD_CLASSES && mine[threadIdx.x])
        first[threadIdx.x] = atomicAdd(&counters[2 + ORD_CLASSES + threadIdx.x], mine[threadIdx.x]);
    __syncthreads();
    if (t >= ntiles) return;
    unsigned k = ~0u;
    if (cls) {
        k = base[cls] + first[cls] + local;
        if (k < cap) heavy[k] = t;
        else k = ~0u;  // (a list that is full leaves the tile to the walk)
    }
    rank[t] = k;
}

int cgk_tile_order(cg_ctx *c, const unsigned *start, const unsigned *count) {
    if (c->tile_order_mode < 0) {
        // CONCEPT_GPU_TILE_ORDER_MIN: particles from which a tile can be heavy (1536); 0: the
        // plain walk; negative: |value|, also on boxes of a few tiles (the tests)
        const char *env = getenv("CONCEPT_GPU_TILE_ORDER_MIN");
        const int v = env ? atoi(env) : 1536;
        c->tile_order_mode = v == 0 ? 0 : (v < 0 ? 2 : 1);
        c->tile_order_floor = (unsigned)(v < 0 ? -v : v);
    }
    const unsigned ntiles = (unsigned)c->ntiles;
    if (!c->tile_order_mode || ntiles < (c->tile_order_mode == 2 ? 8u : 4096u)) {
        c->tile_order_on = false;
        return 0;
    }
    if (c->tile_order_on && c->tile_order_src[0] == start && c->tile_order_src[1] == count)
        return 0;  // made from these tables by the step's previous kernel
    const unsigned nblk = (ntiles + ORD_B - 1) / ORD_B;
    const size_t ncounters = 2 + 2 * ORD_CLASSES;
    // the list's length: an eighth of the tiles, a multiple of 8 (the walk behind it keeps its
    // blocks on the XCDs it had)
    const unsigned cap = ((ntiles / 8u) + 7u) & ~7u;
    if (!c->tile_order_buf) {
        CG_HIP(hipMalloc(&c->tile_order_buf,
                         sizeof(unsigned) * (2 * (size_t)ntiles + cap + ncounters)));
        c->tile_order = c->tile_order_buf + ntiles;
        c->tile_order_cap = cap;
        // (pinned memory is a convenience: without it every launch takes the full number of
        // blocks in front)
        if (hipHostMalloc((void **)&c->tile_order_seen, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            c->tile_order_seen = nullptr;
        } else {
            *c->tile_order_seen = ~0u;  // no build seen yet
        }
    }
    unsigned *pops = c->tile_order_buf, *heavy = c->tile_order, *counters = heavy + cap;
    unsigned *rank = counters + ncounters, *seen_dev = nullptr;
    if (!c->tile_order_seen ||
        hipHostGetDevicePointer((void **)&seen_dev, c->tile_order_seen, 0) != hipSuccess) {
        (void)hipGetLastError();
        seen_dev = counters + 1;  // (the kernel's own word: nothing for the host to see)
    }
    CG_HIP(hipMemsetAsync(counters, 0, sizeof(unsigned) * ncounters, c->stream));
    hipLaunchKernelGGL(k_order_pops, dim3((ntiles + 1023u) / 1024u), dim3(1024), 0, c->stream, start, count, ntiles,
                       ((uintptr_t)count & 15u) == 0, pops, counters);
    hipLaunchKernelGGL(k_order_classes, dim3(nblk), dim3(ORD_B), 0, c->stream, pops, ntiles,
                       c->tile_order_floor, counters);
    hipLaunchKernelGGL(k_order_place, dim3(nblk), dim3(ORD_B), 0, c->stream, pops, ntiles, cap,
                       counters, heavy, rank, seen_dev);
    CG_LAUNCH_CHECK();
    c->tile_order_on = true;
    c->tile_order_src[0] = start;
    c->tile_order_src[1] = count;
    return 0;
}
// what a tile kernel is launched with (list == null: the plain walk, no extra blocks)
struct TileOrder {
    const unsigned *list;        // heavy tiles, *n of them
    const unsigned *n;
    const unsigned *rank;        // per tile: its place on the list (all ones: not on it)
    unsigned cap;                // blocks in front of the walk (a multiple of 8)
};
static TileOrder tile_order_args(const cg_ctx *c) {
    TileOrder o{};
    if (!c->tile_order_on) return o;
    const unsigned seen = c->tile_order_seen ? *(volatile unsigned *)c->tile_order_seen : ~0u;
    if (seen == 0) return o;  // no heavy tiles the last time we looked: the plain launch
    unsigned front = c->tile_order_cap;
    if (seen != ~0u && seen + seen / 4u + 64u < front) front = (seen + seen / 4u + 64u + 7u) & ~7u;
    o.list = c->tile_order;
    o.n = c->tile_order + c->tile_order_cap + 1;
    o.rank = c->tile_order + c->tile_order_cap + 2 + 2 * ORD_CLASSES;
    o.cap = front;
    return o;
}

// ---------------------------------------------------------------------------
// tiled deposit, owner-computes ("pull") form
//
// Workgroup of tile t accumulates in LDS every contribution to the T^3 cells
// it owns: from its own particles (all 8 buckets) and from the particles of
// its 7 lower neighbour tiles whose CIC cloud reaches into t — exactly the
// buckets f of neighbour d (d = which dimensions step back one tile) with
// (f & d) == d, found through the tile table without touching other
// particles.  Then the tile is written once with plain, line-aligned stores:
// no global atomics, no zero-fill pass, ~1.2x particle reads (the boundary
// buckets are re-read from L2 by the neighbour).
// ---------------------------------------------------------------------------
template <int T, bool ACCUMULATE>
__global__ __launch_bounds__(512) void k_deposit_cic_pull(
    const double *__restrict__ pos, const unsigned *__restrict__ table,
    const unsigned *__restrict__ count /* populations of regions with gaps, or null: dense */,
    double *__restrict__ mesh, i64 N, i64 ny, i64 pad, int g, int ntx, int nt, unsigned nblocks,
    XMap xm, CicGeom geo, double contribution, TileOrder ord, unsigned nordered) {
    // tiles: ntx rows along x (this domain's), nt along y and z.  An x-slab domain has
    // one extra row ta == ntx: the ghost layer that receives the CIC clouds sticking out
    // of the last owned layer (sent to the next domain and added there).
    constexpr int NL = T * T * T;
    __shared__ double lds[NL];
    __shared__ unsigned seg_beg[64], seg_end_prefix[65];
    // (ord: heavy tiles first, see cgk_tile_order; it knows the first `nordered` tiles — those
    // that hold particles; a slab domain's ghost row behind them is walked as ever)
    unsigned tile;
    if (ord.list && blockIdx.x < ord.cap) {
        if (blockIdx.x >= (unsigned)__builtin_amdgcn_readfirstlane((int)*ord.n)) return;
        tile = (unsigned)__builtin_amdgcn_readfirstlane((int)ord.list[blockIdx.x]);
    } else {
        const unsigned b = blockIdx.x - (ord.list ? ord.cap : 0u);
        tile = tile_for_block(b, nblocks);
        if (ord.list && tile < nordered &&
            (unsigned)__builtin_amdgcn_readfirstlane((int)ord.rank[tile]) < ord.cap)
            return;
    }
    const int tc = tile % nt, tb = (tile / nt) % nt, ta = tile / (nt * nt);
    const int Ni = (int)N;
    const int T0a = (int)xm.x0 + ta * T, T0b = tb * T, T0c = tc * T;
    for (int idx = threadIdx.x; idx < NL; idx += 512) lds[idx] = 0;
    if (threadIdx.x < 64) {
        // segment (d, f): bucket f of the neighbour d steps back
        int d = threadIdx.x >> 3, f = threadIdx.x & 7;
        int na = ta - ((d >> 2) & 1), nb = tb - ((d >> 1) & 1), nc = tc - (d & 1);
        if (xm.periodic) na = na < 0 ? na + ntx : na;
        nb = nb < 0 ? nb + nt : nb;
        nc = nc < 0 ? nc + nt : nc;
        // slab domain: the row below the first one is another domain's (its clouds
        // arrive through the halo add), the ghost row itself holds no particles
        bool have = (f & d) == d && na >= 0 && na < ntx;
        unsigned e = have ? ((unsigned)((na * nt + nb) * nt + nc)) * 8u + (unsigned)f : 0u;
        unsigned b0 = table[e], cnt = have ? (count ? count[e] : table[e + 1] - b0) : 0u;
        // inclusive scan of cnt over the 64 lanes of this (first) wave
        unsigned incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned v = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o) incl += v;
        }
        seg_beg[threadIdx.x] = b0;
        seg_end_prefix[threadIdx.x + 1] = incl;
        if (threadIdx.x == 0) seg_end_prefix[0] = 0;
    }
    __syncthreads();
    const unsigned total = seg_end_prefix[64];
    for (unsigned q = threadIdx.x; q < total; q += 512) {
        // segment of flat index q: largest s with prefix[s] <= q
        int s = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (seg_end_prefix[s + step] <= q) s += step;
        const i64 p = (i64)seg_beg[s] + (q - seg_end_prefix[s]);
        Cic1 cx = cic1(pos[3 * p + 0], geo.off[0], geo.scale);
        Cic1 cy = cic1(pos[3 * p + 1], geo.off[1], geo.scale);
        Cic1 cz = cic1(pos[3 * p + 2], geo.off[2], geo.scale);
        // mesh.py:5142-5155: ((w_x*contribution)*w_y)*w_z
        double wi[2] = {cx.w0 * contribution, cx.w1 * contribution};
        double wy[2] = {cy.w0, cy.w1}, wz[2] = {cz.w0, cz.w1};
        // lower cell relative to this tile: -1 .. T-1 (periodic)
        int la = wrap(cx.index - g, Ni) - T0a, lb = wrap(cy.index - g, Ni) - T0b,
            lc = wrap(cz.index - g, Ni) - T0c;
        la = (xm.periodic && la >= T) ? la - Ni : la;
        lb = lb >= T ? lb - Ni : lb;
        lc = lc >= T ? lc - Ni : lc;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                double wij = wi[i] * wy[j];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    int a = la + i, b = lb + j, c = lc + k;
                    if (a >= 0 && a < T && b >= 0 && b < T && c >= 0 && c < T)
                        atomicAdd(&lds[(a * T + b) * T + c], wij * wz[k]);
                }
            }
    }
    __syncthreads();
    const int a_end = (ta == ntx) ? 1 : T;  // of the ghost row only its first layer exists
    for (int idx = threadIdx.x; idx < NL; idx += 512) {
        int c = idx % T, b = (idx / T) % T, a = idx / (T * T);
        if (a >= a_end) break;
        i64 layer = xm.periodic ? (i64)(T0a + a) : ((i64)ta * T + a + xm.G);
        double *dst = mesh + (layer * ny + (T0b + b)) * pad + (T0c + c);
        if (ACCUMULATE) *dst += lds[idx];  // cells are exclusively owned: no atomics needed
        else *dst = lds[idx];
    }
}

template <int T>
static int launch_deposit(cg_ctx *c, const double *pos, const unsigned *table,
                          const unsigned *count, double contribution, int accumulate) {
    int rows = c->tiles.ntx + (c->xmap.periodic ? 0 : 1);
    unsigned nb = (unsigned)((i64)rows * c->tiles.nty * c->tiles.ntz);
    // the populations are new with every deposit of a step: the order is made here and the
    // gather of the same tables takes it over
    c->tile_order_on = false;
    if (cgk_tile_order(c, table, count)) return 1;
    const TileOrder order = tile_order_args(c);
    const unsigned nordered = (unsigned)c->ntiles;
    if (accumulate)
        hipLaunchKernelGGL((k_deposit_cic_pull<T, true>), dim3(nb + order.cap), dim3(512), 0, c->stream, pos,
                           table, count, c->mesh, c->N, c->ny, c->pad, c->p.nghosts, c->tiles.ntx, c->tiles.nty,
                           nb, c->xmap, c->geom_deposit, contribution, order, nordered);
    else
        hipLaunchKernelGGL((k_deposit_cic_pull<T, false>), dim3(nb + order.cap), dim3(512), 0, c->stream, pos,
                           table, count, c->mesh, c->N, c->ny, c->pad, c->p.nghosts, c->tiles.ntx, c->tiles.nty,
                           nb, c->xmap, c->geom_deposit, contribution, order, nordered);
    return 0;
}

int cgk_deposit_cic_tiled(cg_ctx *c, const double *pos, i64 n, const unsigned *tile_offset,
                          const unsigned *count, double contribution, int accumulate) {
    (void)n;
    const int T = c->tiles.tx;
    switch (T) {
        case 16: if (launch_deposit<16>(c, pos, tile_offset, count, contribution, accumulate)) return 1; break;
        case 8: if (launch_deposit<8>(c, pos, tile_offset, count, contribution, accumulate)) return 1; break;
        case 4: if (launch_deposit<4>(c, pos, tile_offset, count, contribution, accumulate)) return 1; break;
        case 2: if (launch_deposit<2>(c, pos, tile_offset, count, contribution, accumulate)) return 1; break;
        default: cg_set_error("cgk_deposit_cic_tiled: tile extent %d", T); return 1;
    }
    CG_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// tiled gather + finite difference + kick
// ---------------------------------------------------------------------------
// One force-cell value per dimension exactly as diff_domaingrid forms it
// (mesh.py:4966-4981); P(da, db, dc) reads the potential at an offset from
// the cell.
template <int ORDER, class P>
__device__ __forceinline__ void force_cell(const P &phi, double c1, double c2, double &fx,
                                           double &fy, double &fz) {
    if (ORDER == 2) {
        fx = c1 * (phi(1, 0, 0) - phi(-1, 0, 0));
        fy = c1 * (phi(0, 1, 0) - phi(0, -1, 0));
        fz = c1 * (phi(0, 0, 1) - phi(0, 0, -1));
    } else {
        fx = c1 * (phi(1, 0, 0) - phi(-1, 0, 0)) - c2 * (phi(2, 0, 0) - phi(-2, 0, 0));
        fy = c1 * (phi(0, 1, 0) - phi(0, -1, 0)) - c2 * (phi(0, 2, 0) - phi(0, -2, 0));
        fz = c1 * (phi(0, 0, 1) - phi(0, 0, -1)) - c2 * (phi(0, 0, 2) - phi(0, 0, -2));
    }
}

// PREP: additionally histogram where every particle will be after the NEXT drift
// (pos + mom_new*next_dtm, the arithmetic of k_tile_histogram<true>) into `count`, so
// that cg_drift_sort can skip its own histogram pass over pos and mom.
struct PrepArgs {
    double next_dtm, boxsize;
    CicGeom geo_sort;
    TileGeom tiles;
    unsigned *count;
    // x-slab domains: the particles that will leave the slab with the next drift (no tile
    // key) are listed — emig_idx[0 .. min(*emig_count, emig_cap)) — so that the exchange
    // needs no pass of its own to find them (cg_set_emigrant_list; null = off)
    i64 *emig_idx;
    unsigned *emig_count;
    i64 emig_cap;
    // MODE 2 (fused kick + drift + scatter): the drifted particles with their kicked momenta go
    // straight to their place in the next tile order.  start_out[k] = first slot of the region
    // reserved for (tile, bucket) k (capacities predicted from the present populations,
    // cg_predict_regions), count_out[k] = slots taken so far (zeroed before the launch; it ends
    // as the new population).  A run that does not fit sets CG_ERR_BUCKET_OVERFLOW.
    const unsigned *start_out;
    unsigned *count_out;
    double *pos_out, *mom_out;
    const i64 *ids_in;
    i64 *ids_out;
    const i64 *aux_in;  // a second 64-bit column travelling with the particles (or null)
    i64 *aux_out;
    i64 out_capacity;   // rows of pos_out / mom_out (/ ids_out, aux_out)
    unsigned *err_flags;
    // input populations when the input itself is in regions with gaps (null: dense tile order)
    const unsigned *count_in;
    // x-slab domains: particles leaving the slab go to this row buffer (8 doubles each)
    double *emig_rows;
    unsigned *emig_rows_count;
    i64 emig_rows_cap;
    // heavy tiles first (cgk_tile_order; list null: the plain walk)
    TileOrder order;
    // MODE 2: the sum of |mom|^2 over the KICKED momenta, one partial per wavefront
    // (mom2_out[8 * block + wave]; zeroed before the launch; cg_set_momentum_sum; null = off)
    double *mom2_out;
};

// sum over the 64 lanes of a wave, in DPP moves (the lanes' order is fixed: reproducible);
// returned in lane 63
__device__ __forceinline__ double gk_wave_sum(double v) {
#define GK_DPP_ADD(ctrl, rows)                                                                    \
    do {                                                                                          \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rows, 0xf, false); \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rows, 0xf, false); \
        v += __hiloint2double(hi_, lo_);                                                          \
    } while (0)
    GK_DPP_ADD(0x111, 0xf);  // row_shr 1, 2, 4, 8: inclusive scan inside the rows of 16 lanes
    GK_DPP_ADD(0x112, 0xf);
    GK_DPP_ADD(0x114, 0xf);
    GK_DPP_ADD(0x118, 0xf);
    GK_DPP_ADD(0x142, 0xa);  // row_bcast 15, 31: the row totals forward
    GK_DPP_ADD(0x143, 0xc);
#undef GK_DPP_ADD
    return v;
}

// LDS of the staged block.  A workgroup's allocation is rounded up to 1280 B and three workgroups
// per CU need <= 53,760 B each (measured with a probe allocation: 53,760 B runs three, 54,016 B
// two).  T = 16, order 2: the full 19^3 block is 54,872 B.  The 6-point stencil never reads an
// entry with two or three coordinates on the block's boundary, so the two boundary PLANES
// a = 0 and a = E - 1 are kept compact — only their 17 x 17 interior, row stride 17, behind
// the 17 ordinary planes a = 1 .. 17 — 6137 + 2 * 289 = 6715 doubles = 53,720 B.  Only the
// x-difference of a particle in the first (last) layer of cells of the tile reaches a compact
// plane: one select per (i, j) on the plane offset.  Other tile sizes and order 4 keep the
// plain block (they fit two workgroups either way).
template <int ORDER, int T>
struct GatherLds {
    static constexpr int H = ORDER / 2, E = T + 1 + 2 * H, PL = E * E;
    static constexpr bool compact = (ORDER == 2 && T == 16);
    static constexpr int main_planes = compact ? E - 2 : E;       // planes kept as E x E
    static constexpr int IN = E - 2;                                // interior of a compact plane
    static constexpr int P_LO = main_planes * PL;                   // compact plane a = 0
    static constexpr int P_HI = P_LO + IN * IN;                     // compact plane a = E - 1
    static constexpr int doubles = compact ? P_HI + IN * IN : E * E * E;
    // index of block entry (a, b, c) for a in the main planes (compact: a = 1 .. E - 2)
    __device__ static constexpr int main(int a, int b, int c) {
        return ((a - (compact ? 1 : 0)) * E + b) * E + c;
    }
};
// Wavefronts per SIMD the register allocation aims at (T = 16, order 2; 512-lane workgroups: 6
// waves per SIMD = three workgroups per CU, which the compact LDS block above admits).  Measured
// at 2^28 particles / 1024^3: the fused pass (MODE 2) 9.9 -> 8.8 ms with three instead of two;
// it fits 80 registers without a spill once the x loop of its stencil is not unrolled and the
// particle indices are 32-bit.  A build that spills loses more than the third workgroup gives
// (MODE 2 with 4 spilled registers: 10.0 ms; the histogramming variant, MODE 1, squeezed into 80
// spills 72 B per lane: 9.0 instead of 7.3 ms) — so MODE 1 stays at 4.
template <int ORDER, int T, int MODE>
constexpr int gk_waves() {
    if (ORDER != 2 || T != 16) return 4;
    return MODE == 1 ? 4 : 6;
}

// MODE 0: gather + kick in place.  1: also histogram the tile keys after the next drift (PREP).
// 2: kick, drift and scatter into the next tile order in one pass (nothing written in place).
// MOM2 (MODE 2 only): the pass also leaves the sum of |mom|^2 (prep.mom2_out) — its own
// instantiation, so that the pass without the sum is the code it was before the sum existed.
template <int ORDER, int T, int MODE, bool MOM2 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(
    gk_waves<ORDER, T, MODE>(), 8))) void k_gather_kick_tiled(
    const double *__restrict__ pos, double *__restrict__ mom,
    const unsigned *__restrict__ tile_offset, const double *__restrict__ mesh, i64 N, i64 ny,
    i64 pad, int g, int nt, unsigned ntiles, XMap xm, CicGeom geo, double c1, double c2,
    double factor, PrepArgs prep) {
    constexpr int H = ORDER / 2;
    constexpr int E = T + 1 + 2 * H;  // cells [T0-H, T0+T+H]
    using BL = GatherLds<ORDER, T>;
    constexpr bool COMPACT = BL::compact;
    extern __shared__ double lds[];
    constexpr bool PREP = MODE == 1, FUSED = MODE == 2;
    unsigned tile;
    unsigned place = ~0u;  // of the walk's tile on the list of heavy tiles
    if (prep.order.list && blockIdx.x < prep.order.cap) {
        if (blockIdx.x >= (unsigned)__builtin_amdgcn_readfirstlane((int)*prep.order.n)) return;
        tile = (unsigned)__builtin_amdgcn_readfirstlane((int)prep.order.list[blockIdx.x]);
    } else {
        tile = tile_for_block_grouped(blockIdx.x - (prep.order.list ? prep.order.cap : 0u),
                                      ntiles, (unsigned)nt, (unsigned)nt);
        // (no wait here: the byte arrives with the tile's offsets)
        if (prep.order.list) place = prep.order.rank[tile];
    }
    // the tile's particles: dense tile order -> one range; regions with gaps (prep.count_in)
    // -> its 8 buckets' (start, population), walked as one flat index
    // (FUSED only: the plain kernel's LDS block is sized so that three workgroups fit a CU
    // to the byte — nothing may be added to it)
    // They live in entries of the staged block that no stencil reads and the staging leaves
    // alone: compact layout — the corners (a, 0, 0) of the planes a = 1 .. 9, two words each;
    // plain layout — rows b = E - 1 (8 words) and b = 0 (9 words) of plane a = 0 (a particle's
    // stencil reaches plane 0 only at rows H .. E-1-H; E >= 5 doubles per row).
    constexpr int SEG_AT = (E - 1) * E;   // plain layout: block index of row (0, E - 1)
    static_assert(E * 8 >= 9 * 4, "segment tables do not fit");
    auto seg_beg = [&](int f) -> unsigned & {
        return COMPACT ? ((unsigned *)(lds + BL::main(1 + (f >> 1), 0, 0)))[f & 1]
                       : ((unsigned *)(lds + SEG_AT))[f];
    };
    auto seg_pre = [&](int f) -> unsigned & {
        return COMPACT ? ((unsigned *)(lds + BL::main(5 + (f >> 1), 0, 0)))[f & 1]
                       : ((unsigned *)lds)[f];
    };
    const bool gapped = FUSED && prep.count_in != nullptr;
    // (particle indices fit 32 bits: the tile tables are uint32)
    typedef unsigned pidx;
    pidx beg = tile_offset[8 * tile], end = tile_offset[8 * tile + 8];
    if (gapped) {
        if (threadIdx.x < 64) {
            const int f = threadIdx.x & 7;
            unsigned cnt = prep.count_in[8 * tile + f];
            unsigned incl = cnt;
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                unsigned v = __shfl_up(incl, o);
                if (f >= o) incl += v;
            }
            if (threadIdx.x < 8) {
                seg_beg(f) = tile_offset[8 * tile + f];
                seg_pre(f + 1) = incl;
                if (f == 0) seg_pre(0) = 0;
            }
        }
        // the total by scalar loads (the tile is uniform): the table above is published by the
        // barrier that ends the staging, which starts without waiting for it (a barrier here
        // cost the pass from regions with gaps 0.2 ms against the dense order)
        unsigned total = 0;
#pragma unroll
        for (int f = 0; f < 8; f++)
            total += (unsigned)__builtin_amdgcn_readfirstlane((int)prep.count_in[8 * tile + f]);
        beg = 0;
        end = total;
    }
    // (uniform; a tile with a place among the blocks in front was theirs)
    if (beg == end || (unsigned)__builtin_amdgcn_readfirstlane((int)place) < prep.order.cap) return;
    // FUSED: a workgroup is a chain of dependent round trips (stage the block, load a batch of
    // particles, gather, reserve places, store) and only two fit a CU: the particle loads of a
    // batch are issued one stage ahead — the first batch's under the staging of the block,
    // the next one's under the arithmetic of the present one
    pidx pre_p = 0;
    bool pre_valid = false;
    double pre_x = 0, pre_y = 0, pre_z = 0, pre_mx = 0, pre_my = 0, pre_mz = 0;
    auto fetch = [&](pidx pbase) {
        pidx p = pbase + threadIdx.x;
        pre_valid = p < end;
        if (gapped && pre_valid) {  // flat index -> slot of its bucket's region
            // bucket 0 (the tile's interior, (15/16)^3 of its particles) first: a wave that
            // lies in it entirely — most do — skips the search (uniform branch)
            if (p < seg_pre(1)) {
                p = seg_beg(0) + p;
            } else {
                int f = 0;
#pragma unroll
                for (int step = 4; step > 0; step >>= 1)
                    if (seg_pre(f + step) <= (unsigned)p) f += step;
                p = seg_beg(f) + (p - seg_pre(f));
            }
        }
        pre_p = p;
        if (pre_valid) {
            // a record is 24 B at an 8-byte boundary: one 16-byte and one 8-byte access (global
            // memory takes vector accesses at their element's alignment) — 4 requests per
            // particle instead of 6
            const d2u8 a = *(const d2u8 *)(pos + 3 * (i64)p), b = *(const d2u8 *)(mom + 3 * (i64)p);
            pre_x = a.x, pre_y = a.y, pre_z = pos[3 * (i64)p + 2];
            pre_mx = b.x, pre_my = b.y, pre_mz = mom[3 * (i64)p + 2];
        }
    };
    // (Issuing a batch's particle loads one stage ahead — the first batch's under the staging of
    // the block, the next one's under the arithmetic of the present one — costs 4 spilled
    // registers at the 80 that three workgroups per CU allow: 9.4 against 8.4 ms.)
    const int tc = tile % nt, tb = (tile / nt) % nt, ta = tile / (nt * nt);
    const int Ni = (int)N;
    const int T0a = (int)xm.x0 + ta * T, T0b = tb * T, T0c = tc * T;
    {
        // Stage the potential tile + stencil halo.  Wave w takes the planes a = w, w+8, ...
        // of the E^3 block: the layer index and its 64-bit row base are wave-uniform
        // (scalar unit), a lane's (row, column) offsets inside a plane are 32-bit and the
        // same for every plane, so a load costs one instruction (scalar base + lane
        // offset) instead of the ~80 of per-element 64-bit index arithmetic this loop
        // used to spend — that arithmetic, not the stencil, was two thirds of the kernel's
        // VALU work (rocprofv3: 1060 VALU instructions per particle, VALUBusy 78 %).
        // All loads of a lane are issued before its first LDS store.
        constexpr int PL = E * E;                // points per plane
        constexpr int NP = (PL + 63) / 64;       // 64-lane passes per plane
        constexpr int NA = (E + 7) / 8;          // planes per wave
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int b0 = T0b - H, c0 = T0c - H;
        const bool seam = b0 < 0 || c0 < 0 || b0 + E > Ni || c0 + E > Ni;
        unsigned off[NP];  // gj*pad + gk
        int cmp[COMPACT ? NP : 1];  // compact planes: (b-1)*IN + (c-1) of the interior, else -1
#pragma unroll
        for (int q = 0; q < NP; q++) {
            int i2 = lane + 64 * q;
            int b = i2 / E, c = i2 - b * E;
            int gj = b0 + b, gk = c0 + c;
            if (seam) {  // (tile-uniform) the block reaches across the periodic seam in y or z
                gj += (gj < 0 ? Ni : 0) - (gj >= Ni ? Ni : 0);
                gk += (gk < 0 ? Ni : 0) - (gk >= Ni ? Ni : 0);
            }
            off[q] = (unsigned)gj * (unsigned)pad + (unsigned)gk;
            if (COMPACT)
                cmp[q] = (b >= 1 && b <= E - 2 && c >= 1 && c <= E - 2)
                             ? (b - 1) * BL::IN + (c - 1) : -1;
        }
        double v[NA][NP];
#pragma unroll
        for (int s = 0; s < NA; s++) {
            const int a = __builtin_amdgcn_readfirstlane(wave) + 8 * s;  // wave-uniform
            if (a < E) {
                const double *plane = mesh + cg_xlayer(xm, (i64)(T0a - H + a), N) * ny * pad;
                const bool edge = COMPACT && (a == 0 || a == E - 1);  // (uniform)
#pragma unroll
                for (int q = 0; q < NP; q++) {
                    const bool want = (PL % 64 == 0 || lane + 64 * q < PL) &&
                                      (!edge || cmp[COMPACT ? q : 0] >= 0);
                    if (want) v[s][q] = plane[off[q]];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NA; s++) {
            const int a = __builtin_amdgcn_readfirstlane(wave) + 8 * s;
            if (a >= E) continue;
            if (COMPACT && (a == 0 || a == E - 1)) {
                // a compact boundary plane: its interior only (the branch is wave-uniform)
                const int base = a == 0 ? BL::P_LO : BL::P_HI;
#pragma unroll
                for (int q = 0; q < NP; q++)
                    if ((PL % 64 == 0 || lane + 64 * q < PL) && cmp[COMPACT ? q : 0] >= 0)
                        lds[base + cmp[COMPACT ? q : 0]] = v[s][q];
            } else {
                const int base = COMPACT ? BL::main(a, 0, 0) : a * PL;
#pragma unroll
                for (int q = 0; q < NP; q++) {
                    const int i2 = lane + 64 * q;
                    // the entries that hold the segment tables (FUSED) are left alone
                    const bool seg = FUSED && (COMPACT ? (i2 == 0 && a <= 9)
                                                       : (a == 0 && (i2 >= SEG_AT || i2 < E)));
                    if ((PL % 64 == 0 || i2 < PL) && !seg) lds[base + i2] = v[s][q];
                }
            }
        }
    }
    __syncthreads();
    int m2_lo = 0, m2_hi = 0;  // (FUSED, prep.mom2_out: this wave's sum of |mom|^2, wave-uniform)
    for (pidx pbase = beg; pbase < end; pbase += 512) {
        pidx p = pbase + threadIdx.x;
        bool pvalid = p < end;
        double px = 0, py = 0, pz = 0, qx = 0, qy = 0, qz = 0;
        if (FUSED) {
            fetch(pbase);
            p = pre_p;
            pvalid = pre_valid;
            px = pre_x, py = pre_y, pz = pre_z, qx = pre_mx, qy = pre_my, qz = pre_mz;
        } else if (pvalid) {
            px = pos[3 * (i64)p + 0], py = pos[3 * (i64)p + 1], pz = pos[3 * (i64)p + 2];
        }
        unsigned next_key = kNoTile;
        double nx = 0, ny_ = 0, nz = 0, n0 = 0, n1 = 0, n2 = 0;  // FUSED: what travels
        if (pvalid) {
        Cic1 cx = cic1(px, geo.off[0], geo.scale);
        Cic1 cy = cic1(py, geo.off[1], geo.scale);
        Cic1 cz = cic1(pz, geo.off[2], geo.scale);
        // (a position in [0, boxsize) has its lower cell in [-1, N - 1]: only the lower wrap)
        int ga = cx.index - g, gb = cy.index - g, gc = cz.index - g;
        ga += ga < 0 ? Ni : 0;
        gb += gb < 0 ? Ni : 0;
        gc += gc < 0 ? Ni : 0;
        int la = ga - T0a, lb = gb - T0b, lc = gc - T0c;
        double wx[2] = {cx.w0, cx.w1}, wy[2] = {cy.w0, cy.w1}, wz[2] = {cz.w0, cz.w1};
        double val[3] = {0, 0, 0};
        if (la >= 0 && la < T && lb >= 0 && lb < T && lc >= 0 && lc < T) {
            const double *base = lds + BL::main(la + H, lb + H, lc + H);
            // compact layout: offsets from the central cell (la+1+i, lb+1+j, lc+1+k) to its
            // x-neighbours when those lie in a compact boundary plane (row stride IN instead of
            // E: the offset depends on j through -2j)
            // (x_hi - x_lo is a constant: one register for both)
            const int x_lo = BL::P_LO + lb * BL::IN + lc - BL::main(1, lb + 1, lc + 1);
            constexpr int X_HI_LO = BL::P_HI - BL::P_LO - (BL::main(E - 2, 0, 0) - BL::main(1, 0, 0));
            // FUSED: the x loop stays a loop (the 24 stencil reads of one x at a time): 80
            // registers without a spill, i.e. three workgroups per CU
#pragma unroll PREP ? 2 : 1
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    double wij = wx[i] * wy[j];
                    const int xm = (COMPACT && la + i == 0) ? x_lo - 2 * j : -E * E;
                    const int xp = (COMPACT && la + i == T) ? x_lo + X_HI_LO - 2 * j : E * E;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const double *cell = base + (i * E + j) * E + k;
                        auto phi = [&](int da, int db, int dc) {
                            if (COMPACT && da != 0) return cell[da < 0 ? xm : xp];
                            return cell[(da * E + db) * E + dc];
                        };
                        double fx, fy, fz;
                        force_cell<ORDER>(phi, c1, c2, fx, fy, fz);
                        double w = wij * wz[k];
                        val[0] += fx * w;
                        val[1] += fy * w;
                        val[2] += fz * w;
                    }
                }
        } else if (FUSED) {
            // The fused pass reads what the last fused pass, cg_region_insert or the tile sort
            // wrote: every particle is in its tile.  One that is not (positions changed since)
            // gets no kick here and raises CG_ERR_NOT_IN_TILE; the caller repeats the step on
            // the exact path, whose kernels read the mesh directly for such particles (below:
            // 24 more registers that this variant does without, 9.9 -> 9.6 ms).
            atomicOr(prep.err_flags, (unsigned)CG_ERR_NOT_IN_TILE);
        } else {
            // outside its tile (array drifted since the sort): read the mesh directly.  Rare:
            // the addresses are formed per read and the loops stay loops, so that this branch
            // holds no registers beside the staged path's (index tables per dimension cost the
            // order-4 kernel 12 B of scratch per lane)
#pragma unroll 1
            for (int i = 0; i < 2; i++)
#pragma unroll 1
                for (int j = 0; j < 2; j++) {
                    double wij = wx[i] * wy[j];
#pragma unroll 1
                    for (int k = 0; k < 2; k++) {
                        auto phi = [&](int da, int db, int dc) {
                            return mesh[cg_xlayer(xm, (i64)(ga + i + da), N) * ny * pad +
                                        (i64)wrap(gb + j + db, Ni) * pad + wrap(gc + k + dc, Ni)];
                        };
                        double fx, fy, fz;
                        force_cell<ORDER>(phi, c1, c2, fx, fy, fz);
                        double w = wij * wz[k];
                        val[0] += fx * w;
                        val[1] += fy * w;
                        val[2] += fz * w;
                    }
                }
        }
        if (factor != 1) {
            val[0] *= factor;
            val[1] *= factor;
            val[2] *= factor;
        }
        if (!FUSED) qx = mom[3 * (i64)p + 0], qy = mom[3 * (i64)p + 1], qz = mom[3 * (i64)p + 2];
        const double m0 = qx + val[0], m1 = qy + val[1], m2 = qz + val[2];
        if (!FUSED) {
            mom[3 * (i64)p + 0] = m0;
            mom[3 * (i64)p + 1] = m1;
            mom[3 * (i64)p + 2] = m2;
        }
        if (PREP || FUSED) {
            // Component.drift (species.py:2194-2196) of the kicked particle
            nx = ref_mod(px + m0 * prep.next_dtm, prep.boxsize);
            ny_ = ref_mod(py + m1 * prep.next_dtm, prep.boxsize);
            nz = ref_mod(pz + m2 * prep.next_dtm, prep.boxsize);
            n0 = m0;
            n1 = m1;
            n2 = m2;
            next_key = tile_of(nx, ny_, nz, prep.geo_sort, g, N, prep.tiles, xm.x0);
        }
        }
        if (PREP) {
            int rs, rl;
            wave_runs(next_key, threadIdx.x & 63, rs, rl);
            if ((int)(threadIdx.x & 63) == rs && next_key != kNoTile)
                atomicAdd(&prep.count[next_key], (unsigned)rl);
            if (prep.emig_idx && pvalid && next_key == kNoTile) {
                const unsigned slot = atomicAdd(prep.emig_count, 1u);
                if ((i64)slot < prep.emig_cap) prep.emig_idx[slot] = p;
            }
        }
        if (FUSED) {
            // the scatter of cg_particles.hip k_tile_scatter: one atomic per run of equal keys
            // among consecutive lanes, runs stored cooperatively
            const int lane = threadIdx.x & 63;
            int rs, rl;
            wave_runs(next_key, lane, rs, rl);
            unsigned first = kNoTile;
            if (lane == rs && next_key != kNoTile) {
                // (a region that reaches beyond the output arrays has no room at all: the
                // populations outgrew what the arrays were sized for)
                const unsigned o0 = prep.start_out[next_key], o1 = prep.start_out[next_key + 1],
                               room = (i64)o1 <= prep.out_capacity ? o1 - o0 : 0u;
                const unsigned local = atomicAdd(&prep.count_out[next_key], (unsigned)rl);
                if (local + (unsigned)rl > room) atomicOr(prep.err_flags, 2u);  // overflow
                else first = o0 + local;
            }
            first = __shfl(first, rs);
            const bool valid = pvalid && next_key != kNoTile && first != kNoTile;
            // every lane stores its own record (24-byte stride; the runs of a wave are
            // consecutive records, L2 merges the partial lines).  The cooperative run store of
            // the stand-alone scatter (store_run there: contiguous stores through 36 lane
            // permutes per particle) is slower HERE — 10.4 vs 9.9 ms — because the permutes go
            // through the LDS crossbar, which the stencil reads already keep busy.
            if (valid) {
                const i64 q = 3 * ((i64)first + (lane - rs));
                d2u8 a, b;
                a.x = nx, a.y = ny_, b.x = n0, b.y = n1;
                *(d2u8 *)(prep.pos_out + q) = a;
                prep.pos_out[q + 2] = nz;
                *(d2u8 *)(prep.mom_out + q) = b;
                prep.mom_out[q + 2] = n2;
            }
            if (valid && prep.ids_in) prep.ids_out[(i64)first + (lane - rs)] = prep.ids_in[p];
            if (valid && prep.aux_in) prep.aux_out[(i64)first + (lane - rs)] = prep.aux_in[p];
            if (prep.emig_rows && pvalid && next_key == kNoTile) {
                // leaves this domain's slab: exchange() (communication.py:135-517) takes it
                // from here, already kicked and drifted
                const unsigned slot = atomicAdd(prep.emig_rows_count, 1u);
                if ((i64)slot < prep.emig_rows_cap) {
                    double *row = prep.emig_rows + 8 * (i64)slot;
                    row[0] = nx, row[1] = ny_, row[2] = nz, row[3] = n0, row[4] = n1, row[5] = n2;
                    row[6] = prep.ids_in ? __longlong_as_double(prep.ids_in[p]) : 0.0;
                    row[7] = prep.aux_in ? __longlong_as_double(prep.aux_in[p]) : 0.0;
                } else {  // more leavers than the row buffer holds: the pass must be repeated
                    atomicOr(prep.err_flags, (unsigned)CG_ERR_BUCKET_OVERFLOW);
                }
            }
            if (MOM2) {
                // analysis.measure(component, 'v_rms') (analysis.py:3902-3910) of the momenta
                // this pass leaves: the sum of a batch's |mom|^2 over the wave's lanes, carried
                // on in scalar registers (the values are dead by now: no register of the
                // pass's 80 is held for it — a per-lane accumulator cost the third workgroup
                // per CU its place)
                const double e = pvalid ? (n0 * n0 + n1 * n1) + n2 * n2 : 0.0;
                const double t = gk_wave_sum(e);
                const double acc = __hiloint2double(m2_hi, m2_lo) +
                                   __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), 63),
                                                    __builtin_amdgcn_readlane(__double2loint(t), 63));
                m2_lo = __builtin_amdgcn_readfirstlane(__double2loint(acc));
                m2_hi = __builtin_amdgcn_readfirstlane(__double2hiint(acc));
            }
        }
    }
    // (the wave's index from a scalar read of the lane index: kept in a vector register across
    // the loop it was the pass's one spilled value)
    if (FUSED && MOM2 && (threadIdx.x & 63) == 0)
        prep.mom2_out[8 * (i64)blockIdx.x + (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6)] =
            __hiloint2double(m2_hi, m2_lo);
}

// the wavefronts' partial sums added in a fixed order: 256 workgroups each sum a slice (256 strided
// sums, then a tree), one workgroup sums the 256 results the same way
__global__ __launch_bounds__(256) void k_sum_partials(const double *__restrict__ part, i64 n,
                                                      double *__restrict__ out) {
    __shared__ double red[256];
    const i64 per = (n + gridDim.x - 1) / gridDim.x;
    const i64 lo = (i64)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double s = 0;
    for (i64 i = lo + threadIdx.x; i < hi; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

template <int ORDER, int T>
static int launch_gather(cg_ctx *c, const double *pos, double *mom, const unsigned *tile_offset,
                         double c1, double c2, double factor, const PrepArgs *prep) {
    const size_t lds = sizeof(double) * GatherLds<ORDER, T>::doubles;
    auto kern = k_gather_kick_tiled<ORDER, T, 0>;
    auto kern_prep = k_gather_kick_tiled<ORDER, T, 1>;
    auto kern_fused = k_gather_kick_tiled<ORDER, T, 2>;
    auto kern_fused_m2 = k_gather_kick_tiled<ORDER, T, 2, true>;
    // the attribute belongs to the function ON A DEVICE: once per device of this process
    static bool attr_set[64] = {};
    const int dev = c->p.device & 63;
    if (!attr_set[dev] && lds > 64 * 1024) {
        CG_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        CG_HIP(hipFuncSetAttribute((const void *)kern_prep,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CG_HIP(hipFuncSetAttribute((const void *)kern_fused,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CG_HIP(hipFuncSetAttribute((const void *)kern_fused_m2,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    unsigned nt = (unsigned)c->ntiles;
    PrepArgs plain{};
    plain.order = prep ? prep->order : tile_order_args(c);
    const unsigned nblocks = nt + plain.order.cap;
    if (prep && prep->start_out)
        hipLaunchKernelGGL(prep->mom2_out ? kern_fused_m2 : kern_fused, dim3(nblocks), dim3(512), lds, c->stream, pos, mom, tile_offset,
                           c->mesh, c->N, c->ny, c->pad, c->p.nghosts, c->tiles.nty, nt, c->xmap,
                           c->geom_gather, c1, c2, factor, *prep);
    else if (prep)
        hipLaunchKernelGGL(kern_prep, dim3(nblocks), dim3(512), lds, c->stream, pos, mom, tile_offset,
                           c->mesh, c->N, c->ny, c->pad, c->p.nghosts, c->tiles.nty, nt, c->xmap,
                           c->geom_gather, c1, c2, factor, *prep);
    else
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), lds, c->stream, pos, mom, tile_offset,
                           c->mesh, c->N, c->ny, c->pad, c->p.nghosts, c->tiles.nty, nt, c->xmap,
                           c->geom_gather, c1, c2, factor, plain);
    return 0;
}

int cgk_gather_kick_tiled(cg_ctx *c, const double *pos, double *mom, i64 n,
                          const unsigned *tile_offset, int diff_order, double factor,
                          int prepare, double next_dtm, const FusedScatter *fs) {
    PrepArgs prep_args{next_dtm, c->p.boxsize, c->geom_deposit, c->tiles, c->tile_count,
                       c->emig_idx, c->emig_count, c->emig_cap};
    if (fs) {
        prep_args.start_out = fs->start_out;
        prep_args.count_out = fs->count_out;
        prep_args.pos_out = fs->pos_out;
        prep_args.mom_out = fs->mom_out;
        prep_args.ids_in = fs->ids_in;
        prep_args.ids_out = fs->ids_out;
        prep_args.aux_in = fs->aux_in;
        prep_args.aux_out = fs->aux_out;
        prep_args.out_capacity = fs->out_capacity;
        prep_args.err_flags = c->err_flags;
        prep_args.count_in = fs->count_in;
        prep_args.emig_rows = c->emig_rows;
        prep_args.emig_rows_count = c->emig_rows_count;
        prep_args.emig_rows_cap = c->emig_rows_cap;
        if (c->emig_rows) CG_HIP(hipMemsetAsync(c->emig_rows_count, 0, 4, c->stream));
        if (c->mom2_sum_out) {
            const size_t need = sizeof(double) * (8 * ((size_t)c->ntiles + (c->ntiles / 8 + 8)) + 256);
            if (need > c->mom2_partial_bytes) {
                CG_HIP(hipStreamSynchronize(c->stream));
                (void)hipFree(c->mom2_partial);
                c->mom2_partial = nullptr;
                c->mom2_partial_bytes = 0;
                CG_HIP(hipMalloc((void **)&c->mom2_partial, need));
                c->mom2_partial_bytes = need;
            }
            CG_HIP(hipMemsetAsync(c->mom2_partial, 0, need, c->stream));
            prep_args.mom2_out = c->mom2_partial;
        }
        prepare = 0;
        CG_HIP(hipMemsetAsync(fs->count_out, 0, 4 * (8 * c->ntiles), c->stream));
    }
    if (prepare && c->emig_idx)
        CG_HIP(hipMemsetAsync(c->emig_count, 0, sizeof(unsigned), c->stream));
    if (cgk_tile_order(c, tile_offset, fs ? fs->count_in : nullptr)) return 1;
    prep_args.order = tile_order_args(c);
    const PrepArgs *prep = (prepare || fs) ? &prep_args : nullptr;
    if (prepare)
        CG_HIP(hipMemsetAsync(c->tile_count, 0, 4 * (8 * c->ntiles + 1), c->stream));
    const int T = c->tiles.tx;
    double dx = c->p.boxsize / (double)c->N;  // interactions.py:2133
    double c1, c2 = 0;
    if (diff_order == 2) c1 = (1.0 / 2) / dx;  // mesh.py:4967
    else {
        c1 = (2.0 / 3) / dx;  // mesh.py:4973
        c2 = (1.0 / 12) / dx;  // mesh.py:4977
    }
    int rc = 1;
#define CG_GATHER_CASE(TT)                                                                       \
    case TT:                                                                                     \
        rc = diff_order == 2                                                                     \
                 ? launch_gather<2, TT>(c, pos, mom, tile_offset, c1, c2, factor, prep)          \
                 : launch_gather<4, TT>(c, pos, mom, tile_offset, c1, c2, factor, prep);         \
        break;
    switch (T) {
        CG_GATHER_CASE(16)
        CG_GATHER_CASE(8)
        CG_GATHER_CASE(4)
        CG_GATHER_CASE(2)
        default: cg_set_error("cgk_gather_kick_tiled: tile extent %d", T); return 1;
    }
#undef CG_GATHER_CASE
    if (rc) return rc;
    CG_LAUNCH_CHECK();
    if (fs && c->mom2_sum_out) {
        // (the cap the pass was LAUNCHED with: tile_order_args() reads a word the device updates)
        const i64 nparts = 8 * ((i64)c->ntiles + (i64)prep_args.order.cap);
        // (the first 256 doubles of the partials' buffer past its nparts entries take stage one)
        double *stage = c->mom2_partial + nparts;
        hipLaunchKernelGGL(k_sum_partials, dim3(256), dim3(256), 0, c->stream, c->mom2_partial,
                           nparts, stage);
        CG_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, stage, (i64)256,
                           c->mom2_sum_out);
        CG_LAUNCH_CHECK();
    }
    c->prep_valid = prepare != 0;
    c->prep_pos = pos;
    c->prep_mom = mom;
    c->prep_n = n;
    c->prep_dtm = next_dtm;
    return 0;
}
