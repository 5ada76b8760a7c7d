// cg_shortrange.hip — P3M short-range sweep over half-tile cells (A13-A15) on gfx950.
//
//   Tiling.sort                 species.py:707-823      particle -> tile
//   particle_particle           interactions.py:1563-1791  tile neighbours, periodic offset,
//                                                         x_ji = xi - xj
//   gravity_pairwise_shortrange gravity.py:263-354      r2 cut, r2-indexed table, Δmom
//
// Form: one-sided.  The reference visits every unordered pair once and updates
// both partners (Δmom_r += r*f, Δmom_s -= r*f); here every receiver particle
// sums over all its partners itself — twice the arithmetic, but no atomics, no
// write conflicts (the order of partners inside a cell follows the cell list's
// atomics, so sums are reproducible to rounding, not bit for bit).  The pair
// vector, r2 and the table index are evaluated with the reference's expression
// and operation order ((xi - xj) + offset; x*x + y*y + z*z; int(r2*scaling)):
// exact negation symmetry makes the two directions of a pair bit-consistent.
// (The tiles of 64 particles and more go to cg_shortrange_dense.hip; earlier forms of the
// sweep — one wavefront per tile, a single-precision pre-test, a matrix-core range filter —
// were measured slower and are gone: profiles/README.md keeps their numbers.)
#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <type_traits>

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

struct SrParams {
    double boxsize, r2_index_scaling, r2_max, factor;
    int nt;
    // adaptive rungs (null = every particle on rung 0 with `factor`): a receiver on an
    // active rung (rung >= lowest_active) is kicked with factors[rung_jumped]; inactive
    // receivers are skipped.  One-sided form of interactions.py:1688-1761 +
    // gravity.py:318-349: each particle's kick uses its OWN rung's integral.
    const double *factors;
    const signed char *rung, *rung_jumped;
    int lowest_active;
    // one byte per tile, non-zero where the cells sweep has receivers to kick: a tile with a
    // receiver on an active rung (k_sr_tile_activity) that the dense tiles' sweep has not
    // taken (null = every tile takes part)
    const unsigned char *tile_active;
    // cg_shortrange_stats: [0] pair tests executed (lanes that hold a receiver and a supplier of
    // the range), [1] of them in range, [2] wavefront trips (64 lane slots each) — counted by the
    // STATS instantiations only, which run while the statistics are switched on
    unsigned long long *stats;
    // the table entries a pair in range can ask for: 0 .. (int)(r2_max * r2_index_scaling)
    int table_n;
    // receivers' list with the active particles first in every cell (cgk_shortrange_cells with
    // rungs): nact[cell] of them, rj_sorted[row] = their jumped rung index (null = plain list)
    const unsigned *nact;
    const signed char *rj_sorted;
};
struct SrCount {
    unsigned tests = 0, hits = 0, trips = 0;
};

// ===========================================================================
// Sweep over a cell list at HALF-tile granularity (the reference prunes below the tile level
// with subtiles, interactions.py:1141-1278, species.py:4031-4142).
//
// Why: a sweep with one wavefront per tile tests every receiver against all 27 neighbouring
// tiles (~600 pair tests per particle at the default parameters, ~15 % hits) with 22 receivers
// x 2 supplier groups = 44 of 64 lanes busy, and it is bound by FP64 VALU issue.  What helps
// is fewer wave-iterations, i.e. fewer tests and fuller wavefronts:
//  * cells of half a tile (extent >= range/2): a receiver cell needs supplier cells no further
//    than 2 away;
//  * particles are SORTED by cell (z fastest), positions copied in that order: a column of
//    cells along z is one contiguous run, staged with plain coalesced loads (no index
//    indirection, no dependent loads);
//  * one workgroup per BLOCK of tiles (4 x 2, or 2 x 2) stages the columns around the block
//    ONCE (6 cells along z each), a wavefront takes cell columns of the block (2 cells along z,
//    ~5.5 receivers): a receiver group needs the 5 x 5 columns around its own over all 6 cells
//    — per x that is one contiguous range of the staged array — 412 tests per receiver instead
//    of 594, with R x floor(64/R) ~ 60 of 64 lanes busy (k_sr_sweep_blocks below);
//  * no self test: a particle paired with itself has x_ji = 0 exactly and contributes
//    0 * table[0] = 0, as would two distinct particles at one position (the reference skips
//    i == j by index, interactions.py:1722; the sums are the same);
//  * the accumulation a += x_ji * f is a fused multiply-add (the order of partners already
//    differs from the reference's, Δmom is compared to 1e-12); x_ji, r2 and the table index
//    keep the reference's operation order and are bit-identical.
// Periodic images: the tiles on the box faces are swept in blocks that reach across the face,
// the offset added as the reference does, (xi - xj) + offset (SrWrapSeg below).
// ===========================================================================
constexpr int kSrSlack = 128; // masked lanes read up to 2*S - 1 < 128 entries past a range

__device__ __forceinline__ unsigned sr_cell_of(double x0, double x1, double x2, double inv,
                                               double ext, unsigned nt) {
    // tile index exactly as Tiling.sort (species.py:775-780); the half inside the tile from
    // the distance to the tile's lower face
    unsigned c[3];
    const double xs[3] = {x0, x1, x2};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const double x = xs[d];
        unsigned t = (unsigned)(i64)((x - 0.0) * inv);
        t = t >= nt ? nt - 1 : t;
        const unsigned half = (x - (double)t * ext) >= 0.5 * ext ? 1u : 0u;
        c[d] = 2u * t + half;
    }
    const unsigned nc = 2u * nt;
    return (c[0] * nc + c[1]) * nc + c[2];
}

// The cell list is a counting sort by cell: a counting pass, a scan, a placing pass.  Device-scope
// atomics execute memory-side on MI355X (2^31 of them take 77 ms): one per particle was most of
// the list's time.  Particle memory is in mesh-tile order (16^3 mesh cells), so the 1024
// consecutive particles a workgroup takes fall into a few hundred short-range cells at most:
// they are counted in an LDS hash table first (key -> slot by open addressing, the slot's counter
// hands every particle its rank among the workgroup's particles of that cell), and the
// workgroup then issues ONE device atomic per cell it met.  Particles in any other order still
// sort correctly (up to 1024 distinct cells fit the table: one per particle).
constexpr int kSrHashSlots = 2048, kSrPerThread = 4;
// ACT (a sub-step of the rung loop, main.py:1347-1624, that kicks the rungs >= lowest_active
// only): inside every cell the particles on active rungs come FIRST, nact[cell] of them — the
// list by tile AND rung of the reference (species.py tiles_rungs_N) in the sweep's layout.  A
// receiver group is then the first nact rows of its two cells, known from two words per cell:
// no pass over the tiles' rungs, no gathers of rung[order[row]] in front of every chunk.  The
// rows' jumped rung indices (what selects a receiver's factor, gravity.py:318-349) are written
// in list order beside the positions.
struct SrHash {
    unsigned key[kSrHashSlots], cnt[kSrHashSlots];
};
struct SrHashAct : SrHash {
    unsigned act[kSrHashSlots];
};
template <class H_>
__device__ __forceinline__ void sr_hash_clear(H_ &H) {
    constexpr bool ACT = sizeof(H_) > sizeof(SrHash);
    for (int i = threadIdx.x; i < kSrHashSlots; i += 256) {
        H.key[i] = 0xffffffffu;
        H.cnt[i] = 0;
        if constexpr (ACT) H.act[i] = 0;
    }
}
// returns the slot of `key`
__device__ __forceinline__ unsigned sr_hash_slot(SrHash &H, unsigned key) {
    unsigned slot = (key * 2654435761u) >> (32 - 11);
    for (;;) {
        const unsigned old = atomicCAS(&H.key[slot], 0xffffffffu, key);
        if (old == 0xffffffffu || old == key) break;
        slot = (slot + 1) & (kSrHashSlots - 1);
    }
    return slot;
}
struct SrActive {  // (ACT) which particles are receivers of this sub-step
    const signed char *rung, *rung_jumped;
    int lowest;
};
// Pass 1, counting.  The workgroup's particles are counted per cell in the LDS table (a slot's
// counter hands every particle its rank among the workgroup's particles of that cell); then ONE
// device atomic per cell the workgroup met adds its count to the cell's — and what the atomic
// returns is where the workgroup's particles start INSIDE the cell.  cellrel[p] = (cell, that
// start + the particle's rank): the second pass has nothing left to count — no table, no atomic
// (it had one per cell and workgroup too: the two passes' 13 million device atomics at 256^3
// were most of their time; the memory-side atomic units take 28e9 a second).
// ACT: count[] takes the inactive particles, nact[] the active ones; bit 31 of the second word
// marks an active particle.
// BEGIN: the particle's sub-step pass first (cg_substep.h: drift, flag_rung_jumps, nullify_Δ — the
// time loop deferred it to this list): the drifted position is binned without being read again.
template <bool ACT, bool BEGIN>
__global__ __launch_bounds__(256) void k_sr_cell_count(const double *__restrict__ pos, i64 n,
                                                       double inv, double ext, unsigned nt,
                                                       unsigned *__restrict__ count, SrActive A,
                                                       unsigned *__restrict__ nact,
                                                       uint2 *__restrict__ cellrel,
                                                       SubstepBegin B) {
    __shared__ std::conditional_t<ACT, SrHashAct, SrHash> H;
    __shared__ unsigned s_cnt[64];  // (BEGIN: the workgroup's particles per rung after the sub-step)
    sr_hash_clear(H);
    if (BEGIN && threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const i64 base = (i64)blockIdx.x * (256 * kSrPerThread);
    unsigned slot[kSrPerThread], rank[kSrPerThread], cell[kSrPerThread];
    bool act[kSrPerThread];
    int r_after[kSrPerThread];
    // (first the four particles' own work — their loads are in flight together — then the table:
    // its compare-and-swap loops would otherwise stand between one particle's loads and the next's)
#pragma unroll
    for (int u = 0; u < kSrPerThread; u++) {
        const i64 p = base + threadIdx.x + 256 * u;
        slot[u] = rank[u] = cell[u] = 0, act[u] = false;
        r_after[u] = 255;
        if (p < n) {
            double x, y, z;
            if (BEGIN) r_after[u] = cg_substep_begin_particle(B, p, x, y, z);
            else x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
            cell[u] = sr_cell_of(x, y, z, inv, ext, nt);
            if constexpr (ACT) act[u] = A.rung[p] >= A.lowest;
        }
    }
#pragma unroll
    for (int u = 0; u < kSrPerThread; u++) {
        const i64 p = base + threadIdx.x + 256 * u;
        if (p < n) {
            slot[u] = sr_hash_slot(H, cell[u]);
            bool taken = false;
            if constexpr (ACT)
                if (act[u]) rank[u] = atomicAdd(&H.act[slot[u]], 1u), taken = true;
            if (!taken) rank[u] = atomicAdd(&H.cnt[slot[u]], 1u);
        }
        if (BEGIN && B.partial) cg_count_rungs(r_after[u], B.N_rungs, s_cnt);
    }
    __syncthreads();
    if (BEGIN && B.partial && threadIdx.x < (unsigned)B.N_rungs)
        B.partial[(i64)B.N_rungs * blockIdx.x + threadIdx.x] = s_cnt[threadIdx.x];
    for (int i = threadIdx.x; i < kSrHashSlots; i += 256) {
        const unsigned key = H.key[i];
        if (H.cnt[i]) H.cnt[i] = atomicAdd(&count[key], H.cnt[i]);
        if constexpr (ACT)
            if (H.act[i]) H.act[i] = atomicAdd(&nact[key], H.act[i]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kSrPerThread; u++) {
        const i64 p = base + threadIdx.x + 256 * u;
        if (p < n) {
            unsigned first = H.cnt[slot[u]];
            if constexpr (ACT)
                if (act[u]) first = H.act[slot[u]];
            cellrel[p] = make_uint2(cell[u], (first + rank[u]) | (act[u] ? 0x80000000u : 0u));
        }
    }
}
// (ACT) the cells' populations for the scan: the inactive + the active
__global__ __launch_bounds__(256) void k_sr_cell_total(unsigned *__restrict__ count,
                                                       const unsigned *__restrict__ nact,
                                                       unsigned ncells) {
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncells) count[c] += nact[c];
}
// Pass 2, placing: row = the cell's offset + the particle's place inside the cell (ACT: the active
// ones first, then the others)
template <bool ACT>
__global__ __launch_bounds__(256) void k_sr_cell_place(const double *__restrict__ pos, i64 n,
                                                       const unsigned *__restrict__ offset,
                                                       const uint2 *__restrict__ cellrel,
                                                       unsigned *__restrict__ order,
                                                       double *__restrict__ pos_sorted,
                                                       SrActive A, const unsigned *__restrict__ nact,
                                                       signed char *__restrict__ rj_sorted) {
    // (four particles per thread, every load of the four issued before the first store: the
    // chain cell -> the cell's offset -> the row's address is waited for once, not four times)
    // Workgroups are handed to the eight XCDs in turn; the rows of consecutive stretches of
    // particles share lines of the sorted arrays (a cell's particles sit together in the store's
    // tile order), and a line written from two L2s goes out twice, partially: an XCD takes a
    // contiguous eighth of the particles (grid: a multiple of 8).
    const unsigned per_xcd = gridDim.x >> 3;
    const i64 base = (i64)((blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3)) * (256 * kSrPerThread);
    uint2 cr[kSrPerThread];
    double x[kSrPerThread], y[kSrPerThread], z[kSrPerThread];
    i64 q[kSrPerThread];
#pragma unroll
    for (int u = 0; u < kSrPerThread; u++) {
        const i64 p = base + threadIdx.x + 256 * u;
        cr[u] = p < n ? cellrel[p] : make_uint2(0u, 0u);
        if (p < n) x[u] = pos[3 * p], y[u] = pos[3 * p + 1], z[u] = pos[3 * p + 2];
    }
#pragma unroll
    for (int u = 0; u < kSrPerThread; u++) {
        const i64 p = base + threadIdx.x + 256 * u;
        q[u] = 0;
        if (p < n) {
            q[u] = (i64)offset[cr[u].x] + (cr[u].y & 0x7fffffffu);
            if (ACT && !(cr[u].y >> 31)) q[u] += nact[cr[u].x];
        }
    }
#pragma unroll
    for (int u = 0; u < kSrPerThread; u++) {
        const i64 p = base + threadIdx.x + 256 * u;
        if (p < n) {
            order[q[u]] = (unsigned)p;
            pos_sorted[3 * q[u]] = x[u];
            pos_sorted[3 * q[u] + 1] = y[u];
            pos_sorted[3 * q[u] + 2] = z[u];
            if constexpr (ACT)
                if (rj_sorted) rj_sorted[q[u]] = A.rung_jumped[p];
        }
    }
}

int cgk_shortrange_cells(cg_ctx *c, const double *pos, i64 n, i64 nt, double tile_extent,
                         unsigned *order, unsigned *offset, double *pos_sorted,
                         const signed char *rung, const signed char *rung_jumped,
                         int lowest_active, unsigned *nact, signed char *rj_sorted) {
    const double eps = 2.220446049250313e-16;
    double inv = (1 / tile_extent) * (1 - 2 * eps);
    i64 ncells = 8 * nt * nt * nt;
    const bool act = nact != nullptr;
    // the sub-step's first pass, if the time loop left it to this list (same particles)
    const bool begin = c->sub_pending && c->sub_begin->pos == pos && c->sub_begin->n == n && n > 0;
    if (!begin && cgk_substep_flush(c)) return 1;
    const size_t words = (size_t)(ncells + 4) + 2 * (size_t)(n + 1);  // counts | (cell, place)
    if (4 * words > c->sr_tmp_bytes) {
        CG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->sr_tmp);
        c->sr_tmp = nullptr;
        c->sr_tmp_bytes = 0;
        CG_HIP(hipMalloc(&c->sr_tmp, 4 * words));
        c->sr_tmp_bytes = 4 * words;
    }
    unsigned *count = (unsigned *)c->sr_tmp;
    uint2 *cellrel = (uint2 *)(count + ((ncells + 2) & ~(i64)1));
    CG_HIP(hipMemsetAsync(count, 0, 4 * (size_t)(ncells + 1), c->stream));
    if (act) CG_HIP(hipMemsetAsync(nact, 0, 4 * (size_t)ncells, c->stream));
    const SrActive A{rung, rung_jumped, lowest_active};
    i64 blocks = (n + 256 * kSrPerThread - 1) / (256 * kSrPerThread);
    if (n > 0) {
        auto kern = act ? (begin ? k_sr_cell_count<true, true> : k_sr_cell_count<true, false>)
                        : (begin ? k_sr_cell_count<false, true> : k_sr_cell_count<false, false>);
        c->sub_pending = false;
        i64 nwg = 0;
        c->sub_begin->partial = nullptr;
        if (begin && c->sub_counts) {
            if (cgk_substep_partial(c, n, 256 * kSrPerThread, c->sub_begin->N_rungs, &nwg)) return 1;
            c->sub_begin->partial = c->sub_partial;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, c->stream, pos, n, inv,
                           tile_extent, (unsigned)nt, count, A, nact, cellrel, *c->sub_begin);
        CG_LAUNCH_CHECK();
        if (begin && c->sub_counts &&
            cgk_substep_populations(c, nwg, c->sub_begin->N_rungs, c->sub_counts))
            return 1;
        if (act) {
            hipLaunchKernelGGL(k_sr_cell_total, dim3((unsigned)((ncells + 255) / 256)), dim3(256),
                               0, c->stream, count, nact, (unsigned)ncells);
            CG_LAUNCH_CHECK();
        }
    }
    size_t need = 0;
    CG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, count, offset, (int)(ncells + 1),
                                            c->stream));
    if (need > c->scan_tmp_bytes) {
        CG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->scan_tmp);
        c->scan_tmp = nullptr;
        CG_HIP(hipMalloc(&c->scan_tmp, need));
        c->scan_tmp_bytes = need;
    }
    CG_HIP(hipcub::DeviceScan::ExclusiveSum(c->scan_tmp, need, count, offset, (int)(ncells + 1),
                                            c->stream));
    if (n > 0) {
        hipLaunchKernelGGL(act ? k_sr_cell_place<true> : k_sr_cell_place<false>,
                           dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(256), 0, c->stream, pos, n,
                           offset, cellrel, order, pos_sorted, A, nact, rj_sorted);
        CG_LAUNCH_CHECK();
    }
    // what the dense tiles' sweep wants to know about this list, while it is being made
    return cgk_shortrange_dense_look(c, offset, nt);
}

// The periodic images of a block that reaches across a face of the box (WRAP).  The reference
// adds the image's offset to the pair vector, (xi - xj) + offset (gravity.py:299-302,
// interactions.py:1615-1621); for a staged block the offset of a pair is
// (the receiver column's box shift - the supplier column's) * boxsize per dimension:
//  x: a range is the five columns of ONE x — the offset is the same for the whole range;
//  y: the five columns of a range cross the face at most once — the staged rows before `ys`
//     take oyA, the others oyB;
//  z: a column may come in two pieces (below and above the face): one staged double per
//     supplier, `soz` (0 for most).
// Off the faces the three are (+0, +0, +0): the same sums as the plain loop's.
struct SrWrapSeg {
    double ox;   // (uniform over the range)
    int ys;      // first staged row (window-relative) of the columns beyond the y face
};
// d box lengths, d in {-1, 0, 1}, by selects: wave-uniform d and L stay in scalar registers (a
// conversion and a product would go through the vector unit and keep registers there)
__device__ __forceinline__ double sr_image(int d, double L) {
    const double v = d == 0 ? 0.0 : (d > 0 ? L : -L);
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// pair tests of this lane's receiver against the staged suppliers [a, b), lane group `sub` of S
// taking every S-th.  Accumulates sum x_ji * table[...] — the receiver's factor is applied once
// at the end.  NB pairs per trip: their table loads (hits only) are all issued before the first
// is used (two: 64 VGPRs, 8 wavefronts per SIMD, 8.6 ms at 256^3 / 512^3; four: 80 VGPRs, 9.4 ms;
// eight: 11.8 ms).
template <bool WRAP, bool MASK, int NB, bool STATS>
__device__ __forceinline__ void sr_cell_batch(int k, int S, int b, double xi, double yi, double zi,
                                              const double *sx, const double *sy,
                                              const double *sz, double r2_max,
                                              double r2_index_scaling,
                                              const double *__restrict__ table, double &ax,
                                              double &ay, double &az, bool counted, SrCount &cnt,
                                              SrWrapSeg w, double oyA, double oyB,
                                              const double *soz) {
    // NB pairs of this lane: their table loads (hits only) are all issued before the first use
    double xj[NB], yj[NB], zj[NB], r2[NB], t[NB];
    bool hit[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const int kj = k + j * S;
        xj[j] = xi - sx[kj];                             // interactions.py:1787-1789
        yj[j] = yi - sy[kj];
        zj[j] = zi - sz[kj];
        if (WRAP) {                                      // gravity.py:299-302
            xj[j] += w.ox;
            yj[j] += kj < w.ys ? oyA : oyB;
            zj[j] += soz[kj];
        }
        r2[j] = sr_r2(xj[j], yj[j], zj[j]);  // gravity.py:306
        hit[j] = r2[j] <= r2_max;                        // gravity.py:311: skip r2 > r2_max
        if (MASK) hit[j] &= kj < b;                      // (read past the range: staged slack)
        if (STATS) {
            const bool valid = counted && (!MASK || kj < b);
            cnt.tests += (unsigned)__popcll(__ballot(valid));
            cnt.hits += (unsigned)__popcll(__ballot(valid && hit[j]));
            cnt.trips++;
        }
    }
#pragma unroll
    for (int j = 0; j < NB; j++) {
        t[j] = 0.0;
        if (hit[j]) t[j] = table[(unsigned)(int)(r2[j] * r2_index_scaling)];  // gravity.py:316-321
    }
#pragma unroll
    for (int j = 0; j < NB; j++) {
        ax = __builtin_fma(xj[j], t[j], ax);
        ay = __builtin_fma(yj[j], t[j], ay);
        az = __builtin_fma(zj[j], t[j], az);
    }
}

// One trip over the end of a range and the start of the next: the lanes whose turn i = sub + j S
// comes after the `rem` suppliers left at `pos` take theirs from the next range [na, nb) instead
// of idling — the five ranges of a receiver chunk (~82 suppliers each, 7.5 rows of its S lanes)
// would otherwise end in a mostly empty trip each.
template <bool WRAP, bool STATS>
__device__ __forceinline__ void sr_cell_straddle(int pos, int rem, int na, int nb, int sub, int S,
                                                 double xi, double yi, double zi, const double *sx,
                                                 const double *sy, const double *sz, double r2_max,
                                                 double r2_index_scaling,
                                                 const double *__restrict__ table, double &ax,
                                                 double &ay, double &az, bool counted,
                                                 SrCount &cnt, SrWrapSeg w, SrWrapSeg wn,
                                                 double oyA, double oyB, const double *soz) {
    double xj[2], yj[2], zj[2], r2[2], t[2];
    bool hit[2];
    const int nxt = na - rem;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int i = sub + j * S;
        const bool here = i < rem;
        const int kj = (here ? pos : nxt) + i;
        const bool valid = here || kj < nb;
        xj[j] = xi - sx[kj];
        yj[j] = yi - sy[kj];
        zj[j] = zi - sz[kj];
        if (WRAP) {
            xj[j] += here ? w.ox : wn.ox;
            yj[j] += kj < (here ? w.ys : wn.ys) ? oyA : oyB;
            zj[j] += soz[kj];
        }
        r2[j] = sr_r2(xj[j], yj[j], zj[j]);
        hit[j] = r2[j] <= r2_max && valid;
        if (STATS) {
            cnt.tests += (unsigned)__popcll(__ballot(counted && valid));
            cnt.hits += (unsigned)__popcll(__ballot(counted && hit[j]));
            cnt.trips++;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        t[j] = 0.0;
        if (hit[j]) t[j] = table[(unsigned)(int)(r2[j] * r2_index_scaling)];
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        ax = __builtin_fma(xj[j], t[j], ax);
        ay = __builtin_fma(yj[j], t[j], ay);
        az = __builtin_fma(zj[j], t[j], az);
    }
}

// The NSEG ranges [sa[s], sb[s]) of the staged window as ONE sequence of suppliers: full trips
// inside a range, one trip across each boundary, a masked end.
template <bool WRAP, bool STATS, int NSEG>
__device__ __forceinline__ void sr_cell_ranges(const int (&sa)[NSEG], const int (&sb)[NSEG],
                                               int sub, int S, double xi, double yi, double zi,
                                               const double *sx, const double *sy,
                                               const double *sz, double r2_max,
                                               double r2_index_scaling,
                                               const double *__restrict__ table, double &ax,
                                               double &ay, double &az, bool counted,
                                               SrCount &cnt, const SrWrapSeg (&ws)[NSEG],
                                               double oyA, double oyB, const double *soz) {
    int pos = sa[0];
#pragma unroll
    for (int s = 0; s < NSEG; s++) {
        const int b = sb[s];
        for (; pos + 2 * S <= b; pos += 2 * S)
            sr_cell_batch<WRAP, false, 2, STATS>(pos + sub, S, b, xi, yi, zi, sx, sy, sz, r2_max,
                                                 r2_index_scaling, table, ax, ay, az, counted,
                                                 cnt, ws[s], oyA, oyB, soz);
        const int rem = b - pos;  // < 2 S
        if (s == NSEG - 1) {
            if (rem > S)
                sr_cell_batch<WRAP, true, 2, STATS>(pos + sub, S, b, xi, yi, zi, sx, sy, sz,
                                                    r2_max, r2_index_scaling, table, ax, ay, az,
                                                    counted, cnt, ws[s], oyA, oyB, soz);
            else if (rem > 0)
                sr_cell_batch<WRAP, true, 1, STATS>(pos + sub, S, b, xi, yi, zi, sx, sy, sz,
                                                    r2_max, r2_index_scaling, table, ax, ay, az,
                                                    counted, cnt, ws[s], oyA, oyB, soz);
        } else if (rem <= 0) {  // nothing left here (an empty range, or one the trip before used up)
            pos = sa[s + 1];
        } else {
            sr_cell_straddle<WRAP, STATS>(pos, rem, sa[s + 1], sb[s + 1], sub, S, xi, yi, zi, sx,
                                          sy, sz, r2_max, r2_index_scaling, table, ax, ay, az,
                                          counted, cnt, ws[s], ws[s + 1], oyA, oyB, soz);
            pos = sa[s + 1] + (2 * S - rem);
        }
    }
}

// inclusive scan over the 64 lanes of a wave in DPP adds (no LDS round trips): row_shr 1, 2, 4,
// 8 inside the rows of 16 lanes, then the row totals are broadcast forward (row_bcast 15, 31)
__device__ __forceinline__ unsigned sr_wave_scan(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// One receiver chunk's lanes: R receivers x S supplier groups
struct SrChunk {
    int R, S, sub, rl;
    bool active;
    unsigned pi;
    double xi, yi, zi, factor;
};
// (RUNGS: the instantiation for particles on adaptive rungs — the plain sweep carries none of it:
// with the packing below in one kernel the single-rung sweep measured 7.84 against 7.62 ms)
// RUNGS 2 (ACT): the list has the active particles first in each of the group's two cells — the
// group's receivers are the rows rbeg + k (k < n0) and rbeg + gap + k (the second cell's), `base`
// and `rend` count k from rbeg; nothing of the per-particle rung arrays is read.
// PACK: a plain list swept for a subset of the rungs (RUNGS 1, lowest_active > 0) — the 2 x 2
// blocks' instantiation only; the 4 x 2 blocks are launched for sweeps in which every rung is
// active and carry none of it (with it they spilled 12 B per lane at their 64 registers).
template <int RUNGS, bool PACK>
__device__ __forceinline__ SrChunk sr_chunk_load(unsigned base, unsigned rend, int lane,
                                                 const double *__restrict__ pos_r,
                                                 const unsigned *__restrict__ order_r,
                                                 const SrParams &P, unsigned rbeg = 0,
                                                 unsigned n0 = 0, unsigned gap = 0) {
    SrChunk c;
    c.pi = 0;
    c.factor = P.factor;
    c.xi = c.yi = c.zi = 0;
    const unsigned cand = min(64u, rend - base);  // this chunk's rows of the sorted order
    unsigned qi = base;
    if (RUNGS == 1 && PACK && P.lowest_active > 0) {
        // With rungs only the receivers on an active rung take part (gravity.py:318-349 through
        // the tiles' active rungs): they are packed to the front — lane l of the active ones
        // hands its row to lane rank(l) — so that R counts them alone and every one of them
        // gets 64 / R lanes (a sub-step for the upper rungs kicks a third of a tile's
        // particles or fewer)
        unsigned pi = 0;
        bool act = false;
        if ((unsigned)lane < cand) {
            pi = order_r[base + lane];
            act = P.rung[pi] >= P.lowest_active;
        }
        const unsigned long long mask = __ballot(act);
        const int R = __builtin_amdgcn_readfirstlane(__popcll(mask));
        if (R == 0) {
            c.R = 1, c.S = 1, c.sub = lane, c.rl = 0, c.active = false;
            return c;
        }
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        // (an inactive lane parks its row at lane 63, which no active lane targets unless all
        // 64 are active — and then nobody is inactive)
        const int target = act ? rank : 63;
        const int q_packed = __builtin_amdgcn_ds_permute(4 * target, (int)(base + lane));
        const int p_packed = __builtin_amdgcn_ds_permute(4 * target, (int)pi);
        c.R = R;
        c.S = min(64 / R, 32);
        c.sub = lane / R;
        c.rl = lane - c.sub * R;
        c.active = c.sub < c.S;
        qi = (unsigned)__shfl(q_packed, c.rl);
        c.pi = (unsigned)__shfl(p_packed, c.rl);
        if (c.active) {
            c.xi = pos_r[3 * (i64)qi];
            c.yi = pos_r[3 * (i64)qi + 1];
            c.zi = pos_r[3 * (i64)qi + 2];
            c.factor = P.factors[P.rung_jumped[c.pi]];
        } else {
            c.pi = 0;
        }
        return c;
    }
    // wave-uniform by construction; tell the compiler (scalar registers, scalar loops)
    c.R = __builtin_amdgcn_readfirstlane((int)cand);
    c.S = min(64 / c.R, 32);
    c.sub = lane / c.R;
    c.rl = lane - c.sub * c.R;
    c.active = c.sub < c.S;
    qi = base + c.rl;  // receiver's row in the sorted order
    if (RUNGS == 2) qi += qi - rbeg >= n0 ? gap : 0u;
    if (c.active) {
        c.pi = order_r[qi];
        c.xi = pos_r[3 * (i64)qi];
        c.yi = pos_r[3 * (i64)qi + 1];
        c.zi = pos_r[3 * (i64)qi + 2];
        if (RUNGS == 1) c.factor = P.factors[P.rung_jumped[c.pi]];  // (every rung is active)
        // (the rows' jumped rung indices in list order: a load beside the others — read through
        // order_r it would wait for that load first, 0.5 ms of a sweep with half the receivers
        // active)
        if (RUNGS == 2) c.factor = P.factors[P.rj_sorted[qi]];
    }
    return c;
}

// One thread per tile: does any of its receivers (the 2 x 2 columns of 2 cells each) sit on an
// active rung?
__global__ __launch_bounds__(256) void k_sr_tile_activity(const unsigned *__restrict__ order_r,
                                                          const unsigned *__restrict__ off_r,
                                                          const signed char *__restrict__ rung,
                                                          int lowest_active, int nt,
                                                          unsigned char *__restrict__ tile_active) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned ntiles = (unsigned)nt * nt * nt;
    if (t >= ntiles) return;
    const unsigned tc = t % nt, tb = (t / nt) % nt, ta = t / ((unsigned)nt * nt), nc = 2u * nt;
    bool any = false;
    for (unsigned w = 0; w < 4 && !any; w++) {
        const unsigned cell = ((2 * ta + (w >> 1)) * nc + (2 * tb + (w & 1))) * nc + 2 * tc;
        for (unsigned q = off_r[cell], e = off_r[cell + 2]; q < e && !any; q++)
            any = rung[order_r[q]] >= lowest_active;
    }
    tile_active[t] = any ? 1 : 0;
}

// ===========================================================================
// The interior of the box, 4 x 2 tiles per workgroup, the look-up table in LDS (round 5).
//
// What the sweep above costs apart from its pair tests was measured with a build whose pair
// loop is empty (256^3 / 512^3, 22 particles per tile): 3.2 of 8.1 ms — the chain a tile's
// workgroup goes through before its first pair (offsets of its 36 columns -> their rows ->
// barrier), ~0.3 us of arithmetic in ~3 us of memory round trips, paid 750,000 times, which the
// eight workgroups of a CU do not hide.  Tiles that are neighbours in x and y share most of what
// they stage: a block of 4 x 2 tiles needs the 12 x 8 columns around it (6 cells along z each,
// the same window for all of them, so that a receiver's five columns of one x stay ONE range of
// the staged array) — 1584 suppliers on average instead of 8 x 594 — and ONE chain.  Sixteen
// wavefronts, two receiver groups (a cell column of a tile: 2 cells, ~5.5 receivers) each;
// the pair loop is the one above.  The tiles on the box faces go to the WRAP instantiation: blocks
// of 2 x 2 tiles that reach across the face (round 6; before, one tile per workgroup with a
// staged offset per supplier and dimension: 6.4 % of the tiles cost 11 % of the uniform sweep and a
// quarter of the rung loop's kernel time).  An interior that is not a multiple of the block: the
// last block of a dimension ends with the last interior tile and leaves the tiles it shares with
// its neighbour out.
// The table (gravity.py:416-437; 4096 entries by default) is copied into LDS by every workgroup:
// what the look-up costs as a global load was measured with variant builds
// (profiles/r05_sr_lookup_ab.txt) — 7.3 ms as it was, 6.3 without it, 6.3 with a look-up in LDS —
// and 32 KB of table beside the staged window are what sets the block's size: two workgroups of
// 80 KB and sixteen wavefronts per CU.  A longer table stays in global memory (TABLDS = false).
// Measured at 256^3 / 512^3 on one box: 8.05 ms one tile per workgroup, 7.3 ms with blocks of
// 2 x 2 tiles and the table in global memory.  Also built and measured there: two receivers per
// lane, so that a supplier read from LDS serves two pair tests: 9.3 ms — a chunk's ~5.5 receivers
// become 3 lane rows x 21 supplier groups, a range of ~82 suppliers is four trips of which the
// last is mostly empty, and the fold over 21 groups of six sums costs what the reads saved.
// ===========================================================================
// Two shapes: BX = 4 (4 x 2 tiles, 1024 lanes, the table in LDS) and BX = 2 (2 x 2 tiles, 512
// lanes, 8 x 8 columns, 1056 suppliers on average, 35 KB of LDS, the table where it is) — the
// latter for the sub-steps that kick the upper rungs only (lowest_active > 0) from the plain list,
// where most tiles have no receiver on an active rung and what a block costs before its first pair
// decides
// (tools/soak_p3m.py, 30 base steps of the P3M loop with 8 rungs at 256^3 / 512^3: 2.95 s with
// 4 x 2 blocks for every sweep, 2.76 s with 2 x 2 for the sub-steps — provided its instantiation
// with rungs stays within 64 registers: at 67, seven wavefronts per SIMD, it took 2.95 s too), and
// for boxes of fewer than six tiles a side.
constexpr int kSbY = 2;                                  // tiles per block along y
constexpr int kSbColsY = 2 * kSbY + 4;                   // staged columns along y: 8
constexpr int kSbTable = 4096;                           // table entries that fit into LDS
constexpr int sb_cols(int bx) { return (2 * bx + 4) * kSbColsY; }   // 12 x 8 = 96 | 8 x 8 = 64
constexpr int sb_waves(int bx) { return (2 * bx) * (2 * kSbY) / 2; }  // groups / 2: 16 | 8
// suppliers staged per window (mean 1584 | 1056 at 22 per tile; more take further windows; 1824:
// with the group tables of the active-first form two workgroups still fit a CU's 160 KB)
constexpr int sb_cap(int bx) { return bx == 4 ? 1824 : 1344; }
// (WRAP: one more staged double per supplier, and every column in two pieces)
constexpr size_t sb_lds_bytes(int bx, bool tab, bool wrap) {
    return sizeof(double) * ((wrap ? 4 : 3) * (sb_cap(bx) + kSrSlack) + (tab ? kSbTable : 0)) +
           // (the 4 x 2 blocks with the table: 81,280 B — two workgroups per CU)
           sizeof(unsigned) * ((wrap ? 8 : 3) * sb_cols(bx) + 8 * sb_waves(bx));
}
// Which tiles a workgroup takes.  Plain: the block (bx, by) of the interior's (nt - 2)^2 tiles
// in x and y, tile tc of its nt - 2 in z.  WRAP: the tiles on the faces of the box in blocks of
// 2 x 2 that reach ACROSS the face (tile nt - 1 and tile 0 are neighbours), as a 1-D grid of
// three slabs — 0: x tiles {nt - 1, 0}, every y and z; 1: y tiles {nt - 1, 0}, x inside, every
// z; 2: z tile 0 or nt - 1, x and y inside.  ta0, tb0: the block's first tile (its tiles are
// (ta0 + i) mod nt); skipx, skipy: tiles at the low side of a last block that its neighbour has.
struct SbBlock {
    int ta0, tb0, tc, skipx, skipy;
};
template <int BX, bool WRAP>
__device__ __forceinline__ SbBlock sb_block(int nt) {
    const int m = nt - 2;
    const int nbx = (m + BX - 1) / BX, nby = (m + kSbY - 1) / kSbY;
    SbBlock B;
    int jx, jy;
    bool inx = true, iny = true;  // the interior's blocks in this dimension
    if (!WRAP) {
        jx = blockIdx.z, jy = blockIdx.y, B.tc = (int)blockIdx.x + 1;
    } else {
        const unsigned nyb = (unsigned)(nt + kSbY - 1) / kSbY;
        const unsigned n0 = (unsigned)nt * nyb, n1 = (unsigned)nt * (unsigned)nbx;
        unsigned b = blockIdx.x;
        if (b < n0) {
            B.tc = (int)(b % (unsigned)nt), jy = (int)(b / (unsigned)nt), jx = 0;
            inx = iny = false;
            B.ta0 = nt - 1, B.skipx = 0;
            B.tb0 = min(kSbY * jy, nt - kSbY), B.skipy = kSbY * jy - B.tb0;
        } else if (b < n0 + n1) {
            b -= n0;
            B.tc = (int)(b % (unsigned)nt), jx = (int)(b / (unsigned)nt), jy = 0;
            iny = false;
            B.tb0 = nt - 1, B.skipy = 0;
        } else {
            b -= n0 + n1;
            B.tc = (b & 1u) ? nt - 1 : 0;
            b >>= 1;
            jy = (int)(b % (unsigned)nby), jx = (int)(b / (unsigned)nby);
        }
    }
    if (inx) {
        const bool last = jx == nbx - 1;
        B.ta0 = last ? nt - 1 - BX : 1 + BX * jx;
        B.skipx = last ? BX * nbx - m : 0;
    }
    if (iny) {
        const bool last = jy == nby - 1;
        B.tb0 = last ? nt - 1 - kSbY : 1 + kSbY * jy;
        B.skipy = last ? kSbY * nby - m : 0;
    }
    return B;
}
static unsigned sb_wrap_blocks(unsigned nt) {
    const unsigned m = nt - 2, nbx = (m + 1) / 2, nby = (m + kSbY - 1) / kSbY;
    return nt * ((nt + kSbY - 1) / kSbY) + nt * nbx + 2 * nbx * nby;
}

template <int BX, int RUNGS, bool STATS, bool TABLDS, bool WRAP>
__global__ __launch_bounds__(64 * sb_waves(BX)) __attribute__((amdgpu_waves_per_eu(BX == 4 ? 8 : 4, 8))) void
k_sr_sweep_blocks(const double *__restrict__ pos_r, const unsigned *__restrict__ order_r,
                  const unsigned *__restrict__ off_r, double *__restrict__ dmom_r,
                  const double *__restrict__ pos_s, const unsigned *__restrict__ off_s,
                  const double *__restrict__ table, SrParams P) {
    constexpr int kSbX = BX, kSbCols = sb_cols(BX), kSbWaves = sb_waves(BX), kSbCap = sb_cap(BX);
    static_assert(kSbCols <= 96 && 2 * kSbWaves <= 32, "the prefix below: 64 + 32 columns; the groups: 32 bits");
    static_assert(!WRAP || (BX == 2 && !TABLDS), "the blocks across the faces: 2 x 2 tiles, 64 columns");
    constexpr int kLen = kSbCap + kSrSlack;
    constexpr int kPieces = (WRAP ? 2 : 1) * kSbCols;  // (WRAP: a column may wrap around in z)
    extern __shared__ double sb_lds[];
    double *sx = sb_lds, *sy = sx + kLen, *sz = sy + kLen, *soz = sz + kLen,
           *stab = soz + (WRAP ? kLen : 0);
    unsigned *p_beg = (unsigned *)(stab + (TABLDS ? kSbTable : 0)), *p_cnt = p_beg + kPieces,
             *p_off = p_cnt + kPieces, *grp_n = p_off + kPieces, *grp_b = grp_n + 2 * kSbWaves,
             *grp_n0 = grp_b + 2 * kSbWaves, *grp_gap = grp_n0 + 2 * kSbWaves;  // (ACT)
    int *p_oz = (int *)(grp_gap + 2 * kSbWaves);  // (WRAP) a piece's image in z: -1, 0, +1 boxes
    constexpr bool ACT = RUNGS == 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = P.nt, nc = 2 * nt;
    const SbBlock B = sb_block<BX, WRAP>(nt);
    const int ta0 = B.ta0, tb0 = B.tb0, tc = B.tc;
    // a cell index of the block's frame (it may lie beyond a face) -> the cell of the box, and
    // how many box lengths lie between them
    auto wrapc = [nc](int u) { return !WRAP ? u : (u < 0 ? u + nc : (u >= nc ? u - nc : u)); };
    auto shiftc = [nc](int u) { return !WRAP ? 0 : (u < 0 ? -1 : (u >= nc ? 1 : 0)); };
    // this wave's two receiver groups: cell columns (gx, gy) of the block's 8 x 4, cells 2 tc and
    // 2 tc + 1.  A group whose tile is left out (shared with the neighbour block, no active
    // receiver, taken by the dense tiles' sweep) has no receivers.
    // (scalars selected by the group number, not arrays indexed by it: those end up in scratch
    // memory and take the wave-uniformity of everything derived from them with them)
    // (ACT: the active rows of the group's two cells — n0 of the first, then `gap` rows of its
    // inactive particles, then the second cell's; rend counts the active ones from rbeg)
    unsigned rbeg0 = 0, rend0 = 0, rbeg1 = 0, rend1 = 0;
    unsigned n00 = 0, gap0 = 0, n01 = 0, gap1 = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int g = wave + kSbWaves * h, gx = g / (2 * kSbY), gy = g % (2 * kSbY);
        int ta = ta0 + (gx >> 1), tb = tb0 + (gy >> 1);
        if (WRAP) ta -= ta >= nt ? nt : 0, tb -= tb >= nt ? nt : 0;
        bool take = (gx >> 1) >= B.skipx && (gy >> 1) >= B.skipy;
        if (take && P.tile_active)
            take = P.tile_active[((unsigned)ta * nt + (unsigned)tb) * nt + (unsigned)tc] != 0;
        const unsigned rcell =
            ((unsigned)wrapc(2 * ta0 + gx) * nc + (unsigned)wrapc(2 * tb0 + gy)) * nc + 2 * tc;
        if (take) {
            const unsigned rb = __builtin_amdgcn_readfirstlane(off_r[rcell]);
            unsigned re, n0 = 0, gap = 0;
            if (ACT) {
                const unsigned b1 = __builtin_amdgcn_readfirstlane(off_r[rcell + 1]);
                n0 = __builtin_amdgcn_readfirstlane(P.nact[rcell]);
                gap = b1 - rb - n0;
                re = rb + n0 + __builtin_amdgcn_readfirstlane(P.nact[rcell + 1]);
            } else {
                re = __builtin_amdgcn_readfirstlane(off_r[rcell + 2]);
            }
            if (h == 0) rbeg0 = rb, rend0 = re, n00 = n0, gap0 = gap;
            else rbeg1 = rb, rend1 = re, n01 = n0, gap1 = gap;
        }
    }
    if (lane == 0) {
        grp_n[wave] = rend0 - rbeg0;
        grp_n[wave + kSbWaves] = rend1 - rbeg1;
        grp_b[wave] = rbeg0;
        grp_b[wave + kSbWaves] = rbeg1;
        if (ACT) {
            grp_n0[wave] = n00, grp_n0[wave + kSbWaves] = n01;
            grp_gap[wave] = gap0, grp_gap[wave + kSbWaves] = gap1;
        }
    }
    // supplier pieces: column (cx, cy) of the 12 x 8 around the block, cells 2 tc - 2 .. 2 tc + 3
    // (WRAP: cut in two where the column wraps around the box in z — piece kSbCols + column is
    // the part beyond the face, staged behind the first)
    if (tid < kPieces) {
        const int half = WRAP && tid >= kSbCols, col = tid - (half ? kSbCols : 0);
        const int cx = col / kSbColsY, cy = col % kSbColsY;
        const unsigned base =
            ((unsigned)wrapc(2 * ta0 - 2 + cx) * nc + (unsigned)wrapc(2 * tb0 - 2 + cy)) * nc;
        const int z0 = 2 * tc - 2, z1 = 2 * tc + 3;  // inclusive
        int a = z0, b = z1, oz = 0;                  // this piece's cells [a, b]
        if (WRAP) {
            if (z0 < 0) {
                if (half == 0) { a = z0 + nc; b = nc - 1; oz = 1; } else { a = 0; }
            } else if (z1 >= nc) {
                if (half == 0) { b = nc - 1; } else { a = 0; b = z1 - nc; oz = -1; }
            } else if (half) {
                b = a - 1;
            }
            p_oz[tid] = oz;
        }
        unsigned beg = 0, cnt = 0;
        if (b >= a) {
            beg = off_s[base + a];
            cnt = off_s[base + b + 1] - beg;
        }
        p_beg[tid] = beg;
        p_cnt[tid] = cnt;
    }
    // the first group's first chunk does not depend on the staging: its loads (and the Δmom it
    // will be added to) are in flight while the suppliers are staged
    SrChunk ch = {};
    double d0 = 0, d1 = 0, d2 = 0;
    if (rend0 > rbeg0) {
        ch = sr_chunk_load<RUNGS, BX == 2>(rbeg0, rend0, lane, pos_r, order_r, P, rbeg0, n00, gap0);
        if (ch.active && ch.sub == 0) {
            d0 = dmom_r[3 * (i64)ch.pi];
            d1 = dmom_r[3 * (i64)ch.pi + 1];
            d2 = dmom_r[3 * (i64)ch.pi + 2];
        }
    }
    __syncthreads();
    // which of the 32 groups have receivers (bit g): none -> nothing to do; some -> only the
    // columns within reach of one of them are staged (a block at the edge of the dense tiles,
    // or with few tiles on an active rung, would otherwise stage its neighbours' thousands of
    // suppliers for nobody)
    constexpr unsigned kAll = (unsigned)((1ull << (2 * kSbWaves)) - 1ull);
    const unsigned gm = (unsigned)__ballot(grp_n[lane & (2 * kSbWaves - 1)] != 0) & kAll;
    if (gm == 0) return;
    // The groups that have receivers are dealt out again, the k-th of them to wave k % 16: in a
    // block of which only a part takes part, its wavefronts share what there is instead of
    // keeping the two groups of their places.  (All 32 present: wave w keeps w and w + 16, and
    // the chunk it has loaded.)
    const int g_first = wave;
    const bool all_groups = gm == kAll;  // (the usual case: nothing to deal out or leave out)
    int g0 = wave, g1 = wave + kSbWaves;
    if (!all_groups) {
        g0 = g1 = -1;
        const int n = __popc(gm);
        unsigned t = gm;
        if (wave < n) {
            for (int i = 0; i < wave; i++) t &= t - 1;
            g0 = __builtin_ctz(t);
            if (wave + kSbWaves < n) {
                for (int i = 0; i < kSbWaves; i++) t &= t - 1;
                g1 = __builtin_ctz(t);
            }
        }
        rbeg0 = rend0 = rbeg1 = rend1 = 0;
        if (g0 >= 0) {
            rbeg0 = __builtin_amdgcn_readfirstlane(grp_b[g0]);
            rend0 = rbeg0 + __builtin_amdgcn_readfirstlane(grp_n[g0]);
            if (ACT) n00 = __builtin_amdgcn_readfirstlane(grp_n0[g0]),
                     gap0 = __builtin_amdgcn_readfirstlane(grp_gap[g0]);
        }
        if (g1 >= 0) {
            rbeg1 = __builtin_amdgcn_readfirstlane(grp_b[g1]);
            rend1 = rbeg1 + __builtin_amdgcn_readfirstlane(grp_n[g1]);
            if (ACT) n01 = __builtin_amdgcn_readfirstlane(grp_n0[g1]),
                     gap1 = __builtin_amdgcn_readfirstlane(grp_gap[g1]);
        }
    }
    auto reach = [gm, all_groups](int col) {  // column (cx, cy) is within reach of groups gx in
        if (all_groups) return true;          // [cx - 4, cx], gy in [cy - 4, cy]
        const int cx = col / kSbColsY, cy = col % kSbColsY;
        unsigned rows = 0;
#pragma unroll
        for (int d = 0; d < 5; d++) {
            const int gx = cx - d;
            if (gx >= 0 && gx < 2 * kSbX) rows |= gm >> (2 * kSbY * gx);
        }
        const unsigned ymask = ((2u << cy) - 1u) & ~((1u << max(cy - 4, 0)) - 1u) & ((1u << (2 * kSbY)) - 1u);
        return (rows & ymask) != 0;
    };
    // the table: its loads travel with those of the staging below (the barrier behind the
    // staging is the one the look-ups wait for)
    // (a block with few receivers — the edge of the dense tiles, the upper rungs — would pay
    // more for the copy than its look-ups gain: it reads the table where it is)
    const bool tab_lds = TABLDS && __popc(gm) >= kSbWaves / 2;
    if (tab_lds)
        for (int i = tid; i < P.table_n; i += 64 * kSbWaves) stab[i] = table[i];
    // exclusive prefix of the 96 column sizes by every wave for itself: lane l keeps the bounds
    // of columns l and 64 + l in registers, the range bounds below are v_readlane's
    // (the counts that are not needed are zeroed in place, by every wave with the same result:
    // a wave that reads another's zero would have made it one itself)
    const bool want0 = reach(lane), want1 = !WRAP && lane < kSbCols - 64 && reach(64 + lane);
    // (WRAP: 64 columns; the second count is the column's piece beyond the z face)
    const unsigned ca_ = want0 ? p_cnt[lane] : 0u,
                   cb_ = WRAP ? (want0 ? p_cnt[kSbCols + lane] : 0u) : (want1 ? p_cnt[64 + lane] : 0u);
    const unsigned c0 = WRAP ? ca_ + cb_ : ca_, c1 = WRAP ? 0u : cb_;
    p_cnt[lane] = ca_;
    if (WRAP) p_cnt[kSbCols + lane] = cb_;
    else if (lane < kSbCols - 64) p_cnt[64 + lane] = cb_;
    const unsigned i0 = sr_wave_scan(c0);                                   // inclusive
    const unsigned i1 = WRAP ? i0 : sr_wave_scan(c1) + __builtin_amdgcn_readlane(i0, 63);
    const unsigned e0 = i0 - c0, e1 = i1 - c1;                              // exclusive
    p_off[lane] = e0;  // (identical values from every wave: a wave reads what it wrote itself)
    if (WRAP) p_off[kSbCols + lane] = e0 + ca_;
    else if (lane < kSbCols - 64) p_off[64 + lane] = e1;
    const unsigned total = __builtin_amdgcn_readlane(i1, 63);
    const bool one_window = total <= (unsigned)kSbCap;
    // WRAP, y: the staged columns cy >= cyb lie one box length beyond those below (8 or more:
    // the block's columns do not cross a y face)
    const int shA_y = shiftc(2 * tb0 - 2);
    const int cyb = WRAP ? (shA_y + 1) * nc - (2 * tb0 - 2) : 99;
    SrCount cnt;
    for (unsigned w0 = 0; w0 < total; w0 += kSbCap) {
        const unsigned w1 = min(total, w0 + (unsigned)kSbCap);
        const int sw0 = __builtin_amdgcn_readfirstlane((int)w0),
                  sw1 = __builtin_amdgcn_readfirstlane((int)w1);
        if (w0) __syncthreads();  // everybody is done with the previous window
        // staging: 16 lanes per piece, a wave takes 4 pieces at a time.  (All of a lane's
        // rows loaded before the first is stored — every entry's piece found by bisection of the
        // prefix sums, one round trip to memory instead of a turn per 16 rows of a piece —
        // measured no faster with 2 x 2 tiles: 7.6 against 7.5 ms, and 70 registers.)
        for (int p0 = wave * 4; p0 < kPieces; p0 += 4 * kSbWaves) {
            const int p = p0 + (lane >> 4);
            const unsigned beg = p_beg[p], o0 = p_off[p], o1 = o0 + p_cnt[p];
            const unsigned lo = max(o0, w0), hi = min(o1, w1);  // the part inside this window
            const double oz = WRAP ? sr_image(p_oz[p], P.boxsize) : 0.0;
            for (unsigned tn = 0; __any(lo + 16u * tn < hi); tn++) {
                const unsigned q = lo + (lane & 15) + 16u * tn;
                if (q < hi) {
                    const i64 g = (i64)beg + (q - o0);
                    sx[q - w0] = pos_s[3 * g];
                    sy[q - w0] = pos_s[3 * g + 1];
                    sz[q - w0] = pos_s[3 * g + 2];
                    if (WRAP) soz[q - w0] = oz;
                }
            }
        }
        if (tid < kSrSlack) {  // the slack read by masked lanes: finite values
            sx[w1 - w0 + tid] = 0;
            sy[w1 - w0 + tid] = 0;
            sz[w1 - w0 + tid] = 0;
            if (WRAP) soz[w1 - w0 + tid] = 0;
        }
        __syncthreads();
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const int g = h ? g1 : g0, gx = g / (2 * kSbY), gy = g % (2 * kSbY);
            const unsigned rbeg = h ? rbeg1 : rbeg0, rend = h ? rend1 : rend0;
            const unsigned n0 = h ? n01 : n00, gap = h ? gap1 : gap0;
            const bool simple = one_window && rend - rbeg <= 64;  // one chunk, one window
            for (unsigned base = rbeg; base < rend; base += 64) {
                if (h || base != rbeg0 || w0 || g0 != g_first) {
                    ch = sr_chunk_load<RUNGS, BX == 2>(base, rend, lane, pos_r, order_r, P, rbeg, n0, gap);
                    if (simple && ch.active && ch.sub == 0) {
                        d0 = dmom_r[3 * (i64)ch.pi];
                        d1 = dmom_r[3 * (i64)ch.pi + 1];
                        d2 = dmom_r[3 * (i64)ch.pi + 2];
                    }
                }
                const int R = ch.R, S = ch.S, sub = ch.sub;
                const bool active = ch.active;
                if (!__any(active)) continue;
                const double xi = ch.xi, yi = ch.yi, zi = ch.zi;
                double ax = 0, ay = 0, az = 0;
                // (every lane walks the ranges, active or not: the bounds are v_readlane's of
                // registers whose 64 lanes must be live, i.e. uniform control flow)
                int ra[5], rb[5];
                SrWrapSeg ws[5];
                // the receivers' own box shifts (a group of tile 0 in a block that starts at
                // tile nt - 1 lies one box length up in the block's frame)
                const int shr_x = shiftc(2 * ta0 + gx), shr_y = shiftc(2 * tb0 + gy);
                const double oyA = sr_image(shr_y - shA_y, P.boxsize),
                             oyB = sr_image(shr_y - shA_y - 1, P.boxsize);
#pragma unroll
                for (int xg = 0; xg < 5; xg++) {
                    // first and last of the 5 columns of this x (one staged range)
                    const int ca = (gx + xg) * kSbColsY + gy, cb = ca + 4;
                    int ea, ib;
                    if (kSbCols <= 64 || ca < 64) ea = __builtin_amdgcn_readlane((int)e0, ca);
                    else ea = __builtin_amdgcn_readlane((int)e1, ca - 64);
                    if (kSbCols <= 64 || cb < 64) ib = __builtin_amdgcn_readlane((int)i0, cb);
                    else ib = __builtin_amdgcn_readlane((int)i1, cb - 64);
                    // (a range of another window: empty, and where this window's rows end)
                    ra[xg] = min(max(ea, sw0), sw1) - sw0;
                    rb[xg] = max(min(ib, sw1) - sw0, ra[xg]);
                    ws[xg].ox = 0.0, ws[xg].ys = 0;
                    if (WRAP) {
                        ws[xg].ox = sr_image(shr_x - shiftc(2 * ta0 - 2 + gx + xg), P.boxsize);
                        ws[xg].ys = __builtin_amdgcn_readfirstlane(
                            cyb <= gy ? -0x7fffffff
                            : cyb > gy + 4
                                ? 0x7fffffff
                                : __builtin_amdgcn_readlane((int)e0, (gx + xg) * kSbColsY + min(cyb, kSbColsY - 1)) - sw0);
                    }
                }
                if (tab_lds)
                    sr_cell_ranges<WRAP, STATS, 5>(ra, rb, sub, S, xi, yi, zi, sx, sy, sz, P.r2_max,
                                                   P.r2_index_scaling, stab, ax, ay, az, active,
                                                   cnt, ws, oyA, oyB, soz);
                else
                    sr_cell_ranges<WRAP, STATS, 5>(ra, rb, sub, S, xi, yi, zi, sx, sy, sz, P.r2_max,
                                                   P.r2_index_scaling, table, ax, ay, az, active,
                                                   cnt, ws, oyA, oyB, soz);
                // fold the S partial sums of each receiver (lanes rl, rl + R, ...) into lane rl: a
                // tree over the groups, ceil(log2 S) shuffle steps.  A node whose partner group
                // does not exist (sub + d >= S) reads lane 63 instead, which holds zeros whenever
                // such a node exists: R*S = 64 only for powers of two, where every partner exists.
                if (sub >= S) ax = ay = az = 0;
                for (int d = 1; d < S; d <<= 1) {  // S is wave-uniform
                    const int src = sub + d < S ? lane + d * R : 63;
                    ax += __shfl(ax, src);
                    ay += __shfl(ay, src);
                    az += __shfl(az, src);
                }
                ax *= ch.factor;  // gravity.py:321 (total_factor = factors[rung] * table[...])
                ay *= ch.factor;
                az *= ch.factor;
                if (active && sub == 0) {
                    const i64 o = 3 * (i64)ch.pi;
                    if (simple) {  // the Δmom read with the chunk
                        dmom_r[o] = d0 + ax;
                        dmom_r[o + 1] = d1 + ay;
                        dmom_r[o + 2] = d2 + az;
                    } else {
                        dmom_r[o] += ax;
                        dmom_r[o + 1] += ay;
                        dmom_r[o + 2] += az;
                    }
                }
            }
        }
    }
    if (STATS && lane == 0) {
        atomicAdd(&P.stats[0], (unsigned long long)cnt.tests);
        atomicAdd(&P.stats[1], (unsigned long long)cnt.hits);
        atomicAdd(&P.stats[2], (unsigned long long)cnt.trips);
    }
}

// ===========================================================================
// A handful of active receivers (the top rungs of a base step's sub-steps: sixteen of the 32
// sub-steps of a run with six rungs kick the few particles of the highest one): no cell list at
// all.  Every workgroup takes a slice of the suppliers, tests each against the K <= 8 receivers
// (nearest periodic image: (xi - xj) + offset with offset = -L, 0 or +L, the same operations as
// the tile offsets of interactions.py:1615-1621 give for a range below a quarter of the box) with
// the sweep's pair arithmetic, and reduces its sums in a fixed tree; a second kernel adds the
// slices' partials in order and applies the receiver's rung factor.  Bit-reproducible; equal to
// the cells sweep up to the order of the additions.
// ===========================================================================
constexpr int kSrSparseMax = 8, kSrSparseBlocks = 1024;
__global__ __launch_bounds__(256) void k_sr_sparse(const double *__restrict__ pos_r,
                                                   const i64 *__restrict__ active, int K,
                                                   const double *__restrict__ pos_s, i64 n_s,
                                                   const double *__restrict__ table,
                                                   SrParams P, double *__restrict__ partial) {
    __shared__ double rx[kSrSparseMax], ry[kSrSparseMax], rz[kSrSparseMax];
    __shared__ double red[4][3 * kSrSparseMax];
    if (threadIdx.x < (unsigned)K) {
        const i64 i = active[threadIdx.x];
        rx[threadIdx.x] = pos_r[3 * i];
        ry[threadIdx.x] = pos_r[3 * i + 1];
        rz[threadIdx.x] = pos_r[3 * i + 2];
    }
    __syncthreads();
    const i64 per = (n_s + gridDim.x - 1) / gridDim.x;
    const i64 lo = (i64)blockIdx.x * per, hi = lo + per < n_s ? lo + per : n_s;
    const double L = P.boxsize, half = 0.5 * L;
    double acc[kSrSparseMax][3] = {};
    for (i64 j = lo + threadIdx.x; j < hi; j += 256) {
        const double sx = pos_s[3 * j], sy = pos_s[3 * j + 1], sz = pos_s[3 * j + 2];
#pragma unroll
        for (int r = 0; r < kSrSparseMax; r++) {
            if (r >= K) break;
            double x = rx[r] - sx, y = ry[r] - sy, z = rz[r] - sz;   // interactions.py:1787-1789
            x = x + (x > half ? -L : (x < -half ? L : 0.0));       // + the image's offset
            y = y + (y > half ? -L : (y < -half ? L : 0.0));
            z = z + (z > half ? -L : (z < -half ? L : 0.0));
            const double r2 = sr_r2(x, y, z);               // gravity.py:306
            if (r2 <= P.r2_max) {                                   // gravity.py:311
                const double t = table[(unsigned)(int)(r2 * P.r2_index_scaling)];
                acc[r][0] = __builtin_fma(x, t, acc[r][0]);
                acc[r][1] = __builtin_fma(y, t, acc[r][1]);
                acc[r][2] = __builtin_fma(z, t, acc[r][2]);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < kSrSparseMax; r++) {
        if (r >= K) break;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            double v = acc[r][d];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0) red[wave][3 * r + d] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)(3 * K)) {
        const int e = threadIdx.x;
        partial[(i64)e * gridDim.x + blockIdx.x] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    }
}
__global__ __launch_bounds__(64) void k_sr_sparse_final(const double *__restrict__ partial, int nb,
                                                        const i64 *__restrict__ active, int K,
                                                        SrParams P, double *__restrict__ dmom_r) {
    // one wave: lane e < 3K adds the nb partials of its entry in index order
    const int e = threadIdx.x;
    if (e >= 3 * K) return;
    double sum = 0;
    for (int b = 0; b < nb; b++) sum += partial[(i64)e * nb + b];
    const i64 i = active[e / 3];
    const double factor = P.rung ? P.factors[P.rung_jumped[i]] : P.factor;
    dmom_r[3 * i + e % 3] += sum * factor;   // gravity.py:321-349 (factors[rung] * x * f)
}
int cgk_shortrange_sparse(cg_ctx *c, const double *pos_r, const i64 *active, int K, double *dmom_r,
                          const double *pos_s, i64 n_s, const double *table,
                          double r2_index_scaling, double r2_max, double factor,
                          const double *factors, const signed char *rung_jumped) {
    if (K < 1 || K > kSrSparseMax) {
        cg_set_error("cg_shortrange_sparse: %d active receivers (1..%d)", K, kSrSparseMax);
        return 1;
    }
    if (n_s == 0) return 0;
    SrParams P{c->p.boxsize, r2_index_scaling, r2_max, factor, 0,
               factors,      rung_jumped /* non-null = rungs in use */, rung_jumped, 0, nullptr,
               nullptr,      0};
    if (!factors) P.rung = nullptr;
    if (!c->sr_sparse_partial)
        CG_HIP(hipMalloc((void **)&c->sr_sparse_partial,
                         sizeof(double) * 3 * kSrSparseMax * kSrSparseBlocks));
    i64 nbl = (n_s + 255) / 256;
    const int nb = (int)(nbl < kSrSparseBlocks ? nbl : kSrSparseBlocks);
    hipLaunchKernelGGL(k_sr_sparse, dim3(nb), dim3(256), 0, c->stream, pos_r, active, K, pos_s, n_s,
                       table, P, c->sr_sparse_partial);
    CG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sr_sparse_final, dim3(1), dim3(64), 0, c->stream, c->sr_sparse_partial, nb,
                       active, K, P, dmom_r);
    CG_LAUNCH_CHECK();
    return 0;
}

// ===========================================================================
// A sub-step with FEW active receivers (the upper rungs of a base step's sub-steps: at 256^3 the
// eight sub-steps that kick the highest of five rungs have 1 % of the particles active): the
// blocks above stage ~1000 suppliers for a receiver or two and wait for a table look-up per
// trip with nobody to hide it (1.2 ms for 6e7 pair tests).  Here the active receivers are listed
// (k_sr_active_cells, from the nact words of the active-first list: cell and row of each) and
// every one gets a wavefront of its own — a cell in a clump may hold hundreds of them: the 5 x 5
// columns x 5 cells around the cell are 25 runs of the suppliers' list (50 where the z range
// wraps around the box) read where they are — 16 lanes per run, four runs per trip, nothing
// staged, no barrier.  Same pair arithmetic ((xi - xj) + offset, r2, table
// index bit-identical to the reference's); a receiver's sum is reduced over the wave in a fixed
// order.
// ===========================================================================
constexpr int kSaRuns = 52;  // 25 columns x 2 pieces, rounded up to whole trips of 4
// (one atomic on the list's counter per 4096 cells: an atomic per wavefront — 94,000 of them on
// ONE address at 256^3 — took 0.9 ms, three times the sweep it feeds)
constexpr int kSaPerThread = 16;
// list[i] = the cell of the i-th active receiver, rows[i] = its row in the receivers' list
// (a cell's nact active rows are its first ones)
__global__ __launch_bounds__(256) void k_sr_active_cells(const unsigned *__restrict__ nact,
                                                         const unsigned *__restrict__ off_r,
                                                         unsigned ncells,
                                                         unsigned *__restrict__ list,
                                                         unsigned *__restrict__ rows,
                                                         unsigned *__restrict__ count,
                                                         unsigned cap,
                                                         unsigned *__restrict__ err_flags) {
    __shared__ unsigned w_tot[4], w_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave w of the workgroup takes 1024 consecutive cells, 64 at a time
    const unsigned first = (blockIdx.x * 4u + (unsigned)wave) * (64u * kSaPerThread);
    unsigned mine[kSaPerThread], total = 0;
#pragma unroll
    for (int i = 0; i < kSaPerThread; i++) {
        const unsigned c = first + 64u * i + lane;
        mine[i] = c < ncells ? nact[c] : 0u;
        total += mine[i];
    }
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    if (lane == 0) w_tot[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) w_base = atomicAdd(count, w_tot[0] + w_tot[1] + w_tot[2] + w_tot[3]);
    __syncthreads();
    unsigned off = w_base;
    for (int w = 0; w < wave; w++) off += w_tot[w];
#pragma unroll
    for (int i = 0; i < kSaPerThread; i++) {
        const unsigned n = mine[i];
        const unsigned incl = sr_wave_scan(n);
        if (n) {
            const unsigned c = first + 64u * i + lane, r0 = off_r[c];
            unsigned slot = off + incl - n;
            for (unsigned k = 0; k < n && slot < cap; k++, slot++) {
                list[slot] = c;
                rows[slot] = r0 + k;
            }
            // (more active receivers than the caller's bound: they are not swept, and it shows)
            if (off + incl > cap) atomicOr(err_flags, (unsigned)CG_ERR_ACTIVE_OVERFLOW);
        }
        off += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    }
}

template <bool STATS>
__global__ __launch_bounds__(256) void k_sr_sweep_active_cells(
    const double *__restrict__ pos_r, const unsigned *__restrict__ order_r,
    const unsigned *__restrict__ off_r, double *__restrict__ dmom_r,
    const double *__restrict__ pos_s, const unsigned *__restrict__ off_s,
    const double *__restrict__ table, SrParams P, const unsigned *__restrict__ list,
    const unsigned *__restrict__ rows, const unsigned *__restrict__ nlist, unsigned cap) {
    __shared__ unsigned r_beg[4][kSaRuns], r_cnt[4][kSaRuns];
    __shared__ int r_img[4][kSaRuns];  // the run's image: (ox + 1) | (oy + 1) << 2 | (oz + 1) << 4
    __shared__ unsigned r_pre[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (no barrier below; the count may exceed the list's room — CG_ERR_ACTIVE_OVERFLOW — the list
    // holds the first `cap`)
    const unsigned nrec = min((unsigned)__builtin_amdgcn_readfirstlane((int)*nlist), cap);
    // Neighbours on the list read the same supplier runs: an XCD (workgroup b runs on XCD b % 8)
    // takes a contiguous eighth of the receivers, so that its L2 serves them — dealt out in turn,
    // each of the eight L2s fetched every run
    const unsigned per_xcd = ((nrec + 3u) / 4u + 7u) / 8u;
    if ((blockIdx.x >> 3) >= per_xcd) return;
    const unsigned idx = ((blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3)) * 4u + (unsigned)wave;
    if (idx >= nrec) return;
    const unsigned cell = (unsigned)__builtin_amdgcn_readfirstlane((int)list[idx]);
    const int nt = P.nt, nc = 2 * nt;
    const int Z = (int)(cell % (unsigned)nc), Y = (int)((cell / (unsigned)nc) % (unsigned)nc),
              X = (int)(cell / ((unsigned)nc * (unsigned)nc));
    // (the dense tiles' sweep has taken the tile)
    if (P.tile_active &&
        !P.tile_active[((unsigned)(X >> 1) * nt + (unsigned)(Y >> 1)) * nt + (unsigned)(Z >> 1)])
        return;
    // the runs: slot = column (dx, dy) of the 5 x 5, + 25 for the piece beyond a z face
    const bool zwrap = Z < 2 || Z + 2 >= nc;  // (uniform)
    // (uniform) does any of the receiver's runs lie beyond a face of the box?  (17 of a trip's
    // ~70 vector instructions turn the image bits into offsets; 6 % of the cells need them)
    const bool faces = zwrap || X < 2 || X + 2 >= nc || Y < 2 || Y + 2 >= nc;
    if (lane < kSaRuns) {
        const int half = lane >= 25, col = lane - 25 * half;
        unsigned beg = 0, cnt = 0;
        int img = 1 | 1 << 2 | 1 << 4;
        if (lane < 50 && (half == 0 || zwrap)) {
            int gx = X + col / 5 - 2, gy = Y + col % 5 - 2;
            int ox = 0, oy = 0, oz = 0;
            if (gx < 0) { gx += nc; ox = 1; } else if (gx >= nc) { gx -= nc; ox = -1; }
            if (gy < 0) { gy += nc; oy = 1; } else if (gy >= nc) { gy -= nc; oy = -1; }
            const int z0 = Z - 2, z1 = Z + 2;  // inclusive
            int a = z0, b = z1;
            if (z0 < 0) {
                if (half == 0) { a = z0 + nc; b = nc - 1; oz = 1; } else { a = 0; }
            } else if (z1 >= nc) {
                if (half == 0) { b = nc - 1; } else { a = 0; b = z1 - nc; oz = -1; }
            }
            const unsigned base = ((unsigned)gx * nc + (unsigned)gy) * nc;
            beg = off_s[base + a];
            cnt = off_s[base + b + 1] - beg;
            img = (ox + 1) | (oy + 1) << 2 | (oz + 1) << 4;
        }
        r_beg[wave][lane] = beg;
        r_cnt[wave][lane] = cnt;
        r_img[wave][lane] = img;
    }
    // The runs as ONE sequence of suppliers: r_pre[s] = suppliers in the runs before s (the
    // wave's own scan of the counts it has just written: lanes 0..51 hold them).  A lane takes
    // supplier t = trip * 64 + lane and finds its run by bisection of the 64 prefix sums (six
    // LDS reads) — 6 trips of 64 for the ~350 suppliers of a receiver instead of 11 trips of
    // 4 runs x 16 lanes, most of them half empty (0.48 of the lane slots used).
    {
        const unsigned mine = lane < kSaRuns ? r_cnt[wave][lane] : 0u;
        const unsigned incl = sr_wave_scan(mine);
        r_pre[wave][lane] = incl - mine;   // (lanes >= 52: the total)
    }
    const unsigned total = r_pre[wave][63];
    const double L = P.boxsize;
    SrCount cnt;
    {
        const i64 row = (i64)(unsigned)__builtin_amdgcn_readfirstlane((int)rows[idx]);
        const double xi = pos_r[3 * row], yi = pos_r[3 * row + 1], zi = pos_r[3 * row + 2];
        // (the receiver's particle index and its factor: asked for now, needed after the loop —
        // as a chain behind the loop they were three round trips at the end of every wavefront)
        const i64 o = 3 * (i64)order_r[row];
        const double f = P.factors[P.rj_sorted ? P.rj_sorted[row] : P.rung_jumped[o / 3]];  // gravity.py:318-349
        double ax = 0, ay = 0, az = 0;
        // One trip = 64 suppliers: the lane's run (six LDS reads one behind the other), its
        // supplier's row (a round trip to memory), the look-up of a pair in range (another) — a
        // wavefront spent its time waiting for them in turn.  Two trips go together (their chains
        // side by side), and the rows of the NEXT two are asked for before this pair's look-ups
        // are waited for; a lane beyond the end reads the last supplier and counts for nothing.
        // (All trips' loads issued together — 126 registers, four wavefronts per SIMD — measured
        // slower than a trip at a time at eight.)
        struct Row { double x, y, z; int sl; bool valid; };
        auto ask = [&](unsigned t0, Row &ra, Row &rb) {
            const unsigned ta = t0 + (unsigned)lane, tb = ta + 64u;
            ra.valid = ta < total, rb.valid = tb < total;
            const unsigned ua = min(ta, total - 1u), ub = min(tb, total - 1u);
            // the runs of the two suppliers: the last s with r_pre[s] <= t (side by side)
            int sa = 0, sb = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const unsigned pa = r_pre[wave][sa + step], pb = r_pre[wave][sb + step];
                if (pa <= ua) sa += step;   // (entries beyond the runs hold the total)
                if (pb <= ub) sb += step;
            }
            sa = min(sa, kSaRuns - 1), sb = min(sb, kSaRuns - 1);
            const i64 ga = (i64)r_beg[wave][sa] + (ua - r_pre[wave][sa]),
                      gb = (i64)r_beg[wave][sb] + (ub - r_pre[wave][sb]);
            ra.x = pos_s[3 * ga], ra.y = pos_s[3 * ga + 1], ra.z = pos_s[3 * ga + 2];
            rb.x = pos_s[3 * gb], rb.y = pos_s[3 * gb + 1], rb.z = pos_s[3 * gb + 2];
            ra.sl = sa, rb.sl = sb;
        };
        if (total) {
            Row c0, c1;
            ask(0, c0, c1);
            for (unsigned t0 = 0; t0 < total; t0 += 128) {
                Row n0, n1;
                ask(t0 + 128, n0, n1);
                double x[2], y[2], z[2], r2[2], tv[2];
                bool hit[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const Row &c = j ? c1 : c0;
                    x[j] = xi - c.x;                   // interactions.py:1787-1789
                    y[j] = yi - c.y;
                    z[j] = zi - c.z;
                    if (faces) {                       // gravity.py:299-302 (elsewhere: + 0)
                        const int img = r_img[wave][c.sl];
                        x[j] += (double)((img & 3) - 1) * L;
                        y[j] += (double)((img >> 2 & 3) - 1) * L;
                        z[j] += (double)((img >> 4 & 3) - 1) * L;
                    }
                    r2[j] = sr_r2(x[j], y[j], z[j]);       // gravity.py:306
                    hit[j] = c.valid && r2[j] <= P.r2_max; // gravity.py:311
                    if (STATS) {
                        const bool any = __builtin_amdgcn_readfirstlane((int)(t0 + 64u * j < total));
                        cnt.tests += (unsigned)__popcll(__ballot(c.valid));
                        cnt.hits += (unsigned)__popcll(__ballot(hit[j]));
                        cnt.trips += any ? 1u : 0u;
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    tv[j] = 0.0;
                    if (hit[j]) tv[j] = table[(unsigned)(int)(r2[j] * P.r2_index_scaling)];
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    ax = __builtin_fma(x[j], tv[j], ax);
                    ay = __builtin_fma(y[j], tv[j], ay);
                    az = __builtin_fma(z[j], tv[j], az);
                }
                c0 = n0, c1 = n1;
            }
        }
        // the wave's lanes added in a fixed order (DPP: rows of 16, then the row totals)
        auto wave_sum = [](double v) {
#define SA_DPP_ADD(ctrl, rows)                                                                    \
    do {                                                                                          \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rows, 0xf, false); \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rows, 0xf, false); \
        v += __hiloint2double(hi_, lo_);                                                          \
    } while (0)
            SA_DPP_ADD(0x111, 0xf);
            SA_DPP_ADD(0x112, 0xf);
            SA_DPP_ADD(0x114, 0xf);
            SA_DPP_ADD(0x118, 0xf);
            SA_DPP_ADD(0x142, 0xa);
            SA_DPP_ADD(0x143, 0xc);
#undef SA_DPP_ADD
            return v;  // (the total in lane 63)
        };
        ax = wave_sum(ax), ay = wave_sum(ay), az = wave_sum(az);
        if (lane == 63) {
            dmom_r[o] += ax * f;
            dmom_r[o + 1] += ay * f;
            dmom_r[o + 2] += az * f;
        }
    }
    if (STATS && lane == 0) {
        atomicAdd(&P.stats[0], (unsigned long long)cnt.tests);
        atomicAdd(&P.stats[1], (unsigned long long)cnt.hits);
        atomicAdd(&P.stats[2], (unsigned long long)cnt.trips);
    }
}

// the instantiation for (shape, rungs mode, statistics, table in LDS, across the faces)
typedef void (*SbKernel)(const double *, const unsigned *, const unsigned *, double *,
                         const double *, const unsigned *, const double *, SrParams);
template <int BX, bool TABLDS, bool WRAP>
static SbKernel sb_kernel(int mode, bool stats) {
    switch (mode) {
        case 0: return stats ? k_sr_sweep_blocks<BX, 0, true, TABLDS, WRAP> : k_sr_sweep_blocks<BX, 0, false, TABLDS, WRAP>;
        case 1: return stats ? k_sr_sweep_blocks<BX, 1, true, TABLDS, WRAP> : k_sr_sweep_blocks<BX, 1, false, TABLDS, WRAP>;
        default:
            return stats ? k_sr_sweep_blocks<BX, 2, true, TABLDS, WRAP> : k_sr_sweep_blocks<BX, 2, false, TABLDS, WRAP>;
    }
}

// nact_r, rj_sorted_r: the receivers' list was made with the active particles first
// (cgk_shortrange_cells with this rung array and this lowest active rung); null: a plain list.
// n_active_max >= 0: no more than that many receivers are active and the caller wants them swept
// cell by cell (k_sr_sweep_active_cells) instead of in blocks of tiles
int cgk_shortrange_sweep_cells(cg_ctx *c, const double *pos_r_sorted, const unsigned *order_r,
                               const unsigned *off_r, double *dmom_r, const double *pos_s_sorted,
                               const unsigned *off_s, i64 nt, const double *table,
                               double r2_index_scaling, double r2_max, double factor,
                               const double *factors, const signed char *rung,
                               const signed char *rung_jumped, int lowest_active,
                               const unsigned *nact_r, const signed char *rj_sorted_r,
                               i64 n_active_max) {
    SrParams P{c->p.boxsize, r2_index_scaling, r2_max, factor, (int)nt,
               factors,      rung,             rung_jumped, lowest_active, nullptr, nullptr,
               (int)(r2_max * r2_index_scaling) + 1, nullptr, nullptr};
    const bool partial = rung && lowest_active > 0;
    // (the blocks take the active-first form only with the jumped rung indices in list order; a
    // list without them is swept as a plain list — its order inside the cells does not matter
    // there — unless the caller asks for the sweep by active receiver)
    const bool act = partial && nact_r && (rj_sorted_r || n_active_max >= 0);
    if (act) P.nact = nact_r, P.rj_sorted = rj_sorted_r;
    if (partial && !act) {
        // which tiles have a receiver on an active rung (the others leave at once)
        const size_t ntl = (size_t)nt * nt * nt;
        if (c->sr_tile_active_cap < ntl) {
            (void)hipFree(c->sr_tile_active);
            c->sr_tile_active = nullptr;
            c->sr_tile_active_cap = 0;
            CG_HIP(hipMalloc((void **)&c->sr_tile_active, ntl));
            c->sr_tile_active_cap = ntl;
        }
        hipLaunchKernelGGL(k_sr_tile_activity, dim3((unsigned)((ntl + 255) / 256)), dim3(256), 0,
                           c->stream, order_r, off_r, rung, lowest_active, (int)nt,
                           c->sr_tile_active);
        CG_LAUNCH_CHECK();
        P.tile_active = c->sr_tile_active;
    }
    // The densely populated tiles go to the sweep with sub-cell order and box culling
    // (cg_shortrange_dense.hip); `take` then keeps this sweep off them.  With a subset of active
    // rungs: the tiles that hold many ACTIVE receivers (the upper rungs live where the
    // particles are dense, and are kicked 2^rung times per base step).
    const unsigned char *take = nullptr;
    if (cgk_shortrange_dense(c, pos_r_sorted, order_r, off_r, dmom_r, pos_s_sorted, off_s, nt,
                             table, r2_index_scaling, r2_max, factor, factors, rung, rung_jumped,
                             rung ? lowest_active : 0, P.tile_active, &take))
        return 1;
    if (take) P.tile_active = take;
    const unsigned m = (unsigned)(nt - 2);  // (nt >= 4: checked by the caller)
    // Two launches that touch different receivers: the blocks across the box faces go to a side
    // stream (forked from and joined back into the context's stream) and run beside the
    // interior's.
    if (!c->sr_fork) {
        CG_HIP(hipEventCreateWithFlags(&c->sr_fork, hipEventDisableTiming));
        CG_HIP(hipStreamCreateWithFlags(&c->sr_streams[0], hipStreamNonBlocking));
        CG_HIP(hipEventCreateWithFlags(&c->sr_join[0], hipEventDisableTiming));
    }
    const int mode = !rung ? 0 : (act ? 2 : 1);
    P.stats = c->sr_stats;
    if (act && n_active_max >= 0) {
        // few active receivers: one wavefront per cell that holds one
        const i64 ncells = 8 * nt * nt * nt;
        const size_t need = 4 * (2 * (size_t)n_active_max + 64);
        if (need > c->sr_active_bytes) {
            CG_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(c->sr_active);
            c->sr_active = nullptr;
            c->sr_active_bytes = 0;
            CG_HIP(hipMalloc((void **)&c->sr_active, need));
            c->sr_active_bytes = need;
        }
        unsigned *count = c->sr_active, *list = c->sr_active + 64, *rows = list + n_active_max;
        CG_HIP(hipMemsetAsync(count, 0, 4, c->stream));
        if (n_active_max > 0) {
            hipLaunchKernelGGL(k_sr_active_cells,
                               dim3((unsigned)((ncells + 256 * kSaPerThread - 1) / (256 * kSaPerThread))), dim3(256),
                               0, c->stream, nact_r, off_r, (unsigned)ncells, list, rows, count,
                               (unsigned)n_active_max, c->err_flags);
            CG_LAUNCH_CHECK();
            hipLaunchKernelGGL(P.stats ? k_sr_sweep_active_cells<true> : k_sr_sweep_active_cells<false>,
                               dim3((unsigned)(((n_active_max + 3) / 4 + 7) / 8 * 8)), dim3(256), 0,
                               c->stream, pos_r_sorted, order_r, off_r, dmom_r, pos_s_sorted, off_s, table, P,
                               list, rows, count, (unsigned)n_active_max);
            CG_LAUNCH_CHECK();
        }
        if (take && cgk_shortrange_dense_join(c)) return 1;
        return 0;
    }
    {
        CG_HIP(hipEventRecord(c->sr_fork, c->stream));
        CG_HIP(hipStreamWaitEvent(c->sr_streams[0], c->sr_fork, 0));
        SbKernel faces = sb_kernel<2, false, true>(mode, P.stats != nullptr);
        hipLaunchKernelGGL(faces, dim3(sb_wrap_blocks((unsigned)nt)), dim3(64 * sb_waves(2)),
                           sb_lds_bytes(2, false, true), c->sr_streams[0], pos_r_sorted, order_r,
                           off_r, dmom_r, pos_s_sorted, off_s, table, P);
        CG_LAUNCH_CHECK();
        CG_HIP(hipEventRecord(c->sr_join[0], c->sr_streams[0]));
    }
    {
        // 4 x 2 tiles per workgroup; 2 x 2 for the sub-steps of the upper rungs and for boxes of
        // fewer than 6 tiles a side
        // (an active-first list is swept in blocks only when many receivers are active — the
        // sparse sub-steps go receiver by receiver — and then the 4 x 2 shape wins as it does for a
        // full sweep: 4.7 against 5.2 ms with half the receivers active, 3.8 against 4.1 with a
        // quarter)
        const bool small = m < 4 || (partial && !act);
        const bool lds = !small && P.table_n <= kSbTable;
        const int bx = small ? 2 : 4;
        const unsigned nbx = (m + bx - 1) / bx, nby = (m + kSbY - 1) / kSbY;
        SbKernel blocks = small ? sb_kernel<2, false, false>(mode, P.stats != nullptr)
                          : lds ? sb_kernel<4, true, false>(mode, P.stats != nullptr)
                                : sb_kernel<4, false, false>(mode, P.stats != nullptr);
        const size_t bytes = sb_lds_bytes(bx, lds, false);
        // (the attribute belongs to the function ON A DEVICE; setting it again costs nothing
        // next to a sweep)
        if (bytes > 64 * 1024)
            CG_HIP(hipFuncSetAttribute((const void *)blocks,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL(blocks, dim3(m, nby, nbx), dim3(64 * sb_waves(bx)), bytes, c->stream,
                           pos_r_sorted, order_r, off_r, dmom_r, pos_s_sorted, off_s, table, P);
    }
    CG_LAUNCH_CHECK();
    CG_HIP(hipStreamWaitEvent(c->stream, c->sr_join[0], 0));
    if (take && cgk_shortrange_dense_join(c)) return 1;
    return 0;
}
