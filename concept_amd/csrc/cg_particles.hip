// cg_particles.hip — particle-side kernels: drift (A11) and the tile sort
// that keeps particle memory in mesh-tile order.
#include <hipcub/hipcub.hpp>

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

// ---------------------------------------------------------------------------
// A11 drift: pos[r] = mod(pos[r] + mom[r]*dt_over_mass, boxsize) for all 3N
// reals (species.py:2194-2196).  mod is the reference's pure-Python one
// (commons.py:5103-5110): numpy floored modulo, then a result that rounded
// to exactly boxsize becomes 0.  Pure streaming kernel: 2 reads + 1 write of
// 8 B per real, 16-B vector accesses.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_drift(double *__restrict__ pos,
                                               const double *__restrict__ mom, i64 n3,
                                               double dt_over_mass, double L) {
    i64 stride = (i64)gridDim.x * blockDim.x;
    i64 nv = n3 / 2;
    double2 *p2 = (double2 *)pos;
    const double2 *m2 = (const double2 *)mom;
    for (i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
        double2 p = p2[v], m = m2[v];
        p.x = ref_mod(p.x + m.x * dt_over_mass, L);
        p.y = ref_mod(p.y + m.y * dt_over_mass, L);
        p2[v] = p;
    }
    if ((n3 & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        pos[n3 - 1] = ref_mod(pos[n3 - 1] + mom[n3 - 1] * dt_over_mass, L);
}

int cgk_drift(cg_ctx *c, double *pos, const double *mom, i64 n, double dt_over_mass) {
    i64 n3 = 3 * n;
    CG_CHECK(((uintptr_t)pos % 16 == 0) && ((uintptr_t)mom % 16 == 0),
             "cg_drift: particle arrays must be 16-byte aligned");
    i64 blocks = (n3 / 2 + 255) / 256;  // no cap: one workgroup per 4 KiB streams fastest
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_drift, dim3((unsigned)blocks), dim3(256), 0, c->stream, pos, mom, n3,
                       dt_over_mass, c->p.boxsize);
    CG_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// measure(component, 'v_rms' | 'v_max') of a particle component (analysis.py:3902-3910,
// 3965-3972): sum of mom[r]^2 over all 3N reals and the largest |mom_i|^2 of a particle — the
// inputs of the time loop's PM / P3M step-size limiters (main.py:842-912).  Two stages so that
// the sum has a fixed order (bit-reproducible): kMeasureBlocks workgroups each reduce a
// contiguous share to one partial, one workgroup adds the partials in index order.
// ---------------------------------------------------------------------------
constexpr int kMeasureBlocks = 1024;
__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off));
    return v;
}
__global__ __launch_bounds__(256) void k_measure_mom(const double *__restrict__ mom, i64 n,
                                                     double *__restrict__ partial) {
    __shared__ double s_sum[4], s_max[4];
    const i64 per = (n + gridDim.x - 1) / gridDim.x;
    const i64 p0 = (i64)blockIdx.x * per, p1 = p0 + per < n ? p0 + per : n;
    double sum = 0, mx = 0;
    for (i64 p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        double x = mom[3 * p], y = mom[3 * p + 1], z = mom[3 * p + 2];
        double m2 = x * x + y * y + z * z;
        sum += m2;
        mx = fmax(mx, m2);
    }
    sum = wave_sum(sum);
    mx = wave_max(mx);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_sum[w] = sum;
        s_max[w] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        partial[2 * blockIdx.x + 1] = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
    }
}
__global__ __launch_bounds__(64) void k_measure_final(const double *__restrict__ partial, int nb,
                                                      double *__restrict__ out) {
    // lane l adds partials l, l + 64, ... in order; then the fixed shuffle tree
    double sum = 0, mx = 0;
    for (int b = threadIdx.x; b < nb; b += 64) {
        sum += partial[2 * b];
        mx = fmax(mx, partial[2 * b + 1]);
    }
    sum = wave_sum(sum);
    mx = wave_max(mx);
    if (threadIdx.x == 0) {
        out[0] = sum;
        out[1] = mx;
    }
}

// The same for particles kept in tile regions with gaps (the streaming form of the time loop):
// workgroup b walks the regions [first, first + per) region by region, live rows only.
__global__ __launch_bounds__(256) void k_measure_mom_regions(const double *__restrict__ mom,
                                                             const unsigned *__restrict__ start,
                                                             const unsigned *__restrict__ count,
                                                             i64 nregions,
                                                             double *__restrict__ partial) {
    __shared__ double s_sum[4], s_max[4];
    const i64 per = (nregions + gridDim.x - 1) / gridDim.x;
    const i64 r0 = (i64)blockIdx.x * per, r1 = r0 + per < nregions ? r0 + per : nregions;
    double sum = 0, mx = 0;
    for (i64 r = r0; r < r1; r++) {
        const i64 p0 = start[r], n = count ? count[r] : start[r + 1] - start[r];
        for (i64 k = threadIdx.x; k < n; k += blockDim.x) {
            const i64 p = p0 + k;
            double x = mom[3 * p], y = mom[3 * p + 1], z = mom[3 * p + 2];
            double m2 = x * x + y * y + z * z;
            sum += m2;
            mx = fmax(mx, m2);
        }
    }
    sum = wave_sum(sum);
    mx = wave_max(mx);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_sum[w] = sum;
        s_max[w] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        partial[2 * blockIdx.x + 1] = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
    }
}

int cgk_measure_mom_regions(cg_ctx *c, const double *mom, const unsigned *start,
                            const unsigned *count, double *out, double *scratch) {
    const i64 nregions = 8 * c->ntiles;
    int nb = (int)(nregions < kMeasureBlocks ? nregions : kMeasureBlocks);
    hipLaunchKernelGGL(k_measure_mom_regions, dim3(nb), dim3(256), 0, c->stream, mom, start, count,
                       nregions, scratch);
    CG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_measure_final, dim3(1), dim3(64), 0, c->stream, scratch, nb, out);
    CG_LAUNCH_CHECK();
    return 0;
}

int cgk_measure_mom(cg_ctx *c, const double *mom, i64 n, double *out, double *scratch) {
    int nb = (int)((n + 255) / 256 < kMeasureBlocks ? (n + 255) / 256 : kMeasureBlocks);
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_measure_mom, dim3(nb), dim3(256), 0, c->stream, mom, n, scratch);
    CG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_measure_final, dim3(1), dim3(64), 0, c->stream, scratch, nb, out);
    CG_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Tile sort.  Key = the mesh tile that holds the particle's lower CIC cell
// (the reference's own index map, so the tile of a particle is consistent
// with the cells the deposit / gather kernels touch).  Counting sort:
// histogram -> exclusive scan (hipCUB) -> scatter.  Order inside a tile is
// arbitrary (like the reference's tile_sort, particle order is not part of
// the contract; ids travel with the particles).
//
// Device-scope atomics execute memory-side on MI355X and are the scarce
// resource here, so both passes aggregate per wavefront: lanes holding a run
// of equal keys (the common case, the array is nearly sorted from the
// previous step) elect the run's first lane to issue ONE atomic for the run.
// ---------------------------------------------------------------------------
// Position of particle p, optionally drifted on the fly (A11, same arithmetic as
// k_drift): the fused drift + sort reads the undrifted arrays twice instead of writing
// the drifted positions in between.
template <bool DRIFT>
__device__ __forceinline__ void load_pos(const double *__restrict__ pos,
                                         const double *__restrict__ mom, i64 p, double dtm,
                                         double L, double &x, double &y, double &z) {
    x = pos[3 * p];
    y = pos[3 * p + 1];
    z = pos[3 * p + 2];
    if (DRIFT) {
        x = ref_mod(x + mom[3 * p] * dtm, L);
        y = ref_mod(y + mom[3 * p + 1] * dtm, L);
        z = ref_mod(z + mom[3 * p + 2] * dtm, L);
    }
}

template <bool DRIFT>
__global__ __launch_bounds__(256) void k_tile_histogram(const double *__restrict__ pos,
                                                        const double *__restrict__ mom, i64 n,
                                                        double dtm, double L, CicGeom geo, int g,
                                                        i64 N, TileGeom t, i64 x0,
                                                        unsigned *__restrict__ count) {
    i64 stride = (i64)gridDim.x * blockDim.x;
    int lane = threadIdx.x & 63;
    for (i64 base = (i64)blockIdx.x * blockDim.x; base < n; base += stride) {
        i64 p = base + threadIdx.x;
        unsigned key = kNoTile;
        if (p < n) {
            double x, y, z;
            load_pos<DRIFT>(pos, mom, p, dtm, L, x, y, z);
            key = tile_of(x, y, z, geo, g, N, t, x0);
        }
        int rs, rl;
        wave_runs(key, lane, rs, rl);
        if (lane == rs && key != kNoTile) atomicAdd(&count[key], (unsigned)rl);
    }
}

// Write the 3*rl doubles of a run of rl consecutive records cooperatively: lane r of the
// run stores doubles r, r + rl, r + 2 rl of the run's output block, fetched from the
// owning lanes with shuffles — every store instruction covers a contiguous range instead
// of every third double (measured: 21 GB -> 13 GB of HBM writes at 2^28 particles).
__device__ __forceinline__ void store_run(double *__restrict__ out, i64 first, int rs, int rl,
                                          int lane, bool valid, double a, double b, double c) {
    int r = lane - rs;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int d = r + k * rl;
        int src = rs + d / 3, comp = d - 3 * (d / 3);
        src = src > 63 ? 63 : src;  // lanes of an invalid run compute garbage, never store
        double va = __shfl(a, src), vb = __shfl(b, src), vc = __shfl(c, src);
        double v = comp == 0 ? va : (comp == 1 ? vb : vc);
        if (valid) out[3 * first + d] = v;
    }
}

template <bool DRIFT>
__global__ __launch_bounds__(256) void k_tile_scatter(
    const double *__restrict__ pos, const double *__restrict__ mom, const i64 *__restrict__ ids,
    double *__restrict__ pos_out, double *__restrict__ mom_out, i64 *__restrict__ ids_out, i64 n,
    double dtm, double L, CicGeom geo, int g, i64 N, TileGeom t, i64 x0,
    const unsigned *__restrict__ offset, unsigned *__restrict__ cursor,
    unsigned *__restrict__ err_flags) {
    i64 stride = (i64)gridDim.x * blockDim.x;
    int lane = threadIdx.x & 63;
    for (i64 base = (i64)blockIdx.x * blockDim.x; base < n; base += stride) {
        i64 p = base + threadIdx.x;
        unsigned key = kNoTile;
        double x = 0, y = 0, z = 0, mx = 0, my = 0, mz = 0;
        if (p < n) {
            load_pos<DRIFT>(pos, mom, p, dtm, L, x, y, z);
            key = tile_of(x, y, z, geo, g, N, t, x0);
            mx = mom[3 * p];
            my = mom[3 * p + 1];
            mz = mom[3 * p + 2];
        }
        int rs, rl;
        wave_runs(key, lane, rs, rl);
        unsigned first = 0;
        if (lane == rs && key != kNoTile) {
            // A run that does not fit its bucket means the histogram was not made from these
            // positions and momenta (a prepared histogram gone stale: something changed mom
            // between cg_gather_kick_tiled_prepare and this sort).  Equal totals with unequal
            // buckets always overflow one of them, so this check is complete; the run is
            // dropped (never stored outside its bucket) and the context's error word is set
            // (cg_error_flags).
            const unsigned o0 = offset[key], room = offset[key + 1] - o0;
            const unsigned local = atomicAdd(&cursor[key], (unsigned)rl);
            first = o0 + local;
            if (local + (unsigned)rl > room) {
                atomicOr(err_flags, (unsigned)CG_ERR_STALE_HISTOGRAM);
                first = kNoTile;
            }
        }
        first = __shfl(first, rs);
        // particles of other domains (kNoTile) are dropped: the host exchanges them before
        // sorting; table[last] = number kept
        const bool valid = p < n && key != kNoTile && first != kNoTile;
        store_run(pos_out, (i64)first, rs, rl, lane, valid, x, y, z);
        store_run(mom_out, (i64)first, rs, rl, lane, valid, mx, my, mz);
        if (valid && ids) ids_out[(i64)first + (lane - rs)] = ids[p];
    }
}

// Owner domain of each particle AFTER the drift pos + mom*dtm (the arithmetic of
// load_pos<true>, i.e. of k_tile_scatter<true>): lets the x-slab path ship the particles that
// are about to leave before the fused drift + sort instead of drifting in a pass of its own.
__global__ void k_owner_rank_drifted(const double *__restrict__ pos,
                                     const double *__restrict__ mom, i64 n, double dtm, double L,
                                     CicGeom geo, int g, int N, int nxl, int *__restrict__ owner) {
    i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double x, y, z;
    load_pos<true>(pos, mom, p, dtm, L, x, y, z);
    owner[p] = lower_cell(x, geo.off[0], geo.scale, g, N) / nxl;
}
int cgk_owner_rank_drifted(cg_ctx *c, const double *pos, const double *mom, i64 n, double dtm,
                           int *owner) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_owner_rank_drifted, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       c->stream, pos, mom, n, dtm, c->p.boxsize, c->geom_deposit, c->p.nghosts,
                       (int)c->N, (int)c->xmap.nxl, owner);
    CG_LAUNCH_CHECK();
    return 0;
}

// Add the tile keys (after the drift) of `n` more particles to a prepared histogram and bind it
// to new arrays: the x-slab path after its particle exchange (immigrants were not counted by
// the previous gather-kick; emigrants never were, tile_of() gives them no key).
int cgk_prepare_rebind(cg_ctx *c, const double *pos, const double *mom, i64 n_total,
                       const double *add_pos, const double *add_mom, i64 n_add) {
    if (!c->prep_valid) return 0;
    if (n_add > 0) {
        i64 blocks = (n_add + 255) / 256;
        hipLaunchKernelGGL(k_tile_histogram<true>, dim3((unsigned)blocks), dim3(256), 0, c->stream,
                           add_pos, add_mom, n_add, c->prep_dtm, c->p.boxsize, c->geom_deposit,
                           c->p.nghosts, c->N, c->tiles, c->xmap.x0, c->tile_count);
        CG_LAUNCH_CHECK();
    }
    c->prep_pos = pos;
    c->prep_mom = mom;
    c->prep_n = n_total;
    return 0;
}

int cgk_sort(cg_ctx *c, const double *pos_in, const double *mom_in, const i64 *ids_in,
             double *pos_out, double *mom_out, i64 *ids_out, i64 n, unsigned *tile_offset_out,
             int drift, double dt_over_mass, int use_prepared) {
    i64 nt = 8 * c->ntiles;  // table entries (8 buckets per tile)
    if (!use_prepared) CG_HIP(hipMemsetAsync(c->tile_count, 0, 4 * (nt + 1), c->stream));
    CG_HIP(hipMemsetAsync(c->tile_cursor, 0, 4 * (nt + 1), c->stream));
    c->prep_valid = false;
    // One 256-lane workgroup per 256 particles, no grid-stride loop: with the grid capped at
    // 4096 workgroups the scatter took 5.9 ms at 2^28 particles, uncapped 5.1 ms (the same
    // holds for a plain copy kernel, tools/copy_probe.cpp: 4.9 vs 5.6 TB/s).
    const i64 blocks = (n + 255) / 256;
    if (n > 0 && !use_prepared) {
        if (drift)
            hipLaunchKernelGGL(k_tile_histogram<true>, dim3((unsigned)blocks), dim3(256), 0,
                               c->stream, pos_in, mom_in, n, dt_over_mass, c->p.boxsize,
                               c->geom_deposit, c->p.nghosts, c->N, c->tiles, c->xmap.x0,
                               c->tile_count);
        else
            hipLaunchKernelGGL(k_tile_histogram<false>, dim3((unsigned)blocks), dim3(256), 0,
                               c->stream, pos_in, mom_in, n, 0.0, c->p.boxsize, c->geom_deposit,
                               c->p.nghosts, c->N, c->tiles, c->xmap.x0, c->tile_count);
        CG_LAUNCH_CHECK();
    }
    size_t need = 0;
    CG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, c->tile_count, tile_offset_out,
                                            (int)(nt + 1), c->stream));
    if (need > c->scan_tmp_bytes) {
        CG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->scan_tmp);
        c->scan_tmp = nullptr;
        CG_HIP(hipMalloc(&c->scan_tmp, need));
        c->scan_tmp_bytes = need;
    }
    CG_HIP(hipcub::DeviceScan::ExclusiveSum(c->scan_tmp, need, c->tile_count, tile_offset_out,
                                            (int)(nt + 1), c->stream));
    if (n > 0) {
        if (drift)
            hipLaunchKernelGGL(k_tile_scatter<true>, dim3((unsigned)blocks), dim3(256), 0,
                               c->stream, pos_in, mom_in, ids_in, pos_out, mom_out, ids_out, n,
                               dt_over_mass, c->p.boxsize, c->geom_deposit, c->p.nghosts, c->N,
                               c->tiles, c->xmap.x0, tile_offset_out, c->tile_cursor,
                               c->err_flags);
        else
            hipLaunchKernelGGL(k_tile_scatter<false>, dim3((unsigned)blocks), dim3(256), 0,
                               c->stream, pos_in, mom_in, ids_in, pos_out, mom_out, ids_out, n,
                               0.0, c->p.boxsize, c->geom_deposit, c->p.nghosts, c->N, c->tiles,
                               c->xmap.x0, tile_offset_out, c->tile_cursor, c->err_flags);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

// Regions with gaps for the next tile order: every (tile, bucket) gets room for its present
// population plus a quarter plus 32 (the populations change by a few per cent per step: the
// drift moves a particle a fraction of a cell); start_out = exclusive sum of the capacities.
__global__ void k_region_caps(const unsigned *__restrict__ start_in,
                              const unsigned *__restrict__ count_in, i64 nb,
                              unsigned *__restrict__ cap) {
    i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nb) return;
    unsigned cnt = 0;
    if (k < nb) cnt = count_in ? count_in[k] : start_in[k + 1] - start_in[k];
    cap[k] = k < nb ? cnt + (cnt >> 2) + 32u : 0u;
}

// cg_predict_regions: capacities -> exclusive sum
int cgk_predict_regions(cg_ctx *c, const unsigned *start_in, const unsigned *count_in,
                        unsigned *start_out) {
    const i64 nb = 8 * c->ntiles;
    hipLaunchKernelGGL(k_region_caps, dim3((unsigned)((nb + 256) / 256)), dim3(256), 0, c->stream,
                       start_in, count_in, nb, c->tile_cursor);
    CG_LAUNCH_CHECK();
    size_t need = 0;
    CG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, c->tile_cursor, start_out,
                                            (int)(nb + 1), c->stream));
    if (need > c->scan_tmp_bytes) {
        CG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->scan_tmp);
        c->scan_tmp = nullptr;
        CG_HIP(hipMalloc(&c->scan_tmp, need));
        c->scan_tmp_bytes = need;
    }
    CG_HIP(hipcub::DeviceScan::ExclusiveSum(c->scan_tmp, need, c->tile_cursor, start_out,
                                            (int)(nb + 1), c->stream));
    return 0;
}

// ---------------------------------------------------------------------------
// x-slab domains with the fused kick + drift + scatter: the leavers sit in a row buffer,
// already drifted.  k_rows_dest: owner domain of every row (the x-slab holding its lower CIC
// cell) and the number bound for each domain; k_region_insert: immigrants take their places in
// the regions of their (tile, bucket) through the same cursors the scatter used.
// ---------------------------------------------------------------------------
__global__ void k_rows_dest(const double *__restrict__ rows, const unsigned *__restrict__ count,
                            i64 cap, CicGeom geo, int g, int N, int nxl, int *__restrict__ dest,
                            int *__restrict__ send_counts) {
    i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 n = (i64)*count < cap ? (i64)*count : cap;
    if (t >= n) return;
    int d = lower_cell(rows[8 * t], geo.off[0], geo.scale, g, N) / nxl;
    dest[t] = d;
    atomicAdd(&send_counts[d], 1);
}
int cgk_emigrant_rows_dest(cg_ctx *c, const double *rows, const unsigned *count, i64 cap,
                           int *dest, int *send_counts) {
    CG_HIP(hipMemsetAsync(send_counts, 0, sizeof(int) * c->p.nprocs, c->stream));
    if (cap == 0) return 0;
    hipLaunchKernelGGL(k_rows_dest, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, c->stream,
                       rows, count, cap, c->geom_deposit, c->p.nghosts, (int)c->N,
                       (int)c->xmap.nxl, dest, send_counts);
    CG_LAUNCH_CHECK();
    return 0;
}

__global__ void k_region_insert(const double *__restrict__ rows, i64 m, CicGeom geo, int g, i64 N,
                                TileGeom t, i64 x0, const unsigned *__restrict__ start,
                                unsigned *__restrict__ count, double *__restrict__ pos_out,
                                double *__restrict__ mom_out, i64 *__restrict__ ids_out,
                                i64 *__restrict__ aux_out, i64 capacity,
                                unsigned *__restrict__ err_flags) {
    i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const double *row = rows + 8 * r;
    unsigned key = tile_of(row[0], row[1], row[2], geo, g, N, t, x0);
    if (key == kNoTile) {  // a row that does not belong here: the sender's owner map differs
        atomicOr(err_flags, (unsigned)CG_ERR_BUCKET_OVERFLOW);
        return;
    }
    const unsigned o0 = start[key], o1 = start[key + 1],
                   room = (i64)o1 <= capacity ? o1 - o0 : 0u;
    const unsigned local = atomicAdd(&count[key], 1u);
    if (local >= room) {
        atomicOr(err_flags, (unsigned)CG_ERR_BUCKET_OVERFLOW);
        return;
    }
    const i64 q = (i64)o0 + local;
    pos_out[3 * q] = row[0], pos_out[3 * q + 1] = row[1], pos_out[3 * q + 2] = row[2];
    mom_out[3 * q] = row[3], mom_out[3 * q + 1] = row[4], mom_out[3 * q + 2] = row[5];
    if (ids_out) ids_out[q] = __double_as_longlong(row[6]);
    if (aux_out) aux_out[q] = __double_as_longlong(row[7]);
}
int cgk_region_insert(cg_ctx *c, const double *rows, i64 m, const unsigned *start,
                      unsigned *count, double *pos_out, double *mom_out, i64 *ids_out,
                      i64 *aux_out, i64 capacity) {
    if (m == 0) return 0;
    hipLaunchKernelGGL(k_region_insert, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                       rows, m, c->geom_deposit, c->p.nghosts, c->N, c->tiles, c->xmap.x0, start,
                       count, pos_out, mom_out, ids_out, aux_out, capacity, c->err_flags);
    CG_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// The columns that follow a sort (Δmom, identifiers, the populated order, the two rung arrays of a
// Component: species.py:955-996 keeps them in step with pos and mom through every reordering):
// dst[c][q] = src[c][perm[q]] for all columns in one pass — the permutation is read once per row,
// the stores are contiguous per column.  (One torch.index_select per column: five launches that
// read the permutation five times, 5.7 of the 8.0 ms of BASELINE configs[4]'s drift + sort.)
// ---------------------------------------------------------------------------
constexpr int kPermuteCols = 8;
struct PermuteCols {
    const char *src[kPermuteCols];
    char *dst[kPermuteCols];
    int bytes[kPermuteCols];  // of a row
    int n;
};
__global__ __launch_bounds__(256) void k_permute_rows(const i64 *__restrict__ perm, i64 n,
                                                      PermuteCols C) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const i64 p = perm[q];
#pragma unroll
    for (int c = 0; c < kPermuteCols; c++) {
        if (c >= C.n) break;
        const int b = C.bytes[c];
        const char *s = C.src[c] + p * b;
        char *d = C.dst[c] + q * b;
        if (b == 24) {  // three doubles
            const double *s8 = (const double *)s;
            double *d8 = (double *)d;
            const double v0 = s8[0], v1 = s8[1], v2 = s8[2];
            d8[0] = v0, d8[1] = v1, d8[2] = v2;
        } else if (b == 8) {
            *(unsigned long long *)d = *(const unsigned long long *)s;
        } else if (b == 4) {
            *(unsigned *)d = *(const unsigned *)s;
        } else if (b == 1) {
            *d = *s;
        } else if (b % 8 == 0) {
            for (int k = 0; k < b; k += 8)
                *(unsigned long long *)(d + k) = *(const unsigned long long *)(s + k);
        } else {
            for (int k = 0; k < b; k++) d[k] = s[k];
        }
    }
}
int cgk_permute_rows(cg_ctx *c, const i64 *perm, i64 n, int ncols, const void *const *src,
                     void *const *dst, const int *row_bytes) {
    if (n == 0 || ncols == 0) return 0;
    for (int c0 = 0; c0 < ncols; c0 += kPermuteCols) {
        PermuteCols C{};
        C.n = ncols - c0 < kPermuteCols ? ncols - c0 : kPermuteCols;
        for (int k = 0; k < C.n; k++) {
            C.src[k] = (const char *)src[c0 + k];
            C.dst[k] = (char *)dst[c0 + k];
            C.bytes[k] = row_bytes[c0 + k];
        }
        hipLaunchKernelGGL(k_permute_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           c->stream, perm, n, C);
        CG_LAUNCH_CHECK();
    }
    return 0;
}
