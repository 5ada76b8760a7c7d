// cg_tiled_kernels.hip — LDS-tiled CIC deposit (A1/A2) and fused finite
// difference + CIC gather + kick (A9/A10) for particles kept in mesh-tile
// order (cg_sort_particles).  One 256-lane workgroup per T^3-cell tile.
//
// Why tiles: on MI355X device-scope FP64 atomics execute memory-side, one
// fabric transaction per lane (measured: 2^31 direct atomic adds = 77 ms at
// 2^28 particles).  Accumulating a tile in LDS (ds_add_f64) and writing it
// out once turns the scatter into a streaming store of the mesh: tile
// interiors are exclusively owned by one workgroup and are ASSIGNED with
// plain stores (no zero-fill pass), only the tile faces — the cells that also
// receive the neighbouring tiles' halo layer — are summed with atomics.
//
// Arithmetic per particle and per cell is the reference's (see
// cg_mesh_kernels.hip); only the order in which particles are added to a
// cell differs.  Compiled with -ffp-contract=off.
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

struct Cic1 {
    i64 index;
    double w0, w1;
};
__device__ __forceinline__ Cic1 cic1(double pos, double off, double scale) {
    double x = (pos - off) * scale;
    Cic1 r;
    r.index = (i64)x;
    double dist = x - (double)r.index;
    r.w0 = 1 - dist;
    r.w1 = dist;
    return r;
}
__device__ __forceinline__ i64 wrap(i64 a, i64 n) {
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

// ---------------------------------------------------------------------------
// zero the tile faces (planes i, j or k = 0 mod T) ahead of an assigning deposit
// ---------------------------------------------------------------------------
__global__ void k_zero_tile_faces(double *__restrict__ mesh, i64 N, i64 pad, int T) {
    i64 row = blockIdx.x;
    i64 i = row / N, j = row - i * N;
    double *r = mesh + row * pad;
    if (i % T == 0 || j % T == 0) {
        for (i64 k = threadIdx.x; k < N; k += blockDim.x) r[k] = 0;
    } else {
        for (i64 k = (i64)threadIdx.x * T; k < N; k += (i64)blockDim.x * T) r[k] = 0;
    }
}

// ---------------------------------------------------------------------------
// tiled deposit
// ---------------------------------------------------------------------------
template <int T, bool ACCUMULATE>
__global__ __launch_bounds__(256) void k_deposit_cic_tiled(
    const double *__restrict__ pos, const unsigned *__restrict__ tile_offset,
    double *__restrict__ mesh, i64 N, i64 pad, int g, int nt, unsigned ntiles, CicGeom geo,
    double contribution) {
    constexpr int E = T + 1;
    constexpr int NL = E * E * E;
    __shared__ double lds[NL];
    const unsigned tile = tile_for_block(blockIdx.x, ntiles);
    const int tc = tile % nt, tb = (tile / nt) % nt, ta = tile / (nt * nt);
    const i64 T0a = (i64)ta * T, T0b = (i64)tb * T, T0c = (i64)tc * T;
    for (int idx = threadIdx.x; idx < NL; idx += 256) lds[idx] = 0;
    __syncthreads();
    const i64 beg = tile_offset[tile], end = tile_offset[tile + 1];
    for (i64 p = beg + threadIdx.x; p < end; p += 256) {
        Cic1 cx = cic1(pos[3 * p + 0], geo.off[0], geo.scale);
        Cic1 cy = cic1(pos[3 * p + 1], geo.off[1], geo.scale);
        Cic1 cz = cic1(pos[3 * p + 2], geo.off[2], geo.scale);
        // mesh.py:5142-5155: ((w_x*contribution)*w_y)*w_z
        double wi0 = cx.w0 * contribution, wi1 = cx.w1 * contribution;
        double w00 = wi0 * cy.w0, w01 = wi0 * cy.w1, w10 = wi1 * cy.w0, w11 = wi1 * cy.w1;
        i64 ga = wrap(cx.index - g, N), gb = wrap(cy.index - g, N), gc = wrap(cz.index - g, N);
        i64 la = ga - T0a, lb = gb - T0b, lc = gc - T0c;
        if (la >= 0 && la < T && lb >= 0 && lb < T && lc >= 0 && lc < T) {
            double *l = lds + (la * E + lb) * E + lc;
            atomicAdd(l, w00 * cz.w0);
            atomicAdd(l + 1, w00 * cz.w1);
            atomicAdd(l + E, w01 * cz.w0);
            atomicAdd(l + E + 1, w01 * cz.w1);
            atomicAdd(l + E * E, w10 * cz.w0);
            atomicAdd(l + E * E + 1, w10 * cz.w1);
            atomicAdd(l + E * E + E, w11 * cz.w0);
            atomicAdd(l + E * E + E + 1, w11 * cz.w1);
        } else if (ACCUMULATE) {
            // a particle outside its tile (array drifted since the sort): still correct
            // when every cell is summed atomically
            i64 a1 = wrap(ga + 1, N), b1 = wrap(gb + 1, N), c1 = wrap(gc + 1, N);
            double *r00 = mesh + (ga * N + gb) * pad, *r01 = mesh + (ga * N + b1) * pad;
            double *r10 = mesh + (a1 * N + gb) * pad, *r11 = mesh + (a1 * N + b1) * pad;
            unsafeAtomicAdd(r00 + gc, w00 * cz.w0);
            unsafeAtomicAdd(r00 + c1, w00 * cz.w1);
            unsafeAtomicAdd(r01 + gc, w01 * cz.w0);
            unsafeAtomicAdd(r01 + c1, w01 * cz.w1);
            unsafeAtomicAdd(r10 + gc, w10 * cz.w0);
            unsafeAtomicAdd(r10 + c1, w10 * cz.w1);
            unsafeAtomicAdd(r11 + gc, w11 * cz.w0);
            unsafeAtomicAdd(r11 + c1, w11 * cz.w1);
        }
        // (assign mode requires exact tile order: cg_sort_particles on this array)
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NL; idx += 256) {
        int c = idx % E, b = (idx / E) % E, a = idx / (E * E);
        double v = lds[idx];
        i64 gi = T0a + a, gj = T0b + b, gk = T0c + c;
        gi = gi >= N ? gi - N : gi;
        gj = gj >= N ? gj - N : gj;
        gk = gk >= N ? gk - N : gk;
        double *dst = mesh + (gi * N + gj) * pad + gk;
        bool face = (a == 0) | (a == T) | (b == 0) | (b == T) | (c == 0) | (c == T);
        if (ACCUMULATE || face) {
            if (v != 0) unsafeAtomicAdd(dst, v);
        } else {
            *dst = v;
        }
    }
}

template <int T>
static int launch_deposit(cg_ctx *c, const double *pos, const unsigned *tile_offset,
                          double contribution, int accumulate) {
    unsigned nt = (unsigned)c->ntiles;
    if (accumulate)
        hipLaunchKernelGGL((k_deposit_cic_tiled<T, true>), dim3(nt), dim3(256), 0, c->stream, pos,
                           tile_offset, c->mesh, c->N, c->pad, c->p.nghosts, c->tiles.ntx, nt,
                           c->geom_deposit, contribution);
    else
        hipLaunchKernelGGL((k_deposit_cic_tiled<T, false>), dim3(nt), dim3(256), 0, c->stream, pos,
                           tile_offset, c->mesh, c->N, c->pad, c->p.nghosts, c->tiles.ntx, nt,
                           c->geom_deposit, contribution);
    return 0;
}

int cgk_deposit_cic_tiled(cg_ctx *c, const double *pos, i64 n, const unsigned *tile_offset,
                          double contribution, int accumulate) {
    (void)n;
    const int T = c->tiles.tx;
    if (!accumulate) {
        hipLaunchKernelGGL(k_zero_tile_faces, dim3((unsigned)(c->N * c->N)), dim3(64), 0,
                           c->stream, c->mesh, c->N, c->pad, T);
        CG_LAUNCH_CHECK();
    }
    switch (T) {
        case 16: launch_deposit<16>(c, pos, tile_offset, contribution, accumulate); break;
        case 8: launch_deposit<8>(c, pos, tile_offset, contribution, accumulate); break;
        case 4: launch_deposit<4>(c, pos, tile_offset, contribution, accumulate); break;
        case 2: launch_deposit<2>(c, pos, tile_offset, contribution, accumulate); break;
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

template <int ORDER, int T>
__global__ __launch_bounds__(256) void k_gather_kick_tiled(
    const double *__restrict__ pos, double *__restrict__ mom,
    const unsigned *__restrict__ tile_offset, const double *__restrict__ mesh, i64 N, i64 pad,
    int g, int nt, unsigned ntiles, CicGeom geo, double c1, double c2, double factor) {
    constexpr int H = ORDER / 2;
    constexpr int E = T + 1 + 2 * H;  // cells [T0-H, T0+T+H]
    constexpr int NL = E * E * E;
    extern __shared__ double lds[];
    const unsigned tile = tile_for_block(blockIdx.x, ntiles);
    const i64 beg = tile_offset[tile], end = tile_offset[tile + 1];
    if (beg == end) return;  // uniform for the workgroup
    const int tc = tile % nt, tb = (tile / nt) % nt, ta = tile / (nt * nt);
    const i64 T0a = (i64)ta * T, T0b = (i64)tb * T, T0c = (i64)tc * T;
    for (int idx = threadIdx.x; idx < NL; idx += 256) {
        int c = idx % E, b = (idx / E) % E, a = idx / (E * E);
        i64 gi = wrap(T0a - H + a, N), gj = wrap(T0b - H + b, N), gk = wrap(T0c - H + c, N);
        lds[idx] = mesh[(gi * N + gj) * pad + gk];
    }
    __syncthreads();
    for (i64 p = beg + threadIdx.x; p < end; p += 256) {
        Cic1 cx = cic1(pos[3 * p + 0], geo.off[0], geo.scale);
        Cic1 cy = cic1(pos[3 * p + 1], geo.off[1], geo.scale);
        Cic1 cz = cic1(pos[3 * p + 2], geo.off[2], geo.scale);
        i64 ga = wrap(cx.index - g, N), gb = wrap(cy.index - g, N), gc = wrap(cz.index - g, N);
        i64 la = ga - T0a, lb = gb - T0b, lc = gc - T0c;
        double wx[2] = {cx.w0, cx.w1}, wy[2] = {cy.w0, cy.w1}, wz[2] = {cz.w0, cz.w1};
        double val[3] = {0, 0, 0};
        if (la >= 0 && la < T && lb >= 0 && lb < T && lc >= 0 && lc < T) {
            const double *base = lds + ((la + H) * E + (lb + H)) * E + (lc + H);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    double wij = wx[i] * wy[j];
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const double *cell = base + (i * E + j) * E + k;
                        auto phi = [&](int da, int db, int dc) {
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
        } else {
            // outside its tile (array drifted since the sort): read the mesh directly
            constexpr int W = 2 + 2 * H;
            i64 ix[W], iy[W], iz[W];
#pragma unroll
            for (int s = 0; s < W; s++) {
                ix[s] = wrap(ga - H + s, N) * N * pad;
                iy[s] = wrap(gb - H + s, N) * pad;
                iz[s] = wrap(gc - H + s, N);
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    double wij = wx[i] * wy[j];
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        auto phi = [&](int da, int db, int dc) {
                            return mesh[ix[H + i + da] + iy[H + j + db] + iz[H + k + dc]];
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
        mom[3 * p + 0] += val[0];
        mom[3 * p + 1] += val[1];
        mom[3 * p + 2] += val[2];
    }
}

template <int ORDER, int T>
static int launch_gather(cg_ctx *c, const double *pos, double *mom, const unsigned *tile_offset,
                         double c1, double c2, double factor) {
    constexpr int E = T + 1 + 2 * (ORDER / 2);
    size_t lds = sizeof(double) * E * E * E;
    auto kern = k_gather_kick_tiled<ORDER, T>;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        CG_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        attr_set = true;
    }
    unsigned nt = (unsigned)c->ntiles;
    hipLaunchKernelGGL(kern, dim3(nt), dim3(256), lds, c->stream, pos, mom, tile_offset, c->mesh,
                       c->N, c->pad, c->p.nghosts, c->tiles.ntx, nt, c->geom_gather, c1, c2,
                       factor);
    return 0;
}

int cgk_gather_kick_tiled(cg_ctx *c, const double *pos, double *mom, i64 n,
                          const unsigned *tile_offset, int diff_order, double factor) {
    (void)n;
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
        rc = diff_order == 2 ? launch_gather<2, TT>(c, pos, mom, tile_offset, c1, c2, factor)    \
                             : launch_gather<4, TT>(c, pos, mom, tile_offset, c1, c2, factor);   \
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
    return 0;
}
