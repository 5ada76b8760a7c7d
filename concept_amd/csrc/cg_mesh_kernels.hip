// cg_mesh_kernels.hip — hand-written gfx950 kernels of the PM mesh path:
// CIC deposit (A1/A2), k-space Poisson kernel (A5/A6), CIC gather + finite
// difference + kick (A9/A10).  FP64 throughout; compiled with
// -ffp-contract=off so every expression is evaluated in the reference's
// operation order (no FMA contraction).
#include "cg_internal.h"
#include "cg_kspace.h"

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
// CIC index + weights, set_weights_CIC (mesh.py:5319-5324) under the
// coordinate map of mesh.py:1604-1606 / :430-432.
// ---------------------------------------------------------------------------
struct Cic1 {
    i64 index;  // reference (ghosted) index of the lower cell
    double w0, w1;
};
__device__ __forceinline__ Cic1 cic1(double pos, double off, double scale) {
    double x = (pos - off) * scale;
    Cic1 r;
    r.index = (i64)x;  // int(x); x > 0 always (mesh.py:1602-1603)
    double dist = x - (double)r.index;
    r.w0 = 1 - dist;
    r.w1 = dist;
    return r;
}
__device__ __forceinline__ i64 wrap(i64 a, i64 n) {
    // a in [-n, 2n)
    a = a < 0 ? a + n : a;
    return a >= n ? a - n : a;
}

// ---------------------------------------------------------------------------
// A1/A2 deposit, direct form (cg_deposit_cic: particles in any order).  Index and weight
// arithmetic is the reference's; only the summation order across particles differs.
// Made for particles whose memory order is compact in space but is not THIS mesh's tile
// order — BASELINE configs[4]: 512^3 particles, sorted by the tiles of their own 1024^3
// P3M mesh, deposited onto the 256^3 PM mesh of the (fluid <- particles, fluid) interaction
// (interactions.py:1985-2335; param/example_nonlinnu:36-45).  Device-scope FP64 atomics execute
// memory-side on MI355X (8 per particle: 31.7 ms for 2^27 particles).  Here a workgroup takes
// 2048 consecutive particles, finds the box of their lower cells (relative to the first one's:
// the periodic seam is no seam), and — when the box with its CIC overhang fits 4096 cells of LDS —
// accumulates there (ds_add_f64) and adds the box to the mesh once: one device atomic per
// touched cell instead of eight per particle.  A chunk that is spread out (unsorted particles)
// takes the direct atomics: 8 device-scope FP64 atomic adds per particle.
// ---------------------------------------------------------------------------
constexpr int kDcPer = 4, kDcLanes = 512, kDcCells = 4096;
__global__ __launch_bounds__(kDcLanes) void k_deposit_cic_chunks(
    const double *__restrict__ pos, i64 n, double *__restrict__ mesh, i64 N, i64 ny, i64 pad,
    int g, XMap xm, CicGeom geo, double contribution) {
    __shared__ double blk[kDcCells];
    __shared__ int s_ref[3], s_lo[3], s_hi[3];
    const int tid = threadIdx.x;
    const i64 base = (i64)blockIdx.x * (kDcLanes * kDcPer);
    const int Ni = (int)N;
    double px[kDcPer], py[kDcPer], pz[kDcPer];
    bool valid[kDcPer];
#pragma unroll
    for (int u = 0; u < kDcPer; u++) {
        const i64 p = base + tid + (i64)kDcLanes * u;
        valid[u] = p < n;
        px[u] = py[u] = pz[u] = 0;
        if (valid[u]) px[u] = pos[3 * p], py[u] = pos[3 * p + 1], pz[u] = pos[3 * p + 2];
    }
    auto cells = [&](int u, int (&cell)[3]) {
        cell[0] = (int)wrap(cic1(px[u], geo.off[0], geo.scale).index - g, N);
        cell[1] = (int)wrap(cic1(py[u], geo.off[1], geo.scale).index - g, N);
        cell[2] = (int)wrap(cic1(pz[u], geo.off[2], geo.scale).index - g, N);
    };
    if (tid == 0) {  // (the chunk's first particle exists: base < n)
        int c0[3];
        cells(0, c0);
        for (int d = 0; d < 3; d++) {
            s_ref[d] = c0[d];
            s_lo[d] = 0;
            s_hi[d] = 0;
        }
    }
    __syncthreads();
    const int r0 = s_ref[0], r1 = s_ref[1], r2 = s_ref[2];
    // lower cells relative to the first particle's, through the periodic seam: in [-N/2, N/2)
    auto rel = [&](int cell, int ref) {
        int d = cell - ref;
        d += d < -(Ni / 2) ? Ni : 0;
        d -= d >= Ni - Ni / 2 ? Ni : 0;
        return d;
    };
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
#pragma unroll
    for (int u = 0; u < kDcPer; u++) {
        if (!valid[u]) continue;
        int cell[3];
        cells(u, cell);
        const int d[3] = {rel(cell[0], r0), rel(cell[1], r1), rel(cell[2], r2)};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            lo[k] = min(lo[k], d[k]);
            hi[k] = max(hi[k], d[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            lo[k] = min(lo[k], __shfl_xor(lo[k], m));
            hi[k] = max(hi[k], __shfl_xor(hi[k], m));
        }
        if ((tid & 63) == 0) {
            atomicMin(&s_lo[k], lo[k]);
            atomicMax(&s_hi[k], hi[k]);
        }
    }
    __syncthreads();
    const int l0 = s_lo[0], l1 = s_lo[1], l2 = s_lo[2];
    const int e0 = s_hi[0] - l0 + 2, e1 = s_hi[1] - l1 + 2, e2 = s_hi[2] - l2 + 2;
    const bool fits = (i64)e0 * e1 * e2 <= kDcCells && e0 <= Ni && e1 <= Ni && e2 <= Ni;
    const int ncells = fits ? e0 * e1 * e2 : 0;
    for (int i = tid; i < ncells; i += kDcLanes) blk[i] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kDcPer; u++) {
        if (!valid[u]) continue;
        const Cic1 cx = cic1(px[u], geo.off[0], geo.scale), cy = cic1(py[u], geo.off[1], geo.scale),
                   cz = cic1(pz[u], geo.off[2], geo.scale);
        // mesh.py:5142-5155: ((w_x*contribution)*w_y)*w_z
        const double wi0 = cx.w0 * contribution, wi1 = cx.w1 * contribution;
        const double w00 = wi0 * cy.w0, w01 = wi0 * cy.w1, w10 = wi1 * cy.w0, w11 = wi1 * cy.w1;
        if (fits) {
            const int a = rel((int)wrap(cx.index - g, N), r0) - l0,
                      b = rel((int)wrap(cy.index - g, N), r1) - l1,
                      c = rel((int)wrap(cz.index - g, N), r2) - l2;
            double *q = blk + (a * e1 + b) * e2 + c;
            unsafeAtomicAdd(q, w00 * cz.w0);
            unsafeAtomicAdd(q + 1, w00 * cz.w1);
            unsafeAtomicAdd(q + e2, w01 * cz.w0);
            unsafeAtomicAdd(q + e2 + 1, w01 * cz.w1);
            unsafeAtomicAdd(q + e1 * e2, w10 * cz.w0);
            unsafeAtomicAdd(q + e1 * e2 + 1, w10 * cz.w1);
            unsafeAtomicAdd(q + e1 * e2 + e2, w11 * cz.w0);
            unsafeAtomicAdd(q + e1 * e2 + e2 + 1, w11 * cz.w1);
        } else {
            const i64 i0 = cg_xlayer(xm, cx.index - g, N), i1 = cg_xlayer(xm, cx.index - g + 1, N);
            const i64 j0 = wrap(cy.index - g, N), j1 = wrap(cy.index - g + 1, N);
            const i64 k0 = wrap(cz.index - g, N), k1 = wrap(cz.index - g + 1, N);
            double *r00 = mesh + (i0 * ny + j0) * pad, *r01 = mesh + (i0 * ny + j1) * pad;
            double *r10 = mesh + (i1 * ny + j0) * pad, *r11 = mesh + (i1 * ny + j1) * pad;
            unsafeAtomicAdd(r00 + k0, w00 * cz.w0);
            unsafeAtomicAdd(r00 + k1, w00 * cz.w1);
            unsafeAtomicAdd(r01 + k0, w01 * cz.w0);
            unsafeAtomicAdd(r01 + k1, w01 * cz.w1);
            unsafeAtomicAdd(r10 + k0, w10 * cz.w0);
            unsafeAtomicAdd(r10 + k1, w10 * cz.w1);
            unsafeAtomicAdd(r11 + k0, w11 * cz.w0);
            unsafeAtomicAdd(r11 + k1, w11 * cz.w1);
        }
    }
    if (!fits) return;  // (uniform)
    __syncthreads();
    for (int i = tid; i < ncells; i += kDcLanes) {
        const double v = blk[i];
        if (v == 0) continue;
        const int c = i % e2, b = (i / e2) % e1, a = i / (e2 * e1);
        // the cell's index on the mesh: in [-N, 2N) before the wrap (|rel| <= N/2, e <= N)
        const i64 gi = cg_xlayer(xm, wrap((i64)r0 + l0 + a, N), N), gj = wrap((i64)r1 + l1 + b, N),
                  gk = wrap((i64)r2 + l2 + c, N);
        unsafeAtomicAdd(mesh + (gi * ny + gj) * pad + gk, v);
    }
}

int cgk_deposit_cic(cg_ctx *c, const double *pos, i64 n, double contribution) {
    if (n <= 0) return 0;
    const i64 blocks = (n + kDcLanes * kDcPer - 1) / (kDcLanes * kDcPer);
    hipLaunchKernelGGL(k_deposit_cic_chunks, dim3((unsigned)blocks), dim3(kDcLanes), 0, c->stream,
                       pos, n, c->mesh, c->N, c->ny, c->pad, c->p.nghosts, c->xmap,
                       c->geom_deposit, contribution);
    CG_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// A5/A6 k-space kernel on rocFFT's natural output layout complex[i][j][kk]
// (not transposed: on one GPU the transposition FFTW-MPI produces is only a
// storage order; every mode (ki,kj,kk) gets the reference's factor).
//   nullify_modes('nyquist')  mesh.py:3591-3622 -> planes ki,kj = -N/2, kk = N/2
//   factor                    mesh.py:2775-2856, interactions.py:2096-2116
//   nullify_modes('origin')   interactions.py:2118
// One lane per complex mode (16 B load + 16 B store, coalesced along kk).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kspace(double2 *__restrict__ four, i64 N, i64 si,
                                                i64 pitch, int j0, int nj, KspaceParams P) {
    const i64 nk = N / 2 + 1;
    i64 row = blockIdx.x;  // i*nj + j_local over the Fourier view (cg_ctx::four)
    i64 i = row / nj, jl = row - i * nj;
    double2 *r = four + i * si + jl * pitch;
    for (i64 kk = threadIdx.x; kk < nk; kk += blockDim.x) {
        double factor = kspace_factor(P, N, i, j0 + jl, kk);
        double2 v = make_double2(0, 0);
        if (factor != 0) {
            v = r[kk];
            v.x *= factor;
            v.y *= factor;
        }
        r[kk] = v;
    }
}

int cgk_kspace(cg_ctx *c, int deconv_order, double C, int long_range, double E) {
    i64 rows = c->N * c->f_nj;
    int block = c->N / 2 + 1 >= 256 ? 256 : (c->N / 2 + 1 > 64 ? 128 : 64);
    hipLaunchKernelGGL(k_kspace, dim3((unsigned)rows), dim3(block), 0, c->stream, c->four, c->N,
                       c->f_si, c->pad / 2, c->f_j0, c->f_nj,
                       KspaceParams{c->ktab_n, c->ktab_s, c->ktab_q, deconv_order, long_range, C, E});
    CG_LAUNCH_CHECK();
    return 0;
}

// complex[i][j][kk] -> the reference's transposed double[j][i][N+2] (debug fetch)
__global__ void k_transpose_fourier(const double2 *__restrict__ src, double2 *__restrict__ dst,
                                    i64 N, i64 ny, i64 pitch) {
    i64 nk = N / 2 + 1;
    i64 row = blockIdx.x;
    i64 i = row / N, j = row - i * N;
    for (i64 kk = threadIdx.x; kk < nk; kk += blockDim.x)
        dst[(j * N + i) * nk + kk] = src[(i * ny + j) * pitch + kk];
}
int cgk_transpose_fourier(cg_ctx *c, const double *src, double *dst) {
    hipLaunchKernelGGL(k_transpose_fourier, dim3((unsigned)(c->N * c->N)), dim3(64), 0, c->stream,
                       (const double2 *)src, (double2 *)dst, c->N, c->ny, c->pad / 2);
    CG_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// A9/A10 gather + finite difference + kick, direct form (cg_gather_kick: particles in any
// order).  For each dim the 8 force-cell values are formed exactly as diff_domaingrid forms
// them (mesh.py:4966-4981) and accumulated in the order of mesh.py:5142-5155 / :445, then
// value *= factor, mom += value (:456-459).
// Like the deposit above, a workgroup takes 2048 consecutive particles and, when the box of
// their cells with the stencil's reach fits 4096 cells, copies that box of the potential into
// LDS first: particles that are compact in space but not in THIS mesh's tile order (configs[4]:
// the matter particles, sorted by the tiles of their 1024^3 mesh, kicked by the fluid's
// potential on the 256^3 mesh) then read the mesh once per box instead of 48 scattered values
// per particle (5.5 -> 2.x ms for 2^27 particles).  A chunk that is spread out reads the mesh
// directly.
// ---------------------------------------------------------------------------
template <int ORDER, class Phi>
__device__ __forceinline__ void gather_force(const Phi &phi, const double (&wx)[2],
                                             const double (&wy)[2], const double (&wz)[2],
                                             double c1, double c2, double (&val)[3]) {
    constexpr int H = ORDER / 2;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const double wij = wx[i] * wy[j];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const double w = wij * wz[k];
                const int a = H + i, b = H + j, cc = H + k;
                double fx, fy, fz;
                if (ORDER == 2) {
                    fx = c1 * (phi(a + 1, b, cc) - phi(a - 1, b, cc));
                    fy = c1 * (phi(a, b + 1, cc) - phi(a, b - 1, cc));
                    fz = c1 * (phi(a, b, cc + 1) - phi(a, b, cc - 1));
                } else {
                    fx = c1 * (phi(a + 1, b, cc) - phi(a - 1, b, cc)) -
                         c2 * (phi(a + 2, b, cc) - phi(a - 2, b, cc));
                    fy = c1 * (phi(a, b + 1, cc) - phi(a, b - 1, cc)) -
                         c2 * (phi(a, b + 2, cc) - phi(a, b - 2, cc));
                    fz = c1 * (phi(a, b, cc + 1) - phi(a, b, cc - 1)) -
                         c2 * (phi(a, b, cc + 2) - phi(a, b, cc - 2));
                }
                val[0] += fx * w;
                val[1] += fy * w;
                val[2] += fz * w;
            }
        }
}

template <int ORDER>
__global__ __launch_bounds__(kDcLanes) void k_gather_kick_chunks(
    const double *__restrict__ pos, double *__restrict__ mom, i64 n,
    const double *__restrict__ mesh, i64 N, i64 ny, i64 pad, int g, XMap xm, CicGeom geo,
    double c1, double c2, double factor) {
    constexpr int H = ORDER / 2;      // stencil half width
    constexpr int W = 2 + 2 * H;      // cells needed per dimension
    __shared__ double blk[kDcCells];
    __shared__ int s_ref[3], s_lo[3], s_hi[3];
    const int tid = threadIdx.x;
    const i64 base = (i64)blockIdx.x * (kDcLanes * kDcPer);
    const int Ni = (int)N;
    double px[kDcPer], py[kDcPer], pz[kDcPer];
    bool valid[kDcPer];
#pragma unroll
    for (int u = 0; u < kDcPer; u++) {
        const i64 p = base + tid + (i64)kDcLanes * u;
        valid[u] = p < n;
        px[u] = py[u] = pz[u] = 0;
        if (valid[u]) px[u] = pos[3 * p], py[u] = pos[3 * p + 1], pz[u] = pos[3 * p + 2];
    }
    auto cells = [&](int u, int (&cell)[3]) {
        cell[0] = (int)wrap(cic1(px[u], geo.off[0], geo.scale).index - g, N);
        cell[1] = (int)wrap(cic1(py[u], geo.off[1], geo.scale).index - g, N);
        cell[2] = (int)wrap(cic1(pz[u], geo.off[2], geo.scale).index - g, N);
    };
    if (tid == 0) {
        int c0[3];
        cells(0, c0);
        for (int d = 0; d < 3; d++) {
            s_ref[d] = c0[d];
            s_lo[d] = 0;
            s_hi[d] = 0;
        }
    }
    __syncthreads();
    const int r0 = s_ref[0], r1 = s_ref[1], r2 = s_ref[2];
    auto rel = [&](int cell, int ref) {
        int d = cell - ref;
        d += d < -(Ni / 2) ? Ni : 0;
        d -= d >= Ni - Ni / 2 ? Ni : 0;
        return d;
    };
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
#pragma unroll
    for (int u = 0; u < kDcPer; u++) {
        if (!valid[u]) continue;
        int cell[3];
        cells(u, cell);
        const int d[3] = {rel(cell[0], r0), rel(cell[1], r1), rel(cell[2], r2)};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            lo[k] = min(lo[k], d[k]);
            hi[k] = max(hi[k], d[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            lo[k] = min(lo[k], __shfl_xor(lo[k], m));
            hi[k] = max(hi[k], __shfl_xor(hi[k], m));
        }
        if ((tid & 63) == 0) {
            atomicMin(&s_lo[k], lo[k]);
            atomicMax(&s_hi[k], hi[k]);
        }
    }
    __syncthreads();
    // the box: cells lo - H .. hi + 1 + H of every dimension
    const int l0 = s_lo[0] - H, l1 = s_lo[1] - H, l2 = s_lo[2] - H;
    const int e0 = s_hi[0] - s_lo[0] + W, e1 = s_hi[1] - s_lo[1] + W, e2 = s_hi[2] - s_lo[2] + W;
    const bool fits = (i64)e0 * e1 * e2 <= kDcCells && e0 <= Ni && e1 <= Ni && e2 <= Ni;
    if (fits) {
        const int ncells = e0 * e1 * e2;
        for (int i = tid; i < ncells; i += kDcLanes) {
            const int c = i % e2, b = (i / e2) % e1, a = i / (e2 * e1);
            const i64 gi = cg_xlayer(xm, wrap((i64)r0 + l0 + a, N), N),
                      gj = wrap((i64)r1 + l1 + b, N), gk = wrap((i64)r2 + l2 + c, N);
            blk[i] = mesh[(gi * ny + gj) * pad + gk];
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < kDcPer; u++) {
        const i64 p = base + tid + (i64)kDcLanes * u;
        if (p >= n) continue;
        const Cic1 cx = cic1(px[u], geo.off[0], geo.scale), cy = cic1(py[u], geo.off[1], geo.scale),
                   cz = cic1(pz[u], geo.off[2], geo.scale);
        const double wx[2] = {cx.w0, cx.w1}, wy[2] = {cy.w0, cy.w1}, wz[2] = {cz.w0, cz.w1};
        double val[3] = {0, 0, 0};
        if (fits) {
            // entry (s_a, s_b, s_c) of the particle's W^3 cells = cell index - g - H + s
            const double *q = blk +
                ((rel((int)wrap(cx.index - g, N), r0) - H - l0) * e1 +
                 (rel((int)wrap(cy.index - g, N), r1) - H - l1)) * e2 +
                (rel((int)wrap(cz.index - g, N), r2) - H - l2);
            gather_force<ORDER>([&](int a, int b, int c) { return q[(a * e1 + b) * e2 + c]; }, wx,
                                wy, wz, c1, c2, val);
        } else {
            i64 ix[W], iy[W], iz[W];
#pragma unroll
            for (int s = 0; s < W; s++) {
                ix[s] = cg_xlayer(xm, cx.index - g - H + s, N) * ny * pad;
                iy[s] = wrap(cy.index - g - H + s, N) * pad;
                iz[s] = wrap(cz.index - g - H + s, N);
            }
            gather_force<ORDER>([&](int a, int b, int c) { return mesh[ix[a] + iy[b] + iz[c]]; },
                                wx, wy, wz, c1, c2, val);
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

int cgk_gather_kick(cg_ctx *c, const double *pos, double *mom, i64 n, int diff_order,
                    double factor) {
    if (n <= 0) return 0;
    const i64 blocks = (n + kDcLanes * kDcPer - 1) / (kDcLanes * kDcPer);
    double dx = c->p.boxsize / (double)c->N;  // interactions.py:2133
    if (diff_order == 2) {
        double c1 = (1.0 / 2) / dx;  // mesh.py:4967
        hipLaunchKernelGGL(k_gather_kick_chunks<2>, dim3((unsigned)blocks), dim3(kDcLanes), 0,
                           c->stream, pos, mom, n, c->mesh, c->N, c->ny, c->pad, c->p.nghosts,
                           c->xmap, c->geom_gather, c1, 0.0, factor);
    } else {
        double c1 = (2.0 / 3) / dx, c2 = (1.0 / 12) / dx;  // mesh.py:4973-4977
        hipLaunchKernelGGL(k_gather_kick_chunks<4>, dim3((unsigned)blocks), dim3(kDcLanes), 0,
                           c->stream, pos, mom, n, c->mesh, c->N, c->ny, c->pad, c->p.nghosts,
                           c->xmap, c->geom_gather, c1, c2, factor);
    }
    CG_LAUNCH_CHECK();
    return 0;
}

// CIC indices as set_weights_CIC returns them (parity tests: bit-exact bar)
__global__ void k_cic_indices(const double *__restrict__ pos, i64 n, CicGeom geo,
                              i64 *__restrict__ idx) {
    i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    for (int d = 0; d < 3; d++) idx[3 * p + d] = cic1(pos[3 * p + d], geo.off[d], geo.scale).index;
}
int cgk_cic_indices(cg_ctx *c, const double *pos, i64 n, int for_gather, i64 *idx) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_cic_indices, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       pos, n, for_gather ? c->geom_gather : c->geom_deposit, idx);
    CG_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// x-slab domains: halo layers.  A layer is a contiguous [N][pad] block, so the
// ghost exchange (communication.py:563-660) needs no packing: the host comm
// layer sends/receives whole layers; '+=' (deposit fold) is this add kernel,
// '=' (ghost fill) a plain copy.  layer0 is relative to the first owned layer.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_layers_add(double2 *__restrict__ dst,
                                                    const double2 *__restrict__ src, i64 n2) {
    i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x; v < n2; v += stride) {
        double2 a = dst[v], b = src[v];
        a.x += b.x;
        a.y += b.y;
        dst[v] = a;
    }
}
int cgk_layers_write(cg_ctx *c, i64 layer0, i64 nlayers, const double *src, int add) {
    i64 per = c->ny * c->pad;  // whole layers, the unused row included
    double *dst = c->mesh0 + layer0 * per;
    if (!add) {
        CG_HIP(hipMemcpyAsync(dst, src, 8 * per * nlayers, hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    i64 n2 = per * nlayers / 2;
    i64 blocks = (n2 + 255) / 256;
    hipLaunchKernelGGL(k_layers_add, dim3((unsigned)blocks), dim3(256), 0, c->stream,
                       (double2 *)dst, (const double2 *)src, n2);
    CG_LAUNCH_CHECK();
    return 0;
}

// Owner domain of each particle: the x-slab that holds its lower CIC cell
// (counterpart of which_domain(), communication.py:756-772, for slab domains).
__global__ void k_owner_rank(const double *__restrict__ pos, i64 n, CicGeom geo, int g, i64 N,
                             i64 nxl, int *__restrict__ owner) {
    i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    i64 a = wrap(cic1(pos[3 * p], geo.off[0], geo.scale).index - g, N);
    owner[p] = (int)(a / nxl);
}
// Destination domain of every listed emigrant (the rows cg_gather_kick_tiled_prepare found
// leaving the slab with the prepared drift) and the number going to each domain: what
// exchange() (communication.py:135-517) needs before it can size its messages.  *count is read
// on the device, so the host enqueues this right behind the gather-kick without waiting.
__global__ void k_emigrant_dest(const double *__restrict__ pos, const double *__restrict__ mom,
                                const i64 *__restrict__ idx, const unsigned *__restrict__ count,
                                i64 cap, double dtm, double L, CicGeom geo, int g, i64 N, i64 nxl,
                                int *__restrict__ dest, int *__restrict__ send_counts) {
    i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 n = (i64)*count < cap ? (i64)*count : cap;
    if (t >= n) return;
    i64 p = idx[t];
    // the arithmetic of the fused drift (load_pos<true> in cg_particles.hip): x only
    double x = pos[3 * p] + mom[3 * p] * dtm;
    if (!(x > 0 && x < L)) {
        double m = fmod(x, L);
        if (m != 0) {
            if (m < 0) m += L;
        } else {
            m = 0.0;
        }
        if (m == L) m = 0;
        x = m;
    }
    int d = (int)(wrap(cic1(x, geo.off[0], geo.scale).index - g, N) / nxl);
    dest[t] = d;
    atomicAdd(&send_counts[d], 1);
}
int cgk_emigrant_dest(cg_ctx *c, const double *pos, const double *mom, const i64 *idx,
                      const unsigned *count, i64 cap, double dtm, int *dest, int *send_counts) {
    CG_HIP(hipMemsetAsync(send_counts, 0, sizeof(int) * c->p.nprocs, c->stream));
    if (cap == 0) return 0;
    hipLaunchKernelGGL(k_emigrant_dest, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0,
                       c->stream, pos, mom, idx, count, cap, dtm, c->p.boxsize, c->geom_deposit,
                       c->p.nghosts, c->N, c->xmap.nxl, dest, send_counts);
    CG_LAUNCH_CHECK();
    return 0;
}

int cgk_owner_rank(cg_ctx *c, const double *pos, i64 n, int *owner) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_owner_rank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       pos, n, c->geom_deposit, c->p.nghosts, c->N, c->xmap.nxl, owner);
    CG_LAUNCH_CHECK();
    return 0;
}
