// cg_general.hip — the pieces of the GENERAL particle_mesh() that the fused
// fast path does not need: several suppliers and
// receivers, particles and fluids, added together in Fourier space.
//   k_fluid_add         add_fluid_to_grid                 mesh.py:1685-1753
//   k_nullify_nyquist   nullify_modes('nyquist')          mesh.py:3591-3622
//   k_fourier_operate   fourier_operate / copy_modes      mesh.py:3327-3400, 1038-1092
//                       (equal grid sizes) over fourier_loop   mesh.py:2615-2890
//   k_fluid_kick        diff_domaingrid + the fluid branch of
//                       apply_particle_mesh_force         mesh.py:4874-5030,
//                                                         interactions.py:2388-2401
// Fluid grids are double[N][N][N] (the reference's grid_noghosts order); the
// mesh is double[N][N][pad] / complex[N][N][pad/2], un-transposed: the first
// index is x (the reference's ki), the second y (kj).
// Compiled with -ffp-contract=off; every expression keeps the reference's order.
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

// Fluid grids of an x-slab domain are its own layers double[nxl][N][N]; `mesh` is then the
// first OWNED layer of the local buffer.
__global__ __launch_bounds__(256) void k_fluid_add(double *__restrict__ mesh,
                                                   const double *__restrict__ fluid, int N,
                                                   int nxl, i64 ny, i64 pad, double factor,
                                                   int op_add) {
    // one thread per (i, j, k-pair): rows of the fluid grid are contiguous
    const i64 total = (i64)nxl * N * N;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (i64)gridDim.x * blockDim.x) {
        i64 row = t / N;
        int k = (int)(t - row * N);
        double v = fluid[t];
        if (factor != 1) v = v * factor;  // mesh.py:1737-1751
        i64 i = row / N;
        double *dst = mesh + (i * ny + (row - i * N)) * pad + k;
        if (op_add) *dst += v;
        else *dst = v;
    }
}

__global__ __launch_bounds__(256) void k_nullify_nyquist(double2 *__restrict__ four, int N,
                                                         i64 si, i64 cp, int j0, int nj) {
    // Fourier view (cg_ctx::four): planes a == nyq, b == nyq, kk == nyq
    const int nyq = N / 2;
    const i64 rows = (i64)N * nj;
    for (i64 row = (i64)blockIdx.x; row < rows; row += gridDim.x) {
        int a = (int)(row / nj), bl = (int)(row - (i64)a * nj), b = j0 + bl;
        double2 *r = four + (i64)a * si + (i64)bl * cp;
        if (a == nyq || b == nyq) {
            for (int kk = threadIdx.x; kk <= nyq; kk += blockDim.x) r[kk] = make_double2(0, 0);
        } else if (threadIdx.x == 0) {
            r[nyq] = make_double2(0, 0);
        }
    }
}

struct FourierOp {
    const double *tab_n, *tab_s;
    int deconv_order, nlattice, diff_dim, shifted, op_add, zero_nyquist;
    double A, B, Cc;  // -2*pi/N*shift[d]
    double k_fundamental, inv_lat;
};

__global__ __launch_bounds__(256) void k_fourier_operate(const double2 *__restrict__ from,
                                                         double2 *__restrict__ onto, int N,
                                                         i64 si, i64 cp, int j0, int nj,
                                                         FourierOp P) {
#pragma clang fp contract(off)
    const int nyq = N / 2;
    const i64 rows = (i64)N * nj;
    for (i64 row = (i64)blockIdx.x; row < rows; row += gridDim.x) {
        const int a = (int)(row / nj), bl = (int)(row - (i64)a * nj), b = j0 + bl;
        const double2 *src = from + (i64)a * si + (i64)bl * cp;
        double2 *dst = onto + (i64)a * si + (i64)bl * cp;
        const bool dead_row = (a == nyq) || (b == nyq);
        const int ka = a - (a >= nyq ? N : 0), kb = b - (b >= nyq ? N : 0);
        double dab_n = 0, dab_d = 0;
        if (P.deconv_order) {
            dab_n = P.tab_n[a] * P.tab_n[b];  // mesh.py:2797-2798
            dab_d = P.tab_s[a] * P.tab_s[b];
        }
        for (int kk = threadIdx.x; kk <= nyq; kk += blockDim.x) {
            if (dead_row || kk == nyq) {
                if (P.zero_nyquist) dst[kk] = make_double2(0, 0);
                continue;
            }
            double factor = 1;
            if (P.deconv_order) {
                factor = (dab_n * P.tab_n[kk]) / (dab_d * P.tab_s[kk]);  // mesh.py:2850-2853
                double f = factor;
                for (int o = 1; o < P.deconv_order; o++) factor *= f;   // factor **= order
            }
            factor *= P.inv_lat;  // mesh.py:2856
            double re = src[kk].x, im = src[kk].y;
            if (P.shifted) {
                double theta = ((double)ka * P.A + (double)kb * P.B) + (double)kk * P.Cc;
                double c = cos(theta), s = sin(theta);
                double re2 = re * c - im * s, im2 = re * s + im * c;  // mesh.py:3376-3379
                re = re2;
                im = im2;
            }
            if (P.diff_dim >= 0) {
                int kl = P.diff_dim == 0 ? ka : (P.diff_dim == 1 ? kb : kk);
                factor *= P.k_fundamental * (double)kl;  // mesh.py:3391-3392
                double t = re;
                re = -im;
                im = t;
            }
            re *= factor;
            im *= factor;
            if (P.op_add) dst[kk] = make_double2(dst[kk].x + re, dst[kk].y + im);
            else dst[kk] = make_double2(re, im);
        }
    }
}

// copy_modes between DIFFERENT grid sizes (mesh.py:1094-1326): one workgroup per (a, b)
// row of the smaller grid, lanes over kk.  Deconvolution and lattice phase use the grid
// size of `from` (fourier_loop's gridsize_corrections, mesh.py:1245-1250); the extra
// rotation theta = (pi/N_onto - pi/N_from)*((ki + kj) + kk) re-centres the cell-centred
// grid values (mesh.py:1302).
struct CopyModes {
    const double *tab_n, *tab_s;  // of the `from` context (indexed by ITS array index)
    int deconv_order, shifted, op_add;
    double A, B, Cc, dtheta, inv_lat;
};
static CopyModes copy_modes_params(cg_ctx *onto, cg_ctx *from, int deconv_order, int nlattice,
                                    const double *shift, int op_add);
__global__ __launch_bounds__(64) void k_copy_modes(const double2 *__restrict__ from, int N_from,
                                                   i64 ny_from, i64 cp_from,
                                                   double2 *__restrict__ onto, int N_onto,
                                                   i64 ny_onto, i64 cp_onto, CopyModes P) {
#pragma clang fp contract(off)
    const int N_small = N_from < N_onto ? N_from : N_onto;
    const int nyq = N_small / 2;
    const i64 rows = (i64)N_small * N_small;
    for (i64 row = (i64)blockIdx.x; row < rows; row += gridDim.x) {
        const int as = (int)(row / N_small), bs = (int)(row - (i64)as * N_small);
        if (as == nyq || bs == nyq) continue;
        const int ka = as - (as >= nyq ? N_small : 0), kb = bs - (bs >= nyq ? N_small : 0);
        const int af = ka + (ka < 0 ? N_from : 0), bf = kb + (kb < 0 ? N_from : 0);
        const int ao = ka + (ka < 0 ? N_onto : 0), bo = kb + (kb < 0 ? N_onto : 0);
        const double2 *src = from + ((i64)af * ny_from + bf) * cp_from;
        double2 *dst = onto + ((i64)ao * ny_onto + bo) * cp_onto;
        double dab_n = 0, dab_d = 0;
        if (P.deconv_order) {
            dab_n = P.tab_n[af] * P.tab_n[bf];
            dab_d = P.tab_s[af] * P.tab_s[bf];
        }
        for (int kk = threadIdx.x; kk < nyq; kk += blockDim.x) {
            double factor = 1;
            if (P.deconv_order) {
                factor = (dab_n * P.tab_n[kk]) / (dab_d * P.tab_s[kk]);
                double f = factor;
                for (int o = 1; o < P.deconv_order; o++) factor *= f;
            }
            factor *= P.inv_lat;
            double re = src[kk].x, im = src[kk].y;
            double theta = P.dtheta * (double)((ka + kb) + kk);  // mesh.py:1302
            if (P.shifted) theta += ((double)ka * P.A + (double)kb * P.B) + (double)kk * P.Cc;
            double c = cos(theta), s = sin(theta);
            double re2 = factor * (re * c - im * s), im2 = factor * (re * s + im * c);
            if (P.op_add) dst[kk] = make_double2(dst[kk].x + re2, dst[kk].y + im2);
            else dst[kk] = make_double2(re2, im2);
        }
    }
}

// copy_modes between different grid sizes over x-slab domains (get_subslabs / the sub-slab
// exchange of mesh.py:1327-1468): a row kj of the small cube lives on different domains in
// the two grids' decompositions, so it travels.  Pack: the small-cube part of the listed
// local rows of `from`, out[r][as][kk] (as = array index of ki in the SMALL grid, kk < nyq).
// Unpack: what k_copy_modes does per row, reading the packed rows.
__global__ __launch_bounds__(64) void k_cm_pack(const double2 *__restrict__ from, int N_from,
                                                i64 si, i64 cp, const int *__restrict__ rows,
                                                i64 n_rows, int NS, double2 *__restrict__ out) {
    const int nyq = NS / 2;
    for (i64 w = blockIdx.x; w < n_rows * NS; w += gridDim.x) {
        const i64 r = w / NS;
        const int as = (int)(w - r * NS);
        if (as == nyq) continue;
        const int ka = as - (as >= nyq ? NS : 0);
        const int af = ka + (ka < 0 ? N_from : 0);
        const double2 *src = from + (i64)af * si + (i64)rows[r] * cp;
        double2 *dst = out + (r * NS + as) * nyq;
        for (int kk = threadIdx.x; kk < nyq; kk += blockDim.x) dst[kk] = src[kk];
    }
}
__global__ __launch_bounds__(64) void k_cm_unpack(const double2 *__restrict__ in, int NS,
                                                  const int *__restrict__ rows, i64 n_rows,
                                                  double2 *__restrict__ onto, int N_onto, i64 si,
                                                  i64 cp, int j0, int N_from, CopyModes P) {
#pragma clang fp contract(off)
    const int nyq = NS / 2;
    for (i64 w = blockIdx.x; w < n_rows * NS; w += gridDim.x) {
        const i64 r = w / NS;
        const int as = (int)(w - r * NS);
        if (as == nyq) continue;
        const int bo = j0 + rows[r];
        const int kb = bo - (bo >= N_onto / 2 ? N_onto : 0);
        const int ka = as - (as >= nyq ? NS : 0);
        const int af = ka + (ka < 0 ? N_from : 0), bf = kb + (kb < 0 ? N_from : 0);
        const int ao = ka + (ka < 0 ? N_onto : 0);
        const double2 *src = in + (r * NS + as) * nyq;
        double2 *dst = onto + (i64)ao * si + (i64)rows[r] * cp;
        double dab_n = 0, dab_d = 0;
        if (P.deconv_order) {
            dab_n = P.tab_n[af] * P.tab_n[bf];
            dab_d = P.tab_s[af] * P.tab_s[bf];
        }
        for (int kk = threadIdx.x; kk < nyq; kk += blockDim.x) {
            double factor = 1;
            if (P.deconv_order) {
                factor = (dab_n * P.tab_n[kk]) / (dab_d * P.tab_s[kk]);
                double f = factor;
                for (int o = 1; o < P.deconv_order; o++) factor *= f;
            }
            factor *= P.inv_lat;
            double re = src[kk].x, im = src[kk].y;
            double theta = P.dtheta * (double)((ka + kb) + kk);  // mesh.py:1302
            if (P.shifted) theta += ((double)ka * P.A + (double)kb * P.B) + (double)kk * P.Cc;
            double c = cos(theta), s = sin(theta);
            double re2 = factor * (re * c - im * s), im2 = factor * (re * s + im * c);
            if (P.op_add) dst[kk] = make_double2(dst[kk].x + re2, dst[kk].y + im2);
            else dst[kk] = make_double2(re2, im2);
        }
    }
}

// diff_domaingrid (mesh.py:4874-5030): the value of one force cell from the potential along
// one dimension, P(s) = the potential s cells away.  Coefficients and the order of the terms
// are the reference's; order 1 is its one-sided 'forward' difference.
struct FdCoef {
    double c1, c2, c3, c4;
};
template <int ORDER, class P>
__device__ __forceinline__ double fd_value(const P &phi, const FdCoef &c) {
#pragma clang fp contract(off)
    if (ORDER == 0) return phi(0);  // the mesh already holds the force (Fourier-space gradient)
    if (ORDER == 1) return c.c1 * (phi(1) - phi(0));                                // mesh.py:4962
    if (ORDER == 2) return c.c1 * (phi(1) - phi(-1));                               // mesh.py:4967
    if (ORDER == 4) return c.c1 * (phi(1) - phi(-1)) - c.c2 * (phi(2) - phi(-2));   // mesh.py:4973
    if (ORDER == 6)                                                                  // mesh.py:4984
        return (c.c1 * (phi(1) - phi(-1)) - c.c2 * (phi(2) - phi(-2))) + c.c3 * (phi(3) - phi(-3));
    return ((c.c1 * (phi(1) - phi(-1)) - c.c2 * (phi(2) - phi(-2))) +                // mesh.py:4999
            c.c3 * (phi(3) - phi(-3))) - c.c4 * (phi(4) - phi(-4));
}
static bool fd_coefficients(int order, double dx, FdCoef &c) {
    c = FdCoef{0, 0, 0, 0};
    switch (order) {
        case 0: return true;
        case 1: c.c1 = 1 / dx; return true;
        case 2: c.c1 = (1.0 / 2) / dx; return true;
        case 4: c.c1 = (2.0 / 3) / dx; c.c2 = (1.0 / 12) / dx; return true;
        case 6: c.c1 = (3.0 / 4) / dx; c.c2 = (3.0 / 20) / dx; c.c3 = (1.0 / 60) / dx; return true;
        case 8:
            c.c1 = (4.0 / 5) / dx; c.c2 = (1.0 / 5) / dx; c.c3 = (4.0 / 105) / dx;
            c.c4 = (1.0 / 280) / dx;
            return true;
    }
    return false;
}
#define CG_FD_SWITCH(order, KERNEL, ...)                                          \
    switch (order) {                                                              \
        case 0: hipLaunchKernelGGL(KERNEL<0>, __VA_ARGS__); break;                \
        case 1: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;                \
        case 2: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;                \
        case 4: hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); break;                \
        case 6: hipLaunchKernelGGL(KERNEL<6>, __VA_ARGS__); break;                \
        default: hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__); break;               \
    }

template <int ORDER>
__global__ __launch_bounds__(256) void k_fluid_kick(double *__restrict__ J,
                                                    const double *__restrict__ rho,
                                                    const double *__restrict__ P,
                                                    const double *__restrict__ mesh, int N,
                                                    XMap xm, i64 ny, i64 pad, int dim, FdCoef fc,
                                                    double mdt, double inv_c2) {
#pragma clang fp contract(off)
    // `mesh` is the local buffer (ghost layers included); the fluid grids hold the owned layers
    const i64 total = (i64)xm.nxl * N * N;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (i64)gridDim.x * blockDim.x) {
        i64 row = t / N;
        int k = (int)(t - row * N);
        int i = (int)(row / N), j = (int)(row - (i64)i * N);
        auto phi = [&](int s) {
            i64 ii = cg_xlayer(xm, xm.x0 + i + (dim == 0 ? s : 0), N);
            int jj = j, kk = k;
            if (dim == 1) jj = (j + s + N) % N;
            else if (dim == 2) kk = (k + s + N) % N;
            return mesh[(ii * ny + jj) * pad + kk];
        };
        const double g = fd_value<ORDER>(phi, fc);
        // Jᵢ += ℝ[-ᔑdt]*(ϱ + ℝ[c⁻²]*𝒫)*grid   (interactions.py:2397-2400)
        J[t] += mdt * (rho[t] + inv_c2 * P[t]) * g;
    }
}

// ---------------------------------------------------------------------------
// Particle interpolation of every order: set_weights_NGP / CIC /
// TSC / PCS (mesh.py:5305-5394) and the loops over ORDER^3 grid points
// (mesh.py:5052-5283), with the coordinate map of interpolate_particles
// (mesh.py:1577-1606, lattice shift subtracted) or interpolate_domaingrid_to_particles
// (mesh.py:409-432, lattice shift added) in `geo`.  Direct form: one lane per particle,
// device-scope FP64 atomics for the deposit (these orders are accuracy options, not the
// benchmarked path; the tiled CIC kernels keep the default configuration).
// ---------------------------------------------------------------------------
template <int ORDER>
__device__ __forceinline__ int set_weights(double x, double (&w)[4]) {
#pragma clang fp contract(off)
    if (ORDER == 1) {
        int index = (int)(x + 0.5);
        w[0] = 1;
        return index;
    }
    if (ORDER == 2) {
        int index = (int)x;
        double dist = x - (double)index;
        w[0] = 1 - dist;
        w[1] = dist;
        return index;
    }
    if (ORDER == 3) {
        int index = (int)(x + 0.5);
        double dist = x - (double)index;
        index -= 1;
        double dist2 = dist * dist;
        double weight0 = 0.125 + 0.5 * (dist2 - dist);
        double weight1 = 0.75 - dist2;
        w[0] = weight0;
        w[1] = weight1;
        w[2] = 1 - weight0 - weight1;
        return index;
    }
    int index = (int)x;
    index -= 1;
    double dist = x - (double)index;
    double tmp = 2 - dist;
    double tmp2 = tmp * tmp;
    double tmp3 = tmp * tmp2;
    double weight0 = 1. / 6. * tmp3;
    double weight2 = 2. / 3. - tmp2 + 0.5 * tmp3;
    double d1 = dist - 1;
    double weight3 = 1. / 6. * (d1 * d1 * d1);
    w[0] = weight0;
    w[1] = 1 - weight0 - weight2 - weight3;
    w[2] = weight2;
    w[3] = weight3;
    return index;
}
__device__ __forceinline__ int wrap32(int a, int n) {
    a = a < 0 ? a + n : a;
    return a >= n ? a - n : a;
}

// ---------------------------------------------------------------------------
// Deposit and scalar gather of order 1-4 with lattice shifts (mesh.py:1512-1636, :376-459),
// round 5: through LDS boxes, like cg_deposit_cic / cg_gather_kick (cg_mesh_kernels.hip).  A
// workgroup takes 2048 consecutive particles, finds the box of their stencils' first cells
// relative to the first particle's (the periodic seam is no seam) and, when the box with the
// stencil's width fits 4096 cells of LDS, accumulates there and adds the box to the mesh once — one
// device-scope FP64 atomic (memory-side on MI355X) per touched cell instead of ORDER^3 per
// particle — or copies the box of the field in before the gather.  Particles in tile order of
// ANY mesh are compact enough; a chunk that is spread out takes the direct accesses.  Index and
// weight arithmetic unchanged (the reference's expressions, set_weights above).
// ---------------------------------------------------------------------------
constexpr int kGcPer = 4, kGcLanes = 512, kGcCells = 4096;

// the box of a chunk: s_ref = first particle's first cell, s_lo / s_hi = extremes of the others'
// first cells relative to it (in [-N/2, N/2) through the seam); called by all lanes
template <int ORDER>
struct GcBox {
    int r[3], l[3], e[3];
    bool fits;
};
__device__ __forceinline__ int gc_rel(int cell, int ref, int N) {
    int d = cell - ref;
    d += d < -(N / 2) ? N : 0;
    d -= d >= N - N / 2 ? N : 0;
    return d;
}
template <int ORDER>
__device__ __forceinline__ GcBox<ORDER> gc_box(const int (&cell)[kGcPer][3],
                                               const bool (&valid)[kGcPer], int N, int *s_ref,
                                               int *s_lo, int *s_hi) {
    const int tid = threadIdx.x;
    if (tid == 0) {  // (the chunk's first particle exists)
        for (int d = 0; d < 3; d++) {
            s_ref[d] = cell[0][d];
            s_lo[d] = 0;
            s_hi[d] = 0;
        }
    }
    __syncthreads();
    GcBox<ORDER> B;
    for (int d = 0; d < 3; d++) B.r[d] = s_ref[d];
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
#pragma unroll
    for (int u = 0; u < kGcPer; u++) {
        if (!valid[u]) continue;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const int v = gc_rel(cell[u][d], B.r[d], N);
            lo[d] = min(lo[d], v);
            hi[d] = max(hi[d], v);
        }
    }
#pragma unroll
    for (int d = 0; d < 3; d++) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            lo[d] = min(lo[d], __shfl_xor(lo[d], m));
            hi[d] = max(hi[d], __shfl_xor(hi[d], m));
        }
        if ((tid & 63) == 0) {
            atomicMin(&s_lo[d], lo[d]);
            atomicMax(&s_hi[d], hi[d]);
        }
    }
    __syncthreads();
    for (int d = 0; d < 3; d++) {
        B.l[d] = s_lo[d];
        B.e[d] = s_hi[d] - s_lo[d] + ORDER;
    }
    B.fits = (i64)B.e[0] * B.e[1] * B.e[2] <= kGcCells && B.e[0] <= N && B.e[1] <= N && B.e[2] <= N;
    return B;
}

template <int ORDER>
__global__ __launch_bounds__(kGcLanes) void k_deposit_general(const double *__restrict__ pos, i64 n,
                                                              double *__restrict__ mesh, int N,
                                                              i64 ny, i64 pad, int g, XMap xm,
                                                              CicGeom geo, double contribution) {
#pragma clang fp contract(off)
    __shared__ double blk[kGcCells];
    __shared__ int s_ref[3], s_lo[3], s_hi[3];
    const int tid = threadIdx.x;
    const i64 base = (i64)blockIdx.x * (kGcLanes * kGcPer);
    double px[kGcPer][3];
    int cell[kGcPer][3];
    bool valid[kGcPer];
#pragma unroll
    for (int u = 0; u < kGcPer; u++) {
        const i64 p = base + tid + (i64)kGcLanes * u;
        valid[u] = p < n;
        double w[4];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            px[u][d] = valid[u] ? pos[3 * p + d] : 0.0;
            cell[u][d] = wrap32(set_weights<ORDER>((px[u][d] - geo.off[d]) * geo.scale, w) - g, N);
        }
    }
    const GcBox<ORDER> B = gc_box<ORDER>(cell, valid, N, s_ref, s_lo, s_hi);
    const int ncells = B.fits ? B.e[0] * B.e[1] * B.e[2] : 0;
    for (int i = tid; i < ncells; i += kGcLanes) blk[i] = 0;
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < kGcPer; u++) {
        if (!valid[u]) continue;
        double wx[4], wy[4], wz[4];
        const int ii = set_weights<ORDER>((px[u][0] - geo.off[0]) * geo.scale, wx) - g;
        const int jj = set_weights<ORDER>((px[u][1] - geo.off[1]) * geo.scale, wy) - g;
        const int kk = set_weights<ORDER>((px[u][2] - geo.off[2]) * geo.scale, wz) - g;
        const int a0 = gc_rel(cell[u][0], B.r[0], N) - B.l[0],
                  b0 = gc_rel(cell[u][1], B.r[1], N) - B.l[1],
                  c0 = gc_rel(cell[u][2], B.r[2], N) - B.l[2];
#pragma unroll
        for (int i = 0; i < ORDER; i++) {
            double weight_i = wx[i];
            weight_i *= contribution;  // apply_factor = True
            const i64 ri = B.fits ? 0 : cg_xlayer(xm, (i64)(ii + i), N) * ny;
#pragma unroll
            for (int j = 0; j < ORDER; j++) {
                const double wij = weight_i * wy[j];
                double *row = B.fits ? blk + ((a0 + i) * B.e[1] + (b0 + j)) * B.e[2] + c0
                                     : mesh + (ri + wrap32(jj + j, N)) * pad;
#pragma unroll
                for (int k = 0; k < ORDER; k++)
                    unsafeAtomicAdd(row + (B.fits ? k : wrap32(kk + k, N)),
                                    ORDER == 1 ? weight_i : wij * wz[k]);
            }
        }
    }
    if (!B.fits) return;  // (uniform)
    __syncthreads();
    for (int i = tid; i < ncells; i += kGcLanes) {
        const double v = blk[i];
        if (v == 0) continue;
        const int c = i % B.e[2], b = (i / B.e[2]) % B.e[1], a = i / (B.e[2] * B.e[1]);
        // (cell index before the wrap in [-N, 2N): |rel| <= N/2, extent <= N)
        const i64 gi = cg_xlayer(xm, (i64)wrap32(wrap32(B.r[0] + B.l[0], N) + a, N), N);
        const int gj = wrap32(wrap32(B.r[1] + B.l[1], N) + b, N),
                  gk = wrap32(wrap32(B.r[2] + B.l[2], N) + c, N);
        unsafeAtomicAdd(mesh + (gi * ny + gj) * pad + gk, v);
    }
}

template <int ORDER>
__global__ __launch_bounds__(kGcLanes) void k_gather_scalar(const double *__restrict__ pos,
                                                            double *__restrict__ mom, i64 n,
                                                            int dim,
                                                            const double *__restrict__ mesh, int N,
                                                            i64 ny, i64 pad, int g, XMap xm,
                                                            CicGeom geo, double factor) {
#pragma clang fp contract(off)
    __shared__ double blk[kGcCells];
    __shared__ int s_ref[3], s_lo[3], s_hi[3];
    const int tid = threadIdx.x;
    const i64 base = (i64)blockIdx.x * (kGcLanes * kGcPer);
    double px[kGcPer][3];
    int cell[kGcPer][3];
    bool valid[kGcPer];
#pragma unroll
    for (int u = 0; u < kGcPer; u++) {
        const i64 p = base + tid + (i64)kGcLanes * u;
        valid[u] = p < n;
        double w[4];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            px[u][d] = valid[u] ? pos[3 * p + d] : 0.0;
            cell[u][d] = wrap32(set_weights<ORDER>((px[u][d] - geo.off[d]) * geo.scale, w) - g, N);
        }
    }
    const GcBox<ORDER> B = gc_box<ORDER>(cell, valid, N, s_ref, s_lo, s_hi);
    if (B.fits) {
        const int ncells = B.e[0] * B.e[1] * B.e[2];
        for (int i = tid; i < ncells; i += kGcLanes) {
            const int c = i % B.e[2], b = (i / B.e[2]) % B.e[1], a = i / (B.e[2] * B.e[1]);
            const i64 gi = cg_xlayer(xm, (i64)wrap32(wrap32(B.r[0] + B.l[0], N) + a, N), N);
            const int gj = wrap32(wrap32(B.r[1] + B.l[1], N) + b, N),
                      gk = wrap32(wrap32(B.r[2] + B.l[2], N) + c, N);
            blk[i] = mesh[(gi * ny + gj) * pad + gk];
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < kGcPer; u++) {
        const i64 p = base + tid + (i64)kGcLanes * u;
        if (p >= n) continue;
        double wx[4], wy[4], wz[4];
        const int ii = set_weights<ORDER>((px[u][0] - geo.off[0]) * geo.scale, wx) - g;
        const int jj = set_weights<ORDER>((px[u][1] - geo.off[1]) * geo.scale, wy) - g;
        const int kk = set_weights<ORDER>((px[u][2] - geo.off[2]) * geo.scale, wz) - g;
        const int a0 = gc_rel(cell[u][0], B.r[0], N) - B.l[0],
                  b0 = gc_rel(cell[u][1], B.r[1], N) - B.l[1],
                  c0 = gc_rel(cell[u][2], B.r[2], N) - B.l[2];
        double value = 0;
#pragma unroll
        for (int i = 0; i < ORDER; i++) {
            const i64 ri = B.fits ? 0 : cg_xlayer(xm, (i64)(ii + i), N) * ny;
#pragma unroll
            for (int j = 0; j < ORDER; j++) {
                const double wij = wx[i] * wy[j];
                const double *row = B.fits ? blk + ((a0 + i) * B.e[1] + (b0 + j)) * B.e[2] + c0
                                           : mesh + (ri + wrap32(jj + j, N)) * pad;
#pragma unroll
                for (int k = 0; k < ORDER; k++)
                    value += row[B.fits ? k : wrap32(kk + k, N)] * (ORDER == 1 ? 1.0 : wij * wz[k]);
            }
        }
        if (factor != 1) value *= factor;  // mesh.py:456-458
        mom[3 * p + dim] += value;
    }
}

// diff_domaingrid (mesh.py:4874-5030) of the real-space mesh of `src` into `dst`
template <int ORDER>
__global__ __launch_bounds__(256) void k_mesh_diff(double *__restrict__ dst,
                                                   const double *__restrict__ src, int N,
                                                   XMap xm, i64 ny, i64 pad, int dim, FdCoef fc) {
#pragma clang fp contract(off)
    // both are local buffers (ghost layers included); the owned layers of dst are written
    const i64 total = (i64)xm.nxl * N * N;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (i64)gridDim.x * blockDim.x) {
        i64 row = t / N;
        int k = (int)(t - row * N);
        int i = (int)(row / N), j = (int)(row - (i64)i * N);
        auto phi = [&](int s) {
            i64 ii = cg_xlayer(xm, xm.x0 + i + (dim == 0 ? s : 0), N);
            int jj = j, kk = k;
            if (dim == 1) jj = wrap32(j + s, N);
            else if (dim == 2) kk = wrap32(k + s, N);
            return src[(ii * ny + jj) * pad + kk];
        };
        const double gval = fd_value<ORDER>(phi, fc);
        dst[(((i64)i + xm.G) * ny + j) * pad + k] = gval;
    }
}

static unsigned blocks_for(i64 n, int per) {
    i64 b = (n + per - 1) / per;
    if (b > 256 * 64) b = 256 * 64;
    if (b < 1) b = 1;
    return (unsigned)b;
}

int cgk_fluid_add(cg_ctx *c, const double *fluid, double factor, int op_add) {
    i64 total = c->xmap.nxl * c->N * c->N;
    hipLaunchKernelGGL(k_fluid_add, dim3(blocks_for(total, 256)), dim3(256), 0, c->stream,
                       c->mesh0, fluid, (int)c->N, (int)c->xmap.nxl, c->ny, c->pad, factor, op_add);
    CG_LAUNCH_CHECK();
    return 0;
}

int cgk_nullify_nyquist(cg_ctx *c) {
    hipLaunchKernelGGL(k_nullify_nyquist, dim3(blocks_for(c->N * c->f_nj, 1)), dim3(64), 0,
                       c->stream, c->four, (int)c->N, c->f_si, c->pad / 2, c->f_j0, c->f_nj);
    CG_LAUNCH_CHECK();
    return 0;
}

int cgk_fourier_operate(cg_ctx *onto, cg_ctx *from, int deconv_order, int nlattice,
                        const double *shift, int diff_dim, int op_add) {
    const double kPiLocal = 3.141592653589793;
    FourierOp P{};
    P.tab_n = onto->ktab_n;
    P.tab_s = onto->ktab_s;
    P.deconv_order = deconv_order;
    P.nlattice = nlattice;
    P.diff_dim = diff_dim;
    P.op_add = op_add;
    P.shifted = shift && (shift[0] != 0 || shift[1] != 0 || shift[2] != 0);
    P.zero_nyquist = (from != onto) && !op_add;
    if (P.shifted) {
        // ℝ[-2*π/gridsize_corrections*interlace_lattice.shift[d]]  (mesh.py:2863-2878)
        P.A = -2 * kPiLocal / (double)onto->N * shift[0];
        P.B = -2 * kPiLocal / (double)onto->N * shift[1];
        P.Cc = -2 * kPiLocal / (double)onto->N * shift[2];
    }
    P.k_fundamental = 2 * kPiLocal / onto->p.boxsize;  // mesh.py:3362
    P.inv_lat = 1.0 / (double)nlattice;
    hipLaunchKernelGGL(k_fourier_operate, dim3(blocks_for(onto->N * onto->f_nj, 1)), dim3(256), 0,
                       onto->stream, (const double2 *)from->four, onto->four, (int)onto->N,
                       onto->f_si, onto->pad / 2, onto->f_j0, onto->f_nj, P);
    CG_LAUNCH_CHECK();
    return 0;
}

int cgk_copy_modes(cg_ctx *onto, cg_ctx *from, int deconv_order, int nlattice,
                   const double *shift, int op_add) {
    CopyModes P = copy_modes_params(onto, from, deconv_order, nlattice, shift, op_add);
    i64 ns = onto->N < from->N ? onto->N : from->N;
    hipLaunchKernelGGL(k_copy_modes, dim3(blocks_for(ns * ns, 1)), dim3(64), 0, onto->stream,
                       (const double2 *)from->mesh0, (int)from->N, from->ny, from->pad / 2,
                       (double2 *)onto->mesh0, (int)onto->N, onto->ny, onto->pad / 2, P);
    CG_LAUNCH_CHECK();
    return 0;
}

static CopyModes copy_modes_params(cg_ctx *onto, cg_ctx *from, int deconv_order, int nlattice,
                                    const double *shift, int op_add) {
    const double kPiLocal = 3.141592653589793;
    CopyModes P{};
    P.tab_n = from->ktab_n;  // kk*pi/N_from + eps by array index of the `from` grid
    P.tab_s = from->ktab_s;
    P.deconv_order = deconv_order;
    P.op_add = op_add;
    P.shifted = shift && (shift[0] != 0 || shift[1] != 0 || shift[2] != 0);
    if (P.shifted) {
        P.A = -2 * kPiLocal / (double)from->N * shift[0];
        P.B = -2 * kPiLocal / (double)from->N * shift[1];
        P.Cc = -2 * kPiLocal / (double)from->N * shift[2];
    }
    P.dtheta = kPiLocal / (double)onto->N - kPiLocal / (double)from->N;
    P.inv_lat = 1.0 / (double)nlattice;
    return P;
}

int cgk_copy_modes_pack(cg_ctx *from, i64 n_small, const int *rows_local, i64 n_rows,
                        double *out) {
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(k_cm_pack, dim3(blocks_for(n_rows * n_small, 1)), dim3(64), 0, from->stream,
                       (const double2 *)from->four, (int)from->N, from->f_si, from->pad / 2,
                       rows_local, n_rows, (int)n_small, (double2 *)out);
    CG_LAUNCH_CHECK();
    return 0;
}

int cgk_copy_modes_unpack(cg_ctx *onto, cg_ctx *from, i64 n_small, const int *rows_local,
                          i64 n_rows, const double *in, int deconv_order, int nlattice,
                          const double *shift, int op_add) {
    if (n_rows == 0) return 0;
    CopyModes P = copy_modes_params(onto, from, deconv_order, nlattice, shift, op_add);
    hipLaunchKernelGGL(k_cm_unpack, dim3(blocks_for(n_rows * n_small, 1)), dim3(64), 0,
                       onto->stream, (const double2 *)in, (int)n_small, rows_local, n_rows,
                       onto->four, (int)onto->N, onto->f_si, onto->pad / 2, onto->f_j0,
                       (int)from->N, P);
    CG_LAUNCH_CHECK();
    return 0;
}

int cgk_fluid_kick(cg_ctx *c, double *J, const double *rho, const double *P, int dim,
                   int diff_order, double minus_dt, double inv_c2) {
    i64 total = c->xmap.nxl * c->N * c->N;
    double dx = c->p.boxsize / (double)c->N;  // interactions.py:2133
    FdCoef fc;
    if (!fd_coefficients(diff_order, dx, fc)) {
        cg_set_error("cg_fluid_kick: differentiation order %d", diff_order);
        return 1;
    }
    CG_FD_SWITCH(diff_order, k_fluid_kick, dim3(blocks_for(total, 256)), dim3(256), 0, c->stream,
                 J, rho, P, c->mesh, (int)c->N, c->xmap, c->ny, c->pad, dim, fc, minus_dt, inv_c2)
    CG_LAUNCH_CHECK();
    return 0;
}

#define CG_ORDER_SWITCH(order, KERNEL, ...)                                       \
    switch (order) {                                                              \
        case 1: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;                \
        case 2: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;                \
        case 3: hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); break;                \
        default: hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); break;               \
    }

int cgk_deposit_general(cg_ctx *c, const double *pos, i64 n, double contribution, int order,
                        const CicGeom &geo) {
    if (n == 0) return 0;
    CG_ORDER_SWITCH(order, k_deposit_general,
                    dim3((unsigned)((n + kGcLanes * kGcPer - 1) / (kGcLanes * kGcPer))),
                    dim3(kGcLanes), 0, c->stream, pos, n, c->mesh, (int)c->N, c->ny, c->pad,
                    c->p.nghosts, c->xmap, geo, contribution)
    CG_LAUNCH_CHECK();
    return 0;
}

int cgk_gather_scalar(cg_ctx *c, const double *pos, double *mom, i64 n, int dim, int order,
                      const CicGeom &geo, double factor) {
    if (n == 0) return 0;
    CG_ORDER_SWITCH(order, k_gather_scalar,
                    dim3((unsigned)((n + kGcLanes * kGcPer - 1) / (kGcLanes * kGcPer))),
                    dim3(kGcLanes), 0, c->stream, pos, mom, n, dim, c->mesh, (int)c->N, c->ny,
                    c->pad, c->p.nghosts, c->xmap, geo, factor)
    CG_LAUNCH_CHECK();
    return 0;
}

int cgk_mesh_diff(cg_ctx *dst, cg_ctx *src, int dim, int diff_order) {
    i64 total = src->xmap.nxl * src->N * src->N;
    double dx = src->p.boxsize / (double)src->N;
    FdCoef fc;
    if (diff_order == 0 || !fd_coefficients(diff_order, dx, fc)) {
        cg_set_error("cg_mesh_diff: differentiation order %d", diff_order);
        return 1;
    }
    CG_FD_SWITCH(diff_order, k_mesh_diff, dim3(blocks_for(total, 256)), dim3(256), 0, dst->stream,
                 dst->mesh, src->mesh, (int)src->N, src->xmap, src->ny, src->pad, dim, fc)
    CG_LAUNCH_CHECK();
    return 0;
}
