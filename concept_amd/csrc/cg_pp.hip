// cg_pp.hip — direct summation gravity with the Ewald correction.
//   k_ewald_tabulate   ewald.tabulate -> summation           ewald.py:62-118, 226-231
//   k_pp_kick          gravity_pairwise (periodic: nearest image + Ewald look-up,
//                      gravity.py:160-182, ewald.py:146-197 with the CIC vector
//                      interpolation of mesh.py:293-337) and
//                      gravity_pairwise_nonperiodic (gravity.py:520-533)
// One-sided like the short-range sweep (cg_shortrange.hip): every receiver sums over all
// suppliers; the reference visits a pair once and updates both members.  O(N^2): the
// method of small particle numbers; suppliers are staged through LDS in chunks.
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

__global__ __launch_bounds__(64) void k_ewald_tabulate(double *__restrict__ grid, int gs) {
#pragma clang fp contract(off)
    // ewald.py:245-271
    const double rs = 0.25, maxdist = 3.6, pi = 3.141592653589793;
    const int maxh2 = 10, h_lower = -3, h_upper = 4, n_lower = -4, n_upper = 5;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= gs * gs * gs) return;
    const int k = t % gs, j = (t / gs) % gs, i = t / (gs * gs);
    const double factor = 0.5 / (double)(gs - 1);
    const double x = (double)i * factor, y = (double)j * factor, z = (double)k * factor;
    double fx = 0, fy = 0, fz = 0;
    if (!(x == 0 && y == 0 && z == 0)) {
        double r3 = x * x + y * y + z * z;
        r3 *= sqrt(r3);
        fx += x * (1 / r3);
        fy += y * (1 / r3);
        fz += z * (1 / r3);
        for (int sx = n_lower; sx < n_upper; sx++) {
            double dx = x - sx;
            for (int sy = n_lower; sy < n_upper; sy++) {
                double dy = y - sy;
                for (int sz = n_lower; sz < n_upper; sz++) {
                    double dz = z - sz;
                    double dist2 = (dx * dx + dy * dy) + dz * dz;
                    if (dist2 > maxdist * maxdist) continue;
                    double dist = sqrt(dist2);
                    double inv3 = 1 / (dist * dist * dist);  // dist**(-3)
                    double scalarpart =
                        -inv3 * (erfc(dist * (1 / (2 * rs))) +
                                 dist * (1 / (sqrt(pi) * rs)) * exp(dist2 * (-1 / (4 * rs * rs))));
                    fx += dx * scalarpart;
                    fy += dy * scalarpart;
                    fz += dz * scalarpart;
                }
            }
        }
        for (int sx = h_lower; sx < h_upper; sx++) {
            double kx = 2 * pi * sx;
            for (int sy = h_lower; sy < h_upper; sy++) {
                double ky = 2 * pi * sy;
                for (int sz = h_lower; sz < h_upper; sz++) {
                    int h2 = (sx * sx + sy * sy) + sz * sz;
                    if (h2 > maxh2 || h2 == 0) continue;
                    double kz = 2 * pi * sz;
                    double k2 = (kx * kx + ky * ky) + kz * kz;
                    double scalarpart =
                        -4 * pi / k2 * exp(-k2 * (rs * rs)) * sin(kx * x + ky * y + kz * z);
                    fx += kx * scalarpart;
                    fy += ky * scalarpart;
                    fz += kz * scalarpart;
                }
            }
        }
    }
    grid[3 * (i64)t + 0] = fx;
    grid[3 * (i64)t + 1] = fy;
    grid[3 * (i64)t + 2] = fz;
}

struct PpParams {
    double boxsize, half_box, scale, inv_box2, softening, factor;
    const double *ewald;  // [gs][gs][gs][3] or null (non-periodic)
    int gs, kernel, same;
    // adaptive rungs (null = one rung): per-rung factors, receiver rung arrays
    const double *factors;
    const signed char *rung, *rung_jumped;
    int lowest_active;
};

__device__ __forceinline__ double pp_softened_r3inv(double r2, double eps, int kernel) {
#pragma clang fp contract(off)
    // get_softened_r3inv, interactions.py:1847-1914
    if (kernel == 0) {
        double r3 = r2 * sqrt(r2);
        return r3 == 0 ? 0 : 1 / r3;
    }
    if (kernel == 1) {
        double s = r2 + eps * eps;
        return 1 / (s * sqrt(s));
    }
    double h = 2.8 * eps, r = sqrt(r2);
    if (r >= h) return 1 / (r2 * r);
    double u = r / h;
    if (u < 0.5) return 32 / (h * h * h) * (1. / 3. + u * u * (-6. / 5. + u));
    return 32 / (3 * (r * r * r)) *
           ((u * u * u) * (2 + u * (-9. / 2. + u * (18. / 5. - u))) - 3. / 480.);
}

__global__ __launch_bounds__(256) void k_pp_kick(const double *__restrict__ pos_r, i64 n_r,
                                                 double *__restrict__ dmom_r,
                                                 const double *__restrict__ pos_s, i64 n_s,
                                                 PpParams P) {
#pragma clang fp contract(off)
    __shared__ double sx[256], sy[256], sz[256];
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    bool active = i < n_r;
    double my_factor = P.factor;
    if (active && P.rung) {
        if (P.rung[i] < P.lowest_active) active = false;
        else my_factor = P.factors[P.rung_jumped[i]];
    }
    double xi = 0, yi = 0, zi = 0;
    if (i < n_r) {
        xi = pos_r[3 * i];
        yi = pos_r[3 * i + 1];
        zi = pos_r[3 * i + 2];
    }
    double ax = 0, ay = 0, az = 0;
    for (i64 base = 0; base < n_s; base += 256) {
        __syncthreads();
        i64 j = base + threadIdx.x;
        if (j < n_s) {
            sx[threadIdx.x] = pos_s[3 * j];
            sy[threadIdx.x] = pos_s[3 * j + 1];
            sz[threadIdx.x] = pos_s[3 * j + 2];
        }
        __syncthreads();
        const int cnt = (int)(n_s - base < 256 ? n_s - base : 256);
        if (!active) continue;
        for (int k = 0; k < cnt; k++) {
            if (P.same && base + k == i) continue;
            double x = xi - sx[k], y = yi - sy[k], z = zi - sz[k];
            double f0 = 0, f1 = 0, f2 = 0;
            if (P.ewald) {
                // nearest image (gravity.py:160-171)
                if (x > P.half_box) x -= P.boxsize;
                else if (x < -P.half_box) x += P.boxsize;
                if (y > P.half_box) y -= P.boxsize;
                else if (y < -P.half_box) y += P.boxsize;
                if (z > P.half_box) z -= P.boxsize;
                else if (z < -P.half_box) z += P.boxsize;
                // ewald(x, y, z), ewald.py:146-197
                double c[3] = {x, y, z};
                bool neg[3];
                int idx[3];
                double w[3][2];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    neg[d] = !(c[d] > 0);
                    if (neg[d]) c[d] *= -1;
                    double g = c[d] * P.scale;
                    idx[d] = (int)g;
                    double dist = g - (double)idx[d];
                    w[d][0] = 1 - dist;
                    w[d][1] = dist;
                }
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    double wi = w[0][a];
                    wi *= P.inv_box2;
#pragma unroll
                    for (int b = 0; b < 2; b++) {
                        double wij = wi * w[1][b];
#pragma unroll
                        for (int cc = 0; cc < 2; cc++) {
                            double weight = wij * w[2][cc];
                            const double *g =
                                P.ewald +
                                (((i64)(idx[0] + a) * P.gs + (idx[1] + b)) * P.gs + (idx[2] + cc)) * 3;
                            f0 += g[0] * weight;
                            f1 += g[1] * weight;
                            f2 += g[2] * weight;
                        }
                    }
                }
                if (neg[0]) f0 *= -1;
                if (neg[1]) f1 *= -1;
                if (neg[2]) f2 *= -1;
            }
            double r2 = x * x + y * y + z * z;
            double r3inv = pp_softened_r3inv(r2, P.softening, P.kernel);
            ax += my_factor * (f0 - x * r3inv);  // gravity.py:175-182
            ay += my_factor * (f1 - y * r3inv);
            az += my_factor * (f2 - z * r3inv);
        }
    }
    if (active) {
        dmom_r[3 * i] += ax;
        dmom_r[3 * i + 1] += ay;
        dmom_r[3 * i + 2] += az;
    }
}

int cgk_ewald_tabulate(cg_ctx *c, int gridsize, double *grid) {
    int total = gridsize * gridsize * gridsize;
    hipLaunchKernelGGL(k_ewald_tabulate, dim3((total + 63) / 64), dim3(64), 0, c->stream, grid,
                       gridsize);
    CG_LAUNCH_CHECK();
    return 0;
}

int cgk_pp_kick(cg_ctx *c, const double *pos_r, i64 n_r, double *dmom_r, const double *pos_s,
                i64 n_s, int same, const double *ewald_grid, int ewald_gridsize,
                double softening, int kernel, double factor, const double *factors,
                const signed char *rung, const signed char *rung_jumped, int lowest_active) {
    if (n_r == 0 || n_s == 0) return 0;
    PpParams P{};
    P.boxsize = c->p.boxsize;
    P.half_box = 0.5 * c->p.boxsize;
    // ℝ[2/boxsize*(ewald_gridsize - 1)*(1 - machine_ϵ)], ℝ[1/boxsize**2]   (ewald.py:180-188)
    const double kMachineEps = 2.220446049250313e-16;  // np.finfo(float64).eps, commons.py:1814
    P.scale = 2 / c->p.boxsize * (double)(ewald_gridsize - 1) * (1 - kMachineEps);
    P.inv_box2 = 1 / (c->p.boxsize * c->p.boxsize);
    P.softening = softening;
    P.factor = factor;
    P.ewald = ewald_grid;
    P.gs = ewald_gridsize;
    P.kernel = kernel;
    P.same = same;
    P.factors = factors;
    P.rung = rung;
    P.rung_jumped = rung_jumped;
    P.lowest_active = lowest_active;
    hipLaunchKernelGGL(k_pp_kick, dim3((unsigned)((n_r + 255) / 256)), dim3(256), 0, c->stream,
                       pos_r, n_r, dmom_r, pos_s, n_s, P);
    CG_LAUNCH_CHECK();
    return 0;
}
