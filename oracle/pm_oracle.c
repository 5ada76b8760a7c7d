/*
 * oracle/pm_oracle.c — CPU restatement of the reference's PM gravity path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / CPU
 * baseline.  The product (concept_amd/, libconcept_gpu.so) never links,
 * loads or calls it.
 *
 * Parity status: PINNED.  oracle/oracle.py drives these functions and
 * tests/test_oracle_golden.py checks every intermediate against the golden
 * vectors in tests/golden/*.npz, which were produced by importing the
 * reference's pure-Python path (tests/golden/make_golden.py).
 *
 * Every function cites the reference lines (under /root/reference/src) it
 * restates.  Plain IEEE double; build with -ffp-contract=off (no FMA
 * contraction) so the operation order below is the operation order executed.
 * FFTs are not here: oracle.py uses numpy's pocketfft exactly as the
 * reference's pure-Python fft() does (mesh.py:4035-4143).
 */
#include <math.h>
#include <time.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t i64;

/* The `omp parallel for` pragmas below are active only in the all-cores timing build
 * (liboracle_omp.so, -fopenmp): the parity build ignores them and runs the reference's serial
 * loops in the reference's order.  With threads the deposit is slab-privatised (deposit_colored;
 * `omp atomic` only as the small-case fallback), i.e. cell sums are added in
 * scheduling order — a CPU baseline for bench.py, not a parity instrument. */
int orc_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* set_weights_CIC, mesh.py:5319-5324 */
static inline i64 set_weights_cic(double x, double *w) {
    i64 index = (i64)x; /* int(x), x >= 0 here */
    double dist = x - (double)index;
    w[0] = 1 - dist;
    w[1] = dist;
    return index;
}

/*
 * A1  CIC deposit.  interpolate_particles (mesh.py:1512-1636) with the inlined
 * particle_interpolation_loop_CIC (mesh.py:5101-5155).
 * offset[d] and scale are computed by the caller exactly as mesh.py:1577-1589
 * and the R[(1/cellsize)*(1 - machine_eps)] constant of mesh.py:1604.
 * grid is the ghosted domain grid double[size_i][size_j][size_k].
 * idx_out (nullable): the three set_weights_CIC indices per particle.
 */
#ifdef _OPENMP
#include <stdlib.h>
/* All-cores timing build only: the deposit without atomics.  The reference runs one MPI rank per
 * core, each depositing into its own ghosted domain grid (no write sharing at all,
 * mesh.py:1512-1636 + communicate_ghosts '+='); the OpenMP counterpart is a slab decomposition
 * along i with two colours: particles are binned by the slab of their lower cell (counting sort
 * of indices), then all even slabs are deposited concurrently, then all odd ones — a slab writes
 * the layers [s*W, (s+1)*W], so slabs of one colour never touch the same cell.  Same arithmetic
 * per particle as the loop below; only the order of additions to a cell differs. */
static int deposit_colored(const double *pos, i64 N, double *grid, i64 size_j, i64 size_k,
                           const double *offset, double scale, double contribution) {
    const int T = omp_get_max_threads();
    if (T < 2 || N < 4096) return 0;
    i64 imax = 0;
#pragma omp parallel for schedule(static) reduction(max : imax)
    for (i64 p = 0; p < N; p++) {
        i64 ii = (i64)((pos[3 * p + 0] - offset[0]) * scale);
        if (ii > imax) imax = ii;
    }
    const i64 nlayers = imax + 1;
    i64 S = 4 * (i64)T;  /* slabs: a few per thread and colour, for balance */
    if (S > nlayers) S = nlayers;
    if (S < 2) return 0;
    const i64 W = (nlayers + S - 1) / S;
    S = (nlayers + W - 1) / W;
    i64 *count = (i64 *)calloc((size_t)(S + 1) * (size_t)T, sizeof(i64));
    i64 *start = (i64 *)malloc((size_t)(S + 1) * sizeof(i64));
    int32_t *order = (int32_t *)malloc((size_t)N * sizeof(int32_t));
    if (!count || !start || !order || N > 2147483647) {
        free(count); free(start); free(order);
        return 0;
    }
#pragma omp parallel
    {
        const int t = omp_get_thread_num();
        i64 *mine = count + (size_t)t * (size_t)(S + 1);
#pragma omp for schedule(static)
        for (i64 p = 0; p < N; p++) mine[(i64)((pos[3 * p + 0] - offset[0]) * scale) / W]++;
#pragma omp single
        {
            i64 run = 0;
            for (i64 s = 0; s < S; s++) {
                start[s] = run;
                for (int u = 0; u < T; u++) {
                    i64 c = count[(size_t)u * (size_t)(S + 1) + s];
                    count[(size_t)u * (size_t)(S + 1) + s] = run;
                    run += c;
                }
            }
            start[S] = run;
        }
#pragma omp for schedule(static)
        for (i64 p = 0; p < N; p++)
            order[mine[(i64)((pos[3 * p + 0] - offset[0]) * scale) / W]++] = (int32_t)p;
        for (int colour = 0; colour < 2; colour++) {
#pragma omp for schedule(dynamic, 1)
            for (i64 s = colour; s < S; s += 2) {
                for (i64 q = start[s]; q < start[s + 1]; q++) {
                    const i64 p = order[q];
                    double wx[2], wy[2], wz[2];
                    i64 ii = set_weights_cic((pos[3 * p + 0] - offset[0]) * scale, wx);
                    i64 jj = set_weights_cic((pos[3 * p + 1] - offset[1]) * scale, wy);
                    i64 kk = set_weights_cic((pos[3 * p + 2] - offset[2]) * scale, wz);
                    i64 index_i = ((ii - 1) * size_j + (jj - 1)) * size_k + kk - 1;
                    for (int i = 0; i < 2; i++) {
                        double weight_i = wx[i] * contribution;
                        index_i += size_j * size_k;
                        i64 index_j = index_i;
                        for (int j = 0; j < 2; j++) {
                            index_j += size_k;
                            double wij = weight_i * wy[j];
                            grid[index_j + 1] += wij * wz[0];
                            grid[index_j + 2] += wij * wz[1];
                        }
                    }
                }
            }
        }
    }
    free(count); free(start); free(order);
    return 1;
}
#endif

void orc_cic_deposit(const double *pos, i64 N, double *grid, i64 size_j, i64 size_k,
                     const double *offset, double scale, double contribution, i64 *idx_out) {
#ifdef _OPENMP
    if (!idx_out && deposit_colored(pos, N, grid, size_j, size_k, offset, scale, contribution))
        return;
#endif
#pragma omp parallel for schedule(static)
    for (i64 p = 0; p < N; p++) {
        double wx[2], wy[2], wz[2];
        double x = (pos[3 * p + 0] - offset[0]) * scale;
        double y = (pos[3 * p + 1] - offset[1]) * scale;
        double z = (pos[3 * p + 2] - offset[2]) * scale;
        i64 ii = set_weights_cic(x, wx);
        i64 jj = set_weights_cic(y, wy);
        i64 kk = set_weights_cic(z, wz);
        if (idx_out) {
            idx_out[3 * p + 0] = ii;
            idx_out[3 * p + 1] = jj;
            idx_out[3 * p + 2] = kk;
        }
        /* mesh.py:5138-5155 */
        i64 index_i = ((ii - 1) * size_j + (jj - 1)) * size_k + kk - 1;
        for (int i = 0; i < 2; i++) {
            double weight_i = wx[i];
            weight_i *= contribution; /* apply_factor=True, mesh.py:5145-5146 */
            index_i += size_j * size_k;
            i64 index_j = index_i;
            for (int j = 0; j < 2; j++) {
                index_j += size_k;
                i64 index = index_j;
                double wij = weight_i * wy[j];
                for (int k = 0; k < 2; k++) {
                    index += 1;
#pragma omp atomic
                    grid[index] += wij * wz[k];
                }
            }
        }
    }
}

/*
 * A10  CIC gather + kick of one momentum component.
 * interpolate_domaingrid_to_particles (mesh.py:376-459): value accumulated
 * over the 8 cells in i,j,k order with weight (wx[i]*wy[j])*wz[k], then
 * value *= factor (if factor != 1), then mom[3p+dim] += value.
 */
void orc_cic_gather_kick(const double *grid, i64 size_j, i64 size_k, const double *pos,
                         double *mom, i64 N, int dim, const double *offset, double scale,
                         double factor, i64 *idx_out) {
#pragma omp parallel for schedule(static)
    for (i64 p = 0; p < N; p++) {
        double wx[2], wy[2], wz[2];
        double x = (pos[3 * p + 0] - offset[0]) * scale;
        double y = (pos[3 * p + 1] - offset[1]) * scale;
        double z = (pos[3 * p + 2] - offset[2]) * scale;
        i64 ii = set_weights_cic(x, wx);
        i64 jj = set_weights_cic(y, wy);
        i64 kk = set_weights_cic(z, wz);
        if (idx_out) {
            idx_out[3 * p + 0] = ii;
            idx_out[3 * p + 1] = jj;
            idx_out[3 * p + 2] = kk;
        }
        double value = 0;
        i64 index_i = ((ii - 1) * size_j + (jj - 1)) * size_k + kk - 1;
        for (int i = 0; i < 2; i++) {
            double weight_i = wx[i];
            index_i += size_j * size_k;
            i64 index_j = index_i;
            for (int j = 0; j < 2; j++) {
                index_j += size_k;
                i64 index = index_j;
                double wij = weight_i * wy[j];
                for (int k = 0; k < 2; k++) {
                    index += 1;
                    value += grid[index] * (wij * wz[k]);
                }
            }
        }
        if (factor != 1) value *= factor;
        mom[3 * p + dim] += value;
    }
}

/*
 * A2 / A8  communicate_ghosts on ONE rank (communication.py:563-660): all 26
 * neighbours are the domain itself.  op_add=1: '+=' (ghost values are added
 * onto the interior cells of the periodic neighbour); op_add=0: '=' (reverse
 * direction: ghosts are assigned from the opposite interior cells).
 * The 26 blocks are visited in the reference's i,j,k order so that cells
 * receiving several contributions are summed in the same order.
 */
static void block_range(int d, i64 n, i64 g, i64 *sb, i64 *se, i64 *rb, i64 *re) {
    if (d == -1) { *sb = 0; *se = g; *rb = n - 2 * g; *re = n - g; }
    else if (d == 0) { *sb = g; *se = n - g; *rb = g; *re = n - g; }
    else { *sb = n - g; *se = n; *rb = g; *re = 2 * g; }
}
void orc_communicate_ghosts(double *grid, i64 ni, i64 nj, i64 nk, i64 g, int op_add) {
    for (int i = -1; i < 2; i++) for (int j = -1; j < 2; j++) for (int k = -1; k < 2; k++) {
        if (i == 0 && j == 0 && k == 0) continue;
        i64 sbi, sei, rbi, rei, sbj, sej, rbj, rej, sbk, sek, rbk, rek;
        block_range(i, ni, g, &sbi, &sei, &rbi, &rei);
        block_range(j, nj, g, &sbj, &sej, &rbj, &rej);
        block_range(k, nk, g, &sbk, &sek, &rbk, &rek);
        for (i64 a = 0; a < sei - sbi; a++) for (i64 b = 0; b < sej - sbj; b++)
            for (i64 c = 0; c < sek - sbk; c++) {
                i64 s = ((sbi + a) * nj + (sbj + b)) * nk + (sbk + c);
                i64 r = ((rbi + a) * nj + (rbj + b)) * nk + (rbk + c);
                if (op_add) grid[r] += grid[s];
                else grid[s] = grid[r];
            }
    }
}

/*
 * A3  slab_decompose on one rank (mesh.py:2284-2411): interior of the ghosted
 * domain grid -> x-slab double[N][N][N+2], padding zeroed (mesh.py:2338-2339).
 */
void orc_slab_decompose(const double *grid, i64 N, i64 g, double *slab) {
    i64 n = N + 2 * g, pad = N + 2;
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < N; i++) for (i64 j = 0; j < N; j++) {
        const double *src = grid + ((i + g) * n + (j + g)) * n + g;
        double *dst = slab + (i * N + j) * pad;
        memcpy(dst, src, sizeof(double) * N);
        dst[N] = 0; dst[N + 1] = 0;
    }
}
/* The layout steps around the FFT for the threaded timing path (oracle.fft_forward /
 * fft_backward with workers): the (i, j) transposition of FFTW-MPI's transposed output
 * (fft.c:240-257) on rows of `row` doubles, and the copy of the real result into the padded
 * slab.  Same data movement as the numpy expressions of the parity path, over all threads. */
void orc_transpose_ij(const double *in, double *out, i64 N, i64 row) {
#pragma omp parallel for schedule(static)
    for (i64 j = 0; j < N; j++) for (i64 i = 0; i < N; i++)
        memcpy(out + (j * N + i) * row, in + (i * N + j) * row, sizeof(double) * row);
}
void orc_copy_pad(const double *in, double *out, i64 N) {
    i64 pad = N + 2;
#pragma omp parallel for schedule(static)
    for (i64 r = 0; r < N * N; r++) {
        memcpy(out + r * pad, in + r * N, sizeof(double) * N);
        out[r * pad + N] = 0; out[r * pad + N + 1] = 0;
    }
}
/* A8  domain_decompose on one rank (mesh.py:2138-2244), interior only */
void orc_domain_decompose(const double *slab, i64 N, i64 g, double *grid) {
    i64 n = N + 2 * g, pad = N + 2;
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < N; i++) for (i64 j = 0; j < N; j++)
        memcpy(grid + ((i + g) * n + (j + g)) * n + g, slab + (i * N + j) * pad,
               sizeof(double) * N);
}

/*
 * A5  nullify_modes 'nyquist' (mesh.py:3591-3622) on the transposed slab
 * double[j][i][N+2] (one rank: j is global).
 */
void orc_nullify_nyquist(double *slab, i64 N) {
    i64 nyq = N / 2, pad = N + 2;
#pragma omp parallel for schedule(static)
    for (i64 j = 0; j < N; j++) for (i64 i = 0; i < N; i++) {
        double *row = slab + (j * N + i) * pad;
        if (i == nyq || j == nyq) memset(row, 0, sizeof(double) * pad);
        /* note: the reference leaves k = N (kk = nyquist) of those rows to the
           third loop, which zeroes it anyway */
        row[2 * nyq] = 0; row[2 * nyq + 1] = 0;
    }
}

/*
 * A6  Poisson / deconvolution kernel: fourier_loop (mesh.py:2615-2890) inlined
 * into particle_mesh (interactions.py:2092-2118), one rank, slab in the
 * transposed layout double[j][i][N+2].  Nyquist planes are skipped (they
 * were nullified before), the origin is skipped and then nullified.
 *   C = -boxsize**2*G_Newton/pi               (interactions.py:2105)
 *   E = -(2*pi/boxsize*scale)**2 or unused    (interactions.py:2112)
 */
void orc_kspace_poisson(double *slab, i64 N, int deconv_order, double C, int long_range,
                        double E, double machine_eps) {
    const double pi = 3.141592653589793; /* float(np.pi), commons.py:1816 */
    double pi_over_n = pi / (double)N;
    i64 nyq = N / 2, pad = N + 2;
#pragma omp parallel for schedule(static)
    for (i64 j = 0; j < N; j++) {
        if (j == nyq) continue;
        i64 kj = j - (j >= nyq ? N : 0);
        double dj_n = (double)kj * pi_over_n + machine_eps; /* mesh.py:2775 */
        double dj_d = sin(dj_n);
        for (i64 i = 0; i < N; i++) {
            if (i == nyq) continue;
            i64 ki = i - (i >= nyq ? N : 0);
            double di_n = (double)ki * pi_over_n + machine_eps; /* mesh.py:2795 */
            double di_d = sin(di_n);
            double dij_n = di_n * dj_n; /* mesh.py:2797 */
            double dij_d = di_d * dj_d;
            double *row = slab + (j * N + i) * pad;
            for (i64 kk = (i == 0 && j == 0) ? 1 : 0; kk < nyq; kk++) {
                double factor = 1;
                if (deconv_order) {
                    double dk_n = (double)kk * pi_over_n + machine_eps;
                    double dk_d = sin(dk_n);
                    factor = (dij_n * dk_n) / (dij_d * dk_d); /* mesh.py:2850-2853 */
                    factor = pow(factor, (double)deconv_order); /* mesh.py:2855 */
                }
                factor *= 1.0; /* 1/len(lattice), 'sc' lattice, mesh.py:2856 */
                i64 k2 = (kj * kj + ki * ki) + kk * kk; /* interactions.py:2096 */
                if (!long_range) factor *= C / (double)k2; /* interactions.py:2105 */
                else factor *= C / (double)k2 * exp((double)k2 * E); /* :2110-2113 */
                row[2 * kk] *= factor;
                row[2 * kk + 1] *= factor;
            }
        }
    }
    slab[0] = 0; slab[1] = 0; /* nullify_modes('origin'), mesh.py:3585-3590 */
}

/*
 * A9  diff_domaingrid (mesh.py:4874-5030): symmetric finite difference of the
 * ghosted grid along dim, interior only; orders 2 and 4 (1, 6, 8 are not on
 * the path's defaults and are rejected by the host).
 */
void orc_diff_domaingrid(const double *grid, double *out, i64 ni, i64 nj, i64 nk, i64 g, int dim,
                         int order, double dx) {
    /* diff_domaingrid, mesh.py:4874-5030: symmetric differences of order 2, 4, 6, 8 with the
     * coefficients written as the reference writes them, and the one-sided order 1
     * (direction = 'forward', the default: mesh.py:4903-4910) */
    i64 step = dim == 0 ? nj * nk : (dim == 1 ? nk : 1);
    double c1 = 1 / dx, c2 = (1.0 / 2) / dx, c4a = (2.0 / 3) / dx, c4b = (1.0 / 12) / dx;
    double c6a = (3.0 / 4) / dx, c6b = (3.0 / 20) / dx, c6c = (1.0 / 60) / dx;
    double c8a = (4.0 / 5) / dx, c8b = (1.0 / 5) / dx, c8c = (4.0 / 105) / dx,
           c8d = (1.0 / 280) / dx;
#pragma omp parallel for schedule(static)
    for (i64 i = g; i < ni - g; i++) for (i64 j = g; j < nj - g; j++)
        for (i64 k = g; k < nk - g; k++) {
            i64 ix = (i * nj + j) * nk + k;
            if (order == 1)
                out[ix] = c1 * (grid[ix + step] - grid[ix]);
            else if (order == 2)
                out[ix] = c2 * (grid[ix + step] - grid[ix - step]);
            else if (order == 4)
                out[ix] = c4a * (grid[ix + step] - grid[ix - step])
                        - c4b * (grid[ix + 2 * step] - grid[ix - 2 * step]);
            else if (order == 6)
                out[ix] = (c6a * (grid[ix + step] - grid[ix - step])
                           - c6b * (grid[ix + 2 * step] - grid[ix - 2 * step]))
                        + c6c * (grid[ix + 3 * step] - grid[ix - 3 * step]);
            else
                out[ix] = ((c8a * (grid[ix + step] - grid[ix - step])
                            - c8b * (grid[ix + 2 * step] - grid[ix - 2 * step]))
                           + c8c * (grid[ix + 3 * step] - grid[ix - 3 * step]))
                        - c8d * (grid[ix + 4 * step] - grid[ix - 4 * step]);
        }
}

/*
 * A11  Component.drift (species.py:2179-2199) with the pure-Python mod()
 * (commons.py:5103-5110): np.mod (floored modulo) then x == length -> 0.
 */
static inline double np_mod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0) { if ((b < 0) != (m < 0)) m += b; }
    else m = copysign(0.0, b);
    return m;
}
/* First touch for the timing baseline (bench.py): dst[i] = fmod(block[i % nb] + (i / nb) * shift,
 * period) written by the threads that will later stream it (static schedule, as every particle
 * loop here), so that the pages of a large array spread over the NUMA domains instead of
 * landing where a single generating thread ran.  Not part of any parity path. */
void orc_fill_tiled(double *dst, i64 n, const double *block, i64 nb, double shift, double period) {
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; i++) {
        double v = block[i % nb] + (double)(i / nb) * shift;
        if (period > 0) {
            v = fmod(v, period);
            if (v < 0) v += period;
            if (v >= period) v = 0;
        }
        dst[i] = v;
    }
}
/* STREAM triad a = b + s c on arrays of n doubles touched first by the threads that sweep
 * them (static schedule): what this host's memory gives the thread count in use — the ceiling
 * bench.py prints beside the port's phases.  Returns the best of `reps` sweeps in seconds. */
double orc_stream_triad(double *a, double *b, double *c, i64 n, int reps) {
    double best = 1e300;
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; i++) {
        a[i] = 0.0;
        b[i] = 1.0;
        c[i] = 2.0;
    }
    for (int r = 0; r < reps; r++) {
#ifdef _OPENMP
        const double t0 = omp_get_wtime();
#else
        const clock_t c0 = clock();
#endif
#pragma omp parallel for schedule(static)
        for (i64 i = 0; i < n; i++) a[i] = b[i] + 3.0 * c[i];
#ifdef _OPENMP
        const double dt = omp_get_wtime() - t0;
#else
        const double dt = (double)(clock() - c0) / (double)CLOCKS_PER_SEC;
#endif
        if (dt < best) best = dt;
    }
    return best;
}
void orc_zero(double *dst, i64 n) {
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; i++) dst[i] = 0.0;
}

void orc_drift(double *pos, const double *mom, i64 n3, double dt_over_mass, double boxsize) {
#pragma omp parallel for schedule(static)
    for (i64 r = 0; r < n3; r++) {
        double x = np_mod(pos[r] + mom[r] * dt_over_mass, boxsize);
        if (x == boxsize) x = 0;
        pos[r] = x;
    }
}

/*
 * fourier_operate (mesh.py:3327-3400) and the equal-size branch of copy_modes
 * (mesh.py:1038-1092), one rank, transposed slab double[j][i][N+2]: every mode
 * off the Nyquist planes (fourier_loop, mesh.py:2615-2890, covers j, i in
 * [0, nyq) U (nyq, N) and kk in [0, nyq)) of `from` is
 *   rotated by theta = (ki*A + kj*B) + kk*Cc, A = -2*pi/N*shift[0] ...  (shift != 0)
 *   differentiated: factor *= k_fundamental*k_dim, (re, im) = (-im, re)  (diff_dim >= 0)
 *   scaled by factor = deconvolution**order * (1/nlattice)
 * and stored to (op_add = 0) or added to (op_add = 1) `onto`; from == onto with
 * op_add = 0 is the in-place fourier_operate.  Both early exits of the reference
 * (nothing to do; pure 1/nlattice scaling of every element) are the caller's.
 */
void orc_fourier_operate(const double *from, double *onto, i64 N, int deconv_order, int nlattice,
                         const double *shift, int diff_dim, double k_fundamental, int op_add,
                         double machine_eps) {
    const double pi = 3.141592653589793;
    double pi_over_n = pi / (double)N;
    double inv_lat = 1.0 / (double)nlattice;
    int shifted = shift && (shift[0] != 0 || shift[1] != 0 || shift[2] != 0);
    double A = 0, B = 0, Cc = 0;
    if (shifted) {
        A = -2 * pi / (double)N * shift[0];
        B = -2 * pi / (double)N * shift[1];
        Cc = -2 * pi / (double)N * shift[2];
    }
    i64 nyq = N / 2, pad = N + 2;
    for (i64 j = 0; j < N; j++) {
        if (j == nyq) continue;
        i64 kj = j - (j >= nyq ? N : 0);
        double dj_n = (double)kj * pi_over_n + machine_eps;
        double dj_d = sin(dj_n);
        for (i64 i = 0; i < N; i++) {
            if (i == nyq) continue;
            i64 ki = i - (i >= nyq ? N : 0);
            double di_n = (double)ki * pi_over_n + machine_eps;
            double di_d = sin(di_n);
            double dij_n = di_n * dj_n, dij_d = di_d * dj_d;
            const double *src = from + (j * N + i) * pad;
            double *dst = onto + (j * N + i) * pad;
            for (i64 kk = 0; kk < nyq; kk++) {
                double factor = 1;
                if (deconv_order) {
                    double dk_n = (double)kk * pi_over_n + machine_eps;
                    double dk_d = sin(dk_n);
                    factor = (dij_n * dk_n) / (dij_d * dk_d);
                    factor = pow(factor, (double)deconv_order);
                }
                factor *= inv_lat;
                double re = src[2 * kk], im = src[2 * kk + 1];
                if (shifted) {
                    double theta = ((double)ki * A + (double)kj * B) + (double)kk * Cc;
                    double c = cos(theta), s = sin(theta);
                    double re2 = re * c - im * s, im2 = re * s + im * c;
                    re = re2; im = im2;
                }
                if (diff_dim >= 0) {
                    i64 kl = diff_dim == 0 ? ki : (diff_dim == 1 ? kj : kk);
                    factor *= k_fundamental * (double)kl;
                    double t = re; re = -im; im = t;
                }
                re *= factor;
                im *= factor;
                if (op_add) { dst[2 * kk] += re; dst[2 * kk + 1] += im; }
                else { dst[2 * kk] = re; dst[2 * kk + 1] = im; }
            }
        }
    }
}

/*
 * copy_modes between slabs of DIFFERENT grid sizes (mesh.py:1094-1326), one rank.
 * Every mode of the smaller grid off its Nyquist planes is read from `from` and stored
 * to / added onto `onto` (both transposed slabs double[j][i][N+2] of their own size) as
 *   theta_total = (pi/N_onto - pi/N_from)*((ki + kj) + kk)  [+ theta of the lattice]
 *   (re, im) <- (factor*(re*cos - im*sin), factor*(re*sin + im*cos))
 * with factor and the lattice phase from fourier_loop(gridsize_small,
 * gridsize_corrections = N_from, ...) (mesh.py:1245-1250): deconvolution and shift use
 * the grid size the data was interpolated on.
 */
void orc_copy_modes(const double *from, i64 N_from, double *onto, i64 N_onto, int deconv_order,
                    int nlattice, const double *shift, int op_add, double machine_eps) {
    const double pi = 3.141592653589793;
    i64 N_small = N_from < N_onto ? N_from : N_onto;
    double pi_over_n = pi / (double)N_from; /* gridsize_corrections = gridsize_from */
    double inv_lat = 1.0 / (double)nlattice;
    int shifted = shift && (shift[0] != 0 || shift[1] != 0 || shift[2] != 0);
    double A = 0, B = 0, Cc = 0;
    if (shifted) {
        A = -2 * pi / (double)N_from * shift[0];
        B = -2 * pi / (double)N_from * shift[1];
        Cc = -2 * pi / (double)N_from * shift[2];
    }
    double dtheta = pi / (double)N_onto - pi / (double)N_from;
    i64 nyq = N_small / 2;
    i64 pad_f = N_from + 2, pad_o = N_onto + 2;
    for (i64 js = 0; js < N_small; js++) {
        if (js == nyq) continue;
        i64 kj = js - (js >= nyq ? N_small : 0);
        double dj_n = (double)kj * pi_over_n + machine_eps;
        double dj_d = sin(dj_n);
        i64 jf = kj + (kj < 0 ? N_from : 0), jo = kj + (kj < 0 ? N_onto : 0);
        for (i64 is = 0; is < N_small; is++) {
            if (is == nyq) continue;
            i64 ki = is - (is >= nyq ? N_small : 0);
            double di_n = (double)ki * pi_over_n + machine_eps;
            double di_d = sin(di_n);
            double dij_n = di_n * dj_n, dij_d = di_d * dj_d;
            i64 i_f = ki + (ki < 0 ? N_from : 0), i_o = ki + (ki < 0 ? N_onto : 0);
            const double *src = from + (jf * N_from + i_f) * pad_f;
            double *dst = onto + (jo * N_onto + i_o) * pad_o;
            for (i64 kk = 0; kk < nyq; kk++) {
                double factor = 1;
                if (deconv_order) {
                    double dk_n = (double)kk * pi_over_n + machine_eps;
                    double dk_d = sin(dk_n);
                    factor = (dij_n * dk_n) / (dij_d * dk_d);
                    factor = pow(factor, (double)deconv_order);
                }
                factor *= inv_lat;
                double re = src[2 * kk], im = src[2 * kk + 1];
                double theta_total = dtheta * (double)((ki + kj) + kk);
                if (shifted) theta_total += ((double)ki * A + (double)kj * B) + (double)kk * Cc;
                double c = cos(theta_total), s = sin(theta_total);
                double re2 = factor * (re * c - im * s), im2 = factor * (re * s + im * c);
                if (op_add) { dst[2 * kk] += re2; dst[2 * kk + 1] += im2; }
                else { dst[2 * kk] = re2; dst[2 * kk + 1] = im2; }
            }
        }
    }
}

/*
 * set_weights_{NGP,CIC,TSC,PCS} (mesh.py:5305-5394) and the particle interpolation
 * loops of every order (mesh.py:5052-5283): `order` points per dimension starting
 * at the returned index, visited i, j, k ascending; weight = (w_x[i]*w_y[j])*w_z[k]
 * with w_x[i] first multiplied by the contribution when depositing.
 * The reference's `tmp**2` is Python float pow (libm pow): pow() here.
 */
/* Python's dist**2 is libm pow() at run time; keep the compiler from folding pow(x, 2.0)
 * into x*x (which can differ from glibc's pow in the last bit) */
static volatile double exponent_two = 2.0, exponent_three = 3.0;
static i64 set_weights_order(double x, double *w, int order) {
    i64 index;
    double dist;
    if (order == 1) {
        index = (i64)(x + 0.5);
        w[0] = 1;
        return index;
    }
    if (order == 2) return set_weights_cic(x, w);
    if (order == 3) {
        index = (i64)(x + 0.5);
        dist = x - (double)index;
        index -= 1;
        double dist2 = pow(dist, exponent_two);
        double weight0 = 0.125 + 0.5 * (dist2 - dist);
        double weight1 = 0.75 - dist2;
        w[0] = weight0;
        w[1] = weight1;
        w[2] = 1 - weight0 - weight1;
        return index;
    }
    index = (i64)x;
    index -= 1;
    dist = x - (double)index;
    double tmp = 2 - dist;
    double tmp2 = pow(tmp, exponent_two);
    double tmp3 = tmp * tmp2;
    double weight0 = 1. / 6. * tmp3;
    double weight2 = 2. / 3. - tmp2 + 0.5 * tmp3;
    double weight3 = 1. / 6. * pow(dist - 1, exponent_three);
    w[0] = weight0;
    w[1] = 1 - weight0 - weight2 - weight3;
    w[2] = weight2;
    w[3] = weight3;
    return index;
}

void orc_interp_deposit(const double *pos, i64 N, double *grid, i64 size_j, i64 size_k,
                        const double *offset, double scale, double contribution, int order) {
    double wx[4], wy[4], wz[4];
    for (i64 p = 0; p < N; p++) {
        double x = (pos[3 * p + 0] - offset[0]) * scale;
        double y = (pos[3 * p + 1] - offset[1]) * scale;
        double z = (pos[3 * p + 2] - offset[2]) * scale;
        i64 ii = set_weights_order(x, wx, order);
        i64 jj = set_weights_order(y, wy, order);
        i64 kk = set_weights_order(z, wz, order);
        for (int i = 0; i < order; i++) {
            double weight_i = wx[i];
            weight_i *= contribution;
            if (order == 1) weight_i = contribution; /* NGP: weight = multiplier, mesh.py:5078 */
            for (int j = 0; j < order; j++) {
                double wij = weight_i * wy[j];
                for (int k = 0; k < order; k++) {
                    i64 index = ((ii + i) * size_j + (jj + j)) * size_k + (kk + k);
                    grid[index] += (order == 1) ? weight_i : wij * wz[k];
                }
            }
        }
    }
}

void orc_interp_gather(const double *grid, i64 size_j, i64 size_k, const double *pos,
                       double *mom, i64 N, int dim, const double *offset, double scale,
                       double factor, int order) {
    double wx[4], wy[4], wz[4];
    for (i64 p = 0; p < N; p++) {
        double x = (pos[3 * p + 0] - offset[0]) * scale;
        double y = (pos[3 * p + 1] - offset[1]) * scale;
        double z = (pos[3 * p + 2] - offset[2]) * scale;
        i64 ii = set_weights_order(x, wx, order);
        i64 jj = set_weights_order(y, wy, order);
        i64 kk = set_weights_order(z, wz, order);
        double value = 0;
        for (int i = 0; i < order; i++)
            for (int j = 0; j < order; j++) {
                double wij = wx[i] * wy[j];
                for (int k = 0; k < order; k++) {
                    i64 index = ((ii + i) * size_j + (jj + j)) * size_k + (kk + k);
                    double weight = (order == 1) ? 1.0 : wij * wz[k];
                    value += grid[index] * weight;
                }
            }
        if (factor != 1) value *= factor;
        mom[3 * p + dim] += value;
    }
}
