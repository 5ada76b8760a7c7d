// cg_context.hip — context, rocFFT plans, Poisson solve driver, debug fetch.
// MI355X / gfx950 only.  See include/concept_gpu.h for the reference lines
// each entry point replaces.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "cg_internal.h"
#include "cg_substep.h"

static thread_local std::string g_err;

void cg_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" const char *cg_last_error(void) { return g_err.c_str(); }
extern "C" int cg_abi_version(void) { return CG_ABI_VERSION; }

static const double kMachineEps = 2.220446049250313e-16;  // np.finfo(float64).eps, commons.py:1814
static const double kPi = 3.141592653589793;              // float(np.pi), commons.py:1816

// mesh.py:1577-1589 (deposit, -shift) and mesh.py:408-420 (gather, +shift); no
// interlacing lattice is built yet, so shift = 0 and only cellsize differs.
static CicGeom make_geom(double cellsize, int nghosts, int cell_centered, const double bgn[3]) {
    CicGeom g;
    for (int d = 0; d < 3; d++)
        g.off[d] = bgn[d] - (1 + kMachineEps) * (nghosts - 0.5 * cell_centered - 0) * cellsize;
    g.scale = (1 / cellsize) * (1 - kMachineEps);
    return g;
}
// the same with an interlacing lattice shift: sign = -1 for interpolate_particles
// (mesh.py:1577-1589), +1 for interpolate_domaingrid_to_particles (mesh.py:409-420)
static CicGeom make_geom_shift(double cellsize, int nghosts, int cell_centered,
                               const double *shift, int sign) {
    CicGeom g;
    for (int d = 0; d < 3; d++) {
        double sh = shift ? shift[d] : 0.0;
        g.off[d] = 0.0 - (1 + kMachineEps) * (nghosts - 0.5 * cell_centered + sign * sh) * cellsize;
    }
    g.scale = (1 / cellsize) * (1 - kMachineEps);
    return g;
}

static int make_plans(cg_ctx *c) {
    // In-place 3-D R2C / C2R on the padded layout FFTW uses (fft.c:34-73):
    // real double[N][N][N+2] <-> hermitian complex[N][N][N/2+1], unnormalised
    // both ways (mesh.py:4015-4022).  rocFFT lengths are fastest-first.
    const size_t N = (size_t)c->N;
    size_t lengths[3] = {N, N, N};
    // row pitch c->pad >= N + 2 doubles (a multiple of 16 doubles when N is, so that
    // rows and 16-cell tile rows start on 128-byte lines)
    const size_t P = (size_t)c->pad;
    size_t rstr[3] = {1, P, P * (size_t)c->ny};
    size_t cstr[3] = {1, P / 2, (P / 2) * (size_t)c->ny};
    size_t off[1] = {0};
    rocfft_plan_description d = nullptr;
    CG_FFT(rocfft_plan_description_create(&d));
    CG_FFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real,
                                                   rocfft_array_type_hermitian_interleaved, off,
                                                   off, 3, rstr, P * N * N, 3, cstr,
                                                   (P / 2) * N * N));
    CG_FFT(rocfft_plan_create(&c->plan_fwd, rocfft_placement_inplace,
                              rocfft_transform_type_real_forward, rocfft_precision_double, 3,
                              lengths, 1, d));
    CG_FFT(rocfft_plan_description_destroy(d));
    CG_FFT(rocfft_plan_description_create(&d));
    CG_FFT(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved,
                                                   rocfft_array_type_real, off, off, 3, cstr,
                                                   (P / 2) * N * N, 3, rstr, P * N * N));
    CG_FFT(rocfft_plan_create(&c->plan_bwd, rocfft_placement_inplace,
                              rocfft_transform_type_real_inverse, rocfft_precision_double, 3,
                              lengths, 1, d));
    CG_FFT(rocfft_plan_description_destroy(d));
    size_t wf = 0, wb = 0;
    CG_FFT(rocfft_plan_get_work_buffer_size(c->plan_fwd, &wf));
    CG_FFT(rocfft_plan_get_work_buffer_size(c->plan_bwd, &wb));
    c->fft_work_bytes = wf > wb ? wf : wb;
    if (c->fft_work_bytes) {
        CG_HIP(hipMalloc(&c->fft_work, c->fft_work_bytes));
        c->device_bytes += (i64)c->fft_work_bytes;
    }
    CG_FFT(rocfft_execution_info_create(&c->info_fwd));
    CG_FFT(rocfft_execution_info_create(&c->info_bwd));
    if (c->fft_work_bytes) {
        CG_FFT(rocfft_execution_info_set_work_buffer(c->info_fwd, c->fft_work, c->fft_work_bytes));
        CG_FFT(rocfft_execution_info_set_work_buffer(c->info_bwd, c->fft_work, c->fft_work_bytes));
    }
    return 0;
}


// ---------------------------------------------------------------------------
// x-slab domains on the rocFFT backend: grids that are not a power of two (the reference
// accepts any size divisible by the decomposition, communication.py:692-741,
// mesh.py:1898-1905; its own tests use 24 and 36).  Same stages and the same transpose-buffer
// layout as the hand-written path (cg_fft.hip fft_dist): 2-D real transforms of the owned
// layers, rows packed by destination domain, all-to-all (the caller's), 1-D transforms along x
// on complex[N][JB + 1][cp], and back.  The pack / unpack is a pass of its own here (the
// hand-written y pass writes the blocked layout directly).
// ---------------------------------------------------------------------------
#include <map>
struct DistPlans {
    std::map<i64, std::pair<rocfft_plan, rocfft_plan>> layers2d;  // by number of layers
    rocfft_plan x_fwd = nullptr, x_bwd = nullptr;
    rocfft_execution_info info = nullptr;
    void *work = nullptr;
    size_t work_bytes = 0;
};

static int dist_work(cg_ctx *c, rocfft_plan plan) {
    DistPlans *d = c->dist_plans;
    size_t w = 0;
    CG_FFT(rocfft_plan_get_work_buffer_size(plan, &w));
    if (w > d->work_bytes) {
        CG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(d->work);
        d->work = nullptr;
        CG_HIP(hipMalloc(&d->work, w));
        d->work_bytes = w;
    }
    if (!d->info) CG_FFT(rocfft_execution_info_create(&d->info));
    if (d->work_bytes)
        CG_FFT(rocfft_execution_info_set_work_buffer(d->info, d->work, d->work_bytes));
    CG_FFT(rocfft_execution_info_set_stream(d->info, c->stream));
    return 0;
}

static int dist_plans_2d(cg_ctx *c, i64 nlayers, rocfft_plan *fwd, rocfft_plan *bwd) {
    if (!c->dist_plans) c->dist_plans = new DistPlans();
    auto &m = c->dist_plans->layers2d;
    auto it = m.find(nlayers);
    if (it == m.end()) {
        const size_t N = (size_t)c->N, P = (size_t)c->pad;
        size_t lengths[2] = {N, N};  // fastest first: z, y
        size_t rstr[2] = {1, P}, cstr[2] = {1, P / 2}, off[1] = {0};
        rocfft_plan pf = nullptr, pb = nullptr;
        rocfft_plan_description d = nullptr;
        CG_FFT(rocfft_plan_description_create(&d));
        CG_FFT(rocfft_plan_description_set_data_layout(
            d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, off, off, 2, rstr,
            P * (size_t)c->ny, 2, cstr, (P / 2) * (size_t)c->ny));
        CG_FFT(rocfft_plan_create(&pf, rocfft_placement_inplace, rocfft_transform_type_real_forward,
                                  rocfft_precision_double, 2, lengths, (size_t)nlayers, d));
        CG_FFT(rocfft_plan_description_destroy(d));
        CG_FFT(rocfft_plan_description_create(&d));
        CG_FFT(rocfft_plan_description_set_data_layout(
            d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, off, off, 2, cstr,
            (P / 2) * (size_t)c->ny, 2, rstr, P * (size_t)c->ny));
        CG_FFT(rocfft_plan_create(&pb, rocfft_placement_inplace, rocfft_transform_type_real_inverse,
                                  rocfft_precision_double, 2, lengths, (size_t)nlayers, d));
        CG_FFT(rocfft_plan_description_destroy(d));
        it = m.emplace(nlayers, std::make_pair(pf, pb)).first;
    }
    *fwd = it->second.first;
    *bwd = it->second.second;
    return 0;
}

static int dist_plans_x(cg_ctx *c) {
    if (!c->dist_plans) c->dist_plans = new DistPlans();
    DistPlans *dp = c->dist_plans;
    if (dp->x_fwd) return 0;
    const size_t N = (size_t)c->N, cp = (size_t)c->pad / 2, JB = N / (size_t)c->p.nprocs;
    size_t lengths[1] = {N}, str[1] = {(JB + 1) * cp}, off[1] = {0};
    for (int inv = 0; inv < 2; inv++) {
        rocfft_plan_description d = nullptr;
        CG_FFT(rocfft_plan_description_create(&d));
        // one transform per (row of the own block, kk): consecutive complex numbers
        CG_FFT(rocfft_plan_description_set_data_layout(
            d, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, off,
            off, 1, str, 1, 1, str, 1));
        CG_FFT(rocfft_plan_create(inv ? &dp->x_bwd : &dp->x_fwd, rocfft_placement_inplace,
                                  inv ? rocfft_transform_type_complex_inverse
                                      : rocfft_transform_type_complex_forward,
                                  rocfft_precision_double, 1, lengths, JB * cp, d));
        CG_FFT(rocfft_plan_description_destroy(d));
    }
    return 0;
}

// rows j of the layers [layer0, layer0 + nlayers) between the slab's Fourier layout
// complex[layer][ny][cp] and the transpose buffer blocked by destination domain
// buf[q = j / JB][layer][j - q JB][cp] (JB + 1 rows per layer)
template <bool PACK>
__global__ __launch_bounds__(256) void k_dist_rows(double2 *__restrict__ slab,
                                                   double2 *__restrict__ buf, i64 N, i64 ny,
                                                   i64 cp, i64 nxl, i64 JB, i64 layer0) {
    const i64 row = blockIdx.x;  // (layer - layer0) * N + j
    const i64 l = layer0 + row / N, j = row % N, q = j / JB;
    double2 *a = slab + (l * ny + j) * cp;
    double2 *b = buf + ((q * nxl + l) * (JB + 1) + (j - q * JB)) * cp;
    for (i64 kk = threadIdx.x; kk < N / 2 + 1; kk += blockDim.x) {
        if (PACK) b[kk] = a[kk];
        else a[kk] = b[kk];
    }
}

static int dist_generic(cg_ctx *c, int what, double2 *buf, int deconv_order, double C,
                        int long_range, double E, i64 layer0, i64 nlayers) {
    const i64 cp = c->pad / 2, N = c->N, nxl = c->xmap.nxl, JB = N / c->p.nprocs;
    if (nlayers < 0) nlayers = nxl - layer0;
    double2 *m = (double2 *)c->mesh0;
    if (what == 0 || what == 1) {
        rocfft_plan pf, pb;
        if (dist_plans_2d(c, nlayers, &pf, &pb)) return 1;
        void *io[1] = {(void *)(m + layer0 * c->ny * cp)};
        if (what == 0) {
            if (dist_work(c, pf)) return 1;
            CG_FFT(rocfft_execute(pf, io, nullptr, c->dist_plans->info));
            hipLaunchKernelGGL((k_dist_rows<true>), dim3((unsigned)(nlayers * N)), dim3(256), 0,
                               c->stream, m, buf, N, c->ny, cp, nxl, JB, layer0);
        } else {
            hipLaunchKernelGGL((k_dist_rows<false>), dim3((unsigned)(nlayers * N)), dim3(256), 0,
                               c->stream, m, buf, N, c->ny, cp, nxl, JB, layer0);
            if (dist_work(c, pb)) return 1;
            CG_FFT(rocfft_execute(pb, io, nullptr, c->dist_plans->info));
        }
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) {
            cg_set_error("cg_dist_fft (rocFFT backend): pack / unpack launch failed: %s",
                         hipGetErrorString(e_));
            return 1;
        }
        return 0;
    }
    if (dist_plans_x(c)) return 1;
    void *io[1] = {(void *)buf};
    if (what == 2 || what == 3) {
        if (dist_work(c, c->dist_plans->x_fwd)) return 1;
        CG_FFT(rocfft_execute(c->dist_plans->x_fwd, io, nullptr, c->dist_plans->info));
    }
    if (what == 2) {
        // the Poisson / deconvolution kernel on the rows of this domain (the Fourier view of
        // cg_dist_bind_fourier, for the duration of the call on `buf`)
        double2 *four = c->four;
        const i64 f_si = c->f_si;
        const int f_j0 = c->f_j0, f_nj = c->f_nj;
        c->four = buf;
        c->f_si = (JB + 1) * cp;
        c->f_j0 = (int)(JB * c->p.rank);
        c->f_nj = (int)JB;
        int rc = cgk_kspace(c, deconv_order, C, long_range, E);
        c->four = four, c->f_si = f_si, c->f_j0 = f_j0, c->f_nj = f_nj;
        if (rc) return rc;
    }
    if (what == 2 || what == 4) {
        if (dist_work(c, c->dist_plans->x_bwd)) return 1;
        CG_FFT(rocfft_execute(c->dist_plans->x_bwd, io, nullptr, c->dist_plans->info));
    }
    return 0;
}

static void dist_plans_destroy(cg_ctx *c) {
    DistPlans *d = c->dist_plans;
    if (!d) return;
    for (auto &kv : d->layers2d) {
        rocfft_plan_destroy(kv.second.first);
        rocfft_plan_destroy(kv.second.second);
    }
    if (d->x_fwd) rocfft_plan_destroy(d->x_fwd);
    if (d->x_bwd) rocfft_plan_destroy(d->x_bwd);
    if (d->info) rocfft_execution_info_destroy(d->info);
    (void)hipFree(d->work);
    delete d;
    c->dist_plans = nullptr;
}

static bool g_rocfft_ready = false;

extern "C" int cg_create(const cg_params *p, cg_ctx **out) {
    CG_CHECK(p && out, "cg_create: null argument");
    *out = nullptr;
    CG_CHECK(p->gridsize >= 4 && p->gridsize % 2 == 0,
             "cg_create: gridsize %lld must be even and >= 4 (mesh.py:1898-1905)",
             (long long)p->gridsize);
    CG_CHECK(p->interp_order == 2, "cg_create: the tiled kernels of a context are CIC (interp_order 2), got %d; other orders go through cg_deposit / cg_gather_scalar",
             p->interp_order);
    CG_CHECK(p->nghosts >= 1 && p->nghosts <= 4, "cg_create: nghosts %d out of range", p->nghosts);
    CG_CHECK(p->boxsize > 0, "cg_create: boxsize must be positive");
    CG_CHECK(p->nprocs >= 1 && p->rank >= 0 && p->rank < p->nprocs,
             "cg_create: rank %d of %d", p->rank, p->nprocs);
    CG_CHECK(p->subdiv[0] == p->nprocs && p->subdiv[1] == 1 && p->subdiv[2] == 1,
             "cg_create: domains are x-slabs, subdiv must be (%d, 1, 1) (DESIGN.md section 6)",
             p->nprocs);
    CG_CHECK(p->gridsize % p->nprocs == 0 && (p->gridsize / p->nprocs) % 2 == 0,
             "cg_create: gridsize %lld must be divisible by 2*nprocs (mesh.py:1898-1905,3779)",
             (long long)p->gridsize);
    // (x-slab domains: power-of-two grids 16..2048 take the hand-written passes, any other
    // admissible size rocFFT per slab + a pack pass — dist_generic above)
    // the potential halo is filled from the neighbour's OWNED layers in one hop
    // (communicate_ghosts(grid,'='), communication.py:563-660): a slab thinner than the G = 3
    // halo layers would hand on its own ghost layers
    CG_CHECK(p->nprocs == 1 || p->gridsize / p->nprocs >= 3,
             "cg_create: %d domains leave slabs of %lld layers, thinner than the 3 halo layers "
             "(use fewer domains or a larger grid)", p->nprocs,
             (long long)(p->gridsize / p->nprocs));
    CG_HIP(hipSetDevice(p->device));
    if (!g_rocfft_ready) {
        CG_FFT(rocfft_setup());
        g_rocfft_ready = true;
    }
    cg_ctx *c = new cg_ctx();
    c->sub_begin = new SubstepBegin();
    c->p = *p;
    c->N = p->gridsize;
    c->pad = (c->N % 16 == 0) ? c->N + 16 : c->N + 2;
    if (p->nprocs == 1) c->xmap = XMap{0, c->N, 0, 1};
    else c->xmap = XMap{(c->N / p->nprocs) * p->rank, c->N / p->nprocs, 3, 0};
    c->ny = (c->N % 16 == 0) ? c->N + 1 : c->N;
    c->mesh_doubles = (c->xmap.nxl + 2 * c->xmap.G) * c->ny * c->pad;
    auto fail = [&]() {
        cg_destroy(c);
        return 1;
    };
    if (hipMalloc(&c->mesh, sizeof(double) * c->mesh_doubles) != hipSuccess) {
        cg_set_error("cg_create: hipMalloc of the %lld^3 mesh (%.2f GB) failed", (long long)c->N,
                     c->mesh_doubles * 8e-9);
        return fail();
    }
    c->device_bytes += 8 * c->mesh_doubles;
    c->mesh0 = c->mesh + (i64)c->xmap.G * c->ny * c->pad;
    if (p->nprocs == 1) {  // Fourier space shares the mesh in place, all rows
        c->four = (double2 *)c->mesh0;
        c->f_si = c->ny * (c->pad / 2);
        c->f_j0 = 0;
        c->f_nj = (int)c->N;
    }
    // geometry, reference expressions
    const double bgn[3] = {0, 0, 0};
    double cellsize_dep = p->boxsize / (double)p->gridsize;             // mesh.py:1577
    double domain_size_x = p->boxsize / (double)p->subdiv[0];           // communication.py:1777
    double cellsize_gat = domain_size_x / (double)(p->gridsize / p->subdiv[0]);  // mesh.py:408
    c->geom_deposit = make_geom(cellsize_dep, p->nghosts, p->cell_centered, bgn);
    c->geom_gather = make_geom(cellsize_gat, p->nghosts, p->cell_centered, bgn);
    // k-space tables by array index: k = idx - (idx >= N/2 ? N : 0)
    // n(k) = k*R[pi/gridsize] + machine_eps, s(k) = sin(n(k))   (mesh.py:2775-2776)
    {
        // q = n/s: the per-dimension sinc^-1 of the deconvolution, for the fused FFT pass
        std::vector<double> tn(c->N), ts(c->N), tq(c->N);
        double pi_over_n = kPi / (double)c->N;
        for (i64 i = 0; i < c->N; i++) {
            i64 k = i - (i >= c->N / 2 ? c->N : 0);
            tn[i] = (double)k * pi_over_n + kMachineEps;
            ts[i] = sin(tn[i]);
            tq[i] = tn[i] / ts[i];
        }
        if (hipMalloc(&c->ktab_n, 8 * c->N) != hipSuccess ||
            hipMalloc(&c->ktab_s, 8 * c->N) != hipSuccess ||
            hipMalloc(&c->ktab_q, 8 * c->N) != hipSuccess ||
            hipMemcpy(c->ktab_n, tn.data(), 8 * c->N, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c->ktab_q, tq.data(), 8 * c->N, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c->ktab_s, ts.data(), 8 * c->N, hipMemcpyHostToDevice) != hipSuccess) {
            cg_set_error("cg_create: k-space table upload failed");
            return fail();
        }
    }
    // tiles: cubic, 16 cells when the grid allows (at least 2 tiles per dimension),
    // for the particle memory order and the LDS-tiled deposit / gather kernels
    {
        int t = 16;
        while (t > 2 && (c->N % t || c->N / t < 2 || c->xmap.nxl % t)) t /= 2;
        if (c->N % t || c->xmap.nxl % t) {
            cg_set_error("cg_create: gridsize %lld / %d domains is not divisible by 2",
                         (long long)c->N, p->nprocs);
            return fail();
        }
        c->tiles = {t, t, t, (int)(c->xmap.nxl / t), (int)(c->N / t), (int)(c->N / t)};
        c->ntiles = (i64)c->tiles.ntx * c->tiles.nty * c->tiles.ntz;
        // 8 buckets per tile (which of the +x/+y/+z neighbour tiles a particle's CIC
        // cloud reaches), see cg_particles.hip
        if (hipMalloc(&c->tile_count, 4 * (8 * c->ntiles + 1)) != hipSuccess ||
            hipMalloc(&c->tile_cursor, 4 * (8 * c->ntiles + 1)) != hipSuccess) {
            cg_set_error("cg_create: tile table allocation failed");
            return fail();
        }
        c->device_bytes += 8 * (8 * c->ntiles + 1);
        if (hipMalloc(&c->err_flags, 4) != hipSuccess ||
            hipMemset(c->err_flags, 0, 4) != hipSuccess) {
            cg_set_error("cg_create: error word allocation failed");
            return fail();
        }
    }
    // FFT backend: the hand-written passes for power-of-two grids, rocFFT otherwise
    // (CONCEPT_GPU_FFT=rocfft forces the library, for A/B measurements)
    {
        const char *env = getenv("CONCEPT_GPU_FFT");
        bool force_rocfft = env && std::string(env) == "rocfft";
        c->custom_fft = cgk_fft_supported(c->N) && !force_rocfft;
    }
    if (c->custom_fft) {
        std::vector<double> tw(2 * c->N);
        for (i64 k = 0; k < c->N; k++) {
            long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k /
                            (long double)c->N;
            tw[2 * k] = (double)cosl(a);
            tw[2 * k + 1] = (double)sinl(a);
        }
        if (hipMalloc(&c->fft_tw, 16 * c->N) != hipSuccess ||
            hipMemcpy(c->fft_tw, tw.data(), 16 * c->N, hipMemcpyHostToDevice) != hipSuccess) {
            cg_set_error("cg_create: FFT twiddle upload failed");
            return fail();
        }
    } else if (p->nprocs == 1 && make_plans(c)) {  // (x-slab domains: plans made on first use)
        return fail();
    }
    *out = c;
    return 0;
}

extern "C" int cg_destroy(cg_ctx *c) {
    if (!c) return 0;
    dist_plans_destroy(c);
    if (c->plan_fwd) rocfft_plan_destroy(c->plan_fwd);
    if (c->plan_bwd) rocfft_plan_destroy(c->plan_bwd);
    if (c->info_fwd) rocfft_execution_info_destroy(c->info_fwd);
    if (c->info_bwd) rocfft_execution_info_destroy(c->info_bwd);
    (void)hipFree(c->fft_work);
    (void)hipFree(c->fft_tw);
    (void)hipFree(c->mesh);
    (void)hipFree(c->fetch_tmp);
    (void)hipFree(c->ktab_n);
    (void)hipFree(c->ktab_s);
    (void)hipFree(c->ktab_q);
    (void)hipFree(c->tile_count);
    (void)hipFree(c->tile_cursor);
    (void)hipFree(c->err_flags);
    (void)hipFree(c->sr_tile_active);
    (void)hipFree(c->mom2_partial);
    (void)hipFree(c->sr_stats);
    (void)hipFree(c->sr_sparse_partial);
    (void)hipFree(c->scan_tmp);
    (void)hipFree(c->tile_order_buf);
    if (c->tile_order_seen) (void)hipHostFree(c->tile_order_seen);
    (void)hipFree(c->sr_tmp);
    (void)hipFree(c->sr_sub_tmp);
    (void)hipFree(c->srd_small);
    (void)hipFree(c->srd_buf);
    (void)hipFree(c->srd_rung);
    for (auto &look : c->srd_look)
        if (look.ev) (void)hipEventDestroy(look.ev);
    if (c->srd_host) (void)hipHostFree(c->srd_host);
    if (c->srd_stream) {
        (void)hipStreamDestroy(c->srd_stream);
        (void)hipEventDestroy(c->srd_fork);
        (void)hipEventDestroy(c->srd_join);
    }
    for (int i = 0; i < 3; i++) {
        if (c->sr_streams[i]) (void)hipStreamDestroy(c->sr_streams[i]);
        if (c->sr_join[i]) (void)hipEventDestroy(c->sr_join[i]);
    }
    if (c->sr_fork) (void)hipEventDestroy(c->sr_fork);
    (void)hipFree(c->sr_active);
    delete c->sub_begin;
    (void)hipFree(c->sub_partial);
    delete c;
    return 0;
}

extern "C" int cg_set_stream(cg_ctx *c, void *s) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c, "cg_set_stream: null context");
    c->stream = (hipStream_t)s;
    return 0;
}

extern "C" int cg_synchronize(cg_ctx *c) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c, "cg_synchronize: null context");
    CG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int64_t cg_device_bytes(const cg_ctx *c) { return c ? c->device_bytes : 0; }

extern "C" int cg_error_flags(cg_ctx *c, uint32_t *flags_out) {
    CG_CHECK(c && flags_out, "cg_error_flags: null argument");
    CG_HIP(hipMemcpyAsync(flags_out, c->err_flags, 4, hipMemcpyDeviceToHost, c->stream));
    CG_HIP(hipMemsetAsync(c->err_flags, 0, 4, c->stream));
    CG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int cg_prepare_invalidate(cg_ctx *c) {
    CG_CHECK(c, "cg_prepare_invalidate: null context");
    c->prep_valid = false;
    return 0;
}

extern "C" int cg_mesh_zero(cg_ctx *c) {
    CG_CHECK(c, "cg_mesh_zero: null context");
    CG_HIP(hipMemsetAsync(c->mesh, 0, sizeof(double) * c->mesh_doubles, c->stream));
    return 0;
}

extern "C" int cg_deposit_cic(cg_ctx *c, const double *pos, int64_t n, double contribution) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && (pos || n == 0), "cg_deposit_cic: null argument");
    CG_CHECK(n >= 0, "cg_deposit_cic: negative particle count");
    if (n == 0) return 0;
    return cgk_deposit_cic(c, pos, n, contribution);
}

// ---- general particle_mesh() pieces (cg_general.hip) ----
#define CG_SINGLE(c, name) \
    CG_CHECK((c)->p.nprocs == 1, name ": single-domain entry point")
// k-space operations work on the context's Fourier view: the mesh itself on one domain, the
// buffer bound with cg_dist_bind_fourier on x-slab domains
#define CG_FOURIER(c, name) \
    CG_CHECK((c)->four != nullptr, name ": no Fourier buffer bound (cg_dist_bind_fourier)")

extern "C" int cg_fluid_add(cg_ctx *c, const double *fluid, double factor, int op_add) {
    CG_CHECK(c && fluid, "cg_fluid_add: null argument");
    return cgk_fluid_add(c, fluid, factor, op_add ? 1 : 0);
}

extern "C" int cg_fourier_nullify_nyquist(cg_ctx *c) {
    CG_CHECK(c, "cg_fourier_nullify_nyquist: null context");
    CG_FOURIER(c, "cg_fourier_nullify_nyquist");
    return cgk_nullify_nyquist(c);
}

extern "C" int cg_fourier_operate(cg_ctx *onto, cg_ctx *from, int deconv_order, int nlattice,
                                  const double *shift, int diff_dim, int op_add) {
    CG_CHECK(onto && from, "cg_fourier_operate: null context");
    CG_FOURIER(onto, "cg_fourier_operate");
    CG_FOURIER(from, "cg_fourier_operate");
    CG_CHECK(onto->p.nprocs == from->p.nprocs && onto->p.rank == from->p.rank,
             "cg_fourier_operate: the two meshes belong to different domain decompositions");
    CG_CHECK(onto->N == from->N && onto->pad == from->pad,
             "cg_fourier_operate: grid sizes %lld and %lld differ (different sizes go through "
             "cg_copy_modes)", (long long)from->N, (long long)onto->N);
    CG_CHECK(deconv_order >= 0 && deconv_order <= 8, "cg_fourier_operate: deconv_order %d",
             deconv_order);
    CG_CHECK(nlattice == 1 || nlattice == 2 || nlattice == 4,
             "cg_fourier_operate: nlattice %d not in {1, 2, 4}", nlattice);
    CG_CHECK(diff_dim >= -1 && diff_dim < 3,
             "fourier_operate() called with diff_dim = %d not in {-1, 0, 1, 2}", diff_dim);
    return cgk_fourier_operate(onto, from, deconv_order, nlattice, shift, diff_dim,
                               op_add ? 1 : 0);
}

extern "C" int cg_copy_modes(cg_ctx *onto, cg_ctx *from, int deconv_order, int nlattice,
                             const double *shift, int op_add) {
    CG_CHECK(onto && from, "cg_copy_modes: null context");
    CG_FOURIER(onto, "cg_copy_modes");
    CG_FOURIER(from, "cg_copy_modes");
    CG_CHECK(onto->N == from->N || (onto->p.nprocs == 1 && from->p.nprocs == 1),
             "cg_copy_modes: on x-slab domains rows of different grid sizes live on different "
             "domains: exchange them with cg_copy_modes_pack / cg_copy_modes_unpack");
    CG_CHECK(onto->p.boxsize == from->p.boxsize, "cg_copy_modes: the two meshes span different boxes");
    CG_CHECK(deconv_order >= 0 && deconv_order <= 8, "cg_copy_modes: deconv_order %d",
             deconv_order);
    CG_CHECK(nlattice == 1 || nlattice == 2 || nlattice == 4,
             "cg_copy_modes: nlattice %d not in {1, 2, 4}", nlattice);
    if (onto->N == from->N)
        return cgk_fourier_operate(onto, from, deconv_order, nlattice, shift, -1, op_add ? 1 : 0);
    return cgk_copy_modes(onto, from, deconv_order, nlattice, shift, op_add ? 1 : 0);
}

extern "C" int cg_copy_modes_pack(cg_ctx *from, int64_t n_small, const int32_t *rows_local,
                                  int64_t n_rows, double *out) {
    CG_CHECK(from && (n_rows == 0 || (rows_local && out)), "cg_copy_modes_pack: null argument");
    CG_FOURIER(from, "cg_copy_modes_pack");
    CG_CHECK(n_small >= 2 && n_small <= from->N && n_small % 2 == 0,
             "cg_copy_modes_pack: small grid size %lld", (long long)n_small);
    return cgk_copy_modes_pack(from, n_small, rows_local, n_rows, out);
}

extern "C" int cg_copy_modes_unpack(cg_ctx *onto, cg_ctx *from, int64_t n_small,
                                    const int32_t *rows_local, int64_t n_rows, const double *in,
                                    int deconv_order, int nlattice, const double *shift,
                                    int op_add) {
    CG_CHECK(onto && from && (n_rows == 0 || (rows_local && in)),
             "cg_copy_modes_unpack: null argument");
    CG_FOURIER(onto, "cg_copy_modes_unpack");
    CG_CHECK(onto->p.boxsize == from->p.boxsize,
             "cg_copy_modes_unpack: the two meshes span different boxes");
    CG_CHECK(n_small == (onto->N < from->N ? onto->N : from->N),
             "cg_copy_modes_unpack: n_small must be the smaller of the two grid sizes");
    CG_CHECK(deconv_order >= 0 && deconv_order <= 8, "cg_copy_modes_unpack: deconv_order %d",
             deconv_order);
    CG_CHECK(nlattice == 1 || nlattice == 2 || nlattice == 4,
             "cg_copy_modes_unpack: nlattice %d not in {1, 2, 4}", nlattice);
    return cgk_copy_modes_unpack(onto, from, n_small, rows_local, n_rows, in, deconv_order,
                                 nlattice, shift, op_add ? 1 : 0);
}

extern "C" int cg_deposit(cg_ctx *c, const double *pos, int64_t n, double contribution, int order,
                          const double *shift) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && (pos || n == 0), "cg_deposit: null argument");
    CG_CHECK(order >= 1 && order <= 4,
             "interpolate_particles() called with order = %d not in {1 (NGP), 2 (CIC), 3 (TSC), 4 (PCS)}",
             order);
    double cellsize = c->p.boxsize / (double)c->p.gridsize;  // mesh.py:1576
    return cgk_deposit_general(c, pos, n, contribution, order,
                               make_geom_shift(cellsize, c->p.nghosts, c->p.cell_centered, shift, -1));
}

extern "C" int cg_gather_scalar(cg_ctx *c, const double *pos, double *mom, int64_t n, int dim,
                                int order, const double *shift, double factor) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && ((pos && mom) || n == 0), "cg_gather_scalar: null argument");
    CG_CHECK(order >= 1 && order <= 4,
             "interpolate_domaingrid_to_particles() called with order = %d not in {1, 2, 3, 4}",
             order);
    CG_CHECK(dim >= 0 && dim < 3,
             "apply_particle_mesh_force() called with dim = %d not in {0, 1, 2}", dim);
    double cellsize = c->p.boxsize / (double)c->p.gridsize;  // mesh.py:408 (one domain)
    c->prep_valid = false;
    return cgk_gather_scalar(c, pos, mom, n, dim, order,
                             make_geom_shift(cellsize, c->p.nghosts, c->p.cell_centered, shift, +1),
                             factor);
}

extern "C" int cg_mesh_diff(cg_ctx *dst, cg_ctx *src, int dim, int diff_order) {
    CG_CHECK(dst && src && dst != src, "cg_mesh_diff: two distinct contexts are needed");
    CG_CHECK(dst->N == src->N && dst->pad == src->pad && dst->mesh_doubles == src->mesh_doubles,
             "cg_mesh_diff: the two meshes differ in shape");
    CG_CHECK(dim >= 0 && dim < 3, "diff_domaingrid() called with dim = %d not in {0, 1, 2}", dim);
    CG_CHECK(diff_order == 1 || diff_order == 2 || diff_order == 4 || diff_order == 6 ||
                 diff_order == 8,
             "diff_domaingrid() called with order = %d not in {1, 2, 4, 6, 8}", diff_order);
    // the stencil reaches (order + 1)/2 cells: across a slab face that is halo (3 layers)
    CG_CHECK(src->p.nprocs == 1 || (diff_order + 1) / 2 <= src->xmap.G,
             "cg_mesh_diff: differentiation order %d reaches %d layers beyond a slab face, the "
             "halo holds %d", diff_order, (diff_order + 1) / 2, src->xmap.G);
    return cgk_mesh_diff(dst, src, dim, diff_order);
}

extern "C" int cg_ewald_tabulate(cg_ctx *c, int gridsize, double *grid) {
    CG_CHECK(c && grid, "cg_ewald_tabulate: null argument");
    CG_CHECK(gridsize >= 2 && gridsize <= 512, "cg_ewald_tabulate: ewald_gridsize %d", gridsize);
    return cgk_ewald_tabulate(c, gridsize, grid);
}

extern "C" int cg_pp_kick(cg_ctx *c, const double *pos_r, int64_t n_r, double *dmom_r,
                          const double *pos_s, int64_t n_s, int same, const double *ewald_grid,
                          int ewald_gridsize, double softening, int kernel, double factor,
                          const double *factors, const signed char *rung,
                          const signed char *rung_jumped, int lowest_active) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && ((pos_r && dmom_r) || n_r == 0) && (pos_s || n_s == 0),
             "cg_pp_kick: null argument");
    CG_CHECK(kernel >= 0 && kernel <= 2, "Softening kernel %d not understood", kernel);
    CG_CHECK(!ewald_grid || ewald_gridsize >= 2, "cg_pp_kick: ewald_gridsize %d", ewald_gridsize);
    CG_CHECK(!same || n_r <= n_s, "cg_pp_kick: same = 1: the first n_r suppliers are the receivers");
    CG_CHECK(!rung || (factors && rung_jumped), "cg_pp_kick: rungs need factors and jumped indices");
    return cgk_pp_kick(c, pos_r, n_r, dmom_r, pos_s, n_s, same ? 1 : 0, ewald_grid, ewald_gridsize,
                       softening, kernel, factor, factors, rung, rung_jumped, lowest_active);
}

extern "C" int cg_mesh_copy(cg_ctx *dst, cg_ctx *src) {
    CG_CHECK(dst && src, "cg_mesh_copy: null context");
    CG_CHECK(dst->N == src->N && dst->mesh_doubles == src->mesh_doubles,
             "cg_mesh_copy: the two meshes differ in shape");
    CG_HIP(hipMemcpyAsync(dst->mesh, src->mesh, sizeof(double) * src->mesh_doubles,
                          hipMemcpyDeviceToDevice, dst->stream));
    return 0;
}

extern "C" int cg_fluid_kick(cg_ctx *c, double *J, const double *rho, const double *P, int dim,
                             int diff_order, double minus_dt, double inv_c2) {
    CG_CHECK(c && J && rho && P, "cg_fluid_kick: null argument");
    CG_CHECK(dim >= 0 && dim < 3,
             "apply_particle_mesh_force() called with dim = %d not in {0, 1, 2}", dim);
    CG_CHECK(diff_order == 0 || diff_order == 1 || diff_order == 2 || diff_order == 4 ||
                 diff_order == 6 || diff_order == 8,
             "cg_fluid_kick: differentiation order %d not in {0 (the mesh holds the force), 1, 2, "
             "4, 6, 8}", diff_order);
    CG_CHECK(c->p.nprocs == 1 || (diff_order + 1) / 2 <= c->xmap.G,
             "cg_fluid_kick: differentiation order %d reaches %d layers beyond a slab face, the "
             "halo holds %d", diff_order, (diff_order + 1) / 2, c->xmap.G);
    return cgk_fluid_kick(c, J, rho, P, dim, diff_order, minus_dt, inv_c2);
}

extern "C" int cg_poisson_forward(cg_ctx *c, int deconv_order, double C, int long_range, double E,
                                  int apply_kernel) {
    CG_CHECK(c, "cg_poisson_forward: null context");
    CG_CHECK(c->p.nprocs == 1, "cg_poisson_forward: single-domain entry point; x-slab domains use cg_dist_fft_*");
    CG_CHECK(deconv_order >= 0 && deconv_order <= 8, "cg_poisson_forward: deconv_order %d",
             deconv_order);
    if (c->custom_fft) {
        if (cgk_fft(c, 0, 0, 0.0, 0, 0.0)) return 1;
    } else {
        CG_FFT(rocfft_execution_info_set_stream(c->info_fwd, c->stream));
        void *buf[1] = {c->mesh};
        CG_FFT(rocfft_execute(c->plan_fwd, buf, nullptr, c->info_fwd));
    }
    if (apply_kernel) return cgk_kspace(c, deconv_order, C, long_range, E);
    return 0;
}

extern "C" int cg_poisson_kernel(cg_ctx *c, int deconv_order, double C, int long_range, double E) {
    CG_CHECK(c, "cg_poisson_kernel: null context");
    CG_FOURIER(c, "cg_poisson_kernel");
    CG_CHECK(deconv_order >= 0 && deconv_order <= 8, "cg_poisson_kernel: deconv_order %d",
             deconv_order);
    return cgk_kspace(c, deconv_order, C, long_range, E);
}

extern "C" int cg_poisson_backward(cg_ctx *c) {
    CG_CHECK(c, "cg_poisson_backward: null context");
    CG_CHECK(c->p.nprocs == 1, "cg_poisson_backward: single-domain entry point; x-slab domains use cg_dist_fft_*");
    if (c->custom_fft) return cgk_fft(c, 1, 0, 0.0, 0, 0.0);
    CG_FFT(rocfft_execution_info_set_stream(c->info_bwd, c->stream));
    void *buf[1] = {c->mesh};
    CG_FFT(rocfft_execute(c->plan_bwd, buf, nullptr, c->info_bwd));
    return 0;
}

extern "C" int cg_poisson_solve(cg_ctx *c, int deconv_order, double C, int long_range, double E) {
    CG_CHECK(c, "cg_poisson_solve: null context");
    CG_CHECK(c->p.nprocs == 1, "cg_poisson_solve: single-domain entry point; x-slab domains use cg_dist_fft_*");
    CG_CHECK(deconv_order >= 0 && deconv_order <= 8, "cg_poisson_solve: deconv_order %d",
             deconv_order);
    // hand-written FFT: the k-space kernel is fused into the x pass (5 passes in all)
    if (c->custom_fft) return cgk_fft(c, 2, deconv_order, C, long_range, E);
    if (cg_poisson_forward(c, deconv_order, C, long_range, E, 1)) return 1;
    return cg_poisson_backward(c);
}

extern "C" int cg_poisson_solve_timed(cg_ctx *c, int deconv_order, double C, int long_range,
                                      double E, double pass_ms[5]) {
    CG_CHECK(c && pass_ms, "cg_poisson_solve_timed: null argument");
    CG_CHECK(c->custom_fft && c->p.nprocs == 1,
             "cg_poisson_solve_timed: hand-written single-domain FFT only");
    hipEvent_t ev[6];
    for (auto &e : ev) CG_HIP(hipEventCreate(&e));
    c->pass_events = ev;
    int rc = cg_poisson_solve(c, deconv_order, C, long_range, E);
    c->pass_events = nullptr;
    if (!rc) {
        CG_HIP(hipEventSynchronize(ev[5]));
        for (int i = 0; i < 5; i++) {
            float ms = 0;
            CG_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            pass_ms[i] = ms;
        }
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}

extern "C" int cg_gather_kick(cg_ctx *c, const double *pos, double *mom, int64_t n, int diff_order,
                              double factor) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && ((pos && mom) || n == 0), "cg_gather_kick: null argument");
    CG_CHECK(diff_order == 2 || diff_order == 4,
             "cg_gather_kick: differentiation order %d not built (2 and 4 are)", diff_order);
    CG_CHECK((diff_order + 1) / 2 <= c->p.nghosts,
             "cg_gather_kick: differentiation order %d needs nghosts >= %d (commons.py:4411-4432)",
             diff_order, (diff_order + 1) / 2);
    if (n == 0) return 0;
    c->prep_valid = false;
    return cgk_gather_kick(c, pos, mom, n, diff_order, factor);
}

extern "C" int cg_drift(cg_ctx *c, double *pos, const double *mom, int64_t n,
                        double dt_over_mass) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && ((pos && mom) || n == 0), "cg_drift: null argument");
    if (n == 0) return 0;
    c->prep_valid = false;
    return cgk_drift(c, pos, mom, n, dt_over_mass);
}

extern "C" int cg_measure_momentum(cg_ctx *c, const double *mom, int64_t n, double *out,
                                   double *scratch) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && out && scratch && (mom || n == 0), "cg_measure_momentum: null argument");
    CG_CHECK(n >= 0, "cg_measure_momentum: n out of range");
    return cgk_measure_mom(c, mom, n, out, scratch);
}

extern "C" int cg_measure_momentum_regions(cg_ctx *c, const double *mom, const uint32_t *start,
                                           const uint32_t *count, double *out, double *scratch) {
    CG_CHECK(c && mom && start && out && scratch, "cg_measure_momentum_regions: null argument");
    return cgk_measure_mom_regions(c, mom, start, count, out, scratch);
}

extern "C" int cg_tile_info(const cg_ctx *c, int64_t info[3]) {
    CG_CHECK(c && info, "cg_tile_info: null argument");
    info[0] = c->tiles.tx;
    info[1] = c->tiles.nty;
    info[2] = 8 * c->ntiles + 1;
    return 0;
}

extern "C" int cg_deposit_cic_tiled(cg_ctx *c, const double *pos, int64_t n,
                                    const uint32_t *tile_offset, double contribution,
                                    int accumulate) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && tile_offset && (pos || n == 0), "cg_deposit_cic_tiled: null argument");
    CG_CHECK(n >= 0 && n < (1ll << 32), "cg_deposit_cic_tiled: n out of range");
    return cgk_deposit_cic_tiled(c, pos, n, tile_offset, nullptr, contribution, accumulate);
}

extern "C" int cg_deposit_cic_regions(cg_ctx *c, const double *pos, const uint32_t *start,
                                      const uint32_t *count, double contribution,
                                      int accumulate) {
    CG_CHECK(c && pos && start && count, "cg_deposit_cic_regions: null argument");
    return cgk_deposit_cic_tiled(c, pos, 0, start, count, contribution, accumulate);
}

extern "C" int64_t cg_region_capacity(const cg_ctx *c, int64_t n) {
    return c ? n + n / 4 + 32 * 8 * c->ntiles : 0;
}

extern "C" int cg_predict_regions(cg_ctx *c, const uint32_t *start_in, const uint32_t *count_in,
                                  uint32_t *start_out) {
    CG_CHECK(c && start_in && start_out, "cg_predict_regions: null argument");
    return cgk_predict_regions(c, start_in, count_in, start_out);
}

extern "C" int cg_tile_order_read(cg_ctx *c, uint32_t *heavy_out, int64_t capacity,
                                  int64_t *n_heavy) {
    CG_CHECK(c && n_heavy && (heavy_out || capacity == 0), "cg_tile_order_read: null argument");
    *n_heavy = -1;
    if (!c->tile_order_on) return 0;
    CG_HIP(hipStreamSynchronize(c->stream));
    unsigned n = 0;
    CG_HIP(hipMemcpy(&n, c->tile_order + c->tile_order_cap + 1, sizeof n, hipMemcpyDeviceToHost));
    *n_heavy = n;
    const size_t m = (int64_t)n < capacity ? (size_t)n : (size_t)capacity;
    if (m) CG_HIP(hipMemcpy(heavy_out, c->tile_order, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int cg_gather_kick_drift_scatter(
    cg_ctx *c, const double *pos_in, const double *mom_in, const int64_t *ids_in,
    const uint32_t *start_in, const uint32_t *count_in, double *pos_out, double *mom_out,
    int64_t *ids_out, const uint32_t *start_out, uint32_t *count_out, int diff_order,
    double factor, double dt_over_mass, const int64_t *aux_in, int64_t *aux_out,
    int64_t out_capacity) {
    CG_CHECK(c && pos_in && mom_in && start_in && pos_out && mom_out && start_out && count_out,
             "cg_gather_kick_drift_scatter: null argument");
    CG_CHECK(pos_in != pos_out && mom_in != mom_out,
             "cg_gather_kick_drift_scatter: in/out must not alias");
    CG_CHECK((ids_in == nullptr) == (ids_out == nullptr),
             "cg_gather_kick_drift_scatter: ids_in and ids_out must both be given or both be null");
    CG_CHECK((aux_in == nullptr) == (aux_out == nullptr),
             "cg_gather_kick_drift_scatter: aux_in and aux_out must both be given or both be null");
    CG_CHECK(diff_order == 2 || diff_order == 4,
             "cg_gather_kick_drift_scatter: differentiation order %d not built", diff_order);
    CG_CHECK((diff_order + 1) / 2 <= c->p.nghosts,
             "cg_gather_kick_drift_scatter: differentiation order %d needs nghosts >= %d",
             diff_order, (diff_order + 1) / 2);
    CG_CHECK(c->p.nprocs == 1 || c->emig_rows,
             "cg_gather_kick_drift_scatter: on x-slab domains the particles leaving the slab need "
             "a row buffer (cg_set_emigrant_rows)");
    CG_CHECK(out_capacity > 0, "cg_gather_kick_drift_scatter: out_capacity must be positive");
    FusedScatter fs{count_in, start_out, count_out, pos_out, mom_out, ids_in, ids_out,
                    aux_in, aux_out, out_capacity};
    c->prep_valid = false;
    return cgk_gather_kick_tiled(c, pos_in, const_cast<double *>(mom_in), 0, start_in, diff_order,
                                 factor, 0, dt_over_mass, &fs);
}

extern "C" int cg_gather_kick_tiled(cg_ctx *c, const double *pos, double *mom, int64_t n,
                                    const uint32_t *tile_offset, int diff_order, double factor) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && tile_offset && ((pos && mom) || n == 0), "cg_gather_kick_tiled: null argument");
    CG_CHECK(diff_order == 2 || diff_order == 4,
             "cg_gather_kick_tiled: differentiation order %d not built (2 and 4 are)", diff_order);
    CG_CHECK((diff_order + 1) / 2 <= c->p.nghosts,
             "cg_gather_kick_tiled: differentiation order %d needs nghosts >= %d "
             "(commons.py:4411-4432)", diff_order, (diff_order + 1) / 2);
    if (n == 0) return 0;
    return cgk_gather_kick_tiled(c, pos, mom, n, tile_offset, diff_order, factor, 0, 0.0);
}

extern "C" int cg_gather_kick_tiled_prepare(cg_ctx *c, const double *pos, double *mom, int64_t n,
                                            const uint32_t *tile_offset, int diff_order,
                                            double factor, double next_dt_over_mass) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && tile_offset && ((pos && mom) || n == 0),
             "cg_gather_kick_tiled_prepare: null argument");
    CG_CHECK(diff_order == 2 || diff_order == 4,
             "cg_gather_kick_tiled_prepare: differentiation order %d not built", diff_order);
    CG_CHECK((diff_order + 1) / 2 <= c->p.nghosts,
             "cg_gather_kick_tiled_prepare: differentiation order %d needs nghosts >= %d",
             diff_order, (diff_order + 1) / 2);
    if (n == 0) return 0;
    return cgk_gather_kick_tiled(c, pos, mom, n, tile_offset, diff_order, factor, 1,
                                 next_dt_over_mass);
}

extern "C" int cg_set_emigrant_list(cg_ctx *c, int64_t *idx, uint32_t *count, int64_t cap) {
    CG_CHECK(c, "cg_set_emigrant_list: null context");
    CG_CHECK((idx == nullptr) == (count == nullptr) && cap >= 0,
             "cg_set_emigrant_list: idx and count must both be given or both be null");
    c->emig_idx = idx;
    c->emig_count = count;
    c->emig_cap = idx ? cap : 0;
    return 0;
}

extern "C" int cg_set_emigrant_rows(cg_ctx *c, double *rows, uint32_t *count, int64_t cap) {
    CG_CHECK(c, "cg_set_emigrant_rows: null context");
    CG_CHECK((rows == nullptr) == (count == nullptr) && cap >= 0,
             "cg_set_emigrant_rows: rows and count must both be given or both be null");
    c->emig_rows = rows;
    c->emig_rows_count = count;
    c->emig_rows_cap = rows ? cap : 0;
    return 0;
}

extern "C" int cg_set_momentum_sum(cg_ctx *c, double *sum_out) {
    CG_CHECK(c, "cg_set_momentum_sum: null context");
    c->mom2_sum_out = sum_out;
    return 0;
}

extern "C" int cg_emigrant_rows_dest(cg_ctx *c, const double *rows, const uint32_t *count,
                                     int64_t cap, int32_t *dest, int32_t *send_counts) {
    CG_CHECK(c && send_counts && (cap == 0 || (rows && count && dest)),
             "cg_emigrant_rows_dest: null argument");
    return cgk_emigrant_rows_dest(c, rows, count, cap, dest, send_counts);
}

extern "C" int cg_region_insert(cg_ctx *c, const double *rows, int64_t m, const uint32_t *start,
                                uint32_t *count, double *pos_out, double *mom_out,
                                int64_t *ids_out, int64_t *aux_out, int64_t capacity) {
    CG_CHECK(c && start && count && pos_out && mom_out && (m == 0 || rows),
             "cg_region_insert: null argument");
    CG_CHECK(m >= 0, "cg_region_insert: negative row count");
    return cgk_region_insert(c, rows, m, start, count, pos_out, mom_out, ids_out, aux_out,
                             capacity);
}

extern "C" int cg_sort_particles(cg_ctx *c, const double *pos_in, const double *mom_in,
                                 const int64_t *ids_in, double *pos_out, double *mom_out,
                                 int64_t *ids_out, int64_t n, uint32_t *tile_offset_out) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    // (an empty set — a domain without particles — may come with null arrays)
    CG_CHECK(c && tile_offset_out && (n == 0 || (pos_in && mom_in && pos_out && mom_out)),
             "cg_sort_particles: null argument");
    CG_CHECK(n == 0 || (pos_in != pos_out && mom_in != mom_out),
             "cg_sort_particles: in/out must not alias");
    CG_CHECK(n == 0 || (ids_in == nullptr) == (ids_out == nullptr),
             "cg_sort_particles: ids_in and ids_out must both be given or both be null");
    CG_CHECK(n >= 0 && n < (1ll << 32), "cg_sort_particles: n out of range");
    return cgk_sort(c, pos_in, mom_in, ids_in, pos_out, mom_out, ids_out, n, tile_offset_out, 0,
                    0.0, 0);
}

extern "C" int cg_drift_sort(cg_ctx *c, const double *pos_in, const double *mom_in,
                             const int64_t *ids_in, double *pos_out, double *mom_out,
                             int64_t *ids_out, int64_t n, double dt_over_mass,
                             uint32_t *tile_offset_out) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && tile_offset_out && (n == 0 || (pos_in && mom_in && pos_out && mom_out)),
             "cg_drift_sort: null argument");
    CG_CHECK(n == 0 || (pos_in != pos_out && mom_in != mom_out),
             "cg_drift_sort: in/out must not alias");
    CG_CHECK(n == 0 || (ids_in == nullptr) == (ids_out == nullptr),
             "cg_drift_sort: ids_in and ids_out must both be given or both be null");
    CG_CHECK(n >= 0 && n < (1ll << 32), "cg_drift_sort: n out of range");
    // x-slab domains: particles whose drifted position leaves the slab are dropped (the host
    // ships them beforehand: cg_owner_rank_drifted + exchange + cg_prepare_rebind)
    // a histogram prepared by cg_gather_kick_tiled_prepare for exactly these arrays and this
    // drift replaces the first pass (anything else touching pos/mom in between is a caller bug
    // the pointers cannot reveal: the prepared state is consumed by the very next sort only)
    int use_prepared = c->prep_valid && c->prep_pos == pos_in && c->prep_mom == mom_in &&
                       c->prep_n == n && c->prep_dtm == dt_over_mass;
    return cgk_sort(c, pos_in, mom_in, ids_in, pos_out, mom_out, ids_out, n, tile_offset_out, 1,
                    dt_over_mass, use_prepared);
}

extern "C" int cg_owner_rank_drifted(cg_ctx *c, const double *pos, const double *mom, int64_t n,
                                     double dt_over_mass, int32_t *owner) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && (n == 0 || (pos && mom && owner)), "cg_owner_rank_drifted: null argument");
    return cgk_owner_rank_drifted(c, pos, mom, n, dt_over_mass, owner);
}

extern "C" int cg_prepare_rebind(cg_ctx *c, const double *pos, const double *mom, int64_t n_total,
                                 const double *add_pos, const double *add_mom, int64_t n_add) {
    CG_CHECK(c && (n_total == 0 || (pos && mom)) && (n_add == 0 || (add_pos && add_mom)),
             "cg_prepare_rebind: null argument");
    CG_CHECK(n_total >= 0 && n_add >= 0 && n_total < (1ll << 32), "cg_prepare_rebind: sizes");
    return cgk_prepare_rebind(c, pos, mom, n_total, add_pos, add_mom, n_add);
}

extern "C" int cg_cic_indices(cg_ctx *c, const double *pos, int64_t n, int for_gather,
                              int64_t *idx_out) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && (n == 0 || (pos && idx_out)), "cg_cic_indices: null argument");
    if (n == 0) return 0;
    return cgk_cic_indices(c, pos, n, for_gather, idx_out);
}

static int shortrange_cells_checks(cg_ctx *c, const void *pos, int64_t n, int64_t nt,
                                   double tile_extent, const void *order_out,
                                   const void *offset_out, const void *pos_sorted_out) {
    CG_CHECK(c && offset_out && (n == 0 || (pos && order_out && pos_sorted_out)),
             "cg_shortrange_cells: null argument");
    CG_CHECK(nt >= 4, "The global gravity tiling needs to have at least 4 tiles across the box in "
                      "every direction (species.py:3971); got %lld", (long long)nt);
    CG_CHECK(nt <= 512 && n < (1ll << 32), "cg_shortrange_cells: size out of range");
    // the sweeps take the tile extent from the box (species.py:607-609): a list made with another
    // one would put a particle on a tile border into one tile here and into its neighbour there.
    // The list is made with boxsize/nt itself; the argument only has to agree with it (4 ulp: a
    // caller may have formed it as boxsize*(1/nt)).
    const double ext = c->p.boxsize / (double)nt;
    CG_CHECK(fabs(tile_extent - ext) <= 4 * 2.220446049250313e-16 * ext,
             "cg_shortrange_cells: tile_extent must be boxsize/nt (%.17g), got %.17g", ext,
             tile_extent);
    return 0;
}

extern "C" int cg_shortrange_cells(cg_ctx *c, const double *pos, int64_t n, int64_t nt,
                                   double tile_extent, uint32_t *order_out,
                                   uint32_t *offset_out, double *pos_sorted_out) {
    if (shortrange_cells_checks(c, pos, n, nt, tile_extent, order_out, offset_out, pos_sorted_out))
        return 1;
    return cgk_shortrange_cells(c, pos, n, nt, c->p.boxsize / (double)nt, order_out, offset_out,
                                pos_sorted_out, nullptr, nullptr, 0, nullptr, nullptr);
}

extern "C" int cg_shortrange_cells_rungs(cg_ctx *c, const double *pos, int64_t n, int64_t nt,
                                         double tile_extent, const int8_t *rung,
                                         const int8_t *rung_jumped, int lowest_active_rung,
                                         uint32_t *order_out, uint32_t *offset_out,
                                         double *pos_sorted_out, uint32_t *nact_out,
                                         int8_t *rung_jumped_sorted_out) {
    if (shortrange_cells_checks(c, pos, n, nt, tile_extent, order_out, offset_out, pos_sorted_out))
        return 1;
    CG_CHECK(nact_out && (n == 0 || (rung && (rung_jumped || !rung_jumped_sorted_out))),
             "cg_shortrange_cells_rungs: null argument");
    return cgk_shortrange_cells(c, pos, n, nt, c->p.boxsize / (double)nt, order_out, offset_out,
                                pos_sorted_out, (const signed char *)rung,
                                (const signed char *)rung_jumped, lowest_active_rung, nact_out,
                                (signed char *)rung_jumped_sorted_out);
}

static int sweep_cells_checks(cg_ctx *c, const void *a, const void *b, const void *d,
                              const void *e, const void *f, const void *g, const void *t,
                              int64_t nt, int64_t tablesize, double r2_index_scaling,
                              double r2_max) {
    // a = pos_r, b = order_r, d = offset_r, e = dmom_r, f = pos_s, g = offset_s, t = table;
    // the particle arrays of an empty set may be null (only what the offsets span is touched)
    (void)a, (void)b, (void)e, (void)f;
    CG_CHECK(c && d && g && t, "cg_shortrange_sweep_cells: null argument");
    CG_CHECK(nt >= 4 && nt <= 512, "cg_shortrange_sweep_cells: nt = %lld", (long long)nt);
    // the largest index the sweep can form is int(r2_max*scaling): must be inside the table
    CG_CHECK((int64_t)(r2_max * r2_index_scaling) < tablesize,
             "cg_shortrange_sweep_cells: table of %lld entries too short for r2_max*scaling = %g",
             (long long)tablesize, r2_max * r2_index_scaling);
    // a cell is half a tile: the force range must not exceed two cells, i.e. one tile
    CG_CHECK(r2_max <= (c->p.boxsize / (double)nt) * (c->p.boxsize / (double)nt) * (1 + 1e-12),
             "cg_shortrange_sweep_cells: the force range exceeds the tile extent");
    return 0;
}

extern "C" int cg_shortrange_sweep_cells(cg_ctx *c, const double *pos_r_sorted,
                                         const uint32_t *order_r, const uint32_t *offset_r,
                                         double *dmom_r, const double *pos_s_sorted,
                                         const uint32_t *offset_s, int64_t nt,
                                         const double *table, int64_t tablesize,
                                         double r2_index_scaling, double r2_max, double factor) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    if (sweep_cells_checks(c, pos_r_sorted, order_r, offset_r, dmom_r, pos_s_sorted, offset_s,
                           table, nt, tablesize, r2_index_scaling, r2_max))
        return 1;
    return cgk_shortrange_sweep_cells(c, pos_r_sorted, order_r, offset_r, dmom_r, pos_s_sorted,
                                      offset_s, nt, table, r2_index_scaling, r2_max, factor,
                                      nullptr, nullptr, nullptr, 0, nullptr, nullptr, -1);
}

extern "C" int cg_shortrange_sweep_cells_rungs(
    cg_ctx *c, const double *pos_r_sorted, const uint32_t *order_r, const uint32_t *offset_r,
    double *dmom_r, const double *pos_s_sorted, const uint32_t *offset_s, int64_t nt,
    const double *table, int64_t tablesize, double r2_index_scaling, double r2_max,
    const double *factors, const int8_t *rung_r, const int8_t *rung_jumped_r,
    int lowest_active_rung) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    if (sweep_cells_checks(c, pos_r_sorted, order_r, offset_r, dmom_r, pos_s_sorted, offset_s,
                           table, nt, tablesize, r2_index_scaling, r2_max))
        return 1;
    // (rung arrays of an empty receiver set may be null)
    CG_CHECK(factors, "cg_shortrange_sweep_cells_rungs: null argument");
    return cgk_shortrange_sweep_cells(c, pos_r_sorted, order_r, offset_r, dmom_r, pos_s_sorted,
                                      offset_s, nt, table, r2_index_scaling, r2_max, 0.0, factors,
                                      (const signed char *)rung_r,
                                      (const signed char *)rung_jumped_r, lowest_active_rung,
                                      nullptr, nullptr, -1);
}

extern "C" int cg_shortrange_sweep_cells_active(
    cg_ctx *c, const double *pos_r_sorted, const uint32_t *order_r, const uint32_t *offset_r,
    const uint32_t *nact_r, const int8_t *rung_jumped_sorted_r, double *dmom_r,
    const double *pos_s_sorted, const uint32_t *offset_s, int64_t nt, const double *table,
    int64_t tablesize, double r2_index_scaling, double r2_max, const double *factors,
    const int8_t *rung_r, const int8_t *rung_jumped_r, int lowest_active_rung,
    int64_t n_active_max) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    if (sweep_cells_checks(c, pos_r_sorted, order_r, offset_r, dmom_r, pos_s_sorted, offset_s,
                           table, nt, tablesize, r2_index_scaling, r2_max))
        return 1;
    CG_CHECK(factors && nact_r, "cg_shortrange_sweep_cells_active: null argument");
    CG_CHECK(n_active_max < (1ll << 31), "cg_shortrange_sweep_cells_active: n_active_max");
    return cgk_shortrange_sweep_cells(c, pos_r_sorted, order_r, offset_r, dmom_r, pos_s_sorted,
                                      offset_s, nt, table, r2_index_scaling, r2_max, 0.0, factors,
                                      (const signed char *)rung_r,
                                      (const signed char *)rung_jumped_r, lowest_active_rung,
                                      nact_r, (const signed char *)rung_jumped_sorted_r,
                                      lowest_active_rung > 0 ? n_active_max : -1);
}

extern "C" int cg_shortrange_stats(cg_ctx *c, int enable, uint64_t *out) {
    CG_CHECK(c, "cg_shortrange_stats: null context");
    if (enable) {
        if (!c->sr_stats) CG_HIP(hipMalloc((void **)&c->sr_stats, 8 * sizeof(uint64_t)));
        CG_HIP(hipMemsetAsync(c->sr_stats, 0, 8 * sizeof(uint64_t), c->stream));
        return 0;
    }
    if (!c->sr_stats) {
        if (out) memset(out, 0, 8 * sizeof(uint64_t));
        return 0;
    }
    CG_HIP(hipStreamSynchronize(c->stream));
    if (out) CG_HIP(hipMemcpy(out, c->sr_stats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    (void)hipFree(c->sr_stats);
    c->sr_stats = nullptr;
    return 0;
}

extern "C" int cg_shortrange_tiles(cg_ctx *c, const double *pos, int64_t n, int64_t nt,
                                   double tile_extent, const int8_t *rung,
                                   int lowest_active_rung, uint32_t *order_out,
                                   uint32_t *offset_out, double *pos_sorted_out) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && offset_out && (n == 0 || (pos && order_out)),
             "cg_shortrange_tiles: null argument");
    CG_CHECK(nt >= 4, "The global gravity tiling needs to have at least 4 tiles across the box in "
                      "every direction (species.py:3971); got %lld", (long long)nt);
    CG_CHECK(nt <= 1024 && n < (1ll << 32), "cg_shortrange_tiles: size out of range");
    CG_CHECK(fabs(tile_extent - c->p.boxsize / (double)nt) <=
                 4 * 2.220446049250313e-16 * (c->p.boxsize / (double)nt),
             "cg_shortrange_tiles: tile_extent must be boxsize/nt (%.17g), got %.17g",
             c->p.boxsize / (double)nt, tile_extent);
    tile_extent = c->p.boxsize / (double)nt;
    return cgk_shortrange_tiles(c, pos, n, nt, tile_extent, (const signed char *)rung,
                                lowest_active_rung, order_out, offset_out, pos_sorted_out);
}

extern "C" int cg_dmom_nullify(cg_ctx *c, double *dmom, const int8_t *rung, int64_t n,
                               int lowest_active_rung) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && (dmom || n == 0), "cg_dmom_nullify: null argument");
    return cgk_dmom_active(c, nullptr, dmom, (const signed char *)rung, n, lowest_active_rung, 0);
}
extern "C" int cg_dmom_apply(cg_ctx *c, double *mom, const double *dmom, const int8_t *rung,
                             int64_t n, int lowest_active_rung) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && ((mom && dmom) || n == 0), "cg_dmom_apply: null argument");
    c->prep_valid = false;  // mom changes: a prepared drift histogram no longer describes it
    return cgk_dmom_active(c, mom, (double *)dmom, (const signed char *)rung, n,
                           lowest_active_rung, 1);
}
extern "C" int cg_dmom_to_acc(cg_ctx *c, double *dmom, const int8_t *rung,
                              const int8_t *rung_jumped, int64_t n, int lowest_active_rung,
                              const double *conversion_factors, int any_rung_jumps) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && conversion_factors && ((dmom && rung && rung_jumped) || n == 0),
             "cg_dmom_to_acc: null argument");
    return cgk_dmom_to_acc(c, dmom, (const signed char *)rung, (const signed char *)rung_jumped, n,
                           lowest_active_rung, conversion_factors, any_rung_jumps);
}
extern "C" int cg_assign_rungs(cg_ctx *c, const double *acc, int8_t *rung, int8_t *rung_jumped,
                               int64_t n, double rung_factor, int N_rungs) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && ((acc && rung && rung_jumped) || n == 0), "cg_assign_rungs: null argument");
    CG_CHECK(N_rungs >= 1 && N_rungs <= 42, "cg_assign_rungs: N_rungs = %d", N_rungs);
    return cgk_assign_rungs(c, acc, (signed char *)rung, (signed char *)rung_jumped, n, rung_factor,
                            N_rungs);
}
extern "C" int cg_flag_rung_jumps(cg_ctx *c, const double *acc, const int8_t *rung,
                                  int8_t *rung_jumped, int64_t n, int lowest_active_rung,
                                  const double *integrals_1, double rung_factor_up,
                                  double rung_factor_down, int N_rungs, int32_t *any_out) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && integrals_1 && any_out && ((acc && rung && rung_jumped) || n == 0),
             "cg_flag_rung_jumps: null argument");
    CG_CHECK(N_rungs >= 1 && N_rungs <= 42, "cg_flag_rung_jumps: N_rungs = %d", N_rungs);
    return cgk_flag_rung_jumps(c, acc, (const signed char *)rung, (signed char *)rung_jumped, n,
                               lowest_active_rung, integrals_1, rung_factor_up, rung_factor_down,
                               N_rungs, any_out);
}
extern "C" int cg_apply_rung_jumps(cg_ctx *c, int8_t *rung, int8_t *rung_jumped, int64_t n,
                                   int N_rungs) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && ((rung && rung_jumped) || n == 0), "cg_apply_rung_jumps: null argument");
    return cgk_apply_rung_jumps(c, (signed char *)rung, (signed char *)rung_jumped, n, N_rungs);
}

extern "C" int cg_permute_rows(cg_ctx *c, const int64_t *perm, int64_t n, int ncols,
                               const void *const *src, void *const *dst, const int *row_bytes) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && ncols >= 0 && (n == 0 || ncols == 0 || (perm && src && dst && row_bytes)),
             "cg_permute_rows: null argument");
    for (int k = 0; k < ncols && n > 0; k++) {
        CG_CHECK(src[k] && dst[k] && src[k] != dst[k] && row_bytes[k] > 0,
                 "cg_permute_rows: column %d (source, destination — not in place — and row size)", k);
        const int b = row_bytes[k], al = b % 8 == 0 ? 8 : (b == 4 ? 4 : 1);
        CG_CHECK((uintptr_t)src[k] % al == 0 && (uintptr_t)dst[k] % al == 0,
                 "cg_permute_rows: column %d is not aligned to %d bytes", k, al);
    }
    return cgk_permute_rows(c, (const i64 *)perm, n, ncols, src, dst, row_bytes);
}

extern "C" int cg_substep_begin(cg_ctx *c, double *pos, const double *mom, double *dmom,
                                const int8_t *rung, int8_t *rung_jumped, int64_t n, int do_drift,
                                double dt_over_mass, int do_flag, int lowest_active_rung,
                                const double *integrals_1, double rung_factor_up,
                                double rung_factor_down, int N_rungs, int32_t *any_out,
                                int64_t *counts_after, int defer) {
    CG_CHECK(c && (n == 0 || ((!do_drift || (pos && mom)) &&
                              (!do_flag || (dmom && rung && rung_jumped)))) &&
                 (!do_flag || (integrals_1 && any_out)),
             "cg_substep_begin: null argument");
    CG_CHECK(N_rungs >= 1 && 3 * N_rungs - 1 <= CG_RUNG_TABLE_MAX, "cg_substep_begin: N_rungs = %d",
             N_rungs);
    return cgk_substep_begin(c, pos, mom, dmom, (const signed char *)rung,
                             (signed char *)rung_jumped, n, do_drift, dt_over_mass, do_flag,
                             lowest_active_rung, integrals_1, rung_factor_up, rung_factor_down,
                             N_rungs, any_out, (long long *)counts_after, defer);
}
extern "C" int cg_substep_flush(cg_ctx *c) {
    CG_CHECK(c, "cg_substep_flush: null context");
    return cgk_substep_flush(c);
}
extern "C" int cg_substep_end(cg_ctx *c, double *mom, double *dmom, int8_t *rung,
                              int8_t *rung_jumped, int64_t n, int do_apply,
                              int lowest_active_rung, const double *conversion_factors,
                              int N_rungs, int64_t *counts) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && (n == 0 || (rung && rung_jumped && (!do_apply || (mom && dmom)))) &&
                 (!do_apply || conversion_factors),
             "cg_substep_end: null argument");
    CG_CHECK(N_rungs >= 1 && 3 * N_rungs - 1 <= CG_RUNG_TABLE_MAX, "cg_substep_end: N_rungs = %d",
             N_rungs);
    return cgk_substep_end(c, mom, dmom, (signed char *)rung, (signed char *)rung_jumped, n,
                           do_apply, lowest_active_rung, conversion_factors, N_rungs,
                           (long long *)counts);
}

extern "C" int cg_shortrange_sparse(cg_ctx *c, const double *pos_r, const int64_t *active, int k,
                                   double *dmom_r, const double *pos_s, int64_t n_s,
                                   const double *table, int64_t tablesize,
                                   double r2_index_scaling, double r2_max, double factor,
                                   const double *factors, const int8_t *rung_jumped) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && pos_r && active && dmom_r && table && (pos_s || n_s == 0),
             "cg_shortrange_sparse: null argument");
    CG_CHECK(tablesize >= 2 && (double)(tablesize - 1) >= r2_max * r2_index_scaling,
             "cg_shortrange_sparse: table too short for r2_max * r2_index_scaling");
    CG_CHECK((factors == nullptr) == (rung_jumped == nullptr),
             "cg_shortrange_sparse: factors and rung_jumped go together");
    CG_CHECK(16 * r2_max <= c->p.boxsize * c->p.boxsize,
             "cg_shortrange_sparse: the range must stay below a quarter of the box");
    return cgk_shortrange_sparse(c, pos_r, (const i64 *)active, k, dmom_r, pos_s, n_s, table,
                                 r2_index_scaling, r2_max, factor, factors,
                                 (const signed char *)rung_jumped);
}

extern "C" int cg_rung_populations(cg_ctx *c, const int8_t *rung, int64_t n, int N_rungs,
                                   int64_t *counts) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && counts && (rung || n == 0), "cg_rung_populations: null argument");
    CG_CHECK(N_rungs >= 1 && N_rungs <= 64, "cg_rung_populations: N_rungs out of range");
    return cgk_rung_populations(c, (const signed char *)rung, n, N_rungs, (long long *)counts);
}

extern "C" int cg_local_info(const cg_ctx *c, int64_t info[6]) {
    CG_CHECK(c && info, "cg_local_info: null argument");
    info[0] = c->xmap.x0;
    info[1] = c->xmap.nxl;
    info[2] = c->xmap.G;
    info[3] = c->N;
    info[4] = c->pad;
    // transpose buffers: complex[P][nxl][N/P + 1][pad/2] (one unused row per layer, cg_fft.hip)
    info[5] = c->xmap.nxl * (c->N + c->p.nprocs) * c->pad;
    return 0;
}

extern "C" int64_t cg_layer_doubles(const cg_ctx *c) { return c ? c->ny * c->pad : 0; }

extern "C" int cg_layers_read(cg_ctx *c, int64_t layer0, int64_t nlayers, double *dst) {
    CG_CHECK(c && dst, "cg_layers_read: null argument");
    CG_CHECK(layer0 >= -(i64)c->xmap.G && layer0 + nlayers <= c->xmap.nxl + c->xmap.G &&
                 nlayers >= 0, "cg_layers_read: layers [%lld, %lld) outside the local buffer",
             (long long)layer0, (long long)(layer0 + nlayers));
    i64 per = c->ny * c->pad;  // whole layers, the unused row included
    CG_HIP(hipMemcpyAsync(dst, c->mesh0 + layer0 * per, 8 * per * nlayers,
                          hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

extern "C" int cg_layers_write(cg_ctx *c, int64_t layer0, int64_t nlayers, const double *src,
                               int add) {
    CG_CHECK(c && src, "cg_layers_write: null argument");
    CG_CHECK(layer0 >= -(i64)c->xmap.G && layer0 + nlayers <= c->xmap.nxl + c->xmap.G &&
                 nlayers >= 0, "cg_layers_write: layers [%lld, %lld) outside the local buffer",
             (long long)layer0, (long long)(layer0 + nlayers));
    return cgk_layers_write(c, layer0, nlayers, src, add);
}

extern "C" int cg_dist_fft_forward(cg_ctx *c, double *send_buf) {
    CG_CHECK(c && send_buf, "cg_dist_fft_forward: null argument");
    if (!c->custom_fft) return dist_generic(c, 0, (double2 *)send_buf, 0, 0.0, 0, 0.0, 0, -1);
    return cgk_fft_dist_forward(c, send_buf, 0, -1);
}
extern "C" int cg_dist_fft_forward_layers(cg_ctx *c, double *send_buf, int64_t layer0,
                                          int64_t nlayers) {
    CG_CHECK(c && send_buf, "cg_dist_fft_forward_layers: null argument");
    CG_CHECK(layer0 >= 0 && nlayers >= 1 && layer0 + nlayers <= c->xmap.nxl,
             "cg_dist_fft_forward_layers: layers [%lld, %lld) outside the %lld owned ones",
             (long long)layer0, (long long)(layer0 + nlayers), (long long)c->xmap.nxl);
    if (!c->custom_fft)
        return dist_generic(c, 0, (double2 *)send_buf, 0, 0.0, 0, 0.0, layer0, nlayers);
    return cgk_fft_dist_forward(c, send_buf, layer0, nlayers);
}
extern "C" int cg_dist_fft_xsolve(cg_ctx *c, double *buf, int deconv_order, double C,
                                  int long_range, double E) {
    CG_CHECK(c && buf, "cg_dist_fft_xsolve: null argument");
    CG_CHECK(deconv_order >= 0 && deconv_order <= 8, "cg_dist_fft_xsolve: deconv_order %d",
             deconv_order);
    if (!c->custom_fft)
        return dist_generic(c, 2, (double2 *)buf, deconv_order, C, long_range, E, 0, -1);
    return cgk_fft_dist_xsolve(c, buf, deconv_order, C, long_range, E);
}
extern "C" int cg_dist_bind_fourier(cg_ctx *c, double *buf) {
    CG_CHECK(c, "cg_dist_bind_fourier: null context");
    if (c->p.nprocs == 1 && buf == nullptr) {  // back to the in-place view
        c->four = (double2 *)c->mesh0;
        c->f_si = c->ny * (c->pad / 2);
        c->f_j0 = 0;
        c->f_nj = (int)c->N;
        return 0;
    }
    CG_CHECK(buf, "cg_dist_bind_fourier: x-slab domains need a buffer");
    // the transpose-buffer layout of cg_fft.hip: complex[N][JB + 1][cp]
    const i64 JB = c->N / c->p.nprocs;
    c->four = (double2 *)buf;
    c->f_si = (JB + 1) * (c->pad / 2);
    c->f_j0 = (int)(JB * c->p.rank);
    c->f_nj = (int)JB;
    return 0;
}
extern "C" int cg_dist_fft_x(cg_ctx *c, double *buf, int inverse) {
    CG_CHECK(c && buf, "cg_dist_fft_x: null argument");
    if (!c->custom_fft)
        return dist_generic(c, inverse ? 4 : 3, (double2 *)buf, 0, 0.0, 0, 0.0, 0, -1);
    return cgk_fft_dist_x(c, buf, inverse ? 1 : 0);
}
extern "C" int cg_emigrant_dest(cg_ctx *c, const double *pos, const double *mom,
                                const int64_t *idx, const uint32_t *count, int64_t cap,
                                double dt_over_mass, int32_t *dest, int32_t *send_counts) {
    CG_CHECK(c && send_counts && (cap == 0 || (pos && mom && idx && count && dest)),
             "cg_emigrant_dest: null argument");
    return cgk_emigrant_dest(c, pos, mom, idx, count, cap, dt_over_mass, dest, send_counts);
}
extern "C" int cg_dist_fft_backward(cg_ctx *c, const double *recv_buf) {
    CG_CHECK(c && recv_buf, "cg_dist_fft_backward: null argument");
    if (!c->custom_fft)
        return dist_generic(c, 1, (double2 *)const_cast<double *>(recv_buf), 0, 0.0, 0, 0.0, 0, -1);
    return cgk_fft_dist_backward(c, recv_buf, 0, -1);
}
extern "C" int cg_dist_fft_backward_layers(cg_ctx *c, const double *recv_buf, int64_t layer0,
                                           int64_t nlayers) {
    CG_CHECK(c && recv_buf, "cg_dist_fft_backward_layers: null argument");
    CG_CHECK(layer0 >= 0 && nlayers >= 1 && layer0 + nlayers <= c->xmap.nxl,
             "cg_dist_fft_backward_layers: layers [%lld, %lld) outside the %lld owned ones",
             (long long)layer0, (long long)(layer0 + nlayers), (long long)c->xmap.nxl);
    if (!c->custom_fft)
        return dist_generic(c, 1, (double2 *)const_cast<double *>(recv_buf), 0, 0.0, 0, 0.0,
                            layer0, nlayers);
    return cgk_fft_dist_backward(c, recv_buf, layer0, nlayers);
}
extern "C" int cg_owner_rank(cg_ctx *c, const double *pos, int64_t n, int32_t *owner_out) {
    if (c && cgk_substep_flush(c)) return 1;  // (a deferred sub-step pass first)
    CG_CHECK(c && (n == 0 || (pos && owner_out)), "cg_owner_rank: null argument");
    return cgk_owner_rank(c, pos, n, owner_out);
}

extern "C" int cg_fetch(cg_ctx *c, int which, double *out, int64_t n_doubles) {
    CG_CHECK(c && out, "cg_fetch: null argument");
    const i64 ref_doubles = c->xmap.nxl * c->N * (c->N + 2);  // the reference's slab shape
    CG_CHECK(n_doubles == ref_doubles, "cg_fetch: expected %lld doubles, got %lld",
             (long long)ref_doubles, (long long)n_doubles);
    const size_t row = 8 * (size_t)(c->N + 2);
    if (which == CG_FETCH_MESH_REAL) {
        CG_HIP(hipStreamSynchronize(c->stream));
        for (i64 layer = 0; layer < c->xmap.nxl; layer++)  // N of the ny rows of every layer
            CG_HIP(hipMemcpy2D(out + layer * c->N * (c->N + 2), row,
                               c->mesh0 + layer * c->ny * c->pad, 8 * (size_t)c->pad, row,
                               (size_t)c->N, hipMemcpyDeviceToHost));
        return 0;
    }
    if (which == CG_FETCH_MESH_FOURIER) {
        CG_CHECK(c->p.nprocs == 1, "cg_fetch: the Fourier slab fetch is single-domain only");
        if (!c->fetch_tmp) CG_HIP(hipMalloc(&c->fetch_tmp, 8 * ref_doubles));
        if (cgk_transpose_fourier(c, c->mesh, c->fetch_tmp)) return 1;
        CG_HIP(hipStreamSynchronize(c->stream));
        CG_HIP(hipMemcpy(out, c->fetch_tmp, 8 * ref_doubles, hipMemcpyDeviceToHost));
        return 0;
    }
    cg_set_error("cg_fetch: unknown selector %d", which);
    return 1;
}
