// cg_fft.hip — hand-written FP64 3-D real FFT for the Poisson solve on
// gfx950, power-of-two grid sizes 16..2048.
//
// Why not rocFFT here: for 1024^3 doubles its 3-D real plan runs 6 passes over
// the mesh per transform (3 FFT kernels + 3 transposes/post-process, measured
// 19.9 ms forward, 22.1 ms backward; its strided 1-D plans are slower still).
// The mesh is HBM-bound, so the pass count is the cost.  Here every dimension
// is ONE pass: a workgroup stages whole pencils in LDS (a 1024-point complex
// pencil is 16 KB), transforms them there (Stockham autosort, radix 4 with a
// radix-2 tail; inputs held in registers between the read and write halves of
// a pass so one LDS buffer suffices) and writes them back in place.  The
// x-pass is fused: forward transform, multiply by the Poisson/deconvolution
// factor (cg_kspace.h), inverse transform — the k-space field never travels to
// HBM.  A Poisson solve is 5 passes (z, y, x-fused, y, z) instead of 13.
//
// Layout (unchanged): real double[N][N][pad] <-> complex[N][N][pad/2] in
// place, pad >= N+2, unnormalised both ways (FFTW's convention, fft.c:34-73,
// mesh.py:4015-4022).
//   z: real<->half-complex along the contiguous dimension via the packed
//      half-length complex transform + split/merge step
//   y, x: complex pencils with stride pad/2 and N*pad/2; 4 adjacent kk per
//      workgroup (64-byte segments per row)
#include <cstdlib>

#include "cg_internal.h"
#include "cg_kspace.h"  // its factor function pins fp-contract off itself

#define CG_LAUNCH_CHECK()                                                                     \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) {                                                               \
            cg_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                           \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) {
    return make_double2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ double2 csub(double2 a, double2 b) {
    return make_double2(a.x - b.x, a.y - b.y);
}
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }

// ---------------------------------------------------------------------------
// In-LDS transform of W interleaved pencils of n = 2^LOGN complex points:
// element m of pencil w lives at lds[m*W + w].  tw[k*tws] = exp(-2 pi i k/n).
// All NT threads of the workgroup must call it.  Natural order in and out.
// ---------------------------------------------------------------------------
template <int LOGN, int W, int NT, bool INV>
__device__ __forceinline__ void fft_lds(double2 *lds, const double2 *__restrict__ tw, int tws,
                                        int tid) {
    constexpr int n = 1 << LOGN;
    int Ns = 1;
#pragma unroll 1
    for (int pass = 0; pass < LOGN / 2; pass++) {
        // radix 4
        constexpr int NB = (n / 4) * W;             // butterflies in the workgroup
        constexpr int B = (NB + NT - 1) / NT;       // per thread
        double2 v[B][4];
        double2 t1s[B];
        // the pass's twiddles come from global memory (L1/L2 resident table): issue
        // those loads first so their latency hides behind the LDS reads and the barrier
#pragma unroll
        for (int b = 0; b < B; b++) {
            int f = tid + b * NT;
            t1s[b] = make_double2(1, 0);
            if ((NB % NT == 0 || f < NB) && Ns > 1) {
                int j = f / W;
                t1s[b] = tw[(size_t)(j & (Ns - 1)) * (n / (4 * Ns)) * tws];
            }
        }
#pragma unroll
        for (int b = 0; b < B; b++) {
            int f = tid + b * NT;
            if (NB % NT == 0 || f < NB) {
                int w = f % W, j = f / W;
#pragma unroll
                for (int r = 0; r < 4; r++) v[b][r] = lds[(j + r * (n / 4)) * W + w];
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < B; b++) {
            int f = tid + b * NT;
            if (NB % NT == 0 || f < NB) {
                int w = f % W, j = f / W;
                int k = j & (Ns - 1);
                double2 x0 = v[b][0], x1 = v[b][1], x2 = v[b][2], x3 = v[b][3];
                if (Ns > 1) {
                    // twiddles exp(-+2 pi i k r/(4 Ns)), r = 1..3
                    double2 t1 = t1s[b];
                    if (INV) t1 = cconj(t1);
                    double2 t2 = cmul(t1, t1), t3 = cmul(t2, t1);
                    x1 = cmul(x1, t1);
                    x2 = cmul(x2, t2);
                    x3 = cmul(x3, t3);
                }
                double2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3),
                        d = csub(x1, x3);
                // (x1 - x3) * (-+ i)
                double2 a3 = INV ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);
                int j0 = ((j - k) << 2) + k;  // (j/Ns)*Ns*4 + k
                lds[(j0)*W + w] = cadd(a0, a2);
                lds[(j0 + Ns) * W + w] = cadd(a1, a3);
                lds[(j0 + 2 * Ns) * W + w] = csub(a0, a2);
                lds[(j0 + 3 * Ns) * W + w] = csub(a1, a3);
            }
        }
        __syncthreads();
        Ns <<= 2;
    }
    if (LOGN & 1) {
        // radix-2 tail, Ns = n/2
        constexpr int NB = (n / 2) * W;
        constexpr int B = (NB + NT - 1) / NT;
        double2 v[B][2];
#pragma unroll
        for (int b = 0; b < B; b++) {
            int f = tid + b * NT;
            if (NB % NT == 0 || f < NB) {
                int w = f % W, j = f / W;
                v[b][0] = lds[j * W + w];
                v[b][1] = lds[(j + n / 2) * W + w];
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < B; b++) {
            int f = tid + b * NT;
            if (NB % NT == 0 || f < NB) {
                int w = f % W, j = f / W;  // k = j (Ns = n/2)
                double2 t1 = tw[(size_t)j * tws];
                if (INV) t1 = cconj(t1);
                double2 x1 = cmul(v[b][1], t1);
                lds[j * W + w] = cadd(v[b][0], x1);
                lds[(j + n / 2) * W + w] = csub(v[b][0], x1);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// z pass, forward: N reals -> N/2+1 complex per (i,j) row, in place.
// One row per workgroup.  tw[k] = exp(-2 pi i k/N), k < N.
// ---------------------------------------------------------------------------
template <int LOGN /* log2 N */, int NT>
__global__ __launch_bounds__(NT) void k_fft_z_forward(double *__restrict__ mesh, i64 pad,
                                                      const double2 *__restrict__ tw) {
    constexpr int N = 1 << LOGN, H = N / 2;
    __shared__ double2 lds[H];
    double2 *row = (double2 *)(mesh + (i64)blockIdx.x * pad);
    const int tid = threadIdx.x;
    {
        constexpr int PER = (H + NT - 1) / NT;
        double2 v[PER];
#pragma unroll
        for (int r = 0; r < PER; r++) {
            int m = tid + r * NT;
            if (H % NT == 0 || m < H) v[r] = row[m];  // (x[2m], x[2m+1])
        }
#pragma unroll
        for (int r = 0; r < PER; r++) {
            int m = tid + r * NT;
            if (H % NT == 0 || m < H) lds[m] = v[r];
        }
    }
    __syncthreads();
    fft_lds<LOGN - 1, 1, NT, false>(lds, tw, 2, tid);
    // split: X[k] = (Z[k] + conj Z[H-k])/2 - (i/2) w^k (Z[k] - conj Z[H-k]), k = 0..H
    for (int k = tid; k <= H; k += NT) {
        double2 zk = lds[k & (H - 1)], zc = cconj(lds[(H - k) & (H - 1)]);
        double2 s = cadd(zk, zc), d = csub(zk, zc);
        double2 wd = cmul(tw[k % N], d);  // k = H < N always
        // -(i/2)*wd = (wd.y/2, -wd.x/2)
        row[k] = make_double2(0.5 * s.x + 0.5 * wd.y, 0.5 * s.y - 0.5 * wd.x);
    }
}

// z pass, backward: N/2+1 complex -> N reals (unnormalised), in place.
template <int LOGN, int NT>
__global__ __launch_bounds__(NT) void k_fft_z_backward(double *__restrict__ mesh, i64 pad,
                                                       const double2 *__restrict__ tw) {
    constexpr int N = 1 << LOGN, H = N / 2;
    __shared__ double2 lds[H];
    double2 *row = (double2 *)(mesh + (i64)blockIdx.x * pad);
    const int tid = threadIdx.x;
    // merge: Z[k] = (X[k] + conj X[H-k]) + i conj(w^k) (X[k] - conj X[H-k]), k = 0..H-1
    for (int k = tid; k < H; k += NT) {
        double2 xk = row[k], xc = cconj(row[H - k]);
        double2 s = cadd(xk, xc), d = csub(xk, xc);
        double2 wd = cmul(cconj(tw[k]), d);
        // + i*wd = (-wd.y, wd.x)
        lds[k] = make_double2(s.x - wd.y, s.y + wd.x);
    }
    __syncthreads();
    fft_lds<LOGN - 1, 1, NT, true>(lds, tw, 2, tid);
    for (int m = tid; m < H; m += NT) row[m] = lds[m];
}

// ---------------------------------------------------------------------------
// strided pass (y or x): n = N points per pencil, W = 4 adjacent kk per
// workgroup.  MODE 0: forward, 1: backward, 2: forward * k-space factor *
// backward (the fused Poisson pass).
// Workgroup -> (outer index o, kk block).  Element m of the pencil is read from
//   src + o*src.ostride + kk0 + (m >> src.sh)*src.blk + (m & mask)*src.es
// and written to the same expression on dst: sh = 31 gives the plain strided
// pencil; a finite sh addresses the pencil in blocks of 2^sh points, which is
// the layout of the all-to-all transpose buffers of the multi-GPU path (the
// pack / unpack of the transpose is fused into this pass).
// (Persistent variants that prefetch the next tile into registers while the current
// one is transformed were measured twice: with 4 pencils 4.57 vs 4.60 ms per pass — the
// registers cost the second resident workgroup what the prefetch gains; with 8 pencils
// the 8 staged double2 per lane on top of the butterflies spill (5.8 ms).  Not used.)
// For the factor, the pencil index m and the outer index o + o_off are the two
// full dimensions (the expression is symmetric in them).
// ---------------------------------------------------------------------------
struct PencilMap {
    i64 ostride;  // complex elements between consecutive outer indices
    i64 es;       // between consecutive pencil points inside a block
    i64 blk;      // between consecutive blocks
    int sh;       // log2(points per block); 31 = one block
};
__device__ __forceinline__ i64 pencil_off(const PencilMap &pm, int m) {
    return (i64)(m >> pm.sh) * pm.blk + (i64)((unsigned)m & ((1u << pm.sh) - 1u)) * pm.es;
}

template <int LOGN, int NT, int MODE, int W>
__global__ __launch_bounds__(NT) void k_fft_strided(const double2 *__restrict__ src,
                                                    double2 *__restrict__ dst, PencilMap smap,
                                                    PencilMap dmap, int nkb, i64 o_off,
                                                    const double2 *__restrict__ tw,
                                                    KspaceParams P) {
    constexpr int N = 1 << LOGN;
    extern __shared__ double2 lds_dyn[];
    double2 *lds = lds_dyn;
    const int tid = threadIdx.x;
    const i64 o = blockIdx.x / nkb;
    const int kb = blockIdx.x - (int)o * nkb;
    const int kk0 = kb * W;
    const int nk = N / 2 + 1;  // valid kk: 0..N/2
    const double2 *sbase = src + o * smap.ostride + kk0;
    double2 *dbase = dst + o * dmap.ostride + kk0;
    constexpr int TOT = N * W;
    constexpr int PER = (TOT + NT - 1) / NT;
    {
        double2 v[PER];
#pragma unroll
        for (int r = 0; r < PER; r++) {
            int f = tid + r * NT;
            int w = f % W, m = f / W;
            bool ok = (TOT % NT == 0 || f < TOT) && (kk0 + w < nk);
            v[r] = ok ? sbase[pencil_off(smap, m) + w] : make_double2(0, 0);
        }
#pragma unroll
        for (int r = 0; r < PER; r++) {
            int f = tid + r * NT;
            if (TOT % NT == 0 || f < TOT) lds[f] = v[r];
        }
    }
    __syncthreads();
    if (MODE == 0 || MODE == 2) fft_lds<LOGN, W, NT, false>(lds, tw, 1, tid);
    if (MODE == 2) {
        // NT is a multiple of W: a lane's kk (= kk0 + tid % W) and the outer index are the
        // same for all its elements; only the pencil index m changes
        const int wl = tid % W;
        const int kkl = kk0 + wl;
        if (kkl < nk) {
            const KspaceFixed F = kspace_fix(P, N, o + o_off, kkl);
#pragma unroll
            for (int r = 0; r < PER; r++) {
                int f = tid + r * NT;
                if (TOT % NT == 0 || f < TOT) {
                    double fac = kspace_factor_fixed(P, F, N, f / W);
                    double2 x = lds[f];
                    lds[f] = make_double2(x.x * fac, x.y * fac);
                }
            }
        }
        __syncthreads();
    }
    if (MODE == 1 || MODE == 2) fft_lds<LOGN, W, NT, true>(lds, tw, 1, tid);
#pragma unroll
    for (int r = 0; r < PER; r++) {
        int f = tid + r * NT;
        int w = f % W, m = f / W;
        if ((TOT % NT == 0 || f < TOT) && (kk0 + w < nk)) dbase[pencil_off(dmap, m) + w] = lds[f];
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <int LOGN>
static int run_z(cg_ctx *c, bool inverse) {
    constexpr int N = 1 << LOGN;
    constexpr int NT = (N / 8) < 64 ? 64 : ((N / 8) > 256 ? 256 : (N / 8));
    unsigned rows = (unsigned)(c->xmap.nxl * c->N);  // owned layers only
    if (!inverse)
        hipLaunchKernelGGL((k_fft_z_forward<LOGN, NT>), dim3(rows), dim3(NT), 0, c->stream,
                           c->mesh0, c->pad, (const double2 *)c->fft_tw);
    else
        hipLaunchKernelGGL((k_fft_z_backward<LOGN, NT>), dim3(rows), dim3(NT), 0, c->stream,
                           c->mesh0, c->pad, (const double2 *)c->fft_tw);
    CG_LAUNCH_CHECK();
    return 0;
}

template <int LOGN, int MODE, int W>
static int run_strided_w(cg_ctx *c, const double2 *src, double2 *dst, PencilMap smap,
                         PencilMap dmap, i64 nouter, i64 o_off, const KspaceParams &P) {
    constexpr int N = 1 << LOGN;
    // lanes: N*W/8 points per lane (two radix-4 butterflies per pass), 64..1024
    constexpr int NTW = N * W / 8;
    constexpr int NT = NTW < 64 ? 64 : (NTW > 1024 ? 1024 : NTW);
    const int nkb = (int)((c->N / 2 + 1 + W - 1) / W);
    size_t lds = sizeof(double2) * N * W;
    auto kern = k_fft_strided<LOGN, NT, MODE, W>;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        CG_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(nouter * nkb)), dim3(NT), lds, c->stream, src, dst,
                       smap, dmap, nkb, o_off, (const double2 *)c->fft_tw, P);
    CG_LAUNCH_CHECK();
    return 0;
}

// W adjacent kk per workgroup.  Measured at 1024^3 (ms per y pass / fused x pass):
//   W = 2 (32 B row segments, 4 workgroups per CU)  9.4 / 9.6
//   W = 4 (64 B, 2 per CU)                          4.6 / 7.9
//   W = 8 (128 B = one full cache line, 1 per CU)   3.95 / 7.4
// The row-segment width, not the occupancy, decides: default 8 (N <= 1024; 8 pencils of
// 2048 points exceed the 160 KB LDS, so 4 there).  CONCEPT_GPU_FFT_W overrides for A/B.
template <int LOGN, int MODE>
static int run_strided(cg_ctx *c, const double2 *src, double2 *dst, PencilMap smap,
                       PencilMap dmap, i64 nouter, i64 o_off, const KspaceParams &P) {
    static int w = 0;
    if (!w) {
        const char *env = getenv("CONCEPT_GPU_FFT_W");
        w = env ? atoi(env) : 8;
        if (w != 2 && w != 4 && w != 8) w = 8;
    }
    if (w == 2) return run_strided_w<LOGN, MODE, 2>(c, src, dst, smap, dmap, nouter, o_off, P);
    if (w == 8 && LOGN <= 10)  // 8 pencils of 2048 points would not fit the 160 KB LDS
        return run_strided_w<(LOGN <= 10 ? LOGN : 10), MODE, 8>(c, src, dst, smap, dmap, nouter,
                                                                o_off, P);
    return run_strided_w<LOGN, MODE, 4>(c, src, dst, smap, dmap, nouter, o_off, P);
}

static PencilMap plain_map(i64 ostride, i64 es) { return PencilMap{ostride, es, 0, 31}; }

template <int LOGN>
static int fft3d(cg_ctx *c, int what, const KspaceParams &P) {
    // single domain.  what: 0 forward, 1 backward, 2 forward + kernel + backward (fused x)
    const i64 cp = c->pad / 2, N = c->N;
    double2 *m = (double2 *)c->mesh0;
    PencilMap ymap = plain_map(cp * N, cp), xmap = plain_map(cp, cp * N);
    if (what == 0) {
        if (run_z<LOGN>(c, false)) return 1;
        if (run_strided<LOGN, 0>(c, m, m, ymap, ymap, N, 0, P)) return 1;
        return run_strided<LOGN, 0>(c, m, m, xmap, xmap, N, 0, P);
    }
    if (what == 1) {
        if (run_strided<LOGN, 1>(c, m, m, xmap, xmap, N, 0, P)) return 1;
        if (run_strided<LOGN, 1>(c, m, m, ymap, ymap, N, 0, P)) return 1;
        return run_z<LOGN>(c, true);
    }
    auto mark = [&](int i) {
        if (c->pass_events) (void)hipEventRecord(c->pass_events[i], c->stream);
    };
    mark(0);
    if (run_z<LOGN>(c, false)) return 1;
    mark(1);
    if (run_strided<LOGN, 0>(c, m, m, ymap, ymap, N, 0, P)) return 1;
    mark(2);
    if (run_strided<LOGN, 2>(c, m, m, xmap, xmap, N, 0, P)) return 1;
    mark(3);
    if (run_strided<LOGN, 1>(c, m, m, ymap, ymap, N, 0, P)) return 1;
    mark(4);
    int rc = run_z<LOGN>(c, true);
    mark(5);
    return rc;
}

// x-slab domains (one per GPU).  The local slab complex[nxl][N][cp] is
// transformed along z and y here; the y pass writes straight into the
// all-to-all send buffer, blocked by destination domain:
//   send[q][i_local][j - q*JB][kk],  JB = N/P
// so that after the exchange every domain holds complex[i (N)][j_local (JB)][cp]
// and the x pass is again a plain strided pencil (stride JB*cp).  The way back
// mirrors it: the inverse y pass reads the blocked layout.
template <int LOGN>
static int fft_dist(cg_ctx *c, int what, double2 *buf, const KspaceParams &P) {
    const i64 cp = c->pad / 2, N = c->N, nxl = c->xmap.nxl, JB = N / c->p.nprocs;
    int sh = 0;
    while ((1 << sh) < JB) sh++;
    double2 *m = (double2 *)c->mesh0;
    PencilMap ymap = plain_map(cp * N, cp);
    PencilMap bmap{JB * cp, cp, nxl * JB * cp, sh};
    if (what == 0) {  // forward z, forward y -> send buffer
        if (run_z<LOGN>(c, false)) return 1;
        return run_strided<LOGN, 0>(c, m, buf, ymap, bmap, nxl, 0, P);
    }
    if (what == 2) {  // fused x pass on complex[N][JB][cp]
        PencilMap xmap = plain_map(cp, JB * cp);
        return run_strided<LOGN, 2>(c, buf, buf, xmap, xmap, JB, (i64)c->p.rank * JB, P);
    }
    // backward y from the returned buffer, backward z
    if (run_strided<LOGN, 1>(c, buf, m, bmap, ymap, nxl, 0, P)) return 1;
    return run_z<LOGN>(c, true);
}

bool cgk_fft_supported(i64 N) { return N >= 16 && N <= 2048 && (N & (N - 1)) == 0; }

#define CG_FFT_DISPATCH(FN, ...)                                                              \
    switch (c->N) {                                                                           \
        case 16: return FN<4>(__VA_ARGS__);                                                   \
        case 32: return FN<5>(__VA_ARGS__);                                                   \
        case 64: return FN<6>(__VA_ARGS__);                                                   \
        case 128: return FN<7>(__VA_ARGS__);                                                  \
        case 256: return FN<8>(__VA_ARGS__);                                                  \
        case 512: return FN<9>(__VA_ARGS__);                                                  \
        case 1024: return FN<10>(__VA_ARGS__);                                                \
        case 2048: return FN<11>(__VA_ARGS__);                                                \
    }                                                                                         \
    cg_set_error("grid size %lld not supported by the hand-written FFT", (long long)c->N);    \
    return 1;

int cgk_fft(cg_ctx *c, int what, int deconv_order, double C, int long_range, double E) {
    KspaceParams P{c->ktab_n, c->ktab_s, deconv_order, long_range, C, E};
    CG_FFT_DISPATCH(fft3d, c, what, P)
}
int cgk_fft_dist_forward(cg_ctx *c, double *send_buf) {
    KspaceParams P{c->ktab_n, c->ktab_s, 0, 0, 0.0, 0.0};
    CG_FFT_DISPATCH(fft_dist, c, 0, (double2 *)send_buf, P)
}
int cgk_fft_dist_xsolve(cg_ctx *c, double *buf, int deconv_order, double C, int long_range,
                        double E) {
    KspaceParams P{c->ktab_n, c->ktab_s, deconv_order, long_range, C, E};
    CG_FFT_DISPATCH(fft_dist, c, 2, (double2 *)buf, P)
}
int cgk_fft_dist_backward(cg_ctx *c, const double *recv_buf) {
    KspaceParams P{c->ktab_n, c->ktab_s, 0, 0, 0.0, 0.0};
    CG_FFT_DISPATCH(fft_dist, c, 1, (double2 *)recv_buf, P)
}
