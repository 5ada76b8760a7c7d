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
// Stockham autosort passes: a butterfly of radix R at position j (k = j mod Ns)
// reads points j + r*n/R, multiplies by exp(-+2 pi i k r/(R Ns)) and writes
// (j - k)*R + k + r*Ns.  The inputs sit in registers between the read and the
// write half of a pass, so one LDS buffer suffices (two barriers per pass).
// Schedules: R16 = false: radix 4 (+ radix-2 tail); R16 = true: radix 16 passes,
// then radix 4 / radix 2 for the remaining bits (1024 = 16*16*4: 3 passes
// instead of 5, i.e. 6 instead of 10 LDS round trips).
// ---------------------------------------------------------------------------
template <int LOGN, int W, int NT, bool INV>
__device__ __forceinline__ void fft_pass_r4(double2 *lds, const double2 *__restrict__ tw,
                                            int tws, int tid, int Ns) {
    constexpr int n = 1 << LOGN;
    constexpr int NB = (n / 4) * W;        // butterflies in the workgroup
    constexpr int B = (NB + NT - 1) / NT;  // per thread
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
            double2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), d = csub(x1, x3);
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
}

// radix-2 pass with Ns = n/2 (the last pass of an odd LOGN)
template <int LOGN, int W, int NT, bool INV>
__device__ __forceinline__ void fft_pass_r2_last(double2 *lds, const double2 *__restrict__ tw,
                                                 int tws, int tid) {
    constexpr int n = 1 << LOGN;
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

// x * exp(-+2 pi i M/16) for the M the 4x4 decomposition of a 16-point butterfly needs
template <int M, bool INV>
__device__ __forceinline__ double2 mul_w16(double2 x) {
    constexpr double C = 0.92387953251128674, S = 0.38268343236508977,
                     R = 0.70710678118654752;
    // forward factor (c, -s); inverse its conjugate
    if (M == 0) return x;
    if (M == 4) return INV ? make_double2(-x.y, x.x) : make_double2(x.y, -x.x);
    if (M == 2)
        return INV ? make_double2(R * (x.x - x.y), R * (x.x + x.y))
                   : make_double2(R * (x.x + x.y), R * (x.y - x.x));
    if (M == 6)
        return INV ? make_double2(-R * (x.x + x.y), R * (x.x - x.y))
                   : make_double2(R * (x.y - x.x), -R * (x.x + x.y));
    double c = M == 1 ? C : (M == 3 ? S : -C);   // M = 1, 3, 9
    double s = M == 1 ? S : (M == 3 ? C : -S);   // sin(2 pi M/16)
    double2 t = make_double2(c, INV ? s : -s);
    return cmul(x, t);
}
template <bool INV>
__device__ __forceinline__ void bfly4(double2 &x0, double2 &x1, double2 &x2, double2 &x3) {
    double2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), d = csub(x1, x3);
    double2 a3 = INV ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);
    x0 = cadd(a0, a2);
    x1 = cadd(a1, a3);
    x2 = csub(a0, a2);
    x3 = csub(a1, a3);
}
// 16-point DFT in registers, n = a + 4m, k = b + 4c:
//   X[b+4c] = sum_a w4^(ac) [ w16^(ab) sum_m w4^(mb) x[a+4m] ]
// in place: after the first stage v[a+4b] holds the inner sum for (a, b); after the second
// v[b'+4c] with b' the register row — the result index is returned by out16().
template <bool INV>
__device__ __forceinline__ void fft16_regs(double2 (&v)[16]) {
#pragma unroll
    for (int a = 0; a < 4; a++) bfly4<INV>(v[a], v[a + 4], v[a + 8], v[a + 12]);
    // v[a + 4b] *= w16^(a b)
    v[5] = mul_w16<1, INV>(v[5]);
    v[6] = mul_w16<2, INV>(v[6]);
    v[7] = mul_w16<3, INV>(v[7]);
    v[9] = mul_w16<2, INV>(v[9]);
    v[10] = mul_w16<4, INV>(v[10]);
    v[11] = mul_w16<6, INV>(v[11]);
    v[13] = mul_w16<3, INV>(v[13]);
    v[14] = mul_w16<6, INV>(v[14]);
    v[15] = mul_w16<9, INV>(v[15]);
    // per b: DFT over a of v[a + 4b] -> output c lands in register c + 4b = X[b + 4c]
#pragma unroll
    for (int b = 0; b < 4; b++) bfly4<INV>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);
}

template <int LOGN, int W, int NT, bool INV>
__device__ __forceinline__ void fft_pass_r16(double2 *lds, const double2 *__restrict__ tw,
                                             int tws, int tid, int Ns) {
    constexpr int n = 1 << LOGN;
    constexpr int NB = (n / 16) * W;
    constexpr int B = (NB + NT - 1) / NT;
#pragma unroll
    for (int b = 0; b < B; b++) {
        static_assert(B == 1, "radix-16 passes are launched with one butterfly per lane");
        int f = tid + b * NT;
        const bool on = (NB % NT == 0 || f < NB);
        int w = f % W, j = f / W;
        int k = j & (Ns - 1);
        double2 t1 = make_double2(1, 0);
        if (on && Ns > 1) t1 = tw[(size_t)k * (n / (16 * Ns)) * tws];
        double2 v[16];
        if (on) {
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = lds[(j + r * (n / 16)) * W + w];
        }
        __syncthreads();
        if (on) {
            if (Ns > 1) {
                if (INV) t1 = cconj(t1);
                // powers t^r by a balanced product tree (depth 4)
                double2 t2 = cmul(t1, t1), t3 = cmul(t2, t1), t4 = cmul(t2, t2);
                double2 t5 = cmul(t3, t2), t6 = cmul(t3, t3), t7 = cmul(t4, t3);
                double2 t8 = cmul(t4, t4);
                v[1] = cmul(v[1], t1);
                v[2] = cmul(v[2], t2);
                v[3] = cmul(v[3], t3);
                v[4] = cmul(v[4], t4);
                v[5] = cmul(v[5], t5);
                v[6] = cmul(v[6], t6);
                v[7] = cmul(v[7], t7);
                v[8] = cmul(v[8], t8);
                v[9] = cmul(v[9], cmul(t8, t1));
                v[10] = cmul(v[10], cmul(t8, t2));
                v[11] = cmul(v[11], cmul(t8, t3));
                v[12] = cmul(v[12], cmul(t8, t4));
                v[13] = cmul(v[13], cmul(t8, t5));
                v[14] = cmul(v[14], cmul(t8, t6));
                v[15] = cmul(v[15], cmul(t8, t7));
            }
            fft16_regs<INV>(v);
            int j0 = ((j - k) << 4) + k;  // (j/Ns)*Ns*16 + k
            // register c + 4b holds X[b + 4c]
#pragma unroll
            for (int bb = 0; bb < 4; bb++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++)
                    lds[(j0 + (bb + 4 * cc) * Ns) * W + w] = v[cc + 4 * bb];
        }
        __syncthreads();
    }
}

template <int LOGN, int W, int NT, bool INV, bool R16 = false>
__device__ __forceinline__ void fft_lds(double2 *lds, const double2 *__restrict__ tw, int tws,
                                        int tid) {
    int Ns = 1;
    if constexpr (R16) {
#pragma unroll 1
        for (int pass = 0; pass < LOGN / 4; pass++) {
            fft_pass_r16<LOGN, W, NT, INV>(lds, tw, tws, tid, Ns);
            Ns <<= 4;
        }
        if ((LOGN & 3) >= 2) fft_pass_r4<LOGN, W, NT, INV>(lds, tw, tws, tid, Ns);
    } else {
#pragma unroll 1
        for (int pass = 0; pass < LOGN / 2; pass++) {
            fft_pass_r4<LOGN, W, NT, INV>(lds, tw, tws, tid, Ns);
            Ns <<= 2;
        }
    }
    if (LOGN & 1) fft_pass_r2_last<LOGN, W, NT, INV>(lds, tw, tws, tid);
}

// ---------------------------------------------------------------------------
// z pass, forward: N reals -> N/2+1 complex per (i,j) row, in place.
// One row per workgroup.  tw[k] = exp(-2 pi i k/N), k < N.
// ---------------------------------------------------------------------------
template <int LOGN /* log2 N */, int NT>
__global__ __launch_bounds__(NT) void k_fft_z_forward(double *__restrict__ mesh, i64 ny,
                                                      i64 pad, const double2 *__restrict__ tw) {
    constexpr int N = 1 << LOGN, H = N / 2;
    __shared__ double2 lds[H];
    // blockIdx.x = layer*N + j; a layer is ny >= N rows (cg_ctx::ny)
    double2 *row = (double2 *)(mesh + ((i64)(blockIdx.x >> LOGN) * ny + (blockIdx.x & (N - 1))) * pad);
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
__global__ __launch_bounds__(NT) void k_fft_z_backward(double *__restrict__ mesh, i64 ny,
                                                       i64 pad, const double2 *__restrict__ tw) {
    constexpr int N = 1 << LOGN, H = N / 2;
    __shared__ double2 lds[H];
    // rows from the end (workgroups are dispatched in ascending order): the inverse y pass
    // before this one finished there
    const unsigned b = gridDim.x - 1 - blockIdx.x;
    double2 *row = (double2 *)(mesh + ((i64)(b >> LOGN) * ny + (b & (N - 1))) * pad);
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

template <int LOGN, int NT, int MODE, int W, bool R16>
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
    if (MODE == 0 || MODE == 2) fft_lds<LOGN, W, NT, false, R16>(lds, tw, 1, tid);
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
                    double fac = kspace_factor_fixed(P, F, N, f / W, P.tab_n[f / W], P.tab_s[f / W]);
                    double2 x = lds[f];
                    lds[f] = make_double2(x.x * fac, x.y * fac);
                }
            }
        }
        __syncthreads();
    }
    if (MODE == 1 || MODE == 2) fft_lds<LOGN, W, NT, true, R16>(lds, tw, 1, tid);
#pragma unroll
    for (int r = 0; r < PER; r++) {
        int f = tid + r * NT;
        int w = f % W, m = f / W;
        if ((TOT % NT == 0 || f < TOT) && (kk0 + w < nk)) dbase[pencil_off(dmap, m) + w] = lds[f];
    }
}

// Persistent form of the same pass: one workgroup per CU walks tiles t, t + G, ... and
// loads tile t + G into registers before transforming tile t in LDS, so the CU's memory
// pipe stays busy during the transform (with one 128 KB tile per CU nothing else would
// overlap them).  Needs the 256-VGPR budget of <= 512 lanes (radix-16 schedule: 126 VGPRs
// for the transform + 64 staged) and no vmcnt-tracked access between the prefetch and
// its use — the twiddle and k-space tables are therefore copied to LDS once per workgroup
// (vector memory returns in order: a global twiddle load would wait for the prefetch).
typedef double d2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// Tile transform with register input and output, for NT = n*W/16 lanes (one radix-16
// butterfly per lane): lane (ml = tid / W, wl = tid % W) enters holding the 16 points
// ml + r*n/16 of pencil wl — exactly the operands of its first-pass butterfly AND what it
// loaded from global memory — and leaves holding the same 16 positions of the result —
// exactly what the last pass's butterflies of that lane produce (q = b + r*B below).
// So the first pass never reads LDS and the last never writes it: a 1024-point transform
// (16*16*4) costs 2 LDS writes + 2 LDS reads of the tile instead of 4 + 4 (LDS stores run
// at ~79 B/clk/CU against 256 B/clk for loads: they are the transform's cost), and in the
// fused pass the k-space factor is applied in registers between the last forward pass and
// the first inverse pass.
// ---------------------------------------------------------------------------
template <int LOGN, int W, int NT, bool INV, bool IN_REGS, bool OUT_REGS, int Ns>
__device__ __forceinline__ void tile_r16(double2 (&x)[16], double2 *lds,
                                         const double2 *__restrict__ tw, int tid) {
    constexpr int n = 1 << LOGN;
    static_assert((n / 16) * W == NT, "one radix-16 butterfly per lane");
    const int w = tid % W, j = tid / W;
    const int k = j & (Ns - 1);
    double2 t1 = make_double2(1, 0);
    if (!IN_REGS) {
        t1 = tw[k * (n / (16 * Ns))];
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = lds[(j + r * (n / 16)) * W + w];
    }
    __syncthreads();  // LDS reads of this pass (or of the previous transform) are done
    if (!IN_REGS) {
        if (INV) t1 = cconj(t1);
        double2 t2 = cmul(t1, t1), t3 = cmul(t2, t1), t4 = cmul(t2, t2);
        double2 t5 = cmul(t3, t2), t6 = cmul(t3, t3), t7 = cmul(t4, t3);
        double2 t8 = cmul(t4, t4);
        x[1] = cmul(x[1], t1);
        x[2] = cmul(x[2], t2);
        x[3] = cmul(x[3], t3);
        x[4] = cmul(x[4], t4);
        x[5] = cmul(x[5], t5);
        x[6] = cmul(x[6], t6);
        x[7] = cmul(x[7], t7);
        x[8] = cmul(x[8], t8);
        x[9] = cmul(x[9], cmul(t8, t1));
        x[10] = cmul(x[10], cmul(t8, t2));
        x[11] = cmul(x[11], cmul(t8, t3));
        x[12] = cmul(x[12], cmul(t8, t4));
        x[13] = cmul(x[13], cmul(t8, t5));
        x[14] = cmul(x[14], cmul(t8, t6));
        x[15] = cmul(x[15], cmul(t8, t7));
    }
    fft16_regs<INV>(x);  // register c + 4b holds X[b + 4c]
    if (OUT_REGS) {
        // last pass (Ns = n/16): X[q] is point j + q*n/16 — reorder in registers
        double2 y[16];
#pragma unroll
        for (int bb = 0; bb < 4; bb++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) y[bb + 4 * cc] = x[cc + 4 * bb];
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = y[r];
    } else {
        const int j0 = ((j - k) << 4) + k;
#pragma unroll
        for (int bb = 0; bb < 4; bb++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
                lds[(j0 + (bb + 4 * cc) * Ns) * W + w] = x[cc + 4 * bb];
        __syncthreads();
    }
}

// radix-4 pass, 4 butterflies per lane (j = tid/W + b*n/16); OUT_REGS: last pass (Ns = n/4),
// output r of butterfly b is point tid/W + (b + 4r)*n/16
template <int LOGN, int W, int NT, bool INV, bool OUT_REGS, int Ns>
__device__ __forceinline__ void tile_r4(double2 (&x)[16], double2 *lds,
                                        const double2 *__restrict__ tw, int tid) {
    constexpr int n = 1 << LOGN;
    const int w = tid % W, jl = tid / W;
    double2 t1s[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        int j = jl + b * (n / 16);
        t1s[b] = tw[(j & (Ns - 1)) * (n / (4 * Ns))];
#pragma unroll
        for (int r = 0; r < 4; r++) x[4 * b + r] = lds[(j + r * (n / 4)) * W + w];
    }
    __syncthreads();
    double2 y[16];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        int j = jl + b * (n / 16);
        int k = j & (Ns - 1);
        double2 t1 = INV ? cconj(t1s[b]) : t1s[b];
        double2 t2 = cmul(t1, t1), t3 = cmul(t2, t1);
        double2 x0 = x[4 * b], x1 = cmul(x[4 * b + 1], t1), x2 = cmul(x[4 * b + 2], t2),
                x3 = cmul(x[4 * b + 3], t3);
        bfly4<INV>(x0, x1, x2, x3);
        if (OUT_REGS) {
            y[b] = x0;
            y[b + 4] = x1;
            y[b + 8] = x2;
            y[b + 12] = x3;
        } else {
            int j0 = ((j - k) << 2) + k;
            lds[(j0)*W + w] = x0;
            lds[(j0 + Ns) * W + w] = x1;
            lds[(j0 + 2 * Ns) * W + w] = x2;
            lds[(j0 + 3 * Ns) * W + w] = x3;
        }
    }
    if (OUT_REGS) {
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = y[r];
    } else {
        __syncthreads();
    }
}

// radix-2 last pass (Ns = n/2), 8 butterflies per lane, always to registers:
// output r of butterfly b is point tid/W + (b + 8r)*n/16
template <int LOGN, int W, int NT, bool INV>
__device__ __forceinline__ void tile_r2_last(double2 (&x)[16], double2 *lds,
                                             const double2 *__restrict__ tw, int tid) {
    constexpr int n = 1 << LOGN;
    const int w = tid % W, jl = tid / W;
    double2 a[8], c[8], t[8];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        int j = jl + b * (n / 16);
        t[b] = tw[j];
        a[b] = lds[j * W + w];
        c[b] = lds[(j + n / 2) * W + w];
    }
#pragma unroll
    for (int b = 0; b < 8; b++) {
        double2 t1 = INV ? cconj(t[b]) : t[b];
        double2 x1 = cmul(c[b], t1);
        x[b] = cadd(a[b], x1);
        x[b + 8] = csub(a[b], x1);
    }
}

// radix-8 last pass (Ns = n/8), 2 butterflies per lane (j = tid/W + b*n/16), always to
// registers: output q of butterfly b is point tid/W + (b + 2q)*n/16.  Replaces a radix-4 pass
// through LDS followed by the radix-2 pass when log2(n) mod 4 = 3 (2048 = 16*16*8: two LDS
// exchanges of the tile instead of three).
template <int LOGN, int W, int NT, bool INV>
__device__ __forceinline__ void tile_r8_last(double2 (&x)[16], double2 *lds,
                                             const double2 *__restrict__ tw, int tid) {
    constexpr int n = 1 << LOGN;
    const int w = tid % W, jl = tid / W;
    const double h = 0.70710678118654752440;  // cos(pi/4)
    double2 v[2][8], t[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        int j = jl + b * (n / 16);
        t[b] = tw[j];
#pragma unroll
        for (int r = 0; r < 8; r++) v[b][r] = lds[(j + r * (n / 8)) * W + w];
    }
#pragma unroll
    for (int b = 0; b < 2; b++) {
        double2 t1 = INV ? cconj(t[b]) : t[b];
        double2 t2 = cmul(t1, t1), t3 = cmul(t2, t1), t4 = cmul(t2, t2);
        double2 x0 = v[b][0], x1 = cmul(v[b][1], t1), x2 = cmul(v[b][2], t2),
                x3 = cmul(v[b][3], t3), x4 = cmul(v[b][4], t4), x5 = cmul(v[b][5], cmul(t4, t1)),
                x6 = cmul(v[b][6], cmul(t4, t2)), x7 = cmul(v[b][7], cmul(t4, t3));
        // decimation in frequency: sums feed the even outputs, differences times w8^r the odd
        double2 a0 = cadd(x0, x4), a1 = cadd(x1, x5), a2 = cadd(x2, x6), a3 = cadd(x3, x7);
        double2 b0 = csub(x0, x4), d1 = csub(x1, x5), d2 = csub(x2, x6), d3 = csub(x3, x7);
        // w8 = exp(-+ 2 pi i/8): w8 = h(1 -+ i), w8^2 = -+ i, w8^3 = h(-1 -+ i)
        double2 b1 = INV ? make_double2(h * (d1.x - d1.y), h * (d1.x + d1.y))
                         : make_double2(h * (d1.x + d1.y), h * (d1.y - d1.x));
        double2 b2 = INV ? make_double2(-d2.y, d2.x) : make_double2(d2.y, -d2.x);
        double2 b3 = INV ? make_double2(-h * (d3.x + d3.y), h * (d3.x - d3.y))
                         : make_double2(h * (d3.y - d3.x), -h * (d3.x + d3.y));
        bfly4<INV>(a0, a1, a2, a3);  // X[0], X[2], X[4], X[6]
        bfly4<INV>(b0, b1, b2, b3);  // X[1], X[3], X[5], X[7]
        x[b + 0] = a0;
        x[b + 2] = b0;
        x[b + 4] = a1;
        x[b + 6] = b1;
        x[b + 8] = a2;
        x[b + 10] = b2;
        x[b + 12] = a3;
        x[b + 14] = b3;
    }
}

template <int LOGN, int W, int NT, bool INV>
__device__ __forceinline__ void fft_tile(double2 (&x)[16], double2 *lds,
                                         const double2 *__restrict__ tw, int tid) {
    constexpr int n16 = LOGN / 4, rem = LOGN % 4;
    constexpr int npass = n16 + (rem ? 1 : 0);
    static_assert(n16 >= 1 && n16 <= 2, "tile transform: 16 <= n <= 2048");
    // the stride Ns of each pass is a compile-time constant: LDS addresses become one base
    // register plus immediate offsets
    tile_r16<LOGN, W, NT, INV, true, npass == 1, 1>(x, lds, tw, tid);
    if constexpr (n16 == 2) tile_r16<LOGN, W, NT, INV, false, npass == 2, 16>(x, lds, tw, tid);
    if constexpr (rem == 3) tile_r8_last<LOGN, W, NT, INV>(x, lds, tw, tid);
    if constexpr (rem == 2)
        tile_r4<LOGN, W, NT, INV, true, (n16 == 2 ? 256 : 16)>(x, lds, tw, tid);
    if constexpr (rem == 1) tile_r2_last<LOGN, W, NT, INV>(x, lds, tw, tid);
}

// Persistent form of the strided pass: one workgroup per CU walks tiles t, t + G, ... and
// loads tile t + G into registers before transforming tile t, so the CU's memory pipe
// stays busy during the transform (with one 128 KB tile per CU nothing else would overlap
// them).  Needs the 256-VGPR budget of <= 512 lanes (radix-16 schedule) and no
// vmcnt-tracked access between the prefetch and its use — the twiddle and k-space tables
// are therefore copied to LDS once per workgroup (vector memory returns in order: a global
// twiddle load would wait for the prefetch).
template <int LOGN, int NT, int MODE, int W>
__global__ __launch_bounds__(NT) void k_fft_strided_p(const double2 *__restrict__ src,
                                                      double2 *__restrict__ dst, i64 s_ostride,
                                                      i64 s_es, i64 d_ostride, i64 d_es, int nkb,
                                                      i64 ntiles, i64 o_off,
                                                      const double2 *__restrict__ tw,
                                                      KspaceParams P) {
    // plain pencils only (element m at base + m*es): point f = tid + r*NT of the tile is
    // pencil tid % W, element tid / W + r*(NT/W), i.e. a per-lane 32-bit offset on top of a
    // wave-uniform base per r — no per-element 64-bit address registers
    constexpr int N = 1 << LOGN;
    constexpr int TOT = N * W;
    static_assert(TOT == 16 * NT && NT % W == 0, "persistent pass: 16 points per lane");
    constexpr int PER = 16;
    constexpr int MSTEP = NT / W;  // = N/16
    extern __shared__ double2 lds_dyn[];
    double2 *lds = lds_dyn;
    double2 *twl = lds_dyn + TOT;                      // N/2 twiddles (no pass indexes beyond)
    double *tabq = (double *)(lds_dyn + TOT + N / 2);  // MODE 2: N doubles
    const int tid = threadIdx.x;
    const int nk = N / 2 + 1;
    for (int i = tid; i < N; i += NT) {
        if (i < N / 2) twl[i] = tw[i];
        if (MODE == 2) tabq[i] = kspace_tab_sep(P, N, i, P.tab_q[i]);  // t[a], cg_kspace.h
    }
    const int wl = tid % W, ml = tid / W;
    const unsigned voff_s = (unsigned)(ml * s_es + wl), voff_d = (unsigned)(ml * d_es + wl);
    // staged tiles are held as native vectors (plain loads/stores the optimiser keeps in
    // registers; arrays of the double2 class are copied with memcpy and end up in scratch)
    d2 v[PER], u[PER];
    const bool reverse = (MODE != 2) && (P.long_range & 2);  // walk the tiles from the end
    auto load = [&](i64 t) {
        if (reverse) t = ntiles - 1 - t;
        const i64 o = t / nkb;
        const int kk0 = (int)(t - o * nkb) * W;
        const double2 *sbase = src + o * s_ostride + kk0;
        // unconditional: pencils kk >= N/2+1 of the last tile lie in the row padding
        // (nkb*W <= pad/2), are transformed as garbage and written back there — straight-line
        // loads and stores let the compiler count them (vmcnt)
#pragma unroll
        for (int r = 0; r < PER; r++)
            v[r] = ((const d2 *)(sbase + (i64)r * MSTEP * s_es))[voff_s];
    };
    auto store = [&](i64 t) {
        if (reverse) t = ntiles - 1 - t;
        const i64 o = t / nkb;
        const int kk0 = (int)(t - o * nkb) * W;
        double2 *dbase = dst + o * d_ostride + kk0;
#pragma unroll
        for (int r = 0; r < PER; r++) ((d2 *)(dbase + (i64)r * MSTEP * d_es))[voff_d] = u[r];
    };
    // Software pipeline per tile:  x <- v (tile t) | store u (results of the previous tile)
    // | issue loads of the next tile into v | transform x | u <- x.
    // The stores are issued BEFORE the prefetch loads so that, at the next tile's wait for
    // v, everything older in the (in-order) vector memory queue is stores of the tile before —
    // the wait counts only the loads and never drains fresh stores.
    i64 tprev = -1;
    i64 t = blockIdx.x;
    if (t >= ntiles) return;
    load(t);
    __syncthreads();  // tables
    for (; t < ntiles; t += gridDim.x) {
        double2 x[16];
#pragma unroll
        for (int r = 0; r < PER; r++) x[r] = make_double2(v[r].x, v[r].y);
        if (tprev >= 0) store(tprev);
        // unconditional (the last tile of a workgroup re-reads its own input): every path
        // into the next tile carries the same loads in flight
        load(t + gridDim.x < ntiles ? t + gridDim.x : t);
        const i64 o = t / nkb;
        const int kk0 = (int)(t - o * nkb) * W;
        if (MODE == 0 || MODE == 2) fft_tile<LOGN, W, NT, false>(x, lds, twl, tid);
        if (MODE == 2) {
            const int kkl = kk0 + wl;
            if (kkl < nk) {
                const int b = (int)(o + o_off);
                const KspaceFixedS F = kspace_fix_sep(P, N, b, kkl, tabq[b], tabq[kkl]);
                // the per-element table values do not depend on the tile: keep the optimiser
                // from hoisting 16 x (q, ka^2, ...) out of the tile loop into registers it
                // does not have (they would be spilled to scratch); re-reading LDS is cheap
                int ml_t = ml;
                asm volatile("" : "+v"(ml_t));
#pragma unroll
                for (int r = 0; r < PER; r++) {
                    int a = ml_t + r * MSTEP;
                    double fac = kspace_factor_sep(F, N, a, tabq[a]);
                    x[r] = make_double2(x[r].x * fac, x[r].y * fac);
                    // keep the 16 factor evaluations from being interleaved: their
                    // temporaries on top of x and the staged v would spill
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (MODE == 1 || MODE == 2) fft_tile<LOGN, W, NT, true>(x, lds, twl, tid);
#pragma unroll
        for (int r = 0; r < PER; r++) {
            u[r].x = x[r].x;
            u[r].y = x[r].y;
        }
        tprev = t;
    }
    store(tprev);
}

// ---------------------------------------------------------------------------
// 2048-point strided passes at full line width.  Eight pencils of 2048 points (256 KB) do not
// fit the LDS, and four (the persistent pass above with W = 4) leave every row segment at 64
// bytes — half a line: `tools/stride_probe2048.cpp` measures 53 ms per in-place sweep of the
// 2048^3 mesh in 64-byte segments against 28.5 ms in 128-byte ones, whatever the stride.  So
// the transform is split once by radix 2 over even and odd rows:
//     X[k]     = E[k] + w^k O[k]         E = FFT_1024(x[2m]),  O = FFT_1024(x[2m + 1])
//     X[k + H] = E[k] - w^k O[k]
// A tile is 8 pencils wide (full lines); its even rows are transformed as one 1024-point tile
// (the 16*16*4 schedule above, 128 KB of LDS) and stay in registers while the odd rows —
// prefetched meanwhile — go through the same LDS; lane (ml, wl) holds E and O at the same 16
// positions k = ml + 64 r, so the last stage is register arithmetic with w^k = w^ml * w32^r.
// The fused x pass continues with the k-space factor on all 32 values, the mirrored split
// (decimation in frequency: A = X[k] + X[k + H], B = (X[k] - X[k + H]) conj(w)^k) and two
// inverse 1024-point tiles whose results are the even and odd rows.  Per tile: two loads of
// 16 values per lane (each prefetched during a transform), one store of 32, deferred to the
// next tile like in the pass above.  Registers: 3 x 16 values + temporaries < 256.
// ---------------------------------------------------------------------------
template <int LOGN, int NT, int MODE, bool BLOCKED>
__global__ __launch_bounds__(NT) void k_fft_strided_h(const double2 *__restrict__ src,
                                                      double2 *__restrict__ dst, PencilMap smap,
                                                      PencilMap dmap, int nkb, i64 ntiles,
                                                      i64 o_off,
                                                      const double2 *__restrict__ tw,
                                                      KspaceParams P) {
    constexpr int N = 1 << LOGN, H = N / 2, W = 8, LOGH = LOGN - 1;
    constexpr int TOT = H * W;
    static_assert(TOT == 16 * NT && NT % W == 0, "split pass: 16 points per lane and half");
    constexpr int PER = 16;
    constexpr int MSTEP = NT / W;  // = H/16
    constexpr bool INV1 = MODE == 1;  // direction of the first (or only) transform
    // Pencil maps: plain (point m at m*es) or blocked by destination domain (the all-to-all
    // buffers of the x-slab decomposition, fft_dist): a lane's points are m = l + c with the
    // lane part l < 2*MSTEP and the wave-uniform c a multiple of 2*MSTEP (MSTEP for the
    // natural-order stores); blocks hold a multiple of 2*MSTEP points (checked by the
    // launcher), so block number and offset inside the block of c are scalar and the lane
    // part never leaves the block.
    const i64 s_es = smap.es, d_es = dmap.es;
    // (blocked maps: the 64 block offsets of a tile are scalar arithmetic redone at every use —
    // hoisted out of the tile loop they would be 128 scalar registers, spilled)
    auto opaque = [](int c) {
        asm volatile("" : "+s"(c));
        return c;
    };
    extern __shared__ double2 lds_dyn[];
    double2 *lds = lds_dyn;
    double2 *twh = lds_dyn + TOT;                      // H/2 twiddles of the H-point tiles
    double *tabq = (double *)(lds_dyn + TOT + H / 2);  // MODE 2: N doubles, t[a] of cg_kspace.h
    const int tid = threadIdx.x;
    const int nk = N / 2 + 1;
    for (int i = tid; i < N; i += NT) {
        if (i < H / 2) twh[i] = tw[2 * i];  // w_H^i = w_N^(2i)
        if (MODE == 2) tabq[i] = kspace_tab_sep(P, N, i, P.tab_q[i]);  // t[a], cg_kspace.h
    }
    // w_N^ml (ml < NT/W): w_N^(ml + 64 r) = w_N^ml * w32^r
    double2 *wm = lds_dyn + TOT + H / 2 + (MODE == 2 ? N / 2 : 0);
    if (tid < NT / W) wm[tid] = tw[tid];
    // Everything a lane derives from its number (pencil wl, first point ml, the offsets of its
    // rows, the LDS addresses inside the tile transforms) is recomputed where it is used, from
    // a value the optimiser cannot see through: kept across the tile they would be spilled,
    // and a scratch reload behind a prefetch waits for the prefetch (vector memory returns in
    // order), which would serialise memory and arithmetic.
    int tid_t = tid;
    auto fresh_tid = [&]() {
        asm volatile("" : "+v"(tid_t));
        return tid_t;
    };
    d2 v[PER], u[2 * PER];
    const bool reverse = (MODE != 2) && (P.long_range & 2);  // walk the tiles from the end
    auto load = [&](i64 t, int odd) {
        if (reverse) t = ntiles - 1 - t;
        const i64 o = t / nkb;
        const int kk0 = (int)(t - o * nkb) * W;
        const double2 *sbase = src + o * smap.ostride + kk0;
        // rows of the halves: element m = ml + 64 r of the even half is row 2m, of the odd 2m + 1
        const int ft = fresh_tid();
        const unsigned voff_h = (unsigned)(2 * (ft / W) * s_es + ft % W);
#pragma unroll
        for (int r = 0; r < PER; r++)
            v[r] = ((const d2 *)(sbase + (BLOCKED ? pencil_off(smap, opaque(r * (2 * MSTEP)) + odd)
                                                  : ((i64)r * (2 * MSTEP) + odd) * s_es)))[voff_h];
    };
    // the 32 results of a tile leave in two halves (hi = 0: u[0..15], hi = 1: u[16..31]), one
    // at the start of each phase of the next tile, so that stores and loads alternate
    auto store = [&](i64 t, int hi) {
        if (reverse) t = ntiles - 1 - t;
        const i64 o = t / nkb;
        const int kk0 = (int)(t - o * nkb) * W;
        // MODE 2: u[r] is row 2 (ml + 64 r), u[16 + r] the row after it;
        // otherwise u[r] is row ml + 64 r, u[16 + r] row ml + 64 r + H
        double2 *dbase = dst + o * dmap.ostride + kk0;
        constexpr int RS = (MODE == 2 ? 2 : 1) * MSTEP;
        const int c0 = hi ? (MODE == 2 ? 1 : H) : 0;
        const int ft = fresh_tid();
        const unsigned voff_d = (unsigned)((MODE == 2 ? 2 : 1) * (ft / W) * d_es + ft % W);
#pragma unroll
        for (int r = 0; r < PER; r++)
            ((d2 *)(dbase + (BLOCKED ? pencil_off(dmap, opaque(r * RS) + c0)
                                     : ((i64)r * RS + c0) * d_es)))[voff_d] = u[(hi ? PER : 0) + r];
    };
    // w32^r = exp(-2 pi i r / 32), r = 0 .. 15
    constexpr double c32[16] = {1.0,
                                0.98078528040323044913,
                                0.92387953251128675613,
                                0.83146961230254523708,
                                0.70710678118654752440,
                                0.55557023301960222474,
                                0.38268343236508977173,
                                0.19509032201612826785,
                                0.0,
                                -0.19509032201612826785,
                                -0.38268343236508977173,
                                -0.55557023301960222474,
                                -0.70710678118654752440,
                                -0.83146961230254523708,
                                -0.92387953251128675613,
                                -0.98078528040323044913};
    constexpr double s32[16] = {0.0,
                                -0.19509032201612826785,
                                -0.38268343236508977173,
                                -0.55557023301960222474,
                                -0.70710678118654752440,
                                -0.83146961230254523708,
                                -0.92387953251128675613,
                                -0.98078528040323044913,
                                -1.0,
                                -0.98078528040323044913,
                                -0.92387953251128675613,
                                -0.83146961230254523708,
                                -0.70710678118654752440,
                                -0.55557023301960222474,
                                -0.38268343236508977173,
                                -0.19509032201612826785};
    i64 tprev = -1;
    i64 t = blockIdx.x;
    if (t >= ntiles) return;
    load(t, 0);
    __syncthreads();  // tables
    for (; t < ntiles; t += gridDim.x) {
        double2 x[16], e[16];
        // ---- even rows
#pragma unroll
        for (int r = 0; r < PER; r++) e[r] = make_double2(v[r].x, v[r].y);
        if (tprev >= 0) store(tprev, 0);
        load(t, 1);
        fft_tile<LOGH, W, NT, INV1>(e, lds, twh, fresh_tid());
        // ---- odd rows (the next tile's even rows are fetched meanwhile; in the fused pass
        // only during the inverse transforms)
#pragma unroll
        for (int r = 0; r < PER; r++) x[r] = make_double2(v[r].x, v[r].y);
        if (tprev >= 0) store(tprev, 1);
        const i64 tnext = t + gridDim.x < ntiles ? t + gridDim.x : t;
        if (MODE != 2) load(tnext, 0);
        fft_tile<LOGH, W, NT, INV1>(x, lds, twh, fresh_tid());
        const i64 o = t / nkb;
        const int kk0 = (int)(t - o * nkb) * W;
        const int ftc = fresh_tid();
        const int wl = ftc % W, ml = ftc / W;
        const int kkl = kk0 + wl;
        const double2 wml = wm[ml];
        KspaceFixedS F = {};
        if (MODE == 2) {
            const int b = (int)(o + o_off);
            const int kks = kkl < nk ? kkl : 0;  // (pencils in the row padding: any finite value)
            F = kspace_fix_sep(P, N, b, kks, tabq[b], tabq[kks]);
        }
        const int ml_t = ml;
        const double wmx = wml.x, wmy = wml.y;
#pragma unroll
        for (int r = 0; r < PER; r++) {
            // w^k, k = ml + 64 r (conjugated for the inverse transform)
            double2 wk = cmul(make_double2(wmx, wmy), make_double2(c32[r], s32[r]));
            if (INV1) wk = cconj(wk);
            const double2 y = cmul(x[r], wk);
            double2 x0 = cadd(e[r], y), x1 = csub(e[r], y);
            if (MODE == 2) {
                const int a = ml_t + r * MSTEP;
                const double f0 = kspace_factor_sep(F, N, a, tabq[a]);
                const double f1 = kspace_factor_sep(F, N, a + H, tabq[a + H]);
                x0 = make_double2(x0.x * f0, x0.y * f0);
                x1 = make_double2(x1.x * f1, x1.y * f1);
                // decimation in frequency for the way back: even rows from the sums, odd rows
                // from the differences times conj(w)^k
                e[r] = cadd(x0, x1);
                x[r] = cmul(csub(x0, x1), cconj(wk));
                // (pinned here: the optimiser would sink the odd half's products below the
                // first inverse transform and keep x0, x1 alive — in scratch — instead)
                asm volatile("" : "+v"(x[r].x), "+v"(x[r].y), "+v"(e[r].x), "+v"(e[r].y));
                __builtin_amdgcn_sched_barrier(0);
            } else {
                e[r] = x0;
                x[r] = x1;
            }
        }
        if (MODE == 2) {
            load(tnext, 0);  // (not earlier: three arrays of 16 values are the register budget)
            fft_tile<LOGH, W, NT, true>(e, lds, twh, fresh_tid());
            fft_tile<LOGH, W, NT, true>(x, lds, twh, fresh_tid());
        }
#pragma unroll
        for (int r = 0; r < PER; r++) {
            u[r].x = e[r].x;
            u[r].y = e[r].y;
            u[PER + r].x = x[r].x;
            u[PER + r].y = x[r].y;
        }
        tprev = t;
    }
    store(tprev, 0);
    store(tprev, 1);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <int LOGN>
static int run_z(cg_ctx *c, bool inverse, i64 layer0 = 0, i64 nlayers = -1) {
    constexpr int N = 1 << LOGN;
    constexpr int NT = (N / 8) < 64 ? 64 : ((N / 8) > 256 ? 256 : (N / 8));
    if (nlayers < 0) nlayers = c->xmap.nxl;  // owned layers only
    unsigned rows = (unsigned)(nlayers * c->N);
    double *base = c->mesh0 + layer0 * c->ny * c->pad;
    if (!inverse)
        hipLaunchKernelGGL((k_fft_z_forward<LOGN, NT>), dim3(rows), dim3(NT), 0, c->stream,
                           base, c->ny, c->pad, (const double2 *)c->fft_tw);
    else
        hipLaunchKernelGGL((k_fft_z_backward<LOGN, NT>), dim3(rows), dim3(NT), 0, c->stream,
                           base, c->ny, c->pad, (const double2 *)c->fft_tw);
    CG_LAUNCH_CHECK();
    return 0;
}

template <int LOGN, int MODE, int W, bool R16>
static int run_strided_r(cg_ctx *c, const double2 *src, double2 *dst, PencilMap smap,
                         PencilMap dmap, i64 nouter, i64 o_off, const KspaceParams &P) {
    constexpr int N = 1 << LOGN;
    // lanes: radix 4: N*W/8 (two butterflies per lane and pass); radix 16: N*W/16 (one);
    // 64..1024
    constexpr int NTW = N * W / (R16 ? 16 : 8);
    constexpr int NT = NTW < 64 ? 64 : (NTW > 1024 ? 1024 : NTW);
    const int nkb = (int)((c->N / 2 + 1 + W - 1) / W);
    size_t lds = sizeof(double2) * N * W;
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        CG_HIP(hipGetDeviceProperties(&prop, c->p.device));
        ncu = prop.multiProcessorCount;
    }
    const i64 ntiles = nouter * nkb;
    // persistent form: tables in LDS on top of the tile; only where that fits the 160 KB
    // and pays (many tiles per CU)
    constexpr size_t lds_p = sizeof(double2) * N * W + sizeof(double2) * (N / 2) +
                             (MODE == 2 ? sizeof(double) * N : 0);
    // per-lane offsets are 32-bit byte offsets: (NT/W) pencil points must span < 4 GB
    // (2048 points x 4 pencils + tables = exactly 160 KB in the fused pass)
    constexpr bool can_persist = R16 && LOGN <= 11 && N * W == 16 * NT && NT % W == 0 &&
                                 lds_p <= 160 * 1024;
    if constexpr (can_persist) {
        if (ntiles >= 4 * (i64)ncu && smap.sh == 31 && dmap.sh == 31) {
            auto kern = k_fft_strided_p<LOGN, NT, MODE, W>;
            static bool attr_set_p[64] = {};  // per device: one process may drive several
            const int dev_p = c->p.device & 63;
            if (!attr_set_p[dev_p] && lds_p > 64 * 1024) {
                CG_HIP(hipFuncSetAttribute((const void *)kern,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
                attr_set_p[dev_p] = true;
            }
            // workgroups per CU the LDS footprint allows
            int per_cu = (int)((160 * 1024) / lds_p);
            if (per_cu < 1) per_cu = 1;
            if (per_cu > 4) per_cu = 4;
            hipLaunchKernelGGL(kern, dim3((unsigned)(ncu * per_cu)), dim3(NT), lds_p, c->stream,
                               src, dst, smap.ostride, smap.es, dmap.ostride, dmap.es, nkb,
                               ntiles, o_off, (const double2 *)c->fft_tw, P);
            CG_LAUNCH_CHECK();
            return 0;
        }
    }
    auto kern = k_fft_strided<LOGN, NT, MODE, W, R16>;
    static bool attr_set[64] = {};
    const int dev = c->p.device & 63;
    if (!attr_set[dev] && lds > 64 * 1024) {
        CG_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(nouter * nkb)), dim3(NT), lds, c->stream, src, dst,
                       smap, dmap, nkb, o_off, (const double2 *)c->fft_tw, P);
    CG_LAUNCH_CHECK();
    return 0;
}

// The split pass (k_fft_strided_h): plain pencil maps, enough tiles to keep every CU busy.
template <int LOGN, int MODE>
static int run_strided_h(cg_ctx *c, const double2 *src, double2 *dst, PencilMap smap,
                         PencilMap dmap, i64 nouter, i64 o_off, const KspaceParams &P, bool *done) {
    constexpr int N = 1 << LOGN, W = 8, NT = (N / 2) * W / 16;
    *done = false;
    static int enabled = -1, ncu = 0;
    if (enabled < 0) {
        // CONCEPT_GPU_FFT_SPLIT=0: the whole-pencil passes everywhere (what small meshes and the
        // blocked maps of narrow slabs take anyway: the tests run them at size this way);
        // 2: the split pass at 512 points too (it does not pay there: 0.48 against 0.46 ms)
        const char *env = getenv("CONCEPT_GPU_FFT_SPLIT");
        enabled = env ? atoi(env) : 1;
        if (LOGN < 10 && enabled < 2) enabled = 0;
        hipDeviceProp_t prop;
        CG_HIP(hipGetDeviceProperties(&prop, c->p.device));
        ncu = prop.multiProcessorCount;
    }
    const int nkb = (int)((c->N / 2 + 1 + W - 1) / W);
    const i64 ntiles = nouter * nkb;
    // per-lane offsets are 32-bit byte offsets: rows 2 ml (ml < NT/W) must span < 4 GB
    const i64 reach = ((2 * (i64)(NT / W - 1)) * (smap.es > dmap.es ? smap.es : dmap.es) + W) * 16;
    // blocked maps (fft_dist): whole multiples of the 2*(NT/W) rows a lane spans per block
    auto blocks_ok = [](const PencilMap &m) { return m.sh == 31 || (1 << m.sh) >= 2 * (NT / W); };
    if (!enabled || ntiles < 2 * (i64)ncu || !blocks_ok(smap) || !blocks_ok(dmap) ||
        reach >= ((i64)1 << 32) || (i64)nkb * W > c->pad / 2)
        return 0;
    constexpr size_t lds = sizeof(double2) * (N / 2) * W + sizeof(double2) * (N / 4) +
                           (MODE == 2 ? sizeof(double) * N : 0) + sizeof(double2) * (NT / W);
    static_assert(lds <= 160 * 1024, "split pass: LDS");
    const bool blocked = smap.sh != 31 || dmap.sh != 31;
    auto kern = blocked ? k_fft_strided_h<LOGN, NT, MODE, true> : k_fft_strided_h<LOGN, NT, MODE, false>;
    static bool attr_set[64][2] = {};
    const int dev = c->p.device & 63;
    if (!attr_set[dev][blocked]) {
        CG_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        attr_set[dev][blocked] = true;
    }
    // workgroups per CU the LDS footprint allows (2048: one; 1024: two, each the other's cover)
    const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(ncu * per_cu)), dim3(NT), lds, c->stream, src, dst,
                       smap, dmap, nkb, ntiles, o_off, (const double2 *)c->fft_tw, P);
    CG_LAUNCH_CHECK();
    *done = true;
    return 0;
}

// (radix-16 schedule of the in-LDS transform: the radix-4 one of round 1 lost)
template <int LOGN, int MODE, int W>
static int run_strided_w(cg_ctx *c, const double2 *src, double2 *dst, PencilMap smap,
                         PencilMap dmap, i64 nouter, i64 o_off, const KspaceParams &P) {
    return run_strided_r<LOGN, MODE, W, true>(c, src, dst, smap, dmap, nouter, o_off, P);
}

// W adjacent kk per workgroup.  Measured at 1024^3 (ms per y pass / fused x pass):
//   W = 2 (32 B row segments, 4 workgroups per CU)  9.4 / 9.6
//   W = 4 (64 B, 2 per CU)                          4.6 / 7.9
//   W = 8 (128 B = one full cache line, 1 per CU)   3.95 / 7.4
// The row-segment width, not the occupancy, decides: default 8 (N <= 1024; 8 pencils of
// 2048 points exceed the 160 KB LDS, so 4 there).
template <int LOGN, int MODE>
static int run_strided(cg_ctx *c, const double2 *src, double2 *dst, PencilMap smap,
                       PencilMap dmap, i64 nouter, i64 o_off, const KspaceParams &P) {
    if constexpr (LOGN >= 9 && LOGN <= 11) {  // the even/odd split pass (k_fft_strided_h)
        bool done = false;
        if (int rc = run_strided_h<LOGN, MODE>(c, src, dst, smap, dmap, nouter, o_off, P, &done))
            return rc;
        if (done) return 0;
    }
    if (LOGN <= 10)  // 8 pencils of 2048 points would not fit the 160 KB LDS
        return run_strided_w<(LOGN <= 10 ? LOGN : 10), MODE, 8>(c, src, dst, smap, dmap, nouter,
                                                                o_off, P);
    return run_strided_w<LOGN, MODE, 4>(c, src, dst, smap, dmap, nouter, o_off, P);
}

static PencilMap plain_map(i64 ostride, i64 es) { return PencilMap{ostride, es, 0, 31}; }

// Layers per chunk of the interleaved z / y passes.  Measured at 1024^3 (8.5 MB per layer, ms
// for z + y forward and inverse together): 16 layers 13.4, 24: 12.2, 27: 11.6, 28: 11.7,
// 31: 11.3, 32: 11.9, 36: 13.2, 40: 13.6, no chunks: 14.4 — the chunk must fit the 256 MB
// infinity cache (the cliff sits just above 32 layers = 273 MB), and among the sizes that do,
// those whose tile count nl*ceil((N/2+1)/8) fills whole rounds of the persistent y pass (one
// workgroup per CU) win: 31 layers = 2015 tiles = 7.9 rounds of 256 beats 32 = 8.1 rounds.
static i64 zy_chunk_layers(cg_ctx *c) {
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        ncu = hipGetDeviceProperties(&prop, c->p.device) == hipSuccess ? prop.multiProcessorCount
                                                                        : 256;
    }
    const i64 plane_bytes = c->ny * c->pad * 8;
    i64 cap = 266000000ll / plane_bytes;
    if (cap < 1) cap = 1;
    if (cap >= c->N) return c->N + 1;  // the whole mesh fits: no chunks
    const i64 nkb = (c->N / 2 + 1 + 7) / 8;
    i64 best = cap;
    double best_eff = 0;
    for (i64 nl = cap; nl >= cap - cap / 4 && nl >= 1; nl--) {
        i64 tiles = nl * nkb, rounds = (tiles + ncu - 1) / ncu;
        double eff = (double)tiles / (double)(rounds * ncu);
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best = nl;
        }
    }
    return best;
}

// Split `total` layers into chunks of about `chunk` layers whose sizes differ by at most one
// (no tiny last chunk): chunk i covers [chunk_begin(i), chunk_begin(i + 1)).
struct ChunkPlan {
    i64 n, base, extra;
    ChunkPlan(i64 total, i64 chunk) {
        n = chunk >= total ? 1 : (total + chunk / 2) / chunk;
        if (n < 1) n = 1;
        base = total / n;
        extra = total % n;
    }
    i64 begin(i64 i) const { return i * base + (i < extra ? i : extra); }
};

template <int LOGN>
static int fft3d(cg_ctx *c, int what, const KspaceParams &P) {
    // single domain.  what: 0 forward, 1 backward, 2 forward + kernel + backward (fused x)
    const i64 cp = c->pad / 2, N = c->N;
    double2 *m = (double2 *)c->mesh0;
    // y pencils: one per (layer, kk) with stride cp; x pencils: one per (row, kk) with the
    // layer stride cp*ny (ny rows per layer, cg_ctx::ny)
    PencilMap ymap = plain_map(cp * c->ny, cp), xmap = plain_map(cp, cp * c->ny);
    if (what == 0 || what == 1) {
        const i64 chunk0 = zy_chunk_layers(c), ls0 = cp * c->ny;
        if (what == 1 && run_strided<LOGN, 1>(c, m, m, xmap, xmap, N, 0, P)) return 1;
        const ChunkPlan plan0(N, chunk0);  // one chunk = the whole mesh when it fits
        for (i64 ci = 0; ci < plan0.n; ci++) {
            const i64 l0 = plan0.begin(ci), nl = plan0.begin(ci + 1) - l0;
            if (what == 0) {
                if (run_z<LOGN>(c, false, l0, nl)) return 1;
                if (run_strided<LOGN, 0>(c, m + l0 * ls0, m + l0 * ls0, ymap, ymap, nl, 0, P))
                    return 1;
            } else {
                if (run_strided<LOGN, 1>(c, m + l0 * ls0, m + l0 * ls0, ymap, ymap, nl, 0, P))
                    return 1;
                if (run_z<LOGN>(c, true, l0, nl)) return 1;
            }
        }
        if (what == 0) return run_strided<LOGN, 0>(c, m, m, xmap, xmap, N, 0, P);
        return 0;
    }
    auto mark = [&](int i) {
        if (c->pass_events) (void)hipEventRecord(c->pass_events[i], c->stream);
    };
    const i64 chunk = zy_chunk_layers(c);
    if (chunk < N) {
        // z and y passes interleaved over chunks of layers that fit the 256 MB infinity
        // cache: the second pass of a chunk reads what the first just wrote from the cache,
        // and its own stores overwrite those (still cached, dirty) lines — per chunk HBM sees
        // one read and one write instead of two of each.  1024^3: z + y 7.3 -> 5.9 ms forward,
        // 7.1 -> 5.8 ms inverse with 28 layers (238 MB) per chunk; 36 layers and the gain is
        // gone.  The two entries of pass_ms that the chunks merge are reported as one.
        const i64 ls = cp * c->ny;  // complex elements per layer
        KspaceParams Pr = P;
        Pr.long_range |= 2;  // y forward walks a chunk from its end, where z just stopped
        const ChunkPlan plan(N, chunk);
        mark(0);
        for (i64 ci = 0; ci < plan.n; ci++) {
            const i64 l0 = plan.begin(ci), nl = plan.begin(ci + 1) - l0;
            if (run_z<LOGN>(c, false, l0, nl)) return 1;
            if (run_strided<LOGN, 0>(c, m + l0 * ls, m + l0 * ls, ymap, ymap, nl, 0, Pr)) return 1;
        }
        mark(1);
        mark(2);
        if (run_strided<LOGN, 2>(c, m, m, xmap, xmap, N, 0, P)) return 1;
        mark(3);
        for (i64 ci = 0; ci < plan.n; ci++) {
            const i64 l0 = plan.begin(ci), nl = plan.begin(ci + 1) - l0;
            if (run_strided<LOGN, 1>(c, m + l0 * ls, m + l0 * ls, ymap, ymap, nl, 0, P)) return 1;
            if (run_z<LOGN>(c, true, l0, nl)) return 1;
        }
        mark(4);
        mark(5);
        return 0;
    }
    mark(0);
    if (run_z<LOGN>(c, false)) return 1;
    mark(1);
    {
        // the y pass walks the layers from the end, where the z pass just stopped: the tail
        // of the mesh is still in the 256 MB infinity cache (3.73 -> 3.62 ms); the inverse z
        // pass does the same after the inverse y pass
        KspaceParams Pr = P;
        Pr.long_range |= 2;
        if (run_strided<LOGN, 0>(c, m, m, ymap, ymap, N, 0, Pr)) return 1;
    }
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
static int fft_dist(cg_ctx *c, int what, double2 *buf, const KspaceParams &P, i64 layer0,
                    i64 nlayers) {
    const i64 cp = c->pad / 2, N = c->N, nxl = c->xmap.nxl, JB = N / c->p.nprocs;
    if (nlayers < 0) nlayers = nxl - layer0;  // [layer0, layer0 + nlayers) of the owned layers
    // rows per x layer in the transpose buffers: JB used + 1 unused, for the same reason as
    // cg_ctx::ny (the x pencils' stride JB*cp*16 B would be a large power of two times 65)
    const i64 JBp = JB + 1;
    int sh = 0;
    while ((1 << sh) < JB) sh++;
    double2 *m = (double2 *)c->mesh0;
    PencilMap ymap = plain_map(cp * c->ny, cp);
    PencilMap bmap{JBp * cp, cp, nxl * JBp * cp, sh};
    const i64 chunk = zy_chunk_layers(c), ls = cp * c->ny, lb = JBp * cp;
    if (what == 0) {  // forward z, forward y -> send buffer, chunk by chunk (see fft3d)
        const ChunkPlan plan(nlayers, chunk);
        for (i64 ci = 0; ci < plan.n; ci++) {
            const i64 l0 = layer0 + plan.begin(ci), nl = plan.begin(ci + 1) - plan.begin(ci);
            if (run_z<LOGN>(c, false, l0, nl)) return 1;
            if (run_strided<LOGN, 0>(c, m + l0 * ls, buf + l0 * lb, ymap, bmap, nl, 0, P)) return 1;
        }
        return 0;
    }
    if (what == 2) {  // fused x pass on complex[N][JB][cp]
        PencilMap xmap = plain_map(cp, JBp * cp);
        return run_strided<LOGN, 2>(c, buf, buf, xmap, xmap, JB, (i64)c->p.rank * JB, P);
    }
    if (what == 3 || what == 4) {  // the x pass alone (general particle_mesh: the Fourier slab
        PencilMap xmap = plain_map(cp, JBp * cp);  // is operated on between the two)
        if (what == 3)
            return run_strided<LOGN, 0>(c, buf, buf, xmap, xmap, JB, (i64)c->p.rank * JB, P);
        return run_strided<LOGN, 1>(c, buf, buf, xmap, xmap, JB, (i64)c->p.rank * JB, P);
    }
    // backward y from the returned buffer, backward z
    const ChunkPlan plan(nlayers, chunk);
    for (i64 ci = 0; ci < plan.n; ci++) {
        const i64 l0 = layer0 + plan.begin(ci), nl = plan.begin(ci + 1) - plan.begin(ci);
        if (run_strided<LOGN, 1>(c, buf + l0 * lb, m + l0 * ls, bmap, ymap, nl, 0, P)) return 1;
        if (run_z<LOGN>(c, true, l0, nl)) return 1;
    }
    return 0;
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
    KspaceParams P{c->ktab_n, c->ktab_s, c->ktab_q, deconv_order, long_range, C, E};
    CG_FFT_DISPATCH(fft3d, c, what, P)
}
int cgk_fft_dist_forward(cg_ctx *c, double *send_buf, i64 layer0, i64 nlayers) {
    KspaceParams P{c->ktab_n, c->ktab_s, c->ktab_q, 0, 0, 0.0, 0.0};
    CG_FFT_DISPATCH(fft_dist, c, 0, (double2 *)send_buf, P, layer0, nlayers)
}
int cgk_fft_dist_xsolve(cg_ctx *c, double *buf, int deconv_order, double C, int long_range,
                        double E) {
    KspaceParams P{c->ktab_n, c->ktab_s, c->ktab_q, deconv_order, long_range, C, E};
    CG_FFT_DISPATCH(fft_dist, c, 2, (double2 *)buf, P, 0, -1)
}
int cgk_fft_dist_x(cg_ctx *c, double *buf, int inverse) {
    KspaceParams P{c->ktab_n, c->ktab_s, c->ktab_q, 0, 0, 0.0, 0.0};
    CG_FFT_DISPATCH(fft_dist, c, inverse ? 4 : 3, (double2 *)buf, P, 0, -1)
}
int cgk_fft_dist_backward(cg_ctx *c, const double *recv_buf, i64 layer0, i64 nlayers) {
    KspaceParams P{c->ktab_n, c->ktab_s, c->ktab_q, 0, 0, 0.0, 0.0};
    CG_FFT_DISPATCH(fft_dist, c, 1, (double2 *)recv_buf, P, layer0, nlayers)
}
