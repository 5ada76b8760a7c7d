// cg_kspace.h — the per-mode factor of the Poisson / deconvolution kernel,
// shared by the stand-alone k-space kernel and the fused FFT pass.
//   fourier_loop            mesh.py:2615-2890   (deconvolution factor)
//   particle_mesh           interactions.py:2092-2118 (Poisson factor, cut-off)
//   nullify_modes           mesh.py:3585-3622   (Nyquist planes, origin)
#pragma once
#include "cg_internal.h"

struct KspaceParams {
    const double *tab_n;  // n(k) = k*R[pi/gridsize] + machine_eps by array index
    const double *tab_s;  // sin(n(k))
    const double *tab_q;  // n(k)/sin(n(k)) (may be null where only kspace_factor is used)
    int deconv_order;
    int long_range;
    double C;  // -boxsize**2*G_Newton/pi
    double E;  // -(2*pi/boxsize*scale)**2
};

// Factor multiplying re and im of mode (array indices a, b in the two full
// dimensions — the expression is symmetric in them — and kk in the half
// dimension).  Returns 0 for the nullified modes.
__device__ __forceinline__ double kspace_factor(const KspaceParams &P, i64 N, i64 a, i64 b,
                                                i64 kk) {
#pragma clang fp contract(off)  // the reference's operation order, also inside cg_fft.hip
    const i64 nyq = N / 2;
    if (a == nyq || b == nyq || kk == nyq) return 0;  // nullify_modes('nyquist')
    i64 ka = a - (a >= nyq ? N : 0), kb = b - (b >= nyq ? N : 0);
    i64 kab2 = kb * kb + ka * ka;  // interactions.py:2096 (integer, order-free)
    if (kab2 == 0 && kk == 0) return 0;  // nullify_modes('origin')
    double factor = 1;
    if (P.deconv_order) {
        double dab_n = P.tab_n[a] * P.tab_n[b];  // mesh.py:2797
        double dab_d = P.tab_s[a] * P.tab_s[b];  // mesh.py:2798
        factor = (dab_n * P.tab_n[kk]) / (dab_d * P.tab_s[kk]);  // mesh.py:2850-2853
        double f = factor;
        for (int o = 1; o < P.deconv_order; o++) factor *= f;  // factor **= deconv_order
    }
    i64 k2 = kab2 + kk * kk;
    double pk = P.C / (double)k2;  // interactions.py:2105
    if (P.long_range) pk = pk * exp((double)k2 * P.E);  // interactions.py:2110-2113
    return factor * pk;
}

// The same factor with everything that does not depend on the pencil index `a` hoisted
// by the caller (the fused FFT pass: b and kk are fixed per lane).  Operation order of the
// per-mode arithmetic is unchanged (n_a*n_b, s_a*s_b, (.*n_kk)/(.*s_kk), **order, C/k2 ...).
struct KspaceFixed {
    double n_b, s_b, n_kk, s_kk;
    i64 kb2;       // kb*kb (ka*ka is added per mode: kab2 = kb*kb + ka*ka as in the reference)
    i64 kk2;
    bool dead;     // b or kk on a Nyquist plane
    bool b0kk0;    // kb == 0 and kk == 0 (origin when ka == 0 too)
};
__device__ __forceinline__ KspaceFixed kspace_fix(i64 N, i64 b, i64 kk, double n_b, double s_b,
                                                  double n_kk, double s_kk) {
    const i64 nyq = N / 2;
    KspaceFixed F;
    F.dead = (b == nyq) || (kk == nyq);
    i64 kb = b - (b >= nyq ? N : 0);
    F.kb2 = kb * kb;
    F.kk2 = kk * kk;
    F.b0kk0 = (kb == 0) && (kk == 0);
    F.n_b = n_b;
    F.s_b = s_b;
    F.n_kk = n_kk;
    F.s_kk = s_kk;
    return F;
}
__device__ __forceinline__ KspaceFixed kspace_fix(const KspaceParams &P, i64 N, i64 b, i64 kk) {
    return kspace_fix(N, b, kk, P.tab_n[b], P.tab_s[b], P.tab_n[kk], P.tab_s[kk]);
}
// n_a = tab_n[a], s_a = tab_s[a] are fetched by the caller (global or LDS copy of the tables)
__device__ __forceinline__ double kspace_factor_fixed(const KspaceParams &P, const KspaceFixed &F,
                                                      i64 N, i64 a, double n_a, double s_a) {
#pragma clang fp contract(off)
    const i64 nyq = N / 2;
    if (F.dead || a == nyq) return 0;
    i64 ka = a - (a >= nyq ? N : 0);
    i64 kab2 = F.kb2 + ka * ka;  // interactions.py:2096 (integer)
    if (F.b0kk0 && ka == 0) return 0;
    double factor = 1;
    if (P.deconv_order) {
        double dab_n = n_a * F.n_b;  // mesh.py:2797
        double dab_d = s_a * F.s_b;  // mesh.py:2798
        factor = (dab_n * F.n_kk) / (dab_d * F.s_kk);  // mesh.py:2850-2853
        double f = factor;
        for (int o = 1; o < P.deconv_order; o++) factor *= f;
    }
    i64 k2 = kab2 + F.kk2;
    double pk = P.C / (double)k2;
    if (P.long_range) pk = pk * exp((double)k2 * P.E);
    return factor * pk;
}

// Separable arrangement of the same factor for the fused x passes.  With
// k2 = ka^2 + kb^2 + kk^2 the Gaussian of the long-range split factorises,
// exp(k2 E) = exp(ka^2 E) exp(kb^2 E) exp(kk^2 E), and so does the deconvolution, so one table
// per dimension index, t[a] = q_a^order * exp(ka^2 E), leaves a mode with one table value, one
// multiplication and the division by k2 — no exp and no power loop per mode (they are most of
// the arithmetic of a fused pass: 16 to 32 modes per lane and tile).  The sinc ratio of a
// dimension comes from the table q = n/sin(n); the mode numbers are 32-bit (N <= 2048:
// k2 < 2^24).  Differs from kspace_factor by rounding only (the three ratios and the three
// Gaussians are rounded separately, a few ulp); the Poisson solve is a floating-point path
// (FFT rounding already differs from FFTW's), tested against the CPU restatement to the
// tolerance stated in tests/test_gpu_pm.py.
__device__ __forceinline__ double kspace_tab_sep(const KspaceParams &P, int N, int a, double q_a) {
    const int nyq = N / 2;
    const int ka = a - (a >= nyq ? N : 0);
    double f = 1;
    if (P.deconv_order) {
        f = q_a;
        for (int o = 1; o < P.deconv_order; o++) f *= q_a;
    }
    if (P.long_range & 1) f *= exp((double)(ka * ka) * P.E);
    return f;
}
struct KspaceFixedS {
    double g;     // C * t_b * t_kk
    int kb2_kk2;  // kb*kb + kk*kk
    bool dead;    // b or kk on a Nyquist plane
};
__device__ __forceinline__ KspaceFixedS kspace_fix_sep(const KspaceParams &P, int N, int b, int kk,
                                                       double t_b, double t_kk) {
    const int nyq = N / 2;
    KspaceFixedS F;
    F.dead = (b == nyq) || (kk == nyq);
    const int kb = b - (b >= nyq ? N : 0);
    F.kb2_kk2 = kb * kb + kk * kk;
    F.g = P.C * t_b * t_kk;
    return F;
}
__device__ __forceinline__ double kspace_factor_sep(const KspaceFixedS &F, int N, int a,
                                                    double t_a) {
    const int nyq = N / 2;
    const int ka = a - (a >= nyq ? N : 0);
    const int k2 = F.kb2_kk2 + ka * ka;
    // branch-free (a select): nullify_modes('nyquist'), ('origin')
    const double f = t_a * F.g / (double)(k2 == 0 ? 1 : k2);
    return (F.dead | (a == nyq) | (k2 == 0)) ? 0.0 : f;
}
