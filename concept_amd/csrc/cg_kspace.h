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
    i64 kb2_kk2;   // kb*kb (ka*ka is added per mode: kab2 = kb*kb + ka*ka as in the reference)
    i64 kb2;
    i64 kk2;
    bool dead;     // b or kk on a Nyquist plane
    bool b0kk0;    // kb == 0 and kk == 0 (origin when ka == 0 too)
};
__device__ __forceinline__ KspaceFixed kspace_fix(const KspaceParams &P, i64 N, i64 b, i64 kk) {
    const i64 nyq = N / 2;
    KspaceFixed F;
    F.dead = (b == nyq) || (kk == nyq);
    i64 kb = b - (b >= nyq ? N : 0);
    F.kb2 = kb * kb;
    F.kk2 = kk * kk;
    F.kb2_kk2 = 0;
    F.b0kk0 = (kb == 0) && (kk == 0);
    F.n_b = P.tab_n[b];
    F.s_b = P.tab_s[b];
    F.n_kk = P.tab_n[kk];
    F.s_kk = P.tab_s[kk];
    return F;
}
__device__ __forceinline__ double kspace_factor_fixed(const KspaceParams &P, const KspaceFixed &F,
                                                      i64 N, i64 a) {
#pragma clang fp contract(off)
    const i64 nyq = N / 2;
    if (F.dead || a == nyq) return 0;
    i64 ka = a - (a >= nyq ? N : 0);
    i64 kab2 = F.kb2 + ka * ka;  // interactions.py:2096 (integer)
    if (F.b0kk0 && ka == 0) return 0;
    double factor = 1;
    if (P.deconv_order) {
        double dab_n = P.tab_n[a] * F.n_b;  // mesh.py:2797
        double dab_d = P.tab_s[a] * F.s_b;  // mesh.py:2798
        factor = (dab_n * F.n_kk) / (dab_d * F.s_kk);  // mesh.py:2850-2853
        double f = factor;
        for (int o = 1; o < P.deconv_order; o++) factor *= f;
    }
    i64 k2 = kab2 + F.kk2;
    double pk = P.C / (double)k2;
    if (P.long_range) pk = pk * exp((double)k2 * P.E);
    return factor * pk;
}
