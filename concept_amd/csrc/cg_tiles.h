// cg_tiles.h — device helpers shared by the particle kernels: the reference's mod(),
// the tile / bucket key of a particle and per-wavefront run detection.
#pragma once
#include "cg_internal.h"

// Component.drift's mod (species.py:2194-2196, commons.py:5103-5110): numpy floored
// modulo, then a result that rounded to exactly boxsize becomes 0.
__device__ __forceinline__ double ref_mod(double x, double L) {
    if (x > 0 && x < L) return x;  // fmod(x, L) == x, same sign as L
    double m = fmod(x, L);          // exact
    if (m != 0) {
        if (m < 0) m += L;          // npy_divmod: sign fix-up (L > 0)
    } else {
        m = 0.0;                    // copysign(0, L)
    }
    if (m == L) m = 0;              // commons.py:5108-5109
    return m;
}

// a pair's squared distance (gravity.py:306, r2 = x**2 + y**2 + z**2 in that order): the two
// additions fused with their products — three FP64 instructions instead of five in the loops
// that are bound by their issue rate; within an ulp of the unfused sum (the reference's own
// build contracts them as its compiler sees fit: -O3 -ffast-math -march=native, src/Makefile:175-189)
__device__ __forceinline__ double sr_r2(double x, double y, double z) {
    return __builtin_fma(z, z, __builtin_fma(y, y, x * x));
}

// 32-bit: x is positive and < gridsize + 2*nghosts + 1, so truncating to int equals the
// reference's truncation to Py_ssize_t (one v_cvt_i32_f64 instead of the double -> int64
// sequence)
__device__ __forceinline__ int lower_cell(double pos, double off, double scale, int g, int N) {
    double x = (pos - off) * scale;
    int a = (int)x - g;
    a = a < 0 ? a + N : a;
    return a >= N ? a - N : a;
}
// key = 8*tile + bucket; bucket bit d is set when the lower cell is the last of
// the tile in dimension d, i.e. the CIC cloud reaches the next tile there
// (bit 2: x, bit 1: y, bit 0: z).  The pull-deposit of a tile reads its own 8
// buckets plus the matching boundary buckets of its 7 lower neighbours.
constexpr unsigned kNoTile = 0xffffffffu;
__device__ __forceinline__ unsigned tile_of(double x, double y, double z, const CicGeom &geo,
                                            int g, i64 N, const TileGeom &t, i64 x0) {
    const int Ni = (int)N;
    int cx = lower_cell(x, geo.off[0], geo.scale, g, Ni) - (int)x0;  // local layer of this domain
    if (cx < 0 || cx >= t.ntx * t.tx) return kNoTile;                 // not owned here
    unsigned ca = (unsigned)cx;
    unsigned cb = (unsigned)lower_cell(y, geo.off[1], geo.scale, g, Ni);
    unsigned cc = (unsigned)lower_cell(z, geo.off[2], geo.scale, g, Ni);
    // the tile extent is a power of two (cg_create): shifts and masks, not divisions
    const unsigned sh = (unsigned)(__ffs(t.tx) - 1), last = (unsigned)t.tx - 1u;
    unsigned a = ca >> sh, b = cb >> sh, c = cc >> sh;
    unsigned f = (((ca & last) == last) ? 4u : 0u) | (((cb & last) == last) ? 2u : 0u) |
                 (((cc & last) == last) ? 1u : 0u);
    return ((a * t.nty + b) * t.ntz + c) * 8u + f;
}

// For the calling wave (all 64 lanes active): runs of consecutive lanes with
// equal key.  Returns the first lane of this lane's run and the run length.
__device__ __forceinline__ void wave_runs(unsigned key, int lane, int &run_start, int &run_len) {
    unsigned prev = __shfl_up(key, 1);
    bool head = (lane == 0) || (key != prev);
    unsigned long long mask = __ballot(head);
    unsigned long long below = mask & (~0ull >> (63 - lane));  // heads at lanes <= lane
    run_start = 63 - __clzll(below);
    unsigned long long above = (lane == 63) ? 0ull : (mask >> (lane + 1));
    int next = above ? (lane + 1 + (__ffsll((long long)above) - 1)) : 64;
    run_len = next - run_start;
}

