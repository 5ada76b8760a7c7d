// Can a layer of the mesh be handed from the z pass to the y pass of the FFT INSIDE an XCD — the 32
// workgroups of an XCD keep a layer's rows in registers, exchange them through a scratch window
// that stays in that XCD's 4 MB L2, and write the layer once — instead of two in-place passes over
// HBM ("fewer fabric passes")?  This probe moves the data of that schedule without
// any transform and checks that every value arrives:
//   A  "z pass": every row (512 complex = 8 KB, contiguous) read and written in place
//   B  "y pass": tiles of 8 adjacent kz x 1024 y (128-byte segments at a stride of 8 KB) read and
//      written in place
//   C  hand-over: 256 workgroups of 1024 lanes, one per CU (LDS allocation), group g = block % 8 =
//      the 32 workgroups of one XCD (checked with XCC_ID); a group takes the layers x = g, g + 8, ..;
//      per layer a workgroup loads its 32 rows into registers (16 complex per lane), then in R
//      rounds: the lanes whose kz column belongs to the round write their 16 values into the
//      group's scratch window [kz block][y][8 kz] (two windows, alternating) -> barrier of the 32
//      workgroups (device atomic + bounded spin; stores waited for before, L1 invalidated after)
//      -> every workgroup reads half a kz block's column (512 y x 8 kz) from the window and stores
//      it to the layer in the y pass's pattern.  HBM sees the layer once each way; the window
//      traffic should stay in L2.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/xcd_handover_probe tools/xcd_handover_probe.cpp
// run:   tools/xcd_handover_probe [layers=1024] [rounds=4] [mode: 1 = window read with sc1 loads,
//        2 = the CU's L1 invalidated by one wave (buffer_inv sc0) + plain loads, 0 = buffer_inv sc1]
//        tools/xcd_handover_probe column [mode]     (the second question, further down)
// Measured (MI355X, round 4; profiles/r04_xcd_handover_probe.txt, tools/pmc_handover.sh): A + B 7.0 ms,
// C 5.5-6.0 ms with every value in place, but two passes' WRITE_SIZE in every variant and two
// passes' FETCH_SIZE unless the window is <= 1 MB per XCD: the window does not stay in the L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NY = 1024, NZ = 512;              // rows per layer, complex values per row
constexpr size_t LAYER = (size_t)NY * NZ;       // complex values per layer (8 MB)
constexpr int GROUPS = 8, WG_PER_GROUP = 32, LANES = 1024, PER = 16;
constexpr unsigned SPIN_LIMIT = 1u << 22;
__device__ int g_l2mode = 0;   // C: window loads that bypass the L1 (sc1) instead of an invalidate, layer traffic non-temporal
__device__ int g_inv_sc0 = 0;   // the invalidate after a barrier at workgroup scope (the CU's L1 only) instead of agent scope
__device__ int g_no_inv = 0, g_no_sleep = 0;   // parts of the barrier switched off (timing only)

__device__ __forceinline__ d2 pattern(unsigned x, unsigned y, unsigned kz) {
    d2 v;
    v.x = (double)(x * 1024u + y) + 0.5;
    v.y = (double)kz - 0.25 * (double)x;
    return v;
}
__global__ void k_fill(d2 *mesh, unsigned nx) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nx * LAYER) return;
    unsigned kz = i % NZ, y = (i / NZ) % NY, x = (unsigned)(i / LAYER);
    mesh[i] = pattern(x, y, kz);
}
__global__ void k_check(const d2 *mesh, unsigned nx, unsigned long long *bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nx * LAYER) return;
    unsigned kz = i % NZ, y = (i / NZ) % NY, x = (unsigned)(i / LAYER);
    d2 w = pattern(x, y, kz), v = mesh[i];
    if (v.x != w.x || v.y != w.y) atomicAdd(bad, 1ull);
}

// A: rows in place (a workgroup per 32 rows of a layer, like C's loads)
__global__ __launch_bounds__(LANES) void k_rows(d2 *mesh) {
    d2 *p = mesh + (size_t)blockIdx.x * (32 * NZ);
    d2 v[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) v[j] = p[threadIdx.x + LANES * j];
#pragma unroll
    for (int j = 0; j < PER; j++) p[threadIdx.x + LANES * j] = -v[j];   // (B negates back)
}
// B: column tiles in place: tile = (layer, kz block of 8); lane -> (y = t / 8 + 128 j, kz = t % 8)
__global__ __launch_bounds__(LANES) void k_cols(d2 *mesh) {
    const unsigned tile = blockIdx.x, x = tile / (NZ / 8), kb = tile % (NZ / 8);
    d2 *p = mesh + (size_t)x * LAYER + 8 * kb;
    const unsigned y0 = threadIdx.x / 8, c = threadIdx.x % 8;
    d2 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = p[(size_t)(y0 + 128 * j) * NZ + c];
#pragma unroll
    for (int j = 0; j < 8; j++) p[(size_t)(y0 + 128 * j) * NZ + c] = -v[j];
}

// the 32 workgroups of a group: arrive, wait (bounded), L1 invalidated on the way out
__device__ __forceinline__ bool group_barrier(unsigned *counter, unsigned target, unsigned *abort_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have reached the L2
    __syncthreads();
    __shared__ unsigned ok;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0, good = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (!g_no_sleep) __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
        }
        ok = good;
    }
    // what the others wrote is read from the L2: the CU's L1 is invalidated ONCE, by the first
    // wave, before the others go on (every wave doing it: 30 us per barrier instead of ~2)
    if (threadIdx.x < 64 && !g_no_inv) {
        if (g_inv_sc0) asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return ok != 0;
}

// the same with a flag per member instead of one counter: a member publishes its epoch, the first
// 32 lanes of its first wave each watch one member's flag (no read-modify-write on a shared word)
__device__ __forceinline__ bool group_barrier_flags(unsigned *flags, unsigned w, unsigned epoch,
                                                    unsigned *abort_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned ok2;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(flags + 16 * w, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0, good = 1;
        while (true) {
            unsigned f = threadIdx.x < WG_PER_GROUP
                             ? __hip_atomic_load(flags + 16 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : epoch;
            if (__all((int)(f - epoch) >= 0)) break;
            if (++spins > SPIN_LIMIT) { good = 0; break; }
        }
        if (threadIdx.x == 0) {
            if (!good) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok2 = good;
        }
        if (!g_no_inv) {
            if (g_inv_sc0) asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    return ok2 != 0;
}
// barriers alone: what `n` of them cost a group
template <bool FLAGS>
__global__ __launch_bounds__(LANES) void k_barriers(unsigned n, unsigned *counters, unsigned *flags,
                                                    unsigned *abort_flag) {
    extern __shared__ char force_one_per_cu[];
    const unsigned g = blockIdx.x % GROUPS, w = blockIdx.x / GROUPS;
    for (unsigned e = 1; e <= n; e++) {
        const bool ok = FLAGS ? group_barrier_flags(flags + 16 * WG_PER_GROUP * g, w, e, abort_flag)
                              : group_barrier(counters + 64 * g, e * WG_PER_GROUP, abort_flag);
        if (!ok) return;
    }
}

template <int R, bool PIPE>
__global__ __launch_bounds__(LANES) void k_handover(d2 *mesh, unsigned nx, d2 *scratch,
                                                    unsigned *counters, unsigned *abort_flag,
                                                    unsigned *xcc_of_block, unsigned *flags) {
    extern __shared__ char force_one_per_cu[];
    constexpr int KZ_ROUND = NZ / R, KB_ROUND = KZ_ROUND / 8;   // kz values / blocks per round
    constexpr size_t WINDOW = (size_t)KZ_ROUND * NY;            // complex values per window
    const unsigned g = blockIdx.x % GROUPS, w = blockIdx.x / GROUPS;   // group, member 0..31
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc_of_block[blockIdx.x] = id & 0xf;
    }
    d2 *win = scratch + (size_t)g * 2 * WINDOW;
    unsigned *counter = counters + 64 * g;   // (a line of its own)
    unsigned epoch = 0;
    const unsigned col = threadIdx.x % NZ, row0 = threadIdx.x / NZ;   // lane's kz, first row (0/1)
    d2 v[PER];
    // the member's 32 rows, y = 32 w + row0 + 2 j
    // (PIPE: the first layer here — a lane's registers are free once its round has written them
    // to the window, and are loaded with the NEXT layer's values right then)
    if (PIPE && g < nx) {
        const d2 *first = mesh + (size_t)g * LAYER;
#pragma unroll
        for (int j = 0; j < PER; j++) v[j] = first[32u * w * NZ + threadIdx.x + LANES * j];
    }
    for (unsigned x = g; x < nx; x += GROUPS) {
        d2 *layer = mesh + (size_t)x * LAYER;
        if (!PIPE) {
            if (g_l2mode) {
#pragma unroll
                for (int j = 0; j < PER; j++) v[j] = __builtin_nontemporal_load(&layer[32u * w * NZ + threadIdx.x + LANES * j]);
            } else {
#pragma unroll
                for (int j = 0; j < PER; j++) v[j] = layer[32u * w * NZ + threadIdx.x + LANES * j];
            }
        }
#pragma unroll 1
        for (int r = 0; r < R; r++) {
            d2 *buf = win + (size_t)((epoch & 1u) * WINDOW);
            if ((int)(col / KZ_ROUND) == r) {
                const unsigned kl = col - r * KZ_ROUND;   // kz inside the round
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const unsigned y = 32 * w + row0 + 2 * j;
                    buf[((kl / 8) * NY + y) * 8 + kl % 8] = v[j];
                }
            }
            epoch++;
            if (!(flags ? group_barrier_flags(flags + 16 * WG_PER_GROUP * g, w, epoch, abort_flag)
                        : group_barrier(counter, epoch * WG_PER_GROUP, abort_flag))) return;
            if (PIPE && (int)(col / KZ_ROUND) == r && x + GROUPS < nx) {
                // (in place, under the branch's lane mask: written as an instruction because the
                // compiler would give the loaded values registers of their own beside the live
                // ones of the other lanes — 35 spilled registers; the wait is the next barrier's)
                const d2 *next = layer + (size_t)GROUPS * LAYER;
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const unsigned off = 16u * (32u * w * NZ + threadIdx.x + LANES * j);
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(v[j]) : "v"(off), "s"(next) : "memory");
                }
            }
            // read: 2 * KB_ROUND half columns over 32 members (R = 4: 32 halves, one each)
            constexpr int HALVES = 2 * KB_ROUND;
            for (int h = w; h < HALVES; h += WG_PER_GROUP) {
                const unsigned kb = h / 2, half = h % 2;
                const d2 *src = buf + (kb * NY + 512 * half) * 8;
                d2 u[4];
                if (g_l2mode == 1) {
                    // past the L1, from the L2 the other members' stores went to
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const unsigned off = 16u * (threadIdx.x + LANES * j);
                        asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(u[j]) : "v"(off), "s"(src) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) u[j] = src[threadIdx.x + LANES * j];
                }
                // value (y = 512 half + (t + 1024 j) / 8, kz = 8 (kb + r KB_ROUND) + t % 8)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned y = 512 * half + (threadIdx.x + LANES * j) / 8;
                    if (g_l2mode) __builtin_nontemporal_store(u[j], &layer[y * NZ + 8 * (kb + r * KB_ROUND) + threadIdx.x % 8]);
                    else layer[y * NZ + 8 * (kb + r * KB_ROUND) + threadIdx.x % 8] = u[j];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Second question (argv[1] = "column"): a hand-over between NEIGHBOURING kernels.  The deposit
// writes the mesh tile by tile (16^3 doubles: 256 segments of 128 bytes), the forward z pass reads
// and rewrites it row by row (1024 doubles).  A column of 64 tiles along z is 256 such rows = 2 MB:
// if the 32 workgroups of an XCD write a column's tiles and transform its rows right away, the rows
// are read from the L2 they were just written to and overwritten there — one write-back per line
// instead of write, read, write.
//   D  tile stores over the whole mesh (a workgroup per tile, the tiles of an XCD's eighth in order)
//   A  rows in place (k_rows above, on the same 8.6 GB)
//   F  fused: XCD g owns the columns (ta in [8g, 8g+8), tb); member w stores tiles tc = 2w, 2w+1,
//      barrier, then reads, negates and stores rows 8w .. 8w+7 of the column
// Measured: D + A 4.7 ms; F 4.6 / 5.0 / 5.5 ms with columns of 2 MB / 1 MB / 512 KB (-DCOL_A=16, 8, 4).
// WRITE_SIZE of F: 1.74 / 1.01 / 1.00 passes (the tiles' write-back is saved when the column
// fits), FETCH_SIZE: one full pass in all three — a row that is read is fetched from the fabric
// although its line was stored into the same L2 just before.  (Mode 1 — sc1 loads without any
// invalidate — once returned 24 % stale rows here: use mode 2.)
// ---------------------------------------------------------------------------
#ifndef COL_A
#define COL_A 16   // x extent of a tile (-DCOL_A=8, 4: columns of 1 MB, 512 KB instead of 2 MB)
#endif
constexpr int MT = 16, MN = 1024, MNT = MN / MT;   // tile edge, mesh edge (doubles), tiles per edge
constexpr int MTA = COL_A, MNTA = MN / MTA;
__device__ __forceinline__ double rpattern(unsigned a, unsigned b, unsigned c) {
    return (double)(a * 1024u + b) + (double)c / 2048.0;
}
__device__ __forceinline__ void store_tile(double *mesh, unsigned ta, unsigned tb, unsigned tc, unsigned t, unsigned nthreads) {
    // 16 x 16 segments of 16 doubles = 2048 d2; thread -> d2 number i: segment i / 8, pair i % 8
    for (unsigned i = t; i < 128u * MTA; i += nthreads) {
        const unsigned seg = i / 8u, pair = i % 8u, a = MTA * ta + seg / MT, b = MT * tb + seg % MT,
                       c = MT * tc + 2u * pair;
        d2 v;
        v.x = rpattern(a, b, c);
        v.y = rpattern(a, b, c + 1u);
        *(d2 *)(mesh + ((size_t)a * MN + b) * MN + c) = v;
    }
}
__global__ __launch_bounds__(512) void k_tiles(double *mesh) {
    const unsigned ntiles = MNTA * MNT * MNT, b = blockIdx.x, tile = (b % 8u) * (ntiles / 8u) + b / 8u;
    store_tile(mesh, tile / (MNT * MNT), (tile / MNT) % MNT, tile % MNT, threadIdx.x, 512u);
}
__global__ void k_check_neg(const double *mesh, unsigned long long *bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned c = i % MN, b = (i / MN) % MN, a = (unsigned)(i / ((size_t)MN * MN));
    if (mesh[i] != -rpattern(a, b, c)) atomicAdd(bad, 1ull);
}
__global__ __launch_bounds__(LANES) void k_column(double *mesh, unsigned *flags, unsigned *abort_flag, int mode) {
    extern __shared__ char force_one_per_cu[];
    const unsigned g = blockIdx.x % GROUPS, w = blockIdx.x / GROUPS;
    unsigned epoch = 0;
    for (unsigned col = 0; col < (MNTA / 8u) * MNT; col++) {
        const unsigned ta = (MNTA / 8u) * g + col / MNT, tb = col % MNT;
        store_tile(mesh, ta, tb, 2u * w, threadIdx.x, LANES);
        store_tile(mesh, ta, tb, 2u * w + 1u, threadIdx.x, LANES);
        epoch++;
        if (!group_barrier_flags(flags + 16 * WG_PER_GROUP * g, w, epoch, abort_flag)) return;
        // rows 8w .. 8w+7 of the column's 256 (row r: a = 16 ta + r / 16, b = 16 tb + r % 16)
        constexpr int NJ = MTA / 4;   // rows per member: MTA * 16 / 32, of 512 d2 each
        d2 u[NJ];
        const d2 *rows[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const unsigned i = threadIdx.x + LANES * j, r = (MTA / 2u) * w + i / 512u, a = MTA * ta + r / MT,
                           b = MT * tb + r % MT;
            rows[j] = (const d2 *)(mesh + ((size_t)a * MN + b) * MN) + i % 512u;
            if (mode == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(u[j]) : "v"(rows[j]) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(u[j]) : "v"(rows[j]) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NJ; j++) *(d2 *)rows[j] = -u[j];
    }
}
int main_column(int mode) {
    double *mesh; unsigned *flags, *abort_flag; unsigned long long *bad;
    const size_t n = (size_t)MN * MN * MN;
    CK(hipMalloc(&mesh, n * 8)); CK(hipMalloc(&flags, 16 * WG_PER_GROUP * GROUPS * 4)); CK(hipMalloc(&abort_flag, 4));
    CK(hipMalloc(&bad, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = 96 * 1024;
    CK(hipFuncSetAttribute((const void *)k_column, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    { const int noinv = mode == 1, sc0 = mode == 2;
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_no_inv), &noinv, 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_inv_sc0), &sc0, 4)); }
    printf("F: %s\n", mode == 1 ? "rows read with sc1 loads, no invalidate" : mode == 2 ? "the CU's L1 invalidated by one wave (buffer_inv sc0), plain loads" : "L1 invalidated at agent scope (buffer_inv sc1), plain loads");
    float ms_d, ms_a, ms;
    const double gb = n * 8 / 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_tiles, dim3(MNTA * MNT * MNT), dim3(512), 0, 0, mesh);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_d, e0, e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rows, dim3(MN * (NY / 32)), dim3(LANES), 0, 0, (d2 *)mesh);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_a, e0, e1));
        printf("D tile stores %.3f ms (%.2f TB/s written)   A rows in place %.3f ms   D + A %.3f ms\n", ms_d,
               gb / ms_d, ms_a, ms_d + ms_a);
    }
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemset(flags, 0, 16 * WG_PER_GROUP * GROUPS * 4)); CK(hipMemset(abort_flag, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_column, dim3(GROUPS * WG_PER_GROUP), dim3(LANES), lds, 0, mesh, flags, abort_flag, mode);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned ab; CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
        printf("F column of tiles (%d KB) written and its rows rewritten by the same XCD: %.3f ms%s\n", MTA * 128, ms,
               ab ? "  ABORTED (a barrier timed out)" : "");
        if (ab) break;
    }
    CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(k_check_neg, dim3((unsigned)(n / 256)), dim3(256), 0, 0, mesh, bad);
    unsigned long long nbad; CK(hipMemcpy(&nbad, bad, 8, hipMemcpyDeviceToHost));
    printf("values that are not the negated pattern: %llu of %zu\n", nbad, n);
    return nbad != 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 'c') return main_column(argc > 2 ? atoi(argv[2]) : 2);
    const unsigned nx = argc > 1 ? (unsigned)atoi(argv[1]) : 1024;
    const int rounds = argc > 2 ? atoi(argv[2]) : 4;
    const int l2mode = argc > 3 ? atoi(argv[3]) : 1;   // 0: L1 invalidate after the barrier, plain accesses
    d2 *mesh, *scratch; unsigned *counters, *abort_flag, *xcc; unsigned long long *bad;
    const size_t total = (size_t)nx * LAYER;
    CK(hipMalloc(&mesh, total * sizeof(d2)));
    CK(hipMalloc(&scratch, (size_t)GROUPS * 2 * LAYER * sizeof(d2)));   // (enough for R = 1)
    CK(hipMalloc(&counters, 64 * GROUPS * 4)); CK(hipMalloc(&abort_flag, 4)); CK(hipMalloc(&xcc, 256 * 4));
    CK(hipMalloc(&bad, 8));
    unsigned *flags; CK(hipMalloc(&flags, 16 * WG_PER_GROUP * GROUPS * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned fill_blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_fill, dim3(fill_blocks), dim3(256), 0, 0, mesh, nx);
    CK(hipDeviceSynchronize());
    float ms;
    const double gb = 2.0 * total * sizeof(d2) / 1e9;   // one read + one write of the mesh
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rows, dim3(nx * (NY / 32)), dim3(LANES), 0, 0, mesh);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        float ms_a = ms;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_cols, dim3(nx * (NZ / 8)), dim3(LANES), 0, 0, mesh);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("A rows in place %.3f ms (%.2f TB/s)   B column tiles in place %.3f ms (%.2f TB/s)   A + B %.3f ms\n",
               ms_a, gb / ms_a, ms, gb / ms, ms_a + ms);
    }
    auto kern = rounds == 2 ? k_handover<2, false> : rounds == 8 ? k_handover<8, false> : k_handover<4, false>;
    auto kern_pipe = rounds == 2 ? k_handover<2, true> : rounds == 8 ? k_handover<8, true> : k_handover<4, true>;
    // (PIPE: the next layer's loads under the present layer's rounds)
    const int lds = 96 * 1024;   // one workgroup per CU
    CK(hipFuncSetAttribute((const void *)k_barriers<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void *)k_barriers<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int variant = 0; variant < 3; variant++)
    for (int fl = 0; fl < 2; fl++) {
        const unsigned nb = 2000;
        const int no_inv = variant == 1, no_sleep = variant == 2;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_no_inv), &no_inv, 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_no_sleep), &no_sleep, 4));
        CK(hipMemset(counters, 0, 64 * GROUPS * 4)); CK(hipMemset(abort_flag, 0, 4));
        CK(hipMemset(flags, 0, 16 * WG_PER_GROUP * GROUPS * 4));
        CK(hipEventRecord(e0));
        if (fl) hipLaunchKernelGGL(k_barriers<true>, dim3(GROUPS * WG_PER_GROUP), dim3(LANES), lds, 0, nb, counters, flags, abort_flag);
        else hipLaunchKernelGGL(k_barriers<false>, dim3(GROUPS * WG_PER_GROUP), dim3(LANES), lds, 0, nb, counters, flags, abort_flag);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned ab; CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
        printf("barrier of an XCD's 32 workgroups, %s%s: %.2f us each%s\n", fl ? "a flag per member" : "one counter",
               variant == 1 ? ", no L1 invalidate" : variant == 2 ? ", no s_sleep" : "", 1e3 * ms / nb, ab ? "  ABORTED" : "");
    }
    { const int z = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_no_inv), &z, 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_no_sleep), &z, 4)); }
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void *)kern_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    { const int m = l2mode, noinv = l2mode == 1, sc0 = l2mode == 2;
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_l2mode), &m, 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_no_inv), &noinv, 4));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_inv_sc0), &sc0, 4)); }
    printf("C: %s\n", l2mode == 1 ? "window loads past the L1 (sc1), no invalidate, layer loads and stores non-temporal"
                       : l2mode == 2 ? "the CU's L1 invalidated after every barrier by one wave (buffer_inv sc0), plain window loads, layer loads and stores non-temporal"
                                     : "L1 invalidated after every barrier (buffer_inv sc1), plain accesses");
    for (int rep = 0; rep < 8; rep++) {
        const bool fl = rep >= 2 && rep != 4 && rep != 5;
        const int pipe = rep >= 4;
        CK(hipMemset(counters, 0, 64 * GROUPS * 4)); CK(hipMemset(abort_flag, 0, 4));
        CK(hipMemset(flags, 0, 16 * WG_PER_GROUP * GROUPS * 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(pipe ? kern_pipe : kern, dim3(GROUPS * WG_PER_GROUP), dim3(LANES), lds, 0, mesh, nx, scratch,
                           counters, abort_flag, xcc, fl ? flags : (unsigned *)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned ab; CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
        printf("C hand-over, %d rounds per layer, %s%s: %.3f ms (%.2f TB/s of one read + one write)%s\n",
               rounds == 2 || rounds == 8 ? rounds : 4, fl ? "flags" : "counter",
               pipe ? ", next layer loaded under the rounds" : "", ms, gb / ms,
               ab ? "  ABORTED (a barrier timed out)" : "");
        if (ab) break;
    }
    std::vector<unsigned> hx(256);
    CK(hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost));
    int mixed = 0;
    for (int b = 0; b < 256; b++) if (hx[b] != hx[b % 8]) mixed++;
    printf("XCC_ID of blocks 0..7: %u %u %u %u %u %u %u %u; blocks whose XCC differs from block (b %% 8)'s: %d\n",
           hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7], mixed);
    CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(k_check, dim3(fill_blocks), dim3(256), 0, 0, mesh, nx, bad);
    unsigned long long nbad; CK(hipMemcpy(&nbad, bad, 8, hipMemcpyDeviceToHost));
    printf("values that did not arrive: %llu of %zu\n", nbad, total);
    return nbad != 0;
}
