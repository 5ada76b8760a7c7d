// cg_internal.h — shared declarations of libconcept_gpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "concept_gpu.h"

typedef int64_t i64;

void cg_set_error(const char *fmt, ...);

#define CG_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            cg_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,     \
                         __LINE__);                                                           \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

#define CG_FFT(call)                                                                          \
    do {                                                                                      \
        rocfft_status s_ = (call);                                                            \
        if (s_ != rocfft_status_success) {                                                    \
            cg_set_error("%s failed: rocfft status %d (%s:%d)", #call, (int)s_, __FILE__,     \
                         __LINE__);                                                           \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

#define CG_CHECK(cond, ...)                                                                   \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            cg_set_error(__VA_ARGS__);                                                        \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

// Geometry of the CIC index map for one domain, computed on the host with
// the reference's own expressions (mesh.py:1577-1606 deposit, :408-432 gather).
struct CicGeom {
    double off[3];  // offset_x/y/z
    double scale;   // (1/cellsize)*(1 - machine_eps)
};

// How a global x cell index maps to a layer of the local mesh buffer.
// Single domain: the buffer is the whole periodic mesh (layer = x mod N).
// x-slab domains (one per GPU): the buffer holds layers [x0-G, x0+nxl+G) of
// the global mesh — nxl owned layers between G ghost layers — and y, z stay
// periodic.  Cell (x, j, k) lives at mesh[(layer*ny + j)*pad + k]: a layer is ny >= N rows
// of pad doubles (ny = N + 1 when N is a multiple of 16, see cg_ctx::ny).
struct XMap {
    i64 x0;    // first owned global layer
    i64 nxl;   // owned layers
    int G;     // ghost layers on each side (0 when periodic)
    int periodic;
};
__host__ __device__ inline i64 cg_xlayer(const XMap &m, i64 a, i64 N) {
    if (m.periodic) {
        a = a < 0 ? a + N : a;
        return a >= N ? a - N : a;
    }
    i64 l = a - m.x0;
    if (l < -(i64)m.G) l += N;             // periodic image across the box seam
    else if (l >= m.nxl + m.G) l -= N;
    l += m.G;
    // a cell outside the local layers can only come from a particle this domain does
    // not own (host-side contract: exchange before use); never index out of the buffer
    l = l < 0 ? 0 : l;
    return l >= m.nxl + 2 * m.G ? m.nxl + 2 * m.G - 1 : l;
}

// Tile decomposition used for the particle memory order and the LDS-tiled
// deposit / gather kernels.
struct TileGeom {
    int tx, ty, tz;     // tile extent in cells
    int ntx, nty, ntz;  // tiles per dimension
};

struct cg_ctx {
    cg_params p;
    hipStream_t stream = nullptr;
    i64 N = 0, pad = 0;          // grid size and padded innermost length N+2
    XMap xmap{};                 // local layers of the mesh buffer
    double *mesh = nullptr;      // double[layers][N][pad], layers = nxl + 2G
    double *mesh0 = nullptr;     // first OWNED layer (= mesh + G*N*pad)
    i64 mesh_doubles = 0;
    // rows per x layer.  N rows are used; with N a power of two the layer stride N*pad*8 B
    // is 2^17 x odd at 1024^3 and the 1024 row segments an x-pass tile touches alias onto
    // few memory channels (measured with tools/stride_probe.cpp: 3.99 ms per in-place
    // sweep at stride 8,519,680 B, 3.52 ms at 8,528,000 B): one unused row per layer
    // breaks the pattern.
    i64 ny = 0;
    // Fourier view: where the modes of this context live and which of them.  Mode (ki, kj, kk)
    // with array indices (a, b, kk), b in [f_j0, f_j0 + f_nj), sits at
    // four[a*f_si + (b - f_j0)*cp + kk], cp = pad/2.  Single domain: the mesh itself, in place
    // (f_si = ny*cp, all N rows b).  x-slab domains: the transposed buffer of the distributed
    // FFT, complex[N][JB + 1][cp] with the rows b of this domain's block [rank*JB, (rank+1)*JB)
    // (fft.c:55-72: FFTW-MPI's transposed output is distributed the same way), bound by the
    // caller with cg_dist_bind_fourier.
    double2 *four = nullptr;
    i64 f_si = 0;
    int f_j0 = 0, f_nj = 0;
    double *fetch_tmp = nullptr; // lazily allocated, for CG_FETCH_MESH_FOURIER
    // k-space tables: numerator n(k) and denominator sin(n(k)) by array index
    double *ktab_n = nullptr, *ktab_s = nullptr, *ktab_q = nullptr;
    // rocFFT
    rocfft_plan plan_fwd = nullptr, plan_bwd = nullptr;
    struct DistPlans *dist_plans = nullptr;  // x-slab domains on the rocFFT backend (cg_context.hip)
    rocfft_execution_info info_fwd = nullptr, info_bwd = nullptr;
    void *fft_work = nullptr;
    size_t fft_work_bytes = 0;
    // hand-written FFT (cg_fft.hip): twiddles exp(-2 pi i k/N) as double2[N]
    bool custom_fft = false;
    double *fft_tw = nullptr;
    // side streams of the short-range sweep (its interior and face launches are independent and
    // run side by side: a clustered box otherwise waits for the few dense tiles of each launch
    // in turn), created at the first sweep
    // a sub-step's first pass, deferred to the cell list that follows it (cg_substep.h)
    struct SubstepBegin *sub_begin = nullptr;
    bool sub_pending = false;
    long long *sub_counts = nullptr;     // where the pending pass leaves the populations (DEV)
    unsigned *sub_partial = nullptr;     // ... per workgroup first
    size_t sub_partial_bytes = 0;
    unsigned *sr_active = nullptr;   // [0] count, [64..] the cells that hold an active receiver
    size_t sr_active_bytes = 0;
    hipStream_t sr_streams[3] = {nullptr, nullptr, nullptr};
    hipEvent_t sr_fork = nullptr, sr_join[3] = {nullptr, nullptr, nullptr};
    unsigned long long *sr_stats = nullptr;   // cg_shortrange_stats: device counters while on
    unsigned char *sr_tile_active = nullptr;  // cells sweep with rungs: one byte per tile
    double *sr_sparse_partial = nullptr;      // cg_shortrange_sparse: per-workgroup partial sums
    size_t sr_tile_active_cap = 0;
    hipEvent_t *pass_events = nullptr;  // when set: 6 events recorded around the 5 passes
    // particle sort scratch (owned, grown on demand)
    TileGeom tiles{};
    unsigned int *tile_count = nullptr;   // [ntiles + 1]
    unsigned int *tile_cursor = nullptr;  // [ntiles]
    void *scan_tmp = nullptr;
    // tile kernels: heavy tiles first when the tiles' populations are far from equal
    // (cgk_tile_order; off = the plain walk)
    unsigned int *tile_order = nullptr;  // the list [tile_order_cap], counters and bytes behind
    unsigned int *tile_order_buf = nullptr;
    unsigned tile_order_cap = 0;
    unsigned *tile_order_seen = nullptr;  // pinned: length of the list of the last build done
    bool tile_order_on = false;          // list and bytes of one build are there
    int tile_order_mode = -1;            // from CONCEPT_GPU_TILE_ORDER_MIN: 0 off, 1 on, 2 also on small boxes
    unsigned tile_order_floor = 1536;    // |CONCEPT_GPU_TILE_ORDER_MIN|
    const void *tile_order_src[2] = {nullptr, nullptr};  // (start, count) it was made from
    size_t scan_tmp_bytes = 0;
    void *sr_tmp = nullptr;  // short-range cell-list counters
    size_t sr_tmp_bytes = 0;
    // the dense tiles' sweep (cg_shortrange_dense.hip): counters, pinned read-back, lists, stream;
    // srd_look: what is known about the cell lists built last (keyed by their offsets' address)
    struct SrdLook {
        const unsigned *off = nullptr;
        i64 nt = 0;
        int min_pop = 0;
        hipEvent_t ev = nullptr;
        bool harvested = true;   // its result has been counted in srd_quiet
    } srd_look[4];
    int srd_look_next = 0, srd_idle = 0;
    int srd_quiet = 0;   // completed looks in a row that found no dense tile
    bool srd_hilbert_done = false;
    unsigned *srd_host = nullptr;
    void *srd_small = nullptr, *srd_buf = nullptr, *srd_rung = nullptr;
    size_t srd_small_bytes = 0, srd_buf_bytes = 0, srd_rung_bytes = 0;
    hipStream_t srd_stream = nullptr;
    hipEvent_t srd_fork = nullptr, srd_join = nullptr;
    void *sr_sub_tmp = nullptr;  // rows of the dense tiles while they are re-ordered by sub-cell
    size_t sr_sub_bytes = 0;
    i64 ntiles = 0;
    CicGeom geom_deposit{}, geom_gather{};
    // tile histogram of the NEXT drift prepared by cg_gather_kick_tiled_prepare
    bool prep_valid = false;
    const double *prep_pos = nullptr, *prep_mom = nullptr;
    i64 prep_n = 0;
    double prep_dtm = 0;
    // caller-owned list of the particles leaving the slab with the prepared drift
    i64 *emig_idx = nullptr;
    unsigned *emig_count = nullptr;
    i64 emig_cap = 0;
    // fused kick + drift + scatter on x-slab domains: the particles the drift takes out of the
    // slab are appended here as rows of 8 doubles (pos 3, mom 3, id bits, unused) — caller-owned
    // cg_set_momentum_sum: where the fused pass leaves the sum of |mom|^2 (null: not summed)
    double *mom2_sum_out = nullptr, *mom2_partial = nullptr;
    size_t mom2_partial_bytes = 0;
    double *emig_rows = nullptr;
    unsigned *emig_rows_count = nullptr;
    i64 emig_rows_cap = 0;
    // device word of sticky error bits set by kernels (CG_ERR_*), read by cg_error_flags
    unsigned *err_flags = nullptr;
    i64 device_bytes = 0;
};

// kernels (cg_mesh_kernels.hip, cg_particles.hip)
int cgk_deposit_cic(cg_ctx *c, const double *pos, i64 n, double contribution);
int cgk_kspace(cg_ctx *c, int deconv_order, double C, int long_range, double E);
int cgk_gather_kick(cg_ctx *c, const double *pos, double *mom, i64 n, int diff_order,
                    double factor);
int cgk_drift(cg_ctx *c, double *pos, const double *mom, i64 n, double dt_over_mass);
int cgk_measure_mom(cg_ctx *c, const double *mom, i64 n, double *out, double *scratch);
int cgk_measure_mom_regions(cg_ctx *c, const double *mom, const unsigned *start,
                            const unsigned *count, double *out, double *scratch);
int cgk_cic_indices(cg_ctx *c, const double *pos, i64 n, int for_gather, i64 *idx);
int cgk_transpose_fourier(cg_ctx *c, const double *src, double *dst);
int cgk_sort(cg_ctx *c, const double *pos_in, const double *mom_in, const i64 *ids_in,
             double *pos_out, double *mom_out, i64 *ids_out, i64 n, unsigned *tile_offset_out,
             int drift, double dt_over_mass, int use_prepared);
int cgk_permute_rows(cg_ctx *c, const i64 *perm, i64 n, int ncols, const void *const *src,
                     void *const *dst, const int *row_bytes);
int cgk_shortrange_cells(cg_ctx *c, const double *pos, i64 n, i64 nt, double tile_extent,
                         unsigned *order, unsigned *offset, double *pos_sorted,
                         const signed char *rung, const signed char *rung_jumped,
                         int lowest_active, unsigned *nact, signed char *rj_sorted);
int cgk_shortrange_sparse(cg_ctx *c, const double *pos_r, const i64 *active, int K, double *dmom_r,
                          const double *pos_s, i64 n_s, const double *table,
                          double r2_index_scaling, double r2_max, double factor,
                          const double *factors, const signed char *rung_jumped);
int cgk_shortrange_sweep_cells(cg_ctx *c, const double *pos_r_sorted, const unsigned *order_r,
                               const unsigned *off_r, double *dmom_r, const double *pos_s_sorted,
                               const unsigned *off_s, i64 nt, const double *table,
                               double r2_index_scaling, double r2_max, double factor,
                               const double *factors, const signed char *rung,
                               const signed char *rung_jumped, int lowest_active,
                               const unsigned *nact_r, const signed char *rj_sorted_r,
                               i64 n_active_max);
int cgk_shortrange_dense(cg_ctx *c, const double *pos_r_sorted, const unsigned *order_r,
                         const unsigned *off_r, double *dmom_r, const double *pos_s_sorted,
                         const unsigned *off_s, i64 nt, const double *table,
                         double r2_index_scaling, double r2_max, double factor,
                         const double *factors, const signed char *rung,
                         const signed char *rung_jumped, int lowest_active,
                         const unsigned char *active_in, const unsigned char **take_out);
int cgk_shortrange_tiles_phase(cg_ctx *c, int phase, const double *pos, i64 n, i64 nt,
                               double tile_extent, const signed char *rung, int lowest_active,
                               unsigned *order, unsigned *offset, double *pos_sorted,
                               i64 sub_min, i64 sub_rows);
int cgk_shortrange_dense_join(cg_ctx *c);
int cgk_shortrange_dense_look(cg_ctx *c, const unsigned *off_cells, i64 nt);
int cgk_shortrange_tiles(cg_ctx *c, const double *pos, i64 n, i64 nt, double tile_extent,
                         const signed char *rung, int lowest_active, unsigned *order,
                         unsigned *offset, double *pos_sorted);
int cgk_dmom_active(cg_ctx *c, double *mom, double *dmom, const signed char *rung, i64 n,
                    int lowest_active, int op);
int cgk_dmom_to_acc(cg_ctx *c, double *dmom, const signed char *rung,
                    const signed char *rung_jumped, i64 n, int lowest_active, const double *conv,
                    int any_jumps);
int cgk_assign_rungs(cg_ctx *c, const double *dmom, signed char *rung, signed char *rung_jumped,
                     i64 n, double rung_factor, int N_rungs);
int cgk_flag_rung_jumps(cg_ctx *c, const double *dmom, const signed char *rung,
                        signed char *rung_jumped, i64 n, int lowest_active,
                        const double *integrals, double rf_up, double rf_down, int N_rungs,
                        int *any_out);
int cgk_apply_rung_jumps(cg_ctx *c, signed char *rung, signed char *rung_jumped, i64 n,
                         int N_rungs);
// (3 N_rungs - 1 entries of a rung table passed to a kernel by value)
#define CG_RUNG_TABLE_MAX 64
int cgk_substep_begin(cg_ctx *c, double *pos, const double *mom, double *dmom,
                      const signed char *rung, signed char *rung_jumped, i64 n, int do_drift,
                      double dt_over_mass, int do_flag, int lowest_active,
                      const double *integrals_1, double rf_up, double rf_down, int N_rungs,
                      int *any_out, long long *counts_after, int defer);
int cgk_substep_flush(cg_ctx *c);
int cgk_substep_partial(cg_ctx *c, i64 n, i64 per, int N_rungs, i64 *nwg_out);
int cgk_substep_populations(cg_ctx *c, i64 nwg, int N_rungs, long long *counts);
int cgk_substep_end(cg_ctx *c, double *mom, double *dmom, signed char *rung,
                    signed char *rung_jumped, i64 n, int do_apply, int lowest_active,
                    const double *conversion_factors, int N_rungs, long long *counts);
int cgk_rung_populations(cg_ctx *c, const signed char *rung, i64 n, int N_rungs, long long *counts);
bool cgk_fft_supported(i64 N);
int cgk_fft_dist_forward(cg_ctx *c, double *send_buf, i64 layer0, i64 nlayers);
int cgk_fft_dist_xsolve(cg_ctx *c, double *buf, int deconv_order, double C, int long_range,
                        double E);
int cgk_fft_dist_backward(cg_ctx *c, const double *recv_buf, i64 layer0, i64 nlayers);
int cgk_fft_dist_x(cg_ctx *c, double *buf, int inverse);
int cgk_copy_modes_pack(cg_ctx *from, i64 n_small, const int *rows_local, i64 n_rows, double *out);
int cgk_copy_modes_unpack(cg_ctx *onto, cg_ctx *from, i64 n_small, const int *rows_local,
                          i64 n_rows, const double *in, int deconv_order, int nlattice,
                          const double *shift, int op_add);
int cgk_emigrant_dest(cg_ctx *c, const double *pos, const double *mom, const i64 *idx,
                      const unsigned *count, i64 cap, double dtm, int *dest, int *send_counts);
int cgk_layers_write(cg_ctx *c, i64 layer0, i64 nlayers, const double *src, int add);
int cgk_owner_rank(cg_ctx *c, const double *pos, i64 n, int *owner);
// what: 0 forward, 1 backward, 2 forward + Poisson kernel + backward (fused)
int cgk_fluid_add(cg_ctx *c, const double *fluid, double factor, int op_add);
int cgk_nullify_nyquist(cg_ctx *c);
int cgk_fourier_operate(cg_ctx *onto, cg_ctx *from, int deconv_order, int nlattice,
                        const double *shift, int diff_dim, int op_add);
int cgk_copy_modes(cg_ctx *onto, cg_ctx *from, int deconv_order, int nlattice,
                   const double *shift, int op_add);
int cgk_deposit_general(cg_ctx *c, const double *pos, i64 n, double contribution, int order,
                        const CicGeom &geo);
int cgk_gather_scalar(cg_ctx *c, const double *pos, double *mom, i64 n, int dim, int order,
                      const CicGeom &geo, double factor);
int cgk_mesh_diff(cg_ctx *dst, cg_ctx *src, int dim, int diff_order);
int cgk_ewald_tabulate(cg_ctx *c, int gridsize, double *grid);
int cgk_pp_kick(cg_ctx *c, const double *pos_r, i64 n_r, double *dmom_r, const double *pos_s,
                i64 n_s, int same, const double *ewald_grid, int ewald_gridsize,
                double softening, int kernel, double factor, const double *factors,
                const signed char *rung, const signed char *rung_jumped, int lowest_active);
int cgk_owner_rank_drifted(cg_ctx *c, const double *pos, const double *mom, i64 n, double dtm,
                           int *owner);
int cgk_prepare_rebind(cg_ctx *c, const double *pos, const double *mom, i64 n_total,
                       const double *add_pos, const double *add_mom, i64 n_add);
int cgk_fluid_kick(cg_ctx *c, double *J, const double *rho, const double *P, int dim,
                   int diff_order, double minus_dt, double inv_c2);
int cgk_fft(cg_ctx *c, int what, int deconv_order, double C, int long_range, double E);
int cgk_deposit_cic_tiled(cg_ctx *c, const double *pos, i64 n, const unsigned *tile_offset,
                          const unsigned *count, double contribution, int accumulate);
// output side of the fused kick + drift + scatter (cg_gather_kick_drift_scatter)
struct FusedScatter {
    const unsigned *count_in;  // populations of the input regions (null: dense tile order)
    const unsigned *start_out;
    unsigned *count_out;
    double *pos_out, *mom_out;
    const i64 *ids_in;
    i64 *ids_out;
    const i64 *aux_in;
    i64 *aux_out;
    i64 out_capacity;
};
int cgk_gather_kick_tiled(cg_ctx *c, const double *pos, double *mom, i64 n,
                          const unsigned *tile_offset, int diff_order, double factor,
                          int prepare, double next_dtm, const FusedScatter *fs = nullptr);
// workgroup -> tile for the kernels that give a tile to a workgroup (cg_tiled_kernels.hip)
int cgk_tile_order(cg_ctx *c, const unsigned *start, const unsigned *count);
int cgk_predict_regions(cg_ctx *c, const unsigned *start_in, const unsigned *count_in,
                        unsigned *start_out);
int cgk_emigrant_rows_dest(cg_ctx *c, const double *rows, const unsigned *count, i64 cap,
                           int *dest, int *send_counts);
int cgk_region_insert(cg_ctx *c, const double *rows, i64 m, const unsigned *start,
                      unsigned *count, double *pos_out, double *mom_out, i64 *ids_out,
                      i64 *aux_out, i64 capacity);
