/*
 * concept_gpu.h — C ABI of libconcept_gpu.so, the MI355X (gfx950) PM / P3M
 * gravity hot path of CO*N*CEPT.
 *
 * The reference has no FFI for this path: its boundary is the Python-level
 * interaction registry (interactions.py:2646 register(), :2854 gravity()),
 * below which everything is Cython-compiled Python plus FFTW glue (fft.c).
 * This header is the boundary a maintainer would bind from src/gravity.py /
 * src/interactions.py (ctypes stub in INTEGRATION.md): plain pointers and
 * sizes, no torch types.  Every entry point names the reference function it
 * replaces (file:line under reference/src).
 *
 * Conventions
 *  - Every pointer marked DEV is a device (HBM) pointer on the context's
 *    GPU; HOST pointers are ordinary host memory.
 *  - Particle arrays are the reference's own layout (species.py:2010-2064):
 *    AoS double[3*N] "xyzxyz...", FP64.
 *  - All work is enqueued on the context's HIP stream (cg_set_stream) and is
 *    asynchronous with respect to the host unless stated otherwise.
 *  - Return value: 0 on success, non-zero on error with the message available
 *    from cg_last_error() (the reference's abort(), commons.py:1002-1031,
 *    becomes an error return; nothing is printed, nothing exits).
 */
#ifndef CONCEPT_GPU_H
#define CONCEPT_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CG_ABI_VERSION 2  /* 2: region tables, fused kick+drift+scatter, Fourier views, error flags */
/* Empty sets: wherever an entry point takes a particle count n (or, for the short-range sweeps,
 * cell offsets), the particle arrays may be NULL when the count is zero — an x-slab domain of a
 * clustered or partly empty box owns no particles for a while, and the reference's loops simply
 * run zero times there (communication.py:135-517 exchanges into and out of such domains). */

typedef struct cg_ctx cg_ctx;

/* Mesh + domain description.  Parameter names follow the reference's
 * (commons.py:2956 boxsize, :2958-3237 potential_options, :4411-4432 nghosts,
 * :3926 cell_centered; communication.py:692-741,1765-1836 domain layout). */
typedef struct cg_params {
    double  boxsize;
    int64_t gridsize;      /* global PM grid size N (cubic, even) */
    int32_t nghosts;       /* ghost layers of a domain grid; reference default 2 */
    int32_t cell_centered; /* 1 (reference default) */
    int32_t interp_order;  /* 2 = CIC: the order of the context's fused / tiled kernels; the
                              other orders (1 NGP, 3 TSC, 4 PCS) go through cg_deposit and
                              cg_gather_scalar, which take the order per call */
    int32_t device;        /* HIP device ordinal */
    int32_t nprocs;        /* number of domains (= ranks = GPUs) */
    int32_t rank;          /* this domain */
    int32_t subdiv[3];     /* cutout_domains(nprocs), communication.py:692 */
    int32_t reserved;
} cg_params;

/* which = selector for cg_fetch (debug / parity tests only) */
enum {
    CG_FETCH_MESH_REAL = 0,  /* the owned layers of the real-space mesh, reference x-slab
                                layout double[N/P][N][N+2] (mesh.py:1935-1942) */
    CG_FETCH_MESH_FOURIER = 1 /* the local Fourier slab in the reference's transposed
                                layout double[j_local][i][N+2] re/im interleaved
                                (fft.c:55-72, mesh.py:2716-2719) */
};

const char *cg_last_error(void);
int cg_abi_version(void);

/* Context: owns the potential mesh, the rocFFT plans and scratch.
 * Replaces get_fftw_slab()/fftw_setup() (mesh.py:3769-3866, fft.c:105-212)
 * and the named grid buffers (communication.py:1666 get_buffer). */
int cg_create(const cg_params *params, cg_ctx **out);
int cg_destroy(cg_ctx *ctx);
int cg_set_stream(cg_ctx *ctx, void *hip_stream);
int cg_synchronize(cg_ctx *ctx);
/* bytes of HBM owned by the context */
int64_t cg_device_bytes(const cg_ctx *ctx);
/* Sticky error bits set by kernels since the last call (synchronises the stream, clears them).
 * The reference aborts inside the loop that finds the inconsistency (commons.py:1002-1031); a
 * kernel cannot, so it records the condition here and keeps memory safe. */
#define CG_ERR_STALE_HISTOGRAM 1u /* cg_drift_sort: the prepared histogram did not match the
                                     particles (mom was changed after
                                     cg_gather_kick_tiled_prepare); particles were dropped */
int cg_error_flags(cg_ctx *ctx, uint32_t *flags_out /*HOST*/);

/* --- A1/A2: mass deposition ------------------------------------------------
 * cg_mesh_zero: get_buffer(..., nullify=True) (mesh.py:600-603).
 * cg_deposit_cic: interpolate_particles (mesh.py:1512-1636) with the inlined
 * CIC loop (mesh.py:5101-5155, 5319-5324) for `n` particles, each depositing
 * `contribution` (mesh.py:1550-1573, computed by the caller).  On a single
 * domain the ghost fold communicate_ghosts(grid,'+=') (mesh.py:609,
 * communication.py:563-660) is the periodic wrap and is fused here. */
int cg_mesh_zero(cg_ctx *ctx);
int cg_deposit_cic(cg_ctx *ctx, const double *pos /*DEV 3n*/, int64_t n, double contribution);
/* Same deposit for particles in tile order (cg_sort_particles on exactly this
 * `pos`): one workgroup per mesh tile accumulates, in LDS, its own particles
 * and the boundary buckets of its 7 lower neighbour tiles, then writes the tile
 * once with plain stores (no global atomics, no zero-fill pass).
 * accumulate = 0: the mesh is assigned (no prior cg_mesh_zero needed);
 * accumulate = 1: added onto the existing mesh (second and later suppliers of
 * interpolate_upstream, mesh.py:604-608). */
int cg_deposit_cic_tiled(cg_ctx *ctx, const double *pos /*DEV 3n*/, int64_t n,
                         const uint32_t *tile_offset /*DEV tile table*/, double contribution,
                         int accumulate);

/* --- A3..A8: Poisson solve, in place on the context's mesh -----------------
 * slab_decompose + fft('forward') + nullify_modes('nyquist') (mesh.py:665-672,
 * 2284-2411, 4012-4157, 3591-3622; fft.c:240-248), the Poisson/deconvolution
 * kernel (fourier_loop mesh.py:2615-2890 inlined at interactions.py:2092-2118),
 * fft('backward') + domain_decompose (interactions.py:2302-2307).
 *   deconv_order : deconv_order_global (interactions.py:2069-2080), 4 for CIC up+down
 *   C            : -boxsize**2*G_Newton/pi                (interactions.py:2105)
 *   long_range,E : potential 'gravity long-range', E = -(2*pi/boxsize*scale)**2
 *                  (interactions.py:2110-2113) */
int cg_poisson_solve(cg_ctx *ctx, int deconv_order, double C, int long_range, double E);
/* The two halves, for tests: density -> Fourier potential, and back. */
int cg_poisson_forward(cg_ctx *ctx, int deconv_order, double C, int long_range, double E,
                       int apply_kernel);
int cg_poisson_backward(cg_ctx *ctx);
/* cg_poisson_solve with HIP events (on the context's stream) around its five passes
 * (z forward, y forward, x fused, y backward, z backward); synchronises.  bench.py's
 * per-kernel roofline comes from here. */
int cg_poisson_solve_timed(cg_ctx *ctx, int deconv_order, double C, int long_range, double E,
                           double pass_ms[5]);
/* Only the k-space kernel (A5/A6) on a mesh that already holds the forward transform. */
int cg_poisson_kernel(cg_ctx *ctx, int deconv_order, double C, int long_range, double E);

/* --- A9/A10: force differentiation + interpolation + kick ------------------
 * For dim in 0..2: diff_domaingrid(order 2|4) (mesh.py:4874-5030) and
 * interpolate_domaingrid_to_particles (mesh.py:376-459) via
 * apply_particle_mesh_force (interactions.py:2359-2402):
 *   mom[3p+dim] += factor * sum_8 (w * dphi/dx_dim),  factor = -mass*dt_kick.
 * The finite difference is evaluated per particle from the potential mesh
 * (same expression, same order as the reference's force grid cell). */
int cg_gather_kick(cg_ctx *ctx, const double *pos /*DEV 3n*/, double *mom /*DEV 3n*/, int64_t n,
                   int diff_order, double factor);
/* Same, for particles in tile order: the potential tile (+ stencil halo) is
 * staged in LDS once per workgroup.  Particles that have left their tile since
 * the sort are still handled (straight from the mesh). */
int cg_gather_kick_tiled(cg_ctx *ctx, const double *pos /*DEV 3n*/, double *mom /*DEV 3n*/,
                         int64_t n, const uint32_t *tile_offset /*DEV tile table*/, int diff_order,
                         double factor);

/* cg_gather_kick_tiled that also prepares the tile histogram of the NEXT drift
 * (pos + mom_new*next_dt_over_mass): a following cg_drift_sort on the same arrays with the
 * same dt_over_mass then skips its first pass.  The reference fuses kicks and drifts the
 * same way where it can (driftkick_short, main.py:1347).  Contract: nothing else may change
 * pos/mom between the two calls (cg_drift and cg_sort_particles invalidate the
 * preparation, as do this context's other momentum writers; see cg_prepare_invalidate). */
int cg_gather_kick_tiled_prepare(cg_ctx *ctx, const double *pos /*DEV 3n*/,
                                 double *mom /*DEV 3n*/, int64_t n,
                                 const uint32_t *tile_offset /*DEV tile table*/, int diff_order,
                                 double factor, double next_dt_over_mass);
/* The kick, the drift that follows it and the tile sort of the drifted particles in ONE pass
 * (the reference fuses kicks and drifts where it can, driftkick_short, main.py:1347): a
 * workgroup per tile gathers the force for its particles, kicks, drifts and writes position
 * and kicked momentum straight to the particle's place in the next tile order — the momenta
 * are not written in place and re-read, the positions are read once: 96 B per particle + the
 * potential instead of 168 B.  The places exist before the new keys are known because the
 * next order is laid out in REGIONS WITH GAPS: (tile, bucket) k owns the slots
 * [start_out[k], start_out[k+1]), sized by cg_predict_regions from the present populations
 * (population + 25 % + 32), and holds count_out[k] particles afterwards; slots beyond are
 * never read.  A bucket that outgrows its region sets CG_ERR_BUCKET_OVERFLOW (cg_error_flags)
 * and drops particles: the caller then repeats the step on the exact path
 * (cg_gather_kick_tiled + cg_drift_sort) from the untouched input arrays.
 *   cg_region_capacity   upper bound of start_out[8*ntiles] for n particles (array sizes)
 *   cg_predict_regions   start_in / count_in: the present order (count_in NULL: dense tile
 *                        order, populations = differences of start_in) -> start_out[8*ntiles+1]
 *   cg_deposit_cic_regions   cg_deposit_cic_tiled for particles stored in such regions
 *   cg_gather_kick_drift_scatter   mom_out = mom_in + kick (A9/A10), pos_out = drift(pos_in,
 *                        mom_out) (A11), both stored in the regions start_out; count_out is
 *                        zeroed here and ends as the new populations.  A particle found
 *                        outside the tile it is stored under is not kicked and sets
 *                        CG_ERR_NOT_IN_TILE.  On x-slab domains see below. */
#define CG_ERR_BUCKET_OVERFLOW 2u
#define CG_ERR_ACTIVE_OVERFLOW 8u /* cg_shortrange_sweep_cells_active: more active receivers than
                                   * n_active_max; those beyond it were not swept */
#define CG_ERR_NOT_IN_TILE 4u /* cg_gather_kick_drift_scatter met a particle outside the tile it
                                 is stored under (positions changed since the order was made):
                                 not kicked; repeat the step on the exact path */
int64_t cg_region_capacity(const cg_ctx *ctx, int64_t n);
int cg_predict_regions(cg_ctx *ctx, const uint32_t *start_in /*DEV*/,
                       const uint32_t *count_in /*DEV or NULL*/, uint32_t *start_out /*DEV*/);
int cg_deposit_cic_regions(cg_ctx *ctx, const double *pos /*DEV*/, const uint32_t *start /*DEV*/,
                           const uint32_t *count /*DEV*/, double contribution, int accumulate);
int cg_gather_kick_drift_scatter(cg_ctx *ctx, const double *pos_in, const double *mom_in,
                                 const int64_t *ids_in /*nullable*/, const uint32_t *start_in,
                                 const uint32_t *count_in /*NULL: dense*/, double *pos_out,
                                 double *mom_out, int64_t *ids_out /*nullable*/,
                                 const uint32_t *start_out, uint32_t *count_out, int diff_order,
                                 double factor, double dt_over_mass,
                                 const int64_t *aux_in /*nullable*/, int64_t *aux_out,
                                 int64_t out_capacity /* rows of the output arrays: a region
                                 predicted beyond them counts as overflowed */);
/* Launch order of the tile kernels (deposit, gather-kick, the fused pass).  A workgroup's time
 * follows its tile's population; on a clustered box the plain walk (an eighth of the box per
 * XCD) leaves the XCDs unequal loads and lets a tile of a hundred batches start late.
 * cg_deposit_cic_tiled / _regions therefore list the heavy tiles (more than max(1536, 1.5 x
 * mean) particles; at most an eighth of the tiles) of the populations they are given by falling
 * population (in 61 classes); the tile kernels run those first, dealt out to the XCDs in turn,
 * and the walk behind them skips them; the gather-kick of the same tables takes the list over.
 * Results do not depend on it.  CONCEPT_GPU_TILE_ORDER_MIN=0 keeps the plain walk.
 * cg_tile_order_read copies the present list to the host (at most `capacity` tiles) and
 * reports its length; *n_heavy = -1 when no list is in use (switched off, or fewer than 4096
 * tiles). */
int cg_tile_order_read(cg_ctx *ctx, uint32_t *heavy_out /*HOST*/, int64_t capacity,
                       int64_t *n_heavy /*HOST*/);
/* (ids and aux: two 64-bit columns that travel with the particles — a Component's `ids` and
 * the row numbers its host() uses to restore the populated order) */

/* analysis.measure(component, 'v_rms') (analysis.py:3902-3910, what get_base_timestep_size asks
 * for after every kick, main.py:697-916) without a pass of its own: with sum_out set,
 * cg_gather_kick_drift_scatter also leaves the sum of |mom|^2 over the particles it kicked —
 * leavers of a slab included — in sum_out[0] (per-wavefront partials added in a fixed order:
 * reproducible).  NULL switches it off.  cg_measure_momentum[_regions] stays the stand-alone form. */
int cg_set_momentum_sum(cg_ctx *ctx, double *sum_out /*DEV 1, or NULL*/);

/* The same on x-slab domains: a particle whose drift takes it out of the slab has no place in
 * this domain's next order; the kernel appends it — kicked and drifted — to a caller-owned row
 * buffer (8 doubles: pos 3, mom 3, id bits, aux bits), exchange() (communication.py:135-517)
 * ships the rows, and the receiving domain gives each a place in its regions.  No holes to
 * close on the sending side: a leaver was never written there.
 *   cg_set_emigrant_rows   rows[8*cap], *count (DEV; zeroed by every fused launch); NULL = off;
 *                          more than cap leavers in one pass set CG_ERR_BUCKET_OVERFLOW
 *   cg_emigrant_rows_dest  owner domain of every row + rows bound for each domain (counts
 *                          zeroed here; *count read on the device)
 *   cg_region_insert       m received rows -> their (tile, bucket) regions (start / count of the
 *                          order they join); sets CG_ERR_BUCKET_OVERFLOW if one does not fit */
int cg_set_emigrant_rows(cg_ctx *ctx, double *rows /*DEV*/, uint32_t *count /*DEV 1*/,
                         int64_t cap);
int cg_emigrant_rows_dest(cg_ctx *ctx, const double *rows /*DEV*/, const uint32_t *count /*DEV*/,
                          int64_t cap, int32_t *dest /*DEV cap*/, int32_t *send_counts /*DEV P*/);
int cg_region_insert(cg_ctx *ctx, const double *rows /*DEV 8m*/, int64_t m,
                     const uint32_t *start /*DEV*/, uint32_t *count /*DEV*/, double *pos_out,
                     double *mom_out, int64_t *ids_out /*nullable*/,
                     int64_t *aux_out /*nullable*/, int64_t capacity /* rows of the arrays */);

/* Drop the prepared histogram.  Every entry point of this context that writes momenta
 * (cg_gather_kick*, cg_gather_scalar, cg_dmom_apply, cg_drift) does so itself; a caller that
 * changes mom by other means (another context, its own kernels) between the prepare and the
 * sort must call this.  A stale histogram that slips through is caught by the sort
 * (CG_ERR_STALE_HISTOGRAM). */
int cg_prepare_invalidate(cg_ctx *ctx);

/* --- A11: drift ------------------------------------------------------------
 * Component.drift (species.py:2179-2199): pos = mod(pos + mom*dt_over_mass, boxsize)
 * with the reference's mod (commons.py:5103-5135). */
int cg_drift(cg_ctx *ctx, double *pos /*DEV 3n*/, const double *mom /*DEV 3n*/, int64_t n,
             double dt_over_mass);

/* --- A18: what the time loop's step-size limiters measure ------------------
 * measure(component, 'v_rms') and measure(component, 'v_max') of a particle component
 * (analysis.py:3965-3972, 3902-3910; used by get_base_timestep_size, main.py:842-912):
 *   out[0] = sum over the 3n reals of mom^2,   out[1] = max over particles of |mom_i|^2
 * (the caller divides by N, a^2 and the mass; over domains it sums / maximises the ranks'
 * values, the reference's allreduce).  Fixed summation order: bit-reproducible.
 * scratch: DEV double[2048]. */
int cg_measure_momentum(cg_ctx *ctx, const double *mom /*DEV 3n*/, int64_t n,
                        double *out /*DEV 2*/, double *scratch /*DEV 2048*/);
/* the same over particles kept in tile regions with gaps (start / count as for
 * cg_deposit_cic_regions; count = NULL: dense tile order) */
int cg_measure_momentum_regions(cg_ctx *ctx, const double *mom /*DEV rows*/,
                                const uint32_t *start, const uint32_t *count,
                                double *out /*DEV 2*/, double *scratch /*DEV 2048*/);

/* --- particle memory order -------------------------------------------------
 * The reference reorders particle memory for locality (Component.tile_sort,
 * species.py:2598-2810).  cg_sort_particles bins particles by mesh tile and
 * permutes pos/mom (and the optional ids) into tile order.  Scratch buffers
 * are owned by the caller: pos_out/mom_out/ids_out must not alias the inputs. */
int cg_sort_particles(cg_ctx *ctx, const double *pos_in, const double *mom_in,
                      const int64_t *ids_in /*nullable*/, double *pos_out, double *mom_out,
                      int64_t *ids_out /*nullable*/, int64_t n,
                      uint32_t *tile_offset_out /*DEV 8*ntiles+1, see cg_tile_info*/);
/* Fused A11 + sort (single domain): pos_out/mom_out = tile order of the DRIFTED particles,
 * drift arithmetic identical to cg_drift (species.py:2179-2199).  The input arrays are
 * left undrifted (they are scratch afterwards): saves writing and re-reading the drifted
 * positions between the two steps. */
int cg_drift_sort(cg_ctx *ctx, const double *pos_in, const double *mom_in,
                  const int64_t *ids_in /*nullable*/, double *pos_out, double *mom_out,
                  int64_t *ids_out /*nullable*/, int64_t n, double dt_over_mass,
                  uint32_t *tile_offset_out /*DEV tile table*/);
/* info[0] = tile extent T in cells (cubic), info[1] = tiles per dimension nt,
 * info[2] = number of entries of a tile table (8*nt^3 + 1).  Tile
 * t = (ta*nt + tb)*nt + tc holds the particles whose lower CIC cell
 * (set_weights_CIC index - nghosts, wrapped) lies in
 * [ta*T, ta*T+T) x [tb*T, ...) x [tc*T, ...); inside a tile particles are grouped
 * in 8 buckets f = 4*fx + 2*fy + fz, f? = 1 when that lower cell is the tile's last
 * in that dimension (the CIC cloud then reaches the next tile).  Table entry
 * 8*t + f = index of the first particle of bucket f of tile t. */
int cg_tile_info(const cg_ctx *ctx, int64_t info[3]);

/* --- A13..A15: P3M short-range tile sweep -------------------------------------
 * Tiling.sort (species.py:707-823) for the 'gravity (tiles)' tiling of nt^3 tiles (init_tiling,
 * species.py:3943-3983: nt = int(boxsize/tilesize*(1+eps)), at least 4), particle_particle
 * (interactions.py:1563-1791) and gravity_pairwise_shortrange (gravity.py:263-354):
 *   dmom_r[i] += sum_j ((xi - xj) + periodic_offset) * factor * table[int(r2*scaling)]
 * over all supplier particles j of the tiles around i's own with r2 <= r2_max.
 * One-sided: to kick both components of a pair call the sweep twice with the roles swapped.
 *   table            get_shortrange_table (gravity.py:373-424), DEV double[tablesize]
 *   r2_index_scaling (tablesize - 1)/shortrange_table_maxr2      (gravity.py:288)
 *   r2_max           shortrange_range**2                          (gravity.py:286)
 *   factor           G_Newton*mass_r*mass_s*dt_rungs[...][0]      (gravity.py:51-64)
 * With adaptive rungs (interactions.py:1688-1761, gravity.py:318-349; the _rungs entry): a
 * receiver on an active rung (rung_r[i] >= lowest_active_rung) is kicked with
 * factors[rung_jumped_r[i]], factors[k] = G*m_r*m_s*dt_rungs[...][k], k < 3*N_rungs-1 (DEV);
 * receivers on inactive rungs are left alone.
 *
 * The list: a cell list at HALF-tile granularity (the reference prunes below the tile level
 * with subtiles, interactions.py:1141-1278, species.py:4031-4142).
 * cg_shortrange_cells bins the particles into (2 nt)^3 cells — tile index exactly as
 * Tiling.sort (species.py:775-780), then which half of the tile in each dimension — and
 * writes, in cell order (z fastest): order_out[n] = particle indices, pos_sorted_out[3n] =
 * their positions (the sweep stages supplier runs with plain coalesced loads), offset_out[
 * (2 nt)^3 + 1] = first entry of each cell.  tile_extent must be boxsize/nt (species.py:607-609)
 * to within 4 ulp (the sweeps take boxsize/nt themselves; any other extent is refused).
 * cg_shortrange_sweep_cells[_rungs]: x_ji, r2 and the table index bit-identical to the
 * reference's, a receiver only meeting supplier cells at most two away:
 * dmom_r[order_r[q]] += ... for every receiver row q.  The force range
 * must not exceed the tile extent (the reference requires tilesize >= range,
 * species.py:3943-3983).  Receivers and suppliers may be different particle sets (two
 * components, or a component extended by the neighbour domains' boundary particles): there is
 * no self-pair test because a particle paired with itself contributes x_ji * f = 0 * f.
 * Density-adaptive (csrc/cg_shortrange_dense.hip — the counterpart of the reference's
 * automatic subtile refinement, species.py:4031-4142, interactions.py:145-329, :1236-1251): the
 * receivers of tiles holding 64 particles or more are taken off the cells and swept from a list by
 * tile whose rows follow a Hilbert curve through 8^3 sub-cells — 16 consecutive receivers per
 * wavefront against supplier rows in groups of four, a group whose bounding box is out of the
 * 16 receivers' reach is skipped (2.1-2.6 pair tests per pair in range instead of 4.2-4.4) —
 * with the same pair arithmetic; nothing changes for the caller (the lists are built inside the
 * call, only when such tiles exist and hold enough of the pair work to pay for the lists; what
 * decides is counted when cg_shortrange_cells builds the list and read from pinned memory: a
 * sweep waits for that list to be complete, no more).  With rungs: where every rung is active,
 * as above; in a sub-step for the rungs >= lowest_active_rung the tiles that hold 64 and more
 * ACTIVE receivers (a second look, at the histogram of the active receivers' tiles: the upper
 * rungs live where the particles are dense).  Environment:
 * CONCEPT_GPU_SR_DENSE_MIN=<particles per tile> moves the threshold (0: no dense tiles' sweep). */
int cg_shortrange_cells(cg_ctx *ctx, const double *pos /*DEV 3n*/, int64_t n, int64_t nt,
                        double tile_extent, uint32_t *order_out /*DEV n*/,
                        uint32_t *offset_out /*DEV (2nt)^3+1*/,
                        double *pos_sorted_out /*DEV 3n*/);
int cg_shortrange_sweep_cells(cg_ctx *ctx, const double *pos_r_sorted /*DEV*/,
                              const uint32_t *order_r /*DEV*/, const uint32_t *offset_r /*DEV*/,
                              double *dmom_r /*DEV 3n_r, accumulated*/,
                              const double *pos_s_sorted /*DEV*/, const uint32_t *offset_s /*DEV*/,
                              int64_t nt, const double *table /*DEV*/, int64_t tablesize,
                              double r2_index_scaling, double r2_max, double factor);
int cg_shortrange_sweep_cells_rungs(cg_ctx *ctx, const double *pos_r_sorted,
                                    const uint32_t *order_r, const uint32_t *offset_r,
                                    double *dmom_r, const double *pos_s_sorted,
                                    const uint32_t *offset_s, int64_t nt,
                                    const double *table /*DEV*/, int64_t tablesize,
                                    double r2_index_scaling, double r2_max,
                                    const double *factors /*DEV 3*N_rungs-1*/,
                                    const int8_t *rung_r /*DEV*/,
                                    const int8_t *rung_jumped_r /*DEV*/, int lowest_active_rung);

/* The list for a SUB-STEP of the rung loop (driftkick_short, main.py:1347-1624: the rungs >=
 * lowest_active_rung are kicked, the others only supply): as cg_shortrange_cells, with the
 * particles on active rungs FIRST inside every cell — the reference's lists by tile and rung
 * (species.py tiles_rungs_N, interactions.py:1688-1761) in the sweep's layout.  nact_out[cell] =
 * how many of the cell's rows are active, rung_jumped_sorted_out[row] = rung_jumped[order[row]]
 * (what selects the row's factor; may be null, as may rung_jumped_sorted_r of the sweep: the
 * sweep by active receiver reads rung_jumped_r through order_r — one scattered byte per particle
 * less to write — while a sweep in blocks of a list without them takes it as a plain list).  cg_shortrange_sweep_cells_active is
 * cg_shortrange_sweep_cells_rungs for a receivers' list made this way WITH THE SAME rung array
 * and lowest active rung: its receiver groups are the first nact rows of their cells — no pass
 * over the tiles' rungs, no gathers through order_r in front of the pair loop.  The sums are
 * those of the _rungs entry up to the order of the additions.  (As a suppliers' list it is a
 * plain list: the order of the rows inside a cell does not matter there.)
 * n_active_max: -1, or an upper bound of the number of active receivers (the time loop knows
 * the rung populations) with which the caller asks for the sweep by active CELL — one wavefront
 * per cell that holds an active receiver, its 25 supplier columns read where they are — which is
 * the faster one while a few per cent of the particles are active. */
int cg_shortrange_cells_rungs(cg_ctx *ctx, const double *pos /*DEV 3n*/, int64_t n, int64_t nt,
                              double tile_extent, const int8_t *rung /*DEV n*/,
                              const int8_t *rung_jumped /*DEV n*/, int lowest_active_rung,
                              uint32_t *order_out /*DEV n*/, uint32_t *offset_out /*DEV (2nt)^3+1*/,
                              double *pos_sorted_out /*DEV 3n*/, uint32_t *nact_out /*DEV (2nt)^3*/,
                              int8_t *rung_jumped_sorted_out /*DEV n*/);
int cg_shortrange_sweep_cells_active(cg_ctx *ctx, const double *pos_r_sorted,
                                     const uint32_t *order_r, const uint32_t *offset_r,
                                     const uint32_t *nact_r /*DEV*/,
                                     const int8_t *rung_jumped_sorted_r /*DEV*/, double *dmom_r,
                                     const double *pos_s_sorted, const uint32_t *offset_s,
                                     int64_t nt, const double *table /*DEV*/, int64_t tablesize,
                                     double r2_index_scaling, double r2_max,
                                     const double *factors /*DEV 3*N_rungs-1*/,
                                     const int8_t *rung_r /*DEV*/,
                                     const int8_t *rung_jumped_r /*DEV*/, int lowest_active_rung,
                                     int64_t n_active_max);

/* The particles listed by TILE (Tiling.sort, species.py:775-780; z fastest) — the reference's
 * `tiles[tile]` lists, what the parity tests compare tile by tile: order_out[m],
 * offset_out[nt^3 + 1] and, unless null, pos_sorted_out[3m].  With `rung` only the m particles
 * on rungs >= lowest_active_rung are listed (the receivers of a sub-step), otherwise m = n. */
int cg_shortrange_tiles(cg_ctx *ctx, const double *pos /*DEV 3n*/, int64_t n, int64_t nt,
                        double tile_extent, const int8_t *rung /*DEV n or null*/,
                        int lowest_active_rung, uint32_t *order_out /*DEV n*/,
                        uint32_t *offset_out /*DEV nt^3+1*/,
                        double *pos_sorted_out /*DEV 3n or null*/);

/* What the sweeps do per call, for the measurement (bench.py's roofline of the P3M step): with
 * enable = 1 the counters are zeroed and the following cg_shortrange_sweep_cells[_rungs] calls
 * run counting instantiations of their kernels; enable = 0 waits for the stream, copies the
 * counters out and switches back.  out[0..2]: the half-tile cells sweep — pair tests executed
 * (a lane that holds a receiver against a supplier of its range), of them in range (r2 <=
 * r2_max), wavefront trips (64 lane slots each); out[3..5]: the same for the dense tiles'
 * sweep; out[6..7] unused. */
int cg_shortrange_stats(cg_ctx *ctx, int enable, uint64_t *out /*HOST 8, or null*/);

/* The same sums for k <= 8 receivers (rows active[0..k) of pos_r / dmom_r) against ALL n_s
 * suppliers, without a cell list: the sub-steps of driftkick_short (main.py:1347-1624) that kick
 * only the few particles of the highest rungs.  Nearest periodic image; the range must stay
 * below a quarter of the box (it does: >= 4 tiles of at least the range, species.py:3971).
 * factors / rung_jumped as in cg_shortrange_sweep_cells_rungs, or both null with `factor`. */
int cg_shortrange_sparse(cg_ctx *ctx, const double *pos_r /*DEV*/, const int64_t *active /*DEV k*/,
                         int k, double *dmom_r /*DEV, accumulated*/, const double *pos_s /*DEV*/,
                         int64_t n_s, const double *table /*DEV*/, int64_t tablesize,
                         double r2_index_scaling, double r2_max, double factor,
                         const double *factors /*DEV or null*/,
                         const int8_t *rung_jumped_r /*DEV or null*/);

/* --- A16: momentum buffers and adaptive rungs --------------------------------
 * rung / rung_jumped are the reference's `signed char` arrays (species.py:2040-2064);
 * a jumped index is rung + N_rungs (down) or rung + 2*N_rungs (up).  rung = NULL means
 * "all particles" for the first two.
 *   cg_dmom_nullify     Component.nullify_Δ('mom')        species.py:3717-3741
 *   cg_dmom_apply       Component.apply_Δmom()            species.py:2253-2266
 *   cg_dmom_to_acc      Component.convert_Δmom_to_acc()   species.py:2290-2325
 *                       conversion_factors[k] = a**(3 w_eff)/(mass*(eps + dt_rungs['a**2'][k])), DEV
 *   cg_assign_rungs     Component.assign_rungs()          species.py:2422-2445, rung_factor =
 *                       get_rung_factor(dt, fac_softening) (species.py:2376-2400)
 *   cg_flag_rung_jumps  Component.flag_rung_jumps()       species.py:2463-2513, integrals_1 =
 *                       dt_rungs['1'] (DEV), *any_out (DEV int32) != 0 iff a jump was flagged
 *   cg_apply_rung_jumps Component.apply_rung_jumps()      species.py:2526-2549
 *   cg_rung_populations Component.set_rungs_N()           species.py:2560-2587, counts[N_rungs]
 *                       (DEV int64): the particles on each rung of this rank */
int cg_dmom_nullify(cg_ctx *ctx, double *dmom, const int8_t *rung, int64_t n,
                    int lowest_active_rung);
int cg_dmom_apply(cg_ctx *ctx, double *mom, const double *dmom, const int8_t *rung, int64_t n,
                  int lowest_active_rung);
int cg_dmom_to_acc(cg_ctx *ctx, double *dmom, const int8_t *rung, const int8_t *rung_jumped,
                   int64_t n, int lowest_active_rung, const double *conversion_factors,
                   int any_rung_jumps);
int cg_assign_rungs(cg_ctx *ctx, const double *acc, int8_t *rung, int8_t *rung_jumped, int64_t n,
                    double rung_factor, int N_rungs);
int cg_flag_rung_jumps(cg_ctx *ctx, const double *acc, const int8_t *rung, int8_t *rung_jumped,
                       int64_t n, int lowest_active_rung, const double *integrals_1,
                       double rung_factor_up, double rung_factor_down, int N_rungs,
                       int32_t *any_out);
int cg_apply_rung_jumps(cg_ctx *ctx, int8_t *rung, int8_t *rung_jumped, int64_t n, int N_rungs);
int cg_rung_populations(cg_ctx *ctx, const int8_t *rung, int64_t n, int N_rungs, int64_t *counts);

/* The other columns of a Component follow a sort (species.py:955-996: Δmom, ids, the rung arrays
 * stay in step with pos and mom through every reordering): dst[k][q] = src[k][perm[q]] for ncols
 * columns of row_bytes[k] bytes per row in ONE pass.  src, dst, row_bytes: HOST arrays of ncols
 * entries (the pointers in them DEV); not in place. */
int cg_permute_rows(cg_ctx *ctx, const int64_t *perm /*DEV n*/, int64_t n, int ncols,
                    const void *const *src, void *const *dst, const int *row_bytes);

/* A sub-step of driftkick_short (main.py:1347-1624, one domain) in two passes over the particles:
 *   cg_substep_begin = [Component.drift, species.py:2179-2199, if do_drift] ->
 *                      [flag_rung_jumps -> nullify_Δ('mom') for the rungs >= lowest_active_rung,
 *                       if do_flag]
 *   cg_substep_end   = [apply_Δmom -> convert_Δmom_to_acc, if do_apply] -> apply_rung_jumps ->
 *                      set_rungs_N (counts[N_rungs], DEV int64)
 * per particle the calls above in that order (each of them touches only the particle's own
 * rows: the values are the same).  integrals_1 = dt_rungs['1'] and conversion_factors
 * (species.py:2311-2315) are HOST arrays of 3*N_rungs-1 doubles — they travel as kernel
 * arguments, nothing is uploaded and nothing waited for; *any_out (DEV int32) as in
 * cg_flag_rung_jumps.  convert_Δmom_to_acc indexes the factors by the jumped rung index, which
 * IS the rung index of a particle that is not flagged.
 * counts_after (with do_flag): the rung populations as they will be AFTER this sub-step's jumps —
 * a flagged particle counted on the rung it jumps to, which is all apply_rung_jumps does at the
 * sub-step's end: the time loop learns them while the sub-step's sweep is still running and
 * queues the next sub-step behind it (cg_substep_end's counts may then be null). */
int cg_substep_begin(cg_ctx *ctx, double *pos /*DEV 3n*/, const double *mom /*DEV 3n*/,
                     double *dmom /*DEV 3n*/, const int8_t *rung, int8_t *rung_jumped, int64_t n,
                     int do_drift, double dt_over_mass, int do_flag, int lowest_active_rung,
                     const double *integrals_1 /*HOST*/, double rung_factor_up,
                     double rung_factor_down, int N_rungs, int32_t *any_out /*DEV*/,
                     int64_t *counts_after /*DEV N_rungs, or null*/, int defer);
/* defer = 1: cg_substep_begin launches nothing; the cell list the sub-step's sweep asks for next
 * (cg_shortrange_cells[_rungs] on the same pos, n) runs the pass on every particle as it bins it
 * — the drifted positions are not read a second time.  Any other entry that touches particles
 * runs a pass still pending first, as does cg_substep_flush. */
int cg_substep_flush(cg_ctx *ctx);
int cg_substep_end(cg_ctx *ctx, double *mom, double *dmom, int8_t *rung, int8_t *rung_jumped,
                   int64_t n, int do_apply, int lowest_active_rung,
                   const double *conversion_factors /*HOST*/, int N_rungs,
                   int64_t *counts /*DEV N_rungs*/);

/* --- multi-GPU: x-slab domains ----------------------------------------------
 * One context per GPU with params.nprocs = P, rank = r, subdiv = (P,1,1):
 * domain r owns mesh layers x in [r*N/P, (r+1)*N/P) and the particles whose
 * lower CIC cell lies there; its mesh buffer carries G ghost layers on each
 * side; y and z stay periodic on every GPU.  This replaces the reference's
 * 3-D domains + x-slabs (communication.py:692-741, mesh.py:1935-1942) by slabs
 * for both, so the slab<->domain remaps (mesh.py:2138-2411) disappear and
 * every halo is a contiguous block of layers.  The library does the local
 * work; the exchanges themselves (RCCL through torch.distributed, or anything
 * else) are the caller's, on the buffers named here:
 *   ghost fold   communicate_ghosts(grid,'+=') : cg_layers_read(nxl, 1) -> next domain
 *                                                 -> cg_layers_write(0, 1, add=1)
 *   ghost fill   communicate_ghosts(grid,'=')  : cg_layers_read / cg_layers_write(add=0)
 *   FFT          fft.c:240-257 (FFTW-MPI's all-to-all): cg_dist_fft_forward ->
 *                all-to-all (equal blocks) -> cg_dist_fft_xsolve -> all-to-all back
 *                -> cg_dist_fft_backward
 *   exchange()   communication.py:135-517      : cg_owner_rank + caller-side moves
 * info = {x0, nxl, G, N, pad, doubles in one transpose buffer: P blocks of
 *         complex[nxl][N/P + 1][pad/2], the last row of each layer unused} */
int cg_local_info(const cg_ctx *ctx, int64_t info[6]);
/* doubles in one x layer of the local mesh buffer = the unit of cg_layers_read / _write (a
 * layer holds one unused row besides its N rows of `pad` doubles) */
int64_t cg_layer_doubles(const cg_ctx *ctx);
/* layer0 is relative to the first owned layer (-G .. nxl+G-1) */
int cg_layers_read(cg_ctx *ctx, int64_t layer0, int64_t nlayers, double *dst /*DEV*/);
int cg_layers_write(cg_ctx *ctx, int64_t layer0, int64_t nlayers, const double *src /*DEV*/,
                    int add);
int cg_dist_fft_forward(cg_ctx *ctx, double *send_buf /*DEV*/);
int cg_dist_fft_xsolve(cg_ctx *ctx, double *buf /*DEV*/, int deconv_order, double C,
                       int long_range, double E);
int cg_dist_fft_backward(cg_ctx *ctx, const double *recv_buf /*DEV*/);
/* The same two steps for the owned layers [layer0, layer0 + nlayers) only.  In both transpose
 * buffers the block of domain q holds the layers of the SENDING domain outermost, so the data of
 * a layer range is one contiguous piece per peer: the caller can exchange a range while the
 * library transforms the next one (fft.c:240-257 does the whole transpose at once). */
int cg_dist_fft_forward_layers(cg_ctx *ctx, double *send_buf /*DEV*/, int64_t layer0,
                               int64_t nlayers);
int cg_dist_fft_backward_layers(cg_ctx *ctx, const double *recv_buf /*DEV*/, int64_t layer0,
                                int64_t nlayers);
/* The general particle_mesh() on x-slab domains (SURVEY.md §8f rows 1, 1b, 3 over §8e): the
 * Fourier-space slab must persist between the forward and the inverse transform, because
 * fourier_operate / copy_modes / the Poisson kernel / the '+=' of several upstream slabs act
 * on it (interactions.py:2092-2307, mesh.py:654-711).  It lives in a caller-owned buffer of
 * cg_local_info()[5] doubles in the transposed layout complex[N][N/P + 1][pad/2] — rows kj of
 * this domain's block, like FFTW-MPI's transposed output (fft.c:55-72).
 *   cg_dist_bind_fourier  make `buf` the context's Fourier view: cg_fourier_nullify_nyquist,
 *                         cg_fourier_operate, cg_poisson_kernel, cg_copy_modes (equal sizes)
 *                         then work on it (single domain: buf = NULL restores the in-place view)
 *   cg_dist_fft_x         the x pass alone, in place on such a buffer: forward transform =
 *                         cg_dist_fft_forward -> all-to-all -> cg_dist_fft_x(0); inverse =
 *                         cg_dist_fft_x(1) -> all-to-all -> cg_dist_fft_backward
 *   cg_copy_modes_pack / _unpack   copy_modes between DIFFERENT grid sizes (mesh.py:1018-1326
 *                         with the sub-slab exchange of get_subslabs, mesh.py:1327-1468): row kj
 *                         of the small cube belongs to different domains in the two grids, so
 *                         the owner in `from` packs the small-cube part of its rows
 *                         (out[r][N_small][N_small/2] complex, rows_local = row numbers inside
 *                         its block), the caller ships them (all-to-all-v), and the owner in
 *                         `onto` applies factor, phase and '=' / '+=' as cg_copy_modes does
 *                         (rows_local = the rows' numbers inside ITS block; `from` supplies
 *                         the deconvolution tables and grid size). */
int cg_dist_bind_fourier(cg_ctx *ctx, double *buf /*DEV or NULL*/);
int cg_dist_fft_x(cg_ctx *ctx, double *buf /*DEV*/, int inverse);
int cg_copy_modes_pack(cg_ctx *from, int64_t n_small, const int32_t *rows_local /*DEV*/,
                       int64_t n_rows, double *out /*DEV*/);
int cg_copy_modes_unpack(cg_ctx *onto, cg_ctx *from, int64_t n_small,
                         const int32_t *rows_local /*DEV*/, int64_t n_rows,
                         const double *in /*DEV*/, int deconv_order, int nlattice,
                         const double *shift /*HOST 3 or NULL*/, int op_add);
/* exchange() without a host round trip per question: destination domain of every listed
 * emigrant (cg_set_emigrant_list) under the drift pos + mom*dt_over_mass and the number bound
 * for each domain (send_counts[P], zeroed here).  *count is read on the device. */
int cg_emigrant_dest(cg_ctx *ctx, const double *pos /*DEV*/, const double *mom /*DEV*/,
                     const int64_t *idx /*DEV cap*/, const uint32_t *count /*DEV 1*/, int64_t cap,
                     double dt_over_mass, int32_t *dest /*DEV cap*/,
                     int32_t *send_counts /*DEV P*/);
int cg_owner_rank(cg_ctx *ctx, const double *pos /*DEV 3n*/, int64_t n,
                  int32_t *owner_out /*DEV n*/);
/* The fused step of the x-slab path: cg_owner_rank_drifted gives the owner of every particle
 * AFTER the coming drift (same arithmetic as cg_drift_sort) so that the caller can run
 * exchange() (communication.py:135-517) on the undrifted rows first; cg_prepare_rebind then
 * adds the immigrants' tile keys to the histogram prepared by the previous
 * cg_gather_kick_tiled_prepare and binds it to the compacted arrays, and cg_drift_sort
 * drifts and sorts in one pass pair (no stand-alone drift, no histogram pass). */
int cg_owner_rank_drifted(cg_ctx *ctx, const double *pos /*DEV 3n*/, const double *mom /*DEV 3n*/,
                          int64_t n, double dt_over_mass, int32_t *owner_out /*DEV n*/);
/* Optional: cg_gather_kick_tiled_prepare also lists the particles that the prepared drift takes
 * out of this domain's slab (the candidates of exchange(), communication.py:135-517):
 * idx[0 .. min(*count, cap)) are their row numbers, *count the number found (> cap: the list is
 * incomplete, fall back to cg_owner_rank_drifted over all particles).  Caller-owned device
 * buffers; idx = count = NULL switches the list off. */
int cg_set_emigrant_list(cg_ctx *ctx, int64_t *idx /*DEV cap*/, uint32_t *count /*DEV 1*/,
                         int64_t cap);
int cg_prepare_rebind(cg_ctx *ctx, const double *pos /*DEV*/, const double *mom /*DEV*/,
                      int64_t n_total, const double *add_pos /*DEV 3 n_add*/,
                      const double *add_mom /*DEV 3 n_add*/, int64_t n_add);

/* --- general particle_mesh(): several suppliers / receivers, particles and fluids
 *     (interactions.py:1985-2402) ----------------
 * The fused entry points above cover the default configuration (particle
 * components only, all grid sizes equal, CIC, 'sc' lattice).  The general case
 * is assembled by the caller from one context per (grid size, role) — on x-slab
 * domains the real-space operations below act on the local layers (fluid grids are
 * the domain's own double[N/P][N][N]; ghost layers are folded / filled by the caller
 * with cg_layers_read / _write), the k-space ones on the bound Fourier view — the
 * reference's 'slab_global', 'slab_updownstream', 'slab_updownstream_subgroup'
 * buffers — with cg_poisson_forward(apply_kernel = 0) / cg_poisson_kernel /
 * cg_poisson_backward and the operations below.
 *
 * cg_fluid_add: add_fluid_to_grid (mesh.py:1685-1753) / combine_fluids
 *   (mesh.py:1657-1683): mesh (= | +=) fluid*factor for a fluid scalar grid
 *   double[N][N][N] of the context's grid size (the caller folds the quantity's
 *   time-step integral and fft_factor into `factor`).
 * cg_fourier_nullify_nyquist: nullify_modes(slab, 'nyquist') (mesh.py:3591-3622).
 * cg_fourier_operate: in place (from == onto, op_add = 0) it is fourier_operate
 *   (mesh.py:3327-3400); otherwise the equal-size branch of copy_modes
 *   (mesh.py:1038-1092): onto (= | +=) factor * rotated/differentiated modes of
 *   `from`, over fourier_loop's modes (mesh.py:2615-2890: all but the Nyquist
 *   planes; with op_add = 0 and from != onto those are zeroed, the reference's
 *   nullified target).  deconv_order 0..8; nlattice = len(lattice) (1, 2 or 4);
 *   shift = lattice.shift (HOST double[3], may be NULL); diff_dim -1 or 0..2.
 *   The caller applies the reference's early exits (nothing to do; a pure
 *   1/nlattice scaling still goes through this call).
 * cg_copy_modes: copy_modes (mesh.py:1018-1326) between meshes of any two grid sizes
 *   (SURVEY.md §8f row 1b): every mode of the smaller grid off its Nyquist planes goes
 *   to the same wave vector of the other grid, rotated by
 *   (pi/N_onto - pi/N_from)*(ki + kj + kk) (cell-centred grids of different spacing,
 *   mesh.py:1302) plus the lattice phase, scaled by the deconvolution factor evaluated
 *   with the grid size of `from`.  Modes of a larger `onto` beyond the cube, and its
 *   Nyquist planes, are not touched: with op_add = 0 zero it first (cg_mesh_zero — the
 *   reference's nullified get_fftw_slab, mesh.py:686-709).  Equal sizes forward to
 *   cg_fourier_operate.
 * cg_deposit: interpolate_particles (mesh.py:1512-1636) of any order 1..4 (NGP, CIC,
 *   TSC, PCS: set_weights_*, mesh.py:5305-5394) with an interlacing lattice shift
 *   (HOST double[3] in grid units, NULL = none; Lattice, mesh.py:77-182), ghost fold
 *   fused (periodic).  cg_gather_scalar: interpolate_domaingrid_to_particles
 *   (mesh.py:376-459): mom[dim] += factor * (mesh interpolated at the particle), the
 *   mesh holding one force component (after cg_mesh_diff or a Fourier-space
 *   differentiation).  cg_mesh_diff: diff_domaingrid (mesh.py:4874-5030) of `src`'s
 *   real-space mesh into `dst`'s: the symmetric orders 2, 4, 6, 8 and the one-sided
 *   ('forward') order 1.
 * cg_mesh_copy: slab_downstream_subgroup[...] = slab_downstream
 *   (interactions.py:2242-2245, 2276-2279).
 * cg_fluid_kick: the fluid branch of apply_particle_mesh_force
 *   (interactions.py:2388-2401) fused with diff_domaingrid (mesh.py:4874-5030):
 *   J_dim[cell] += minus_dt*(rho[cell] + inv_c2*P[cell]) * d(phi)/dx_dim, the
 *   difference of order 1, 2, 4, 6 or 8 on the context's real-space potential
 *   (diff_order 0: the mesh already holds the force component, Fourier-space
 *   differentiation). */
int cg_fluid_add(cg_ctx *ctx, const double *fluid /*DEV N^3*/, double factor, int op_add);
int cg_fourier_nullify_nyquist(cg_ctx *ctx);
int cg_fourier_operate(cg_ctx *onto, cg_ctx *from, int deconv_order, int nlattice,
                       const double *shift /*HOST 3 or NULL*/, int diff_dim, int op_add);
int cg_copy_modes(cg_ctx *onto, cg_ctx *from, int deconv_order, int nlattice,
                  const double *shift /*HOST 3 or NULL*/, int op_add);
int cg_deposit(cg_ctx *ctx, const double *pos /*DEV 3n*/, int64_t n, double contribution,
               int order, const double *shift /*HOST 3 or NULL*/);
int cg_gather_scalar(cg_ctx *ctx, const double *pos /*DEV 3n*/, double *mom /*DEV 3n*/,
                     int64_t n, int dim, int order, const double *shift /*HOST 3 or NULL*/,
                     double factor);
int cg_mesh_diff(cg_ctx *dst, cg_ctx *src, int dim, int diff_order);
int cg_mesh_copy(cg_ctx *dst, cg_ctx *src);
int cg_fluid_kick(cg_ctx *ctx, double *J_dim /*DEV N^3*/, const double *rho /*DEV N^3*/,
                  const double *P /*DEV N^3*/, int dim, int diff_order, double minus_dt,
                  double inv_c2);

/* --- direct summation, 'pp' and 'ppnonperiodic'  ---------
 * cg_ewald_tabulate: ewald.tabulate() (ewald.py:62-118, 226-231): the Ewald correction
 *   force of a unit box on a gridsize^3 x 3 grid over one octant (DEV double[g][g][g][3]).
 * cg_pp_kick: gravity_pairwise (gravity.py:121-206: nearest image, Ewald look-up
 *   ewald.py:146-197, softened 1/r^3 of interactions.py:1847-1914) when ewald_grid is
 *   given, gravity_pairwise_nonperiodic (gravity.py:491-560) when it is NULL.
 *   dmom_r[i] += factor * sum_j force_ij over all suppliers (one-sided; the reference
 *   visits a pair once and updates both).  same = 1: the first n_r suppliers ARE the
 *   receivers, in the same order (pair i, i skipped); n_s > n_r: the rest of the component, on
 *   other domains (domain_domain pairing of interactions.py:398-590 in one-sided form).  kernel: 0 none, 1 plummer, 2 spline.  With rungs
 *   (factors/rung/rung_jumped DEV, as cg_shortrange_sweep_cells_rungs): receivers below
 *   lowest_active are skipped, the factor is factors[rung_jumped[i]]. */
int cg_ewald_tabulate(cg_ctx *ctx, int gridsize, double *grid /*DEV 3 g^3*/);
int cg_pp_kick(cg_ctx *ctx, const double *pos_r /*DEV 3 n_r*/, int64_t n_r,
               double *dmom_r /*DEV 3 n_r*/, const double *pos_s /*DEV 3 n_s*/, int64_t n_s,
               int same, const double *ewald_grid /*DEV or NULL*/, int ewald_gridsize,
               double softening, int kernel, double factor, const double *factors /*DEV or NULL*/,
               const signed char *rung /*DEV or NULL*/, const signed char *rung_jumped,
               int lowest_active);

/* --- debug fetch (parity tests) -------------------------------------------- */
int cg_fetch(cg_ctx *ctx, int which, double *out /*HOST*/, int64_t n_doubles);
/* CIC cell indices exactly as set_weights_CIC returns them for the deposit
 * (mesh.py:5319-5324 under the offsets of mesh.py:1577-1606) -> int64[3n] DEV */
int cg_cic_indices(cg_ctx *ctx, const double *pos /*DEV 3n*/, int64_t n, int for_gather,
                   int64_t *idx_out /*DEV 3n*/);

#ifdef __cplusplus
}
#endif
#endif /* CONCEPT_GPU_H */
