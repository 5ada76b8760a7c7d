/*
 * oracle/p3m_oracle.c — CPU restatement of the reference's P3M short-range
 * tile sweep.  TEST INFRASTRUCTURE (see pm_oracle.c header): never linked,
 * loaded or called by the product.
 *
 * Restates, for all particles on rung 0 (single-rung time stepping) and one
 * rank:
 *   Tiling.sort                    species.py:707-823   (particle -> tile index)
 *   particle_particle              interactions.py:1563-1791 (tile pairs, periodic offset
 *                                  :1615-1621, x_ji = xi - xj :1787-1789)
 *   gravity_pairwise_shortrange    gravity.py:263-354   (r2 cut, table lookup, Δmom +=/-=)
 * The table itself (gravity.py:373-424, interactions.py:1847-1914) is built
 * in oracle.py with numpy/scipy exactly as the reference does.
 *
 * Every unordered particle pair of neighbouring tiles is visited once and
 * applied to both partners with opposite signs, like the reference.  The
 * ORDER in which pairs are visited differs from the reference's (it walks
 * tile pairings, subtiles and rungs), so Δmom agrees to summation rounding,
 * not bit for bit; r2 and the table index of a pair are bit-identical (same
 * expression, same operation order).
 * Parity status: pinned against tests/golden/p3m_*.npz (dmom_short) by
 * tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef int64_t i64;

/* Tiling.sort, species.py:775-780 with location = 0 (one rank):
 * i = int((x - loc*(1+2eps))*((1/tile_extent)*(1-2eps))) */
static inline i64 tile_index_1d(double x, double inv_extent_guard) {
    return (i64)((x - 0.0) * inv_extent_guard);
}

void orc_shortrange_tiles(const double *pos, i64 N, i64 nt, double tile_extent, double eps,
                          i64 *tile_out) {
    double inv = (1 / tile_extent) * (1 - 2 * eps);
    for (i64 p = 0; p < N; p++) {
        i64 i = tile_index_1d(pos[3 * p + 0], inv);
        i64 j = tile_index_1d(pos[3 * p + 1], inv);
        i64 k = tile_index_1d(pos[3 * p + 2], inv);
        tile_out[p] = (i * nt + j) * nt + k;
    }
}

/*
 * The sweep.  table[tablesize], r2_index_scaling = (tablesize-1)/maxr2,
 * r2_max = range**2 (gravity.py:285-288); factor = G*m_r*m_s*ᔑdt_rungs[...][0]
 * (gravity.py:51-64, all particles on rung 0).
 */
int orc_shortrange_sweep(const double *pos, i64 N, double *dmom, double boxsize, i64 nt,
                         double tile_extent, double eps, const double *table,
                         double r2_index_scaling, double r2_max, double factor) {
    i64 ntiles = nt * nt * nt;
    i64 *tile = malloc(sizeof(i64) * (N > 0 ? N : 1));
    i64 *start = calloc(ntiles + 1, sizeof(i64));
    i64 *order = malloc(sizeof(i64) * (N > 0 ? N : 1));
    i64 *cursor = calloc(ntiles, sizeof(i64));
    if (!tile || !start || !order || !cursor) return 1;
    orc_shortrange_tiles(pos, N, nt, tile_extent, eps, tile);
    for (i64 p = 0; p < N; p++) {
        if (tile[p] < 0 || tile[p] >= ntiles) return 2; /* particle outside the tiling */
        start[tile[p] + 1]++;
    }
    for (i64 t = 0; t < ntiles; t++) start[t + 1] += start[t];
    for (i64 p = 0; p < N; p++) order[start[tile[p]] + cursor[tile[p]]++] = p;
    for (i64 tr = 0; tr < ntiles; tr++) {
        i64 ra = tr / (nt * nt), rb = (tr / nt) % nt, rc = tr % nt;
        for (int da = -1; da <= 1; da++) for (int db = -1; db <= 1; db++)
        for (int dc = -1; dc <= 1; dc++) {
            i64 sa = ra + da, sb = rb + db, sc = rc + dc;
            /* periodic offset from the tile separation, interactions.py:1615-1621:
             * supplier tile wrapped to the far end -> separation > L/2 -> +L */
            double off[3] = {0, 0, 0};
            if (sa < 0) { sa += nt; off[0] = boxsize; } else if (sa >= nt) { sa -= nt; off[0] = -boxsize; }
            if (sb < 0) { sb += nt; off[1] = boxsize; } else if (sb >= nt) { sb -= nt; off[1] = -boxsize; }
            if (sc < 0) { sc += nt; off[2] = boxsize; } else if (sc >= nt) { sc -= nt; off[2] = -boxsize; }
            i64 ts = (sa * nt + sb) * nt + sc;
            if (ts < tr) continue; /* every unordered tile pair once */
            for (i64 a = start[tr]; a < start[tr + 1]; a++) {
                i64 i = order[a];
                double xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
                for (i64 b = (ts == tr ? a + 1 : start[ts]); b < start[ts + 1]; b++) {
                    i64 j = order[b];
                    double x_ji = xi - pos[3 * j];         /* interactions.py:1787-1789 */
                    double y_ji = yi - pos[3 * j + 1];
                    double z_ji = zi - pos[3 * j + 2];
                    if (off[0] != 0 || off[1] != 0 || off[2] != 0) { /* gravity.py:299-302 */
                        x_ji += off[0];
                        y_ji += off[1];
                        z_ji += off[2];
                    }
                    double r2 = x_ji * x_ji + y_ji * y_ji + z_ji * z_ji; /* gravity.py:306 */
                    if (r2 > r2_max) continue;
                    i64 idx = (i64)(r2 * r2_index_scaling);              /* gravity.py:316 */
                    double total_factor = factor * table[idx];          /* gravity.py:321 */
                    double dx = x_ji * total_factor, dy = y_ji * total_factor,
                           dz = z_ji * total_factor;
                    dmom[3 * i] += dx; dmom[3 * i + 1] += dy; dmom[3 * i + 2] += dz;
                    dmom[3 * j] -= dx; dmom[3 * j + 1] -= dy; dmom[3 * j + 2] -= dz;
                }
            }
        }
    }
    free(tile); free(start); free(order); free(cursor);
    return 0;
}

/*
 * The sweep with adaptive rungs (particle_particle's rung logic,
 * interactions.py:1688-1761, and gravity_pairwise_shortrange, gravity.py:318-349):
 * particle i receives r*f*factors[rung_jumped[i]] if its rung is active
 * (rung[i] >= lowest_active_rung); a pair of two inactive particles is skipped.
 * factors[k] = G*m_r*m_s*ᔑdt_rungs[...][k], k < 3*N_rungs - 1 (gravity.py:51-64).
 */
int orc_shortrange_sweep_rungs(const double *pos, i64 N, double *dmom, double boxsize, i64 nt,
                               double tile_extent, double eps, const double *table,
                               double r2_index_scaling, double r2_max, const double *factors,
                               const signed char *rung, const signed char *rung_jumped,
                               int lowest_active_rung) {
    i64 ntiles = nt * nt * nt;
    i64 *tile = malloc(sizeof(i64) * (N > 0 ? N : 1));
    i64 *start = calloc(ntiles + 1, sizeof(i64));
    i64 *order = malloc(sizeof(i64) * (N > 0 ? N : 1));
    i64 *cursor = calloc(ntiles, sizeof(i64));
    if (!tile || !start || !order || !cursor) return 1;
    orc_shortrange_tiles(pos, N, nt, tile_extent, eps, tile);
    for (i64 p = 0; p < N; p++) {
        if (tile[p] < 0 || tile[p] >= ntiles) return 2;
        start[tile[p] + 1]++;
    }
    for (i64 t = 0; t < ntiles; t++) start[t + 1] += start[t];
    for (i64 p = 0; p < N; p++) order[start[tile[p]] + cursor[tile[p]]++] = p;
    for (i64 tr = 0; tr < ntiles; tr++) {
        i64 ra = tr / (nt * nt), rb = (tr / nt) % nt, rc = tr % nt;
        for (int da = -1; da <= 1; da++) for (int db = -1; db <= 1; db++)
        for (int dc = -1; dc <= 1; dc++) {
            i64 sa = ra + da, sb = rb + db, sc = rc + dc;
            double off[3] = {0, 0, 0};
            if (sa < 0) { sa += nt; off[0] = boxsize; } else if (sa >= nt) { sa -= nt; off[0] = -boxsize; }
            if (sb < 0) { sb += nt; off[1] = boxsize; } else if (sb >= nt) { sb -= nt; off[1] = -boxsize; }
            if (sc < 0) { sc += nt; off[2] = boxsize; } else if (sc >= nt) { sc -= nt; off[2] = -boxsize; }
            i64 ts = (sa * nt + sb) * nt + sc;
            if (ts < tr) continue;
            for (i64 a = start[tr]; a < start[tr + 1]; a++) {
                i64 i = order[a];
                int act_i = rung[i] >= lowest_active_rung;
                double xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
                for (i64 b = (ts == tr ? a + 1 : start[ts]); b < start[ts + 1]; b++) {
                    i64 j = order[b];
                    int act_j = rung[j] >= lowest_active_rung;
                    if (!act_i && !act_j) continue; /* interactions.py:1693-1707 */
                    double x_ji = xi - pos[3 * j], y_ji = yi - pos[3 * j + 1],
                           z_ji = zi - pos[3 * j + 2];
                    if (off[0] != 0 || off[1] != 0 || off[2] != 0) {
                        x_ji += off[0]; y_ji += off[1]; z_ji += off[2];
                    }
                    double r2 = x_ji * x_ji + y_ji * y_ji + z_ji * z_ji;
                    if (r2 > r2_max) continue;
                    double f = table[(i64)(r2 * r2_index_scaling)];
                    if (act_i) { /* gravity.py:320-327 */
                        double tf = factors[rung_jumped[i]] * f;
                        dmom[3 * i] += x_ji * tf; dmom[3 * i + 1] += y_ji * tf;
                        dmom[3 * i + 2] += z_ji * tf;
                    }
                    if (act_j) { /* gravity.py:332-349 */
                        double tf = factors[rung_jumped[j]] * f;
                        dmom[3 * j] -= x_ji * tf; dmom[3 * j + 1] -= y_ji * tf;
                        dmom[3 * j + 2] -= z_ji * tf;
                    }
                }
            }
        }
    }
    free(tile); free(start); free(order); free(cursor);
    return 0;
}

/*
 * The same sums for a SAMPLE of receivers, one-sided: receiver i = sample[s] meets every
 * particle of the 27 tiles around its own (particle_particle's tile neighbourhood,
 * interactions.py:1563-1791, periodic offset :1615-1621) with the pair arithmetic of
 * gravity_pairwise_shortrange (gravity.py:299-321) — O(M x 27 tile populations), so that a
 * check at full size (256^3 particles: 1.7e7 x 600 pairs for everything) takes seconds.
 * tile[N] from orc_shortrange_tiles; dmom_out[3 M].  Self pairs contribute x_ji * f = 0 * f.
 */
int orc_shortrange_sample(const double *pos, i64 N, const i64 *tile, const i64 *sample, i64 M,
                          double *dmom_out, double boxsize, i64 nt, const double *table,
                          double r2_index_scaling, double r2_max, double factor) {
    i64 ntiles = nt * nt * nt;
    i64 *start = calloc(ntiles + 1, sizeof(i64));
    i64 *order = malloc(sizeof(i64) * (N > 0 ? N : 1));
    i64 *cursor = calloc(ntiles, sizeof(i64));
    if (!start || !order || !cursor) return 1;
    for (i64 p = 0; p < N; p++) {
        if (tile[p] < 0 || tile[p] >= ntiles) return 2;
        start[tile[p] + 1]++;
    }
    for (i64 t = 0; t < ntiles; t++) start[t + 1] += start[t];
    for (i64 p = 0; p < N; p++) order[start[tile[p]] + cursor[tile[p]]++] = p;
    for (i64 s = 0; s < M; s++) {
        i64 i = sample[s], tr = tile[i];
        i64 ra = tr / (nt * nt), rb = (tr / nt) % nt, rc = tr % nt;
        double xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
        double ax = 0, ay = 0, az = 0;
        for (int da = -1; da <= 1; da++) for (int db = -1; db <= 1; db++)
        for (int dc = -1; dc <= 1; dc++) {
            i64 sa = ra + da, sb = rb + db, sc = rc + dc;
            double off[3] = {0, 0, 0};
            if (sa < 0) { sa += nt; off[0] = boxsize; } else if (sa >= nt) { sa -= nt; off[0] = -boxsize; }
            if (sb < 0) { sb += nt; off[1] = boxsize; } else if (sb >= nt) { sb -= nt; off[1] = -boxsize; }
            if (sc < 0) { sc += nt; off[2] = boxsize; } else if (sc >= nt) { sc -= nt; off[2] = -boxsize; }
            i64 ts = (sa * nt + sb) * nt + sc;
            for (i64 b = start[ts]; b < start[ts + 1]; b++) {
                i64 j = order[b];
                double x_ji = xi - pos[3 * j];         /* interactions.py:1787-1789 */
                double y_ji = yi - pos[3 * j + 1];
                double z_ji = zi - pos[3 * j + 2];
                if (off[0] != 0 || off[1] != 0 || off[2] != 0) { /* gravity.py:299-302 */
                    x_ji += off[0];
                    y_ji += off[1];
                    z_ji += off[2];
                }
                double r2 = x_ji * x_ji + y_ji * y_ji + z_ji * z_ji; /* gravity.py:306 */
                if (r2 > r2_max) continue;
                double total_factor = factor * table[(i64)(r2 * r2_index_scaling)];
                ax += x_ji * total_factor;
                ay += y_ji * total_factor;
                az += z_ji * total_factor;
            }
        }
        dmom_out[3 * s] = ax;
        dmom_out[3 * s + 1] = ay;
        dmom_out[3 * s + 2] = az;
    }
    free(start); free(order); free(cursor);
    return 0;
}
