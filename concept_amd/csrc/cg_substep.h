// cg_substep.h — the first pass of a rung sub-step per particle, shared by its own kernel
// (cg_rungs.hip: k_substep_begin) and by the cell list's counting pass (cg_shortrange.hip), which
// runs it on the particle it is about to bin when the time loop deferred it.
//   Component.drift        species.py:2179-2199
//   flag_rung_jumps        species.py:2463-2513
//   nullify_Δ('mom')       species.py:3717-3741
#pragma once
#include "cg_internal.h"
#include "cg_tiles.h"

struct RungTable {
    double v[CG_RUNG_TABLE_MAX];
};

struct SubstepBegin {
    double *pos;
    const double *mom;
    double *dmom;
    const signed char *rung;
    signed char *rung_jumped;
    i64 n;
    int do_drift, do_flag;
    double dtm, L;
    int lowest_active;
    RungTable integrals;   // dt_rungs['1'], 3 N_rungs - 1 entries
    double rf_up, rf_down;
    int N_rungs;
    int *any_out;
    // the rung populations AFTER this sub-step's jumps (a flagged particle counted on the rung it
    // will jump to: apply_rung_jumps at the end of the sub-step moves exactly the flagged ones) —
    // partial[N_rungs * workgroup + rung], summed into counts[rung] by k_substep_populations;
    // null = not counted.  The time loop knows the next sub-step's populations while this
    // one's sweep is still running.
    unsigned *partial;
};

// get_rung, species.py:2341-2363
__device__ __forceinline__ int cg_get_rung(const double *__restrict__ dmom, i64 p, int current,
                                           double rung_factor, int N_rungs) {
    double ax = dmom[3 * p], ay = dmom[3 * p + 1], az = dmom[3 * p + 2];
    double acc2 = ax * ax + ay * ay + az * az;
    if (acc2 == 0) return current;
    double f = rung_factor + 0.25 * log2(acc2);
    if (f < 0) return 0;
    if (f > N_rungs - 1) return N_rungs - 1;
    return 1 + (int)(signed char)f;
}

// particle p: drifted position in (x, y, z) (written back when it drifted); flags and nullifies
// the particle if it is on an active rung.  Returns the rung it will be on after the sub-step.
__device__ __forceinline__ int cg_substep_begin_particle(const SubstepBegin &B, i64 p, double &x,
                                                         double &y, double &z) {
    x = B.pos[3 * p], y = B.pos[3 * p + 1], z = B.pos[3 * p + 2];
    if (B.do_drift) {
        x = ref_mod(x + B.mom[3 * p] * B.dtm, B.L);
        y = ref_mod(y + B.mom[3 * p + 1] * B.dtm, B.L);
        z = ref_mod(z + B.mom[3 * p + 2] * B.dtm, B.L);
        B.pos[3 * p] = x, B.pos[3 * p + 1] = y, B.pos[3 * p + 2] = z;
    }
    if (!B.do_flag) return 0;
    const int r = B.rung[p];
    if (r < B.lowest_active) return r;  // (flag_rung_jumps and nullify_Δ: active rungs only)
    int r_after = r;
    // flag_rung_jumps, species.py:2476-2512
    if (B.integrals.v[r] != 0) {
        int ought = cg_get_rung(B.dmom, p, r, B.rf_up, B.N_rungs);
        if (ought > r) {
            B.rung_jumped[p] = (signed char)(r + 2 * B.N_rungs);
            *B.any_out = 1;
            r_after = r + 1;
        } else if (B.integrals.v[r + B.N_rungs] != -1) {
            ought = cg_get_rung(B.dmom, p, r, B.rf_down, B.N_rungs);
            if (ought < r) {
                B.rung_jumped[p] = (signed char)(r + B.N_rungs);
                *B.any_out = 1;
                r_after = r - 1;
            }
        }
    }
    // nullify_Δ('mom'), species.py:3717-3741
    B.dmom[3 * p] = 0;
    B.dmom[3 * p + 1] = 0;
    B.dmom[3 * p + 2] = 0;
    return r_after;
}

// the workgroup's particles per rung-after-the-sub-step: `r` of this lane's particle (any value
// >= N_rungs: none), counted by ballots into s_cnt[N_rungs <= 64] (zeroed by the caller, read
// after a barrier)
__device__ __forceinline__ void cg_count_rungs(int r, int N_rungs, unsigned *s_cnt) {
    for (int q = 0; q < N_rungs && q < 64; q++) {
        const unsigned c = (unsigned)__popcll(__ballot(r == q));
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt[q], c);
    }
}
