"""oracle/rungs.py — CPU restatement of the reference's adaptive-rung time stepping
(TEST INFRASTRUCTURE; never imported by the product).

    Component.convert_Δmom_to_acc   species.py:2290-2325
    Component.get_rung              species.py:2341-2363
    Component.get_rung_factor       species.py:2376-2400
    Component.assign_rungs          species.py:2422-2445
    Component.flag_rung_jumps       species.py:2463-2513
    Component.apply_rung_jumps      species.py:2526-2549
    Component.nullify_Δ / apply_Δmom (only active rungs)  species.py:3717-3741, 2253-2266
    main.initialize_rung_populations main.py:1639-1659
    main.kick_short                  main.py:1173-1262
    main.driftkick_short             main.py:1347-1603
for ONE particle component on itself and one rank.  Per-particle functions are numpy,
the rung-aware pair sweep is orc_shortrange_sweep_rungs (p3m_oracle.c).
Pinned against tests/golden/rungs_*.npz, produced by running the reference's own main.py
functions (tests/golden/make_golden.py child_rungs).
"""
import ctypes

import numpy as np

from . import oracle

machine_eps = oracle.machine_eps
KEY_PAIR = 'pair'  # 'a**(-3*w_eff₀-3*w_eff₁-1)'
KEY_A2 = 'a**2'
KEY_1 = '1'


class Particles:
    """pos, mom, Δmom (N,3) float64; rung_indices, rung_indices_jumped int8."""

    def __init__(self, pos, mom, mass, softening, N_rungs):
        self.pos = np.ascontiguousarray(pos, dtype=np.float64).copy()
        self.mom = np.ascontiguousarray(mom, dtype=np.float64).copy()
        self.dmom = np.zeros_like(self.mom)
        self.N = self.pos.shape[0]
        self.mass = mass
        self.softening_length = softening
        self.N_rungs = N_rungs
        self.rung = np.zeros(self.N, dtype=np.int8)
        self.rung_jumped = np.zeros(self.N, dtype=np.int8)
        self.rungs_N = np.zeros(N_rungs, dtype=np.int64)
        self.rungs_N[0] = self.N
        self.lowest_active_rung = 0
        self.lowest_populated_rung = 0
        self.highest_populated_rung = 0

    # species.py:2560-2587
    def set_rungs_N(self):
        self.rungs_N = np.bincount(self.rung, minlength=self.N_rungs).astype(np.int64)
        pop = np.nonzero(self.rungs_N)[0]
        self.lowest_populated_rung = int(pop[0]) if len(pop) else self.N_rungs - 1
        self.highest_populated_rung = int(pop[-1]) if len(pop) else 0

    def active(self):
        return self.rung >= self.lowest_active_rung

    # species.py:3717-3741
    def nullify_dmom(self):
        self.dmom[self.active()] = 0

    # species.py:2253-2266
    def apply_dmom(self):
        a = self.active()
        self.mom[a] += self.dmom[a]

    # species.py:2290-2325 (w_eff = 0, a = 1)
    def convert_dmom_to_acc(self, dt_rungs, any_rung_jumps=False):
        conversion = 1.0/(self.mass*(machine_eps + dt_rungs[KEY_A2]))
        a = self.active()
        idx = (self.rung_jumped if any_rung_jumps else self.rung)[a]
        self.dmom[a] *= conversion[idx][:, None]

    # species.py:2376-2400
    def get_rung_factor(self, dt, fac_softening):
        return 0.5*np.log2(dt**2/(2*fac_softening*self.softening_length))

    # species.py:2341-2363, vectorised
    def get_rung(self, rung_factor):
        acc2 = self.dmom[:, 0]**2 + self.dmom[:, 1]**2 + self.dmom[:, 2]**2
        with np.errstate(divide='ignore'):
            f = rung_factor + 0.25*np.log2(acc2)
        out = np.where(f < 0, 0, np.where(f > self.N_rungs - 1, self.N_rungs - 1,
                                          1 + np.where(np.isfinite(f), f, 0).astype(np.int8)))
        return np.where(acc2 == 0, self.rung, out).astype(np.int8)

    # species.py:2422-2445
    def assign_rungs(self, dt, fac_softening):
        r = self.get_rung(self.get_rung_factor(dt, fac_softening))
        self.rung[:] = r
        self.rung_jumped[:] = r
        self.set_rungs_N()

    # species.py:2463-2513
    def flag_rung_jumps(self, dt, dt_jump_fac, fac_softening, dt_rungs):
        integrals = dt_rungs[KEY_1]
        nr = self.N_rungs
        ought_up = self.get_rung(self.get_rung_factor(dt*dt_jump_fac, fac_softening))
        ought_down = self.get_rung(self.get_rung_factor(dt/dt_jump_fac, fac_softening))
        r = self.rung.astype(np.int64)
        consider = (r >= self.lowest_active_rung) if self.lowest_active_rung > 0 \
            else np.ones(self.N, dtype=bool)
        consider &= integrals[r] != 0
        up = consider & (ought_up > r)
        down_allowed = integrals[r + nr] != -1
        down = consider & ~up & down_allowed & (ought_down < r)
        self.rung_jumped[up] = (r[up] + 2*nr).astype(np.int8)
        self.rung_jumped[down] = (r[down] + nr).astype(np.int8)
        return bool(up.any() or down.any())

    # species.py:2526-2549
    def apply_rung_jumps(self):
        nr = self.N_rungs
        j = self.rung_jumped.astype(np.int64)
        moved = j >= nr
        self.rung[moved] += (2*(j[moved] >= 2*nr) - 1).astype(np.int8)
        self.rung_jumped[:] = self.rung
        self.set_rungs_N()


def shortrange_kick_rungs(p, sr, G_Newton, dt_rungs):
    """gravity('p3m', [c], [c], ᔑdt_rungs, 'short-range') with rungs: accumulates into p.dmom."""
    L = oracle.lib()
    if not hasattr(L, '_p3m_rungs'):
        dp, i64, dbl = oracle._dp, ctypes.c_int64, ctypes.c_double
        bp = ctypes.POINTER(ctypes.c_int8)
        L.orc_shortrange_sweep_rungs.argtypes = [dp, i64, dp, dbl, i64, dbl, dbl, dp, dbl, dbl, dp,
                                                 bp, bp, ctypes.c_int]
        L.orc_shortrange_sweep_rungs.restype = ctypes.c_int
        L._p3m_rungs = True
    factors = np.ascontiguousarray(G_Newton*p.mass*p.mass*dt_rungs[KEY_PAIR])  # gravity.py:63
    rc = L.orc_shortrange_sweep_rungs(
        oracle._p(p.pos), p.N, oracle._p(p.dmom), sr['boxsize'], sr['nt'], sr['boxsize']/sr['nt'],
        machine_eps, oracle._p(sr['table']), (sr['tablesize'] - 1)/sr['maxr2'], sr['range']**2,
        oracle._p(factors), p.rung.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)),
        p.rung_jumped.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)), int(p.lowest_active_rung))
    if rc:
        raise RuntimeError(f'orc_shortrange_sweep_rungs failed ({rc})')


def new_dt_rungs(N_rungs):
    return {k: np.zeros(3*N_rungs - 1) for k in (KEY_1, KEY_A2, KEY_PAIR)}


def kick_short(p, dt, t, sr, G_Newton, dt_rungs, fac_softening, fake=False):
    """main.kick_short (main.py:1173-1262); integrals are t_end - t_start."""
    p.lowest_active_rung = p.lowest_populated_rung
    for rung_index in range(p.highest_populated_rung + 1):
        t_end = t + dt/2**(rung_index + 1)
        for arr in dt_rungs.values():
            arr[rung_index] = t_end - t
    p.nullify_dmom()
    shortrange_kick_rungs(p, sr, G_Newton, dt_rungs)
    if fake:
        p.convert_dmom_to_acc(dt_rungs)
        p.assign_rungs(dt, fac_softening)
    else:
        p.apply_dmom()
        p.convert_dmom_to_acc(dt_rungs)


def initialize_rung_populations(p, dt, t, sr, G_Newton, dt_rungs, fac_softening):
    """main.initialize_rung_populations (main.py:1639-1659)."""
    p.rung[:] = 0
    p.set_rungs_N()
    kick_short(p, dt, t, sr, G_Newton, dt_rungs, fac_softening, fake=True)


def driftkick_short(p, dt, t, sync_time, sr, G_Newton, dt_rungs, fac_softening, dt_jump_fac,
                    dt_reltol, drift):
    """main.driftkick_short (main.py:1347-1603) for one component with a short-range force.
    `drift(p, dt_am2)` performs Component.drift with ᔑdt['a**(-2)'] = dt_am2."""
    nr = p.N_rungs
    tol = dt_reltol*dt + 2*machine_eps

    def clip(tt):
        return sync_time if tt + tol > sync_time else tt
    any_kicks = True
    index_start = 0
    for driftkick_index in range(2**(nr - 1)):
        if any_kicks:
            index_start = 2*driftkick_index
        for rung_index in range(nr):
            if (driftkick_index + 1) % 2**(nr - 1 - rung_index) == 0:
                lowest_active_rung = rung_index
                break
        p.lowest_active_rung = max(lowest_active_rung, p.lowest_populated_rung)
        any_kicks = p.highest_populated_rung >= p.lowest_active_rung
        if not any_kicks:
            continue
        index_end = 2*driftkick_index + 2
        t_start = clip(t + dt*(float(index_start)/2**nr))
        t_end = clip(t + dt*(float(index_end)/2**nr))
        if t_end > t_start:
            drift(p, t_end - t_start)
            p.lowest_active_rung = max(lowest_active_rung, p.lowest_populated_rung)
        highest_populated_rung = p.highest_populated_rung
        for rung_index in range(lowest_active_rung, highest_populated_rung + 1):
            i0 = 2**(nr - 1 - rung_index) + (driftkick_index//2**(nr - 1 - rung_index))*2**(
                nr - rung_index)
            i1 = i0 + 2**(nr - rung_index)
            ts = clip(t + dt*(float(i0)/2**nr))
            te = clip(t + dt*(float(i1)/2**nr))
            for arr in dt_rungs.values():
                arr[rung_index] = te - ts
            if rung_index > 0 and ((driftkick_index + 1) - 2**(nr - 1 - rung_index)) % 2**(
                    nr - rung_index) == 0:
                te = clip(t + dt*(float(i0 + 2**(nr - 1 - rung_index))/2**nr))
                for arr in dt_rungs.values():
                    arr[rung_index + nr] = te - ts
            else:
                for arr in dt_rungs.values():
                    arr[rung_index + nr] = -1
            if rung_index < nr - 1:
                te = clip(t + dt*(float(i0 + 3*2**(nr - 2 - rung_index))/2**nr))
                for arr in dt_rungs.values():
                    arr[rung_index + 2*nr] = te - ts
        if dt_rungs[KEY_1][lowest_active_rung:highest_populated_rung + 1].sum() == 0:
            continue
        any_rung_jumps = p.flag_rung_jumps(dt, dt_jump_fac, fac_softening, dt_rungs)
        p.nullify_dmom()
        shortrange_kick_rungs(p, sr, G_Newton, dt_rungs)
        p.apply_dmom()
        p.convert_dmom_to_acc(dt_rungs, any_rung_jumps)
        if any_rung_jumps:
            p.apply_rung_jumps()
