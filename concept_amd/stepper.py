"""concept_amd.stepper — the callers of gravity() in the reference's time loop,
reduced to what fixes the order of kicks and drifts:

    main.kick_long()        main.py:1104-1144   half / full long-range kick
    main.kick_short()       main.py:1173-1262   nullify Δmom -> short-range gravity -> apply_Δmom
    main.driftkick_short()  main.py:1347-1624   (single rung: one full drift, :1395-1409)
    main.timeloop()         main.py:255-361     init: half long (+ half short) kick;
                                                step: drift, (short kick,) long kick

The time-step integrals ᔑdt[...] (main.get_time_step_integrals, integration.py:712-827)
are inputs: the cosmological background and the Δt limiters are outside the path, the
caller supplies plain numbers with the reference's keys."""
from . import interactions, lib
from .lib import ConceptGPUError


def _method(components):
    methods = {c.forces.get('gravity') for c in components}
    if len(methods) != 1 or None in methods:
        raise ConceptGPUError(f'components must share one gravity method, got {methods}')
    return methods.pop()


def kick_long(components, ᔑdt, printout=False):
    """main.kick_long (main.py:1104-1144)."""
    method = _method(components)
    interactions.gravity(method, components, components, ᔑdt, 'long-range', printout)


def kick_short(components, ᔑdt_rungs, printout=False):
    """main.kick_short (main.py:1173-1262) for particles that all sit on rung 0."""
    if _method(components) != 'p3m':
        return
    for c in components:
        c.nullify_Δ('mom')
    interactions.gravity('p3m', components, components, ᔑdt_rungs, 'short-range', printout)
    for c in components:
        c.apply_Δmom()


def drift(components, ᔑdt, a=1.0):
    """Component.drift for every component, leaving particle memory in tile order
    (the reference re-sorts tiles inside the next short-range kick, species.py:2598)."""
    for c in components:
        c.drift_sort(ᔑdt, a=a)


def timeloop(components, n_steps, integrals, rung_integrals=None, on_step=None):
    """Fixed-Δt KDK loop in the reference's order (main.py:255-361).  `integrals(kind)` and
    `rung_integrals(kind)` return the ᔑdt dicts for kind in {'init', 'full'} (half / full
    step), exactly the role of get_time_step_integrals()."""
    p3m = _method(components) == 'p3m'
    if p3m and rung_integrals is None:
        raise ConceptGPUError('P3M stepping needs the per-rung integrals (ᔑdt_rungs)')
    if on_step is None and not p3m and n_steps > 0:
        # nobody looks at the particles between a long-range kick and the drift after it: the
        # default PM configuration then takes both in one pass (same kicks and drifts, same
        # order: K½ D K D ... K; drift bit-exact, kick to summation order)
        plan = interactions.pm_streaming_plan(components)
        if plan is not None:
            return _timeloop_streaming(components, n_steps, integrals, plan)
    kick_long(components, integrals('init'))
    if p3m:
        kick_short(components, rung_integrals('init'))
    if on_step:
        on_step(0)
    for step in range(1, n_steps + 1):
        drift(components, integrals('full'))
        if p3m:
            kick_short(components, rung_integrals('full'))
        kick_long(components, integrals('full'))
        if on_step:
            on_step(step)


stream_replays = 0  # steps the streaming loop had to undo and take on the exact path


def _timeloop_streaming(components, n_steps, integrals, plan):
    """timeloop() for the default PM configuration with the kick of one step and the drift of
    the next fused.  Per pass: every component deposited from its tile regions
    (mesh.py:1512-1636), one Poisson solve (interactions.py:2092-2118), then per component
    cg_gather_kick_drift_scatter with the kick's ᔑdt['a**(-3*w_eff)', name] and the next
    drift's ᔑdt['a**(-2)'] (zero after the last kick)."""
    mesh = plan['mesh']
    fft_factor = float(plan['gridsize'])**(-3)
    p = components[0].params
    rps = [c.to_regions(mesh) for c in components]
    try:
        for step in range(n_steps + 1):
            ᔑdt_kick = integrals('init' if step == 0 else 'full')
            ᔑdt_drift = integrals('full') if step < n_steps else None
            for k, (c, rp) in enumerate(zip(components, rps)):
                rp.deposit(interactions._particle_contribution(
                    c, ᔑdt_kick, fft_factor, plan['gridsize'], p.boxsize), accumulate=k > 0)
            fold = mesh.fold_ghosts_start()
            mesh.poisson_solve(plan['deconv_order'], plan['C'], plan['long_range'], plan['E'],
                               fold_finish=fold, fill=True)
            before = [rp.snapshot() for rp in rps]
            for c, rp in zip(components, rps):
                order = c.potential_differentiations[plan['force']][plan['method']]
                Δt_over_mass = (ᔑdt_drift['a**(-2)']/c.mass) if ᔑdt_drift is not None else 0.0
                rp.kick_drift_sort(order, c.mass*(-ᔑdt_kick['a**(-3*w_eff)', c.name]),
                                   Δt_over_mass)
            # On several domains: ship the leavers and seat the arrivals now (into the new
            # buffer set), so that everything that can go wrong with this pass is known before
            # the next one starts.
            for rp in rps:
                rp.finish_exchange()
            # The regions of the new order were sized from the present populations; a (tile,
            # bucket) that grew beyond that in one step — by the drift or by arrivals from other
            # domains — or more leavers than the row buffer holds have dropped particles.  The
            # pass and the exchange wrote the other buffer set only: undo them and take the step
            # on the exact path (the potential is still on the mesh), then go on streaming.
            flags = mesh.error_flags()   # reads AND clears the sticky bits
            other = flags & ~(lib.CG_ERR_BUCKET_OVERFLOW | lib.CG_ERR_NOT_IN_TILE)
            if other:
                # e.g. CG_ERR_STALE_HISTOGRAM of a drift_sort on the replay path: that sort has
                # dropped particles, nothing to recover from here
                raise ConceptGPUError(f'streaming time loop: device error flags {other:#x} '
                                      f'in step {step}')
            overflow = bool(flags & (lib.CG_ERR_BUCKET_OVERFLOW | lib.CG_ERR_NOT_IN_TILE))
            if mesh.comm is not None:
                overflow = mesh.comm.any(overflow)
            if overflow:
                global stream_replays
                stream_replays += 1
                for i, c in enumerate(components):
                    rps[i].restore(before[i])
                    c.from_regions(rps[i])
                    interactions._kick_particles(mesh, c, plan['force'], plan['method'],
                                                 ᔑdt_kick, ('a**(-3*w_eff)', 'component'))
                    if ᔑdt_drift is not None:
                        c.drift_sort(ᔑdt_drift, mesh=mesh)
                    rps[i] = c.to_regions(mesh)
    except BaseException:
        # unwinding: hand the particles back without the collective part of from_regions (the
        # other domains may not be unwinding), then let the exception travel
        for c, rp in zip(components, rps):
            try:
                c.from_regions(rp, collective=False)
            except Exception:
                pass
        raise
    for c, rp in zip(components, rps):
        c.from_regions(rp)


# ---------------------------------------------------------------------------
# Adaptive rungs: the reference's kick_short / driftkick_short state machine
# ---------------------------------------------------------------------------
import numpy as np

from . import commons

PAIR_KEY = 'a**(-3*w_eff₀-3*w_eff₁-1)'


def static_integrals(components):
    """get_time_step_integrals() (main.py:998-1073) for a static background (a = 1,
    enable_Hubble = False): every integrand integrates to t_end - t_start."""
    keys = ['1', 'a**2', 'a**(-1)', 'a**(-2)', 'ȧ/a']
    for c in components:
        for k in ('a**(-3*w_eff)', 'a**(-3*(1+w_eff))', 'a**(-3*w_eff-1)', 'a**(3*w_eff-2)',
                  'a**(-3*w_eff)*Γ/H'):
            keys.append((k, c.name))
    for c0 in components:
        for c1 in components:
            keys.append((PAIR_KEY, c0.name, c1.name))

    def integrals(t_start, t_end):
        return {k: (t_end - t_start) for k in keys}
    return integrals


class RungStepper:
    """kick_long / kick_short / driftkick_short / initialize_rung_populations of main.py
    for particle components with a P3M short-range force and N_rungs > 1.
    `integrals(t_start, t_end)` plays get_time_step_integrals(); `t` is universals.t."""

    def __init__(self, components, integrals, t=0.0, fac_softening=None, Δt_jump_fac=0.95,
                 Δt_reltol=1e-9):
        self.components = list(components)
        self.integrals = integrals
        self.t = float(t)
        p = self.components[0].params
        self.N_rungs = p.N_rungs
        # main.py:2366, 2373, 2433 (Δt_rung_factor = 1)
        self.fac_softening = 0.025 if fac_softening is None else fac_softening
        self.Δt_jump_fac = Δt_jump_fac
        self.Δt_reltol = Δt_reltol
        self.ᔑdt_rungs = {}

    # -- helpers --------------------------------------------------------------
    def _store(self, ᔑdt_rung, index):
        for integrand, integral in ᔑdt_rung.items():
            arr = self.ᔑdt_rungs.get(integrand)
            if arr is None:
                arr = self.ᔑdt_rungs[integrand] = np.zeros(3*self.N_rungs - 1)
            arr[index] = integral

    def _clip(self, t, Δt, sync_time):
        return sync_time if t + self.Δt_reltol*Δt + 2*commons.machine_ϵ > sync_time else t

    def _gravity_short(self):
        interactions.gravity('p3m', self.components, self.components, self.ᔑdt_rungs,
                             'short-range', False)

    # -- main.kick_long (main.py:1104-1144) -----------------------------------
    def kick_long(self, Δt, sync_time, step_type):
        t_start = self.t
        t_end = self._clip(t_start + (Δt/2 if step_type == 'init' else Δt), Δt, sync_time)
        if t_start == t_end:
            return
        kick_long(self.components, self.integrals(t_start, t_end))

    # -- main.kick_short (main.py:1173-1262) ----------------------------------
    def kick_short(self, Δt, fake=False):
        comps = self.components
        for c in comps:
            c.lowest_active_rung = c.lowest_populated_rung
        highest_populated_rung = max(c.highest_populated_rung for c in comps)
        t_start = self.t
        for rung_index in range(highest_populated_rung + 1):
            t_end = t_start + Δt/2**(rung_index + 1)
            self._store(self.integrals(t_start, t_end), rung_index)
        for c in comps:
            c.nullify_Δ('mom')
        self._gravity_short()
        if fake:
            for c in comps:
                c.convert_Δmom_to_acc(self.ᔑdt_rungs)
            for c in comps:
                c.assign_rungs(Δt, self.fac_softening)
        else:
            for c in comps:
                c.apply_Δmom()
                c.convert_Δmom_to_acc(self.ᔑdt_rungs)

    # -- main.initialize_rung_populations (main.py:1639-1659) -----------------
    def initialize_rung_populations(self, Δt):
        if Δt == 0:
            raise ConceptGPUError('Cannot initialise rung populations with Δt = 0')
        for c in self.components:
            if c.use_rungs:
                c.rung_indices.zero_()
                c.set_rungs_N()
        self.kick_short(Δt, fake=True)

    # -- main.driftkick_short (main.py:1347-1603) ------------------------------
    def driftkick_short(self, Δt, sync_time):
        comps = self.components
        nr = self.N_rungs
        any_kicks = True
        index_start = 0
        for driftkick_index in range(2**(nr - 1)):
            if any_kicks:
                index_start = 2*driftkick_index
            for rung_index in range(nr):
                if (driftkick_index + 1) % 2**(nr - 1 - rung_index) == 0:
                    lowest_active_rung = rung_index
                    break
            any_kicks = False
            for c in comps:
                c.lowest_active_rung = max(lowest_active_rung, c.lowest_populated_rung)
                if c.highest_populated_rung >= c.lowest_active_rung:
                    any_kicks = True
            if not any_kicks:
                continue
            index_end = 2*driftkick_index + 2
            t_start = self._clip(self.t + Δt*(float(index_start)/2**nr), Δt, sync_time)
            t_end = self._clip(self.t + Δt*(float(index_end)/2**nr), Δt, sync_time)
            if t_end > t_start:
                ᔑdt = self.integrals(t_start, t_end)
                for c in comps:
                    c.drift_sort(ᔑdt)
                    c.lowest_active_rung = max(lowest_active_rung, c.lowest_populated_rung)
            highest_populated_rung = max(c.highest_populated_rung for c in comps)
            for rung_index in range(lowest_active_rung, highest_populated_rung + 1):
                i0 = (2**(nr - 1 - rung_index)
                      + (driftkick_index//2**(nr - 1 - rung_index))*2**(nr - rung_index))
                i1 = i0 + 2**(nr - rung_index)
                ts = self._clip(self.t + Δt*(float(i0)/2**nr), Δt, sync_time)
                te = self._clip(self.t + Δt*(float(i1)/2**nr), Δt, sync_time)
                self._store(self.integrals(ts, te), rung_index)
                # integral for jumping down a rung: only every second kick, else -1
                if rung_index > 0 and (
                        (driftkick_index + 1) - 2**(nr - 1 - rung_index)) % 2**(nr - rung_index) == 0:
                    te = self._clip(self.t + Δt*(float(i0 + 2**(nr - 1 - rung_index))/2**nr), Δt,
                                    sync_time)
                    self._store(self.integrals(ts, te), rung_index + nr)
                else:
                    for arr in self.ᔑdt_rungs.values():
                        arr[rung_index + nr] = -1
                # integral for jumping up a rung
                if rung_index < nr - 1:
                    te = self._clip(self.t + Δt*(float(i0 + 3*2**(nr - 2 - rung_index))/2**nr), Δt,
                                    sync_time)
                    self._store(self.integrals(ts, te), rung_index + 2*nr)
            integrals_1 = self.ᔑdt_rungs['1']
            if sum(integrals_1[lowest_active_rung:highest_populated_rung + 1]) == 0:
                continue
            any_rung_jumps = [c.flag_rung_jumps(Δt, self.Δt_jump_fac, self.fac_softening,
                                                self.ᔑdt_rungs) for c in comps]
            for c in comps:
                c.nullify_Δ('mom')
            self._gravity_short()
            for c, jumps in zip(comps, any_rung_jumps):
                c.apply_Δmom()
                c.convert_Δmom_to_acc(self.ᔑdt_rungs, jumps)
            for c, jumps in zip(comps, any_rung_jumps):
                if jumps:
                    c.apply_rung_jumps()

    # -- one base step of main.timeloop (main.py:335-361) ----------------------
    def base_step(self, Δt, sync_time=float('inf')):
        self.driftkick_short(Δt, sync_time)
        self.t = self._clip(self.t + 0.5*Δt, Δt, sync_time)
        self.kick_long(Δt, sync_time, 'full')
        self.t = self._clip(self.t + 0.5*Δt, Δt, sync_time)
