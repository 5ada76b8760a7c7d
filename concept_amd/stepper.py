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


force_replays = 0   # test hook: that many of the next streaming passes are replayed


def _streaming_pass(plan, components, rps, ᔑdt_kick, ᔑdt_drift, label='', drift_on_replay=True):
    """One pass of the streaming form over particles kept in tile regions: every
    component deposited from its regions (mesh.py:1512-1636), one Poisson solve
    (interactions.py:2092-2118), then per component cg_gather_kick_drift_scatter with the kick's
    ᔑdt['a**(-3*w_eff)', name] and the following drift's ᔑdt['a**(-2)'].  ᔑdt_drift None: a kick
    only (the particles keep their places); ᔑdt_kick None: a drift only (kick factor 0: the
    momenta pass through unchanged, bit for bit).  `rps` is updated in place.  Returns True if the
    pass overflowed and was taken again on the exact path — the entries of `rps` are then NEW
    objects (snapshots of the old ones no longer apply); with drift_on_replay=False the replay
    takes the kick only and leaves the drift to the caller (a pass whose drift is a guess)."""
    mesh = plan['mesh']
    fft_factor = float(plan['gridsize'])**(-3)
    p = components[0].params
    if ᔑdt_kick is not None:
        for k, (c, rp) in enumerate(zip(components, rps)):
            rp.deposit(interactions._particle_contribution(
                c, ᔑdt_kick, fft_factor, plan['gridsize'], p.boxsize), accumulate=k > 0)
        fold = mesh.fold_ghosts_start()
        mesh.poisson_solve(plan['deconv_order'], plan['C'], plan['long_range'], plan['E'],
                           fold_finish=fold, fill=True)
        plan['solved'] = True
    elif not plan.get('solved'):
        # a drift before any kick: the gather multiplies what the mesh holds by 0 — make sure
        # that is a number
        mesh.zero()
        plan['solved'] = True
    before = [rp.snapshot() for rp in rps]
    for c, rp in zip(components, rps):
        order = c.potential_differentiations[plan['force']][plan['method']]
        Δt_over_mass = (ᔑdt_drift['a**(-2)']/c.mass) if ᔑdt_drift is not None else 0.0
        factor = c.mass*(-ᔑdt_kick['a**(-3*w_eff)', c.name]) if ᔑdt_kick is not None else 0.0
        rp.kick_drift_sort(order, factor, Δt_over_mass)
    # On several domains: ship the leavers and seat the arrivals now (into the new buffer
    # set), so that everything that can go wrong with this pass is known before the next one
    # starts.
    for rp in rps:
        rp.finish_exchange()
    # The regions of the new order were sized from the present populations; a (tile, bucket)
    # that grew beyond that in one step — by the drift or by arrivals from other domains — or
    # more leavers than the row buffer holds have dropped particles.  The pass and the exchange
    # wrote the other buffer set only: undo them and take the step on the exact path (the
    # potential is still on the mesh), then go on streaming.
    flags = mesh.error_flags()   # reads AND clears the sticky bits
    other = flags & ~(lib.CG_ERR_BUCKET_OVERFLOW | lib.CG_ERR_NOT_IN_TILE)
    if other:
        # e.g. CG_ERR_STALE_HISTOGRAM of a drift_sort on the replay path: that sort has
        # dropped particles, nothing to recover from here
        raise ConceptGPUError(f'streaming time loop: device error flags {other:#x} {label}')
    overflow = bool(flags & (lib.CG_ERR_BUCKET_OVERFLOW | lib.CG_ERR_NOT_IN_TILE))
    global force_replays
    if force_replays > 0:   # (tests: treat this pass as overflowed)
        force_replays -= 1
        overflow = True
    if mesh.comm is not None:
        overflow = mesh.comm.any(overflow)
    if overflow:
        global stream_replays
        stream_replays += 1
        for i, c in enumerate(components):
            rps[i].restore(before[i])
            c.from_regions(rps[i])
            if ᔑdt_kick is not None:
                interactions._kick_particles(mesh, c, plan['force'], plan['method'],
                                             ᔑdt_kick, ('a**(-3*w_eff)', 'component'))
            if ᔑdt_drift is not None and drift_on_replay:
                c.drift_sort(ᔑdt_drift, mesh=mesh)
            rps[i] = c.to_regions(mesh)
    return overflow


def _timeloop_streaming(components, n_steps, integrals, plan):
    """timeloop() for the default PM configuration with the kick of one step and the drift of
    the next fused: K½ D K D ... K as n_steps + 1 passes of _streaming_pass."""
    mesh = plan['mesh']
    rps = [c.to_regions(mesh) for c in components]
    try:
        for step in range(n_steps + 1):
            ᔑdt_kick = integrals('init' if step == 0 else 'full')
            ᔑdt_drift = integrals('full') if step < n_steps else None
            _streaming_pass(plan, components, rps, ᔑdt_kick, ᔑdt_drift, f'in step {step}')
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


class _LazyIntegrals(dict):
    """{integrand: integral over (t_start, t_end)} whose values are worked out when they are
    asked for: a sub-step of the rung loop reads four of the run's eleven integrands ('1',
    'a**2', 'a**(-2)' and the pair integrand of its components), and there are three such
    dictionaries per populated rung and sub-step (get_time_step_integrals, main.py:998-1073,
    evaluates them all: the values are the same, the unread ones are not missed)."""

    def __init__(self, one, keys, t_start, t_end):
        super().__init__()
        self.one, self.all_keys, self.t = one, keys, (t_start, t_end)

    def __missing__(self, key):
        if key not in self.all_keys:
            raise KeyError(key)
        value = self[key] = self.one(key, *self.t)
        return value

    def __contains__(self, key):
        return key in self.all_keys

    def __iter__(self):
        return iter(self.all_keys)

    def __len__(self):
        return len(self.all_keys)

    def keys(self):
        return list(self.all_keys)

    def items(self):
        return [(k, self[k]) for k in self.all_keys]

    def values(self):
        return [self[k] for k in self.all_keys]

    def get(self, key, default=None):
        return self[key] if key in self.all_keys else default


class _RungIntegrals(dict):
    """ᔑdt_rungs: {integrand: array of 3 N_rungs - 1 integrals}.  Entries set from lazy step
    integrals are filled in when the integrand's array is read."""

    def __init__(self, size):
        super().__init__()
        self.size, self.pending = size, {}

    def set(self, source, index):
        """arr[index] = source[integrand] for every integrand of `source` (a number: for every
        integrand known so far)"""
        if isinstance(source, _LazyIntegrals):
            for key in source.all_keys:
                self.pending.setdefault(key, {})[index] = source
        elif isinstance(source, dict):
            for key, value in source.items():
                self.pending.setdefault(key, {})[index] = value
        else:
            for key in set(self.pending) | set(dict.keys(self)):
                self.pending.setdefault(key, {})[index] = source

    def __missing__(self, key):
        if key not in self.pending:
            raise KeyError(key)
        arr = self[key] = np.zeros(self.size)
        return arr

    def __getitem__(self, key):
        arr = dict.__getitem__(self, key) if dict.__contains__(self, key) else self.__missing__(key)
        todo = self.pending.get(key)
        if todo:
            for index, source in todo.items():
                arr[index] = source[key] if isinstance(source, dict) else source
            todo.clear()
        return arr

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self.pending

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return list(set(dict.keys(self)) | set(self.pending))

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]


class _CachedIntegrals:
    """get_time_step_integrals() behind a small cache: the rung loop asks for a sub-step's
    integrals while the GPU is still busy with the sub-step before it (prefetch), and again
    when it needs them."""

    def __init__(self, func):
        self.func = func
        self.cache = {}

    def __call__(self, t_start, t_end):
        out = self.cache.get((t_start, t_end))
        return out if out is not None else self.func(t_start, t_end)

    def prefetch(self, t_start, t_end):
        key = (t_start, t_end)
        if key not in self.cache:
            if len(self.cache) > 256:
                self.cache.clear()
            self.cache[key] = self.func(t_start, t_end)


class RungStepper:
    """kick_long / kick_short / driftkick_short / initialize_rung_populations of main.py
    for particle components with a P3M short-range force and N_rungs > 1.
    `integrals(t_start, t_end)` plays get_time_step_integrals(); `t` is universals.t."""

    # (False: the separate calls of the reference's loop — what several domains take, and what
    # the tests compare the two-pass form of a sub-step with)
    fuse_substeps = True

    def __init__(self, components, integrals, t=0.0, fac_softening=None, Δt_jump_fac=0.95,
                 Δt_reltol=1e-9):
        self.components = list(components)
        # the short-range machinery (rungs, sub-steps, drifts) is the particles' alone
        # (main.py's particle_components); fluids take part in kick_long()
        self.particles = [c for c in self.components if c.representation == 'particles']
        self.integrals = _CachedIntegrals(integrals)
        self.t = float(t)
        p = self.components[0].params
        self.N_rungs = p.N_rungs
        # main.py:2366, 2373, 2433 (Δt_rung_factor = 1)
        self.fac_softening = 0.025 if fac_softening is None else fac_softening
        self.Δt_jump_fac = Δt_jump_fac
        self.Δt_reltol = Δt_reltol
        self.ᔑdt_rungs = _RungIntegrals(3*self.N_rungs - 1)

    # -- helpers --------------------------------------------------------------
    def _store(self, ᔑdt_rung, index):
        self.ᔑdt_rungs.set(ᔑdt_rung, index)

    def _clip(self, t, Δt, sync_time):
        return sync_time if t + self.Δt_reltol*Δt + 2*commons.machine_ϵ > sync_time else t

    def _shortrange_interactions(self):
        """find_interactions(particle_components, 'short-range') (main.py:1216-1225,
        1395-1403)"""
        return interactions.find_interactions(self.particles, 'short-range')

    def _gravity_short(self):
        """every short-range interaction once (main.py:1249-1262, 1559-1576); returns the
        receivers"""
        receivers_all = []
        for force, method, receivers, suppliers in self._shortrange_interactions():
            getattr(interactions, force)(method, receivers, suppliers, self.ᔑdt_rungs,
                                         'short-range', False)
            receivers_all += [r for r in receivers if r not in receivers_all]
        return receivers_all

    # -- main.kick_long (main.py:1104-1144) -----------------------------------
    def kick_long(self, Δt, sync_time, step_type):
        t_start = self.t
        t_end = self._clip(t_start + (Δt/2 if step_type == 'init' else Δt), Δt, sync_time)
        if t_start == t_end:
            return
        ᔑdt = self.integrals(t_start, t_end)
        for c in self.components:
            # the sub-steps of driftkick_short drift without sorting (one domain): the mesh
            # kernels want tile order — once per base step, not once per sub-step
            if c.representation == 'particles' and not c.tiles_exact and c._store is not None:
                c.tile_sort()
        for force, method, receivers, suppliers in interactions.find_interactions(
                self.components, 'long-range'):
            getattr(interactions, force)(method, receivers, suppliers, ᔑdt, 'long-range', False)

    # -- main.kick_short (main.py:1173-1262) ----------------------------------
    def kick_short(self, Δt, fake=False):
        comps = self.particles
        if not self._shortrange_interactions():
            return
        for c in comps:
            c.lowest_active_rung = c.lowest_populated_rung
        highest_populated_rung = max(c.highest_populated_rung for c in comps)
        t_start = self.t
        for rung_index in range(highest_populated_rung + 1):
            t_end = t_start + Δt/2**(rung_index + 1)
            self._store(self.integrals(t_start, t_end), rung_index)
        for c in comps:
            c.nullify_Δ('mom')
        receivers_all = self._gravity_short()
        if fake:
            for c in receivers_all:
                c.convert_Δmom_to_acc(self.ᔑdt_rungs)
            for c in comps:
                c.assign_rungs(Δt, self.fac_softening)
        else:
            for c in receivers_all:
                c.apply_Δmom()
                c.convert_Δmom_to_acc(self.ᔑdt_rungs)

    # -- main.initialize_rung_populations (main.py:1639-1659) -----------------
    def initialize_rung_populations(self, Δt):
        if Δt == 0:
            raise ConceptGPUError('Cannot initialise rung populations with Δt = 0')
        for c in self.particles:
            if c.use_rungs:
                c.rung_indices.zero_()
                c.set_rungs_N()
        self.kick_short(Δt, fake=True)

    # -- main.driftkick_short (main.py:1347-1603) ------------------------------
    def driftkick_short(self, Δt, sync_time):
        comps = self.particles
        nr = self.N_rungs
        if not self._shortrange_interactions():
            # no short-range interactions at all: the particles drift in one go
            # (main.py:1395-1413)
            t_start = self.t
            t_end = self._clip(t_start + Δt, Δt, sync_time)
            if t_start == t_end:
                return
            ᔑdt = self.integrals(t_start, t_end)
            for c in comps:
                c.drift_sort(ᔑdt)
            return
        # One domain, rungs in use: a sub-step's per-particle calls run as two passes
        # (Component.substep_begin / substep_end) and the host waits once per sub-step
        # (substep_finish), after it has worked out the next sub-step's integrals.
        fused = self.fuse_substeps and all(c.nprocs == 1 and c.use_rungs for c in comps)
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
            ᔑdt_drift = self.integrals(t_start, t_end) if t_end > t_start else None
            if ᔑdt_drift is not None and not fused:
                for c in comps:
                    # (one domain: nothing to exchange, and the short-range cell list does not
                    # need the mesh-tile order — kick_long sorts once before the mesh kernels;
                    # several domains: the fused drift + exchange + sort)
                    if c.nprocs == 1:
                        c.drift(ᔑdt_drift)
                    else:
                        c.drift_sort(ᔑdt_drift)
                    c.lowest_active_rung = max(lowest_active_rung, c.lowest_populated_rung)
            highest_populated_rung = max(c.highest_populated_rung for c in comps)
            self._store_rung_integrals(driftkick_index, lowest_active_rung,
                                       highest_populated_rung, Δt, sync_time)
            integrals_1 = self.ᔑdt_rungs['1']
            kick = sum(integrals_1[lowest_active_rung:highest_populated_rung + 1]) != 0
            if fused:
                for c in comps:
                    # (a kick follows: the pass is left to the cell list of the component's
                    # short-range sweep, which bins the particles it has just drifted)
                    c.substep_begin(ᔑdt_drift, kick, Δt, self.Δt_jump_fac, self.fac_softening,
                                    self.ᔑdt_rungs, defer=kick)
                if not kick:
                    continue
                receivers_all = self._gravity_short()
                for c in comps:
                    c.substep_end(c in receivers_all, self.ᔑdt_rungs)
                # (the GPU is busy with the sweep: the next sub-step's integrals meanwhile)
                self._prefetch_integrals(driftkick_index, Δt, sync_time)
                for c in comps:
                    c.substep_finish()
                continue
            if not kick:
                continue
            any_rung_jumps = [c.flag_rung_jumps(Δt, self.Δt_jump_fac, self.fac_softening,
                                                self.ᔑdt_rungs) for c in comps]
            for c in comps:
                c.nullify_Δ('mom')
            receivers_all = self._gravity_short()
            for c, jumps in zip(comps, any_rung_jumps):
                if c in receivers_all:
                    c.apply_Δmom()
                    c.convert_Δmom_to_acc(self.ᔑdt_rungs, jumps)
            for c, jumps in zip(comps, any_rung_jumps):
                if jumps:
                    c.apply_rung_jumps()
        self._check_sweeps()

    def _check_sweeps(self):
        """once per base step: did a sweep by active receiver meet more of them than the rung
        populations said (cg_error_flags of the meshes that took such sweeps)?"""
        from . import shortrange
        for mesh in list(shortrange.by_receiver_meshes.values()):
            mesh.check_errors()
        shortrange.by_receiver_meshes.clear()

    def _rung_integral_times(self, driftkick_index, rung_index, Δt, sync_time):
        """(index into ᔑdt_rungs, t_start, t_end) of the integrals rung `rung_index` needs in
        sub-step `driftkick_index`, t_end None where the entry is -1 (main.py:1480-1552)"""
        nr = self.N_rungs
        i0 = (2**(nr - 1 - rung_index)
              + (driftkick_index//2**(nr - 1 - rung_index))*2**(nr - rung_index))
        i1 = i0 + 2**(nr - rung_index)
        ts = self._clip(self.t + Δt*(float(i0)/2**nr), Δt, sync_time)
        out = [(rung_index, ts, self._clip(self.t + Δt*(float(i1)/2**nr), Δt, sync_time))]
        # integral for jumping down a rung: only every second kick, else -1
        if rung_index > 0 and (
                (driftkick_index + 1) - 2**(nr - 1 - rung_index)) % 2**(nr - rung_index) == 0:
            out.append((rung_index + nr, ts,
                        self._clip(self.t + Δt*(float(i0 + 2**(nr - 1 - rung_index))/2**nr), Δt,
                                   sync_time)))
        else:
            out.append((rung_index + nr, ts, None))
        # integral for jumping up a rung
        if rung_index < nr - 1:
            out.append((rung_index + 2*nr, ts,
                        self._clip(self.t + Δt*(float(i0 + 3*2**(nr - 2 - rung_index))/2**nr), Δt,
                                   sync_time)))
        return out

    def _store_rung_integrals(self, driftkick_index, lowest_active_rung, highest_populated_rung,
                              Δt, sync_time):
        for rung_index in range(lowest_active_rung, highest_populated_rung + 1):
            for index, ts, te in self._rung_integral_times(driftkick_index, rung_index, Δt,
                                                           sync_time):
                if te is None:
                    self.ᔑdt_rungs.set(-1, index)
                else:
                    self._store(self.integrals(ts, te), index)

    def _prefetch_integrals(self, driftkick_index, Δt, sync_time):
        """the integrals of the sub-step that follows `driftkick_index`, worked out now (they
        land in the integrals' cache).  Which rungs are populated by then is not known yet: the
        next sub-step is taken to be the next one whose lowest active rung is no higher than
        the highest rung populated now, plus one (a wrong guess costs nothing but the work)."""
        nr = self.N_rungs
        if not hasattr(self.integrals, 'prefetch'):
            return
        highest = min(nr - 1, max(c.highest_populated_rung for c in self.particles) + 1)
        for nxt in range(driftkick_index + 1, 2**(nr - 1)):
            for rung_index in range(nr):
                if (nxt + 1) % 2**(nr - 1 - rung_index) == 0:
                    lowest_active_rung = rung_index
                    break
            if lowest_active_rung <= highest:
                break
        else:
            return
        # (the drift runs from the end of this sub-step to the end of that one)
        t_start = self._clip(self.t + Δt*(float(2*driftkick_index + 2)/2**nr), Δt, sync_time)
        t_end = self._clip(self.t + Δt*(float(2*nxt + 2)/2**nr), Δt, sync_time)
        if t_end > t_start:
            self.integrals.prefetch(t_start, t_end)
        for rung_index in range(lowest_active_rung, highest + 1):
            for index, ts, te in self._rung_integral_times(nxt, rung_index, Δt, sync_time):
                if te is not None:
                    self.integrals.prefetch(ts, te)

    # -- one base step of main.timeloop (main.py:335-361) ----------------------
    def base_step(self, Δt, sync_time=float('inf')):
        self.driftkick_short(Δt, sync_time)
        self.t = self._clip(self.t + 0.5*Δt, Δt, sync_time)
        self.kick_long(Δt, sync_time, 'full')
        self.t = self._clip(self.t + 0.5*Δt, Δt, sync_time)


# ---------------------------------------------------------------------------
# main.timeloop() with the cosmic clock: base time step control, synchronisations, dumps
# ---------------------------------------------------------------------------
import collections
import math

from .integration import Cosmology

DumpTime = collections.namedtuple('DumpTime', ('time_param', 't', 'a'))
ထ = math.inf


def measure(component, quantity, a, regions=None):
    """analysis.measure(component, 'v_rms' | 'v_max') for particle components
    (analysis.py:3902-3910, 3965-3972); collective over the domains.  regions: the component's
    particles in streaming form (distributed.RegionParticles)."""
    if component.representation != 'particles':
        raise ConceptGPUError('measure(): particle components only (the fluid solvers are '
                              'outside this path)')
    if regions is not None:
        mom2_sum, mom2_max = regions.measure_momentum(want_max=quantity == 'v_max')
    else:
        mom2_sum, mom2_max = component._store.mesh.measure_momentum(component.mom)
    if component.comm is not None and component.nprocs > 1:
        both = component.comm.all_gather_floats([mom2_sum, mom2_max])
        mom2_sum, mom2_max = float(both[:, 0].sum()), float(both[:, 1].max())
    w_eff = component.w_eff(a=a)
    if quantity == 'v_rms':
        return math.sqrt(mom2_sum/component.N)/(a**(2 - 3*w_eff)*component.mass)
    if quantity == 'v_max':
        return math.sqrt(mom2_max)/(a**(2 - 3*w_eff)*component.mass)
    raise ConceptGPUError(f'measure(): quantity "{quantity}" not implemented')


class Timeloop(RungStepper):
    """main.timeloop() (main.py:102-471) for particle components: from the initial time to the
    last dump time with the reference's base-step control — get_base_timestep_size
    (main.py:697-916: dynamical time, Δa limits, Hubble time, PM and P³M limiters),
    update_base_timestep_size (main.py:922-982), synchronisation of kicks and drifts at dumps and
    at changes of Δt, the init / full step types — on top of kick_long / kick_short /
    driftkick_short (RungStepper) with the time-step integrals of integration.Cosmology.

    on_dump(loop, dump_time) is the reference's dump() (snapshot output is outside the path).

    Streaming (streaming = None: whenever the configuration allows; False: never).  The default
    PM configuration (interactions.pm_streaming_plan) with no short-range force takes every long
    kick together with the drift that follows it as ONE pass over the particles (DESIGN.md
    §4a).  The drift's interval is known when the kick is asked for: inside a segment K½ D K D
    ... K it depends on t, Δt and the dump times only — what get_base_timestep_size measures
    after kick n decides the LENGTH OF KICK n + 1 and whether a synchronisation follows, never
    drift n + 1 — and the last kick of a segment is followed by no drift.  The one exception is
    the drift after an init kick, which a reduction of Δt found right after that kick cuts to
    half a step (main.py:316-321): there the pass is speculative — the drift the loop then asks
    for is compared with the one taken, and a pass whose guess was wrong is undone (it wrote
    the other buffer set only) and retaken as a kick and a drift of their own."""
    # main.py:2336-2378
    Δt_initial_fac = 0.95
    Δt_reduce_fac = 0.94
    Δt_increase_fac = 0.96
    Δt_increase_min_factor = 1.01
    Δt_ratio_warn = 0.7
    Δt_ratio_abort = 0.01
    Δt_period = 1*8

    def __init__(self, components, cosmology=None, on_dump=None, on_step=None, streaming=None,
                 fluid_drift=None, fluid_limiter=None):
        """fluid_drift(component, ᔑdt, a_end): the hook for fluid components.  Their solvers
        (fluid.py: MacCormack / Kurganov-Tadmor on ϱ, J) are outside this path, but the loop
        around them is not: with the hook given, fluid components ride along — they take
        part in every kick_long() through gravity() (deposit of ϱ, fluid kick of J,
        interactions.py:1985-2335) and the hook is called where main.py calls
        Component.drift() for them (drift_fluids, main.py:1279-1299: once per full base step,
        over the whole step, before the particles' sub-steps) with the step's integrals and the
        scale factor at its end; it advances component.ϱ / component.J in place.
        fluid_limiter(component, a) -> Δt: the fluid's own limit on the base step (the
        reference's Courant condition, main.py:773-836), optional."""
        p = components[0].params
        self.cosmo = cosmology or Cosmology(p)
        self.cosmo.init_time()
        self.fluid_drift, self.fluid_limiter = fluid_drift, fluid_limiter
        for c in components:
            if c.representation != 'particles' and fluid_drift is None:
                raise ConceptGPUError(
                    f'Timeloop: {c.name} is a fluid component; fluids take part in gravity() '
                    'but their own evolution (fluid.py) is outside this path: pass '
                    'fluid_drift=callback(component, ᔑdt, a_end) to advance ϱ and J')
        super().__init__(components, self._integrals, t=self.cosmo.t,
                         fac_softening=0.025*p.Δt_rung_factor)
        self.params = p
        self.on_dump, self.on_step = on_dump, on_step
        # main.py:2386-2426
        bg, nl = p.Δt_base_background_factor, p.Δt_base_nonlinear_factor
        self.fac_dynamical = 0.056*bg
        self.fac_hubble = 0.031*bg
        self.fac_pm = 0.13*nl
        self.fac_p3m = 0.14*nl
        self.initial_fac_times = set()
        self.keys = None
        self.static_timestepping_func = None
        if callable(p.static_timestepping):
            def func(a):   # main.py:646-660
                t = self.cosmo.t if a == self.cosmo.a else self.cosmo.cosmic_time(a)
                a_next = a + p.static_timestepping(a)
                return self.cosmo.cosmic_time(a_next) - t if a_next <= 1 else ထ
            self.static_timestepping_func = func
        elif p.static_timestepping is not None:
            raise ConceptGPUError('static_timestepping: only a callable Δa(a) is supported here '
                                  '(recording to / replaying from a file is outside the path)')
        self.time_step = 0
        self.Δt = 0.0
        self.history = []   # (time_step, t, a, Δt) at the beginning of every time step
        self.streaming = streaming
        self._plan = self._rps = self._spec = self._next_drift = None
        self.stream_passes = self.stream_wrong_guesses = 0
        if on_dump is None and p.output_dirs.get('snapshot') and (
                p.snapshot_times['a'] or p.snapshot_times['t']):
            # the parameter file asks for snapshots (output_dirs, output_times['snapshot'],
            # snapshot_type, gadget_snapshot_params: main.dump, main.py:1660-1760)
            if p.snapshot_type != 'gadget':
                import warnings
                warnings.warn(f"snapshot_type = '{p.snapshot_type}': only 'gadget' snapshots are "
                              'written here (the reference\'s own format is HDF5, which needs h5py); '
                              'no snapshots will be dumped')
            else:
                gsp = p.gadget_snapshot_params
                self.on_dump = self.snapshot_dumper(
                    p.output_dirs['snapshot'], p.output_bases.get('snapshot', 'snapshot'),
                    only_snapshot_times=True, snapformat=gsp['snapformat'],
                    dataformat=gsp['dataformat'], header=gsp['header'],
                    particles_per_file=gsp['particles per file'], units=gsp['units'])

    # universals.t / universals.a live in the Cosmology object
    t = property(lambda self: self.cosmo.t, lambda self, v: setattr(self.cosmo, 't', float(v)))

    def _integrals(self, t_start, t_end):
        """get_time_step_integrals(t_start, t_end) (main.py:998-1073), every integrand worked
        out when it is read (the values are those of
        cosmo.get_time_step_integrals(t_start, t_end, components, keys))"""
        if self.keys is None:
            return self.cosmo.get_time_step_integrals(t_start, t_end, self.components, self.keys)
        return _LazyIntegrals(self._one_integral, self.keys, t_start, t_end)

    def _one_integral(self, key, t_start, t_end):
        return self.cosmo.get_time_step_integrals(t_start, t_end, self.components, (key,))[key]

    # -- the streaming form of the loop ----------------------------------------------------
    def _stream_begin(self):
        if self.streaming is False or self._shortrange_interactions():
            return
        plan = interactions.pm_streaming_plan(self.components)
        if plan is None:
            if self.streaming:
                raise ConceptGPUError('Timeloop(streaming=True): not the default PM '
                                      'configuration (interactions.pm_streaming_plan)')
            return
        self._plan = plan
        self._rps = [c.to_regions(plan['mesh']) for c in self.components]

    def _stream_end(self, collective=True):
        if self._rps is not None:
            for c, rp in zip(self.components, self._rps):
                c.from_regions(rp, collective=collective)
        self._plan = self._rps = self._spec = None

    def _predict_drift(self, step_type, Δt, sync_time, dump_time):
        """the drift that follows the long kick about to be asked for, as (t_start, t_end), or
        None if none follows (the kick ends at a synchronisation time)"""
        t = self.cosmo.t
        if step_type == 'full':
            # main.py:353-356: the clock after the kick
            t = t + 0.5*Δt
            if t + self.Δt_reltol*Δt + 2*commons.machine_ϵ > sync_time:
                return None
        # main.py:311-314, 436-440 (v_rms plays no part in these)
        if dump_time.t - t <= 1.5*Δt:
            sync_time = dump_time.t
        t_end = self._clip(t + Δt, Δt, sync_time)
        return (t, t_end) if t_end != t else None

    def kick_long(self, Δt, sync_time, step_type):
        if self._rps is None:
            return super().kick_long(Δt, sync_time, step_type)
        t_start = self.t
        t_end = self._clip(t_start + (Δt/2 if step_type == 'init' else Δt), Δt, sync_time)
        if t_start == t_end:
            return
        ᔑdt = self.integrals(t_start, t_end)
        drift = self._next_drift
        before = [rp.snapshot() for rp in self._rps]
        replayed = _streaming_pass(self._plan, self.components, self._rps, ᔑdt,
                                   self.integrals(*drift) if drift is not None else None,
                                   f'at t = {t_start}', drift_on_replay=False)
        self.stream_passes += 1
        if replayed:
            # the pass overflowed: the kick has been taken on the exact path, the drift not at
            # all, and the region objects are new ones — `before` describes buffers that no
            # longer exist.  No guess is outstanding: the loop's next driftkick_short() takes
            # its drift as a pass of its own.
            self._spec = None
        else:
            self._spec = {'kick': ᔑdt, 'drift': drift, 'before': before}

    def driftkick_short(self, Δt, sync_time):
        if self._rps is None:
            return super().driftkick_short(Δt, sync_time)
        t_start = self.t
        t_end = self._clip(t_start + Δt, Δt, sync_time)
        spec, self._spec = self._spec, None
        if spec is not None and spec['drift'] is not None:
            if t_start == t_end or spec['drift'] != (t_start, t_end):
                # the guess was wrong: undo the pass, take the kick alone
                self.stream_wrong_guesses += 1
                for rp, snap in zip(self._rps, spec['before']):
                    rp.restore(snap)
                _streaming_pass(self._plan, self.components, self._rps, spec['kick'], None)
            else:
                return   # this drift was taken with the kick before it
        if t_start == t_end:
            return
        _streaming_pass(self._plan, self.components, self._rps, None,
                        self.integrals(t_start, t_end))

    # -- main.prepare_for_output (main.py:2188-2310), the dump times -----------------------
    def dump_times(self):
        p, cosmo = self.params, self.cosmo
        for tp, at_begin in (('a', cosmo.a), ('t', cosmo.t)):
            if p.output_times[tp] and min(p.output_times[tp]) < at_begin:
                raise ConceptGPUError(
                    f'Cannot produce output at {tp} = {min(p.output_times[tp])}, as the '
                    f'simulation starts at {tp} = {at_begin}.')
        dumps = [DumpTime('t', t=t, a=None) for t in sorted(set(p.output_times['t']))]
        dumps += [DumpTime('a', a=a, t=None) for a in sorted(set(p.output_times['a']))]
        if cosmo.enable_Hubble:
            for i, d in enumerate(dumps):
                if d.time_param == 't':
                    dumps[i] = DumpTime('t', t=d.t, a=cosmo.scale_factor(d.t))
                else:
                    dumps[i] = DumpTime('a', a=d.a, t=cosmo.cosmic_time(d.a))
        elif any(d.t is None for d in dumps):
            raise ConceptGPUError('output_times given as scale factors with the Hubble '
                                  'expansion disabled')
        dumps.sort(key=lambda d: d.t)
        unique = dumps[:1]
        for d in dumps[1:]:
            if not np.isclose(d.t, unique[-1].t, rtol=1e-6, atol=0):
                unique.append(d)
        return unique

    # -- main.get_base_timestep_size (main.py:697-916) -------------------------------------
    # The base step is the smallest of what a list of limiters allows; a limiter is a method
    # that yields (Δt, what it stands for) candidates.  The first of equal candidates names the
    # bottleneck.  (1/|ẇ| and the decay rate: matter has w = 0 and Γ = 0, never the bottleneck;
    # the Courant condition belongs to fluids.)
    LIMITERS = ('_limit_dynamical', '_limit_background', '_limit_fluids', '_limit_pm',
                '_limit_p3m')

    def get_base_timestep_size(self):
        cosmo = self.cosmo
        if self.static_timestepping_func is not None:
            return self.static_timestepping_func(cosmo.a), 'static time-stepping'
        state = {'t': cosmo.t, 'a': cosmo.a, 'v_rms': {}}
        Δt_max, bottleneck = ထ, ''
        for name in self.LIMITERS:
            for Δt, what in getattr(self, name)(state):
                if Δt < Δt_max:
                    Δt_max, bottleneck = Δt, what
        if state['t'] in self.initial_fac_times:
            Δt_max *= self.Δt_initial_fac
        return Δt_max, bottleneck

    def _limit_dynamical(self, state):
        """the dynamical time scale of the mean density (main.py:724-737)"""
        a, ρ_bar = state['a'], 0
        for c in self.components:
            ρ_bar += a**(-3*(1 + c.w_eff(a=a)))*c.ϱ_bar
        yield (self.fac_dynamical/(math.sqrt(self.params.G_Newton*ρ_bar) + commons.machine_ϵ),
               'the dynamical time scale')

    def _limit_background(self, state):
        """the largest allowed Δa at late times; the Hubble time, overruled by a constant Δa
        at early times (main.py:739-771)"""
        cosmo, p = self.cosmo, self.params
        if not cosmo.enable_Hubble:
            return
        t, a = state['t'], state['a']
        a_next = a + p.Δa_max_late
        if a_next < 1:
            yield (p.Δt_base_background_factor*(cosmo.cosmic_time(a_next) - t),
                   'the maximum allowed Δa (late)')
        Δt_hubble, what = self.fac_hubble/cosmo.hubble(a), 'the Hubble time'
        if p.Δa_max_early > 0:
            a_next = a + p.Δa_max_early
            if a_next < 1:
                Δt_early = p.Δt_base_background_factor*(cosmo.cosmic_time(a_next) - t)
                if Δt_early > Δt_hubble:
                    Δt_hubble, what = Δt_early, 'the maximum allowed Δa (early)'
        yield Δt_hubble, what

    def _v_rms(self, c, state):
        """rms velocity of a component, measured once per call of get_base_timestep_size
        (on the regions when the loop streams); a static component counts as just above 0"""
        if c not in state['v_rms']:
            rp = self._rps[self.components.index(c)] if self._rps is not None else None
            state['v_rms'][c] = measure(c, 'v_rms', state['a'], rp)
        v_rms = state['v_rms'][c]
        return commons.machine_ϵ if v_rms < commons.machine_ϵ else v_rms

    def _limit_fluids(self, state):
        """what the caller's fluid solver allows (its Courant condition, main.py:773-836)"""
        if self.fluid_limiter is None:
            return
        for c in self.components:
            if c.representation == 'fluid':
                yield self.fluid_limiter(c, state['a']), f'the fluid solver of {c.name}'

    def _limit_pm(self, state):
        """a particle may cross a fraction of a cell of its finest PM mesh (main.py:838-872)"""
        p = self.params
        for c in self.particles:
            finest = (0, None)   # (of equal grid sizes the first force names the bottleneck)
            for force, method in c.forces.items():
                for method_, gridsize in c.potential_gridsizes[force].items():
                    if method == method_ == 'pm' and int(np.max(gridsize)) > finest[0]:
                        finest = (int(np.max(gridsize)), force)
            if finest[0] == 0:
                continue
            yield (self.fac_pm*(p.boxsize/finest[0])/self._v_rms(c, state),
                   f'the PM method of the {finest[1]} force for {c.name}')

    def _limit_p3m(self, state):
        """... or of the short-range scale of its P³M force (main.py:873-906); the short-range
        parameters are resolved with the global P³M grid size (commons.py:3254-3300)"""
        p = self.params
        for c in self.particles:
            scales = []
            for force, method in c.forces.items():
                if method != 'p3m':
                    continue
                gs = p.potential_options['gridsize']['global'].get(force, {}).get('p3m', -1)
                if gs == -1:
                    gs = int(np.max(c.potential_gridsizes[force]['p3m']))
                scales.append(commons.resolve_shortrange(p, gs)['scale'])
            if scales:
                yield (self.fac_p3m*min(scales)/self._v_rms(c, state),
                       f'the P³M method of the gravity force for {c.name}')

    # -- main.update_base_timestep_size (main.py:922-982) ----------------------------------
    def update_base_timestep_size(self, Δt, Δt_min, Δt_max, bottleneck, time_step=-1,
                                  time_step_last_sync=-1, *, allow_increase=True,
                                  tolerate_danger=False):
        p, cosmo = self.params, self.cosmo
        if Δt > Δt_max:
            Δt_new = self.Δt_reduce_fac*Δt_max
            Δt_ratio = Δt_new/Δt
            if Δt_ratio < self.Δt_ratio_abort and not tolerate_danger:
                raise ConceptGPUError(
                    f'Due to {bottleneck}, the time step size needs to be rescaled by a factor '
                    f'{Δt_ratio:.1g}. This extreme change is unacceptable.')
            if Δt_new < Δt_min:
                raise ConceptGPUError('Time evolution effectively halted with a time step size '
                                      f'of {Δt_new}')
            return Δt_new, bottleneck
        if not allow_increase:
            return Δt, bottleneck
        Δt_new = self.Δt_increase_fac*Δt_max
        if Δt_new < Δt:
            Δt_new = Δt
        period_frac = (time_step + 1 - time_step_last_sync)*(1/self.Δt_period)
        if period_frac > 1:
            period_frac = 1
        elif period_frac < 0:
            period_frac = 0
        Δt_tmp = (1 + period_frac*(p.Δt_increase_max_factor - 1))*Δt
        if Δt_new > Δt_tmp:
            Δt_new = Δt_tmp
        if cosmo.enable_Hubble and cosmo.t + Δt_new > cosmo.cosmic_time(1):
            return Δt, 'a ≈ 1'
        return Δt_new, ''

    def drift_fluids(self, Δt, sync_time):
        """main.drift_fluids (main.py:1279-1299): always over a full base step"""
        fluids = [c for c in self.components if c.representation == 'fluid']
        if not fluids:
            return
        t_start = self.cosmo.t
        t_end = self._clip(t_start + Δt, Δt, sync_time)
        if t_start == t_end:
            return
        ᔑdt = self.integrals(t_start, t_end)
        a_end = self.cosmo.scale_factor(t_end)
        for c in fluids:
            self.fluid_drift(c, ᔑdt, a_end)

    def _advance(self, Δt, sync_time):
        # universals.t += 0.5*Δt, snapped onto the sync time (main.py:343-346, 353-356)
        cosmo = self.cosmo
        cosmo.t += 0.5*Δt
        if cosmo.t + self.Δt_reltol*Δt + 2*commons.machine_ϵ > sync_time:
            cosmo.t = sync_time
        cosmo.a = cosmo.scale_factor(cosmo.t)

    def _dump(self, dump_time):
        if self.on_dump is not None:
            streaming = self._rps is not None
            if streaming:   # the callback looks at the Components' own arrays
                for c, rp in zip(self.components, self._rps):
                    c.from_regions(rp)
            self.on_dump(self, dump_time)
            if streaming:
                self._rps = [c.to_regions(self._plan['mesh']) for c in self.components]

    def snapshot_dumper(self, output_dir, output_base='snapshot', only_snapshot_times=False,
                        **save_options):
        """An on_dump callback that writes a GADGET snapshot per dump time (main.dump,
        main.py:1660-1760, for snapshots of snapshot_type = 'gadget'; concept_amd.snapshot.save)
        named like the reference's — <output_dir>/<output_base>_<a|t>=<value> with just enough
        digits that neighbouring dumps and the initial time differ (prepare_for_output,
        main.py:2242-2278).  Usage: loop.on_dump = loop.snapshot_dumper('output/run').
        only_snapshot_times: write at the times output_times lists for 'snapshot' only (the
        other output kinds' times are dumps of the loop too)."""
        from . import snapshot
        p = self.params
        fmts = {}
        for kind, begin in (('a', self.cosmo.a), ('t', self.cosmo.t)):
            times = sorted(set((begin,) + tuple(p.output_times[kind])))
            if len(times) < 2 and not p.output_times[kind]:
                continue
            ndigits = 0
            while True:
                fmt = f'{{:.{ndigits}f}}'
                if (len(set(fmt.format(ot) for ot in times)) == len(times)
                        and (fmt.format(times[0]) != fmt.format(0) or not times[0])):
                    break
                ndigits += 1
            fmts[kind] = ndigits
        ndigits = max(fmts.values()) if fmts else 2
        sep = '_' if output_base else ''
        self.snapshots_written = []

        def on_dump(loop, dump_time):
            value = dump_time.a if dump_time.time_param == 'a' else dump_time.t
            if only_snapshot_times and not any(
                    abs(value - v) <= 1e-12*max(abs(v), 1e-300)
                    for v in p.snapshot_times[dump_time.time_param]):
                return
            name = f'{output_dir}/{output_base}{sep}{dump_time.time_param}={value:.{ndigits}f}'
            # GADGET snapshots hold particles: fluid components (riding along through
            # fluid_drift) are left out, as the reference's writer leaves them out
            # (snapshot.py: GadgetSnapshot.populate keeps the particle components)
            particles = [c for c in loop.components if c.representation == 'particles']
            left_out = [c.name for c in loop.components if c.representation != 'particles']
            if left_out and not getattr(loop, '_warned_fluid_snapshot', False):
                import warnings
                warnings.warn('GADGET snapshots hold particle components only: '
                              f'{", ".join(left_out)} not written')
                loop._warned_fluid_snapshot = True
            if not particles:
                return
            fn = snapshot.save(particles, name, a=loop.cosmo.a,
                               **{'output_base': output_base or 'snapshot', **save_options})
            loop.snapshots_written.append(fn)
        return on_dump

    # -- main.timeloop (main.py:102-471) ---------------------------------------------------
    def run(self):
        try:
            self._run()
        except BaseException:
            # unwinding: hand the particles back without the collective part (the other
            # domains may not be unwinding), then let the exception travel
            try:
                self._stream_end(collective=False)
            except Exception:
                pass
            raise
        self._stream_end()

    def _run(self):
        cosmo, components = self.cosmo, self.components
        dump_times = self.dump_times()
        if not dump_times:
            return
        if dump_times[0].t == cosmo.t or dump_times[0].a == cosmo.a:
            self._dump(dump_times[0])
            dump_times.pop(0)
            if not dump_times:
                return
        self._stream_begin()
        self.initial_fac_times.add(cosmo.t)
        Δt_max, bottleneck = self.get_base_timestep_size()
        Δt_begin = Δt_max
        if Δt_begin > dump_times[0].t - cosmo.t:
            Δt_begin = dump_times[0].t - cosmo.t
        Δt = Δt_begin
        Δt_min = 1e-4*Δt_begin
        self.keys = cosmo.integrand_keys(components)
        self.initialize_rung_populations(Δt)
        time_step = time_step_last_sync = 0
        time_step_previous = time_step - 1
        bottleneck = ''
        time_step_type = 'init'
        sync_time = ထ
        recompute_Δt_max = True
        Δt_backup = -1
        for dump_index, dump_time in enumerate(dump_times):
            while True:
                if time_step > time_step_previous:
                    time_step_previous = time_step
                    if time_step_type == 'init':
                        for c in self.particles:
                            c.assign_rungs(Δt, self.fac_softening)
                    self.time_step, self.Δt = time_step, Δt
                    Δt_print = Δt
                    if cosmo.t + Δt*(1 + self.Δt_reltol) + 2*commons.machine_ϵ > sync_time:
                        Δt_print = sync_time - cosmo.t
                    self.history.append((time_step, cosmo.t, cosmo.a, Δt_print))
                    if self.on_step is not None:
                        self.on_step(self)
                if time_step_type == 'init':
                    time_step_type = 'full'
                    self._next_drift = self._predict_drift('init', Δt, sync_time, dump_time)
                    self.kick_long(Δt, sync_time, 'init')
                    self.kick_short(Δt)
                    if dump_time.t - cosmo.t <= 1.5*Δt:
                        sync_time = dump_time.t
                        continue
                    Δt_max, bottleneck = self.get_base_timestep_size()
                    if Δt > Δt_max:
                        sync_time = cosmo.t + 0.5*Δt
                        recompute_Δt_max = False
                        continue
                elif time_step_type == 'full':
                    self.drift_fluids(Δt, sync_time)
                    self.driftkick_short(Δt, sync_time)
                    self._advance(Δt, sync_time)
                    self._next_drift = self._predict_drift('full', Δt, sync_time, dump_time)
                    self.kick_long(Δt, sync_time, 'full')
                    self._advance(Δt, sync_time)
                    if cosmo.t == sync_time:
                        time_step_type = 'init'
                        sync_time = ထ
                        if Δt_backup != -1:
                            if Δt < Δt_backup:
                                Δt = Δt_backup
                            Δt_backup = -1
                        if recompute_Δt_max:
                            Δt_max, bottleneck = self.get_base_timestep_size()
                        recompute_Δt_max = True
                        Δt, bottleneck = self.update_base_timestep_size(
                            Δt, Δt_min, Δt_max, bottleneck, time_step, time_step_last_sync,
                            tolerate_danger=(bottleneck == 'static time-stepping'))
                        time_step += 1
                        time_step_last_sync = time_step
                        if cosmo.t == dump_time.t:
                            self._dump(dump_time)
                            if dump_index != len(dump_times) - 1:
                                Δt_max = dump_times[dump_index + 1].t - cosmo.t
                                if Δt > Δt_max:
                                    Δt_backup = Δt
                                    Δt = Δt_max
                            break
                        Δt_max = dump_time.t - cosmo.t
                        if Δt > Δt_max:
                            Δt_backup = Δt
                            Δt = Δt_max
                        continue
                    time_step += 1
                    if dump_time.t - cosmo.t <= 1.5*Δt:
                        sync_time = dump_time.t
                        continue
                    Δt_max, bottleneck = self.get_base_timestep_size()
                    if Δt > Δt_max:
                        sync_time = cosmo.t + Δt
                        recompute_Δt_max = False
                        continue
                    if (Δt_max > self.Δt_increase_min_factor*Δt
                            and (time_step + 1 - time_step_last_sync) >= self.Δt_period):
                        sync_time = cosmo.t + Δt
                        recompute_Δt_max = False
                        continue
        self.time_step, self.Δt = time_step, Δt
        self.history.append((time_step, cosmo.t, cosmo.a, Δt))


def get_initial_conditions(params=None, device=None):
    """main.get_initial_conditions (main.py:2080-2186) for the case of a snapshot on disk: the
    components of the file `initial_conditions` names (GADGET-2, one file or several; over several
    domains every rank reads its own rows).  Component specifications to be REALISED (dicts in
    `initial_conditions`) need the reference's IC generator and CLASS: outside this path."""
    from . import snapshot
    p = params or commons.params
    ic = p.initial_conditions
    if not ic:
        return []
    if not isinstance(ic, str):
        raise ConceptGPUError('initial_conditions: only the path of a snapshot is supported here '
                              '(realising components needs the reference\'s IC generator)')
    gsp = p.gadget_snapshot_params
    return snapshot.load(ic, params=p, units=gsp['units']).to_components(device=device)
