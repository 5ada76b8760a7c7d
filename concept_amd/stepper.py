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
from . import interactions
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
