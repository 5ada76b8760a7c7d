"""concept_amd.commons — the few parameters, units and constants the PM/P3M
gravity path reads, under the reference's names and defaults.

Mirrors (does not import) the relevant slice of the reference's commons.py:
  unit system            commons.py:1828-1886, 2040-2110 (default Mpc, Gyr, 1e10 m_sun)
  G_Newton               commons.py:2130-2134
  boxsize                commons.py:2956
  potential_options      commons.py:2958-3237  (gridsize / interpolation / deconvolve /
                                                interlace / differentiation)
  shortrange_params      commons.py:3254-3275  (scale, range, tilesize, subtiling, tablesize)
  select_forces          commons.py:3664-3700
  select_softening_length commons.py:3867-3872, softening_kernel :3862
  N_rungs                commons.py:3891
  cell_centered          commons.py:3926
  nghosts                commons.py:4411-4432
A parameter file is Python source executed in a namespace that holds the
units (commons.py:2001-2140); `load_params` does the same for the names above
and ignores everything else (I/O, cosmology tables, ... are out of scope).
"""
import math
import types

import numpy as np

machine_ϵ = float(np.finfo(np.float64).eps)  # commons.py:1814
π = float(np.pi)
τ = 2*π
ထ = float('inf')

interpolation_orders = {'NGP': 1, 'CIC': 2, 'TSC': 3, 'PCS': 4}


def _unit_relations():
    r = {'yr': 1.0, 'pc': 1.0, 'm_sun': 1.0}
    r['kyr'] = 1e+3*r['yr']
    r['Myr'] = 1e+6*r['yr']
    r['Gyr'] = 1e+9*r['yr']
    r['day'] = 1/365.25*r['yr']
    r['hr'] = 1/24*r['day']
    r['minutes'] = 1/60*r['hr']
    r['s'] = 1/60*r['minutes']
    r['kpc'] = 1e+3*r['pc']
    r['Mpc'] = 1e+6*r['pc']
    r['Gpc'] = 1e+9*r['pc']
    r['AU'] = τ/(60*60*360)*r['pc']
    r['m'] = 1/149597870700*r['AU']
    r['km'] = 1e+3*r['m']
    r['kg'] = 1/1.98841e+30*r['m_sun']
    r['G_Newton'] = 6.67430e-11*r['m']**3/(r['kg']*r['s']**2)
    r['light_speed'] = 299792458*r['m']/r['s']
    return r


class Params(types.SimpleNamespace):
    """All parameters of the path, reference names.  Build with load_params()."""


def _make_units(unit_length='Mpc', unit_time='Gyr', unit_mass='10**(10)*m_sun'):
    rel = _unit_relations()
    ns = dict(rel)
    yr = 1/eval(unit_time, {}, ns)
    pc = 1/eval(unit_length, {}, ns)
    m_sun = 1/eval(unit_mass, {}, ns)
    u = {'yr': yr, 'pc': pc, 'm_sun': m_sun}
    for k in ('kyr', 'Myr', 'Gyr', 'day', 'hr', 'minutes', 's'):
        u[k] = rel[k]*yr
    for k in ('kpc', 'Mpc', 'Gpc', 'AU', 'm', 'km'):
        u[k] = rel[k]*pc
    u['kg'] = rel['kg']*m_sun
    G = (rel['G_Newton']/(rel['m']**3/(rel['kg']*rel['s']**2))
         *u['m']**3/(u['kg']*u['s']**2))
    c = rel['light_speed']/(rel['m']/rel['s'])*u['m']/u['s']
    return u, G, c


def _method_dict(user, default_pm, default_p3m):
    """{'gravity': {'pm': x, 'p3m': y}} with reference defaults filled in."""
    out = {'gravity': {'pm': default_pm, 'p3m': default_p3m}}
    if isinstance(user, dict):
        for force, d in user.items():
            if isinstance(d, dict):
                out.setdefault(force, {}).update(d)
            else:
                out.setdefault(force, {})
                for m in ('pm', 'p3m'):
                    out[force][m] = d
    elif user is not None:
        for m in ('pm', 'p3m'):
            out['gravity'][m] = user
    return out


_PATH_PARAMETERS = ('G_Newton', 'N_rungs', 'boxsize', 'cell_centered', 'ewald_gridsize', 'nghosts',
                    'potential_options', 'select_forces', 'select_softening_length',
                    'shortrange_params', 'softening_kernel', 'H0', 'Ωb', 'Ωcdm', 'a_begin',
                    'enable_Hubble', 'Δt_base_background_factor', 'Δt_base_nonlinear_factor',
                    'Δt_increase_max_factor', 'Δt_rung_factor', 'Δa_max_early', 'Δa_max_late',
                    'static_timestepping', 'output_times', 't_begin', 'output_dirs', 'output_bases',
                    'snapshot_type', 'gadget_snapshot_params', 'initial_conditions')


def load_params(source=None, **overrides):
    """`source`: path to a parameter file, parameter text, a dict, or None.
    Returns a Params object and makes it the active one (`commons.params`)."""
    global params
    units, G_Newton, light_speed = _make_units()
    user = {}
    if isinstance(source, dict):
        user.update(source)
    elif isinstance(source, str):
        text = source
        if '\n' not in source and '=' not in source:
            with open(source, encoding='utf-8') as f:
                text = f.read()
        ns = dict(units)
        ns.update(π=π, pi=π, τ=τ, ထ=ထ, inf=ထ, sqrt=math.sqrt, cbrt=np.cbrt, h=1.0)
        # the reference executes the file repeatedly until all names resolve
        # (commons.py:2001-2040); here: the whole file, else statement by statement (whole
        # statements — dict literals span many lines — skipping those that use names outside
        # this path: output paths, CLASS, ...), twice, so that later definitions reach earlier
        # uses.  `path` and `param` (directories and the parameter file's own name in the
        # reference) resolve to placeholders.
        class _Placeholder(str):
            def __getattr__(self, name):
                return _Placeholder(name)
        ns.setdefault('path', _Placeholder('path'))
        if not isinstance(source, str) or '\n' in source or '=' in source:
            ns['param'] = _Placeholder('param')
        else:
            ns['param'] = _Placeholder(source)
        import ast
        try:
            statements = [ast.get_source_segment(text, node) for node in ast.parse(text).body]
        except SyntaxError:
            statements = text.split('\n')
        # passes until nothing changes any more: a statement that failed may succeed once a
        # later one has defined the name it uses (forward references), and h follows H0
        # (commons.py:1790-1792): a file that uses it (boxsize = 200*Mpc/h) before defining H0
        # gets it right on the next pass
        # (the reference re-executes every line until the set of lines that ran stops growing,
        # commons.py:2001-2040; here a pass after the first re-runs only what failed before —
        # a statement with side effects runs once — unless h changed, which every line that
        # uses it must see)
        failed, todo = set(), None   # todo None: everything
        for _ in range(16):
            failed_prev = failed
            failed = set()
            whole = False
            if todo is None:
                try:
                    exec(text, ns)
                    whole = True
                except Exception:
                    pass
            if not whole:
                for i, st in enumerate(statements):
                    if todo is not None and i not in todo:
                        continue
                    try:
                        exec(st, ns)
                    except Exception:
                        failed.add(i)
            h_new = ns['H0']/(100*ns['km']/(ns['s']*ns['Mpc'])) if 'H0' in ns else 1.0
            h_same = (h_new == ns['h'])
            ns['h'] = h_new
            if h_same and (not failed or failed == failed_prev):
                break
            todo = None if not h_same else set(failed)
        # a statement of this path's own parameters that never ran must not pass silently
        for i in sorted(failed):
            target = statements[i].split('=')[0].strip()
            if target in _PATH_PARAMETERS:
                import warnings
                warnings.warn(f'parameter file: the statement assigning {target!r} could not '
                              f'be executed; its default is used')
        user.update({k: v for k, v in ns.items() if not k.startswith('__')})
    user.update(overrides)
    p = Params()
    p.units = types.SimpleNamespace(**units)
    p.G_Newton = float(user.get('G_Newton', G_Newton))
    p.light_speed = light_speed
    p.boxsize = float(user.get('boxsize', 512*units['Mpc']))
    po = user.get('potential_options', {})
    if not isinstance(po, dict):
        po = {'gridsize': po}
    p.potential_options = {
        'gridsize': {'global': _method_dict(po.get('gridsize', {}).get('global')
                                            if isinstance(po.get('gridsize'), dict)
                                            and 'global' in po.get('gridsize', {})
                                            else po.get('gridsize'), -1, -1)},
        'interpolation': _method_dict(po.get('interpolation'), 'CIC', 'CIC'),
        'deconvolve': _method_dict(po.get('deconvolve'), (True, True), (True, True)),
        'interlace': _method_dict(po.get('interlace'), ('sc', 'sc'), ('sc', 'sc')),
        'differentiation': {'default': _method_dict(None, 2, 4)},
    }
    for force, d in p.potential_options['interpolation'].items():
        for m, v in d.items():
            if isinstance(v, str):
                d[m] = interpolation_orders[v.upper()]
    diff = po.get('differentiation')
    if isinstance(diff, dict):
        for name, d in diff.items():
            p.potential_options['differentiation'][name] = _method_dict(d, 2, 4)
    # 'fourier' = differentiation in Fourier space, encoded as order 0 (commons.py:3220-3233)
    for name, d0 in p.potential_options['differentiation'].items():
        for force, d1 in d0.items():
            for m, v in d1.items():
                if isinstance(v, str):
                    if v.lower() != 'fourier':
                        raise ValueError(
                            f'Invalid potential_options["differentiation"] value {v}')
                    d1[m] = 0
                else:
                    d1[m] = int(round(v))
    # component-level (upstream, downstream) grid sizes: every key of
    # potential_options['gridsize'] other than 'global' selects components
    # (commons.py:3095-3207; looked up with is_selected in species.py:1147-1160)
    gs = po.get('gridsize')
    if isinstance(gs, dict) and 'global' in gs:
        for sel, d in gs.items():
            if sel == 'global':
                continue
            entry = {}
            if isinstance(d, dict):
                for force, dm in d.items():
                    if isinstance(dm, dict):
                        entry[force] = dict(dm)
                    else:
                        entry[force] = {'pm': dm, 'p3m': dm}
            else:
                entry['gravity'] = {'pm': d, 'p3m': d}
            p.potential_options['gridsize'][sel] = entry
    # short-range parameters (strings are evaluated per use with the grid size known)
    sr = {'scale': '1.25*boxsize/gridsize', 'range': '4.5*scale', 'tilesize': 'range',
          'subtiling': 'automatic', 'tablesize': 2**12}
    sr.update((user.get('shortrange_params') or {}).get('gravity', {}))
    p.shortrange_params = {'gravity': sr}
    # select_forces (commons.py:3664-3700): {selector: {force: method}}, or {selector: 'force'}
    # for the force's default method; absent, it follows the methods the global grid sizes
    # were given for — only 'pm': everything by PM; 'p3m' (with or without 'pm'): particles by
    # P3M, fluids by PM
    default_force_method = {'gravity': 'p3m'}
    methods_implemented = ('ppnonperiodic', 'pp', 'p3m', 'pm')
    sf = {}
    for key, val in dict(user.get('select_forces') or {}).items():
        key = str(key).strip().lower() if key != 'default' else key
        if isinstance(val, dict):
            sf[key] = {str(f).strip().lower(): str(m).strip().lower() for f, m in val.items()}
        elif isinstance(val, str):
            force = val.strip().lower()
            if force not in default_force_method:
                raise ValueError(f'select_forces: force "{val}" has no default method')
            sf[key] = {force: default_force_method[force]}
        else:
            raise ValueError(f'select_forces[{key!r}] = {val!r} not understood')
        for force, method in sf[key].items():
            if method not in methods_implemented:
                raise ValueError(f'select_forces: method "{method}" for force "{force}" not '
                                 f'implemented (methods: {methods_implemented})')
    if not sf:
        for force, dm in p.potential_options['gridsize']['global'].items():
            methods = {m for m, g in dm.items() if g != -1}
            if not methods:
                continue
            sf.setdefault('particles', {})
            sf.setdefault('fluid', {})
            if methods == {'pm'}:
                sf['particles'].setdefault(force, 'pm')
                sf['fluid'].setdefault(force, 'pm')
            elif methods in ({'p3m'}, {'pm', 'p3m'}):
                sf['particles'].setdefault(force, 'p3m')
                sf['fluid'].setdefault(force, 'pm')
            else:
                raise ValueError(f'Force methods "{methods}" from potential_options'
                                 f'["gridsize"]["global"]["{force}"] not understood')
    p.select_forces = sf
    ssl = user.get('select_softening_length') or {}
    if not isinstance(ssl, dict):
        ssl = {'all': ssl}
    ssl = dict(ssl)
    ssl.setdefault('default', '0.025*boxsize/cbrt(N)')
    p.select_softening_length = ssl
    p.softening_kernel = str(user.get('softening_kernel', 'spline')).lower()
    p.ewald_gridsize = int(user.get('ewald_gridsize', 64))  # commons.py:3061
    p.N_rungs = int(user.get('N_rungs', 8))
    p.cell_centered = bool(user.get('cell_centered', True))
    # nghosts (commons.py:4411-4432): default 2 comes from the PCS default of the
    # power-spectrum options; force interpolation and differentiation orders raise it
    # (powerspec_options default: PCS, interlaced -> 4//2 = 2, + 1 with cell-vertex grids)
    nghosts = 2 + (0 if p.cell_centered else 1)
    for force, d in p.potential_options['interpolation'].items():
        for m, order in d.items():
            lattices = tuple(p.potential_options['interlace'].get(force, {}).get(m, ('sc', 'sc')))
            nghosts = max(nghosts, order//2 + int(lattices != ('sc', 'sc') and order % 2 != 0))
    for name, d0 in p.potential_options['differentiation'].items():
        for force, d1 in d0.items():
            nghosts = max(nghosts, (max(d1.values()) + 1)//2)
    p.nghosts = int(user.get('nghosts', nghosts))
    # cosmology and time stepping (commons.py:3630-3638, 3875-3890, 4313-4319, 4435-4479)
    u = p.units
    p.H0 = float(user.get('H0', 67*u.km/(u.s*u.Mpc)))
    p.Ωb = float(user.get('Ωb', 0.049))
    p.Ωcdm = float(user.get('Ωcdm', 0.27))
    p.Ωm = p.Ωb + p.Ωcdm
    p.a_begin = float(user.get('a_begin', 1))
    p.t_begin = float(user.get('t_begin', 0))
    p.enable_Hubble = bool(user.get('enable_Hubble', True))
    p.ρ_crit = 3*p.H0**2/(8*π*p.G_Newton)
    p.ρ_mbar = p.Ωm*p.ρ_crit
    p.Δt_base_background_factor = float(user.get('Δt_base_background_factor', 1))
    p.Δt_base_nonlinear_factor = float(user.get('Δt_base_nonlinear_factor', 1))
    p.Δt_increase_max_factor = float(user.get('Δt_increase_max_factor', ထ))
    p.Δt_rung_factor = float(user.get('Δt_rung_factor', 1))
    p.Δa_max_early = float(user.get('Δa_max_early', 0.00153))
    p.Δa_max_late = float(user.get('Δa_max_late', 0.022))
    p.static_timestepping = user.get('static_timestepping') or None
    if p.Δt_increase_max_factor <= 1:
        raise ValueError('You must have Δt_increase_max_factor > 1')
    if p.Δa_max_early < 0:
        raise ValueError('You must have Δa_max_early ≥ 0')
    if p.Δa_max_late <= 0:
        raise ValueError('You must have Δa_max_late > 0')
    if p.static_timestepping is not None and not p.enable_Hubble:
        raise ValueError('You may not specify static_timestepping with the Hubble expansion '
                         'disabled')
    # output_times (commons.py: {'a': {kind: (...)}, 't': {kind: (...)}}): the dump times of the
    # time loop.  Accepted: {kind: times} (scale factors when the Hubble expansion is enabled,
    # cosmic times otherwise), {'a': ..., 't': ...} with {kind: times} or plain times inside.
    def _times(v):
        if v is None:
            return ()
        if isinstance(v, dict):
            out = []
            for w in v.values():
                out += list(_times(w))
            return tuple(out)
        if isinstance(v, (int, float)):
            return (float(v),)
        return tuple(float(x) for x in v)
    ot = user.get('output_times') or {}
    default_param = 'a' if p.enable_Hubble else 't'
    p.output_times = {'a': (), 't': ()}
    if isinstance(ot, dict) and set(ot) & {'a', 't'}:
        for tp in ('a', 't'):
            p.output_times[tp] = _times(ot.get(tp))
        rest = {k: v for k, v in ot.items() if k not in ('a', 't')}
        p.output_times[default_param] += _times(rest)
    else:
        p.output_times[default_param] = _times(ot)
    # which of those dump times are snapshot times, per time parameter (the other output kinds —
    # power spectra, renders — are dumps of the time loop too, but nothing is written for them)
    def _kind_times(v, kind):
        if isinstance(v, dict):
            return _times(v.get(kind))
        return ()
    p.snapshot_times = {'a': (), 't': ()}
    if isinstance(ot, dict) and set(ot) & {'a', 't'}:
        for tp in ('a', 't'):
            p.snapshot_times[tp] = _kind_times(ot.get(tp), 'snapshot')
        p.snapshot_times[default_param] += _kind_times(ot, 'snapshot')
    else:
        p.snapshot_times[default_param] = _kind_times(ot, 'snapshot')
    # input / output (commons.py:2547-2572, 2787-2830): where snapshots go, what they are
    # called and which format they have
    od = user.get('output_dirs', {})
    if isinstance(od, str):
        od = {'snapshot': od}
    p.output_dirs = {k: str(v) for k, v in dict(od).items() if v}
    p.output_bases = {'snapshot': 'snapshot'}
    p.output_bases.update({k: str(v) for k, v in dict(user.get('output_bases', {})).items()})
    p.snapshot_type = str(user.get('snapshot_type', 'concept')).lower()
    p.initial_conditions = user.get('initial_conditions', '')
    gsp = {'snapformat': 2, 'dataformat': {'POS': 32, 'VEL': 32, 'ID': 'automatic'},
           'particles per file': 'automatic', 'header': {},
           'units': {'length': 'kpc/h', 'velocity': 'km/s', 'mass': '10**10*m_sun/h'}}
    for key, val in dict(user.get('gadget_snapshot_params', {})).items():
        simple = str(key).lower().replace(' ', '').replace('_', '').replace('-', '')
        for known in gsp:
            if simple == known.replace(' ', ''):
                if isinstance(gsp[known], dict) and isinstance(val, dict):
                    gsp[known] = dict(gsp[known], **val)
                else:
                    gsp[known] = val
                break
    gsp['snapformat'] = int(gsp['snapformat'])
    if gsp['snapformat'] not in (1, 2):
        raise ValueError(f'gadget_snapshot_params["snapformat"] = {gsp["snapformat"]} but must '
                         'be 1 or 2')
    p.gadget_snapshot_params = gsp
    p.user = user
    params = p
    return p


def is_selected(component, d, accumulate=False, default=None):
    """Look a component up in a selection dict (commons.py:5471-5600): keys may be
    'default', 'all', the representation, the species, or the component's name —
    later ones take precedence; accumulate = True merges every match (dicts are
    merged in that order)."""
    lowered = {(k.lower() if isinstance(k, str) else k): v for k, v in d.items()}
    species = component.species.lower()
    keys = ['default', 'all', component.representation.lower()]
    keys += [s_ for s_ in species.split('+') if s_ != species] + [species, component.name.lower()]
    found = [lowered[k] for k in keys if k in lowered]
    if not found:
        return default
    if not accumulate:
        return found[-1]
    if all(isinstance(v, dict) for v in found):
        merged = {}
        for v in found:
            merged.update(v)
        return merged
    return found


def resolve_shortrange(p, gridsize):
    """Numerical shortrange_params['gravity'] for a given P3M grid size
    (commons.py:3254-3275 evaluates the strings with boxsize/gridsize known)."""
    sr = p.shortrange_params['gravity']
    ns = {'boxsize': p.boxsize, 'gridsize': gridsize, **vars(p.units)}
    out = {}
    for key in ('scale', 'range', 'tilesize'):
        v = sr[key]
        out[key] = float(eval(v, {}, ns)) if isinstance(v, str) else float(v)
        ns[key] = out[key]
    out['subtiling'] = sr['subtiling']
    out['tablesize'] = int(sr['tablesize'])
    return out


def softening_length(p, species, N):
    """select_softening_length (commons.py:3867-3872)."""
    expr = p.select_softening_length.get(species, p.select_softening_length.get(
        'all', p.select_softening_length['default']))
    if isinstance(expr, str):
        return float(eval(expr, {}, {'boxsize': p.boxsize, 'N': N, 'cbrt': np.cbrt,
                                     **vars(p.units)}))
    return float(expr)


params = None


def upload(values, device, dtype=None):
    """a small host array as a CUDA tensor without waiting for the GPU: through pinned memory
    (torch's caching host allocator hands the block out again only after the copy has run) —
    torch.tensor(..., device='cuda') copies from pageable memory, which waits for everything
    queued on the stream before it"""
    import numpy as np
    import torch
    h = torch.from_numpy(np.ascontiguousarray(values, dtype=dtype or np.float64))
    if torch.device(device).type != 'cuda':
        return h.clone()
    return h.pin_memory().to(device, non_blocking=True)
