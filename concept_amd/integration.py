"""The cosmic clock of the time loop: Friedmann background, a(t) / t(a) / H(a) splines and the
time-step integrals ᔑ f(a(t)) dt that every kick and drift of the path is scaled with.

Host-side restatement of the reference's
  Spline                      integration.py:40-320   (natural cubic spline, SciPy/GSL)
  remove_doppelgängers        integration.py:403-545
  hubble / scale_factor / cosmic_time / ȧ   integration.py:570-660
  scalefactor_integral        integration.py:712-827
  init_time                   integration.py:864-1001
  solve_matterΛ_background    integration.py:1043-1180 (a(t) part; the growth factors belong to
                              the IC generator, out of scope)
with `enable_class_background = False`: flat matter + Λ, H(a) = H0 sqrt(Ωm a⁻³ + 1 − Ωm).
(The CLASS background needs CLASS; the reference's own P³M validation test runs without it,
test/concept_vs_gadget_p3m/param:41.)  Matter components only: w_eff = 0, Γ = 0.

Nothing here touches the GPU: these are a few hundred scalars per run.
"""
import math

import numpy as np

# the reference's pure-Python mode — the mode the goldens are generated in — takes exp, log and
# sqrt from numpy (commons.py:1234-1243), its compiled mode from libm; they differ in the last
# bit here and there, which the ODE solver and the event root-finder amplify to ~1e-13
exp, log, sqrt = np.exp, np.log, np.sqrt

from . import commons
from .lib import ConceptGPUError

machine_ϵ = commons.machine_ϵ


def remove_doppelgängers(x, y, rel_tol=1e-1):
    """integration.py:403-545 for one x array: consecutive x values that are (nearly) equal —
    closer than rel_tol times the previous accepted spacing — collapse to one point; first and
    last point always kept.  Returns new arrays."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)[:x.shape[0]]
    size = x.shape[0]
    if size < 2:
        return x.copy(), y.copy()
    if np.any(np.diff(x) < 0):
        raise ConceptGPUError('The values in the x array passed to remove_doppelgängers() are '
                              'not in increasing order')
    # right to left
    accepted = [size - 1, size - 2]
    x_prev = x[size - 2]
    xdiff_prev = x[size - 1] - x_prev
    for i in range(size - 3, -1, -1):
        xdiff = x_prev - x[i]
        if xdiff > rel_tol*xdiff_prev:
            accepted.append(i)
            x_prev = x[i]
            xdiff_prev = xdiff
    accepted.reverse()
    xc, yc = x[accepted].copy(), y[accepted].copy()
    xc[0], yc[0] = x[0], y[0]          # always include the first point
    # left to right
    accepted = [0, 1]
    x_prev = xc[1]
    xdiff_prev = x_prev - xc[0]
    for i in range(2, xc.shape[0]):
        xdiff = xc[i] - x_prev
        if xdiff > rel_tol*xdiff_prev:
            accepted.append(i)
            x_prev = xc[i]
            xdiff_prev = xdiff
    last_x, last_y = xc[-1], yc[-1]
    xc, yc = xc[accepted].copy(), yc[accepted].copy()
    xc[-1], yc[-1] = last_x, last_y    # always include the last point
    return xc, yc


class Spline:
    """integration.py:40-320: natural cubic spline of tabulated y(x), optionally of
    (log x, log y); eval, eval_deriv, integrate, with the reference's handling of arguments
    just outside the table."""
    size_min = 3

    def __init__(self, x, y, name='', *, logx=False, logy=False):
        import scipy.interpolate
        self.name = name
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        if x.shape[0] != y.shape[0]:
            raise ConceptGPUError(f'Spline "{name}": arrays of lengths {x.shape[0]} and '
                                  f'{y.shape[0]} were passed')
        if x.shape[0] < self.size_min:
            raise ConceptGPUError(
                f'Spline "{name}": Too few tabulated values ({x.shape[0]}) were given for '
                f'cubic spline interpolation. At least {self.size_min} is needed.')
        self.logx = bool(logx) and not np.any(x <= 0)
        self.logy = bool(logy)
        self.negativey = False
        if self.logy:
            neg, zero, pos = np.any(y < 0), np.any(y == 0), np.any(y > 0)
            if zero or (neg and pos):
                self.logy = False
            elif neg:
                self.negativey = True
                y = -y
        if self.logx:
            x = np.log(x)
        if self.logy:
            y = np.log(y)
        x, y = remove_doppelgängers(x, y)
        self.x = np.exp(x) if self.logx else x.copy()
        self.y = (1 - 2*self.negativey)*np.exp(y) if self.logy else y.copy()
        self.xmin, self.xmax = float(x[0]), float(x[-1])
        abs_tol = 1e-9*(self.xmax - self.xmin) + machine_ϵ
        self.abs_tol_min = abs_tol + 0.5*(x[1] - self.xmin)
        self.abs_tol_max = abs_tol + 0.5*(self.xmax - x[-2])
        # 'natural' boundary conditions: GSL's cspline, and what the reference's pure-Python
        # mode asks SciPy for (integration.py:157-167)
        self.spline = scipy.interpolate.CubicSpline(x.copy(), y.copy(), bc_type='natural')

    def in_interval(self, x, action='interpolate to'):
        if x < self.xmin:
            if x > self.xmin - self.abs_tol_min:
                return self.xmin
        elif x > self.xmax:
            if x < self.xmax + self.abs_tol_max:
                return self.xmax
        else:
            return x
        raise ConceptGPUError(
            f'Spline "{self.name}": Could not {action} {x} because it is outside the tabulated '
            f'interval [{self.xmin}, {self.xmax}]')

    def eval(self, x_in):
        x = log(x_in) if self.logx else x_in
        x = self.in_interval(x, 'interpolate to')
        y = float(self.spline(x))
        if self.logy:
            y = float(exp(y))
            if self.negativey:
                y *= -1
        return y

    def eval_deriv(self, x_in):
        x = log(x_in) if self.logx else x_in
        x = self.in_interval(x, 'differentiate at')
        d = float(self.spline(x, 1))
        if self.logx and self.logy:
            d *= self.eval(x_in)/x_in
        elif self.logx:
            d /= x_in
        elif self.logy:
            d *= self.eval(x_in)
        if self.negativey:
            d *= -1
        return d

    def integrate(self, a, b):
        if self.logx or self.logy:
            raise ConceptGPUError(f'Spline "{self.name}": Spline integration not possible for '
                                  'logged data')
        a = self.in_interval(a, 'integrate from')
        b = self.in_interval(b, 'integrate to')
        sign_flip = False
        if a > b:
            sign_flip = True
            a, b = b, a
        s = float(self.spline.integrate(a, b))
        if sign_flip:
            s *= -1
        if self.negativey:
            s *= -1
        return s


class Cosmology:
    """The reference's module-level clock state — universals.t / universals.a, the
    temporal_splines and the per-integrand splines of scalefactor_integral — as one object.

    `params`: commons.Params (H0, Ωb, Ωcdm, a_begin, t_begin, enable_Hubble).  init_time()
    must be called once before the clock is used (integration.py:864)."""

    def __init__(self, params=None):
        p = params or commons.params
        self.params = p
        self.enable_Hubble = bool(p.enable_Hubble)
        self.H0 = float(p.H0)
        self.Ωm = float(p.Ωb) + float(p.Ωcdm)   # commons.py:4464 (no decaying / warm matter)
        self.a_t = self.t_a = self.a_H = None
        self.spline_t_integrands = {}
        self.t = self.a = None
        self.t_begin = self.a_begin = None

    # -- integration.py:570-596 ------------------------------------------------------------
    def hubble(self, a=-1):
        if not self.enable_Hubble:
            return 0
        if a == -1:
            a = self.a
        ΩΛ = 1 - self.Ωm
        return self.H0*sqrt(self.Ωm*np.power(a, -3) + ΩΛ)

    # -- integration.py:598-640 ------------------------------------------------------------
    def scale_factor(self, t=-1):
        if not self.enable_Hubble:
            return 1
        if t == -1:
            t = self.t
        if self.t_a is None:
            raise ConceptGPUError('The function a(t) has not been tabulated. Have you called '
                                  'init_time?')
        return self.t_a.eval(t)

    def cosmic_time(self, a=-1):
        if not self.enable_Hubble:
            raise ConceptGPUError(
                'The cosmic_time() function was called. A mapping from a to t is only '
                'meaningful when Hubble expansion is enabled.')
        if a == -1:
            a = self.a
        if self.a_t is None:
            raise ConceptGPUError('The function t(a) has not been tabulated. Have you called '
                                  'init_time?')
        return self.a_t.eval(a)

    def ȧ(self, a=-1):
        if not self.enable_Hubble:
            return 0
        if a == -1:
            a = self.a
        return a*self.hubble(a)

    # -- integration.py:1043-1100 ----------------------------------------------------------
    def solve_matterΛ_background(self, a_today=1):
        """a(t) of the flat matter + Λ universe: d ln a / d ln t = t H(a) integrated with
        SciPy's DOP853 at rtol 1e-12 from a = 1e-14 (matter-dominated start t = 2/(3H)),
        tabulated on int(ln(a_today/1e-14)/7e-3) points equidistant in ln t."""
        import scipy.integrate
        a_begin_bg = 1e-14
        kwargs = dict(method='DOP853', rtol=1e-12, atol=0)
        t_begin_bg = 2/(3*self.hubble(a_begin_bg))
        log_a_today = log(a_today)

        def dloga_dlogt(logt, loga):
            t = exp(logt)
            a = exp(loga[0])
            return t*self.hubble(a)

        def event(logt, loga):
            return loga[0] - log_a_today
        event.terminal = True
        t_today = exp(scipy.integrate.solve_ivp(
            dloga_dlogt, (log(t_begin_bg), math.inf), np.asarray([log(a_begin_bg)]),
            events=event, **kwargs).t_events[0][0])
        n_bg = int(log(a_today/a_begin_bg)/7e-3)
        logt_values = np.linspace(log(t_begin_bg), log(t_today), n_bg)
        t_values = np.exp(logt_values)
        a_values = np.exp(scipy.integrate.solve_ivp(
            dloga_dlogt, (log(t_begin_bg), log(t_today)), [log(a_begin_bg)],
            t_eval=logt_values, **kwargs).y[0])
        t_values[0], t_values[-1] = t_begin_bg, t_today
        a_values[0], a_values[-1] = a_begin_bg, a_today
        H_values = np.asarray([self.hubble(a) for a in a_values])
        return a_values, t_values, H_values

    # -- integration.py:864-1001 -----------------------------------------------------------
    def init_time(self, reinitialize=False):
        if self.a_t is not None and not reinitialize:
            return
        p = self.params
        a_today = 1
        if self.enable_Hubble:
            a_values, t_values, H_values = self.solve_matterΛ_background(a_today)
            a_values[-1] = a_today
            H_values[-1] = self.H0
            self.a_t = Spline(a_values, t_values, 't(a)', logx=True, logy=True)
            self.t_a = Spline(t_values, a_values, 'a(t)', logx=True, logy=True)
            self.a_H = Spline(a_values, H_values, 'H(a)', logx=True, logy=True)
            if 'a_begin' in p.user:
                a_begin = float(p.a_begin)
                t_begin = self.cosmic_time(a_begin)
            elif 't_begin' in p.user:
                t_begin = float(p.t_begin)
                if t_begin == 0:
                    raise ConceptGPUError(
                        'You have specified t_begin = 0 while having Hubble expansion enabled. '
                        'Please specify some finite starting time or disable Hubble expansion.')
                a_begin = self.scale_factor(t_begin)
            else:
                raise ConceptGPUError(
                    'No initial scale factor (a_begin) or initial cosmic time (t_begin) '
                    'specified. A specification of one or the other is needed when '
                    'enable_Hubble is True.')
        else:
            t_begin = float(p.t_begin)
            a_begin = 1.0
        self.t_begin, self.a_begin = t_begin, a_begin
        self.t, self.a = t_begin, a_begin

    # -- integration.py:712-827 ------------------------------------------------------------
    # The integrands of the time loop as a table: name -> (number of components it names,
    # function of the background, the scale factor and the components' w_eff(a)).  Stable
    # matter has Γ = 0.
    INTEGRANDS = {
        '1': (0, lambda bg, a: 1.0),
        '': (0, lambda bg, a: 1.0),
        'a**2': (0, lambda bg, a: a**2),
        'a**(-1)': (0, lambda bg, a: 1/a),
        'a**(-2)': (0, lambda bg, a: 1/a**2),
        'ȧ/a': (0, lambda bg, a: bg.hubble(a)),
        'a**(-3*w_eff)': (1, lambda bg, a, w: a**(-3*w)),
        'a**(-3*(1+w_eff))': (1, lambda bg, a, w: a**(-3*(1 + w))),
        'a**(-3*w_eff-1)': (1, lambda bg, a, w: a**(-3*w - 1)),
        'a**(3*w_eff-2)': (1, lambda bg, a, w: a**(3*w - 2)),
        'a**(-3*w_eff)*Γ/H': (1, lambda bg, a, w: 0.0),
        'a**(-3*w_eff₀-3*w_eff₁-1)': (2, lambda bg, a, w0, w1: a**(-3*w0 - 3*w1 - 1)),
    }

    def scalefactor_integral(self, key, t_start, t_end, all_components):
        """ᔑ_t_start^t_end integrand(a(t)) dt; key = 'integrand' or ('integrand', name[, name]).
        The integrand is tabulated on the points of the a(t) table and integrated as a
        natural cubic spline in t (not logged), one spline per key kept for the run."""
        if t_start == t_end:
            return 0
        spline = self.spline_t_integrands.get(key)
        if spline is not None:
            return spline.integrate(t_start, t_end)
        integrand, names = (key, ()) if isinstance(key, str) else (key[0], tuple(key[1:]))
        entry = self.INTEGRANDS.get(integrand)
        if entry is None or entry[0] != len(names):
            known = sorted(k for k, (n, _) in self.INTEGRANDS.items() if n == len(names) and k)
            raise ConceptGPUError(
                f'scalefactor_integral: no integrand {integrand!r} over {len(names)} '
                f'component(s); the time loop knows {known}')
        by_name = {c.name: c for c in all_components}
        missing = [nm for nm in names if nm not in by_name]
        if missing:
            raise ConceptGPUError(f'scalefactor_integral: {key!r} names the component(s) '
                                  f'{missing}, which are not in the run')
        comps = [by_name[nm] for nm in names]
        func = entry[1]
        if self.enable_Hubble:
            a_tab, t_tab = self.a_t.x, self.a_t.y
        else:
            t_tab = np.linspace(t_start, t_end, Spline.size_min)
            a_tab = np.asarray([self.scale_factor(t) for t in t_tab], dtype=np.float64)
        tab = np.fromiter((func(self, a, *(c.w_eff(a=a) for c in comps))
                           for a in map(float, a_tab)), dtype=np.float64, count=len(a_tab))
        spline = Spline(t_tab, tab, integrand)
        if self.enable_Hubble:
            self.spline_t_integrands[key] = spline
        return spline.integrate(t_start, t_end)

    # -- main.py:998-1073 ------------------------------------------------------------------
    SINGLE = ('a**(-3*w_eff)', 'a**(-3*(1+w_eff))', 'a**(-3*w_eff-1)', 'a**(3*w_eff-2)',
              'a**(-3*w_eff)*Γ/H')
    PAIR = ('a**(-3*w_eff₀-3*w_eff₁-1)',)

    def integrand_keys(self, components):
        keys = ['1', 'a**2', 'a**(-1)', 'a**(-2)', 'ȧ/a']
        keys += [(integrand, c.name) for c in components for integrand in self.SINGLE]
        keys += [(integrand, c0.name, c1.name) for c0 in components for c1 in components
                 for integrand in self.PAIR]
        return keys

    def get_time_step_integrals(self, t_start, t_end, components, keys=None):
        """{integrand key: ᔑ_t_start^t_end integrand dt} for every integrand the time loop
        uses; an integrand naming a component outside `components` gives NaN (main.py:1049-
        1066).  `keys`: the key set fixed by the first call of the run (all components)."""
        out = {}
        for key in (keys if keys is not None else self.integrand_keys(components)):
            if isinstance(key, tuple) and any(all(c.name != nm for c in components)
                                              for nm in key[1:]):
                out[key] = math.nan
                continue
            out[key] = self.scalefactor_integral(key, t_start, t_end, components)
        return out
