"""oracle/pp.py — direct summation with the Ewald correction.
TEST INFRASTRUCTURE ONLY (see oracle/oracle.py); C restatement in pp_oracle.c."""
import ctypes

import numpy as np

from . import oracle as O

KERNELS = {'none': 0, 'plummer': 1, 'spline': 2}


def _lib():
    L = O.lib()
    if not getattr(L, '_pp_ready', False):
        d, i64, vp = ctypes.c_double, ctypes.c_int64, ctypes.c_void_p
        L.orc_ewald_summation.argtypes = [d, d, d, vp]
        L.orc_ewald_tabulate.argtypes = [i64, vp]
        L.orc_ewald_lookup.argtypes = [vp, i64, d, d, d, d, d, vp]
        L.orc_pp_kick.argtypes = [vp, i64, vp, d, ctypes.c_int, vp, i64, d, ctypes.c_int, d, d]
        for f in (L.orc_ewald_summation, L.orc_ewald_tabulate, L.orc_ewald_lookup, L.orc_pp_kick):
            f.restype = None
        L._pp_ready = True
    return L


def ewald_tabulate(gridsize):
    """ewald.tabulate() (ewald.py:226-231): double[g][g][g][3] over one octant"""
    grid = np.empty((gridsize, gridsize, gridsize, 3), dtype=np.float64)
    _lib().orc_ewald_tabulate(gridsize, O._p(grid))
    return grid


def ewald_lookup(grid, x, y, z, boxsize):
    out = np.empty(3, dtype=np.float64)
    _lib().orc_ewald_lookup(O._p(grid), grid.shape[0], x, y, z, boxsize, O.machine_eps, O._p(out))
    return out


def pp_kick(pos, *, boxsize, softening, factor, periodic=True, ewald_grid=None,
            kernel='spline'):
    """Δmom of gravity('pp' | 'ppnonperiodic') for one component on itself, all particles
    on rung 0.  factor = G_Newton*mass**2*ᔑdt_rungs[...][0]."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    dmom = np.zeros_like(pos)
    if periodic and ewald_grid is None:
        raise ValueError('periodic direct summation needs the Ewald grid')
    g = np.ascontiguousarray(ewald_grid) if periodic else np.zeros(3)
    _lib().orc_pp_kick(O._p(pos), pos.shape[0], O._p(dmom), boxsize, int(periodic), O._p(g),
                       g.shape[0] if periodic else 0, softening, KERNELS[kernel], factor,
                       O.machine_eps)
    return dmom
