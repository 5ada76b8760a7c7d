"""INTEGRATION.md §1 executed literally: the reference-side stub — ctypes on libconcept_gpu.so
and libamdhip64.so (hipMalloc / hipMemcpy), no PyTorch, no concept_amd package — runs
cg_create -> cg_mesh_zero -> cg_deposit_cic -> cg_poisson_solve -> cg_gather_kick -> cg_drift
on the reference-generated golden `pm_n8_g16` and compares with the golden's momenta and
positions.  Run by tests/test_gpu_boundary_torchfree.py in a fresh interpreter."""
import ctypes
import math
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(REPO, 'concept_amd', 'libconcept_gpu.so'))
hip = ctypes.CDLL('libamdhip64.so')
L.cg_last_error.restype = ctypes.c_char_p


class cg_params(ctypes.Structure):   # include/concept_gpu.h
    _fields_ = [('boxsize', ctypes.c_double), ('gridsize', ctypes.c_int64),
                ('nghosts', ctypes.c_int32), ('cell_centered', ctypes.c_int32),
                ('interp_order', ctypes.c_int32), ('device', ctypes.c_int32),
                ('nprocs', ctypes.c_int32), ('rank', ctypes.c_int32),
                ('subdiv', ctypes.c_int32*3), ('reserved', ctypes.c_int32)]


def ok(rc):
    if rc:
        raise SystemExit('libconcept_gpu: ' + (L.cg_last_error() or b'?').decode())


def hip_ok(rc):
    if rc:
        raise SystemExit(f'hip error {rc}')


def upload(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    d = ctypes.c_void_p()
    hip_ok(hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(a.nbytes)))
    hip_ok(hip.hipMemcpy(d, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.nbytes), 1))
    return d


def download(d, shape):
    out = np.empty(shape, dtype=np.float64)
    hip_ok(hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), d, ctypes.c_size_t(out.nbytes), 2))
    return out


def main():
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'pm_n8_g16.npz'))
    boxsize, gridsize = float(g['boxsize']), int(g['gridsize'])
    mass, G = float(g['mass']), float(g['G_Newton'])
    pos, mom = g['pos_in'], g['mom_in']
    n = pos.shape[0]
    hip_ok(hip.hipSetDevice(0))
    p = cg_params(boxsize, gridsize, int(g['nghosts']), 1, 2, 0, 1, 0, (1, 1, 1), 0)
    c = ctypes.c_void_p()
    ok(L.cg_create(ctypes.byref(p), ctypes.byref(c)))
    d_pos, d_mom = upload(pos), upload(mom)
    # mesh.py:1550-1573
    contribution = float(g['dt_dens'])/float(g['dt_1'])*mass \
        * (float(gridsize)**(-3)*(gridsize/boxsize)**3)
    ok(L.cg_mesh_zero(c))
    ok(L.cg_deposit_cic(c, d_pos, ctypes.c_int64(n), ctypes.c_double(contribution)))
    C = -boxsize**2*G/math.pi
    ok(L.cg_poisson_solve(c, 4, ctypes.c_double(C), 0, ctypes.c_double(0.0)))
    ok(L.cg_gather_kick(c, d_pos, d_mom, ctypes.c_int64(n), int(g['diff_order']),
                        ctypes.c_double(mass*(-float(g['dt_kick'])))))
    ok(L.cg_synchronize(c))
    out = download(d_mom, mom.shape)
    kick = g['mom_after_long'] - mom
    scale = float(np.sqrt((kick**2).mean()))
    err = float(np.abs(out - g['mom_after_long']).max())
    assert err <= 1e-12*scale + 4e-16*np.abs(mom).max(), (err, scale)
    # Component.drift (species.py:2179-2199) on the golden's momenta: bit-exact
    hip_ok(hip.hipMemcpy(d_mom, np.ascontiguousarray(g['mom_after_long']).ctypes.data_as(
        ctypes.c_void_p), ctypes.c_size_t(mom.nbytes), 1))
    ok(L.cg_drift(c, d_pos, d_mom, ctypes.c_int64(n),
                  ctypes.c_double(float(g['drift_dt_over_mass']))))
    ok(L.cg_synchronize(c))
    assert np.array_equal(download(d_pos, pos.shape), g['drift_pos_out'])
    ok(L.cg_destroy(c))
    hip_ok(hip.hipFree(d_pos))
    hip_ok(hip.hipFree(d_mom))
    assert 'torch' not in sys.modules and 'concept_amd' not in sys.modules
    print(f'TORCHFREE-OK kick err {err:.3e} (rms kick {scale:.3e}), drift bit-exact')


if __name__ == '__main__':
    main()
