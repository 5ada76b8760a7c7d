"""ctypes binding of libconcept_gpu.so (include/concept_gpu.h).

The HIP library IS the product path: there is no CPU or PyTorch fallback.  If
the shared object is missing or a symbol is absent this module raises at
import time."""
import ctypes
import os

# PyTorch ships its own libamdhip64.so.7 / libhsa-runtime64 (same SONAMEs as /opt/rocm's).
# The particle arrays this library works on are torch allocations, so both must live in
# ONE HIP runtime: import torch first, then the dynamic loader resolves our NEEDED
# libamdhip64.so.7 / librocfft.so.0 to the copies torch already mapped.  (Loaded the
# other way round the process ends up with two runtimes and the second one finds no device.)
import torch  # noqa: F401  (must precede the CDLL below)

HERE = os.path.dirname(os.path.abspath(__file__))
# CONCEPT_GPU_LIB: load another build of the same library (kernel experiments in tools/)
LIB_PATH = os.environ.get('CONCEPT_GPU_LIB') or os.path.join(HERE, 'libconcept_gpu.so')


class ConceptGPUError(RuntimeError):
    pass


class cg_params(ctypes.Structure):
    _fields_ = [
        ('boxsize', ctypes.c_double),
        ('gridsize', ctypes.c_int64),
        ('nghosts', ctypes.c_int32),
        ('cell_centered', ctypes.c_int32),
        ('interp_order', ctypes.c_int32),
        ('device', ctypes.c_int32),
        ('nprocs', ctypes.c_int32),
        ('rank', ctypes.c_int32),
        ('subdiv', ctypes.c_int32*3),
        ('reserved', ctypes.c_int32),
    ]


CG_FETCH_MESH_REAL = 0
CG_FETCH_MESH_FOURIER = 1
CG_ERR_STALE_HISTOGRAM = 1
CG_ERR_BUCKET_OVERFLOW = 2
CG_ERR_NOT_IN_TILE = 4
CG_ERR_ACTIVE_OVERFLOW = 8

_vp, _i64, _dbl, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int

# name -> (restype, argtypes); every symbol include/concept_gpu.h declares
SYMBOLS = {
    'cg_last_error': (ctypes.c_char_p, []),
    'cg_abi_version': (_int, []),
    'cg_create': (_int, [ctypes.POINTER(cg_params), ctypes.POINTER(_vp)]),
    'cg_destroy': (_int, [_vp]),
    'cg_set_stream': (_int, [_vp, _vp]),
    'cg_synchronize': (_int, [_vp]),
    'cg_device_bytes': (_i64, [_vp]),
    'cg_error_flags': (_int, [_vp, ctypes.POINTER(ctypes.c_uint32)]),
    'cg_prepare_invalidate': (_int, [_vp]),
    'cg_set_emigrant_rows': (_int, [_vp, _vp, _vp, _i64]),
    'cg_emigrant_rows_dest': (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    'cg_region_insert': (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    'cg_region_capacity': (_i64, [_vp, _i64]),
    'cg_predict_regions': (_int, [_vp, _vp, _vp, _vp]),
    'cg_tile_order_read': (_int, [_vp, _vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    'cg_deposit_cic_regions': (_int, [_vp, _vp, _vp, _vp, _dbl, _int]),
    'cg_gather_kick_drift_scatter': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                            _int, _dbl, _dbl, _vp, _vp, _i64]),
    'cg_mesh_zero': (_int, [_vp]),
    'cg_deposit_cic': (_int, [_vp, _vp, _i64, _dbl]),
    'cg_deposit_cic_tiled': (_int, [_vp, _vp, _i64, _vp, _dbl, _int]),
    'cg_poisson_solve': (_int, [_vp, _int, _dbl, _int, _dbl]),
    'cg_poisson_solve_timed': (_int, [_vp, _int, _dbl, _int, _dbl, ctypes.POINTER(ctypes.c_double*5)]),
    'cg_poisson_forward': (_int, [_vp, _int, _dbl, _int, _dbl, _int]),
    'cg_poisson_backward': (_int, [_vp]),
    'cg_poisson_kernel': (_int, [_vp, _int, _dbl, _int, _dbl]),
    'cg_gather_kick': (_int, [_vp, _vp, _vp, _i64, _int, _dbl]),
    'cg_gather_kick_tiled': (_int, [_vp, _vp, _vp, _i64, _vp, _int, _dbl]),
    'cg_gather_kick_tiled_prepare': (_int, [_vp, _vp, _vp, _i64, _vp, _int, _dbl, _dbl]),
    'cg_drift': (_int, [_vp, _vp, _vp, _i64, _dbl]),
    'cg_measure_momentum': (_int, [_vp, _vp, _i64, _vp, _vp]),
    'cg_measure_momentum_regions': (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    'cg_sort_particles': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    'cg_drift_sort': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _vp]),
    'cg_tile_info': (_int, [_vp, ctypes.POINTER(ctypes.c_int64*3)]),
    'cg_shortrange_cells': (_int, [_vp, _vp, _i64, _i64, _dbl, _vp, _vp, _vp]),
    'cg_shortrange_sweep_cells': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _dbl,
                                         _dbl, _dbl]),
    'cg_shortrange_sweep_cells_rungs': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64,
                                               _dbl, _dbl, _vp, _vp, _vp, _int]),
    'cg_permute_rows': (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp]),
    'cg_substep_begin': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _dbl, _int, _int, _vp, _dbl,
                                _dbl, _int, _vp, _vp, _int]),
    'cg_substep_flush': (_int, [_vp]),
    'cg_substep_end': (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _vp, _int, _vp]),
    'cg_shortrange_cells_rungs': (_int, [_vp, _vp, _i64, _i64, _dbl, _vp, _vp, _int, _vp, _vp, _vp,
                                         _vp, _vp]),
    'cg_shortrange_sweep_cells_active': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                                _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _int, _i64]),
    'cg_shortrange_tiles': (_int, [_vp, _vp, _i64, _i64, _dbl, _vp, _int, _vp, _vp, _vp]),
    'cg_shortrange_stats': (_int, [_vp, _int, _vp]),
    'cg_set_momentum_sum': (_int, [_vp, _vp]),
    'cg_dmom_nullify': (_int, [_vp, _vp, _vp, _i64, _int]),
    'cg_dmom_apply': (_int, [_vp, _vp, _vp, _vp, _i64, _int]),
    'cg_dmom_to_acc': (_int, [_vp, _vp, _vp, _vp, _i64, _int, _vp, _int]),
    'cg_assign_rungs': (_int, [_vp, _vp, _vp, _vp, _i64, _dbl, _int]),
    'cg_flag_rung_jumps': (_int, [_vp, _vp, _vp, _vp, _i64, _int, _vp, _dbl, _dbl, _int, _vp]),
    'cg_apply_rung_jumps': (_int, [_vp, _vp, _vp, _i64, _int]),
    'cg_rung_populations': (_int, [_vp, _vp, _i64, _int, _vp]),
    'cg_shortrange_sparse': (_int, [_vp, _vp, _vp, _int, _vp, _vp, _i64, _vp, _i64, _dbl, _dbl, _dbl,
                             _vp, _vp]),
    'cg_local_info': (_int, [_vp, ctypes.POINTER(ctypes.c_int64*6)]),
    'cg_layers_read': (_int, [_vp, _i64, _i64, _vp]),
    'cg_layers_write': (_int, [_vp, _i64, _i64, _vp, _int]),
    'cg_dist_fft_forward': (_int, [_vp, _vp]),
    'cg_dist_fft_xsolve': (_int, [_vp, _vp, _int, _dbl, _int, _dbl]),
    'cg_dist_fft_backward': (_int, [_vp, _vp]),
    'cg_set_emigrant_list': (_int, [_vp, _vp, _vp, _i64]),
    'cg_dist_fft_forward_layers': (_int, [_vp, _vp, _i64, _i64]),
    'cg_dist_fft_backward_layers': (_int, [_vp, _vp, _i64, _i64]),
    'cg_layer_doubles': (_i64, [_vp]),
    'cg_owner_rank': (_int, [_vp, _vp, _i64, _vp]),
    'cg_dist_bind_fourier': (_int, [_vp, _vp]),
    'cg_dist_fft_x': (_int, [_vp, _vp, _int]),
    'cg_copy_modes_pack': (_int, [_vp, _i64, _vp, _i64, _vp]),
    'cg_copy_modes_unpack': (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _int, _int, _vp, _int]),
    'cg_emigrant_dest': (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _dbl, _vp, _vp]),
    'cg_owner_rank_drifted': (_int, [_vp, _vp, _vp, _i64, _dbl, _vp]),
    'cg_prepare_rebind': (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64]),
    'cg_fetch': (_int, [_vp, _int, _vp, _i64]),
    'cg_cic_indices': (_int, [_vp, _vp, _i64, _int, _vp]),
    'cg_fluid_add': (_int, [_vp, _vp, _dbl, _int]),
    'cg_fourier_nullify_nyquist': (_int, [_vp]),
    'cg_fourier_operate': (_int, [_vp, _vp, _int, _int, _vp, _int, _int]),
    'cg_copy_modes': (_int, [_vp, _vp, _int, _int, _vp, _int]),
    'cg_deposit': (_int, [_vp, _vp, _i64, _dbl, _int, _vp]),
    'cg_gather_scalar': (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _dbl]),
    'cg_mesh_diff': (_int, [_vp, _vp, _int, _int]),
    'cg_ewald_tabulate': (_int, [_vp, _int, _vp]),
    'cg_pp_kick': (_int, [_vp, _vp, _i64, _vp, _vp, _i64, _int, _vp, _int, _dbl, _int, _dbl, _vp, _vp,
                          _vp, _int]),
    'cg_mesh_copy': (_int, [_vp, _vp]),
    'cg_fluid_kick': (_int, [_vp, _vp, _vp, _vp, _int, _int, _dbl, _dbl]),
}

if not os.path.exists(LIB_PATH):
    raise ConceptGPUError(
        f'{LIB_PATH} is missing: build it with `python -m concept_amd.build` '
        '(hipcc --offload-arch=gfx950).  There is no fallback path.')
_lib = ctypes.CDLL(LIB_PATH)
for _name, (_res, _args) in SYMBOLS.items():
    _f = getattr(_lib, _name)  # AttributeError if the .so lacks a declared symbol
    _f.restype = _res
    _f.argtypes = _args


def last_error():
    return _lib.cg_last_error().decode('utf-8', 'replace')


def check(rc):
    if rc != 0:
        raise ConceptGPUError(last_error())


def raw():
    return _lib
