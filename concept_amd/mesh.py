"""concept_amd.mesh — host handle of the GPU potential mesh (a cg_ctx).

Counterpart of the reference's global/upstream/downstream grid buffers and
FFTW slabs (mesh.py:492-710 interpolate_upstream, :3769-3866 get_fftw_slab,
communication.py:1666 get_buffer): one persistent mesh per (grid size,
device), living in HBM, reused across calls."""
import ctypes

import numpy as np
import torch

from . import lib
from .lib import cg_params, check

_L = lib.raw()
_meshes = {}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class PotentialMesh:
    def __init__(self, gridsize, boxsize, nghosts=2, cell_centered=True, interp_order=2,
                 device=None, nprocs=1, rank=0):
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device('cuda', device) if isinstance(device, int) else device
        self.gridsize = int(gridsize)
        self.boxsize = float(boxsize)
        self.nghosts = int(nghosts)
        p = cg_params()
        p.boxsize = self.boxsize
        p.gridsize = self.gridsize
        p.nghosts = self.nghosts
        p.cell_centered = int(cell_centered)
        p.interp_order = int(interp_order)
        p.device = self.device.index or 0
        p.nprocs, p.rank = int(nprocs), int(rank)
        p.subdiv[0], p.subdiv[1], p.subdiv[2] = int(nprocs), 1, 1  # x-slab domains
        self._ctx = ctypes.c_void_p()
        check(_L.cg_create(ctypes.byref(p), ctypes.byref(self._ctx)))
        self.use_stream(torch.cuda.current_stream(self.device))
        info = (ctypes.c_int64*3)()
        check(_L.cg_tile_info(self._ctx, ctypes.byref(info)))
        self.tile_extent, self.tiles_per_dim, self.table_entries = (
            int(info[0]), int(info[1]), int(info[2]))
        linfo = (ctypes.c_int64*6)()
        check(_L.cg_local_info(self._ctx, ctypes.byref(linfo)))
        (self.x0, self.nxl, self.ghost_layers, _, self.pad, self.transpose_doubles) = (
            int(v) for v in linfo)
        self.nprocs, self.rank = int(nprocs), int(rank)
        self.ntiles = (self.table_entries - 1)//8
        self.layer_doubles = int(_L.cg_layer_doubles(self._ctx))  # unit of layers_read/write

    def close(self):
        if self._ctx:
            _L.cg_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing -----------------------------------------------------------
    def use_stream(self, stream):
        self._stream = stream
        check(_L.cg_set_stream(self._ctx, ctypes.c_void_p(stream.cuda_stream)))

    def synchronize(self):
        check(_L.cg_synchronize(self._ctx))

    @property
    def device_bytes(self):
        return int(_L.cg_device_bytes(self._ctx))

    def error_flags(self):
        """Sticky error bits set by kernels since the last call (synchronises; clears)."""
        flags = ctypes.c_uint32(0)
        check(_L.cg_error_flags(self._ctx, ctypes.byref(flags)))
        return int(flags.value)

    def check_errors(self):
        flags = self.error_flags()
        if flags & lib.CG_ERR_STALE_HISTOGRAM:
            raise lib.ConceptGPUError(
                'cg_drift_sort: the tile histogram prepared by the last gather-kick did not match '
                'the particles it sorted (momenta were changed in between without '
                'prepare_invalidate()); particles were dropped')

    def prepare_invalidate(self):
        """Momenta were changed outside this mesh: forget the prepared drift histogram."""
        check(_L.cg_prepare_invalidate(self._ctx))

    @staticmethod
    def _check_particles(*tensors):
        n = None
        for t in tensors:
            if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
                raise lib.ConceptGPUError('particle arrays must be contiguous float64 CUDA tensors')
            if t.dim() != 2 or t.shape[1] != 3:
                raise lib.ConceptGPUError('particle arrays must have shape (N, 3) (AoS xyz)')
            n = t.shape[0] if n is None else n
            if t.shape[0] != n:
                raise lib.ConceptGPUError('particle arrays differ in length')
        return n

    # -- the path -----------------------------------------------------------
    def zero(self):
        check(_L.cg_mesh_zero(self._ctx))

    def deposit(self, pos, contribution):
        n = self._check_particles(pos)
        check(_L.cg_deposit_cic(self._ctx, _ptr(pos), n, float(contribution)))

    def deposit_tiled(self, pos, tile_offset, contribution, accumulate=False):
        """`pos` must be in exact tile order (sort_particles on this array)."""
        n = self._check_particles(pos)
        self._check_table(tile_offset)
        check(_L.cg_deposit_cic_tiled(self._ctx, _ptr(pos), n, _ptr(tile_offset),
                                      float(contribution), int(accumulate)))

    def poisson_solve(self, deconv_order, C, long_range=False, E=0.0):
        check(_L.cg_poisson_solve(self._ctx, int(deconv_order), float(C), int(long_range),
                                  float(E)))

    def poisson_solve_timed(self, deconv_order, C, long_range=False, E=0.0):
        """poisson_solve + per-pass milliseconds (HIP events inside the library)."""
        ms = (ctypes.c_double*5)()
        check(_L.cg_poisson_solve_timed(self._ctx, int(deconv_order), float(C), int(long_range),
                                        float(E), ctypes.byref(ms)))
        return list(ms)

    def poisson_forward(self, deconv_order, C, long_range=False, E=0.0, apply_kernel=True):
        check(_L.cg_poisson_forward(self._ctx, int(deconv_order), float(C), int(long_range),
                                    float(E), int(apply_kernel)))

    def poisson_kernel(self, deconv_order, C, long_range=False, E=0.0):
        check(_L.cg_poisson_kernel(self._ctx, int(deconv_order), float(C), int(long_range),
                                   float(E)))

    def poisson_backward(self):
        check(_L.cg_poisson_backward(self._ctx))

    # -- general particle_mesh() pieces ------------
    def _check_fluid(self, *grids):
        n = self.gridsize**3
        for g in grids:
            if (g.dtype != torch.float64 or not g.is_contiguous() or g.device != self.device
                    or g.numel() != n):
                raise lib.ConceptGPUError(
                    f'fluid grids must be contiguous float64 tensors of {self.gridsize}^3 '
                    f'elements on {self.device}')

    def fluid_add(self, fluid, factor=1.0, operation='+='):
        """add_fluid_to_grid (mesh.py:1685-1753)"""
        self._check_fluid(fluid)
        check(_L.cg_fluid_add(self._ctx, _ptr(fluid), float(factor), int(operation == '+=')))

    def fft_forward(self):
        """slab_decompose + fft(slab, 'forward') (mesh.py:665-670)"""
        check(_L.cg_poisson_forward(self._ctx, 0, 0.0, 0, 0.0, 0))

    def nullify_nyquist(self):
        """nullify_modes(slab, 'nyquist') (mesh.py:3591-3622)"""
        check(_L.cg_fourier_nullify_nyquist(self._ctx))

    def fourier_operate(self, deconv_order=0, nlattice=1, shift=(0.0, 0.0, 0.0), diff_dim=-1,
                        source=None, operation='='):
        """fourier_operate (mesh.py:3327-3400) when source is None, else the equal-size
        copy_modes(source, self, ...) (mesh.py:1038-1092); their early exits included."""
        shifted = tuple(shift) != (0, 0, 0)
        if source is None or source is self:
            source = self
            if operation != '=':
                raise lib.ConceptGPUError('fourier_operate(): in place means operation "="')
            if deconv_order == 0 and nlattice == 1 and not shifted and diff_dim == -1:
                return self  # mesh.py:3339-3344
        sh = (ctypes.c_double*3)(*[float(x) for x in shift])
        check(_L.cg_fourier_operate(self._ctx, source._ctx, int(deconv_order), int(nlattice),
                                    sh, int(diff_dim), int(operation == '+=')))
        return self

    def copy_modes_from(self, source, deconv_order=0, nlattice=1, shift=(0.0, 0.0, 0.0),
                        operation='='):
        """copy_modes(source, self, ...) (mesh.py:1018-1326) for any two grid sizes.  With
        operation '=' and different sizes this mesh is nullified first (mesh.py:686-709)."""
        if source.gridsize != self.gridsize and operation == '=':
            self.zero()
        sh = (ctypes.c_double*3)(*[float(x) for x in shift])
        check(_L.cg_copy_modes(self._ctx, source._ctx, int(deconv_order), int(nlattice), sh,
                               int(operation == '+=')))
        return self

    def deposit_general(self, pos, contribution, order=2, shift=(0.0, 0.0, 0.0)):
        """interpolate_particles of order 1..4 with a lattice shift (mesh.py:1512-1636)"""
        n = self._check_particles(pos)
        sh = (ctypes.c_double*3)(*[float(x) for x in shift])
        check(_L.cg_deposit(self._ctx, _ptr(pos), n, float(contribution), int(order), sh))

    def gather_scalar(self, pos, mom, dim, order, shift, factor):
        """interpolate_domaingrid_to_particles (mesh.py:376-459): this mesh holds one
        force component"""
        n = self._check_particles(pos, mom)
        sh = (ctypes.c_double*3)(*[float(x) for x in shift])
        check(_L.cg_gather_scalar(self._ctx, _ptr(pos), _ptr(mom), n, int(dim), int(order), sh,
                                  float(factor)))

    def diff_from(self, source, dim, diff_order):
        """diff_domaingrid (mesh.py:4874-5030) of source's real-space mesh into this one"""
        check(_L.cg_mesh_diff(self._ctx, source._ctx, int(dim), int(diff_order)))

    def ewald_tabulate(self, gridsize):
        """ewald.tabulate() (ewald.py:226-231) -> (g, g, g, 3) tensor in HBM"""
        grid = torch.empty((gridsize, gridsize, gridsize, 3), dtype=torch.float64,
                           device=self.device)
        check(_L.cg_ewald_tabulate(self._ctx, int(gridsize), _ptr(grid)))
        return grid

    def pp_kick(self, pos_r, dmom_r, pos_s, same, ewald_grid, softening, kernel, factor,
                rungs=None):
        """gravity_pairwise / gravity_pairwise_nonperiodic (gravity.py:121-206, 491-560);
        rungs = (factors, rung_indices, rung_indices_jumped, lowest_active_rung) or None"""
        n_r = self._check_particles(pos_r, dmom_r)
        n_s = self._check_particles(pos_s)
        kernels = {'none': 0, 'plummer': 1, 'spline': 2}
        if kernel not in kernels:
            raise lib.ConceptGPUError(f'Softening kernel "{kernel}" not understood')
        gs = 0 if ewald_grid is None else int(ewald_grid.shape[0])
        eg = None if ewald_grid is None else _ptr(ewald_grid)
        if rungs is None:
            f = r = rj = None
            low = 0
        else:
            factors, rung, rung_jumped, low = rungs
            self._check_rungs(n_r, rung, rung_jumped)
            f, r, rj = _ptr(factors), _ptr(rung), _ptr(rung_jumped)
        check(_L.cg_pp_kick(self._ctx, _ptr(pos_r), n_r, _ptr(dmom_r), _ptr(pos_s), n_s,
                            int(bool(same)), eg, gs, float(softening), kernels[kernel],
                            float(factor), f, r, rj, int(low)))

    def copy_from(self, other):
        check(_L.cg_mesh_copy(self._ctx, other._ctx))

    def fluid_kick(self, J_dim, rho, P, dim, diff_order, minus_dt, inv_c2):
        """diff_domaingrid + the fluid branch of apply_particle_mesh_force
        (interactions.py:2388-2401)"""
        self._check_fluid(J_dim, rho, P)
        check(_L.cg_fluid_kick(self._ctx, _ptr(J_dim), _ptr(rho), _ptr(P), int(dim),
                               int(diff_order), float(minus_dt), float(inv_c2)))

    def gather_kick(self, pos, mom, diff_order, factor):
        n = self._check_particles(pos, mom)
        check(_L.cg_gather_kick(self._ctx, _ptr(pos), _ptr(mom), n, int(diff_order),
                                float(factor)))

    def gather_kick_tiled(self, pos, mom, tile_offset, diff_order, factor):
        n = self._check_particles(pos, mom)
        self._check_table(tile_offset)
        check(_L.cg_gather_kick_tiled(self._ctx, _ptr(pos), _ptr(mom), n, _ptr(tile_offset),
                                      int(diff_order), float(factor)))

    def gather_kick_tiled_prepare(self, pos, mom, tile_offset, diff_order, factor,
                                  next_dt_over_mass):
        """gather_kick_tiled + tile histogram of the next drift (see concept_gpu.h)."""
        n = self._check_particles(pos, mom)
        self._check_table(tile_offset)
        check(_L.cg_gather_kick_tiled_prepare(self._ctx, _ptr(pos), _ptr(mom), n,
                                              _ptr(tile_offset), int(diff_order), float(factor),
                                              float(next_dt_over_mass)))

    def new_tile_table(self):
        """uint32[8*ntiles + 1] on the device (stored as int32 bits): first particle
        of each (tile, bucket), see include/concept_gpu.h cg_tile_info."""
        return torch.zeros(self.table_entries, dtype=torch.int32, device=self.device)

    def _check_table(self, t):
        if t.dtype != torch.int32 or not t.is_cuda or t.numel() != self.table_entries:
            raise lib.ConceptGPUError(
                f'tile table must be a CUDA int32 tensor of {self.table_entries} entries')

    def drift(self, pos, mom, dt_over_mass):
        n = self._check_particles(pos, mom)
        check(_L.cg_drift(self._ctx, _ptr(pos), _ptr(mom), n, float(dt_over_mass)))

    def sort_particles(self, pos, mom, ids, pos_out, mom_out, ids_out, tile_offset=None):
        n = self._check_particles(pos, mom, pos_out, mom_out)
        if tile_offset is None:
            tile_offset = self.new_tile_table()
        self._check_table(tile_offset)
        check(_L.cg_sort_particles(
            self._ctx, _ptr(pos), _ptr(mom), _ptr(ids) if ids is not None else None,
            _ptr(pos_out), _ptr(mom_out), _ptr(ids_out) if ids_out is not None else None, n,
            _ptr(tile_offset)))
        return tile_offset

    def drift_sort(self, pos, mom, ids, pos_out, mom_out, ids_out, dt_over_mass,
                   tile_offset=None):
        """Fused drift + tile sort: the outputs hold the drifted particles in tile order."""
        n = self._check_particles(pos, mom, pos_out, mom_out)
        if tile_offset is None:
            tile_offset = self.new_tile_table()
        self._check_table(tile_offset)
        check(_L.cg_drift_sort(
            self._ctx, _ptr(pos), _ptr(mom), _ptr(ids) if ids is not None else None,
            _ptr(pos_out), _ptr(mom_out), _ptr(ids_out) if ids_out is not None else None, n,
            float(dt_over_mass), _ptr(tile_offset)))
        return tile_offset

    # -- debug / parity -----------------------------------------------------
    # -- P3M short range -----------------------------------------------------------
    def shortrange_build(self, pos, nt, tile_extent):
        n = self._check_particles(pos)
        order = torch.empty(max(n, 1), dtype=torch.int32, device=pos.device)
        offset = torch.empty(nt**3 + 1, dtype=torch.int32, device=pos.device)
        check(_L.cg_shortrange_build(self._ctx, _ptr(pos), n, int(nt), float(tile_extent),
                                     _ptr(order), _ptr(offset)))
        return order, offset

    def shortrange_sweep(self, pos_r, cells_r, dmom_r, pos_s, cells_s, nt, same, table,
                         r2_index_scaling, r2_max, factor, rungs=None):
        """`rungs` = (factors[3*N_rungs-1] CUDA float64, rung int8, rung_jumped int8,
        lowest_active_rung) selects the adaptive-rung form (then `factor` is unused)."""
        n = self._check_particles(pos_r, dmom_r)
        self._check_particles(pos_s)
        if table.dtype != torch.float64 or not table.is_cuda:
            raise lib.ConceptGPUError('short-range table must be a float64 CUDA tensor')
        if rungs is None:
            check(_L.cg_shortrange_sweep(
                self._ctx, _ptr(pos_r), _ptr(cells_r[0]), _ptr(cells_r[1]), _ptr(dmom_r),
                _ptr(pos_s), _ptr(cells_s[0]), _ptr(cells_s[1]), int(nt), int(same), _ptr(table),
                table.numel(), float(r2_index_scaling), float(r2_max), float(factor)))
            return
        factors, rung, rung_jumped, lowest = rungs
        self._check_rungs(n, rung, rung_jumped)
        check(_L.cg_shortrange_sweep_rungs(
            self._ctx, _ptr(pos_r), _ptr(cells_r[0]), _ptr(cells_r[1]), _ptr(dmom_r), _ptr(pos_s),
            _ptr(cells_s[0]), _ptr(cells_s[1]), int(nt), int(same), _ptr(table), table.numel(),
            float(r2_index_scaling), float(r2_max), _ptr(factors), _ptr(rung), _ptr(rung_jumped),
            int(lowest)))

    # -- A16: momentum buffers and adaptive rungs -------------------------------------
    @staticmethod
    def _check_rungs(n, *arrays):
        for t in arrays:
            if t.dtype != torch.int8 or not t.is_cuda or t.numel() != n:
                raise lib.ConceptGPUError('rung arrays must be int8 CUDA tensors of length N')

    def dmom_nullify(self, dmom, rung=None, lowest_active_rung=0):
        n = self._check_particles(dmom)
        check(_L.cg_dmom_nullify(self._ctx, _ptr(dmom), _ptr(rung) if rung is not None else None,
                                 n, int(lowest_active_rung)))

    def dmom_apply(self, mom, dmom, rung=None, lowest_active_rung=0):
        n = self._check_particles(mom, dmom)
        check(_L.cg_dmom_apply(self._ctx, _ptr(mom), _ptr(dmom),
                               _ptr(rung) if rung is not None else None, n,
                               int(lowest_active_rung)))

    def dmom_to_acc(self, dmom, rung, rung_jumped, lowest_active_rung, conversion_factors,
                    any_rung_jumps):
        n = self._check_particles(dmom)
        self._check_rungs(n, rung, rung_jumped)
        check(_L.cg_dmom_to_acc(self._ctx, _ptr(dmom), _ptr(rung), _ptr(rung_jumped), n,
                                int(lowest_active_rung), _ptr(conversion_factors),
                                int(bool(any_rung_jumps))))

    def assign_rungs(self, acc, rung, rung_jumped, rung_factor, N_rungs):
        n = self._check_particles(acc)
        self._check_rungs(n, rung, rung_jumped)
        check(_L.cg_assign_rungs(self._ctx, _ptr(acc), _ptr(rung), _ptr(rung_jumped), n,
                                 float(rung_factor), int(N_rungs)))

    def flag_rung_jumps(self, acc, rung, rung_jumped, lowest_active_rung, integrals_1, rf_up,
                        rf_down, N_rungs):
        n = self._check_particles(acc)
        self._check_rungs(n, rung, rung_jumped)
        flag = torch.zeros(1, dtype=torch.int32, device=acc.device)
        check(_L.cg_flag_rung_jumps(self._ctx, _ptr(acc), _ptr(rung), _ptr(rung_jumped), n,
                                    int(lowest_active_rung), _ptr(integrals_1), float(rf_up),
                                    float(rf_down), int(N_rungs), _ptr(flag)))
        return bool(flag.item())

    def apply_rung_jumps(self, rung, rung_jumped, N_rungs):
        n = rung.numel()
        self._check_rungs(n, rung, rung_jumped)
        check(_L.cg_apply_rung_jumps(self._ctx, _ptr(rung), _ptr(rung_jumped), n, int(N_rungs)))

    # -- x-slab domains (multi-GPU) -------------------------------------------
    def layers_read(self, layer0, nlayers, dst):
        check(_L.cg_layers_read(self._ctx, int(layer0), int(nlayers), _ptr(dst)))

    def layers_write(self, layer0, nlayers, src, add=False):
        check(_L.cg_layers_write(self._ctx, int(layer0), int(nlayers), _ptr(src), int(add)))

    def dist_fft_forward(self, send_buf, layer0=None, nlayers=None):
        if layer0 is None:
            check(_L.cg_dist_fft_forward(self._ctx, _ptr(send_buf)))
        else:
            check(_L.cg_dist_fft_forward_layers(self._ctx, _ptr(send_buf), int(layer0),
                                                int(nlayers)))

    def dist_fft_xsolve(self, buf, deconv_order, C, long_range=False, E=0.0):
        check(_L.cg_dist_fft_xsolve(self._ctx, _ptr(buf), int(deconv_order), float(C),
                                    int(long_range), float(E)))

    def dist_fft_backward(self, recv_buf, layer0=None, nlayers=None):
        if layer0 is None:
            check(_L.cg_dist_fft_backward(self._ctx, _ptr(recv_buf)))
        else:
            check(_L.cg_dist_fft_backward_layers(self._ctx, _ptr(recv_buf), int(layer0),
                                                 int(nlayers)))

    def set_emigrant_list(self, idx, count):
        """gather_kick_tiled_prepare also lists (int64 row numbers into `idx`, their number into
        the one-element uint32/int32 tensor `count`) the particles its prepared drift takes out
        of this domain's slab; None, None switches the list off."""
        if idx is None:
            check(_L.cg_set_emigrant_list(self._ctx, None, None, 0))
        else:
            check(_L.cg_set_emigrant_list(self._ctx, _ptr(idx), _ptr(count), int(idx.numel())))

    def owner_rank(self, pos):
        n = self._check_particles(pos)
        out = torch.empty(n, dtype=torch.int32, device=pos.device)
        check(_L.cg_owner_rank(self._ctx, _ptr(pos), n, _ptr(out)))
        return out

    def owner_rank_drifted(self, pos, mom, dt_over_mass):
        n = self._check_particles(pos, mom)
        owner = torch.empty(n, dtype=torch.int32, device=pos.device)
        check(_L.cg_owner_rank_drifted(self._ctx, _ptr(pos), _ptr(mom), n, float(dt_over_mass),
                                       _ptr(owner)))
        return owner

    def prepare_rebind(self, pos, mom, add_pos=None, add_mom=None):
        n = self._check_particles(pos, mom)
        n_add = 0 if add_pos is None else self._check_particles(add_pos, add_mom)
        check(_L.cg_prepare_rebind(self._ctx, _ptr(pos), _ptr(mom), n,
                                   _ptr(add_pos) if n_add else None,
                                   _ptr(add_mom) if n_add else None, n_add))

    def fetch(self, which):
        N = self.gridsize
        out = np.empty((self.nxl, N, N + 2), dtype=np.float64)
        check(_L.cg_fetch(self._ctx, int(which), out.ctypes.data_as(ctypes.c_void_p), out.size))
        return out

    def fetch_real(self):
        return self.fetch(lib.CG_FETCH_MESH_REAL)

    def fetch_fourier(self):
        return self.fetch(lib.CG_FETCH_MESH_FOURIER)

    def cic_indices(self, pos, for_gather=False):
        n = self._check_particles(pos)
        idx = torch.empty((n, 3), dtype=torch.int64, device=pos.device)
        check(_L.cg_cic_indices(self._ctx, _ptr(pos), n, int(for_gather), _ptr(idx)))
        return idx


def get_mesh(gridsize, boxsize, nghosts=2, cell_centered=True, interp_order=2, device=None,
             role='global'):
    """Persistent mesh per configuration and role — the reference's named buffers
    ('slab_global', 'slab_updownstream', 'slab_updownstream_subgroup': mesh.py:593-617,
    interactions.py:2159-2161, 2242-2279)."""
    if device is None:
        device = torch.cuda.current_device()
    key = (int(gridsize), float(boxsize), int(nghosts), bool(cell_centered), int(interp_order),
           int(device) if isinstance(device, int) else device.index, role)
    m = _meshes.get(key)
    if m is None:
        m = _meshes[key] = PotentialMesh(gridsize, boxsize, nghosts, cell_centered, interp_order,
                                         device)
    m.use_stream(torch.cuda.current_stream(m.device))
    return m


def free_meshes():
    for m in _meshes.values():
        m.close()
    _meshes.clear()
