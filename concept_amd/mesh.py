"""concept_amd.mesh — host handle of the GPU potential mesh (a cg_ctx).

Counterpart of the reference's global/upstream/downstream grid buffers and
FFTW slabs (mesh.py:492-710 interpolate_upstream, :3769-3866 get_fftw_slab,
communication.py:1666 get_buffer): one persistent mesh per (grid size,
device), living in HBM, reused across calls."""
import ctypes
import os

import numpy as np
import torch

from . import comm as _comm
from . import lib
from .lib import cg_params, check

_L = lib.raw()
_meshes = {}
_stage_buffers = {}  # (device, doubles) -> transpose staging buffer shared by the meshes of a size
_cm_plans = {}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class PotentialMesh:
    def __init__(self, gridsize, boxsize, nghosts=2, cell_centered=True, interp_order=2,
                 device=None, nprocs=1, rank=0, comm=None):
        """comm: a concept_amd.comm.Comm — the mesh then is one x-slab domain of its group
         and poisson_solve / fft_forward / poisson_backward / copy_modes_from /
        fold_ghosts / fill_ghosts are collective over it."""
        if comm is not None:
            nprocs, rank = comm.world, comm.rank
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device('cuda', device) if isinstance(device, int) else device
        self.gridsize = int(gridsize)
        self.boxsize = float(boxsize)
        self.nghosts = int(nghosts)
        self.cell_centered = int(bool(cell_centered))
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
        self.comm = comm
        self.dist = comm is not None and (comm.world > 1 or getattr(comm, 'force', False))
        self.four = self.stage = None
        if self.dist:
            self._init_dist()

    # -- x-slab domains: buffers of the transposing FFT and of the halos --------------
    def _init_dist(self):
        dev, n = self.device, self.transpose_doubles
        # Fourier-space slab of THIS mesh (persists between forward and inverse transform) and
        # the staging buffer of the transposes, shared by all meshes of this size
        self.four = torch.zeros(n, dtype=torch.float64, device=dev)
        key = (dev, n)
        if key not in _stage_buffers:
            _stage_buffers[key] = torch.empty(n, dtype=torch.float64, device=dev)
        self.stage = _stage_buffers[key]
        check(_L.cg_dist_bind_fourier(self._ctx, _ptr(self.four)))
        per, G = self.layer_doubles, self.ghost_layers
        if self.nprocs > 1:
            self.halo_s = torch.empty(G*per, dtype=torch.float64, device=dev)
            self.halo_r = torch.empty(G*per, dtype=torch.float64, device=dev)
            self.halo_s2 = torch.empty(G*per, dtype=torch.float64, device=dev)
            self.halo_r2 = torch.empty(G*per, dtype=torch.float64, device=dev)
        # The FFT transposes are exchanged in `pieces` layer ranges so that the transform of one
        # range overlaps the exchange of the previous one (CONCEPT_GPU_DIST_PIECES, 1 = one
        # all_to_all_single per transpose).  A piece should stay a large message: >= 8 layers.
        want = int(os.environ.get('CONCEPT_GPU_DIST_PIECES', '4'))
        npieces = max(1, min(want, self.nxl//8))
        if npieces > 1:
            # a piece should also fit the 256 MB infinity cache, like the chunks of the
            # single-GPU schedule (cg_fft.hip zy_chunk_layers): its y pass then reads what its
            # z pass wrote from the cache
            cache_layers = max(1, int(266e6//(per*8)))
            npieces = min(max(npieces, -(-self.nxl//cache_layers)), max(1, self.nxl//8))
        if npieces > 1 and not self.comm.stage:
            # every rank probes the asynchronous list form of all_to_all once, on a few bytes;
            # a transport that rejects it falls back to one all_to_all_single per transpose
            ok = getattr(self.comm, 'async_lists_ok', None)
            if ok is None:
                try:
                    a = torch.zeros(2*self.comm.world, dtype=torch.float64, device=dev)
                    b = torch.empty_like(a)
                    w = self.comm.all_to_all_layers(b, a, 2, 0, 1, async_op=True)
                    if w is not None:
                        w.wait()
                    torch.cuda.synchronize(dev)
                    ok = True
                except Exception as e:  # noqa: BLE001 (any backend error: do not pipeline)
                    print(f'[concept_amd] pipelined transposes disabled: {e}', flush=True)
                    ok = False
                self.comm.async_lists_ok = ok
            if not ok:
                npieces = 1
        edges = [self.nxl*k//npieces for k in range(npieces + 1)]
        self.pieces = [(a, b - a) for a, b in zip(edges[:-1], edges[1:])]

    def fold_ghosts(self, general=False):
        """communicate_ghosts(grid, '+=') after a deposit (mesh.py:609,
        communication.py:563-660).  CIC clouds of owned particles reach one layer above the
        slab; `general` (other orders, shifted lattices: the mesh was zeroed first) folds the
        full halo on both sides.  One periodic domain wraps by itself."""
        if not self.dist or self.nprocs == 1:
            return
        c, per, G, nxl = self.comm, self.layer_doubles, self.ghost_layers, self.nxl
        if not general:
            s, r = self.halo_s[:per], self.halo_r[:per]
            self.layers_read(nxl, 1, s)
            c.sendrecv(s, c.next, r, c.prev)
            self.layers_write(0, 1, r, add=True)
            return
        self.layers_read(nxl, G, self.halo_s)
        c.sendrecv(self.halo_s, c.next, self.halo_r, c.prev)
        self.layers_write(0, G, self.halo_r, add=True)
        self.layers_read(-G, G, self.halo_s)
        c.sendrecv(self.halo_s, c.prev, self.halo_r, c.next)
        self.layers_write(nxl - G, G, self.halo_r, add=True)

    def fold_ghosts_start(self, general=False):
        """fold_ghosts with the message in flight: returns finish() (adds what arrived).  The
        layers it adds to — the first owned ones — must not be read before finish()."""
        if not self.dist or self.nprocs == 1:
            return lambda: None
        if general:
            self.fold_ghosts(general=True)
            return lambda: None
        c, per, nxl = self.comm, self.layer_doubles, self.nxl
        s, r = self.halo_s[:per], self.halo_r[:per]
        self.layers_read(nxl, 1, s)
        done = c.sendrecv_start(s, c.next, r, c.prev)

        def finish():
            done()
            self.layers_write(0, 1, r, add=True)
        return finish

    def _fill_ghosts_start(self):
        """the two halo messages of fill_ghosts posted (the first and last G owned layers must
        be final); returns finish(), which writes the ghost layers"""
        c, G, nxl = self.comm, self.ghost_layers, self.nxl
        self.layers_read(0, G, self.halo_s)
        self.layers_read(nxl - G, G, self.halo_s2)
        d1 = c.sendrecv_start(self.halo_s, c.prev, self.halo_r, c.next)
        d2 = c.sendrecv_start(self.halo_s2, c.next, self.halo_r2, c.prev)

        def finish():
            d1()
            d2()
            self.layers_write(nxl, G, self.halo_r, add=False)
            self.layers_write(-G, G, self.halo_r2, add=False)
        return finish

    def fill_ghosts(self):
        """communicate_ghosts(grid, '=') (interactions.py:2303-2307, mesh.py:5026-5028): the
        G halo layers on both sides from the ring neighbours' owned layers."""
        if not self.dist or self.nprocs == 1:
            return
        c, G, nxl = self.comm, self.ghost_layers, self.nxl
        # my first G layers -> previous rank's upper ghosts [nxl, nxl+G)
        self.layers_read(0, G, self.halo_s)
        c.sendrecv(self.halo_s, c.prev, self.halo_r, c.next)
        self.layers_write(nxl, G, self.halo_r, add=False)
        # my last G layers -> next rank's lower ghosts [-G, 0)
        self.layers_read(nxl - G, G, self.halo_s)
        c.sendrecv(self.halo_s, c.next, self.halo_r, c.prev)
        self.layers_write(-G, G, self.halo_r, add=False)

    def _dist_forward(self, fold_finish=None):
        """z + y transform of the local layers, transpose (fft.c:240-257), pipelined in layer
        ranges; leaves the y-z transformed data, transposed, in self.four.  fold_finish: the
        pending ghost fold (fold_ghosts_start) — it lands in the first owned layer, so the
        range holding that layer is transformed last, behind the message."""
        c = self.comm
        if len(self.pieces) == 1:
            if fold_finish is not None:
                fold_finish()
            self.dist_fft_forward(self.stage)
            c.all_to_all(self.four, self.stage)
            return
        works = []
        order = self.pieces[1:] + self.pieces[:1] if fold_finish is not None else self.pieces
        for l0, nl in order:  # transform range k+1 while range k is on the links
            if l0 == 0 and fold_finish is not None:
                fold_finish()
            self.dist_fft_forward(self.stage, l0, nl)
            works.append(c.all_to_all_layers(self.four, self.stage, self.nxl, l0, nl,
                                             async_op=True))
        for w in works:
            if w is not None:
                w.wait()

    def _dist_backward(self, fill=False):
        """way back; fill: also exchange the potential's halo layers (fill_ghosts) — the two
        boundary ranges come first, their halo messages travel under the transforms of the
        inner ranges.  Returns the pending finish() of the halo (or None)."""
        c = self.comm
        fill = fill and self.nprocs > 1
        if len(self.pieces) == 1:
            c.all_to_all(self.stage, self.four)
            self.dist_fft_backward(self.stage)
            return self._fill_ghosts_start() if fill else None
        order = list(self.pieces)
        if fill and len(order) > 2:
            order = [order[0], order[-1]] + order[1:-1]
        works = [c.all_to_all_layers(self.stage, self.four, self.nxl, l0, nl, async_op=True)
                 for l0, nl in order]
        pending = None
        for k, ((l0, nl), w) in enumerate(zip(order, works)):  # inverse y + z of a range while
            if w is not None:                                  # the next ones arrive
                w.wait()
            self.dist_fft_backward(self.stage, l0, nl)
            if fill and k == min(1, len(order) - 1):
                pending = self._fill_ghosts_start()
        return pending

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

    def poisson_solve(self, deconv_order, C, long_range=False, E=0.0, fold_finish=None,
                      fill=False):
        """fold_finish: a pending fold_ghosts_start(); fill: end with fill_ghosts().  Both halo
        exchanges then overlap the transforms (x-slab domains; one domain needs neither)."""
        if self.dist:
            # A3..A8 with the transposes as all-to-alls; the k-space factor is fused into the
            # x pass, which runs on the transposed buffer
            self._dist_forward(fold_finish)
            self.dist_fft_xsolve(self.four, deconv_order, C, long_range, E)
            pending = self._dist_backward(fill)
            if pending is not None:
                pending()
            return
        if fold_finish is not None:
            fold_finish()
        check(_L.cg_poisson_solve(self._ctx, int(deconv_order), float(C), int(long_range),
                                  float(E)))

    def poisson_solve_timed(self, deconv_order, C, long_range=False, E=0.0):
        """poisson_solve + per-pass milliseconds (HIP events inside the library)."""
        ms = (ctypes.c_double*5)()
        check(_L.cg_poisson_solve_timed(self._ctx, int(deconv_order), float(C), int(long_range),
                                        float(E), ctypes.byref(ms)))
        return list(ms)

    def poisson_forward(self, deconv_order, C, long_range=False, E=0.0, apply_kernel=True):
        if self.dist:
            self._dist_forward()
            check(_L.cg_dist_fft_x(self._ctx, _ptr(self.four), 0))
            if apply_kernel:
                self.poisson_kernel(deconv_order, C, long_range, E)
            return
        check(_L.cg_poisson_forward(self._ctx, int(deconv_order), float(C), int(long_range),
                                    float(E), int(apply_kernel)))

    def poisson_kernel(self, deconv_order, C, long_range=False, E=0.0):
        check(_L.cg_poisson_kernel(self._ctx, int(deconv_order), float(C), int(long_range),
                                   float(E)))

    def poisson_backward(self):
        if self.dist:
            check(_L.cg_dist_fft_x(self._ctx, _ptr(self.four), 1))
            self._dist_backward()
            return
        check(_L.cg_poisson_backward(self._ctx))

    # -- general particle_mesh() pieces ------------
    def _check_fluid(self, *grids):
        n = self.nxl*self.gridsize**2  # a domain holds its own layers of a fluid grid
        for g in grids:
            if (g.dtype != torch.float64 or not g.is_contiguous() or g.device != self.device
                    or g.numel() != n):
                raise lib.ConceptGPUError(
                    f'fluid grids must be contiguous float64 tensors of {self.nxl} x '
                    f'{self.gridsize}^2 elements on {self.device}')

    def fluid_add(self, fluid, factor=1.0, operation='+='):
        """add_fluid_to_grid (mesh.py:1685-1753)"""
        self._check_fluid(fluid)
        check(_L.cg_fluid_add(self._ctx, _ptr(fluid), float(factor), int(operation == '+=')))

    def fft_forward(self):
        """slab_decompose + fft(slab, 'forward') (mesh.py:665-670)"""
        self.poisson_forward(0, 0.0, False, 0.0, apply_kernel=False)

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
            self.zero_fourier()
        sh = (ctypes.c_double*3)(*[float(x) for x in shift])
        if self.dist and source.gridsize != self.gridsize:
            return self._copy_modes_exchange(source, deconv_order, nlattice, sh, operation)
        check(_L.cg_copy_modes(self._ctx, source._ctx, int(deconv_order), int(nlattice), sh,
                               int(operation == '+=')))
        return self

    def zero_fourier(self):
        """The nullified slab copy_modes '=' starts from (get_fftw_slab(nullify=True),
        mesh.py:686-709)."""
        if self.dist:
            self.four.zero_()
        else:
            self.zero()

    def _copy_modes_exchange(self, source, deconv_order, nlattice, sh, operation):
        """copy_modes between different grid sizes over x-slab domains: the sub-slab exchange
        of mesh.py:1327-1468.  Row kj of the small cube is owned by rank (kj mod N)//(N/P),
        which differs between the two grids: the owner in `source` packs the small-cube part
        of its rows, an all-to-all-v carries them, the owner here applies them."""
        c, P = self.comm, self.nprocs
        N_from, N_onto = source.gridsize, self.gridsize
        key = (N_from, N_onto, P, c.rank, str(self.device))
        plan = _cm_plans.get(key)
        if plan is None:
            NS = min(N_from, N_onto)
            JBf, JBo = N_from//P, N_onto//P
            send = [[] for _ in range(P)]  # my rows of `source`, by destination
            recv = [[] for _ in range(P)]  # my rows of this mesh, by sender
            for kb in range(-(NS//2 - 1), NS//2):  # rows off the small grid's Nyquist plane
                bf, bo = kb % N_from, kb % N_onto
                src, dst = bf//JBf, bo//JBo
                if src == c.rank:
                    send[dst].append(bf % JBf)
                if dst == c.rank:
                    recv[src].append(bo % JBo)
            per_row = NS*(NS//2)*2  # doubles: complex[NS][NS/2]
            rows_s = torch.tensor([r for q in send for r in q], dtype=torch.int32,
                                  device=self.device)
            rows_r = torch.tensor([r for q in recv for r in q], dtype=torch.int32,
                                  device=self.device)
            plan = (NS, rows_s, rows_r, [len(q)*per_row for q in send],
                    [len(q)*per_row for q in recv], per_row)
            _cm_plans[key] = plan
        NS, rows_s, rows_r, send_counts, recv_counts, per_row = plan
        out = torch.empty(max(1, rows_s.numel()*per_row), dtype=torch.float64, device=self.device)
        inc = torch.empty(max(1, rows_r.numel()*per_row), dtype=torch.float64, device=self.device)
        if rows_s.numel():
            check(_L.cg_copy_modes_pack(source._ctx, NS, _ptr(rows_s), rows_s.numel(), _ptr(out)))
        c.all_to_all(inc[:sum(recv_counts)], out[:sum(send_counts)], recv_counts, send_counts)
        if rows_r.numel():
            check(_L.cg_copy_modes_unpack(self._ctx, source._ctx, NS, _ptr(rows_r),
                                          rows_r.numel(), _ptr(inc), int(deconv_order),
                                          int(nlattice), sh, int(operation == '+=')))
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
        """slab_downstream_subgroup[...] = slab_downstream (interactions.py:2242-2245): the
        Fourier-space slab — on x-slab domains that is the transposed buffer, not the mesh"""
        if self.dist:
            self.four.copy_(other.four)
            return
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

    # -- regions with gaps: kick + drift + scatter in one pass ---------------------------
    def region_capacity(self, n):
        """rows a particle array needs to hold n particles in predicted regions"""
        return int(_L.cg_region_capacity(self._ctx, int(n)))

    def new_region_table(self):
        """(start int32[8*ntiles+1], count int32[8*ntiles]) on the device"""
        return (torch.zeros(8*self.ntiles + 1, dtype=torch.int32, device=self.device),
                torch.zeros(8*self.ntiles, dtype=torch.int32, device=self.device))

    def predict_regions(self, start_in, count_in, start_out):
        check(_L.cg_predict_regions(self._ctx, _ptr(start_in),
                                    _ptr(count_in) if count_in is not None else None,
                                    _ptr(start_out)))

    def tile_order(self):
        """cg_tile_order_read: the heavy tiles the tile kernels run first (numpy uint32, in
        launch order), or None when they walk the tiles in the plain order"""
        out = np.empty(self.ntiles, dtype=np.uint32)
        nh = ctypes.c_int64(-1)
        check(_L.cg_tile_order_read(self._ctx, out.ctypes.data_as(ctypes.c_void_p), out.size,
                                    ctypes.byref(nh)))
        return out[:nh.value].copy() if nh.value >= 0 else None

    def deposit_regions(self, pos, start, count, contribution, accumulate=False):
        check(_L.cg_deposit_cic_regions(self._ctx, _ptr(pos), _ptr(start), _ptr(count),
                                        float(contribution), int(accumulate)))

    def gather_kick_drift_scatter(self, pos_in, mom_in, ids_in, start_in, count_in, pos_out,
                                  mom_out, ids_out, start_out, count_out, diff_order, factor,
                                  dt_over_mass, aux_in=None, aux_out=None):
        """cg_gather_kick_drift_scatter (see concept_gpu.h): nothing is written in place"""
        opt = lambda t: _ptr(t) if t is not None else None
        check(_L.cg_gather_kick_drift_scatter(
            self._ctx, _ptr(pos_in), _ptr(mom_in), _ptr(ids_in) if ids_in is not None else None,
            _ptr(start_in), _ptr(count_in) if count_in is not None else None, _ptr(pos_out),
            _ptr(mom_out), _ptr(ids_out) if ids_out is not None else None, _ptr(start_out),
            _ptr(count_out), int(diff_order), float(factor), float(dt_over_mass), opt(aux_in),
            opt(aux_out), int(pos_out.shape[0])))

    def check_errors(self):
        flags = self.error_flags()
        if flags & lib.CG_ERR_STALE_HISTOGRAM:
            raise lib.ConceptGPUError(
                'cg_drift_sort: the tile histogram prepared by the last gather-kick did not match '
                'the particles it sorted (momenta were changed in between without '
                'prepare_invalidate()); particles were dropped')
        if flags & lib.CG_ERR_NOT_IN_TILE:
            raise lib.ConceptGPUError(
                'cg_gather_kick_drift_scatter: a particle was not in the tile it is stored under '
                '(positions changed since the order was made) and got no kick — repeat the step '
                'on the exact path')
        if flags & lib.CG_ERR_ACTIVE_OVERFLOW:
            raise lib.ConceptGPUError(
                'cg_shortrange_sweep_cells_active: more receivers on active rungs than the bound '
                'it was given (the rung populations lag behind the rung array); the receivers '
                'beyond it got no short-range kick')
        if flags & lib.CG_ERR_BUCKET_OVERFLOW:
            raise lib.ConceptGPUError(
                'cg_gather_kick_drift_scatter: a (tile, bucket) outgrew its predicted region; '
                'particles were dropped — repeat the step on the exact path')

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

    def measure_momentum(self, mom):
        """(Σ mom², max |mom_i|²) of the local particles (cg_measure_momentum): the inputs of
        measure(component, 'v_rms' | 'v_max'), analysis.py:3902-3972.  Synchronises."""
        n = mom.shape[0]
        if n == 0:
            return 0.0, 0.0
        if getattr(self, '_measure_buf', None) is None:
            self._measure_buf = torch.empty(2048 + 2, dtype=torch.float64, device=self.device)
        buf = self._measure_buf
        check(_L.cg_measure_momentum(self._ctx, _ptr(mom), n, _ptr(buf[2048:]), _ptr(buf)))
        s, m = buf[2048:].tolist()
        return s, m

    def measure_momentum_regions(self, mom, start, count):
        """measure_momentum() for particles kept in tile regions with gaps"""
        if getattr(self, '_measure_buf', None) is None:
            self._measure_buf = torch.empty(2048 + 2, dtype=torch.float64, device=self.device)
        buf = self._measure_buf
        check(_L.cg_measure_momentum_regions(
            self._ctx, _ptr(mom), _ptr(start), _ptr(count) if count is not None else None,
            _ptr(buf[2048:]), _ptr(buf)))
        s, m = buf[2048:].tolist()
        return s, m

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

    def permute_rows(self, perm, pairs):
        """dst[q] = src[perm[q]] for every (src, dst) of `pairs` in one pass (cg_permute_rows):
        the columns that follow a sort."""
        k = len(pairs)
        if k == 0 or perm.numel() == 0:
            return
        n = perm.numel()
        for s_, d_ in pairs:
            if (s_.dtype != d_.dtype or s_.shape[1:] != d_.shape[1:] or s_.shape[0] < n
                    or d_.shape[0] < n or not s_.is_contiguous() or not d_.is_contiguous()):
                raise lib.ConceptGPUError('permute_rows: columns must be contiguous, of one '
                                          'type and shape, with a row per entry of perm')
        src = (ctypes.c_void_p*k)(*[s_.data_ptr() for s_, _ in pairs])
        dst = (ctypes.c_void_p*k)(*[d_.data_ptr() for _, d_ in pairs])
        rb = (ctypes.c_int*k)(*[s_.element_size()*int(np.prod(s_.shape[1:], dtype=np.int64))
                                for s_, _ in pairs])
        check(_L.cg_permute_rows(self._ctx, _ptr(perm), n, k, src, dst, rb))

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
    def shortrange_cells(self, pos, nt, tile_extent, rungs=None, sorted_jumps=False):
        """Cell list at half-tile granularity with the positions copied in cell order
        (cg_shortrange_cells): (order, offset, pos_sorted).  rungs = (rung int8, rung_jumped
        int8, lowest_active_rung) with lowest_active_rung > 0 makes the list of a sub-step
        (cg_shortrange_cells_rungs): the particles on active rungs first in every cell, and the
        tuple continues with (nact, rung_jumped_sorted, rung, lowest_active_rung) — what
        shortrange_sweep_cells() needs to take the active rows as its receivers."""
        n = self._check_particles(pos)
        order = torch.empty(max(n, 1), dtype=torch.int32, device=pos.device)
        offset = torch.empty(8*nt**3 + 1, dtype=torch.int32, device=pos.device)
        pos_sorted = torch.empty((max(n, 1), 3), dtype=torch.float64, device=pos.device)
        if rungs is not None and rungs[2] > 0:
            rung, rung_jumped, lowest = rungs
            self._check_rungs(n, rung, rung_jumped)
            nact = torch.empty(8*nt**3, dtype=torch.int32, device=pos.device)
            # sorted_jumps: the rows' jumped rung indices copied into list order too — worth
            # its scattered byte per particle (0.1 ms per list at 256^3) where many receivers
            # are active and the sweep goes in blocks (5.2 against 5.8 ms with half of them
            # active); the sweep by active receiver reads them through `order`
            rj_sorted = (torch.empty(max(n, 1), dtype=torch.int8, device=pos.device)
                         if sorted_jumps else None)
            check(_L.cg_shortrange_cells_rungs(
                self._ctx, _ptr(pos), n, int(nt), float(tile_extent), _ptr(rung),
                _ptr(rung_jumped), int(lowest), _ptr(order), _ptr(offset), _ptr(pos_sorted),
                _ptr(nact), _ptr(rj_sorted) if rj_sorted is not None else None))
            return order, offset, pos_sorted, nact, rj_sorted, rung, int(lowest)
        check(_L.cg_shortrange_cells(self._ctx, _ptr(pos), n, int(nt), float(tile_extent),
                                     _ptr(order), _ptr(offset), _ptr(pos_sorted)))
        return order, offset, pos_sorted

    def shortrange_sweep_cells(self, cells_r, dmom_r, cells_s, nt, table, r2_index_scaling,
                               r2_max, factor, rungs=None, n_active=None):
        """The sweep over half-tile cells; cells_* from shortrange_cells().  `rungs` =
        (factors[3*N_rungs-1] CUDA float64, rung int8, rung_jumped int8, lowest_active_rung)
        selects the adaptive-rung form (then `factor` is unused); a receivers' list made with
        the active particles first (for these rungs) is swept by its active rows — cell by
        cell instead of in blocks of tiles when `n_active` (an upper bound of the number of
        active receivers) is given."""
        n = self._check_particles(dmom_r)
        if table.dtype != torch.float64 or not table.is_cuda:
            raise lib.ConceptGPUError('short-range table must be a float64 CUDA tensor')
        order_r, off_r, pos_r = cells_r[:3]
        _, off_s, pos_s = cells_s[:3]
        if rungs is None:
            check(_L.cg_shortrange_sweep_cells(
                self._ctx, _ptr(pos_r), _ptr(order_r), _ptr(off_r), _ptr(dmom_r), _ptr(pos_s),
                _ptr(off_s), int(nt), _ptr(table), table.numel(), float(r2_index_scaling),
                float(r2_max), float(factor)))
            return
        factors, rung, rung_jumped, lowest = rungs
        self._check_rungs(n, rung, rung_jumped)
        if len(cells_r) > 3:
            nact, rj_sorted, rung_list, lowest_list = cells_r[3:]
            if rung_list.data_ptr() != rung.data_ptr() or lowest_list != int(lowest):
                raise lib.ConceptGPUError(
                    'shortrange_sweep_cells: the receivers\' list was made for other rungs')
            check(_L.cg_shortrange_sweep_cells_active(
                self._ctx, _ptr(pos_r), _ptr(order_r), _ptr(off_r), _ptr(nact),
                _ptr(rj_sorted) if rj_sorted is not None else None,
                _ptr(dmom_r), _ptr(pos_s), _ptr(off_s), int(nt), _ptr(table), table.numel(),
                float(r2_index_scaling), float(r2_max), _ptr(factors), _ptr(rung),
                _ptr(rung_jumped), int(lowest), -1 if n_active is None else int(n_active)))
            return
        check(_L.cg_shortrange_sweep_cells_rungs(
            self._ctx, _ptr(pos_r), _ptr(order_r), _ptr(off_r), _ptr(dmom_r), _ptr(pos_s),
            _ptr(off_s), int(nt), _ptr(table), table.numel(), float(r2_index_scaling),
            float(r2_max), _ptr(factors), _ptr(rung), _ptr(rung_jumped), int(lowest)))

    def shortrange_tiles(self, pos, nt, tile_extent, active=None):
        """The particles listed by tile (z fastest; cg_shortrange_tiles — the reference's
        `tiles[tile]` lists): (order, offset, pos_sorted).  active = (rung int8,
        lowest_active_rung) lists the particles on active rungs only; the tensors keep room for
        all n, offset[-1] says how many are listed."""
        n = self._check_particles(pos)
        order = torch.empty(max(n, 1), dtype=torch.int32, device=pos.device)
        offset = torch.empty(nt**3 + 1, dtype=torch.int32, device=pos.device)
        pos_sorted = torch.empty((max(n, 1), 3), dtype=torch.float64, device=pos.device)
        rung, lowest = (None, 0) if active is None else active
        if rung is not None:
            self._check_rungs(n, rung)
        check(_L.cg_shortrange_tiles(self._ctx, _ptr(pos), n, int(nt), float(tile_extent),
                                     _ptr(rung) if rung is not None else None, int(lowest),
                                     _ptr(order), _ptr(offset), _ptr(pos_sorted)))
        return order, offset, pos_sorted

    def shortrange_stats(self, enable):
        """cg_shortrange_stats: switch the sweeps' counters on (True), or off (False) and return
        {'cells': (tests, hits, trips), 'dense': (tests, hits, trips)} of the sweeps since."""
        if enable:
            check(_L.cg_shortrange_stats(self._ctx, 1, None))
            return None
        out = (ctypes.c_uint64*8)()
        check(_L.cg_shortrange_stats(self._ctx, 0, out))
        return {'cells': tuple(int(v) for v in out[0:3]), 'dense': tuple(int(v) for v in out[3:6])}

    SHORTRANGE_SPARSE_MAX = 8
    # the sweep by active receiver is taken up to this fraction of the receivers on active rungs
    # (tools/sr_rung_cost.py, 256^3 / 512^3: 0.32 against 1.39 ms in blocks at 0.9 %, 2.5 against
    # 3.5 ms at 12 %, 4.9 against 4.6 ms at 25 %)
    SHORTRANGE_BY_CELL_MAX = 0.16

    def shortrange_sparse(self, pos_r, active, dmom_r, pos_s, table, r2_index_scaling, r2_max,
                          factor, rungs=None):
        """The short-range sums of the receivers in rows `active` (int64 CUDA tensor, at most
        SHORTRANGE_SPARSE_MAX of them) against all suppliers, without a cell list
        (cg_shortrange_sparse).  rungs: (factors, rung_jumped) or None with `factor`."""
        self._check_particles(pos_r, dmom_r)
        self._check_particles(pos_s)
        k = active.numel()
        if active.dtype != torch.int64 or not active.is_cuda:
            raise lib.ConceptGPUError('active rows must be an int64 CUDA tensor')
        factors, rung_jumped = rungs if rungs is not None else (None, None)
        check(_L.cg_shortrange_sparse(
            self._ctx, _ptr(pos_r), _ptr(active), int(k), _ptr(dmom_r), _ptr(pos_s),
            pos_s.shape[0], _ptr(table), table.numel(), float(r2_index_scaling), float(r2_max),
            float(factor), _ptr(factors) if factors is not None else None,
            _ptr(rung_jumped) if rung_jumped is not None else None))

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

    def substep_begin(self, pos, mom, dmom, rung, rung_jumped, dt_over_mass, flag, lowest_active_rung,
                      integrals_1, rf_up, rf_down, N_rungs, any_out, counts_after=None,
                      defer=False):
        """cg_substep_begin: drift (dt_over_mass not None), then flag_rung_jumps + nullify_Δ
        (flag) in one pass; integrals_1 a host array, any_out an int32 CUDA tensor read later.
        counts_after (int64 CUDA tensor [N_rungs]): the rung populations after the sub-step's
        jumps.  defer: the pass is left to the next shortrange_cells() on these positions (or
        to substep_flush(), or to any other call that touches particles)."""
        n = self._check_particles(pos)
        self._check_rungs(n, rung, rung_jumped)
        tab = (ctypes.c_double*(3*N_rungs - 1))(*[float(v) for v in integrals_1]) if flag else None
        check(_L.cg_substep_begin(
            self._ctx, _ptr(pos), _ptr(mom), _ptr(dmom) if dmom is not None else None, _ptr(rung),
            _ptr(rung_jumped), n, int(dt_over_mass is not None), float(dt_over_mass or 0.0),
            int(bool(flag)), int(lowest_active_rung), tab, float(rf_up), float(rf_down),
            int(N_rungs), _ptr(any_out),
            _ptr(counts_after) if counts_after is not None else None, int(bool(defer))))

    def substep_flush(self):
        check(_L.cg_substep_flush(self._ctx))

    def substep_end(self, mom, dmom, rung, rung_jumped, apply, lowest_active_rung,
                    conversion_factors, N_rungs, counts):
        """cg_substep_end: apply_Δmom + convert_Δmom_to_acc (apply), apply_rung_jumps and the
        rung populations in one pass; conversion_factors a host array, counts an int64 CUDA
        tensor [N_rungs] read later."""
        n = self._check_particles(mom)
        self._check_rungs(n, rung, rung_jumped)
        tab = ((ctypes.c_double*(3*N_rungs - 1))(*[float(v) for v in conversion_factors])
               if apply else None)
        check(_L.cg_substep_end(
            self._ctx, _ptr(mom), _ptr(dmom) if dmom is not None else None, _ptr(rung),
            _ptr(rung_jumped), n, int(bool(apply)), int(lowest_active_rung), tab, int(N_rungs),
            _ptr(counts) if counts is not None else None))

    def apply_rung_jumps(self, rung, rung_jumped, N_rungs):
        n = rung.numel()
        self._check_rungs(n, rung, rung_jumped)
        check(_L.cg_apply_rung_jumps(self._ctx, _ptr(rung), _ptr(rung_jumped), n, int(N_rungs)))

    def rung_populations(self, rung, N_rungs):
        """int64 tensor [N_rungs]: the particles on each rung (set_rungs_N, species.py:2560-2587)"""
        counts = torch.empty(int(N_rungs), dtype=torch.int64, device=self.device)
        check(_L.cg_rung_populations(self._ctx, _ptr(rung), rung.numel(), int(N_rungs),
                                     _ptr(counts)))
        return counts

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
        if self.dist:
            raise lib.ConceptGPUError('fetch_fourier(): single-domain debug fetch')
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
        # under an active domain decomposition (comm.init) every mesh is one x-slab domain
        m = _meshes[key] = PotentialMesh(gridsize, boxsize, nghosts, cell_centered, interp_order,
                                         device, comm=_comm.active())
    m.use_stream(torch.cuda.current_stream(m.device))
    return m


def free_meshes():
    for m in _meshes.values():
        m.close()
    _meshes.clear()
    _stage_buffers.clear()
    _cm_plans.clear()
