"""concept_amd.species — the GPU-resident Component.

Mirrors the slice of the reference's Component the gravity path uses
(species.py:852-1040 fields, :1911-1925 populate, :2010-2064 AoS double[3N]
pos/mom/Δmom, :2179-2199 drift, :2253-2266 apply_Δmom, :3717-3741 nullify_Δ,
:2598-2810 tile_sort), with the particle arrays living in HBM as torch CUDA
tensors of shape (N_local, 3), float64 — the reference's own "xyzxyz..." layout.

Under an active domain decomposition (concept_amd.comm.init) a Component is
collective like the reference's: N is the global particle number, N_local this
rank's share (species.py:955-996); the arrays hold the particles of this
rank's x-slab with spare capacity, Component.drift() ends with exchange()
(communication.py:135-517, the reference calls it right after every drift,
main.py:1406) and every per-particle array (pos, mom, Δmom, ids, rung indices)
travels with its particle.  Fluid grids are the rank's own layers
double[gridsize/P][gridsize][gridsize]."""
import collections

import numpy as np
import torch

from . import comm as _comm
from . import commons
from .lib import ConceptGPUError
from .mesh import get_mesh

PotentialGridsizes = collections.namedtuple('PotentialGridsizes', ('upstream', 'downstream'))


_BEGIN_WRITES = ('pos', 'Δmom', 'rung_indices_jumped')


def _store_column(name):
    """Component attribute backed by a column of its ParticleStore: the live rows"""
    def get(self):
        st = self.__dict__.get('_store')
        if st is None or name not in st.cols:
            return None
        # (a sub-step pass left to the next cell list, substep_begin(defer=True), runs before
        # anybody else looks at the rows it changes)
        if self.__dict__.get('_begin_args') is not None and name in _BEGIN_WRITES:
            self.flush_begin()
        return st.cols[name][:st.n]

    def set_(self, value):
        st = self._store
        if value is None:
            st.drop_column(name)
            return
        if name not in st.cols:
            st.add_column(name, dtype=value.dtype,
                          width=(value.shape[1] if value.dim() == 2 else None))
        st.cols[name][:st.n] = value
    return property(get, set_)


class Component:
    pos = _store_column('pos')
    mom = _store_column('mom')
    Δmom = _store_column('Δmom')
    ids = _store_column('ids')
    order = _store_column('order')
    rung_indices = _store_column('rung_indices')
    rung_indices_jumped = _store_column('rung_indices_jumped')

    def __init__(self, name, species, *, N=None, gridsize=None, mass=None, boltzmann_order=-1,
                 device=None, params=None):
        """Component(name, species, N=..., mass=...)                      particles
        Component(name, species, gridsize=..., boltzmann_order=1)        fluid
        (species.py:852-1040).  A fluid component here is what the gravity path needs of
        it: the grids ϱ, J[0..2] and 𝒫 as float64 tensors on the GPU (the reference's
        grid_noghosts, this rank's layers); its own evolution (fluid.py) stays with the
        caller."""
        self.params = p = params or commons.params
        if p is None:
            raise ConceptGPUError('no parameters loaded: call concept_amd.commons.load_params()')
        self.name = name.strip()
        self.species = species
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        self.comm = _comm.active()
        self.nprocs = self.comm.world if self.comm is not None else 1
        self.rank = self.comm.rank if self.comm is not None else 0
        self._store = None
        self._sub_dev = None   # (substep_begin/_end: device words read back by substep_finish)
        self._begin_args = None   # (a substep_begin() left to the next short-range cell list)
        self._sub_counted = False
        self.tile_mesh = None
        self.tiles_exact = False
        self.use_ids = False  # identifiers are the running row numbers until some are populated
        # host() returns the particles in the order they were populated in (an `order` column
        # travels with them).  keep_order = False gives that up where it costs — the streaming
        # time loop then moves no 64-bit column with particles that have no identifiers, as the
        # reference moves none (its particle order is whatever the last tile sort left)
        self.keep_order = True
        if (N is None) == (gridsize is None):
            raise ConceptGPUError(
                f'{self.name}: give N (particle component) or gridsize (fluid component)')
        if N is not None:
            if boltzmann_order != -1:
                raise ConceptGPUError(f'{self.name}: particle components have '
                                      'boltzmann_order = -1')
            self.representation = 'particles'
            self.N = int(N)
            self.mass = float(mass)
            self.softening_length = commons.softening_length(p, self.species, self.N)
        else:
            if boltzmann_order < 0:
                raise ConceptGPUError(f'{self.name}: fluid components need boltzmann_order >= 0')
            self.representation = 'fluid'
            self.boltzmann_order = int(boltzmann_order)
            self._init_fluid(int(gridsize))
        self._init_forces()
        if self.representation == 'particles':
            # one domain: the N particles live here from the start; several: populate()
            # (or populate_local) decides which of them are this rank's
            self._new_store(self.N if self.nprocs == 1 else 0)

    def _init_fluid(self, gridsize):
        if gridsize < 2 or gridsize % 2:
            raise ConceptGPUError(f'{self.name}: fluid grid size {gridsize} must be even and ≥ 2')
        if gridsize % (2*self.nprocs):
            raise ConceptGPUError(f'{self.name}: fluid grid size {gridsize} must be divisible '
                                  f'by 2*nprocs = {2*self.nprocs} (mesh.py:1898-1905)')
        self.gridsize = gridsize
        self.N = 0
        self.mass = -1.0  # species.py: fluids carry no particle mass
        self.nxl = gridsize//self.nprocs        # this rank's layers [x0, x0 + nxl)
        self.x0 = self.nxl*self.rank
        shape = (self.nxl, gridsize, gridsize)
        z = lambda: torch.zeros(shape, dtype=torch.float64, device=self.device)
        self.ϱ, self.𝒫 = z(), z()
        self.J = [z(), z(), z()]
        self.use_rungs = False

    def _new_store(self, n):
        """n zeroed local particles.  ids: the particles' identifiers (species.py:2040-2064; a
        snapshot's ID block); order: where each row stood when the arrays were populated —
        tile_sort and exchange() move both with the particles, host(original_order=True)
        undoes it through `order`."""
        from .distributed import ParticleStore
        dev = self.device
        z = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        base = torch.arange(n, dtype=torch.int64, device=dev)
        had_dmom = self._store is not None and 'Δmom' in self._store.cols
        self._store = ParticleStore(self._mesh(), z, z, None, slack=1.4,
                                    extra={'ids': base, 'order': base.clone()})
        if had_dmom:
            self._store.add_column('Δmom', dtype=torch.float64, width=3)
        if self.use_rungs:
            self._store.add_column('rung_indices', dtype=torch.int8)
            self._store.add_column('rung_indices_jumped', dtype=torch.int8)
        self.tile_mesh = None
        self.tiles_exact = False

    def to_regions(self, mesh):
        """The particle arrays in streaming form (distributed.RegionParticles: tile regions
        with gaps, for the fused kick + drift + sort of stepper.timeloop); the Component's own
        arrays are stale until from_regions()."""
        from .distributed import RegionParticles
        self.tile_sort(mesh)
        # identifiers that are the running row numbers equal `order`: one column travels
        return RegionParticles(self._store, drop_order=not self.use_ids,
                               drop_ids=not self.use_ids and not self.keep_order)

    def from_regions(self, rp, collective=True):
        """Take the particles back from their streaming form (pos, mom, ids, order; Δmom and
        rung columns start from zero as after populate()).  collective=False (unwinding from an
        exception on one rank): the exchange still pending from the last pass — a collective —
        is abandoned, its leavers with it."""
        from .distributed import ParticleStore
        if not collective:
            rp.pending = False
        cols = rp.columns()
        old = self._store
        if 'ids' not in cols:   # (keep_order = False: the rows are renumbered as they lie)
            first = 0
            if self.comm is not None and self.nprocs > 1 and collective:
                counts = self.comm.all_gather_ints([cols['pos'].shape[0]])[:, 0].tolist()
                first = int(sum(counts[:self.rank]))
            cols['ids'] = torch.arange(first, first + cols['pos'].shape[0], dtype=torch.int64,
                                       device=self.device)
        self._store = ParticleStore(old.mesh, cols['pos'], cols['mom'], None, slack=1.4,
                                    extra={'ids': cols['ids'],
                                           'order': cols['order'] if 'order' in cols
                                           else cols['ids'].clone()})
        for name in ('Δmom', 'rung_indices', 'rung_indices_jumped'):
            if name in old.cols:
                t = old.cols[name]
                self._store.add_column(name, dtype=t.dtype, width=(t.shape[1] if t.dim() > 1
                                                                  else None))
        self.tile_mesh = None
        self.tiles_exact = False

    N_local = property(lambda self: self._store.n if self._store is not None else 0)
    tile_table = property(lambda self: self._store.table
                          if self._store is not None and self.tile_mesh is not None else None)

    def _init_forces(self):
        p = self.params
        # forces first: use_rungs depends on them (species.py:1113-1117, 1444-1448)
        self.forces = dict(commons.is_selected(self, p.select_forces, accumulate=True) or {})
        if self.representation == 'particles':
            # short-range rungs (species.py:1443-1460): int8 rung index per particle
            self.N_rungs = p.N_rungs
            self.use_rungs = bool(p.N_rungs > 1 and ({'ppnonperiodic', 'pp', 'p3m'}
                                                     & set(self.forces.values())))
            self.lowest_active_rung = 0
            self.lowest_populated_rung = 0
            self.highest_populated_rung = 0
            self.rungs_N = [0]*p.N_rungs
            self.rungs_N[0] = self.N
        # potential grid sizes (species.py:1147-1203): component-level entries of
        # potential_options['gridsize'] first; fluids default to their own grid size,
        # particles to the global one, else cbrt(N) (2 cbrt(N) for p3m)
        own = commons.is_selected(
            self, {k: v for k, v in p.potential_options['gridsize'].items() if k != 'global'},
            accumulate=True, default={}) or {}
        self.potential_gridsizes = {}
        self.potential_differentiations = {}
        for force, method in self.forces.items():
            methods = [method] + (['pm'] if method == 'p3m' else [])
            for m in methods:
                g = own.get(force, {}).get(m)
                if g is None:
                    if self.representation == 'fluid':
                        g = self.gridsize
                    else:
                        g = p.potential_options['gridsize']['global'].get(force, {}).get(m, -1)
                pair = list(g) if isinstance(g, (tuple, list)) else [g, g]
                for i, v in enumerate(pair):
                    if v == -1:
                        if self.representation == 'fluid':
                            v = self.gridsize
                        else:
                            cb = int(round(self.N**(1/3)))
                            v = 2*cb if m == 'p3m' else cb
                    pair[i] = int(v)
                self.potential_gridsizes.setdefault(force, {})[m] = PotentialGridsizes(*pair)
                dd = commons.is_selected(self, p.potential_options['differentiation'])
                d = (dd or {}).get(force, {}).get(m)
                self.potential_differentiations.setdefault(force, {})[m] = d

    # -- data in / out ------------------------------------------------------
    def populate(self, data, var, multi_index=None):
        """populate(array, 'posx'|'posy'|'posz'|'momx'|...) (species.py:1911-1925);
        also accepts var='pos'/'mom' with an (N, 3) array and var='ids'.  On several
        domains every rank passes the GLOBAL array (N rows / the whole fluid grid) and
        keeps what it owns: 'pos' decides which rows those are (the x-slab holding their
        lower CIC cell), so it comes first; populate_local() takes per-rank data."""
        if self.representation == 'fluid':
            t = torch.as_tensor(np.ascontiguousarray(np.asarray(data, dtype=np.float64)))
            # populate(grid, 'ϱ') | populate(grid, 'J', dim) | populate(grid, '𝒫')
            # (species.py:1926-1990)
            target = {'ϱ': self.ϱ, 'rho': self.ϱ, '𝒫': self.𝒫, 'P': self.𝒫}.get(var)
            if var == 'J':
                if multi_index not in (0, 1, 2):
                    raise ConceptGPUError(f'{self.name}.populate(): J needs a dimension 0, 1 or 2')
                target = self.J[multi_index]
            if target is None:
                raise ConceptGPUError(f'{self.name}.populate(): unknown fluid variable "{var}"')
            if tuple(t.shape) != (self.gridsize,)*3:
                raise ConceptGPUError(
                    f'{self.name}.populate(): grid of shape {tuple(t.shape)} for grid size '
                    f'{self.gridsize}')
            target.copy_(t[self.x0:self.x0 + self.nxl])
            return
        if var == 'ids':
            t = torch.as_tensor(np.ascontiguousarray(np.asarray(data, dtype=np.int64)))
        else:
            t = torch.as_tensor(np.ascontiguousarray(np.asarray(data, dtype=np.float64)))
        if t.shape[0] != self.N:
            raise ConceptGPUError(f'{self.name}.populate(): {t.shape[0]} rows for N = {self.N}')
        if self.nprocs > 1:
            if var in ('posx', 'posy', 'posz'):
                # component-wise positions: ownership needs all three
                pend = self.__dict__.setdefault('_pending_pos', {})
                pend[var] = t
                if len(pend) < 3:
                    return
                t = torch.stack([pend['posx'], pend['posy'], pend['posz']], 1)
                self._pending_pos = {}
                var = 'pos'
            if var == 'pos':
                t = t.reshape(self.N, 3).to(self.device)
                owner = self._mesh().owner_rank(t.contiguous())
                rows = torch.nonzero(owner == self.rank).flatten()
                self._new_store(int(rows.numel()))
                self.pos.copy_(t[rows])
                self.order.copy_(rows)   # global row numbers: host() reassembles by them
                self.ids.copy_(rows)
                self._rows_known = True
                for data_, var_ in self.__dict__.pop('_pending_other', []):
                    self.populate(data_, var_)
                return
            if not getattr(self, '_rows_known', False):
                # the positions decide which rows are local: keep this until they are complete
                self.__dict__.setdefault('_pending_other', []).append((t, var))
                return
        # the caller's row i is the local row whose `order` entry is i: `order` travels with
        # the particles through tile_sort, exchange() and the streaming loop, so data populated
        # after any of them still lands on the right particles (fresh arrays: the identity)
        t = t.to(self.device)[self.order]
        if var.startswith('pos'):
            self.tile_mesh = None
            self.tiles_exact = False
            self._store.touch_mom()
        if var.startswith('mom'):
            self._store.touch_mom()
        if var in ('pos', 'mom'):
            getattr(self, var).copy_(t.reshape(-1, 3))
            return
        if var == 'ids':
            self.ids.copy_(t)
            self.use_ids = True   # (species.py:1253: identifiers given, not running numbers)
            return
        prefix, suffix = var[:-1], var[-1]
        getattr(self, prefix)[:, 'xyz'.index(suffix)] = t.to(self.device)

    def populate_local(self, pos, mom, ids=None, first_row=None):
        """This rank's particles as device tensors (any rows; exchange() then re-homes those
        that belong to other domains).  N_local may differ between ranks; N stays the global
        number given to the constructor.  first_row: the global row number of the first of
        these rows (a rank-wise read of a file: its start_local); default: the ranks' rows
        follow each other in rank order.  host() returns the particles in global row order."""
        n = pos.shape[0]
        self._new_store(n)
        self.pos.copy_(pos)
        self.mom.copy_(mom)
        if first_row is not None:
            first = int(first_row)
        elif self.comm is not None:
            counts = self.comm.all_gather_ints([n])[:, 0].tolist()
            first = sum(counts[:self.rank])
        else:
            first = 0
        base = torch.arange(first, first + n, dtype=torch.int64, device=self.device)
        self.order.copy_(base)
        self.ids.copy_(base if ids is None else ids)
        self.use_ids = ids is not None
        self._rows_known = True
        self.exchange()

    @property
    def ϱ_bar(self):
        """species.py:1790-1836 without CLASS: Ω ρ_crit for the plain matter species, else
        N mass / boxsize³ (particles)."""
        p = self.params
        class_species = {'matter': 'b+cdm', 'baryon': 'b', 'baryons': 'b',
                         'cold dark matter': 'cdm', 'dark matter': 'cdm'}.get(self.species)
        if class_species is not None:
            return sum({'b': p.Ωb, 'cdm': p.Ωcdm}[s_] for s_ in class_species.split('+'))*p.ρ_crit
        if self.representation == 'particles':
            return self.N*self.mass/p.boxsize**3
        if getattr(self, 'ρ', None) is not None:
            # a fluid of another species: the mean of its (conserved) density grid; on several
            # domains every slab has the same number of cells
            mean = float(self.ρ.mean().item())
            if self.comm is not None and self.nprocs > 1:
                mean = float(self.comm.all_gather_floats([mean]).mean())
            return mean
        raise ConceptGPUError(f'Cannot determine ϱ_bar for {self.name}')

    def w_eff(self, a=1.0):
        return 0.0  # matter; decaying species are out of scope

    def host(self, var, original_order=True):
        """Host copy of 'pos' | 'mom' | 'Δmom' | 'ids' | 'rung_indices' — of ALL N particles,
        gathered over the domains — by default in the order the particles were populated in
        (undoing tile_sort and exchange() through `order`); for a fluid component 'ϱ' | '𝒫' |
        'J' (stacked (3, g, g, g)), the whole grid."""
        if self.representation == 'fluid':
            if var == 'J':
                t = torch.stack(self.J)
                if self.comm is not None and self.nprocs > 1:
                    t = self.comm.all_gather_rows(t.transpose(0, 1).contiguous()).transpose(0, 1)
                return t.cpu().numpy()
            # (identifiers are NFKC-normalised by Python: self.ϱ is the attribute 'ρ')
            import unicodedata
            name = unicodedata.normalize('NFKC', {'rho': 'ϱ'}.get(var, var))
            t = getattr(self, name)
            if self.comm is not None and self.nprocs > 1:
                t = self.comm.all_gather_rows(t)
            return t.cpu().numpy()
        t = getattr(self, var)
        order = self.order
        if self.comm is not None and self.nprocs > 1:
            t = self.comm.all_gather_rows(t.contiguous())
            order = self.comm.all_gather_rows(order.contiguous())
        if original_order:
            out = torch.empty_like(t)
            out[order] = t
            t = out
        return t.cpu().numpy()

    # -- dynamics -----------------------------------------------------------
    def _mesh(self):
        p = self.params
        g = None
        for force, d in self.potential_gridsizes.items():
            for m, gs in d.items():
                g = gs.upstream
                break
            break
        if g is None:
            g = 2*int(round(self.N**(1/3)))
        # only the particle bookkeeping (tile order, drift, Δmom) needs this mesh: make any
        # size admissible (even, >= 4; on P domains divisible by 2 P with slabs >= 4 layers:
        # cg_create, mesh.py:1898-1905)
        g = max(4, g + g % 2)
        if self.nprocs > 1:
            step = 2*self.nprocs
            g = max(4*self.nprocs, -(-g//step)*step)
        return get_mesh(g, p.boxsize, p.nghosts, p.cell_centered, 2, self.device)

    def _use_mesh(self, mesh):
        """The store sorts and exchanges on `mesh` (tile tables are per grid size)."""
        st = self._store
        if st.mesh is not mesh:
            if (st.mesh.gridsize, st.mesh.nprocs) != (mesh.gridsize, mesh.nprocs):
                st.table = mesh.new_tile_table()
            st.mesh = mesh
            st._emig_for = None

    def exchange(self):
        """exchange() (communication.py:135-517)"""
        self._store.exchange()
        self.tiles_exact = False
        if self.nprocs > 1:
            self.tile_mesh = None  # rows changed hands: the tile table describes nothing now

    def drift(self, ᔑdt, a_next=-1, a=1.0):
        """species.py:2179-2199, followed on several domains by exchange() as in the
        reference's time loop (main.py:1406).  `a` is universals.a (only enters through
        a**(3*w_eff) = 1 for matter)."""
        Δt_over_mass = ᔑdt['a**(-2)']*a**(3*self.w_eff(a=a))/self.mass
        self._store.drift(Δt_over_mass)
        self.exchange()

    def tile_sort(self, mesh=None):
        """Reorder particle memory into mesh-tile order (the reference's
        tile_sort, species.py:2598-2810, reorders for the same reason)."""
        mesh = mesh or self._mesh()
        self._use_mesh(mesh)
        self._store.tile_sort()
        self.tile_mesh = mesh
        self.tiles_exact = True

    def drift_sort(self, ᔑdt, a_next=-1, a=1.0, mesh=None):
        """drift() (with its exchange()) immediately followed by tile_sort(), fused into one
        pass pair (cg_drift_sort): same result as the two calls."""
        Δt_over_mass = ᔑdt['a**(-2)']*a**(3*self.w_eff(a=a))/self.mass
        mesh = mesh or self._mesh()
        self._use_mesh(mesh)
        self._store.drift_exchange_sort(Δt_over_mass)
        self.tile_mesh = mesh
        self.tiles_exact = True

    def nullify_Δ(self, specifically=None, only_active=True):
        """species.py:3717-3741: Δmom = 0, for particles on active rungs only when rungs
        are in use."""
        if specifically is None:
            raise ConceptGPUError('You must specify "specifically" when calling '
                                  'Component.nullify_Δ() for particle components.')
        if specifically != 'mom':
            raise ConceptGPUError(f'Component.nullify_Δ(): specifically = {specifically} '
                                  'not supported')
        if self.Δmom is None:
            self._store.add_column('Δmom', dtype=torch.float64, width=3)
        elif only_active and self.use_rungs:
            self._mesh().dmom_nullify(self.Δmom, self.rung_indices, self.lowest_active_rung)
        else:
            self.Δmom.zero_()

    def apply_Δmom(self, only_active=True):
        """species.py:2253-2266"""
        if self.Δmom is None:
            return
        self._store.touch_mom()
        if only_active and self.use_rungs:
            self._mesh().dmom_apply(self.mom, self.Δmom, self.rung_indices,
                                    self.lowest_active_rung)
        else:
            self._mesh().dmom_apply(self.mom, self.Δmom)

    # -- adaptive rungs (A16) -----------------------------------------------
    def convert_Δmom_to_acc(self, ᔑdt_rungs, any_rung_jumps=False, a=1.0):
        """species.py:2290-2325"""
        if not self.use_rungs:
            return
        w_eff = self.w_eff(a=a)
        conversion_factors = a**(3*w_eff)/(self.mass*(commons.machine_ϵ
                                                      + np.asarray(ᔑdt_rungs['a**2'])))
        conv = commons.upload(conversion_factors, self.device)
        self._mesh().dmom_to_acc(self.Δmom, self.rung_indices, self.rung_indices_jumped,
                                 self.lowest_active_rung, conv, any_rung_jumps)

    def get_rung_factor(self, Δt, fac_softening):
        """species.py:2376-2400"""
        import math
        return 0.5*math.log2(Δt**2/(2*fac_softening*self.softening_length))

    def assign_rungs(self, Δt, fac_softening):
        """species.py:2422-2445"""
        if not self.use_rungs:
            self.rungs_N[0] = self.N
            return
        self._mesh().assign_rungs(self.Δmom, self.rung_indices, self.rung_indices_jumped,
                                  self.get_rung_factor(Δt, fac_softening), self.N_rungs)
        self.set_rungs_N()

    def flag_rung_jumps(self, Δt, Δt_jump_fac, fac_softening, ᔑdt_rungs):
        """species.py:2463-2513; returns whether any particle (of any domain) was flagged."""
        if not self.use_rungs:
            return False
        integrals = commons.upload(np.asarray(ᔑdt_rungs['1']), self.device)
        flagged = self._mesh().flag_rung_jumps(
            self.Δmom, self.rung_indices, self.rung_indices_jumped, self.lowest_active_rung,
            integrals, self.get_rung_factor(Δt*Δt_jump_fac, fac_softening),
            self.get_rung_factor(Δt/Δt_jump_fac, fac_softening), self.N_rungs)
        if self.comm is not None and self.nprocs > 1:
            flagged = self.comm.any(flagged)  # allreduce(..., op=MPI.LOR), species.py:2511
        return flagged

    def apply_rung_jumps(self):
        """species.py:2526-2549"""
        if not self.use_rungs:
            return
        self._mesh().apply_rung_jumps(self.rung_indices, self.rung_indices_jumped, self.N_rungs)
        self.set_rungs_N()

    # -- a sub-step of driftkick_short in two passes (one domain) -------------------------
    def substep_begin(self, ᔑdt_drift, flag, Δt=None, Δt_jump_fac=None, fac_softening=None,
                      ᔑdt_rungs=None, a=1.0, defer=False):
        """drift(ᔑdt_drift) (unless None), then — `flag` — flag_rung_jumps() and
        nullify_Δ('mom') as one pass over the particles (cg_substep_begin); what
        flag_rung_jumps() returns is read by substep_finish().  For components with rungs in
        use (the time loop takes the separate calls otherwise).  defer: the pass is left to the
        short-range cell list built next from these positions, which runs it on every particle
        as it bins it."""
        if not self.use_rungs:
            raise ConceptGPUError(f'{self.name}: substep_begin() is for components with rungs')
        mesh = self._mesh()
        if self._sub_dev is None:
            self._sub_dev = (torch.zeros(self.N_rungs, dtype=torch.int64, device=self.device),
                             torch.zeros(1, dtype=torch.int32, device=self.device))
            self._sub_host = (torch.zeros(self.N_rungs, dtype=torch.int64).pin_memory(),
                              torch.zeros(1, dtype=torch.int32).pin_memory())
            self._sub_event = torch.cuda.Event()
        dtm = None
        if ᔑdt_drift is not None:
            dtm = ᔑdt_drift['a**(-2)']*a**(3*self.w_eff(a=a))/self.mass
            self._store.sorted = False
            self._store._emig_for = None
            self.tiles_exact = False
        if flag and self.Δmom is None:
            self._store.add_column('Δmom', dtype=torch.float64, width=3)
        flag = bool(flag and self.use_rungs)
        if flag:
            args = (ᔑdt_rungs['1'], self.get_rung_factor(Δt*Δt_jump_fac, fac_softening),
                    self.get_rung_factor(Δt/Δt_jump_fac, fac_softening))
        else:
            args = (None, 0.0, 0.0)
        self._sub_flag_pending = flag
        if dtm is None and not flag:
            return
        self._begin_args = (dtm, flag, self.lowest_active_rung, args[0], args[1], args[2])
        if not defer:
            self.flush_begin()

    def take_begin(self, mesh):
        """hand a deferred substep_begin() to `mesh`, whose next shortrange_cells() on this
        component's positions runs it with its counting pass; True if there was one (the caller
        then reports the list queued: begin_queued())"""
        if self._begin_args is None:
            return False
        self.flush_begin(mesh, defer=True)
        return True

    def flush_begin(self, mesh=None, defer=False):
        """run a deferred substep_begin() now"""
        args, self._begin_args = self._begin_args, None
        if args is None:
            return
        dtm, flag, lowest, integrals_1, rf_up, rf_down = args
        (mesh or self._mesh()).substep_begin(
            self.pos, self.mom, self.Δmom, self.rung_indices, self.rung_indices_jumped, dtm, flag,
            lowest, integrals_1, rf_up, rf_down, self.N_rungs, self._sub_dev[1],
            self._sub_dev[0] if flag else None, defer)
        if not defer:
            self.begin_queued()

    def begin_queued(self):
        """the pass is on the stream: its populations and its flag follow it to the host"""
        if self._sub_flag_pending:
            self._sub_host[0].copy_(self._sub_dev[0], non_blocking=True)
            self._sub_host[1].copy_(self._sub_dev[1], non_blocking=True)
            self._sub_event.record()
            self._sub_counted = True

    def substep_end(self, apply, ᔑdt_rungs, a=1.0):
        """apply_Δmom() + convert_Δmom_to_acc() (`apply`: the component received a kick) and
        apply_rung_jumps() as one pass (cg_substep_end).  The populations the jumps leave were
        counted by the sub-step's first pass: substep_finish() has them."""
        self.flush_begin()
        if apply:
            self._store.touch_mom()
        conv = None
        if apply:
            w_eff = self.w_eff(a=a)
            conv = a**(3*w_eff)/(self.mass*(commons.machine_ϵ + np.asarray(ᔑdt_rungs['a**2'])))
        self._mesh().substep_end(self.mom, self.Δmom, self.rung_indices, self.rung_indices_jumped,
                                 apply, self.lowest_active_rung, conv, self.N_rungs, None)

    def substep_finish(self):
        """wait for the sub-step's FIRST pass (the sweep may still be running): the rung
        populations as the sub-step leaves them (set_rungs_N after apply_rung_jumps) are set,
        and whether any particle was flagged to jump is returned"""
        if not self._sub_counted:
            return False
        self._sub_counted = False
        self._sub_event.synchronize()
        counts = self._sub_host[0].tolist()
        self.rungs_N = counts[:self.N_rungs]
        populated = [r for r, c in enumerate(self.rungs_N) if c > 0]
        self.lowest_populated_rung = populated[0] if populated else self.N_rungs - 1
        self.highest_populated_rung = populated[-1] if populated else 0
        return bool(int(self._sub_host[1][0]))

    def set_rungs_N(self):
        """species.py:2560-2587 (set_rungs_N + set_lowest_highest_populated_rung): the
        populations are global (allreduce over the domains, species.py:2571)"""
        counts = self._mesh().rung_populations(self.rung_indices, self.N_rungs).cpu()
        if self.comm is not None and self.nprocs > 1:
            counts = self.comm.all_gather_ints(counts[:self.N_rungs].tolist()).sum(0)
        counts = counts.tolist()
        self.rungs_N = counts[:self.N_rungs]
        populated = [r for r, c in enumerate(self.rungs_N) if c > 0]
        self.lowest_populated_rung = populated[0] if populated else self.N_rungs - 1
        self.highest_populated_rung = populated[-1] if populated else 0
