"""concept_amd.species — the GPU-resident particle Component.

Mirrors the slice of the reference's Component the gravity path uses
(species.py:852-1040 fields, :1911-1925 populate, :2010-2064 AoS double[3N]
pos/mom/Δmom, :2179-2199 drift, :2253-2266 apply_Δmom, :3717-3741 nullify_Δ,
:2598-2810 tile_sort), with the particle arrays living in HBM as torch CUDA
tensors of shape (N, 3), float64 — the reference's own "xyzxyz..." layout."""
import collections

import numpy as np
import torch

from . import commons
from .lib import ConceptGPUError
from .mesh import get_mesh

PotentialGridsizes = collections.namedtuple('PotentialGridsizes', ('upstream', 'downstream'))


class Component:
    def __init__(self, name, species, *, N=None, gridsize=None, mass=None, boltzmann_order=-1,
                 device=None, params=None):
        """Component(name, species, N=..., mass=...)                      particles
        Component(name, species, gridsize=..., boltzmann_order=1)        fluid
        (species.py:852-1040).  A fluid component here is what the gravity path needs of
        it: the grids ϱ, J[0..2] and 𝒫 as float64 tensors (gridsize,)*3 on the GPU (the
        reference's grid_noghosts); its own evolution (fluid.py) stays with the caller."""
        self.params = p = params or commons.params
        if p is None:
            raise ConceptGPUError('no parameters loaded: call concept_amd.commons.load_params()')
        self.name = name.strip()
        self.species = species
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        if (N is None) == (gridsize is None):
            raise ConceptGPUError(
                f'{self.name}: give N (particle component) or gridsize (fluid component)')
        if N is not None:
            if boltzmann_order != -1:
                raise ConceptGPUError(f'{self.name}: particle components have '
                                      'boltzmann_order = -1')
            self.representation = 'particles'
            self._init_particles(int(N), mass)
        else:
            if boltzmann_order < 0:
                raise ConceptGPUError(f'{self.name}: fluid components need boltzmann_order >= 0')
            self.representation = 'fluid'
            self.boltzmann_order = int(boltzmann_order)
            self._init_fluid(int(gridsize))
        self._init_forces()

    def _init_fluid(self, gridsize):
        if gridsize < 2 or gridsize % 2:
            raise ConceptGPUError(f'{self.name}: fluid grid size {gridsize} must be even and ≥ 2')
        self.gridsize = gridsize
        self.N = self.N_local = 0
        self.mass = -1.0  # species.py: fluids carry no particle mass
        shape = (gridsize,)*3
        z = lambda: torch.zeros(shape, dtype=torch.float64, device=self.device)
        self.ϱ, self.𝒫 = z(), z()
        self.J = [z(), z(), z()]
        self.use_rungs = False
        self.tile_table = None
        self.tile_mesh = None
        self.tiles_exact = False

    def _init_particles(self, N, mass):
        p = self.params
        self.N = self.N_local = N
        self.mass = float(mass)
        self.softening_length = commons.softening_length(p, self.species, self.N)
        self.pos = torch.zeros((self.N, 3), dtype=torch.float64, device=self.device)
        self.mom = torch.zeros((self.N, 3), dtype=torch.float64, device=self.device)
        self.Δmom = None  # allocated on first short-range use
        # ids: the particles' identifiers (species.py:2040-2064 `ids`; a snapshot's ID block);
        # order: where each row stood when the arrays were populated — tile_sort permutes
        # both with the particles, host(original_order=True) undoes it through `order`
        self.ids = torch.arange(self.N, dtype=torch.int64, device=self.device)
        self.order = torch.arange(self.N, dtype=torch.int64, device=self.device)
        self._scratch = None
        # tile order bookkeeping: `tile_table` (first particle of each mesh tile) is
        # exact right after tile_sort() on `tile_mesh`; a drift makes it approximate
        self.tile_table = None
        self.tile_mesh = None
        self.tiles_exact = False

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
            self.rung_indices = torch.zeros(self.N, dtype=torch.int8, device=self.device)
            self.rung_indices_jumped = torch.zeros(self.N, dtype=torch.int8, device=self.device)
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
        also accepts var='pos'/'mom' with an (N, 3) array."""
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(data, dtype=np.float64)))
        if self.representation == 'fluid':
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
            target.copy_(t)
            return
        if var == 'ids':
            self.ids.copy_(torch.as_tensor(np.ascontiguousarray(np.asarray(data, dtype=np.int64))))
            return
        if var.startswith('pos'):
            self.tile_table = None
            self.tiles_exact = False
            # new positions arrive in the caller's order: rows are "as populated" again
            self.order = torch.arange(self.N, dtype=torch.int64, device=self.device)
        if var in ('pos', 'mom'):
            getattr(self, var).copy_(t.reshape(self.N, 3))
            return
        prefix, suffix = var[:-1], var[-1]
        getattr(self, prefix)[:, 'xyz'.index(suffix)] = t.to(self.device)

    def w_eff(self, a=1.0):
        return 0.0  # matter; decaying species are out of scope

    def host(self, var, original_order=True):
        """Host copy of 'pos' | 'mom' | 'Δmom', by default in the order the
        particles were populated in (undoing tile_sort via ids); for a fluid
        component 'ϱ' | '𝒫' | 'J' (stacked (3, g, g, g))."""
        if self.representation == 'fluid':
            if var == 'J':
                return torch.stack(self.J).cpu().numpy()
            # (identifiers are NFKC-normalised by Python: self.ϱ is the attribute 'ρ')
            import unicodedata
            name = unicodedata.normalize('NFKC', {'rho': 'ϱ'}.get(var, var))
            return getattr(self, name).cpu().numpy()
        t = getattr(self, var)
        if original_order:
            out = torch.empty_like(t)
            out[self.order] = t
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
            g = max(4, 2*int(round(self.N**(1/3))))
        return get_mesh(g, p.boxsize, p.nghosts, p.cell_centered, 2, self.device)

    def drift(self, ᔑdt, a_next=-1, a=1.0):
        """species.py:2179-2199.  `a` is universals.a (only enters through
        a**(3*w_eff) = 1 for matter)."""
        Δt_over_mass = ᔑdt['a**(-2)']*a**(3*self.w_eff(a=a))/self.mass
        self._mesh().drift(self.pos, self.mom, Δt_over_mass)
        self.tiles_exact = False

    def _sorted_into(self, mesh, Δt_over_mass=None):
        """Run the (drift +) tile sort into the scratch buffers and swap; everything that is
        indexed by particle (ids, rung indices, Δmom) follows through the slot permutation."""
        if self._scratch is None:
            self._scratch = (torch.empty_like(self.pos), torch.empty_like(self.mom),
                             torch.empty_like(self.ids))
            self._slots = torch.arange(self.N, dtype=torch.int64, device=self.device)
        po, mo, perm = self._scratch
        if self.tile_table is None or self.tile_mesh is not mesh:
            self.tile_table = mesh.new_tile_table()
        if Δt_over_mass is None:
            mesh.sort_particles(self.pos, self.mom, self._slots, po, mo, perm, self.tile_table)
        else:
            mesh.drift_sort(self.pos, self.mom, self._slots, po, mo, perm, Δt_over_mass,
                            self.tile_table)
        self._scratch = (self.pos, self.mom, perm)
        self.pos, self.mom = po, mo
        self.ids = self.ids[perm]
        self.order = self.order[perm]
        if self.use_rungs:
            self.rung_indices = self.rung_indices[perm]
            self.rung_indices_jumped = self.rung_indices_jumped[perm]
        if self.Δmom is not None:
            self.Δmom = self.Δmom[perm]
        self.tile_mesh = mesh
        self.tiles_exact = True

    def tile_sort(self, mesh=None):
        """Reorder particle memory into mesh-tile order (the reference's
        tile_sort, species.py:2598-2810, reorders for the same reason)."""
        self._sorted_into(mesh or self._mesh())

    def drift_sort(self, ᔑdt, a_next=-1, a=1.0, mesh=None):
        """drift() immediately followed by tile_sort(), fused into one pass pair
        (cg_drift_sort): same result as the two calls."""
        Δt_over_mass = ᔑdt['a**(-2)']*a**(3*self.w_eff(a=a))/self.mass
        self._sorted_into(mesh or self._mesh(), Δt_over_mass)

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
            self.Δmom = torch.zeros_like(self.mom)
        elif only_active and self.use_rungs:
            self._mesh().dmom_nullify(self.Δmom, self.rung_indices, self.lowest_active_rung)
        else:
            self.Δmom.zero_()

    def apply_Δmom(self, only_active=True):
        """species.py:2253-2266"""
        if self.Δmom is None:
            return
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
        import numpy as np
        w_eff = self.w_eff(a=a)
        conversion_factors = a**(3*w_eff)/(self.mass*(commons.machine_ϵ
                                                      + np.asarray(ᔑdt_rungs['a**2'])))
        conv = torch.tensor(conversion_factors, dtype=torch.float64, device=self.device)
        self._mesh().dmom_to_acc(self.Δmom, self.rung_indices, self.rung_indices_jumped,
                                 self.lowest_active_rung, conv, any_rung_jumps)

    def get_rung_factor(self, Δt, fac_softening):
        """species.py:2376-2400"""
        import math
        return 0.5*math.log2(Δt**2/(2*fac_softening*self.softening_length))

    def assign_rungs(self, Δt, fac_softening):
        """species.py:2422-2445"""
        if not self.use_rungs:
            self.rungs_N[0] = self.N_local
            return
        self._mesh().assign_rungs(self.Δmom, self.rung_indices, self.rung_indices_jumped,
                                  self.get_rung_factor(Δt, fac_softening), self.N_rungs)
        self.set_rungs_N()

    def flag_rung_jumps(self, Δt, Δt_jump_fac, fac_softening, ᔑdt_rungs):
        """species.py:2463-2513; returns whether any particle was flagged."""
        if not self.use_rungs:
            return False
        import numpy as np
        integrals = torch.tensor(np.asarray(ᔑdt_rungs['1']), dtype=torch.float64,
                                 device=self.device)
        return self._mesh().flag_rung_jumps(
            self.Δmom, self.rung_indices, self.rung_indices_jumped, self.lowest_active_rung,
            integrals, self.get_rung_factor(Δt*Δt_jump_fac, fac_softening),
            self.get_rung_factor(Δt/Δt_jump_fac, fac_softening), self.N_rungs)

    def apply_rung_jumps(self):
        """species.py:2526-2549"""
        if not self.use_rungs:
            return
        self._mesh().apply_rung_jumps(self.rung_indices, self.rung_indices_jumped, self.N_rungs)
        self.set_rungs_N()

    def set_rungs_N(self):
        """species.py:2560-2587 (set_rungs_N + set_lowest_highest_populated_rung)"""
        counts = torch.bincount(self.rung_indices.long(), minlength=self.N_rungs).cpu().tolist()
        self.rungs_N = counts[:self.N_rungs]
        populated = [r for r, c in enumerate(self.rungs_N) if c > 0]
        self.lowest_populated_rung = populated[0] if populated else self.N_rungs - 1
        self.highest_populated_rung = populated[-1] if populated else 0
