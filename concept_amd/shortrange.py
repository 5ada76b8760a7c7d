"""concept_amd.shortrange — host side of the P3M short-range tile sweep.

Counterpart of component_component(..., pairing_level='tile')
(interactions.py:122-329) with gravity_pairwise_shortrange (gravity.py:263-354),
including the adaptive rungs (per-rung factors, active rungs, jumped indices:
cg_shortrange_sweep_rungs), and of pairing_level='domain' with gravity_pairwise /
gravity_pairwise_nonperiodic (direct summation with and without the Ewald
correction, component_component_pp below).  The pair loops run in
libconcept_gpu.so (cg_shortrange.hip, cg_pp.hip)."""
import math
import os

import numpy as np
import torch

from . import commons
from .lib import ConceptGPUError
from .mesh import get_mesh

_tables = {}
PAIR_KEY = 'a**(-3*w_eff₀-3*w_eff₁-1)'


def get_softened_r3inv(r2, ϵ, kernel='spline'):
    """interactions.py:1847-1914"""
    if kernel == 'none':
        r3 = r2*math.sqrt(r2)
        return 0 if r3 == 0 else 1/r3
    if kernel == 'plummer':
        r2_softened = r2 + ϵ**2
        return 1/(r2_softened*math.sqrt(r2_softened))
    if kernel != 'spline':
        raise ConceptGPUError(f'Softening kernel "{kernel}" not understood')
    h = 2.8*ϵ
    r = math.sqrt(r2)
    if r >= h:
        return 1/(r2*r)
    u = r/h
    if u < 0.5:
        return 32/h**3*(1./3. + u**2*(-6./5. + u))
    return 32/(3*r**3)*(u**3*(2 + u*(-9./2. + u*(18./5. - u))) - 3./480.)


def get_shortrange_table(softening, scale, range_, tablesize, kernel, device):
    """gravity.py:373-424.  Tabulated on the host (4096 entries), cached, kept in HBM."""
    key = (softening, scale, range_, tablesize, kernel, str(device))
    hit = _tables.get(key)
    if hit is not None:
        return hit
    maxr2 = (1 + 1/tablesize)*range_**2  # gravity.py:437
    r_tabulation = np.sqrt(np.linspace(0, maxr2, tablesize))
    table = np.empty(tablesize, dtype=np.float64)
    inv_scale = 1/scale
    for i in range(tablesize - 1):
        r2 = float(0.5*(r_tabulation[i]**2 + r_tabulation[i + 1]**2))
        r = math.sqrt(r2)
        x = r*inv_scale
        r3_inv = 1/(r2*r)
        table[i] = (
            - r3_inv*(1/math.sqrt(commons.π)*x*math.exp(-(0.5*x)**2) + (math.erfc(0.5*x) - 1))
            - get_softened_r3inv(r2, softening, kernel)
        )
    table[tablesize - 1] = np.nan  # never accessed (gravity.py:416-421)
    out = (torch.tensor(table, device=device), maxr2)
    _tables[key] = out
    return out


def combine_softening_lengths(ϵᵢ, ϵⱼ):
    """interactions.py:1820-1830"""
    return 0.5*(ϵᵢ + ϵⱼ)


def _pair_integrals(ᔑdt_rungs, rec, sup):
    """ᔑdt_rungs['a**(-3*w_eff₀-3*w_eff₁-1)', receiver, supplier] as a 1-D array: one entry
    per rung index (main.py:1203-1215 fills 3*N_rungs - 1 of them); a plain number — what
    get_time_step_integrals() gives for a single step — counts as rung 0."""
    k = (PAIR_KEY, rec.name, sup.name)
    if k not in ᔑdt_rungs:
        raise ConceptGPUError(
            f'ᔑdt_rungs lacks the integral {k!r} (both orders of a component pair are needed)')
    integrals = np.atleast_1d(np.asarray(ᔑdt_rungs[k], dtype=np.float64))
    if integrals.ndim != 1:
        raise ConceptGPUError(f'ᔑdt_rungs[{k!r}] must be a number or a 1-D array of rung integrals')
    if rec.use_rungs and integrals.size < 3*rec.N_rungs - 1:
        raise ConceptGPUError(
            f'ᔑdt_rungs[{k!r}] holds {integrals.size} integrals, {3*rec.N_rungs - 1} '
            f'(3*N_rungs - 1) are needed with rungs in use')
    return integrals


sparse_sweeps = 0   # sweeps taken without a cell list (a handful of active receivers)
by_receiver_meshes = {}   # meshes that took a sweep by active receiver since the time loop looked


def component_component(force, receivers, suppliers, ᔑdt_rungs, gridsize):
    """Short-range gravity of every (receiver, supplier) component pair, accumulated
    into the components' Δmom buffers (the caller applies them, main.py:1253-1262)."""
    if force != 'gravity':
        raise ConceptGPUError(f'short-range force "{force}" is not built')
    p = receivers[0].params
    sr = commons.resolve_shortrange(p, gridsize)
    nt = int((p.boxsize/1)/sr['tilesize']*(1 + commons.machine_ϵ))  # species.py:3943-3950
    if nt < 4:
        raise ConceptGPUError(
            'The global gravity tiling needs to have at least 4 tiles across the box in every '
            'direction. Consider lowering shortrange_params["gravity"]["tilesize"].')
    if sr['tilesize'] < sr['range']*(1 - 1e-12):
        raise ConceptGPUError('shortrange_params: tilesize must be at least the range')
    tile_extent = p.boxsize/nt  # species.py:607-609
    mesh = get_mesh(gridsize, p.boxsize, p.nghosts, p.cell_centered, 2, receivers[0].device)
    involved = list({id(c): c for c in list(receivers) + list(suppliers)}.values())
    for c in involved:
        if c.representation != 'particles':
            raise ConceptGPUError(f'{c.name}: only particle components have short-range forces')
        # (asked of the store, not through the attribute: reading c.Δmom would run a sub-step
        # pass the time loop has left to this call's cell list)
        if 'Δmom' not in c._store.cols:
            c.Δmom = torch.zeros_like(c.mom)
    # On several domains every supplier component is extended by the neighbour ranks'
    # particles within the force range of this rank's slab (sendrecv_component,
    # communication.py:847-1130: "supplier particles in the boundary tiles").  Positions only:
    # the sweep is one-sided, every rank kicks its OWN receivers with their own rung factors,
    # so neither the suppliers' rung indices nor any Δmom crosses the wire.
    multi = mesh.dist and mesh.nprocs > 1
    cells, supp_pos, supp_cells = {}, {}, {}
    if multi:
        from .distributed import check_shortrange_fits, ship_boundary_positions
        # (components own their particles by the slabs of their own grids: the faces of two
        # components differ by less than half a cell of the coarser one)
        slack = max(c._store.mesh.boxsize/c._store.mesh.gridsize for c in involved)
        for c in involved:
            check_shortrange_fits(c._store.mesh, sr['range'] + slack)
    build = mesh.shortrange_cells
    def get_cells(c):
        """the component's cell list (built when a sweep first asks for it: a sub-step that
        kicks a handful of particles needs none, see sweep() below)"""
        if id(c) not in cells:
            # a sub-step (lowest active rung > 0): the particles on active rungs first in every
            # cell, so that the sweep takes its receivers without looking at the rungs again
            # (the sub-step's first pass over the particles, if the time loop left it to this
            # list: run by the list's counting pass on the particles it bins)
            taken = c.take_begin(mesh)
            rungs = ((c.rung_indices, c.rung_indices_jumped, c.lowest_active_rung)
                     if c.use_rungs and c in receivers else None)
            # (many active receivers: the sweep goes in blocks and wants the jumped rung
            # indices in list order)
            in_blocks = (rungs is not None and c.lowest_active_rung > 0 and
                         sum(c.rungs_N[c.lowest_active_rung:]) > mesh.SHORTRANGE_BY_CELL_MAX*c.N)
            cells[id(c)] = build(c.pos, nt, tile_extent, rungs, in_blocks)
            if taken:
                c.begin_queued()
            supp_cells.setdefault(id(c), cells[id(c)])
        return cells[id(c)]
    for c in involved:
        # every involved component can act as supplier: s of sweep(r, s), and r of the
        # reciprocal sweep(s, r) when s is also a receiver
        if multi:
            get_cells(c)
            ghosts = ship_boundary_positions(c._store.mesh, c.pos,
                                             sr['range']*(1 + 1e-9) + slack + 1e-9*p.boxsize)
            # rows 0..N_local-1 ARE the component's own particles (the sweep's `same`
            # convention), the ghosts follow
            supp_pos[id(c)] = torch.cat([c.pos] + list(ghosts)).contiguous()
            supp_cells[id(c)] = build(supp_pos[id(c)], nt, tile_extent)

    def active_rows(rec):
        """rows of the receiver's particles on active rungs when they are few (one domain:
        rungs_N counts them without looking), else None"""
        if multi or not rec.use_rungs or rec.lowest_active_rung <= 0:
            return None
        n_active = sum(rec.rungs_N[rec.lowest_active_rung:])
        if n_active > mesh.SHORTRANGE_SPARSE_MAX:
            return None
        key_ = (id(rec), rec.lowest_active_rung)
        if key_ not in sparse_rows:
            # (the populations say how many there are: the rows are found without the host
            # waiting for their number — torch.nonzero() would drain the stream, the sweep of
            # the sub-step before included, once per such sub-step)
            mask = rec.rung_indices >= rec.lowest_active_rung
            if hasattr(torch, 'nonzero_static'):
                sparse_rows[key_] = torch.nonzero_static(mask, size=n_active).flatten()
            else:
                sparse_rows[key_] = torch.nonzero(mask).flatten()
        rows_ = sparse_rows[key_]
        # (the populations are bookkeeping of the time loop: should they lag behind the rung
        # array, the cells sweep takes over)
        return rows_ if rows_.numel() <= mesh.SHORTRANGE_SPARSE_MAX else None
    sparse_rows = {}
    key = 'a**(-3*w_eff₀-3*w_eff₁-1)'
    done = set()
    for r in receivers:
        for s in suppliers:
            pair = frozenset((id(r), id(s)))
            if pair in done:
                continue
            done.add(pair)
            softening = combine_softening_lengths(r.softening_length, s.softening_length)
            table, maxr2 = get_shortrange_table(softening, sr['scale'], sr['range'],
                                                sr['tablesize'], p.softening_kernel, r.device)
            scaling = (sr['tablesize'] - 1)/maxr2  # gravity.py:288
            r2_max = sr['range']**2                # gravity.py:286
            same = r is s

            def sweep(rec, sup, same_):
                # compute_factors (gravity.py:51-64): G*m_r*m_s*ᔑdt_rungs[...][k] per rung k
                integrals = _pair_integrals(ᔑdt_rungs, rec, sup)
                if rec.use_rungs:
                    factors = commons.upload(p.G_Newton*rec.mass*sup.mass*integrals, rec.device)
                    rows = active_rows(rec)
                    if rows is not None:
                        # the sub-steps for the highest rungs (main.py:1347-1624): a handful
                        # of receivers against all suppliers, no cell list
                        global sparse_sweeps
                        sparse_sweeps += 1
                        if rows.numel():
                            rec.flush_begin()
                            sup.flush_begin()
                            mesh.shortrange_sparse(rec.pos, rows, rec.Δmom,
                                                   supp_pos[id(sup)] if multi else sup.pos,
                                                   table, scaling, r2_max, 0.0,
                                                   (factors, rec.rung_indices_jumped))
                        return
                    get_cells(sup)
                    rc = get_cells(rec)
                    rungs = (factors, rec.rung_indices, rec.rung_indices_jumped,
                             rec.lowest_active_rung)
                    # few receivers on active rungs (rungs_N counts them — over all domains: an
                    # upper bound of this domain's, which is all the sweep asks for): the sweep
                    # by active receiver
                    n_active = None
                    if rec.lowest_active_rung > 0:
                        n_active = int(sum(rec.rungs_N[rec.lowest_active_rung:]))
                        if n_active > mesh.SHORTRANGE_BY_CELL_MAX*rec.N:
                            n_active = None
                    if n_active is not None:
                        by_receiver_meshes[id(mesh)] = mesh
                    mesh.shortrange_sweep_cells(rc, rec.Δmom, supp_cells[id(sup)], nt, table,
                                                scaling, r2_max, 0.0, rungs, n_active)
                else:
                    rc = get_cells(rec)
                    get_cells(sup)
                    mesh.shortrange_sweep_cells(
                        rc, rec.Δmom, supp_cells[id(sup)], nt, table, scaling, r2_max,
                        p.G_Newton*rec.mass*sup.mass*float(integrals[0]))
            sweep(r, s, same)
            if not same and s in receivers:
                # the reference kicks both partners of a pair (Δmom_s -= ..., gravity.py:341-349)
                sweep(s, r, False)


_ewald_grids = {}


def get_ewald_grid(p, device):
    """ewald.get_ewald_grid() (ewald.py:200-224): the octant table of the Ewald correction,
    tabulated once per (ewald_gridsize, device) on the GPU and kept in HBM (the reference
    caches it on disk)."""
    key = (p.ewald_gridsize, str(device))
    grid = _ewald_grids.get(key)
    if grid is None:
        mesh = get_mesh(16, p.boxsize, p.nghosts, p.cell_centered, 2, device, role='pp')
        grid = _ewald_grids[key] = mesh.ewald_tabulate(p.ewald_gridsize)
    return grid


def component_component_pp(force, receivers, suppliers, ᔑdt_rungs, periodic):
    """component_component(..., pairing_level='domain') (interactions.py:122-329) with
    gravity_pairwise (periodic, Ewald-corrected) or gravity_pairwise_nonperiodic
    (gravity.py:121-206, 491-560): direct summation, accumulated into the components' Δmom
    buffers (the caller applies them, main.py:1253-1262)."""
    if force != 'gravity':
        raise ConceptGPUError(f'direct summation of force "{force}" is not built')
    p = receivers[0].params
    dev = receivers[0].device
    mesh = get_mesh(16, p.boxsize, p.nghosts, p.cell_centered, 2, dev, role='pp')
    ewald_grid = get_ewald_grid(p, dev) if periodic else None
    for c in {id(c): c for c in list(receivers) + list(suppliers)}.values():
        if c.representation != 'particles':
            raise ConceptGPUError(f'{c.name}: only particle components take part in direct '
                                  'summation')
        if c.Δmom is None:
            c.Δmom = torch.zeros_like(c.mom)
    key = 'a**(-3*w_eff₀-3*w_eff₁-1)'

    def all_positions(sup):
        """The supplier's particles of EVERY domain, this domain's first (the reference pairs
        every domain with every other, interactions.py:398-590; direct summation is O(N²)
        anyway, so the positions are simply gathered)."""
        comm = sup.comm
        if comm is None or comm.world == 1:
            return sup.pos
        counts = comm.all_gather_ints([sup.N_local])[:, 0].tolist()
        everything = comm.all_gather_rows(sup.pos.contiguous())
        start = int(sum(counts[:comm.rank]))
        return torch.cat([sup.pos, everything[:start], everything[start + sup.N_local:]])
    done = set()
    for r in receivers:
        for s in suppliers:
            pair = frozenset((id(r), id(s)))
            if pair in done:
                continue
            done.add(pair)
            softening = combine_softening_lengths(r.softening_length, s.softening_length)

            def kick(rec, sup, same):
                # compute_factors (gravity.py:51-64): G*m_r*m_s*ᔑdt_rungs[...][k] per rung k
                integrals = _pair_integrals(ᔑdt_rungs, rec, sup)
                rungs, factor = None, p.G_Newton*rec.mass*sup.mass*float(integrals[0])
                if rec.use_rungs:
                    factors = commons.upload(p.G_Newton*rec.mass*sup.mass*integrals, rec.device)
                    rungs = (factors, rec.rung_indices, rec.rung_indices_jumped,
                             rec.lowest_active_rung)
                    factor = 0.0
                mesh.pp_kick(rec.pos, rec.Δmom, all_positions(sup), same, ewald_grid, softening,
                             p.softening_kernel, factor, rungs)
            kick(r, s, r is s)
            if r is not s and s in receivers:
                kick(s, r, False)  # the reference kicks both partners of a pair
