"""concept_amd.interactions — the drop-in boundary of the gravity path.

Same names, argument meaning and error behaviour as the reference's
interaction layer for this path:
  register()            interactions.py:2646-2677
  get_potential_specs() interactions.py:2786-2827
  gravity()             interactions.py:2838-2961   <- the boundary
  particle_mesh()       interactions.py:1985-2335
Everything below these functions runs in libconcept_gpu.so on the MI355X;
the host code here only evaluates the scalars of the call with the
reference's own expressions.  The reference aborts on misuse
(commons.py:1002-1031); here that is a ConceptGPUError."""
import collections

from . import commons
from .lib import ConceptGPUError
from .mesh import get_mesh

π = commons.π

interactions_registered = {}
Interaction = collections.namedtuple(
    'Interaction', ('force', 'methods', 'conjugated_name', 'dependent', 'affected',
                    'deterministic', 'instantaneous'))


def register(force, methods, conjugated_name=None, *, dependent=('pos', ), affected=('mom', ),
             deterministic=True, instantaneous=False):
    """interactions.py:2646-2677"""
    if isinstance(methods, str):
        methods = [methods]
    interactions_registered[force] = Interaction(
        force, list(methods), conjugated_name or force, list(dependent), list(affected),
        deterministic, instantaneous)


PotentialInfo = collections.namedtuple(
    'PotentialInfo', ('gridsize', 'interpolation_order', 'deconvolve', 'interlace'))
Deconvolve = collections.namedtuple('Deconvolve', ('upstream', 'downstream'))
Interlace = collections.namedtuple('Interlace', ('upstream', 'downstream'))


def get_potential_specs(force, method, receivers, suppliers):
    """interactions.py:2786-2821"""
    p = receivers[0].params
    gridsize = p.potential_options['gridsize']['global'].get(force, {}).get(method, -1)
    if gridsize == -1:
        gridsizes = (
            {s.potential_gridsizes[force][method].upstream for s in suppliers}
            | {r.potential_gridsizes[force][method].downstream for r in receivers})
        if len(gridsizes) != 1:
            raise ConceptGPUError(
                f'No global potential grid size specified for force "{force}" with '
                f'method "{method}". As multiple upstream and/or downstream grid sizes '
                f'are in use, the global grid size could not be set automatically.')
        gridsize = gridsizes.pop()
    return PotentialInfo(
        gridsize,
        p.potential_options['interpolation'][force][method],
        Deconvolve(*p.potential_options['deconvolve'][force][method]),
        Interlace(*p.potential_options['interlace'][force][method]),
    )


def _aligned(component, mesh):
    """On several domains a particle component owns its particles by the x-slabs of ITS grid
    (lower CIC cell inside the slab).  On a mesh of another grid size the slab
    faces sit up to half a (coarser) cell elsewhere: such a component is deposited and
    gathered with the direct kernels through the full halo on both sides, never the tiled
    ones."""
    if not mesh.dist or mesh.nprocs == 1 or component.representation != 'particles':
        return True
    return component._store.mesh.gridsize == mesh.gridsize


def _check_halo_reach(component, mesh, cloud, stencil):
    """The halo is G = 3 layers (cg_create): refuse what would reach beyond it.  cloud: how
    far the interpolation reaches from the lower CIC cell (CIC: 0 below, 1 above), stencil:
    half width of a finite difference fused into the gather."""
    if _aligned(component, mesh):
        return
    import math
    delta = (mesh.gridsize/component._store.mesh.gridsize - 1)/2
    up = math.ceil(max(delta, 0.0)) + cloud[1] + stencil
    down = (1 if delta < 0 else 0) + cloud[0] + stencil
    if max(up, down) > mesh.ghost_layers:
        raise ConceptGPUError(
            f'{component.name}: its particles are distributed by the slabs of its '
            f'{component._store.mesh.gridsize}^3 grid; on the {mesh.gridsize}^3 mesh they reach '
            f'{max(up, down)} layers beyond a slab face, the halo holds {mesh.ghost_layers}. '
            f'Use grid sizes closer to each other or fewer domains.')


def _default_fast_path(receivers, suppliers, gridsize_global, force, method, interpolation_order,
                       interlace_upstream, interlace_downstream):
    """The default configuration: particle components only, every upstream / downstream
    grid size equal to the global one, CIC, 'sc' lattices, finite-difference gradient."""
    for c in list(receivers) + list(suppliers):
        if c.representation != 'particles':
            return False
    if any(s.potential_gridsizes[force][method].upstream != gridsize_global for s in suppliers):
        return False
    if any(r.potential_gridsizes[force][method].downstream != gridsize_global
           for r in receivers):
        return False
    if interpolation_order != 2:
        return False
    if any(len(lattice_shifts(x)) != 1 for x in (interlace_upstream, interlace_downstream)):
        return False
    return all(r.potential_differentiations[force][method] in (2, 4) for r in receivers)


def particle_mesh(receivers, suppliers, gridsize_global, quantity, force, method, potential,
                  interpolation_order, deconvolve_upstream, deconvolve_downstream,
                  interlace_upstream, interlace_downstream, ᔑdt, ᔑdt_key):
    """interactions.py:1985-2335.  The default configuration takes the fused kernels
    (one deposit, the 5-pass Poisson solve, one gather-kick); everything else — fluid
    components, several representations — is assembled as the reference does it
    (particle_mesh_general)."""
    if not receivers or not suppliers:
        return
    if potential not in {'gravity', 'gravity long-range'}:
        raise ConceptGPUError(
            f'particle_mesh() got potential "{potential}" ∉ {{"gravity", "gravity long-range"}}')
    if quantity != 'a²ρ':
        raise ConceptGPUError(f'particle_mesh(): quantity "{quantity}" is not on the gravity path')
    if not _default_fast_path(receivers, suppliers, gridsize_global, force, method,
                              interpolation_order, interlace_upstream, interlace_downstream):
        return particle_mesh_general(
            receivers, suppliers, gridsize_global, quantity, force, method, potential,
            interpolation_order, deconvolve_upstream, deconvolve_downstream,
            interlace_upstream, interlace_downstream, ᔑdt, ᔑdt_key)
    p = receivers[0].params
    boxsize = p.boxsize
    mesh = get_mesh(gridsize_global, boxsize, p.nghosts, p.cell_centered, interpolation_order,
                    receivers[0].device)
    # interactions.py:2069-2080: with particle-only suppliers/receivers on the global
    # grid size both deconvolutions are promoted to the global one
    deconv_order_global = (int(bool(deconvolve_upstream)) + int(bool(deconvolve_downstream)))
    deconv_order_global *= interpolation_order
    # interpolate_upstream (mesh.py:492-635): nullified grid, every supplier deposited
    # Suppliers whose memory is in exact tile order of this mesh (Component.tile_sort)
    # take the LDS-tiled kernel; the first of them assigns the mesh (no zero-fill pass).
    fft_factor = float(gridsize_global)**(-3)  # mesh.py:582
    ordered = sorted(suppliers, key=lambda s: not (s.tiles_exact and s.tile_mesh is mesh))
    mesh_started = False
    aligned = all(_aligned(s, mesh) for s in suppliers)
    if not aligned:  # clouds land in the halo on both sides: it must start from zero
        for s in suppliers:
            _check_halo_reach(s, mesh, (0, 1), 0)
        mesh.zero()
        mesh_started = True
    for supplier in ordered:
        contribution = _particle_contribution(supplier, ᔑdt, fft_factor, gridsize_global, boxsize)
        if supplier.tiles_exact and supplier.tile_mesh is mesh:
            mesh.deposit_tiled(supplier.pos, supplier.tile_table, contribution,
                               accumulate=mesh_started)
        else:
            if not mesh_started:
                mesh.zero()
            mesh.deposit(supplier.pos, contribution)
        mesh_started = True
    # communicate_ghosts(grid, '+=') (mesh.py:609) and, after the solve, communicate_ghosts(grid,
    # '=') (interactions.py:2303-2307): posted here, both travel under the transforms (one
    # domain wraps by itself and needs neither)
    fold = mesh.fold_ghosts_start(general=not aligned)
    # interactions.py:2092-2118 and :2302
    C, long_range, E = _potential_constants(p, potential, gridsize_global)
    mesh.poisson_solve(deconv_order_global, C, long_range, E, fold_finish=fold, fill=True)
    # interactions.py:2311-2332 via apply_particle_mesh_force (:2359-2387)
    for receiver in receivers:
        _kick_particles(mesh, receiver, force, method, ᔑdt, ᔑdt_key)


def pm_streaming_plan(components):
    """What stepper.timeloop needs to run gravity('pm') of the default configuration (the fast
    path of particle_mesh above) together with the drift that follows it, in one pass over
    particles kept in tile regions (cg_gather_kick_drift_scatter): the mesh, the promoted
    deconvolution order and the potential's constants — or None where the configuration is not
    that one (then the loop calls gravity() and drift() one after the other)."""
    force, method = 'gravity', 'pm'
    if any(c.representation != 'particles' or c.forces.get(force) != method or c.use_rungs
           for c in components):
        return None
    specs = get_potential_specs(force, method, components, components)
    if not _default_fast_path(components, components, specs.gridsize, force, method,
                              specs.interpolation_order, specs.interlace.upstream,
                              specs.interlace.downstream):
        return None
    p = components[0].params
    mesh = get_mesh(specs.gridsize, p.boxsize, p.nghosts, p.cell_centered,
                    specs.interpolation_order, components[0].device)
    if not all(_aligned(c, mesh) for c in components):
        return None
    deconv = (int(bool(specs.deconvolve.upstream)) + int(bool(specs.deconvolve.downstream)))
    C, long_range, E = _potential_constants(p, 'gravity', specs.gridsize)
    return {'mesh': mesh, 'gridsize': specs.gridsize, 'deconv_order': deconv*specs.interpolation_order,
            'C': C, 'long_range': long_range, 'E': E, 'force': force, 'method': method}


def _particle_contribution(supplier, ᔑdt, fft_factor, gridsize, boxsize):
    """mesh.py:1550-1573 for quantity 'a²ρ'"""
    contribution = ᔑdt['a**(-3*w_eff-1)', supplier.name]/ᔑdt['1']
    contribution *= supplier.mass
    contribution_factor = fft_factor*(gridsize/boxsize)**3
    contribution *= contribution_factor
    return contribution


def _potential_constants(p, potential, gridsize_global):
    """interactions.py:2105 and :2110-2113"""
    C = -p.boxsize**2*p.G_Newton/π
    if potential == 'gravity':
        return C, False, 0.0
    scale = commons.resolve_shortrange(p, gridsize_global)['scale']
    return C, True, -(2*π/p.boxsize*scale)**2


def _same_tiling(component, mesh):
    """The tile order of a component depends on the grid geometry only: it is valid on
    every mesh context of the same grid size (the roles of the general path)."""
    t = component.tile_mesh
    return (component.tile_table is not None and t is not None
            and (t is mesh or (t.gridsize, t.boxsize, t.nghosts, t.device)
                 == (mesh.gridsize, mesh.boxsize, mesh.nghosts, mesh.device)))


def _kick_particles(mesh, receiver, force, method, ᔑdt, ᔑdt_key):
    key = (ᔑdt_key[0], receiver.name) if isinstance(ᔑdt_key, tuple) else ᔑdt_key
    differentiation_order = receiver.potential_differentiations[force][method]
    factor = receiver.mass*(-ᔑdt[key])
    _check_halo_reach(receiver, mesh, (0, 1), differentiation_order//2)
    if _same_tiling(receiver, mesh):
        # tile order (possibly drifted since the sort: strays are handled)
        mesh.gather_kick_tiled(receiver.pos, receiver.mom, receiver.tile_table,
                               differentiation_order, factor)
    else:
        mesh.gather_kick(receiver.pos, receiver.mom, differentiation_order, factor)


def group_components(components, gridsizes, gridsizes_order=(), split_representations=True):
    """mesh.py:714-790: {gridsize: {representation: [components]}} (or {gridsize: [...]}),
    grid sizes ordered as requested — an Ellipsis in `gridsizes_order` stands for every
    grid size not listed."""
    order = list(gridsizes_order)
    if ... not in order:
        order.append(...)
    if order.count(...) != 1:
        raise ConceptGPUError(f'group_components() got gridsizes_order = {order}')
    present = []
    for g in gridsizes:
        if g not in present:
            present.append(g)
    i = order.index(...)
    listed = [g for g in order if g is not ...]
    rest = [g for g in present if g not in listed]
    final = [g for g in order[:i] if g in present] + rest + [g for g in order[i + 1:]
                                                           if g in present]
    groups = collections.OrderedDict()
    for g in final:
        members = [c for c, gc in zip(components, gridsizes) if gc == g]
        if split_representations:
            groups[g] = collections.OrderedDict()
            for c in members:
                groups[g].setdefault(c.representation, []).append(c)
        else:
            groups[g] = members
    return groups


def lattice_shifts(kind, cell_centered=True):
    """Lattice (mesh.py:77-182): the primitive simple-cubic sub-lattices of 'sc', 'bcc'
    (body-centred) and 'fcc' (face-centred) as shifts in grid units."""
    kind = (kind or 'sc').lower()
    if 'simple' in kind:
        kind = 'sc'
    elif 'body' in kind:
        kind = 'bcc'
    elif 'face' in kind:
        kind = 'fcc'
    a = (1 - 2*cell_centered)*0.5
    table = {'sc': [(0, 0, 0)],
             'bcc': [(0, 0, 0), (a, a, a)],
             'fcc': [(0, 0, 0), (0, a, a), (a, 0, a), (a, a, 0)]}
    if kind not in table:
        raise ConceptGPUError(f'Unrecognized lattice "{kind}" ∉ {set(table)}')
    return table[kind]


def particle_mesh_general(receivers, suppliers, gridsize_global, quantity, force, method,
                          potential, interpolation_order, deconvolve_upstream,
                          deconvolve_downstream, interlace_upstream, interlace_downstream,
                          ᔑdt, ᔑdt_key):
    """interactions.py:1985-2335 with interpolate_upstream (mesh.py:492-635) and
    add_upstream_to_global_slabs (mesh.py:654-711), step for step, on one GPU: one mesh
    context per (grid size, role) plays the reference's named slabs.  Built: particle and
    fluid suppliers / receivers, upstream / downstream grid sizes
    different from the global one (row 1b: copy_modes), interpolation orders NGP / CIC /
    TSC / PCS, interlacing on 'bcc' / 'fcc' lattices, finite-difference (2, 4) and
    Fourier-space (0) gradients (row 3)."""
    p = receivers[0].params
    boxsize = p.boxsize
    dev = receivers[0].device
    if not 1 <= interpolation_order <= 4:
        raise ConceptGPUError(
            f'interpolate_particles() called with order = {interpolation_order} '
            f'∉ {{1 (NGP), 2 (CIC), 3 (TSC), 4 (PCS)}}')
    shifts_upstream = lattice_shifts(interlace_upstream, p.cell_centered)
    gs_up = [s.potential_gridsizes[force][method].upstream for s in suppliers]
    gs_down = [r.potential_gridsizes[force][method].downstream for r in receivers]
    for c, g in list(zip(suppliers, gs_up)) + list(zip(receivers, gs_down)):
        if c.representation == 'fluid' and c.gridsize != g:
            raise ConceptGPUError(
                f'add_fluid_to_grid() got component with global grid size {c.gridsize} and '
                f'non-matching grid of global grid size {g}')

    def mesh_for(gridsize, role):
        return get_mesh(gridsize, boxsize, p.nghosts, p.cell_centered, 2, dev, role=role)
    # interactions.py:2049-2080: which deconvolutions are promoted to the global one
    only_particle_suppliers = all(s.representation == 'particles' for s in suppliers)
    only_particle_receivers = all(r.representation == 'particles' for r in receivers)
    deconv_order_global = 0
    if (deconvolve_upstream and only_particle_suppliers
            and all(g == gridsize_global for g in gs_up)):
        deconvolve_upstream = False
        deconv_order_global += 1
    if (deconvolve_downstream and only_particle_receivers
            and all(g == gridsize_global for g in gs_down)):
        deconvolve_downstream = False
        deconv_order_global += 1
    deconv_order_global *= interpolation_order
    # ---- interpolate_upstream (mesh.py:571-620): per upstream grid size (the global one
    # first), fluids then particles (once per sub-lattice); each upstream slab is added
    # onto the global one in Fourier space (add_upstream_to_global_slabs, mesh.py:654-711)
    slab_global = None

    def upstream_mesh(gridsize_upstream):
        if slab_global is None and gridsize_upstream == gridsize_global:
            return mesh_for(gridsize_global, 'global')
        return mesh_for(gridsize_upstream, 'upstream')

    def add_to_global(up, deconv_order, nlattice=1, shift=(0, 0, 0)):
        nonlocal slab_global
        if slab_global is None and up.gridsize == gridsize_global:
            slab_global = up.fourier_operate(deconv_order, nlattice, shift)
        elif slab_global is None:
            slab_global = mesh_for(gridsize_global, 'global')
            slab_global.copy_modes_from(up, deconv_order, nlattice, shift, operation='=')
        else:
            slab_global.copy_modes_from(up, deconv_order, nlattice, shift, operation='+=')
    groups = group_components(suppliers, gs_up, [gridsize_global, ...])
    for gridsize_upstream, group in groups.items():
        fft_factor = float(gridsize_upstream)**(-3)  # mesh.py:582
        fluid_components = group.get('fluid', [])
        particle_components = group.get('particles', [])
        if fluid_components:
            up = upstream_mesh(gridsize_upstream)
            for i, fluid in enumerate(fluid_components):
                # add_fluid_to_grid, quantity 'a²ρ' (mesh.py:1712-1717)
                factor = fft_factor
                factor *= ᔑdt['a**(-3*w_eff-1)', fluid.name]/ᔑdt['1']
                up.fluid_add(fluid.ϱ, factor, '=' if i == 0 else '+=')
            up.fft_forward()
            up.nullify_nyquist()
            add_to_global(up, 0)
        for shift in (shifts_upstream if particle_components else ()):
            up = upstream_mesh(gridsize_upstream)
            simple = interpolation_order == 2 and shift == (0, 0, 0)
            # tile-sorted suppliers first: the LDS-tiled deposit assigns the mesh (no
            # zero-fill pass), later ones accumulate
            ordered = sorted(particle_components,
                             key=lambda c: not (simple and c.tiles_exact and _same_tiling(c, up)))
            started = False
            aligned = all(_aligned(c, up) for c in particle_components)
            cloud = {1: (0, 1), 2: (0, 1), 3: (1, 2), 4: (1, 2)}[interpolation_order]
            if shift != (0, 0, 0):
                cloud = (cloud[0] + 1, cloud[1] + 1)
            if up.dist and up.nprocs > 1 and max(cloud) > up.ghost_layers:
                raise ConceptGPUError('interpolation reaches beyond the 3 halo layers')
            if not aligned:
                for c in particle_components:
                    _check_halo_reach(c, up, cloud, 0)
                up.zero()
                started = True
            for supplier in ordered:
                contribution = _particle_contribution(supplier, ᔑdt, fft_factor,
                                                      gridsize_upstream, boxsize)
                if simple and supplier.tiles_exact and _same_tiling(supplier, up):
                    up.deposit_tiled(supplier.pos, supplier.tile_table, contribution,
                                     accumulate=started)
                    started = True
                    continue
                if not started:
                    up.zero()
                    started = True
                if simple:
                    up.deposit(supplier.pos, contribution)
                else:
                    up.deposit_general(supplier.pos, contribution, interpolation_order, shift)
            # communicate_ghosts(grid, '+=') (mesh.py:609)
            up.fold_ghosts(general=not (simple and aligned))
            up.fft_forward()
            up.nullify_nyquist()
            add_to_global(up, interpolation_order*int(bool(deconvolve_upstream)),
                          len(shifts_upstream), shift)
    # ---- potential (interactions.py:2092-2120)
    C, long_range, E = _potential_constants(p, potential, gridsize_global)
    slab_global.poisson_kernel(deconv_order_global, C, long_range, E)
    # ---- downstream (interactions.py:2124-2332): per downstream grid size, the global one
    # last (its slab may then be consumed)
    groups = group_components(receivers, gs_down, [..., gridsize_global])
    for gridsize_downstream, group in groups.items():
        if gridsize_downstream == gridsize_global:
            slab_downstream = slab_global
        else:
            slab_downstream = mesh_for(gridsize_downstream, 'downstream')
            slab_downstream.copy_modes_from(slab_global, operation='=')

        def apply_force(grid, dim, subgroup, representation, shift, differentiation_order):
            """apply_particle_mesh_force (interactions.py:2359-2402); `grid` holds the
            potential when differentiation_order is 2 or 4, else the force component"""
            for receiver in subgroup:
                key = (ᔑdt_key[0], receiver.name) if isinstance(ᔑdt_key, tuple) else ᔑdt_key
                if representation == 'fluid':
                    grid.fluid_kick(receiver.J[dim], receiver.ϱ, receiver.𝒫, dim,
                                    differentiation_order, -ᔑdt[key], p.light_speed**(-2))
                    continue
                factor = receiver.mass*(-ᔑdt[key])
                force_grid = grid
                if differentiation_order:
                    force_grid = mesh_for(gridsize_downstream, 'force')
                    force_grid.diff_from(grid, dim, differentiation_order)
                    force_grid.fill_ghosts()  # mesh.py:5026-5028
                force_grid.gather_scalar(receiver.pos, receiver.mom, dim, interpolation_order,
                                         shift, factor)
        for representation in ('fluid', 'particles'):
            if representation not in group:
                continue
            at_last_representation = representation == 'particles' or 'particles' not in group
            deconv_order_downstream = interpolation_order*int(
                representation == 'particles' and bool(deconvolve_downstream))
            shifts_downstream = lattice_shifts(
                interlace_downstream if representation == 'particles' else 'sc', p.cell_centered)
            differentiations = [r.potential_differentiations[force][method]
                                for r in group[representation]]
            for d in differentiations:
                if d not in (0, 1, 2, 4, 6, 8):
                    raise ConceptGPUError(
                        f'diff_domaingrid() called with order = {d} ∉ {{1, 2, 4, 6, 8}}')
                if (d + 1)//2 > p.nghosts:
                    raise ConceptGPUError(
                        f'differentiation order {d} needs nghosts >= {(d + 1)//2} '
                        f'(commons.py:4428-4430), got {p.nghosts}')
            subgroups = group_components(group[representation], differentiations,
                                         sorted(differentiations, reverse=True),
                                         split_representations=False)
            for differentiation_order, subgroup in subgroups.items():
                at_last_order = differentiation_order == min(differentiations)
                for li, shift in enumerate(shifts_downstream):
                    mutate_ok = (li == len(shifts_downstream) - 1 and at_last_representation
                                 and at_last_order)

                    def working_slab(may_mutate):
                        if may_mutate:
                            return slab_downstream  # nobody needs it afterwards
                        slab = mesh_for(gridsize_downstream, 'subgroup')
                        slab.copy_from(slab_downstream)
                        return slab
                    if differentiation_order == 0:
                        # Fourier-space differentiation (interactions.py:2229-2268)
                        for dim in range(3):
                            slab = working_slab(mutate_ok and dim == 2)
                            slab.fourier_operate(deconv_order_downstream, len(shifts_downstream),
                                                 shift, diff_dim=dim)
                            slab.poisson_backward()
                            slab.fill_ghosts()
                            apply_force(slab, dim, subgroup, representation, shift, 0)
                        continue
                    slab = working_slab(mutate_ok)
                    slab.fourier_operate(deconv_order_downstream, len(shifts_downstream), shift)
                    slab.poisson_backward()
                    slab.fill_ghosts()  # communicate_ghosts(grid, '=') (interactions.py:2307)
                    simple = (interpolation_order == 2 and shift == (0, 0, 0)
                              and differentiation_order in (2, 4))  # the fused gather-kick
                    for receiver in (subgroup if representation == 'particles' and simple
                                     else ()):
                        _kick_particles(slab, receiver, force, method, ᔑdt, ᔑdt_key)
                    if representation == 'fluid' or not simple:
                        for dim in range(3):
                            apply_force(slab, dim, subgroup, representation, shift,
                                        differentiation_order)


register('gravity', ['ppnonperiodic', 'pp', 'p3m', 'pm'], 'gravitational')


Pairing = collections.namedtuple('Pairing', ('force', 'method', 'receivers', 'suppliers'))
methods_implemented = ('ppnonperiodic', 'pp', 'p3m', 'pm')


def find_interactions(components, interaction_type='any', instantaneous='both'):
    """Which (force, method, receivers, suppliers) calls one kick consists of
    (interactions.py:2456-2636).  Every component with the force supplies; it receives
    through its own method.  Fluid suppliers of a non-PM interaction are split off into
    a PM interaction of their own; interactions with equal receivers (or equal suppliers)
    are merged; `interaction_type` 'long-range' keeps PM and P³M, 'short-range' everything
    but PM."""
    in_use = collections.defaultdict(set)
    for c in components:
        for force, method in c.forces.items():
            in_use[force].add(method)
    for force, methods in in_use.items():
        info = interactions_registered.get(force)
        if info is None:
            raise ConceptGPUError(f'Force "{force}" is not implemented')
        for method in methods:
            if not method:
                continue
            if method not in methods_implemented:
                raise ConceptGPUError(f'Force method "{method}" not recognised')
            if method not in info.methods:
                raise ConceptGPUError(
                    f'Method "{method}" for force "{force}" is not implemented. '
                    f'Did you mean one of {info.methods}?')
    todo = []
    for force, info in interactions_registered.items():
        for method in info.methods:
            if method not in in_use.get(force, ()):
                continue
            sup = [c for c in components if force in c.forces]
            rec = [c for c in sup if c.forces[force] == method]
            todo.append(Pairing(force, method, rec, sup))

    def simplify(lst):
        changed = True
        while changed:
            changed = False
            # fluids supply through the mesh only
            for i, it in enumerate(lst):
                if it.method == 'pm':
                    continue
                fluid = next((c for c in it.suppliers if c.representation == 'fluid'), None)
                if fluid is not None:
                    it.suppliers.remove(fluid)
                    lst.insert(i + 1, Pairing(it.force, 'pm', list(it.receivers), [fluid]))
                    changed = True
                    break
            if changed:
                continue
            lst[:] = [it for it in lst if it.receivers and it.suppliers]
            for i, a_ in enumerate(lst):
                for j in range(i + 1, len(lst)):
                    b_ = lst[j]
                    if (a_.force, a_.method) != (b_.force, b_.method):
                        continue
                    same_r = set(a_.receivers) == set(b_.receivers)
                    same_s = set(a_.suppliers) == set(b_.suppliers)
                    if same_r and not same_s:
                        for c in b_.suppliers:
                            if c not in a_.suppliers:
                                a_.suppliers.insert(0, c)
                    elif same_s and not same_r:
                        for c in b_.receivers:
                            if c not in a_.receivers:
                                a_.receivers.insert(0, c)
                    else:
                        continue
                    lst.pop(j)
                    changed = True
                    break
                if changed:
                    break
        return lst
    simplify(todo)
    if 'long' in interaction_type:
        for it in todo:
            if it.method not in {'pm', 'p3m'}:
                it.receivers[:] = []
        simplify(todo)
    elif 'short' in interaction_type:
        for it in todo:
            if it.method == 'pm':
                it.receivers[:] = []
        simplify(todo)
    elif 'any' not in interaction_type:
        raise ConceptGPUError(f'find_interactions(): Unknown interaction_type "{interaction_type}"')
    if 'True' in str(instantaneous) or 'False' in str(instantaneous):
        want = 'True' in str(instantaneous)
        for it in todo:
            if interactions_registered[it.force].instantaneous != want:
                it.receivers[:] = []
        simplify(todo)
    elif 'both' not in str(instantaneous):
        raise ConceptGPUError(f'find_interactions(): Unknown instantaneous value "{instantaneous}"')
    return todo


def gravity(method, receivers, suppliers, ᔑdt, interaction_type, printout):
    """interactions.py:2838-2961.  'pm' and the long-range part of 'p3m' run on
    the GPU mesh; the short-range part runs the tile sweep (shortrange.py)."""
    force = 'gravity'
    if method in {'pm', 'p3m'}:
        potential_specs = get_potential_specs(force, method, receivers, suppliers)
        quantity = 'a²ρ'
        ᔑdt_key = ('a**(-3*w_eff)', 'component')
    if method == 'pm':
        if printout:
            print(f'Executing gravitational interaction for '
                  f'{", ".join(c.name for c in receivers)} via the PM method ...')
        particle_mesh(
            receivers, suppliers, potential_specs.gridsize, quantity, force, method, 'gravity',
            potential_specs.interpolation_order,
            potential_specs.deconvolve.upstream, potential_specs.deconvolve.downstream,
            potential_specs.interlace.upstream, potential_specs.interlace.downstream,
            ᔑdt, ᔑdt_key)
    elif method == 'p3m':
        if 'any' in interaction_type or 'long' in interaction_type:
            particle_mesh(
                receivers, suppliers, potential_specs.gridsize, quantity, force, method,
                'gravity long-range', potential_specs.interpolation_order,
                potential_specs.deconvolve.upstream, potential_specs.deconvolve.downstream,
                potential_specs.interlace.upstream, potential_specs.interlace.downstream,
                ᔑdt, ᔑdt_key)
        if 'any' in interaction_type or 'short' in interaction_type:
            from .shortrange import component_component
            component_component(force, receivers, suppliers, ᔑdt, potential_specs.gridsize)
    elif method in {'pp', 'ppnonperiodic'}:
        # direct summation (gravity.py:121-206 with the Ewald correction of ewald.py,
        # gravity.py:491-560 without); SURVEY.md §8f row 4
        from .shortrange import component_component_pp
        component_component_pp(force, receivers, suppliers, ᔑdt, periodic=(method == 'pp'))
    else:
        raise ConceptGPUError(f'gravity() was called with the "{method}" method')
