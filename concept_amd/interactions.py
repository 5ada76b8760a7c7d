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


def particle_mesh(receivers, suppliers, gridsize_global, quantity, force, method, potential,
                  interpolation_order, deconvolve_upstream, deconvolve_downstream,
                  interlace_upstream, interlace_downstream, ᔑdt, ᔑdt_key):
    """interactions.py:1985-2335 for particle components whose upstream and
    downstream grid sizes equal the global one (the default configuration)."""
    if not receivers or not suppliers:
        return
    if potential not in {'gravity', 'gravity long-range'}:
        raise ConceptGPUError(
            f'particle_mesh() got potential "{potential}" ∉ {{"gravity", "gravity long-range"}}')
    if quantity != 'a²ρ':
        raise ConceptGPUError(f'particle_mesh(): quantity "{quantity}" is not on the gravity path')
    for c in list(receivers) + list(suppliers):
        if c.representation != 'particles':
            raise ConceptGPUError(f'{c.name}: fluid components are not built (SURVEY.md §8f)')
    for s in suppliers:
        if s.potential_gridsizes[force][method].upstream != gridsize_global:
            raise ConceptGPUError('upstream grid size ≠ global grid size is not built '
                                  '(copy_modes, SURVEY.md §8f-1b)')
    for r in receivers:
        if r.potential_gridsizes[force][method].downstream != gridsize_global:
            raise ConceptGPUError('downstream grid size ≠ global grid size is not built '
                                  '(copy_modes, SURVEY.md §8f-1b)')
    if interpolation_order != 2:
        raise ConceptGPUError(f'interpolation order {interpolation_order}: only CIC (2) is built')
    if (interlace_upstream, interlace_downstream) != ('sc', 'sc'):
        raise ConceptGPUError('interlacing is not built (SURVEY.md §8f-3)')
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
    for supplier in ordered:
        # mesh.py:1550-1573
        contribution = ᔑdt['a**(-3*w_eff-1)', supplier.name]/ᔑdt['1']
        contribution *= supplier.mass
        contribution_factor = fft_factor*(gridsize_global/boxsize)**3
        contribution *= contribution_factor
        if supplier.tiles_exact and supplier.tile_mesh is mesh:
            mesh.deposit_tiled(supplier.pos, supplier.tile_table, contribution,
                               accumulate=mesh_started)
        else:
            if not mesh_started:
                mesh.zero()
            mesh.deposit(supplier.pos, contribution)
        mesh_started = True
    # interactions.py:2092-2118 and :2302
    C = -boxsize**2*p.G_Newton/π
    if potential == 'gravity':
        mesh.poisson_solve(deconv_order_global, C, False, 0.0)
    else:
        scale = commons.resolve_shortrange(p, gridsize_global)['scale']
        E = -(2*π/boxsize*scale)**2
        mesh.poisson_solve(deconv_order_global, C, True, E)
    # interactions.py:2311-2332 via apply_particle_mesh_force (:2359-2387)
    for receiver in receivers:
        key = (ᔑdt_key[0], receiver.name) if isinstance(ᔑdt_key, tuple) else ᔑdt_key
        differentiation_order = receiver.potential_differentiations[force][method]
        if differentiation_order == 0:
            raise ConceptGPUError('Fourier-space differentiation (order 0) is not built '
                                  '(SURVEY.md §8f-3)')
        factor = receiver.mass*(-ᔑdt[key])
        if receiver.tile_table is not None and receiver.tile_mesh is mesh:
            # tile order (possibly drifted since the sort: strays are handled)
            mesh.gather_kick_tiled(receiver.pos, receiver.mom, receiver.tile_table,
                                   differentiation_order, factor)
        else:
            mesh.gather_kick(receiver.pos, receiver.mom, differentiation_order, factor)


register('gravity', ['ppnonperiodic', 'pp', 'p3m', 'pm'], 'gravitational')


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
        raise ConceptGPUError(f'gravity(): the "{method}" method is not on the GPU path '
                              '(direct summation; SURVEY.md §8f-4)')
    else:
        raise ConceptGPUError(f'gravity() was called with the "{method}" method')
