"""Known-answer tests that need no oracle at all: the symmetric set-ups
of the reference's own integration tests, replayed through gravity() on the GPU with the
reference's tolerances or tighter ones.

K2  test/multicomponent, 'domain' subtest: six particles placed symmetrically around the box
    centre (gen_ic.py:14-20) under non-periodic PP — every kick points at the centre with the
    analytic magnitude G m^2 (1/4 + sqrt(2))/d^2.
K3  test/kick_pp_without_ewald: 2 x 4 particles (gen_ic.py:13-19) — the four of a group must
    keep identical x (analyze.py: relative 1e-9).
K4  test/fluid_gravity: a sine-wave density along x (gen_ic.py) — whatever gravity does to a
    fluid, yz-slices stay uniform (analyze.py: std of a slice <= 1e-6 std of the grid).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pp_component(pos, mass, boxsize, method):
    from concept_amd import commons
    from concept_amd.species import Component
    commons.load_params({
        'boxsize': boxsize,
        'select_forces': {'matter': {'gravity': method}},
        'select_softening_length': {'matter': '0.03*boxsize/cbrt(N)'}})
    c = Component('matter', 'matter', N=pos.shape[0], mass=mass)
    c.populate(pos, 'pos')
    return c


def _pp_kick(c, method, dt):
    from concept_amd import commons, interactions
    n_rungs = int(commons.params.N_rungs)
    sdt_rungs = {('a**(-3*w_eff₀-3*w_eff₁-1)', 'matter', 'matter'):
                 np.full(3*n_rungs - 1, dt)}
    c.nullify_Δ('mom')
    interactions.gravity(method, [c], [c], sdt_rungs, 'any', False)
    return c.host('Δmom')


def test_k2_six_symmetric_particles_fall_to_the_centre():
    from concept_amd import commons
    L, T, dt = 1.0, 3.0, 0.01
    d = 0.4*L
    pos = []
    for dim in range(3):
        for sign in (-1, +1):
            pos.append(0.5*L*np.ones(3) + np.roll([sign*d, 0, 0], dim))
    pos = np.array(pos)
    _pp_component(pos, 1.0, L, 'ppnonperiodic')                 # loads the parameters
    G = float(commons.params.G_Newton)
    mass = np.pi**2/((2 + 8*np.sqrt(2))*G)*d**3/T**2          # gen_ic.py: collide after T
    c = _pp_component(pos, mass, L, 'ppnonperiodic')
    dmom = _pp_kick(c, 'ppnonperiodic', dt)
    # separations (0.8 L, 0.57 L) are far outside the softening radius: pure Newton
    expect = dt*G*mass**2*(0.25 + np.sqrt(2))/d**2
    radial = (0.5*L - pos)/d                                    # unit vectors to the centre
    along = (dmom*radial).sum(1)
    assert np.abs(along/expect - 1).max() < 1e-12
    assert np.abs(dmom - along[:, None]*radial).max() < 1e-13*expect
    assert np.abs(dmom.sum(0)).max() < 1e-13*expect


def test_k3_two_groups_of_four_keep_identical_x():
    L, dt = 21.0, 0.05
    x = np.array([0.26]*4 + [0.74]*4)*L
    y = np.array([0.25, 0.25, 0.75, 0.75]*2)*L
    z = np.array([0.25, 0.75, 0.75, 0.25]*2)*L
    pos = np.stack([x, y, z], 1)
    c = _pp_component(pos, 1.0e3, L, 'ppnonperiodic')
    dmom = _pp_kick(c, 'ppnonperiodic', dt)
    kx = dmom[:, 0]
    assert kx[:4].min() > 0 and kx[4:].max() < 0               # the groups attract each other
    for grp in (kx[:4], kx[4:]):
        assert np.std(grp) <= 1e-9*np.abs(grp).mean()          # the reference's bar
        assert np.ptp(grp) <= 1e-13*np.abs(grp).mean()         # ours
    assert abs(kx[:4].sum() + kx[4:].sum()) <= 1e-13*np.abs(kx).sum()
    # the y and z kicks mirror each other inside a group
    assert np.abs(np.abs(dmom[:, 1]) - np.abs(dmom[0, 1])).max() <= 1e-13*np.abs(dmom[0, 1])
    assert np.abs(np.abs(dmom[:, 2]) - np.abs(dmom[0, 2])).max() <= 1e-13*np.abs(dmom[0, 2])


@pytest.mark.parametrize('gridsize,diff_order', [(16, 2), (32, 4)])
def test_k4_sine_wave_fluid_stays_uniform_in_yz(gridsize, diff_order):
    from concept_amd import commons, interactions
    from concept_amd.species import Component
    L, dt = 64.0, 0.1
    commons.load_params({
        'boxsize': L,
        'potential_options': {
            'gridsize': {'global': {'gravity': {'pm': gridsize}}},
            'differentiation': {'fluid': {'gravity': {'pm': diff_order}}}},
        'select_forces': {'fluid': {'gravity': 'pm'}}})
    fl = Component('fluid', 'matter', gridsize=gridsize, boltzmann_order=1)
    xc = (np.arange(gridsize) + 0.5)*L/gridsize
    rho = np.broadcast_to((2 + np.sin(2*np.pi*xc/L))[:, None, None], (gridsize,)*3).copy()
    fl.populate(rho, 'ϱ')
    fl.populate(np.zeros_like(rho), '𝒫')
    for d in range(3):
        fl.populate(np.zeros_like(rho), 'J', d)
    sdt = {'1': dt, ('a**(-3*w_eff)', 'fluid'): dt, ('a**(-3*w_eff-1)', 'fluid'): dt}
    found = interactions.find_interactions([fl], 'long-range')
    assert [(f, m) for f, m, _, _ in found] == [('gravity', 'pm')]
    for force, method, receivers, suppliers in found:
        getattr(interactions, force)(method, receivers, suppliers, sdt, 'long-range', False)
    J = fl.host('J')
    assert np.array_equal(fl.host('ϱ'), rho)
    Jx = J[0]
    sigma = np.std(Jx)
    assert sigma > 0
    eps = np.finfo(float).eps
    for grid in J:
        for i in range(gridsize):
            # analyze.py:78-95
            assert np.std(grid[i]) <= max(1e-6*sigma, 10*gridsize**2*eps)
    # no force across the wave, and the kick follows -ϱ dφ/dx: towards the crest at x = L/4
    assert np.abs(J[1]).max() <= 1e-12*np.abs(Jx).max()
    assert np.abs(J[2]).max() <= 1e-12*np.abs(Jx).max()
    profile = Jx[:, 0, 0]
    crest = int(np.argmax(rho[:, 0, 0]))
    assert profile[(crest - gridsize//4) % gridsize] > 0 > profile[(crest + gridsize//4) % gridsize]


@pytest.mark.parametrize('ncomponents', [1, 2, 5])
def test_k1_lattice_stays_put_across_components(ncomponents):
    k1_lattice(ncomponents)


def k1_lattice(ncomponents, n_lin=12, gridsize=36):
    """test/multicomponent, 'tile' subtest (gen_ic.py:21-33, param:25-42): 12^3 particles on
    a perfect cubic lattice, dealt round-robin to 1, 2 or 5 components, P3M with the mesh
    of that test (36): long- and short-range kicks cancel by symmetry for every particle —
    which they only do if every pair across tiles, components and the periodic boundary is
    counted exactly once."""
    from concept_amd import commons, interactions
    from concept_amd.species import Component
    L, dt = float(gridsize), 0.1
    commons.load_params({
        'boxsize': L,
        'potential_options': {'gridsize': {'gravity': {'p3m': gridsize}}},
        'select_forces': {'matter': {'gravity': 'p3m'}},
        'select_softening_length': {'matter': f'0.03*boxsize/{n_lin}'}})
    ax = (0.5 + np.arange(n_lin))*L/n_lin
    pos = np.stack(np.meshgrid(ax, ax, ax, indexing='ij'), -1).reshape(-1, 3)
    G = float(commons.params.G_Newton)
    mass = 10*np.pi**2/((2 + 8*np.sqrt(2))*G)*(L/n_lin)**3      # gen_ic.py:36-43 with T = 1
    comps = []
    for n in range(ncomponents):
        sub = pos[n::ncomponents]
        c = Component(f'component{n}', 'matter', N=sub.shape[0], mass=mass)
        c.populate(sub, 'pos')
        c.populate(np.zeros_like(sub), 'mom')
        c.nullify_Δ('mom')
        comps.append(c)
    n_rungs = int(commons.params.N_rungs)
    sdt = {'1': dt}
    sdt_rungs = {}
    for r in comps:
        sdt['a**(-3*w_eff)', r.name] = dt
        sdt['a**(-3*w_eff-1)', r.name] = dt
        for s in comps:
            sdt_rungs['a**(-3*w_eff₀-3*w_eff₁-1)', r.name, s.name] = np.full(3*n_rungs - 1, dt)
    found = interactions.find_interactions(comps, 'any')
    assert len(found) == 1 and found[0][:2] == ('gravity', 'p3m')
    _, _, receivers, suppliers = found[0]
    assert len(receivers) == len(suppliers) == ncomponents
    interactions.gravity('p3m', receivers, suppliers, sdt, 'long-range', False)
    interactions.gravity('p3m', receivers, suppliers, sdt_rungs, 'short-range', False)
    pair = G*mass**2*dt/(L/n_lin)**2           # the kick one nearest neighbour alone would give
    for c in comps:
        assert np.abs(c.host('mom')).max() <= 1e-10*pair       # long range
        assert np.abs(c.host('Δmom')).max() <= 1e-10*pair      # short range
