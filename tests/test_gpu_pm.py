"""GPU parity tests of the PM path (run with -m gpu on an MI355X).

Every test calls the HIP library through its C ABI (concept_amd.lib /
concept_amd.mesh) and checks against (a) the golden vectors produced by the
imported reference and (b) the CPU oracle on the same seeded inputs.

Bars: CIC grid indices bit-exact; drift bit-exact; deposited
density, k-space potential, potential grid and per-particle kick <= 1e-12 of
the field rms for a single kick (summation order of the scatter and the FFT
backend differ from the reference's; its own compiled-vs-pure-Python bar is
1e-10, test/pure_python_pm/analyze.py:125)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12
PM_CASES = ['pm_n8_g16', 'pm_n16_g32', 'pm_edge_g16', 'pm_n8_g16_d4']


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch


def rms(a):
    return float(np.sqrt((np.asarray(a)**2).mean()))


def fold_ghosts(grid, g):
    """communicate_ghosts(grid, '+=') on one rank + interior, in numpy."""
    n = grid.shape[0] - 2*g
    out = np.zeros((n, n, n))
    idx = (np.arange(grid.shape[0]) - g) % n
    np.add.at(out, (idx[:, None, None], idx[None, :, None], idx[None, None, :]), grid)
    return out


def setup_case(torch, g):
    from concept_amd.mesh import PotentialMesh
    from oracle import oracle
    L, N = float(g['boxsize']), int(g['gridsize'])
    mesh = PotentialMesh(N, L, nghosts=int(g['nghosts']))
    pos = torch.tensor(g['pos_in'], device='cuda')
    mom = torch.tensor(g['mom_in'], device='cuda')
    contribution = oracle.deposit_contribution(float(g['mass']), float(g['dt_dens']),
                                               float(g['dt_1']), N, L)
    sc = float(g['shortrange_scale']) if 'shortrange_scale' in g else None
    C, E = oracle.poisson_constants(L, float(g['G_Newton']), sc)
    return mesh, pos, mom, contribution, C, E, sc is not None


@pytest.mark.parametrize('name', PM_CASES + ['pm_n32_g64', 'p3m_n8_g32'])
def test_cic_indices_bit_exact(torch_cuda, golden, name):
    g = golden(name)
    mesh, pos, mom, *_ = setup_case(torch_cuda, g)
    assert np.array_equal(mesh.cic_indices(pos, False).cpu().numpy(), g['cic_index_deposit'])
    assert np.array_equal(mesh.cic_indices(pos, True).cpu().numpy(), g['cic_index_gather'])


@pytest.mark.parametrize('name', PM_CASES + ['p3m_n8_g32'])
def test_pm_intermediates(torch_cuda, golden, name):
    g = golden(name)
    mesh, pos, mom, contribution, C, E, lr = setup_case(torch_cuda, g)
    N, ng = int(g['gridsize']), int(g['nghosts'])
    # A1/A2 deposit (+ fold)
    mesh.zero()
    mesh.deposit(pos, contribution)
    dens = mesh.fetch_real()[:, :, :N]
    ref = fold_ghosts(g['grid_deposit'], ng)
    assert np.abs(dens - ref).max() <= TOL*rms(ref)
    # total mass is conserved to rounding
    assert abs(dens.sum() - ref.sum()) <= 1e-12*abs(ref.sum())
    # A4/A5 forward FFT + Nyquist nullification (kernel applied separately below)
    mesh.poisson_forward(0, 1.0, False, 0.0, apply_kernel=False)
    dk = mesh.fetch_fourier()
    refk = g['slab_density_k']
    nyq = N//2
    keep = np.ones((N, N, N + 2), dtype=bool)
    keep[nyq, :, :] = False
    keep[:, nyq, :] = False
    keep[:, :, N:] = False
    assert np.abs(dk - refk)[keep].max() <= TOL*rms(refk)
    # A6 Poisson kernel: redo from the density
    mesh.zero()
    mesh.deposit(pos, contribution)
    mesh.poisson_forward(4, C, lr, E, apply_kernel=True)
    pk = mesh.fetch_fourier()
    refp = g['slab_potential_k']
    assert np.abs(pk - refp).max() <= TOL*rms(refp)
    assert pk[0, 0, 0] == 0 and pk[0, 0, 1] == 0
    assert not pk[nyq].any() and not pk[:, nyq].any() and not pk[:, :, N:].any()
    # A8 back to real space
    mesh.poisson_backward()
    phi = mesh.fetch_real()[:, :, :N]
    refphi = g['grid_potential'][ng:-ng, ng:-ng, ng:-ng]
    assert np.abs(phi - refphi).max() <= TOL*rms(refphi)


@pytest.mark.parametrize('name', PM_CASES + ['pm_n32_g64', 'p3m_n8_g32', 'p3m_n12_g36_lattice',
                                             'p3m_n16_g48_clustered'])
def test_pm_kick_vs_golden_and_oracle(torch_cuda, golden, name):
    from oracle import oracle
    g = golden(name)
    mesh, pos, mom, contribution, C, E, lr = setup_case(torch_cuda, g)
    mesh.zero()
    mesh.deposit(pos, contribution)
    mesh.poisson_solve(4, C, lr, E)
    mesh.gather_kick(pos, mom, int(g['diff_order']), float(g['mass'])*(-float(g['dt_kick'])))
    out = mom.cpu().numpy()
    kick_ref = g['mom_after_long'] - g['mom_in']
    scale = max(rms(kick_ref), 1e-300)
    if name == 'p3m_n12_g36_lattice':
        # perfect lattice: the kick vanishes by symmetry (test/multicomponent K1);
        # compare against the size of a single-cell force instead
        scale = float(g['mass'])*float(g['dt_kick'])*float(g['G_Newton'])*float(g['mass'])
    assert np.abs(out - g['mom_after_long']).max() <= TOL*scale + 4e-16*np.abs(g['mom_in']).max()
    # and the oracle run here on the same inputs
    mom_o = g['mom_in'].copy()
    sc = float(g['shortrange_scale']) if 'shortrange_scale' in g else None
    oracle.pm_long_range(g['pos_in'].copy(), mom_o, mass=float(g['mass']),
                         boxsize=float(g['boxsize']), gridsize=int(g['gridsize']),
                         G_Newton=float(g['G_Newton']), dt_1=float(g['dt_1']),
                         dt_dens=float(g['dt_dens']), dt_kick=float(g['dt_kick']),
                         diff_order=int(g['diff_order']), shortrange_scale=sc,
                         want_indices=False)
    assert np.abs(out - mom_o).max() <= TOL*scale + 4e-16*np.abs(g['mom_in']).max()


@pytest.mark.parametrize('name', PM_CASES + ['pm_n32_g64', 'p3m_n8_g32'])
def test_drift_bit_exact(torch_cuda, golden, name):
    torch = torch_cuda
    g = golden(name)
    mesh, *_ = setup_case(torch, g)
    pos = torch.tensor(g['drift_pos_in'] if 'drift_pos_in' in g else g['pos_in'], device='cuda')
    mom = torch.tensor(g['drift_mom_in'] if 'drift_mom_in' in g else g['mom_after_long'],
                       device='cuda')
    mesh.drift(pos, mom, float(g['drift_dt_over_mass']))
    assert np.array_equal(pos.cpu().numpy(), g['drift_pos_out'])


def test_drift_wrap_edges(torch_cuda):
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    from oracle import oracle
    L = 10.0
    mesh = PotentialMesh(8, L)
    pos = np.array([[0.0, 9.999999999999998, 5.0], [1e-17, 9.5, 0.25], [0.0, 0.0, 9.0]])
    mom = np.array([[-1e-17, 1.0, 25.0], [-1.0, 0.5, -30.25], [-0.0, 10.0, 1.0]])
    ref = oracle.drift(pos.copy(), mom, 1.0, L)
    p = torch.tensor(pos, device='cuda')
    mesh.drift(p, torch.tensor(mom, device='cuda'), 1.0)
    out = p.cpu().numpy()
    assert np.array_equal(out, ref)
    assert (out >= 0).all() and (out < L).all()


def test_tile_sort_is_a_permutation_in_tile_order(torch_cuda):
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    L, N, n = 50.0, 32, 20000
    mesh = PotentialMesh(N, L)
    rng = np.random.default_rng(3)
    pos = torch.tensor(rng.uniform(0, L, (n, 3)), device='cuda')
    mom = torch.tensor(rng.normal(size=(n, 3)), device='cuda')
    ids = torch.arange(n, device='cuda')
    po, mo, io = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    table = mesh.sort_particles(pos, mom, ids, po, mo, io)
    torch.cuda.synchronize()
    assert np.array_equal(np.sort(io.cpu().numpy()), np.arange(n))
    assert torch.equal(po, pos[io]) and torch.equal(mo, mom[io])
    T, nt = mesh.tile_extent, mesh.tiles_per_dim
    idx = mesh.cic_indices(po).cpu().numpy() - 2
    idx = np.mod(idx, N)//T
    key = (idx[:, 0]*nt + idx[:, 1])*nt + idx[:, 2]
    assert (np.diff(key) >= 0).all()
    # the tile table (8 buckets per tile) is the exclusive scan of the populations
    tab = table.cpu().numpy().astype(np.int64)
    counts = np.bincount(key, minlength=mesh.ntiles)
    assert np.array_equal(tab[::8], np.concatenate([[0], np.cumsum(counts)]))
    assert (np.diff(tab) >= 0).all() and tab[-1] == n
    # bucket f = 4*fx + 2*fy + fz, f? = lower cell is the last one of its tile
    cell = np.mod(mesh.cic_indices(po).cpu().numpy() - 2, N)
    last = (cell % T == T - 1).astype(np.int64)
    f = 4*last[:, 0] + 2*last[:, 1] + last[:, 2]
    assert (np.diff(key*8 + f) >= 0).all()


@pytest.mark.parametrize('name', PM_CASES + ['pm_n32_g64', 'p3m_n8_g32', 'p3m_n12_g36_lattice',
                                             'p3m_n16_g48_clustered'])
def test_tiled_kernels_vs_golden(torch_cuda, golden, name):
    """LDS-tiled deposit (assign mode) + tiled gather on tile-sorted particles:
    same parity bar as the direct kernels, compared through the ids."""
    torch = torch_cuda
    g = golden(name)
    mesh, pos, mom, contribution, C, E, lr = setup_case(torch, g)
    n = pos.shape[0]
    ids = torch.arange(n, device='cuda')
    po, mo, io = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    table = mesh.sort_particles(pos, mom, ids, po, mo, io)
    # poison the mesh: the assigning deposit must not depend on its previous content
    mesh.zero()
    mesh.deposit(pos, 1e30)
    mesh.deposit_tiled(po, table, contribution, accumulate=False)
    N, ng = int(g['gridsize']), int(g['nghosts'])
    if 'grid_deposit' in g:
        dens = mesh.fetch_real()[:, :, :N]
        ref = fold_ghosts(g['grid_deposit'], ng)
        assert np.abs(dens - ref).max() <= TOL*rms(ref)
    mesh.poisson_solve(4, C, lr, E)
    mesh.gather_kick_tiled(po, mo, table, int(g['diff_order']),
                           float(g['mass'])*(-float(g['dt_kick'])))
    out = np.empty((n, 3))
    out[io.cpu().numpy()] = mo.cpu().numpy()
    kick_ref = g['mom_after_long'] - g['mom_in']
    scale = max(rms(kick_ref), 1e-300)
    if name == 'p3m_n12_g36_lattice':
        scale = float(g['mass'])*float(g['dt_kick'])*float(g['G_Newton'])*float(g['mass'])
    assert np.abs(out - g['mom_after_long']).max() <= TOL*scale + 4e-16*np.abs(g['mom_in']).max()


def test_tiled_accumulate_and_strays(torch_cuda):
    """accumulate=1 adds onto an existing mesh (exact tile order required for the
    deposit); the tiled gather still kicks particles right after they have left
    their tile (drift without re-sort)."""
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    L, N, n = 64.0, 64, 50000
    mesh = PotentialMesh(N, L)
    rng = np.random.default_rng(5)
    pos = torch.tensor(rng.uniform(0, L, (n, 3)), device='cuda')
    mom = torch.tensor(rng.normal(0, 3.0, (n, 3)), device='cuda')  # several cells per drift
    po, mo = torch.empty_like(pos), torch.empty_like(mom)
    table = mesh.sort_particles(pos, mom, None, po, mo, None)
    # reference: direct kernels
    mesh.zero()
    mesh.deposit(po, 0.7)
    mesh.deposit(po, 0.3)
    ref = mesh.fetch_real()[:, :, :N].copy()
    mesh.zero()
    mesh.deposit(po, 0.7)
    mesh.deposit_tiled(po, table, 0.3, accumulate=True)
    out = mesh.fetch_real()[:, :, :N]
    assert np.abs(out - ref).max() <= 1e-13*np.abs(ref).max()
    mesh.drift(po, mo, 1.0)  # now many particles are outside their tile
    mesh.poisson_solve(4, -1.0, False, 0.0)
    k_direct = torch.zeros_like(mo)
    k_tiled = torch.zeros_like(mo)
    for order in (2, 4):
        k_direct.zero_()
        k_tiled.zero_()
        mesh.gather_kick(po, k_direct, order, -0.5)
        mesh.gather_kick_tiled(po, k_tiled, table, order, -0.5)
        assert torch.equal(k_direct, k_tiled)  # same arithmetic, same order: bit-exact


def test_empty_and_single_particle(torch_cuda):
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    mesh = PotentialMesh(16, 16.0)
    empty = torch.zeros((0, 3), dtype=torch.float64, device='cuda')
    mesh.zero()
    mesh.deposit(empty, 1.0)
    mesh.gather_kick(empty, empty.clone(), 2, 1.0)
    mesh.drift(empty, empty.clone(), 1.0)
    assert not mesh.fetch_real().any()
    # one particle at a cell centre deposits all its mass in one cell and feels no self-force
    pos = torch.tensor([[4.5, 7.5, 11.5]], dtype=torch.float64, device='cuda')
    mom = torch.zeros_like(pos)
    mesh.deposit(pos, 2.0)
    d = mesh.fetch_real()[:, :, :16]
    # weights carry the reference's (1 -+ machine_eps) guard factors: not exactly 1
    assert abs(d[4, 7, 11] - 2.0) < 1e-12 and abs(d.sum() - 2.0) < 1e-13
    mesh.poisson_solve(4, -1.0, False, 0.0)
    mesh.gather_kick(pos, mom, 2, 1.0)
    assert np.abs(mom.cpu().numpy()).max() < 1e-12


def test_errors_are_loud(torch_cuda):
    torch = torch_cuda
    from concept_amd.lib import ConceptGPUError
    from concept_amd.mesh import PotentialMesh
    with pytest.raises(ConceptGPUError):
        PotentialMesh(15, 1.0)  # odd grid
    mesh = PotentialMesh(16, 16.0)
    pos = torch.zeros((4, 3), dtype=torch.float64, device='cuda')
    with pytest.raises(ConceptGPUError):
        mesh.gather_kick(pos, pos.clone(), 3, 1.0)  # odd differentiation order
    with pytest.raises(ConceptGPUError):
        mesh.deposit(pos.float(), 1.0)  # wrong dtype


def test_gravity_api_pm(torch_cuda, golden):
    """The boundary itself: gravity('pm', [c], [c], ᔑdt, 'long-range', False)
    (interactions.py:2854) with the reference's parameter names."""
    from concept_amd import commons, interactions
    from concept_amd.species import Component
    g = golden('pm_n16_g32')
    commons.load_params({
        'boxsize': float(g['boxsize']),
        'potential_options': {'gridsize': {'gravity': {'pm': int(g['gridsize'])}}},
        'select_forces': {'matter': {'gravity': 'pm'}},
    })
    c = Component('matter', 'matter', N=int(g['N']), mass=float(g['mass']))
    for d, s in enumerate('xyz'):
        c.populate(g['pos_in'][:, d], 'pos' + s)
        c.populate(g['mom_in'][:, d], 'mom' + s)
    sdt = {'1': float(g['dt_1']), 'a**(-2)': float(g['dt_am2']),
           ('a**(-3*w_eff)', 'matter'): float(g['dt_kick']),
           ('a**(-3*w_eff-1)', 'matter'): float(g['dt_dens'])}
    interactions.gravity('pm', [c], [c], sdt, 'long-range', False)
    kick_ref = g['mom_after_long'] - g['mom_in']
    assert np.abs(c.host('mom') - g['mom_after_long']).max() <= TOL*rms(kick_ref)
    assert np.array_equal(c.host('pos'), g['pos_in'])  # gravity must not touch pos
    # drift through the Component API, then sorting must not change the physics
    c.drift(sdt)
    d = np.abs(c.host('pos') - g['drift_pos_out'])
    d = np.minimum(d, float(g['boxsize']) - d)
    assert d.max() <= 1e-13*float(g['boxsize'])
    c.tile_sort()
    before = c.host('mom')
    interactions.gravity('pm', [c], [c], sdt, 'long-range', False)
    c2 = Component('matter', 'matter', N=int(g['N']), mass=float(g['mass']))
    c2.populate(c.host('pos'), 'pos')
    c2.populate(before, 'mom')
    interactions.gravity('pm', [c2], [c2], sdt, 'long-range', False)
    k = c2.host('mom') - before
    assert np.abs(c.host('mom') - c2.host('mom')).max() <= TOL*rms(k)


def test_full_size_properties(torch_cuda):
    """BASELINE config-2 size (256^3 particles / 512^3 mesh): size-independent
    properties — total deposited mass, zero net momentum transfer, linearity
    of the kick in the supplier mass, and invariance under particle order."""
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    L, N, n = 512.0, 512, 256**3
    mesh = PotentialMesh(N, L)
    gen = torch.Generator(device='cuda').manual_seed(11)
    pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen)*L
    pos.clamp_(max=float(np.nextafter(L, 0)))
    mom = torch.zeros_like(pos)
    mesh.zero()
    mesh.deposit(pos, 1.0)
    dens = mesh.fetch_real()[:, :, :N]
    total = dens.sum()
    assert abs(total - n) <= 1e-9*n
    mesh.poisson_solve(4, -1.0, False, 0.0)
    mesh.gather_kick(pos, mom, 2, 1.0)
    k1 = mom.clone()
    # Newton's third law on the mesh: net momentum change ~ 0
    assert float(k1.sum(0).abs().max()) <= 1e-9*float(k1.abs().sum(0).max())
    # linearity: twice the contribution -> twice the kick
    mom.zero_()
    mesh.zero()
    mesh.deposit(pos, 2.0)
    mesh.poisson_solve(4, -1.0, False, 0.0)
    mesh.gather_kick(pos, mom, 2, 1.0)
    assert float((mom - 2*k1).abs().max()) <= 1e-11*float(k1.abs().max())
    # order invariance: sort the particles, kick again, compare through ids
    ids = torch.arange(n, device='cuda')
    po, mo, io = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    mom.zero_()
    table = mesh.sort_particles(pos, mom, ids, po, mo, io)
    mesh.deposit_tiled(po, table, 1.0, accumulate=False)
    dens2 = mesh.fetch_real()[:, :, :N]
    assert np.abs(dens2 - dens).max() <= 1e-12*np.abs(dens).max()
    mesh.poisson_solve(4, -1.0, False, 0.0)
    mesh.gather_kick_tiled(po, mo, table, 2, 1.0)
    assert float((mo - k1[io]).abs().max()) <= 1e-11*float(k1.abs().max())


@pytest.mark.parametrize('N', [16, 64, 128, 256])
def test_handwritten_fft_vs_numpy_and_rocfft(torch_cuda, N, monkeypatch):
    """The hand-written FFT passes (power-of-two grids) against numpy's pocketfft
    — the reference's own pure-Python FFT (mesh.py:4035-4143) — and against the
    rocFFT backend; the fused solve against the unfused one."""
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    L = 100.0
    rng = np.random.default_rng(N)
    n = 4*N**2
    pos = torch.tensor(rng.uniform(0, L, (n, 3)), device='cuda')
    monkeypatch.delenv('CONCEPT_GPU_FFT', raising=False)
    mesh = PotentialMesh(N, L)
    mesh.zero()
    mesh.deposit(pos, 1.0)
    dens = mesh.fetch_real()[:, :, :N].copy()
    mesh.poisson_forward(0, 1.0, False, 0.0, apply_kernel=False)
    fk = mesh.fetch_fourier()
    ref = np.fft.rfftn(dens).transpose(1, 0, 2)
    out = fk[:, :, 0::2] + 1j*fk[:, :, 1::2]
    scale = np.sqrt((np.abs(ref)**2).mean())
    assert np.abs(out - ref).max() <= 1e-13*scale*np.log2(N)**2
    # forward then backward = N^3 * identity (both unnormalised)
    mesh.poisson_backward()
    back = mesh.fetch_real()[:, :, :N]
    assert np.abs(back/N**3 - dens).max() <= 1e-13*np.abs(dens).max()
    # fused solve == forward + kernel + backward
    C, E = -3.7, -1e-4
    for lr in (False, True):
        mesh.zero()
        mesh.deposit(pos, 1.0)
        mesh.poisson_solve(4, C, lr, E)
        phi_fused = mesh.fetch_real()[:, :, :N].copy()
        mesh.zero()
        mesh.deposit(pos, 1.0)
        mesh.poisson_forward(4, C, lr, E, apply_kernel=True)
        mesh.poisson_backward()
        phi_split = mesh.fetch_real()[:, :, :N]
        assert np.abs(phi_fused - phi_split).max() <= 1e-13*np.abs(phi_split).max()
    # rocFFT backend on the same density
    monkeypatch.setenv('CONCEPT_GPU_FFT', 'rocfft')
    mesh2 = PotentialMesh(N, L)
    mesh2.zero()
    mesh2.deposit(pos, 1.0)
    mesh2.poisson_solve(4, C, True, E)
    phi_roc = mesh2.fetch_real()[:, :, :N]
    assert np.abs(phi_fused - phi_roc).max() <= 1e-12*np.abs(phi_roc).max()


def test_fft_2048_roundtrip(torch_cuda):
    """The largest grid of BASELINE.json (2048^3, config 4's mesh; 69 GB on one GPU):
    forward then backward = N^3 * identity, and the fused solve equals the split one."""
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    N, L, n = 2048, 2048.0, 2_000_000
    mesh = PotentialMesh(N, L)
    gen = torch.Generator(device='cuda').manual_seed(5)
    pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen)*(L*(1 - 1e-12))
    probe = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
    mesh.zero()
    mesh.deposit(pos, 1.0)
    mesh.poisson_forward(0, 1.0, False, 0.0, apply_kernel=False)
    mesh.poisson_backward()
    # read the mesh back at the particles: CIC gather of an (N^3-scaled) density is heavy to
    # fetch whole; compare through the gather kernel instead (same on both sides)
    mesh.gather_kick(pos, probe, 2, 1.0)
    a = probe.clone()
    mesh.zero()
    mesh.deposit(pos, float(N)**3)
    probe.zero_()
    mesh.gather_kick(pos, probe, 2, 1.0)
    assert float((a - probe).abs().max()) <= 1e-11*float(probe.abs().max())
    # fused vs split Poisson solve
    k1 = torch.zeros_like(probe)
    k2 = torch.zeros_like(probe)
    mesh.zero()
    mesh.deposit(pos, 1.0)
    mesh.poisson_solve(4, -1.0, False, 0.0)
    mesh.gather_kick(pos, k1, 2, 1.0)
    mesh.zero()
    mesh.deposit(pos, 1.0)
    mesh.poisson_forward(4, -1.0, False, 0.0, apply_kernel=True)
    mesh.poisson_backward()
    mesh.gather_kick(pos, k2, 2, 1.0)
    assert float((k1 - k2).abs().max()) <= 1e-12*float(k2.abs().max())
    assert float(k2.abs().max()) > 0
    mesh.close()


def test_fft_2048_known_answer(torch_cuda):
    """The 2048-point passes against an answer known in closed form — the kernels this size
    instantiates (k_fft_strided_h: even/odd split over two 1024-point tiles, 8 pencils) are
    used by no smaller grid.  The transform of a*delta_p + b*delta_q is
    a*exp(-2 pi i k.p/N) + b*exp(-2 pi i k.q/N): every mode of sampled x layers is compared
    (<= 1e-13: a delta's transform is a single product of twiddles), the inverse transform must
    return the two deltas, and the fused solve must equal forward + kernel + inverse."""
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    N, L = 2048, 2048.0
    mesh = PotentialMesh(N, L)
    per, pad = mesh.layer_doubles, mesh.pad
    (a, p), (b, q) = (1.0, (3, 1029, 2047)), (-0.625, (1024, 7, 1))
    mesh.zero()
    layer = torch.zeros(per, dtype=torch.float64, device='cuda')
    for amp, (px, py, pz) in ((a, p), (b, q)):
        layer.zero_()
        layer[py*pad + pz] = amp
        mesh.layers_write(px, 1, layer)
    mesh.poisson_forward(0, 1.0, False, 0.0, apply_kernel=False)
    j = torch.arange(N, device='cuda').view(N, 1)
    k = torch.arange(N//2 + 1, device='cuda').view(1, N//2 + 1)
    worst = 0.0
    for i in (0, 1, 2, 63, 64, 511, 1023, 1024, 1025, 1536, 2046, 2047):
        mesh.layers_read(i, 1, layer)
        got = torch.view_as_complex(layer[:N*pad].view(N, pad//2, 2))[:, :N//2 + 1]
        ref = torch.zeros_like(got)
        for amp, (px, py, pz) in ((a, p), (b, q)):
            phase = ((i*px + j*py + k*pz) % N).to(torch.float64)*(-2*np.pi/N)
            ref += amp*torch.polar(torch.ones_like(phase), phase)
        worst = max(worst, float((got - ref).abs().max()))
    assert worst <= 1e-13, worst
    mesh.poisson_backward()
    for amp, (px, py, pz) in ((a, p), (b, q)):
        mesh.layers_read(px, 1, layer)
        real = layer[:N*pad].view(N, pad)[:, :N]
        assert abs(float(real[py, pz])/N**3 - amp) <= 1e-13
        real[py, pz] = 0.0
        assert float(real.abs().max())/N**3 <= 1e-13
    # the fused x pass (forward, factor, inverse in one kernel) against the three separate ones
    def prepare():
        mesh.zero()
        for amp, (px, py, pz) in ((a, p), (b, q)):
            layer.zero_()
            layer[py*pad + pz] = amp
            mesh.layers_write(px, 1, layer)
    other = torch.empty_like(layer)
    for lr in (False, True):
        prepare()
        mesh.poisson_solve(4, -2.5, lr, -3e-5)
        fused = {}
        for i in (0, 3, 1024, 2047):
            mesh.layers_read(i, 1, layer)
            fused[i] = layer.clone()
        prepare()
        mesh.poisson_forward(4, -2.5, lr, -3e-5, apply_kernel=True)
        mesh.poisson_backward()
        for i, f in fused.items():
            mesh.layers_read(i, 1, other)
            s = other[:N*pad].view(N, pad)[:, :N]
            d = (f[:N*pad].view(N, pad)[:, :N] - s).abs().max()
            assert float(d) <= 1e-12*float(s.abs().max()), (lr, i, float(d))
    mesh.close()


def test_multi_component_superposition(torch_cuda, golden):
    """particle_mesh with several suppliers/receivers (interactions.py:2029-2035,
    mesh.py:604-608): splitting one component into two of the same particle mass must
    give the single-component kicks; one of the two is tile-sorted (LDS path, assigns
    the mesh), the other unsorted (direct path, accumulates)."""
    from concept_amd import commons, interactions
    from concept_amd.species import Component
    g = golden('pm_n16_g32')
    commons.load_params({
        'boxsize': float(g['boxsize']),
        'potential_options': {'gridsize': {'gravity': {'pm': int(g['gridsize'])}}},
        'select_forces': {'matter': {'gravity': 'pm'}},
    })
    n = int(g['N'])
    half = n//2
    a = Component('a', 'matter', N=half, mass=float(g['mass']))
    b = Component('b', 'matter', N=n - half, mass=float(g['mass']))
    a.populate(g['pos_in'][:half], 'pos')
    a.populate(g['mom_in'][:half], 'mom')
    b.populate(g['pos_in'][half:], 'pos')
    b.populate(g['mom_in'][half:], 'mom')
    a.tile_sort()
    sdt = {'1': float(g['dt_1'])}
    for c in (a, b):
        sdt['a**(-3*w_eff)', c.name] = float(g['dt_kick'])
        sdt['a**(-3*w_eff-1)', c.name] = float(g['dt_dens'])
    interactions.gravity('pm', [a, b], [a, b], sdt, 'long-range', False)
    out = np.concatenate([a.host('mom'), b.host('mom')])
    kick = g['mom_after_long'] - g['mom_in']
    assert np.abs(out - g['mom_after_long']).max() <= TOL*rms(kick)


def test_drift_sort_fused_equals_drift_then_sort(torch_cuda):
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    L, N, n = 40.0, 64, 100_003
    mesh = PotentialMesh(N, L)
    rng = np.random.default_rng(9)
    pos = torch.tensor(rng.uniform(0, L, (n, 3)), device='cuda')
    mom = torch.tensor(rng.normal(0, 2.0, (n, 3)), device='cuda')
    ids = torch.arange(n, device='cuda')
    pos_a, mom_a = pos.clone(), mom.clone()
    mesh.drift(pos_a, mom_a, 0.7)
    po1, mo1, io1 = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    t1 = mesh.sort_particles(pos_a, mom_a, ids, po1, mo1, io1)
    po2, mo2, io2 = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    t2 = mesh.drift_sort(pos, mom, ids, po2, mo2, io2, 0.7)
    assert torch.equal(t1, t2)
    # same particles per slot range (order inside a bucket is arbitrary): compare via ids
    back1 = torch.empty_like(po1)
    back2 = torch.empty_like(po2)
    back1[io1] = po1
    back2[io2] = po2
    assert torch.equal(back1, back2) and torch.equal(back1, pos_a)  # drift arithmetic identical
    m1 = torch.empty_like(mo1)
    m2 = torch.empty_like(mo2)
    m1[io1] = mo1
    m2[io2] = mo2
    assert torch.equal(m1, m2) and torch.equal(m1, mom)


def test_north_star_size_properties(torch_cuda):
    """BASELINE.json's metric configuration itself (2^28 particles / 1024^3 mesh, ~75 GB of
    HBM): size-independent properties of the production kernels — the fused drift+sort is a
    permutation, the LDS pull deposit equals the direct atomic deposit, total mass is
    conserved, the tiled gather equals the direct gather bit for bit, the mesh force
    transfers no net momentum."""
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    N, L, n = 1024, 1024.0, 2**28
    mesh = PotentialMesh(N, L)
    gen = torch.Generator(device='cuda').manual_seed(3)
    pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen)
    pos.mul_(L*(1 - 1e-13))
    mom = torch.zeros_like(pos)
    ids = torch.arange(n, device='cuda')
    po, mo, io = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    table = mesh.drift_sort(pos, mom, ids, po, mo, io, 0.0)
    assert (int(table[-1].item()) & 0xffffffff) == n
    assert int(io.sum().item()) == n*(n - 1)//2
    sample = torch.arange(0, n, 4099, device='cuda')
    assert torch.equal(po[sample], pos[io[sample]])
    del pos, ids
    per = mesh.layer_doubles  # a layer = N rows of `pad` doubles + one unused row
    rows = per//mesh.pad
    a = torch.empty(N*per, dtype=torch.float64, device='cuda')
    b = torch.empty(N*per, dtype=torch.float64, device='cuda')
    mesh.deposit_tiled(po, table, 1.0, accumulate=False)
    mesh.layers_read(0, N, a)
    mesh.zero()
    mesh.deposit(po, 1.0)
    mesh.layers_read(0, N, b)
    av = a.view(N, rows, mesh.pad)[:, :N, :N]
    bv = b.view(N, rows, mesh.pad)[:, :N, :N]
    assert float((av - bv).abs().max()) <= 1e-12*float(bv.abs().max())
    assert abs(float(av.sum()) - n) <= 1e-9*n
    del a, b, av, bv
    mesh.poisson_solve(4, -1.0, False, 0.0)
    k_t = torch.zeros_like(mo)
    mesh.gather_kick_tiled(po, k_t, table, 2, 1.0)
    mesh.gather_kick(po, mo, 2, 1.0)
    assert torch.equal(k_t, mo)
    tot = mo.sum(0).abs().max()
    assert float(tot) <= 1e-9*float(mo.abs().sum(0).max())
    mesh.close()


def test_prepared_histogram_equals_plain(torch_cuda):
    """cg_gather_kick_tiled_prepare + cg_drift_sort (histogram pass skipped) gives exactly
    the tile table and particle arrays of the unfused sequence."""
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    L, N, n = 64.0, 64, 200_001
    mesh = PotentialMesh(N, L)
    rng = np.random.default_rng(21)
    pos = torch.tensor(rng.uniform(0, L, (n, 3)), device='cuda')
    mom = torch.tensor(rng.normal(0, 1.0, (n, 3)), device='cuda')
    ids = torch.arange(n, device='cuda')
    po, mo, io = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    table = mesh.sort_particles(pos, mom, ids, po, mo, io)
    mesh.deposit_tiled(po, table, 1.0)
    mesh.poisson_solve(4, -1.0, False, 0.0)
    dtm = 0.8
    out = {}
    for mode in ('plain', 'prepared'):
        p1, m1 = po.clone(), mo.clone()
        if mode == 'plain':
            mesh.gather_kick_tiled(p1, m1, table, 2, -0.3)
        else:
            mesh.gather_kick_tiled_prepare(p1, m1, table, 2, -0.3, dtm)
        p2, m2, i2 = torch.empty_like(p1), torch.empty_like(m1), torch.empty_like(io)
        t2 = mesh.drift_sort(p1, m1, io, p2, m2, i2, dtm)
        back_p, back_m = torch.empty_like(p2), torch.empty_like(m2)
        back_p[i2] = p2
        back_m[i2] = m2
        out[mode] = (t2.clone(), back_p, back_m)
    assert torch.equal(out['plain'][0], out['prepared'][0])
    assert torch.equal(out['plain'][1], out['prepared'][1])
    assert torch.equal(out['plain'][2], out['prepared'][2])
    # a different dt_over_mass must NOT reuse the prepared histogram
    p1, m1 = po.clone(), mo.clone()
    mesh.gather_kick_tiled_prepare(p1, m1, table, 2, -0.3, dtm)
    p2, m2, i2 = torch.empty_like(p1), torch.empty_like(m1), torch.empty_like(io)
    t3 = mesh.drift_sort(p1, m1, io, p2, m2, i2, 0.5*dtm)
    assert (int(t3[-1].item()) & 0xffffffff) == n
    p1b, m1b = po.clone(), mo.clone()
    mesh.gather_kick_tiled(p1b, m1b, table, 2, -0.3)
    t4 = mesh.drift_sort(p1b, m1b, io, torch.empty_like(p1), torch.empty_like(m1),
                         torch.empty_like(io), 0.5*dtm)
    assert torch.equal(t3, t4)


@pytest.mark.parametrize('order', [2, 4])
def test_fused_kick_drift_scatter_equals_separate(torch_cuda, order):
    """cg_gather_kick_drift_scatter (kick, drift and tile sort in one pass, particles kept in
    regions with gaps) against cg_gather_kick_tiled + cg_drift_sort: the same particles with
    bit-identical positions and momenta, every particle inside the region of its (tile,
    bucket); two consecutive fused steps (the second reads regions with gaps)."""
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    L, N, n = 64.0, 64, 300_007
    mesh = PotentialMesh(N, L)
    rng = np.random.default_rng(31 + order)
    pos = torch.tensor(rng.uniform(0, L, (n, 3)), device='cuda')
    mom = torch.tensor(rng.normal(0, 0.4, (n, 3)), device='cuda')
    ids = torch.arange(n, device='cuda')
    cap = mesh.region_capacity(n)
    assert cap >= n
    p0, m0, i0 = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    table = mesh.sort_particles(pos, mom, ids, p0, m0, i0)
    dtm, kick = 0.9, -0.3

    # both paths read the SAME potential in every step (the tiled deposit adds in hardware
    # order: two deposits of the same particles differ by rounding)
    big = lambda t: torch.cat([t, torch.full((cap - n,) + tuple(t.shape[1:]), -7,
                                             dtype=t.dtype, device='cuda')])
    pa, ma, ia = big(p0), big(m0), big(i0)
    pb, mb, ib = torch.full_like(pa, -7), torch.full_like(ma, -7), torch.full_like(ia, -7)
    start_in, count_in = table, None
    tabs = [mesh.new_region_table(), mesh.new_region_table()]
    pr, mr, ir, tr = p0.clone(), m0.clone(), i0.clone(), table
    for step in range(2):
        # potential from the fused path's layout: dense first, then regions with gaps
        if count_in is None:
            mesh.deposit_tiled(pa[:n], table, 1.0)
        else:
            mesh.deposit_regions(pa, start_in, count_in, 1.0)
        mesh.poisson_solve(4, -1.0, False, 0.0)
        # fused
        start_out, count_out = tabs[step]
        mesh.predict_regions(start_in, count_in, start_out)
        mesh.gather_kick_drift_scatter(pa, ma, ia, start_in, count_in, pb, mb, ib, start_out,
                                       count_out, order, kick, dtm)
        assert mesh.error_flags() == 0
        # reference: the separate kernels on the dense copy
        mesh.gather_kick_tiled(pr, mr, tr, order, kick)
        p2, m2, i2 = torch.empty_like(pr), torch.empty_like(mr), torch.empty_like(ir)
        tr = mesh.drift_sort(pr, mr, ir, p2, m2, i2, dtm)
        pr, mr, ir = p2, m2, i2
        ref_p, ref_m = torch.empty_like(pr), torch.empty_like(mr)
        ref_p[ir] = pr
        ref_m[ir] = mr
        st, ct = start_out.long(), count_out.long()
        assert int(ct.sum()) == n and int(st[-1]) <= cap
        assert bool((st[1:] - st[:-1] >= ct).all())
        # live slots: [start[k], start[k] + count[k])
        slot = torch.arange(cap, device='cuda')
        k = (torch.searchsorted(st, slot, right=True) - 1).clamp(max=ct.numel() - 1)
        live = (slot - st[k]) < ct[k]
        assert int(live.sum()) == n
        got_i = ib[live]
        assert torch.equal(torch.sort(got_i)[0], torch.arange(n, device='cuda'))
        back_p = torch.empty((n, 3), dtype=torch.float64, device='cuda')
        back_m = torch.empty((n, 3), dtype=torch.float64, device='cuda')
        back_p[got_i] = pb[live]
        back_m[got_i] = mb[live]
        assert torch.equal(back_m, ref_m) and torch.equal(back_p, ref_p)
        # the populations are those of the exact sort
        dense = tr.long() & 0xffffffff
        assert torch.equal(ct, dense[1:] - dense[:-1])
        pa, pb, ma, mb, ia, ib = pb, pa, mb, ma, ib, ia
        start_in, count_in = start_out, count_out
    mesh.close()


def test_fused_scatter_overflow_is_flagged(torch_cuda):
    """A bucket that outgrows its predicted region (here: every particle is sent into one
    corner) sets CG_ERR_BUCKET_OVERFLOW and leaves the input arrays untouched."""
    torch = torch_cuda
    from concept_amd import lib
    from concept_amd.mesh import PotentialMesh
    L, N, n = 64.0, 64, 100_000
    mesh = PotentialMesh(N, L)
    rng = np.random.default_rng(5)
    pos = torch.tensor(rng.uniform(0, L, (n, 3)), device='cuda')
    # momenta that carry everybody to the same point under the drift below
    target = torch.tensor([3.0, 3.0, 3.0], dtype=torch.float64, device='cuda')
    mom = (target - pos)
    cap = mesh.region_capacity(n)
    p0, m0 = torch.empty_like(pos), torch.empty_like(mom)
    table = mesh.sort_particles(pos, mom, None, p0, m0, None)
    mesh.zero()   # zero potential: no kick
    pa = torch.cat([p0, torch.zeros((cap - n, 3), dtype=torch.float64, device='cuda')])
    ma = torch.cat([m0, torch.zeros((cap - n, 3), dtype=torch.float64, device='cuda')])
    keep_p, keep_m = pa.clone(), ma.clone()
    pb, mb = torch.empty_like(pa), torch.empty_like(ma)
    start_out, count_out = mesh.new_region_table()
    mesh.predict_regions(table, None, start_out)
    mesh.gather_kick_drift_scatter(pa, ma, None, table, None, pb, mb, None, start_out, count_out,
                                   2, 1.0, 1.0)
    assert mesh.error_flags() & lib.CG_ERR_BUCKET_OVERFLOW
    assert torch.equal(pa, keep_p) and torch.equal(ma, keep_m)
    mesh.close()


@pytest.mark.parametrize('name', ['pm_n8_g16_d1', 'pm_n8_g16_d6', 'pm_n8_g16_d8',
                                  'pm_n8_g16_vertex', 'pm_n8_g16_deconv_up'])
def test_other_differentiation_orders_vs_golden(torch_cuda, golden, name):
    """diff_domaingrid's other orders (mesh.py:4874-5030): 6 and 8 — for which the reference
    raises nghosts to 3 and 4 (commons.py:4428-4430) — and the one-sided order 1, through the
    gravity() boundary (force grid by cg_mesh_diff, then interpolated) and mesh by mesh.
    'pm_n8_g16_vertex': the user parameter cell_centered = False (vertex-centred grids; the
    reference then keeps 3 ghost layers, commons.py:4411-4419) on the fused order-2 path."""
    from concept_amd import comm, commons, interactions
    from concept_amd.mesh import PotentialMesh
    from concept_amd.species import Component
    from oracle import oracle
    g = golden(name)
    order, N, L = int(g['diff_order']), int(g['gridsize']), float(g['boxsize'])
    cc = bool(int(g['cell_centered']))
    deconv = tuple(bool(v) for v in g['deconvolve']) if 'deconvolve' in g else (True, True)
    p = commons.load_params({
        'boxsize': L, 'cell_centered': cc,
        'potential_options': {'gridsize': {'gravity': {'pm': N}},
                              'deconvolve': {'gravity': {'pm': deconv}},
                              'differentiation': {'matter': {'gravity': {'pm': order}}}},
        'select_forces': {'matter': {'gravity': 'pm'}},
    })
    assert p.nghosts == int(g['nghosts'])
    c = Component('matter', 'matter', N=int(g['N']), mass=float(g['mass']))
    c.populate(g['pos_in'], 'pos')
    c.populate(g['mom_in'], 'mom')
    single = comm.active() is None  # (also run over domains by test_gpu_distributed.py)
    if single:
        assert np.array_equal(c._mesh().cic_indices(c.pos, False).cpu().numpy(),
                              g['cic_index_deposit'])
    sdt = {'1': float(g['dt_1']), ('a**(-3*w_eff)', 'matter'): float(g['dt_kick']),
           ('a**(-3*w_eff-1)', 'matter'): float(g['dt_dens'])}
    interactions.gravity('pm', [c], [c], sdt, 'long-range', False)
    kick_ref = g['mom_after_long'] - g['mom_in']
    assert np.abs(c.host('mom') - g['mom_after_long']).max() <= TOL*rms(kick_ref) \
        + 4e-16*np.abs(g['mom_in']).max()
    if not single:
        return
    # the force grids themselves
    ng = int(g['nghosts'])
    mesh = PotentialMesh(N, L, nghosts=ng, cell_centered=cc)
    force = PotentialMesh(N, L, nghosts=ng, cell_centered=cc)
    pos = torch_cuda.tensor(g['pos_in'], device='cuda')
    contribution = oracle.deposit_contribution(float(g['mass']), float(g['dt_dens']),
                                               float(g['dt_1']), N, L)
    C, _ = oracle.poisson_constants(L, float(g['G_Newton']), None)
    mesh.zero()
    mesh.deposit(pos, contribution)
    mesh.poisson_solve(2*sum(deconv), C, False, 0.0)  # (interactions.py:2069-2080)
    for dim in range(3):
        force.diff_from(mesh, dim, order)
        got = force.fetch_real()[:, :, :N]
        ref = g['grid_force'][dim][ng:-ng, ng:-ng, ng:-ng]
        assert np.abs(got - ref).max() <= TOL*rms(ref)


def test_streaming_timeloop_two_components(torch_cuda):
    """stepper.timeloop without a callback takes kick + drift + tile sort in one pass
    (cg_gather_kick_drift_scatter) — here with two components of different mass and
    differentiation order on the shared mesh: same end state, row by row, as the loop that calls
    gravity() and drift() one after the other (main.py:255-361)."""
    from concept_amd import commons, stepper
    from concept_amd.species import Component
    rng = np.random.default_rng(21)
    L, gs, n = 50.0, 32, (3000, 1700)
    commons.load_params({
        'boxsize': L,
        'potential_options': {'gridsize': {'gravity': {'pm': gs}},
                              'differentiation': {'light': {'gravity': {'pm': 4}}}},
        'select_forces': {'all': {'gravity': 'pm'}},
    })
    data = [(rng.uniform(0, L, (k, 3)), rng.normal(0, 2.0, (k, 3))) for k in n]

    def run(stream):
        comps = [Component('heavy', 'matter', N=n[0], mass=3.0),
                 Component('light', 'matter', N=n[1], mass=0.7)]
        for c, (pos, mom) in zip(comps, data):
            c.populate(pos, 'pos')
            c.populate(mom, 'mom')

        def integrals(kind):
            d = 0.05 if kind == 'init' else 0.1
            out = {'1': d, 'a**(-2)': 1.2*d}
            for c in comps:
                out['a**(-3*w_eff)', c.name] = 1.1*d
                out['a**(-3*w_eff-1)', c.name] = 0.9*d
            return out
        stepper.timeloop(comps, 3, integrals, None, None if stream else (lambda step: None))
        return [(c.host('pos'), c.host('mom'), c.host('ids')) for c in comps]
    for (p0, m0, i0), (p1, m1, i1), (_, mom_in) in zip(run(False), run(True), data):
        d = np.abs(p0 - p1)
        assert np.minimum(d, L - d).max() <= 1e-13*L
        assert np.abs(m0 - m1).max() <= TOL*rms(m0 - mom_in) + 4e-16*np.abs(m0).max()
        assert np.array_equal(i0, i1)


def test_streaming_timeloop_region_overflow_is_replayed(torch_cuda):
    """A step in which a (tile, bucket) grows beyond the region predicted for it
    (CG_ERR_BUCKET_OVERFLOW: here a flow converging on one tile) is undone and taken on the
    exact path by stepper.timeloop; the end state equals the non-streaming loop's."""
    from concept_amd import commons, stepper
    from concept_amd.species import Component
    rng = np.random.default_rng(3)
    L, gs, n = 32.0, 32, 20000
    commons.load_params({'boxsize': L, 'potential_options': {'gridsize': {'gravity': {'pm': gs}}},
                         'select_forces': {'all': {'gravity': 'pm'}}})
    pos = rng.uniform(0, L, (n, 3))
    d, mass = 0.1, 2.0
    # the first full drift (a**(-2) integral 1.2 d) takes every particle 90 % of the way to (8, 8, 8)
    mom = (8.0 - pos)*0.9*mass/(1.2*d)

    def integrals(kind):
        s = d/2 if kind == 'init' else d
        return {'1': s, 'a**(-2)': 1.2*s if kind == 'init' else 1.2*d,
                ('a**(-3*w_eff)', 'm'): 1e-3*s, ('a**(-3*w_eff-1)', 'm'): 0.9*s}

    def run(stream):
        c = Component('m', 'matter', N=n, mass=mass)
        c.populate(pos, 'pos')
        c.populate(mom, 'mom')
        stepper.timeloop([c], 2, integrals, None, None if stream else (lambda step: None))
        return c.host('pos'), c.host('mom'), c.host('ids')
    replays = stepper.stream_replays
    (p0, m0, i0), (p1, m1, i1) = run(False), run(True)
    assert stepper.stream_replays > replays  # (the overflow really happened)
    dd = np.abs(p0 - p1)
    assert np.minimum(dd, L - dd).max() <= 1e-13*L
    assert np.abs(m0 - m1).max() <= 1e-12*np.abs(m0).max()
    assert np.array_equal(i0, i1)


@pytest.mark.parametrize('bulk', [False, True, 'point'])
def test_void_domains_vs_oracle(torch_cuda, bulk):
    """All particles in one eighth of the box along x, streaming towards +x: on several x-slab
    domains most ranks start EMPTY and some receive their first particles by exchange()
    (communication.py:135-517).  stepper.timeloop — step by step and in its streaming form —
    against the CPU oracle's K½ D K D K (decomposition-independent).  bulk: 12000 particles
    crossing 13 cells per step — whole slabs change hands at once, more leavers than the
    streaming pass's row buffer holds (the step is then repeated on the exact path).  'point':
    all 5000 particles inside ONE mesh cell, nearly at rest (one tile bucket holds everything).
    Also run on 2 and 4 domains by tests/test_gpu_distributed.py."""
    from concept_amd import commons, stepper
    from concept_amd.species import Component
    from oracle import oracle
    rng = np.random.default_rng(17)
    point = bulk == 'point'
    bulk = bulk is True
    L, gs, n, mass, d = 32.0, 32, (12000 if bulk else 5000 if point else 4000), 1.5, 0.25
    commons.load_params({'boxsize': L, 'potential_options': {'gridsize': {'gravity': {'pm': gs}}},
                         'select_forces': {'all': {'gravity': 'pm'}}})
    p = commons.params
    pos0 = rng.uniform(0, L, (n, 3))
    pos0[:, 0] *= 1/8
    mom0 = rng.normal(0, 0.3, (n, 3))
    mom0[:, 0] += (13.0 if bulk else 2.5)*mass/d  # cells per step towards +x
    if point:
        pos0 = 7.0 + rng.uniform(0.05, 0.95, (n, 3))*(L/gs)
        mom0 = rng.normal(0, 1e-3, (n, 3))

    def integrals(kind):
        s = d/2 if kind == 'init' else d
        return {'1': s, 'a**(-2)': d, ('a**(-3*w_eff)', 'm'): s, ('a**(-3*w_eff-1)', 'm'): s}
    # oracle: K(d/2) D K D K
    pos, mom = pos0.copy(), mom0.copy()
    for step in range(3):
        s = d/2 if step == 0 else d
        oracle.pm_long_range(pos, mom, mass=mass, boxsize=L, gridsize=gs, G_Newton=p.G_Newton,
                             dt_1=s, dt_dens=s, dt_kick=s, diff_order=2, want_indices=False)
        if step < 2:
            oracle.drift(pos, mom, d/mass, L)
    kick = rms(mom - mom0)
    for stream in (False, True):
        c = Component('m', 'matter', N=n, mass=mass)
        c.populate(pos0, 'pos')
        c.populate(mom0, 'mom')
        stepper.timeloop([c], 2, integrals, None, None if stream else (lambda step: None))
        dx = np.abs(c.host('pos') - pos)
        assert np.minimum(dx, L - dx).max() <= 1e-12*L, stream
        assert np.abs(c.host('mom') - mom).max() <= 1e-11*kick + 4e-16*np.abs(mom).max(), stream


@pytest.mark.parametrize('knot', [False, True])
def test_void_domains_p3m(torch_cuda, knot):
    """The same void box with P3M (long-range mesh + short-range sweep, two steps): on several
    domains — most of them empty, boundary suppliers shipped into and out of empty slabs —
    against the single-domain run of the same calls.  (Single domain: the run against itself;
    the multi-rank form is what tests/test_gpu_distributed.py adds.)"""
    from concept_amd import comm, commons, stepper
    from concept_amd.species import Component
    rng = np.random.default_rng(23)
    L, gs, n, mass, d = 64.0, 64, 6000, 1.5, 0.25
    pos0 = rng.uniform(0, L, (n, 3))
    pos0[:, 0] *= 1/8
    mom0 = rng.normal(0, 0.3, (n, 3))
    mom0[:, 0] += 3.0*mass/d
    if knot:
        # a knot of 3000 particles two cells across, centred on the face between two slabs
        # (x = L/2 on 2 and 4 domains): thousands of pairs per tile, every pair across the face
        # met through the shipped boundary suppliers; the rest of the box empty
        n = 3000
        pos0 = np.array([L/2, 20.3, 41.7]) + rng.normal(0, 0.7, (n, 3))
        pos0 %= L
        mom0 = rng.normal(0, 1e-2, (n, 3))

    def integrals(kind):
        s = d/2 if kind == 'init' else d
        return {'1': s, 'a**(-2)': d, ('a**(-3*w_eff)', 'm'): s, ('a**(-3*w_eff-1)', 'm'): s}

    def rung_integrals(kind):
        s = d/2 if kind == 'init' else d
        return {('a**(-3*w_eff₀-3*w_eff₁-1)', 'm', 'm'): np.full(2, s)}

    def run():
        commons.load_params({'boxsize': L, 'N_rungs': 1,
                             'potential_options': {'gridsize': {'gravity': {'p3m': gs}}},
                             'select_forces': {'all': {'gravity': 'p3m'}}})
        c = Component('m', 'matter', N=n, mass=mass)
        c.populate(pos0, 'pos')
        c.populate(mom0, 'mom')
        stepper.timeloop([c], 2, integrals, rung_integrals)
        return c.host('pos'), c.host('mom')
    active = comm.active()
    if active is not None:
        comm.shutdown()
    pos_ref, mom_ref = run()
    if active is not None:
        comm.init()
    pos, mom = run()
    dx = np.abs(pos - pos_ref)
    assert np.minimum(dx, L - dx).max() <= 1e-12*L
    assert np.abs(mom - mom_ref).max() <= 1e-11*rms(mom_ref - mom0) + 4e-16*np.abs(mom_ref).max()


def test_empty_particle_sets(torch_cuda):
    """Zero particles through every particle entry point of the PM path (an x-slab domain of a
    void owns none): nothing is touched, nothing fails, the mesh of an empty deposit is zero."""
    torch = torch_cuda
    from concept_amd.mesh import PotentialMesh
    mesh = PotentialMesh(32, 10.0)
    e3 = torch.empty((0, 3), dtype=torch.float64, device='cuda')
    ids = torch.empty(0, dtype=torch.int64, device='cuda')
    mesh.zero()
    mesh.deposit(e3, 1.0)
    table = mesh.sort_particles(e3, e3.clone(), ids, e3.clone(), e3.clone(), ids.clone())
    assert int(table.long().abs().sum()) == 0
    mesh.deposit_tiled(e3, table, 1.0, accumulate=False)
    assert float(torch.tensor(mesh.fetch_real()).abs().max()) == 0.0
    mesh.poisson_solve(4, -1.0, False, 0.0)
    mesh.gather_kick(e3, e3.clone(), 2, 1.0)
    mesh.gather_kick_tiled(e3, e3.clone(), table, 2, 1.0)
    mesh.drift(e3, e3.clone(), 0.1)
    mesh.drift_sort(e3, e3.clone(), None, e3.clone(), e3.clone(), None, 0.1, mesh.new_tile_table())
    assert mesh.cic_indices(e3).shape == (0, 3)
    mesh.check_errors()


@pytest.mark.parametrize('seed', range(8))
def test_random_streaming_timeloops(torch_cuda, seed):
    """stepper.timeloop in streaming form against the loop that calls gravity() and drift() one
    after the other, over random draws: one to three components of different mass and
    differentiation order, velocities from a fraction of a cell to several tiles per step
    (region overflows are replayed), two to four steps.  Also run over 2 and 4 domains."""
    from concept_amd import commons, stepper
    from concept_amd.species import Component
    rng = np.random.default_rng(3000 + seed)
    L = float(rng.choice([32.0, 64.0]))
    gs = int(rng.choice([32, 64]))
    ncomp = int(rng.integers(1, 4))
    names = ['a', 'b', 'c'][:ncomp]
    orders = {nm: int(rng.choice([2, 4])) for nm in names}
    commons.load_params({
        'boxsize': L,
        'potential_options': {'gridsize': {'gravity': {'pm': gs}},
                              'differentiation': {nm: {'gravity': {'pm': o}}
                                                  for nm, o in orders.items()}},
        'select_forces': {'all': {'gravity': 'pm'}}})
    d = float(rng.uniform(0.05, 0.3))
    nsteps = int(rng.integers(2, 5))
    data = []
    for nm in names:
        n = int(rng.integers(300, 5000))
        mass = float(rng.uniform(0.5, 3.0))
        pos = rng.uniform(0, L, (n, 3))
        if rng.integers(0, 2):
            pos[:n//2] = (rng.uniform(0, L, 3) + rng.normal(0, 0.05*L, (n//2, 3))) % L
        speed = float(rng.choice([0.3, 3.0, 25.0]))*(L/gs)  # cells per step
        mom = rng.normal(0, speed*mass/(1.2*d), (n, 3))
        data.append((nm, n, mass, pos, mom))

    def integrals(kind):
        s = d/2 if kind == 'init' else d
        out = {'1': s, 'a**(-2)': 1.2*d}
        for nm in names:
            out['a**(-3*w_eff)', nm] = 1.1*s
            out['a**(-3*w_eff-1)', nm] = 0.9*s
        return out

    def run(stream):
        comps = []
        for nm, n, mass, pos, mom in data:
            c = Component(nm, 'matter', N=n, mass=mass)
            c.populate(pos, 'pos')
            c.populate(mom, 'mom')
            comps.append(c)
        stepper.timeloop(comps, nsteps, integrals, None, None if stream else (lambda step: None))
        return [(c.host('pos'), c.host('mom'), c.host('ids')) for c in comps]
    for (p0, m0, i0), (p1, m1, i1), (nm, n, mass, pos, mom) in zip(run(False), run(True), data):
        dd = np.abs(p0 - p1)
        assert np.minimum(dd, L - dd).max() <= 1e-12*L, (seed, nm)
        kick = np.abs(m0 - mom).max()
        assert np.abs(m0 - m1).max() <= 1e-11*kick + 4e-16*np.abs(m0).max(), (seed, nm)
        assert np.array_equal(i0, i1)
