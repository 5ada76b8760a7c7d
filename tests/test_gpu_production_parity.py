"""Parity of the PRODUCTION instantiations (run with -m gpu on an MI355X).

The small goldens exercise 16^3..64^3 meshes; the kernels the bench times are other template
instantiations (1024-point register-staged FFT passes over cache-sized chunks of layers, 16^3
tiles with 512-lane workgroups, the prepared drift histogram).  Here they are checked at their
own sizes against something independent of them:

  (i)   1024^3: forward transform and fused Poisson solve of a deposited density against the
        rocFFT backend on the same mesh layout (potential <= 1e-12 of the field rms; the
        largest of the 2^30 mode deviations <= 4e-12 of the rms mode);
  (ii)  BASELINE configs[0] size (128^3 particles / 256^3 mesh): drift + tile sort + PM kick with
        the tiled kernels against the C oracle — CIC indices bit-exact, first drift bit-exact,
        kick <= 1e-12 of the rms kick (the reference's own compiled-vs-pure-Python bar is 1e-10,
        test/pure_python_pm/analyze.py:125);
  (iii) one P3M kick (long-range with the Gaussian cut-off, differentiation order 4, + short-range
        tile sweep) at 64^3 / 128^3 and 128^3 / 256^3 against the C oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rms(t):
    return float((t.double()**2).mean().sqrt())


def _mesh_view(torch, mesh):
    """The owned layers of the mesh as a (N, rows, pad) device tensor (a copy)."""
    N = mesh.gridsize
    per = mesh.layer_doubles
    buf = torch.empty(N*per, dtype=torch.float64, device='cuda')
    mesh.layers_read(0, N, buf)
    return buf.view(N, per//mesh.pad, mesh.pad)[:, :N]


def test_fft_1024_vs_rocfft(monkeypatch):
    """(i) the 1024-point instantiation the bench times (k_fft_z_*<10,128>, the even/odd split
    passes k_fft_strided_h<10,256,*> with two 512-point tiles per 8 pencils, z/y passes
    interleaved over 31-layer chunks) against rocFFT's 3-D plan, forward and fused solve."""
    import torch
    from concept_amd.mesh import PotentialMesh
    N, L, n = 1024, 1024.0, 2**24
    gen = torch.Generator(device='cuda').manual_seed(11)
    pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen)
    # clustered along x so that the spectrum is not flat
    pos[:, 0] = pos[:, 0]**2
    pos.mul_(L*(1 - 1e-13))
    monkeypatch.delenv('CONCEPT_GPU_FFT', raising=False)
    own = PotentialMesh(N, L)
    monkeypatch.setenv('CONCEPT_GPU_FFT', 'rocfft')
    roc = PotentialMesh(N, L)
    monkeypatch.delenv('CONCEPT_GPU_FFT', raising=False)
    own.zero()
    own.deposit(pos, 1.0)
    roc.copy_from(own)  # (the direct deposit adds with atomics: two runs differ by rounding)
    dens = _mesh_view(torch, own)[:, :, :N].clone()
    assert torch.equal(dens, _mesh_view(torch, roc)[:, :, :N])
    # forward transform (A4): both backends leave complex[N][N][N/2+1] un-transposed in place
    own.poisson_forward(0, 1.0, False, 0.0, apply_kernel=False)
    roc.poisson_forward(0, 1.0, False, 0.0, apply_kernel=False)
    a = _mesh_view(torch, own)[:, :, :N + 2]
    b = _mesh_view(torch, roc)[:, :, :N + 2]
    scale = rms(b)
    err = float((a - b).abs().max())
    # the LARGEST deviation among 2^30 modes between two transforms that each round ~30 times
    # per mode (measured 1.1e-12 of the rms mode amplitude); the potential below, which is what
    # the path hands on, holds the 1e-12 bar
    assert err <= 4e-12*scale, (err, scale)
    # Parseval against the real-space density pins the pair to the true transform
    # (sum |F|^2 over the half spectrum, the kk = 0 and kk = N/2 planes counted once)
    re, im = a[:, :, 0::2], a[:, :, 1::2]
    p2 = re**2 + im**2
    total = 2*float(p2.sum()) - float(p2[:, :, 0].sum()) - float(p2[:, :, N//2].sum())
    assert abs(total/N**3 - float((dens**2).sum())) <= 1e-11*float((dens**2).sum())
    del a, b, re, im, p2
    # back: unnormalised round trip = N^3 * identity
    own.poisson_backward()
    back = _mesh_view(torch, own)[:, :, :N]
    assert float((back/N**3 - dens).abs().max()) <= 1e-12*float(dens.abs().max())
    del back
    # fused solve (A4-A8, the k-space factor inside the x pass) vs rocFFT + stand-alone kernel
    C, E = -L**2/np.pi, -(2*np.pi/L*1.25)**2
    for long_range in (False, True):
        own.zero()
        own.deposit(pos, 1.0)
        roc.copy_from(own)
        own.poisson_solve(4, C, long_range, E)
        roc.poisson_forward(4, C, long_range, E, apply_kernel=True)
        roc.poisson_backward()
        a = _mesh_view(torch, own)[:, :, :N]
        b = _mesh_view(torch, roc)[:, :, :N]
        scale = rms(b - b.mean())
        err = float((a - b).abs().max())
        assert err <= 1e-12*scale, (long_range, err, scale)
        del a, b
    own.close()
    roc.close()


def test_fft_1024_vs_rocfft_whole_pencil_kernels():
    """The same with CONCEPT_GPU_FFT_SPLIT=0: the whole-pencil passes
    (k_fft_strided_p<10,512,*,8>, radix 16*16*4) that the split passes replaced as the default
    stay a supported configuration (the switch is read once per process: a fresh one)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from _pytest.monkeypatch import MonkeyPatch\n"
            "import test_gpu_production_parity as t\n"
            "t.test_fft_1024_vs_rocfft(MonkeyPatch())\n"
            "print('WHOLE-PENCIL-OK')\n") % (here, os.path.dirname(here))
    p = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, CONCEPT_GPU_FFT_SPLIT='0'),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0 and b'WHOLE-PENCIL-OK' in p.stdout, p.stdout.decode()[-3000:]


def _oracle_pm_step(oracle, pos, mom, *, mass, L, N, G, dt, order, shortrange_scale=None):
    """drift -> long-range kick as the reference's time loop orders them (main.py:335-358)."""
    pos_d = oracle.drift(pos.copy(), mom, dt/mass, L)
    mom_k = mom.copy()
    out = oracle.pm_long_range(pos_d, mom_k, mass=mass, boxsize=L, gridsize=N, G_Newton=G,
                               dt_1=dt, dt_dens=dt, dt_kick=dt, diff_order=order,
                               shortrange_scale=shortrange_scale, want_indices=True)
    return pos_d, mom_k, out


@pytest.mark.parametrize('order', [2, 4])
def test_pm_step_vs_oracle_configs0(order):
    """(ii) BASELINE configs[0]: 128^3 particles / 256^3 mesh through the bench's own call
    sequence (cg_drift_sort -> cg_deposit_cic_tiled -> cg_poisson_solve ->
    cg_gather_kick_tiled_prepare -> cg_drift_sort on the prepared histogram)."""
    import torch
    from concept_amd.mesh import PotentialMesh
    from oracle import oracle
    N, L, n = 256, 400.0, 128**3
    mass, G, dt = 1.7, 0.9, 0.03
    rng = np.random.default_rng(5 + order)
    pos0 = rng.uniform(0, L, (n, 3))
    # thermal momenta: ~0.3 cells per drift, so particles change tile (and wrap around the box)
    mom0 = rng.normal(0, 0.3*(L/N)*mass/dt/3**0.5, (n, 3))
    pos_o, mom_o, out = _oracle_pm_step(oracle, pos0, mom0, mass=mass, L=L, N=N, G=G, dt=dt,
                                        order=order)
    mesh = PotentialMesh(N, L, nghosts=2)
    pos, mom = torch.tensor(pos0, device='cuda'), torch.tensor(mom0, device='cuda')
    ids = torch.arange(n, device='cuda')
    p1, m1, i1 = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    table = mesh.drift_sort(pos, mom, ids, p1, m1, i1, dt/mass)
    assert (int(table[-1].item()) & 0xffffffff) == n
    perm = i1.cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(n))
    # A11: the fused drift is bit-exact
    assert np.array_equal(p1.cpu().numpy(), pos_o[perm])
    # A1: CIC indices bit-exact (every 7th particle)
    sample = torch.arange(0, n, 7, device='cuda')
    idx = mesh.cic_indices(p1[sample].contiguous(), False).cpu().numpy()
    assert np.array_equal(idx, out['cic_index_deposit'][perm[::7]])
    idx = mesh.cic_indices(p1[sample].contiguous(), True).cpu().numpy()
    assert np.array_equal(idx, out['cic_index_gather'][perm[::7]])
    contribution = oracle.deposit_contribution(mass, dt, dt, N, L)
    C, _ = oracle.poisson_constants(L, G, None)
    mesh.deposit_tiled(p1, table, contribution, accumulate=False)
    # A1/A2: the density mesh against the oracle's folded deposit
    g = 2
    grid = out['grid_deposit']
    dens_o = np.zeros((N, N, N))
    ix = (np.arange(N + 2*g) - g) % N
    np.add.at(dens_o, (ix[:, None, None], ix[None, :, None], ix[None, None, :]), grid)
    dens = mesh.fetch_real()[:, :, :N]
    assert np.abs(dens - dens_o).max() <= 1e-12*np.sqrt((dens_o**2).mean())
    mesh.poisson_solve(4, C, False, 0.0)
    phi = mesh.fetch_real()[:, :, :N]
    phi_o = out['grid_potential'][g:-g, g:-g, g:-g]
    assert np.abs(phi - phi_o).max() <= 1e-12*np.sqrt((phi_o**2).mean())
    mesh.gather_kick_tiled_prepare(p1, m1, table, order, mass*(-dt), dt/mass)
    kick_o = (mom_o - mom0)[perm]
    kick = m1.cpu().numpy() - mom0[perm]
    scale = np.sqrt((kick_o**2).mean())
    # the kick is added to momenta ~1e4 times larger: their rounding (eps * |mom|) is the floor
    floor = 2.3e-16*np.abs(mom0).max()
    assert np.abs(kick - kick_o).max() <= 1e-12*scale + 2*floor
    # second drift on the prepared histogram: a permutation again, positions as the oracle's
    p2, m2, i2 = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    table2 = mesh.drift_sort(p1, m1, i1, p2, m2, i2, dt/mass)
    mesh.check_errors()
    assert (int(table2[-1].item()) & 0xffffffff) == n
    perm2 = i2.cpu().numpy()
    assert np.array_equal(np.sort(perm2), np.arange(n))
    pos_o2 = oracle.drift(pos_o.copy(), mom_o, dt/mass, L)
    d = np.abs(p2.cpu().numpy() - pos_o2[perm2])
    d = np.minimum(d, L - d)  # a particle on the box seam may wrap on one side only
    assert d.max() <= 1e-13*L
    # every particle sits in the tile its (drifted) position belongs to
    t_ext, nt = mesh.tile_extent, mesh.tiles_per_dim
    cells = mesh.cic_indices(p2, False) - g
    cells = torch.remainder(cells, N)
    tile = ((cells[:, 0]//t_ext)*nt + cells[:, 1]//t_ext)*nt + cells[:, 2]//t_ext
    bounds = table2[0::8].long() & 0xffffffff  # first particle of every tile, then n
    where = torch.searchsorted(bounds, torch.arange(n, device='cuda'), right=True) - 1
    assert torch.equal(where, tile)
    mesh.close()


@pytest.mark.parametrize('npart,N', [(64, 128), (128, 256)])
def test_p3m_kick_vs_oracle(npart, N):
    """(iii) one P3M kick at 64^3 / 128^3 and 128^3 / 256^3: long-range mesh part (Gaussian
    cut-off, order-4 differences) and the short-range tile sweep with the default parameters
    (scale 1.25 cells, range 4.5 scale, 4096-entry table, spline softening) vs the C oracle."""
    import torch
    from concept_amd import shortrange
    from concept_amd.mesh import PotentialMesh
    from oracle import oracle
    n, L = npart**3, 250.0
    mass, G, dt = 2.1, 1.3, 0.02
    rng = np.random.default_rng(npart)
    # mildly clustered: half uniform, half in Gaussian blobs (tiles of very different population)
    half = n//2
    centres = rng.uniform(0, L, (32, 3))
    blob = centres[rng.integers(0, 32, n - half)] + rng.normal(0, L/12, (n - half, 3))
    pos0 = np.concatenate([rng.uniform(0, L, (half, 3)), np.mod(blob, L)])
    pos0 = np.minimum(pos0, np.nextafter(L, 0))
    mom0 = rng.normal(0, 1.0, (n, 3))
    scale = 1.25*L/N
    range_ = 4.5*scale
    soft = 0.03*L/npart
    # oracle
    mom_o = mom0.copy()
    oracle.pm_long_range(pos0, mom_o, mass=mass, boxsize=L, gridsize=N, G_Newton=G, dt_1=dt,
                         dt_dens=dt, dt_kick=dt, diff_order=4, shortrange_scale=scale,
                         want_indices=False)
    factor = G*mass*mass*dt
    dmom_o, _ = oracle.shortrange_kick(pos0, boxsize=L, scale=scale, range_=range_,
                                       tilesize=range_, tablesize=4096, softening=soft,
                                       factor=factor)
    # HIP: tile sort, tiled deposit, fused solve with the cut-off, tiled gather (order 4), sweep
    mesh = PotentialMesh(N, L, nghosts=2)
    pos, mom = torch.tensor(pos0, device='cuda'), torch.tensor(mom0, device='cuda')
    ids = torch.arange(n, device='cuda')
    p1, m1, i1 = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    table = mesh.sort_particles(pos, mom, ids, p1, m1, i1)
    perm = i1.cpu().numpy()
    contribution = oracle.deposit_contribution(mass, dt, dt, N, L)
    C, E = oracle.poisson_constants(L, G, scale)
    mesh.deposit_tiled(p1, table, contribution, accumulate=False)
    mesh.poisson_solve(4, C, True, E)
    mesh.gather_kick_tiled(p1, m1, table, 4, mass*(-dt))
    kick_o = (mom_o - mom0)[perm]
    kick = m1.cpu().numpy() - mom0[perm]
    s = np.sqrt((kick_o**2).mean())
    assert np.abs(kick - kick_o).max() <= 1e-12*s + 5e-16*np.abs(mom0).max()
    nt = oracle.shortrange_tiling_shape(L, range_)
    tab, maxr2 = shortrange.get_shortrange_table(soft, scale, range_, 4096, 'spline',
                                                 torch.device('cuda'))
    ref = dmom_o[perm]
    big = max(np.abs(ref).max(), factor/scale**2)
    # the production sweep (half-tile cells)
    cells = mesh.shortrange_cells(p1, nt, L/nt)
    dmom = torch.zeros_like(p1)
    mesh.shortrange_sweep_cells(cells, dmom, cells, nt, tab, 4095/maxr2, range_**2, factor)
    out = dmom.cpu().numpy()
    assert np.abs(out - ref).max() <= 1e-12*big
    assert np.abs(out.sum(0)).max() <= 1e-10*big
    # cell list: a permutation, positions copied in cell order, every particle in its cell
    order, offset, pos_sorted = cells
    o = order.long()
    assert torch.equal(torch.sort(o)[0], torch.arange(n, device='cuda'))
    assert torch.equal(pos_sorted, p1[o])
    # every particle lies in the cell the offsets put it in (to rounding at the cell faces: the
    # tile index itself is pinned bit-exact by test_gpu_p3m.test_table_and_tiles)
    ext, nc = L/nt, 2*nt
    off = offset.long()
    assert int(off[0]) == 0 and int(off[-1]) == n and bool((off[1:] >= off[:-1]).all())
    cell = torch.searchsorted(off, torch.arange(n, device='cuda'), right=True) - 1
    c3 = torch.stack([cell//(nc*nc), (cell//nc) % nc, cell % nc], 1).double()
    lo, hi = c3*(ext/2), (c3 + 1)*(ext/2)
    tol = 1e-12*L
    assert bool(((pos_sorted >= lo - tol) & (pos_sorted <= hi + tol)).all())
    mesh.close()


def test_shortrange_cells_dense_and_two_components():
    """The half-tile sweep where its staging rounds and receiver chunks are exercised: a blob
    with thousands of particles per tile (several staging windows, > 64 receivers per cell
    column), a box face (periodic images), and receivers != suppliers (two components),
    against the oracle: the sums of a receiver set over another supplier set are those of the
    union on itself minus those of the receivers on themselves (the sweep is linear in the
    suppliers)."""
    import torch
    from concept_amd import shortrange
    from concept_amd.mesh import PotentialMesh
    from oracle import oracle
    L, N = 64.0, 64
    rng = np.random.default_rng(12)
    scale = 1.25*L/N
    range_ = 4.5*scale
    nt = int(L/range_*(1 + 2.220446049250313e-16))
    blob = np.mod(rng.normal(0.5, 1.6, (6000, 3)), L)     # dense, wrapped around the box corner
    thin = rng.uniform(0, L, (9000, 3))
    pos_a = torch.tensor(np.concatenate([blob, thin]), device='cuda')
    pos_b = torch.tensor(np.mod(rng.normal(L - 1.0, 3.0, (3000, 3)), L), device='cuda')
    mesh = PotentialMesh(N, L)
    tab, maxr2 = shortrange.get_shortrange_table(0.03, scale, range_, 4096, 'spline',
                                                 torch.device('cuda'))
    sc, r2 = 4095/maxr2, range_**2
    def orc(pos):
        return oracle.shortrange_kick(pos.cpu().numpy(), boxsize=L, scale=scale, range_=range_,
                                      tilesize=range_, tablesize=4096, softening=0.03,
                                      factor=0.7)[0]
    self_a, self_b = orc(pos_a), orc(pos_b)
    both = orc(torch.cat([pos_a, pos_b]))
    na = pos_a.shape[0]
    for rec, sup, ref, own in ((pos_a, pos_a, self_a, None), (pos_a, pos_b, both[:na], self_a),
                               (pos_b, pos_a, both[na:], self_b)):
        cr, cs = mesh.shortrange_cells(rec, nt, L/nt), mesh.shortrange_cells(sup, nt, L/nt)
        got = torch.zeros_like(rec)
        mesh.shortrange_sweep_cells(cr, got, cs, nt, tab, sc, r2, 0.7)
        big = float(np.abs(ref).max())
        want = ref if own is None else ref - own
        assert big > 0 and np.abs(got.cpu().numpy() - want).max() <= 1e-12*big, big
    mesh.close()


def test_stale_prepared_histogram_is_caught():
    """momenta changed between cg_gather_kick_tiled_prepare and cg_drift_sort.
    Through the library's own entry points the prepared state is dropped; changed behind its
    back, the sort notices (CG_ERR_STALE_HISTOGRAM) instead of silently losing particles."""
    import torch
    from concept_amd import lib
    from concept_amd.mesh import PotentialMesh
    L, N, n = 64.0, 64, 150_000
    mesh = PotentialMesh(N, L)
    rng = np.random.default_rng(2)
    pos = torch.tensor(rng.uniform(0, L, (n, 3)), device='cuda')
    mom = torch.tensor(rng.normal(0, 1.0, (n, 3)), device='cuda')
    p1, m1 = torch.empty_like(pos), torch.empty_like(mom)
    table = mesh.sort_particles(pos, mom, None, p1, m1, None)
    mesh.deposit_tiled(p1, table, 1.0)
    mesh.poisson_solve(4, -1.0, False, 0.0)
    dtm = 0.7
    # (a) through the library: cg_dmom_apply invalidates, the sort re-histograms
    mesh.gather_kick_tiled_prepare(p1, m1, table, 2, -0.3, dtm)
    dmom = torch.tensor(rng.normal(0, 3.0, (n, 3)), device='cuda')
    mesh.dmom_apply(m1, dmom)
    p2, m2 = torch.empty_like(p1), torch.empty_like(m1)
    t2 = mesh.drift_sort(p1, m1, None, p2, m2, None, dtm)
    assert (int(t2[-1].item()) & 0xffffffff) == n
    assert mesh.error_flags() == 0
    ref = mesh.drift_sort(p1, m1, None, torch.empty_like(p1), torch.empty_like(m1), None, dtm)
    assert torch.equal(t2, ref)
    # (b) behind its back (a torch op): detected
    mesh.gather_kick_tiled_prepare(p1, m1, table, 2, -0.3, dtm)
    m1.add_(dmom)
    mesh.drift_sort(p1, m1, None, p2, m2, None, dtm)
    assert mesh.error_flags() & lib.CG_ERR_STALE_HISTOGRAM
    assert mesh.error_flags() == 0  # reading clears
    # (c) ... unless the caller says so
    mesh.gather_kick_tiled_prepare(p1, m1, table, 2, -0.3, dtm)
    m1.add_(dmom)
    mesh.prepare_invalidate()
    t3 = mesh.drift_sort(p1, m1, None, p2, m2, None, dtm)
    assert mesh.error_flags() == 0 and (int(t3[-1].item()) & 0xffffffff) == n
    mesh.close()
