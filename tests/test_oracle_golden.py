"""The CPU oracle against the golden vectors produced by importing the
reference (tests/golden/make_golden.py).  This is what pins the oracle.

Bars: CIC grid indices bit-exact; PM ('gravity' potential) intermediates and
momenta bit-exact (same libm sin, same pocketfft); 'gravity long-range'
(exp in the Poisson kernel: numpy's SIMD exp vs libm, 1 ulp) <= 1e-14 of the
field rms."""
import numpy as np
import pytest

from oracle import oracle

PM_CASES = ['pm_n8_g16', 'pm_n16_g32', 'pm_edge_g16', 'pm_n8_g16_d4', 'pm_n8_g16_d6',
            'pm_n8_g16_d8', 'pm_n8_g16_d1', 'pm_n8_g16_vertex', 'pm_n8_g16_deconv_up']
FIELDS = ['grid_deposit', 'slab_density_k', 'slab_potential_k', 'grid_potential', 'grid_force']


def run_oracle(g, **kw):
    pos = g['pos_in'].copy()
    mom = g['mom_in'].copy()
    sc = float(g['shortrange_scale']) if 'shortrange_scale' in g else None
    o = oracle.pm_long_range(
        pos, mom, mass=float(g['mass']), boxsize=float(g['boxsize']), gridsize=int(g['gridsize']),
        G_Newton=float(g['G_Newton']), dt_1=float(g['dt_1']), dt_dens=float(g['dt_dens']),
        dt_kick=float(g['dt_kick']), diff_order=int(g['diff_order']), shortrange_scale=sc,
        nghosts=int(g['nghosts']), cell_centered=bool(int(g['cell_centered'])),
        deconvolve=tuple(bool(v) for v in g['deconvolve']) if 'deconvolve' in g else (True, True),
        **kw)
    return o, mom


@pytest.mark.parametrize('name', PM_CASES)
def test_pm_bit_exact(golden, name):
    g = golden(name)
    o, mom = run_oracle(g)
    assert np.array_equal(o['cic_index_deposit'], g['cic_index_deposit'])
    assert np.array_equal(o['cic_index_gather'], g['cic_index_gather'])
    for k in FIELDS:
        assert np.array_equal(o[k], g[k]), k
    assert np.array_equal(mom, g['mom_after_long'])


def test_pm_large_checksums(golden):
    g = golden('pm_n32_g64')
    o, mom = run_oracle(g)
    assert np.array_equal(o['cic_index_deposit'], g['cic_index_deposit'])
    assert np.array_equal(o['cic_index_gather'], g['cic_index_gather'])
    for k in ['grid_deposit', 'slab_density_k', 'slab_potential_k', 'grid_potential']:
        assert np.array_equal(o[k].ravel()[::97], g[k + '_sample']), k
        assert o[k].sum() == float(g[k + '_sum'])
    assert np.array_equal(mom, g['mom_after_long'])


@pytest.mark.parametrize('name', ['p3m_n8_g32', 'p3m_n12_g36_lattice', 'p3m_n16_g48_clustered'])
def test_p3m_long_range(golden, name):
    g = golden(name)
    o, mom = run_oracle(g)
    assert np.array_equal(o['cic_index_deposit'], g['cic_index_deposit'])
    assert np.array_equal(o['cic_index_gather'], g['cic_index_gather'])
    if 'grid_deposit' in g:
        assert np.array_equal(o['grid_deposit'], g['grid_deposit'])
        assert np.array_equal(o['slab_density_k'], g['slab_density_k'])
        for k in ['slab_potential_k', 'grid_potential', 'grid_force']:
            rms = np.sqrt((g[k]**2).mean())
            assert np.abs(o[k] - g[k]).max() <= 1e-14*rms, k
    kick_ref = g['mom_after_long'] - g['mom_in']
    kick = mom - g['mom_in']
    scale = max(np.sqrt((kick_ref**2).mean()), 1e-300)
    assert np.abs(mom - g['mom_after_long']).max() <= 1e-13*max(scale, np.abs(g['mom_in']).max())


@pytest.mark.parametrize('name', PM_CASES + ['pm_n32_g64', 'p3m_n8_g32'])
def test_drift_bit_exact(golden, name):
    g = golden(name)
    pos = (g['drift_pos_in'] if 'drift_pos_in' in g else g['pos_in']).copy()
    mom = g['drift_mom_in'] if 'drift_mom_in' in g else g['mom_after_long']
    oracle.drift(pos, mom, float(g['drift_dt_over_mass']), float(g['boxsize']))
    assert np.array_equal(pos, g['drift_pos_out'])
    assert (pos >= 0).all() and (pos < float(g['boxsize'])).all()


def test_drift_wrap_edge_cases():
    L = 10.0
    pos = np.array([0.0, 9.999999999999998, 5.0, 1e-17, 9.5, 0.25], dtype=np.float64)
    mom = np.array([-1e-17, 1.0, 25.0, -1.0, 0.5, -30.25], dtype=np.float64)
    ref = np.mod(pos + mom*1.0, L)
    ref[ref == L] = 0
    out = oracle.drift(pos.copy(), mom, 1.0, L)
    assert np.array_equal(out, ref)
    assert (out >= 0).all() and (out < L).all()


def test_fast_build_agrees_to_rounding(golden):
    """The -ffast-math timing build is the CPU baseline; it must still be the
    same algorithm (agreement to rounding, not bit-exact)."""
    g = golden('pm_n16_g32')
    o, mom = run_oracle(g, fast=True)
    kick = g['mom_after_long'] - g['mom_in']
    assert np.abs(mom - g['mom_after_long']).max() <= 1e-10*np.abs(kick).max()


@pytest.mark.parametrize('name', ['p3m_n8_g32', 'p3m_n12_g36_lattice', 'p3m_n16_g48_clustered',
                                  'p3m_n8_g32_plummer'])
def test_p3m_shortrange(golden, name):
    """Short-range tile sweep: the force table bit-exact, Δmom to summation rounding
    (the oracle visits the pairs in a different order than the reference)."""
    g = golden(name)
    factor = float(g['G_Newton'])*float(g['mass'])**2*float(g['dt_rungs_pair'][0])
    dmom, table = oracle.shortrange_kick(
        g['pos_after_short'], boxsize=float(g['boxsize']), scale=float(g['shortrange_scale']),
        range_=float(g['shortrange_range']), tilesize=float(g['shortrange_tilesize']),
        tablesize=int(g['shortrange_tablesize']), softening=float(g['softening_length']),
        factor=factor, kernel=str(g['softening_kernel']) if 'softening_kernel' in g else 'spline')
    assert np.array_equal(table[:-1], g['shortrange_table'][:-1])
    assert oracle.shortrange_tiling_shape(float(g['boxsize']), float(g['shortrange_tilesize'])) \
        == int(g['tiling_shape'][0])
    ref = g['dmom_short']
    # scale: one pair at the force-split scale (the lattice case cancels to ~0 by symmetry,
    # test/multicomponent K1)
    pair = factor/float(g['shortrange_scale'])**2
    assert np.abs(dmom - ref).max() <= 1e-13*max(np.abs(ref).max(), pair)
    # Newton's third law: the sweep conserves momentum
    assert np.abs(dmom.sum(0)).max() <= 1e-12*max(np.abs(ref).max(), pair)


@pytest.mark.parametrize('name', ['p3m_n8_g32', 'p3m_n12_g36_lattice', 'p3m_n16_g48_clustered',
                                  'p3m_n8_g32_plummer'])
def test_p3m_shortrange_sample_is_pinned(golden, name):
    """orc_shortrange_sample (the one-sided sums the full-size GPU tests are checked against)
    pinned like the sweep itself: every particle as a sampled receiver against the
    reference-generated Δmom."""
    g = golden(name)
    factor = float(g['G_Newton'])*float(g['mass'])**2*float(g['dt_rungs_pair'][0])
    pos = g['pos_after_short']
    dmom, tile = oracle.shortrange_sample(
        pos, np.arange(len(pos)), boxsize=float(g['boxsize']),
        scale=float(g['shortrange_scale']), range_=float(g['shortrange_range']),
        tilesize=float(g['shortrange_tilesize']), tablesize=int(g['shortrange_tablesize']),
        softening=float(g['softening_length']), factor=factor,
        kernel=str(g['softening_kernel']) if 'softening_kernel' in g else 'spline')
    ref = g['dmom_short']
    pair = factor/float(g['shortrange_scale'])**2
    assert np.abs(dmom - ref).max() <= 1e-13*max(np.abs(ref).max(), pair)


def _step_scalars(dt):
    return dict(dt_1=dt, dt_am2=dt*1.3, dt_kick=dt*1.1, dt_dens=dt*0.9, dt_rung=dt*0.8)


@pytest.mark.parametrize('name', ['steps_pm_n8_g16', 'steps_p3m_n8_g32'])
def test_caller_sequence(golden, name):
    """A18: the order of kicks and drifts of main.timeloop (main.py:255-361), replayed
    through the oracle against the reference's own sequence (states sorted by x)."""
    g = golden(name)
    L, N, mass = float(g['boxsize']), int(g['gridsize']), float(g['mass'])
    p3m = str(g['method']) == 'p3m'
    pos, mom = g['pos_in'].copy(), g['mom_in'].copy()
    common = dict(mass=mass, boxsize=L, gridsize=N, G_Newton=float(g['G_Newton']),
                  diff_order=int(g['diff_order']), want_indices=False,
                  shortrange_scale=float(g['shortrange_scale']) if p3m else None)

    def kick_long(dt):
        s = _step_scalars(dt)
        oracle.pm_long_range(pos, mom, dt_1=s['dt_1'], dt_dens=s['dt_dens'], dt_kick=s['dt_kick'],
                             **common)

    def kick_short(dt):
        factor = float(g['G_Newton'])*mass**2*_step_scalars(dt)['dt_rung']
        dm, _ = oracle.shortrange_kick(
            pos, boxsize=L, scale=float(g['shortrange_scale']), range_=float(g['shortrange_range']),
            tilesize=float(g['shortrange_range']), tablesize=4096,
            softening=float(g['softening_length']), factor=factor)
        mom[...] += dm

    def check(tag, exact):
        o = np.argsort(pos[:, 0], kind='stable')
        if exact:
            assert np.array_equal(pos[o], g['pos_' + tag]) and np.array_equal(mom[o], g['mom_' + tag])
        else:
            assert np.abs(pos[o] - g['pos_' + tag]).max() <= 1e-13*L
            assert np.abs(mom[o] - g['mom_' + tag]).max() <= 1e-12*np.abs(g['mom_' + tag]).max()

    dt = float(g['dt'])
    kick_long(dt/2)
    if p3m:
        kick_short(dt/2)
    check('init', exact=not p3m)
    for step in (1, 2):
        oracle.drift(pos, mom, _step_scalars(dt)['dt_am2']/mass, L)
        if p3m:
            kick_short(dt)
        kick_long(dt)
        check(f'step{step}', exact=not p3m)


def test_adaptive_rungs_sequence(golden):
    """A14/A16 with adaptive rungs: the oracle's restatement of initialize_rung_populations,
    kick_short and driftkick_short (oracle/rungs.py) against the reference's own main.py
    functions (N_rungs = 4, two base steps with rung jumps).  Rung indices are integers:
    bit-exact at every checkpoint; positions and momenta to rounding."""
    from oracle import rungs
    g = golden('rungs_p3m_n8_g32')
    L, N, mass, nr = float(g['boxsize']), int(g['gridsize']), float(g['mass']), int(g['N_rungs'])
    G, dt = float(g['G_Newton']), float(g['dt'])
    scale, rng_ = float(g['shortrange_scale']), float(g['shortrange_range'])
    table, maxr2 = oracle.shortrange_table(float(g['softening_length']), scale, rng_, 4096)
    sr = dict(boxsize=L, nt=oracle.shortrange_tiling_shape(L, rng_), table=table, tablesize=4096,
              maxr2=maxr2, range=rng_)
    p = rungs.Particles(g['pos_in'], g['mom_in'], mass, float(g['softening_length']), nr)
    dtr = rungs.new_dt_rungs(nr)
    fs, jf, rt = float(g['fac_softening']), float(g['dt_jump_fac']), float(g['dt_reltol'])
    t = 0.0
    rungs.initialize_rung_populations(p, dt, t, sr, G, dtr, fs)
    assert np.array_equal(p.rung, g['rung_init'])
    assert np.array_equal(p.rungs_N, g['rungs_N_init'])
    assert np.abs(p.dmom - g['acc_init']).max() <= 1e-13*np.abs(g['acc_init']).max()

    def kick_long(d):
        oracle.pm_long_range(p.pos, p.mom, mass=mass, boxsize=L, gridsize=N, G_Newton=G, dt_1=d,
                             dt_dens=d, dt_kick=d, diff_order=int(g['diff_order']),
                             shortrange_scale=scale, want_indices=False)

    def drift(p_, dt_am2):
        oracle.drift(p_.pos, p_.mom, dt_am2/mass, L)

    def check(tag, rtag):
        o = np.argsort(p.pos[:, 0], kind='stable')
        assert np.array_equal(p.rung[o], g[rtag]), tag
        assert np.abs(p.pos[o] - g['pos_' + tag]).max() <= 1e-13*L, tag
        assert np.abs(p.mom[o] - g['mom_' + tag]).max() <= 1e-12*np.abs(g['mom_' + tag]).max()

    kick_long(dt/2)
    rungs.kick_short(p, dt, t, sr, G, dtr, fs)
    check('init', 'rungs_after_init')
    for step in (1, 2):
        rungs.driftkick_short(p, dt, t, float('inf'), sr, G, dtr, fs, jf, rt, drift)
        t += 0.5*dt
        kick_long(dt)
        t += 0.5*dt
        check(f'step{step}', f'rungs_step{step}')
        assert np.array_equal(p.rungs_N, g[f'rungs_N_step{step}'])
    assert int(g['rungs_N_step2'][3]) != int(g['rungs_N_init'][3])  # jumps did happen


# ---- fluid coupling (oracle/pm_general.py) --------------------
def _fluid_components(g, particle_diff):
    comps = []
    for c in range(int(g['n_particle_components'])):
        comps.append(dict(kind='particles', pos=g[f'p{c}_pos'], mom=g[f'p{c}_mom_in'].copy(),
                          mass=float(g[f'p{c}_mass']), dt_dens=float(g[f'p{c}_dt_dens']),
                          dt_kick=float(g[f'p{c}_dt_kick']), diff_order=particle_diff))
        if f'gridsizes_particles{c}_pm' in g:
            up, down = (int(v) for v in g[f'gridsizes_particles{c}_pm'])
            comps[-1].update(gridsize_up=up, gridsize_down=down)
    for c in range(int(g['n_fluid_components'])):
        comps.append(dict(kind='fluid', rho=g[f'f{c}_rho'], P=g[f'f{c}_P'],
                          J=g[f'f{c}_J_in'].copy(), dt_dens=float(g[f'f{c}_dt_dens']),
                          dt_kick=float(g[f'f{c}_dt_kick']), diff_order=2))
    return comps


@pytest.mark.parametrize('name', ['fluid_pm_n8_g16', 'fluid2_pm_n6_g12', 'multigrid_n8_g16',
                                  'multigrid_n8_up32_down24', 'tsc_bcc_n8_g16',
                                  'pcs_fcc_fourier_n8_g16', 'ngp_fluid_n8_g16',
                                  'cic_fcc_multigrid_n8', 'multigrid_n8_pow2',
                                  'cic_fcc_multigrid_pow2', 'cic_fcc_multigrid_vertex_pow2',
                                  'tsc_bcc_deconv_down_n8_g16'])
def test_general_particle_mesh_bit_exact(golden, name):
    """gravity('pm') with receivers = suppliers = particles + fluids: momenta, J grids and
    the k-space potential handed to every backward FFT, bit for bit; the multigrid cases
    add upstream / downstream grid sizes different from the global one (copy_modes)."""
    from oracle import pm_general
    g = golden(name)
    comps = _fluid_components(g, int(g['diff_order']))
    extra = {}
    if 'interpolation' in g:  # row 3: NGP / TSC / PCS, interlacing, Fourier differentiation
        extra = dict(interp_order={'NGP': 1, 'CIC': 2, 'TSC': 3, 'PCS': 4}[str(g['interpolation'])],
                     interlace=tuple(str(x) for x in g['interlace']))
    if 'cell_centered' in g:
        extra.update(cell_centered=bool(int(g['cell_centered'])), nghosts=int(g['nghosts']))
    if 'deconvolve' in g:
        extra.update(deconvolve=tuple(bool(v) for v in g['deconvolve']))
    out = pm_general.particle_mesh(
        comps, boxsize=float(g['boxsize']), gridsize=int(g['gridsize']),
        G_Newton=float(g['G_Newton']), dt_1=float(g['dt_1']), light_speed=float(g['light_speed']),
        **extra)
    npc = int(g['n_particle_components'])
    for c in range(npc):
        assert np.array_equal(comps[c]['mom'], g[f'p{c}_mom_out'])
    for c in range(int(g['n_fluid_components'])):
        assert np.array_equal(comps[npc + c]['J'], g[f'f{c}_J_out'])
        assert np.abs(g[f'f{c}_J_out'] - g[f'f{c}_J_in']).max() > 0
    n_slabs = len([k for k in g.files if k.startswith('slab_k_before_backward_')])
    assert len(out['slab_before_backward']) == n_slabs
    for i, slab in enumerate(out['slab_before_backward']):
        assert np.array_equal(slab, g[f'slab_k_before_backward_{i}'])


def test_general_particle_mesh_nonlinnu_shape(golden):
    """The three long-range interactions of an example_nonlinnu-shaped setup."""
    from oracle import pm_general
    g = golden('nonlinnu_like_n8')
    assert list(g['interactions']) == ['gravity|p3m|particles0|particles0',
                                       'gravity|pm|particles0|fluid0',
                                       'gravity|pm|fluid0|particles0,fluid0']
    part, fl = _fluid_components(g, 4)
    kw = dict(boxsize=float(g['boxsize']), G_Newton=float(g['G_Newton']), dt_1=float(g['dt_1']),
              light_speed=float(g['light_speed']))
    gs = int(g['gridsize'])
    scale = float(g['shortrange_scale'])*float(g['boxsize'])/(2*gs)
    up, down = (int(v) for v in g['gridsizes_particles0_p3m'])
    part.update(gridsize_up=up, gridsize_down=down)
    pm_general.particle_mesh([part], [part], gridsize=2*gs, shortrange_scale=scale, **kw)
    part['diff_order'] = int(g['differentiation_particles0_pm'])
    up, down = (int(v) for v in g['gridsizes_particles0_pm'])
    part.update(gridsize_up=up, gridsize_down=down)
    pm_general.particle_mesh([part], [fl], gridsize=gs, **kw)
    pm_general.particle_mesh([fl], [part, fl], gridsize=gs, **kw)
    assert np.array_equal(fl['J'], g['f0_J_out'])
    kick = np.abs(g['p0_mom_out'] - g['p0_mom_in']).max()
    assert np.abs(part['mom'] - g['p0_mom_out']).max() <= 1e-13*kick  # exp() of the cut-off


# ---- direct summation with the Ewald correction (oracle/pp.py) ----
def test_ewald_table_and_lookup(golden):
    from oracle import pp
    g = golden('pp_ewald_n4')
    grid = pp.ewald_tabulate(int(g['ewald_gridsize']))
    # libm here, SciPy/NumPy scalar calls in the pure-Python reference: to rounding
    assert np.abs(grid - g['ewald_grid']).max() <= 1e-14*np.abs(g['ewald_grid']).max()
    assert list(g['ewald_constants']) == [0.25, 3.6, 10, -3, 4, -4, 5]
    L = float(g['boxsize'])
    vals = np.array([pp.ewald_lookup(g['ewald_grid'], *pt, L) for pt in g['ewald_points']])
    assert np.array_equal(vals, g['ewald_values'])   # the CIC look-up itself: bit for bit


@pytest.mark.parametrize('name,periodic', [('pp_ewald_n4', True), ('ppnonperiodic_n4', False)])
def test_pp_kick_bit_exact(golden, name, periodic):
    from oracle import pp
    g = golden(name)
    factor = float(g['G_Newton'])*float(g['mass'])**2*float(g['dt_rungs_pair'][0])
    dm = pp.pp_kick(g['pos_in'], boxsize=float(g['boxsize']),
                    softening=float(g['softening_length']), factor=factor, periodic=periodic,
                    ewald_grid=g['ewald_grid'] if periodic else None)
    assert np.array_equal(dm, g['dmom'])
    assert np.array_equal(g['pos_in'], g['pos_after'])
