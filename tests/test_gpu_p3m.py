"""GPU parity tests of the P3M short-range tile sweep (A13-A15) and of
gravity('p3m', ...) end to end, against the reference-generated goldens and the
CPU oracle.  Bars: tile index of every particle bit-exact; Δmom <= 1e-12 of the
largest kick (pairs are summed in a different order than the reference's
tile/subtile/rung walk; r2 and the table index of a pair are bit-identical)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CASES = ['p3m_n8_g32', 'p3m_n12_g36_lattice', 'p3m_n16_g48_clustered', 'p3m_n8_g32_plummer']


def setup(g):
    import torch
    from concept_amd import commons
    from concept_amd.species import Component
    commons.load_params({
        'boxsize': float(g['boxsize']),
        'potential_options': {'gridsize': {'gravity': {'p3m': int(g['gridsize'])}},
                              'differentiation': {'matter': {'gravity': {'p3m': 4}}}},
        'select_forces': {'matter': {'gravity': 'p3m'}},
        'select_softening_length': {'matter': '0.03*boxsize/cbrt(N)'},
        # (non-default short-range parameters and softening kernel where the golden has them)
        'softening_kernel': str(g['softening_kernel']) if 'softening_kernel' in g else 'spline',
        'shortrange_params': {'gravity': {'scale': float(g['shortrange_scale']),
                                          'range': float(g['shortrange_range']),
                                          'tilesize': float(g['shortrange_tilesize']),
                                          'tablesize': int(g['shortrange_tablesize'])}},
    })
    c = Component('matter', 'matter', N=int(g['N']), mass=float(g['mass']))
    assert abs(c.softening_length - float(g['softening_length'])) < 1e-15
    return torch, c


def dense_everywhere(monkeypatch):
    """The cells sweep hands every tile of 3 particles and more to the dense tiles' sweep
    (cg_shortrange_dense.hip; the default threshold is 64)"""
    monkeypatch.setenv('CONCEPT_GPU_SR_DENSE_MIN', '3')


@pytest.mark.parametrize('sweep', ['cells', 'dense'])
@pytest.mark.parametrize('name', CASES)
def test_shortrange_vs_golden_and_oracle(golden, name, sweep, monkeypatch):
    """(the sweep: the half-tile cells, and the same with the dense tiles' sweep taking every
    populated tile; sweep None — tests/dist_component_worker.py, on several domains — runs
    the default)"""
    from concept_amd import interactions, shortrange
    from oracle import oracle
    if sweep == 'dense':
        dense_everywhere(monkeypatch)
    g = golden(name)
    torch, c = setup(g)
    c.populate(g['pos_after_short'], 'pos')
    c.populate(g['mom_after_short_sorted'], 'mom')
    nr = int(g['N_rungs'])
    sdt_rungs = {('a**(-3*w_eff₀-3*w_eff₁-1)', 'matter', 'matter'): g['dt_rungs_pair']}
    assert len(g['dt_rungs_pair']) == 3*nr - 1
    c.nullify_Δ('mom')
    interactions.gravity('p3m', [c], [c], sdt_rungs, 'short-range', False)
    out = c.host('Δmom')
    ref = g['dmom_short']
    factor = float(g['G_Newton'])*float(g['mass'])**2*float(g['dt_rungs_pair'][0])
    pair = factor/float(g['shortrange_scale'])**2
    scale = max(np.abs(ref).max(), pair)
    assert np.abs(out - ref).max() <= 1e-12*scale
    assert np.array_equal(c.host('pos'), g['pos_after_short'])  # positions untouched
    assert np.array_equal(c.host('mom'), g['mom_after_short_sorted'])  # mom only via apply_Δmom
    dm, _ = oracle.shortrange_kick(
        g['pos_after_short'], boxsize=float(g['boxsize']), scale=float(g['shortrange_scale']),
        range_=float(g['shortrange_range']), tilesize=float(g['shortrange_tilesize']),
        tablesize=int(g['shortrange_tablesize']), softening=float(g['softening_length']),
        factor=factor, kernel=str(g['softening_kernel']) if 'softening_kernel' in g else 'spline')
    assert np.abs(out - dm).max() <= 1e-12*scale
    # momentum conservation of the one-sided sweep
    assert np.abs(out.sum(0)).max() <= 1e-11*scale
    c.apply_Δmom()
    assert np.abs(c.host('mom') - (g['mom_after_short_sorted'] + ref)).max() <= 1e-12*scale


@pytest.mark.parametrize('name', CASES)
def test_table_and_tiles(golden, name):
    import torch
    from concept_amd import shortrange
    from concept_amd.mesh import PotentialMesh
    g = golden(name)
    table, maxr2 = shortrange.get_shortrange_table(
        float(g['softening_length']), float(g['shortrange_scale']), float(g['shortrange_range']),
        int(g['shortrange_tablesize']),
        str(g['softening_kernel']) if 'softening_kernel' in g else 'spline', torch.device('cuda'))
    ref = g['shortrange_table']
    t = table.cpu().numpy()
    assert maxr2 == float(g['shortrange_table_maxr2'])
    # math.erfc/exp (libm) vs scipy/numpy in the reference: last-bit differences only
    assert np.abs(t[:-1] - ref[:-1]).max() <= 4e-16*np.abs(ref[:-1]).max()
    # cell list: every particle in the tile Tiling.sort puts it in (species.py:775-780)
    L, nt = float(g['boxsize']), int(g['tiling_shape'][0])
    mesh = PotentialMesh(int(g['gridsize']), L)
    pos = torch.tensor(g['pos_after_short'], device='cuda')
    order, offset, _ = mesh.shortrange_tiles(pos, nt, L/nt)
    order, offset = order.cpu().numpy(), offset.cpu().numpy()
    eps = np.finfo(float).eps
    idx = ((g['pos_after_short'] - 0.0)*((1/(L/nt))*(1 - 2*eps))).astype(np.int64)
    tile = (idx[:, 0]*nt + idx[:, 1])*nt + idx[:, 2]
    n = len(tile)
    assert np.array_equal(np.sort(order[:n]), np.arange(n))
    assert np.array_equal(offset, np.concatenate([[0], np.cumsum(np.bincount(tile, minlength=nt**3))]))
    for t_ in np.unique(tile)[:50]:
        assert (tile[order[offset[t_]:offset[t_ + 1]]] == t_).all()


def test_p3m_full_kick_any(golden):
    """gravity('p3m', ..., 'any'): long-range (Gaussian cut-off mesh) + short-range in one
    call; long part checked against the golden long-range momenta."""
    from concept_amd import interactions
    g = golden('p3m_n8_g32')
    torch, c = setup(g)
    c.populate(g['pos_in'], 'pos')
    c.populate(g['mom_in'], 'mom')
    sdt = {'1': float(g['dt_1']), ('a**(-3*w_eff)', 'matter'): float(g['dt_kick']),
           ('a**(-3*w_eff-1)', 'matter'): float(g['dt_dens'])}
    interactions.gravity('p3m', [c], [c], sdt, 'long-range', False)
    kick = g['mom_after_long'] - g['mom_in']
    assert np.abs(c.host('mom') - g['mom_after_long']).max() <= 1e-12*np.sqrt((kick**2).mean()) \
        + 4e-16*np.abs(g['mom_in']).max()


def test_too_few_tiles_is_an_error():
    import torch
    from concept_amd import commons, interactions
    from concept_amd.lib import ConceptGPUError
    from concept_amd.species import Component
    commons.load_params({'boxsize': 16.0,
                         'potential_options': {'gridsize': {'gravity': {'p3m': 8}}},
                         'select_forces': {'matter': {'gravity': 'p3m'}}})
    c = Component('matter', 'matter', N=8, mass=1.0)
    c.nullify_Δ('mom')
    key = ('a**(-3*w_eff₀-3*w_eff₁-1)', 'matter', 'matter')
    with pytest.raises(ConceptGPUError):  # 16/(4.5*1.25*2) = 1.4 tiles < 4 (species.py:3971)
        interactions.gravity('p3m', [c], [c], {key: np.ones(23)}, 'short-range', False)


@pytest.mark.parametrize('name', ['steps_pm_n8_g16', 'steps_p3m_n8_g32'])
def test_timeloop_sequence_vs_reference(golden, name):
    """A18: concept_amd.stepper.timeloop (init half kicks, then drift -> kicks) against
    the reference's own sequence of Component.drift / gravity / apply_Δmom calls."""
    import torch
    from concept_amd import commons, interactions, stepper
    from concept_amd.species import Component
    g = golden(name)
    method = str(g['method'])
    commons.load_params({
        'boxsize': float(g['boxsize']),
        'potential_options': {'gridsize': {'gravity': {method: int(g['gridsize'])}},
                              'differentiation': {'matter': {'gravity': {
                                  method: int(g['diff_order'])}}}},
        'select_forces': {'matter': {'gravity': method}},
        'select_softening_length': {'matter': '0.03*boxsize/cbrt(N)'},
    })
    c = Component('matter', 'matter', N=int(g['N']), mass=float(g['mass']))
    c.populate(g['pos_in'], 'pos')
    c.populate(g['mom_in'], 'mom')
    dt, nr = float(g['dt']), int(g['N_rungs'])

    def integrals(kind):
        d = dt/2 if kind == 'init' else dt
        return {'1': d, 'a**(-2)': d*1.3, ('a**(-3*w_eff)', 'matter'): d*1.1,
                ('a**(-3*w_eff-1)', 'matter'): d*0.9}

    def rung_integrals(kind):
        d = dt/2 if kind == 'init' else dt
        return {('a**(-3*w_eff₀-3*w_eff₁-1)', 'matter', 'matter'): np.full(3*nr - 1, d*0.8)}

    L = float(g['boxsize'])
    seen = []

    def on_step(step):
        tag = 'init' if step == 0 else f'step{step}'
        pos, mom = c.host('pos'), c.host('mom')
        o = np.argsort(pos[:, 0], kind='stable')
        dx = np.abs(pos[o] - g['pos_' + tag])
        dx = np.minimum(dx, L - dx)
        assert dx.max() <= 1e-13*L, tag
        dm = np.abs(mom[o] - g['mom_' + tag]).max()
        assert dm <= 1e-12*np.abs(g['mom_' + tag]).max(), tag
        seen.append(tag)

    # drift inside timeloop uses integrals('full')['a**(-2)'] like the reference's scalars(dt)
    stepper.timeloop([c], 2, integrals, rung_integrals if method == 'p3m' else None, on_step)
    assert seen == ['init', 'step1', 'step2']
    if method != 'pm':
        return
    # Without a callback between kick and drift the PM loop streams: the kick of a step and the
    # drift of the next in one pass over particles kept in tile regions
    # (cg_gather_kick_drift_scatter).  Same end state as the reference's sequence — and ids and
    # the populated order come back with the particles.
    c2 = Component('matter', 'matter', N=int(g['N']), mass=float(g['mass']))
    c2.populate(g['pos_in'], 'pos')
    c2.populate(g['mom_in'], 'mom')
    assert interactions.pm_streaming_plan([c2]) is not None
    stepper.timeloop([c2], 2, integrals)
    pos, mom = c2.host('pos'), c2.host('mom')
    o = np.argsort(pos[:, 0], kind='stable')
    dx = np.abs(pos[o] - g['pos_step2'])
    assert np.minimum(dx, L - dx).max() <= 1e-13*L
    assert np.abs(mom[o] - g['mom_step2']).max() <= 1e-12*np.abs(g['mom_step2']).max()
    # row by row against the stepwise run (host() restores the populated order through `order`)
    dx = np.abs(pos - c.host('pos'))
    assert np.minimum(dx, L - dx).max() <= 1e-13*L
    assert np.abs(mom - c.host('mom')).max() <= 1e-12*np.abs(mom).max()
    assert np.array_equal(c2.host('ids'), c.host('ids'))
    assert c2.N_local == c.N_local


def _config3_positions(torch, dist, n, L, gen):
    pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen)
    if dist == 'clustered':  # bench.py --dist clustered: 80 % in 64 Gaussian blobs, σ = L/40
        centres = torch.rand((64, 3), dtype=torch.float64, device='cuda', generator=gen)*L
        which = torch.randint(0, 64, (n,), device='cuda', generator=gen)
        blob = centres[which] + torch.randn((n, 3), dtype=torch.float64, device='cuda',
                                            generator=gen)*(L/40)
        keep = torch.rand(n, dtype=torch.float64, device='cuda', generator=gen) < 0.2
        pos = torch.where(keep[:, None], pos*L, torch.remainder(blob, L))
        return pos.clamp_(0.0, L*(1 - 1e-13)).contiguous()
    return pos*(L*(1 - 1e-13))


@pytest.mark.parametrize('dist', ['uniform', 'clustered'])
def test_config3_size_shortrange_properties(dist):
    """BASELINE configs[2] size (256^3 particles, 512^3 mesh, default short-range
    parameters) through the sweep bench.py --p3m and shortrange.component_component run
    (cg_shortrange_cells + cg_shortrange_sweep_cells: half-tile cells), for the uniform and
    the clustered box of the bench: the one-sided sweep conserves momentum (Newton's third
    law holds pair by pair because both directions of a pair evaluate bit-identical r2 and
    table entries) and is invariant under a permutation of the particle memory.  (Against an
    independent pair enumeration: test_config3_size_shortrange_vs_oracle_sample.)"""
    import torch
    from concept_amd import commons, shortrange
    from concept_amd.mesh import PotentialMesh
    N, L, n = 512, 512.0, 256**3
    mesh = PotentialMesh(N, L)
    gen = torch.Generator(device='cuda').manual_seed(8)
    pos = _config3_positions(torch, dist, n, L, gen)
    scale = 1.25*L/N
    rng_ = 4.5*scale
    nt = int(L/rng_*(1 + commons.machine_ϵ))
    table, maxr2 = shortrange.get_shortrange_table(0.025*L/256, scale, rng_, 4096, 'spline',
                                                   pos.device)
    dm = torch.zeros_like(pos)
    cells = mesh.shortrange_cells(pos, nt, L/nt)
    mesh.shortrange_sweep_cells(cells, dm, cells, nt, table, 4095/maxr2, rng_**2, 1.0)
    scale_f = float(dm.abs().max())
    assert scale_f > 0
    assert float(dm.sum(0).abs().max()) <= 1e-9*float(dm.abs().sum(0).max())
    perm = torch.randperm(n, device='cuda', generator=gen)
    pos2 = pos[perm].contiguous()
    dm2 = torch.zeros_like(pos2)
    cells2 = mesh.shortrange_cells(pos2, nt, L/nt)
    mesh.shortrange_sweep_cells(cells2, dm2, cells2, nt, table, 4095/maxr2, rng_**2, 1.0)
    assert float((dm2 - dm[perm]).abs().max()) <= 1e-12*scale_f
    mesh.close()


@pytest.mark.parametrize('dist', ['uniform', 'clustered'])
def test_config3_size_shortrange_vs_oracle_sample(dist):
    """BASELINE configs[2]'s own size (256^3 particles, 512^3 mesh, default short-range
    parameters, test/concept_vs_gadget_p3m/param:10-46) against the ORACLE, uniform and
    clustered (where tile populations reach the staging limits): the receivers of every 97th
    tile plus those of the 32 most populated tiles (at most 64 of each), summed by the oracle
    over the 27 tiles around their own (orc_shortrange_sample: O(sample x 600) pairs).  Tile
    index of EVERY particle bit-exact (cg_shortrange_tiles against the oracle's tiling); Δmom <=
    1e-12 of the rms kick."""
    import torch
    from concept_amd import commons, shortrange
    from concept_amd.mesh import PotentialMesh
    from oracle import oracle
    N, L, n = 512, 512.0, 256**3
    mesh = PotentialMesh(N, L)
    gen = torch.Generator(device='cuda').manual_seed(8)
    pos = _config3_positions(torch, dist, n, L, gen)
    scale = 1.25*L/N
    rng_ = 4.5*scale
    nt = int(L/rng_*(1 + commons.machine_ϵ))
    soft = 0.025*L/256
    table, maxr2 = shortrange.get_shortrange_table(soft, scale, rng_, 4096, 'spline', pos.device)
    dm = torch.zeros_like(pos)
    lst = mesh.shortrange_cells(pos, nt, L/nt)
    mesh.shortrange_sweep_cells(lst, dm, lst, nt, table, 4095/maxr2, rng_**2, 1.0)
    del lst
    pos_h = pos.cpu().numpy()
    dm_h = dm.cpu().numpy()
    # the sample: by the oracle's own tile indices
    eps = np.finfo(float).eps
    idx = ((pos_h - 0.0)*((1/(L/nt))*(1 - 2*eps))).astype(np.int64)
    tile = (idx[:, 0]*nt + idx[:, 1])*nt + idx[:, 2]
    counts = np.bincount(tile, minlength=nt**3)
    order = np.argsort(tile, kind='stable')
    start = np.concatenate([[0], np.cumsum(counts)])
    chosen = set(range(0, nt**3, 97)) | set(np.argsort(counts)[-32:].tolist())
    sample = np.concatenate([order[start[t]:start[t] + min(counts[t], 64)] for t in sorted(chosen)])
    ref, tile_o = oracle.shortrange_sample(pos_h, sample, boxsize=L, scale=scale, range_=rng_,
                                           tilesize=rng_, tablesize=4096, softening=soft,
                                           factor=1.0)
    assert np.array_equal(tile_o, tile)
    # the list by tile IS the oracle's tiling, particle by particle
    lst = mesh.shortrange_tiles(pos, nt, L/nt)
    off = lst[1].cpu().numpy().astype(np.int64)
    assert np.array_equal(off, start)
    o = lst[0].cpu().numpy().astype(np.int64)[:n]
    assert np.array_equal(tile[o], np.repeat(np.arange(nt**3), counts))
    del lst
    rms = np.sqrt((dm_h**2).mean())
    assert rms > 0 and len(sample) > 50000
    assert counts.max() >= (1000 if dist == 'clustered' else 40)
    err = np.abs(dm_h[sample] - ref).max()
    assert err <= 1e-12*rms, (err/rms, dist)
    mesh.close()


@pytest.mark.parametrize('sweep', ['cells', 'dense'])
def test_adaptive_rungs_vs_reference(golden, sweep, monkeypatch):
    """A14/A16 with adaptive rungs on the GPU: RungStepper (initialize_rung_populations,
    kick_long, kick_short, driftkick_short with rung jumps; N_rungs = 4) against the
    reference's own main.py functions.  Rung indices bit-exact at every checkpoint."""
    import torch
    from concept_amd import commons, shortrange, stepper
    from concept_amd.species import Component
    if sweep == 'dense':
        dense_everywhere(monkeypatch)
    g = golden('rungs_p3m_n8_g32')
    commons.load_params({
        'boxsize': float(g['boxsize']),
        'potential_options': {'gridsize': {'gravity': {'p3m': int(g['gridsize'])}},
                              'differentiation': {'matter': {'gravity': {
                                  'p3m': int(g['diff_order'])}}}},
        'select_forces': {'matter': {'gravity': 'p3m'}},
        'select_softening_length': {'matter': '0.03*boxsize/cbrt(N)'},
        'N_rungs': int(g['N_rungs']),
    })
    c = Component('matter', 'matter', N=int(g['N']), mass=float(g['mass']))
    assert c.use_rungs
    c.populate(g['pos_in'], 'pos')
    c.populate(g['mom_in'], 'mom')
    L, dt = float(g['boxsize']), float(g['dt'])
    rs = stepper.RungStepper([c], stepper.static_integrals([c]),
                             fac_softening=float(g['fac_softening']),
                             Δt_jump_fac=float(g['dt_jump_fac']), Δt_reltol=float(g['dt_reltol']))
    rs.initialize_rung_populations(dt)
    assert np.array_equal(c.host('rung_indices'), g['rung_init'])
    assert c.rungs_N == list(g['rungs_N_init'])
    acc = c.host('Δmom')
    assert np.abs(acc - g['acc_init']).max() <= 1e-12*np.abs(g['acc_init']).max()

    def check(tag, rtag):
        pos, mom = c.host('pos'), c.host('mom')
        rung = c.host('rung_indices')
        o = np.argsort(pos[:, 0], kind='stable')
        assert np.array_equal(rung[o], g[rtag]), tag
        dx = np.abs(pos[o] - g['pos_' + tag])
        assert np.minimum(dx, L - dx).max() <= 1e-13*L, tag
        assert np.abs(mom[o] - g['mom_' + tag]).max() <= 1e-12*np.abs(g['mom_' + tag]).max(), tag

    rs.kick_long(dt, float('inf'), 'init')
    rs.kick_short(dt)
    check('init', 'rungs_after_init')
    for step in (1, 2):
        rs.base_step(dt)
        check(f'step{step}', f'rungs_step{step}')
        assert c.rungs_N == list(g[f'rungs_N_step{step}'])


def test_adaptive_rungs_knot_across_domains():
    """RungStepper (N_rungs = 4, rung jumps) on a box that is empty but for a knot across the
    face between two x-slab domains and a sparse halo around it: particles on high rungs next
    to a domain face, domains without particles, jumped rung indices travelling with the
    shipped suppliers.  Several domains against the single-domain run of the same calls: rung
    indices bit-exact, positions and momenta to summation order.  (On one domain: the run
    against itself; tests/test_gpu_distributed.py adds 2 and 4 domains.)"""
    from concept_amd import comm, commons, stepper
    from concept_amd.species import Component
    rng = np.random.default_rng(31)
    L, gs, mass, dt = 64.0, 64, 300.0, 0.05  # (G m ~ 0.013: the knot's core reaches rung 2-3)
    knot = np.array([L/2, 20.3, 41.7]) + rng.normal(0, 0.6, (1200, 3))
    halo = np.array([L/2, 20.3, 41.7]) + rng.normal(0, 5.0, (800, 3))
    pos0 = np.concatenate([knot, halo]) % L
    mom0 = rng.normal(0, 0.05, pos0.shape)
    n = pos0.shape[0]

    def run():
        commons.load_params({'boxsize': L, 'N_rungs': 4,
                             'potential_options': {'gridsize': {'gravity': {'p3m': gs}}},
                             'select_forces': {'all': {'gravity': 'p3m'}}})
        c = Component('m', 'matter', N=n, mass=mass)
        c.populate(pos0, 'pos')
        c.populate(mom0, 'mom')
        rs = stepper.RungStepper([c], stepper.static_integrals([c]))
        rs.initialize_rung_populations(dt)
        rs.kick_long(dt, float('inf'), 'init')
        rs.kick_short(dt)
        out = [c.host('rung_indices')]
        for step in (1, 2):
            rs.base_step(dt)
            out.append(c.host('rung_indices'))
        return c.host('pos'), c.host('mom'), out, list(c.rungs_N)
    active = comm.active()
    if active is not None:
        comm.shutdown()
    pos_ref, mom_ref, rungs_ref, pop_ref = run()
    if active is not None:
        comm.init()
    pos, mom, rungs, pop = run()
    assert len(set(rungs_ref[0].tolist())) > 1  # (more than one rung is populated)
    for a, b in zip(rungs, rungs_ref):
        assert np.array_equal(a, b)
    assert pop == pop_ref
    dx = np.abs(pos - pos_ref)
    assert np.minimum(dx, L - dx).max() <= 1e-12*L
    assert np.abs(mom - mom_ref).max() <= 1e-11*np.abs(mom_ref - mom0).max()


@pytest.mark.parametrize('seed', range(16))
def test_random_shortrange_vs_oracle(seed):
    _random_shortrange_vs_oracle(seed)


@pytest.mark.parametrize('tablesize', [4095, 4097, 2**13])
def test_shortrange_table_beside_and_beyond_the_lds_copy(tablesize):
    """The sweep's blocks of 4 x 2 tiles keep a copy of the look-up table in LDS when it has at
    most 4096 entries (cg_shortrange.hip: kSbTable); a longer one is read where it is.  Both
    sides of that limit against the oracle, on a box of 9 tiles a side (an interior of 7: blocks
    of four tiles, the last one leaving out the three it shares)."""
    _random_shortrange_vs_oracle(3, tablesize=tablesize, tiles=9)


def _random_shortrange_vs_oracle(seed, tablesize=None, tiles=None):
    """Differential test of the short-range part of gravity('p3m') over its parameters (force
    split scale, range, tile size, table size, softening kernel and length, particle number and
    clustering) against the CPU oracle.  Also run over 2 and 4 domains."""
    from concept_amd import commons, interactions
    from concept_amd.species import Component
    from oracle import oracle
    rng = np.random.default_rng(2000 + seed)
    L = float(rng.choice([32.0, 50.0]))
    gs = int(rng.choice([32, 64]))
    scale = float(rng.uniform(1.0, 1.5))*L/gs
    range_ = float(rng.uniform(3.5, 5.0))*scale
    tilesize = range_*float(rng.choice([1.0, 1.2]))
    tablesize = int(rng.choice([1024, 4096])) if tablesize is None else tablesize
    if tiles is not None:
        L, gs = 50.0, 64
        range_ = L/tiles/1.0001
        scale = range_/4.5
        tilesize = range_
    kernel = str(rng.choice(['spline', 'plummer', 'none']))
    N = int(rng.integers(500, 4000))
    pos = rng.uniform(0, L, (N, 3))
    if rng.integers(0, 2):  # half of the particles in a blob
        k = N//2
        pos[:k] = (rng.uniform(0, L, 3) + rng.normal(0, 0.04*L, (k, 3))) % L
    commons.load_params({
        'boxsize': L, 'N_rungs': 1, 'softening_kernel': kernel,
        'potential_options': {'gridsize': {'gravity': {'p3m': gs}}},
        'select_forces': {'all': {'gravity': 'p3m'}},
        'select_softening_length': {'all': f'{float(rng.uniform(0.01, 0.05))}*boxsize/cbrt(N)'},
        'shortrange_params': {'gravity': {'scale': scale, 'range': range_, 'tilesize': tilesize,
                                          'tablesize': tablesize}},
    })
    p = commons.params
    mass = float(rng.uniform(0.5, 3.0))
    c = Component('m', 'matter', N=N, mass=mass)
    c.populate(pos, 'pos')
    c.populate(np.zeros((N, 3)), 'mom')
    integral = 0.37
    c.nullify_Δ('mom')
    interactions.gravity('p3m', [c], [c],
                         {('a**(-3*w_eff₀-3*w_eff₁-1)', 'm', 'm'): np.full(2, integral)},
                         'short-range', False)
    factor = p.G_Newton*mass**2*integral
    dm, _ = oracle.shortrange_kick(pos, boxsize=L, scale=scale, range_=range_, tilesize=tilesize,
                                   tablesize=tablesize, softening=c.softening_length,
                                   factor=factor, kernel=kernel)
    got = c.host('Δmom')
    ref_scale = max(np.abs(dm).max(), factor/scale**2)
    assert np.abs(got - dm).max() <= 1e-11*ref_scale, (seed, kernel, scale, range_, tablesize, N)


@pytest.mark.parametrize('seed', range(4))
def test_random_p3m_timeloops_across_domains(seed):
    """P3M time loops (long-range + short-range kicks, drift, exchange, tile sort) over random
    draws of box, mesh, particle number, clustering and speed: several domains against the
    single-domain run of the same calls.  (On one domain: the run against itself;
    tests/test_gpu_distributed.py runs it on 2 and 4.)"""
    from concept_amd import comm, commons, stepper
    from concept_amd.species import Component
    rng = np.random.default_rng(4000 + seed)
    L = float(rng.choice([48.0, 64.0]))
    gs = int(rng.choice([32, 64]))  # (several domains: power-of-two meshes)
    n = int(rng.integers(1000, 6000))
    mass, d = float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.05, 0.3))
    nsteps = int(rng.integers(1, 4))
    pos0 = rng.uniform(0, L, (n, 3))
    kind = int(rng.integers(0, 3))
    if kind == 1:    # half in a blob
        pos0[:n//2] = (rng.uniform(0, L, 3) + rng.normal(0, 0.05*L, (n//2, 3))) % L
    elif kind == 2:  # everything in a slab along x (most domains start empty)
        pos0[:, 0] = (rng.uniform(0, L) + rng.uniform(0, L/6, n)) % L
    speed = float(rng.choice([0.2, 1.5, 4.0]))*(L/gs)
    mom0 = rng.normal(0, speed*mass/d, (n, 3))

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
        stepper.timeloop([c], nsteps, integrals, rung_integrals)
        return c.host('pos'), c.host('mom'), c.host('ids')
    active = comm.active()
    if active is not None:
        comm.shutdown()
    pos_ref, mom_ref, ids_ref = run()
    if active is not None:
        comm.init()
    pos, mom, ids = run()
    assert np.array_equal(ids, ids_ref)
    dx = np.abs(pos - pos_ref)
    assert np.minimum(dx, L - dx).max() <= 1e-12*L, seed
    kick = np.abs(mom_ref - mom0).max()
    assert np.abs(mom - mom_ref).max() <= 1e-11*kick + 4e-16*np.abs(mom_ref).max(), seed


@pytest.mark.parametrize('cell_centered', [True, False])
def test_shortrange_two_components_receivers_not_suppliers(cell_centered):
    """gravity('p3m', [A, B], [B], 'short-range'): A is kicked by B, and B — a receiver too —
    by itself and, reciprocally, by A, which is NOT among the suppliers (gravity.py:341-349
    kicks both partners of a pair).  On several domains A's boundary particles therefore have
    to travel as well, and the slab faces follow the grid's centring (vertex-
    centred: [x0*cell, (x0 + nxl)*cell)).  Checked against the oracle on the union
    (equal masses and softening: the kicks are linear in the supplier set) and, on several
    domains, against the single-domain run."""
    from concept_amd import comm, commons, interactions
    from concept_amd.species import Component
    from oracle import oracle
    rng = np.random.default_rng(77)
    L, gs, nA, nB = 64.0, 64, 3000, 2500
    # particles crowd the slab faces of a 4-domain run (x = 0, 16, 32, 48 cells, and half a cell
    # above: the faces of the cell-centred grid)
    def cloud(n):
        pos = rng.uniform(0, L, (n, 3))
        k = n//2
        pos[:k, 0] = (rng.choice([0.0, 16.0, 32.0, 48.0], k) + rng.uniform(-0.2, 0.8, k)) % L
        return pos
    posA, posB = cloud(nA), cloud(nB)
    integral, mass = 0.41, 1.7

    def run():
        commons.load_params({
            'boxsize': L, 'N_rungs': 1, 'cell_centered': cell_centered,
            'potential_options': {'gridsize': {'gravity': {'p3m': gs}}},
            'select_forces': {'all': {'gravity': 'p3m'}},
            'select_softening_length': {'all': 0.02}})
        A = Component('A', 'matter', N=nA, mass=mass)
        B = Component('B', 'matter', N=nB, mass=mass)
        for c, pos in ((A, posA), (B, posB)):
            c.populate(pos, 'pos')
            c.populate(np.zeros_like(pos), 'mom')
            c.nullify_Δ('mom')
        key = 'a**(-3*w_eff₀-3*w_eff₁-1)'
        sdt = {(key, x, y): np.full(2, integral) for x in 'AB' for y in 'AB'}
        interactions.gravity('p3m', [A, B], [B], sdt, 'short-range', False)
        return A.host('Δmom'), B.host('Δmom'), commons.params
    dA, dB, p = run()
    sr = commons.resolve_shortrange(p, gs)
    factor = p.G_Newton*mass**2*integral
    kw = dict(boxsize=L, scale=sr['scale'], range_=sr['range'], tilesize=sr['tilesize'],
              tablesize=sr['tablesize'], softening=0.02, factor=factor, kernel='spline')
    both, _ = oracle.shortrange_kick(np.concatenate([posA, posB]), **kw)
    onlyA, _ = oracle.shortrange_kick(posA, **kw)
    scale = max(np.abs(both).max(), factor/sr['scale']**2)
    assert np.abs(dA - (both[:nA] - onlyA)).max() <= 1e-11*scale   # A by B
    assert np.abs(dB - both[nA:]).max() <= 1e-11*scale             # B by A and B
    active = comm.active()
    if active is not None and active.world > 1:
        comm.shutdown()
        try:
            sA, sB, _ = run()
        finally:
            comm.init()
        assert np.abs(dA - sA).max() <= 1e-12*scale
        assert np.abs(dB - sB).max() <= 1e-12*scale


def test_populate_after_sort_lands_on_the_right_particles():
    """populate() of further columns after the particles have been reordered (tile_sort, a
    drift with its exchange, the streaming loop): data is given in the caller's row order and
    must reach the particles that stood in those rows (species.py populate)."""
    from concept_amd import commons
    from concept_amd.species import Component
    rng = np.random.default_rng(5)
    L, n = 32.0, 4000
    commons.load_params({'boxsize': L, 'potential_options': {'gridsize': {'gravity': {'pm': 32}}},
                         'select_forces': {'all': {'gravity': 'pm'}}})
    c = Component('m', 'matter', N=n, mass=1.0)
    pos, mom = rng.uniform(0, L, (n, 3)), rng.normal(0, 1, (n, 3))
    c.populate(pos, 'pos')
    c.populate(mom, 'mom')
    c.drift({'a**(-2)': 0.7})
    c.tile_sort()
    mom2 = rng.normal(0, 1, (n, 3))
    c.populate(mom2[:, 1], 'momy')
    c.populate(np.arange(n)[::-1].copy(), 'ids')
    want = mom.copy()
    want[:, 1] = mom2[:, 1]
    assert np.array_equal(c.host('mom'), want)
    assert np.array_equal(c.host('ids'), np.arange(n)[::-1])
    assert np.array_equal(c.host('pos'), (pos + mom*0.7) % L)
    # new positions for the same rows: the other columns stay with their rows
    pos2 = rng.uniform(0, L, (n, 3))
    c.populate(pos2[:, 0], 'posx')
    if c.nprocs == 1:
        got = c.host('pos')
        assert np.array_equal(got[:, 0], pos2[:, 0])
        assert np.array_equal(c.host('mom'), want)


@pytest.mark.parametrize('n,N_rungs', [(0, 8), (5, 8), (4096, 1), (1000003, 8), (300007, 10)])
def test_rung_populations(n, N_rungs):
    """cg_rung_populations (Component.set_rungs_N, species.py:2560-2587) against numpy.bincount;
    empty and ragged lengths, one rung only, more rungs than the default"""
    import torch
    from concept_amd.mesh import PotentialMesh
    mesh = PotentialMesh(16, 16.0)
    rng = np.random.default_rng(n + N_rungs)
    # (most particles on the low rungs, as in a run)
    r = np.minimum(rng.geometric(0.5, n) - 1, N_rungs - 1).astype(np.int8)
    got = mesh.rung_populations(torch.as_tensor(r, device='cuda'), N_rungs).cpu().numpy()
    assert got.dtype == np.int64 and got.shape == (N_rungs,)
    assert np.array_equal(got, np.bincount(r.astype(np.int64), minlength=N_rungs))


@pytest.mark.parametrize('k', [1, 3, 8])
def test_sparse_shortrange_equals_the_cells_sweep(k):
    """cg_shortrange_sparse (a handful of receivers on the top rungs against all suppliers, no
    cell list, nearest periodic image) against the cells sweep restricted to the same active
    rungs: the same sums up to the order of the additions; receivers chosen in dense spots and
    next to the box faces and corners (periodic images in one, two and three dimensions), a
    receiver and a supplier component that differ, momentum buffers accumulated not
    overwritten."""
    import torch
    from concept_amd import commons, shortrange
    from concept_amd.mesh import PotentialMesh
    N, L = 64, 64.0
    mesh = PotentialMesh(N, L)
    rng = np.random.default_rng(70 + k)
    n_r, n_s = 20000, 30011
    pos_r = rng.uniform(0, L, (n_r, 3))
    pos_s = rng.uniform(0, L, (n_s, 3))
    # the active receivers: box corner, an edge, a face, a clump of suppliers, the rest anywhere
    special = np.array([[1e-9, L - 1e-9, 0.3], [L - 0.2, 0.1, 31.0], [12.0, 40.0, L - 1e-7],
                        [20.0, 20.0, 20.0]])
    pos_s[:2000] = 20.0 + rng.normal(0, 0.4, (2000, 3))
    active = rng.choice(n_r, k, replace=False)
    pos_r[active[:min(k, 4)]] = special[:min(k, 4)]
    pos_r_t = torch.as_tensor(pos_r, device='cuda')
    pos_s_t = torch.as_tensor(np.mod(pos_s, L), device='cuda')
    scale = 1.25*L/N
    rng_ = 4.5*scale
    nt = int(L/rng_*(1 + commons.machine_ϵ))
    table, maxr2 = shortrange.get_shortrange_table(0.03*L/27, scale, rng_, 4096, 'spline',
                                                   pos_r_t.device)
    N_rungs = 8
    rung = np.zeros(n_r, dtype=np.int8)
    rung[active] = rng.integers(5, 8, k)
    jumped = rung.copy()
    jumped[active[0]] = rung[active[0]] + N_rungs      # flagged to jump up: another factor
    factors = torch.as_tensor(rng.uniform(0.5, 2.0, 3*N_rungs - 1), device='cuda')
    rung_t, jumped_t = torch.as_tensor(rung, device='cuda'), torch.as_tensor(jumped, device='cuda')
    base = torch.as_tensor(rng.normal(0, 1e-3, (n_r, 3)), device='cuda')
    # the cells sweep, rungs >= 5 active
    ref = base.clone()
    cr = mesh.shortrange_cells(pos_r_t, nt, L/nt)
    cs = mesh.shortrange_cells(pos_s_t, nt, L/nt)
    mesh.shortrange_sweep_cells(cr, ref, cs, nt, table, 4095/maxr2, rng_**2, 0.0,
                                (factors, rung_t, jumped_t, 5))
    got = base.clone()
    rows = torch.nonzero(rung_t >= 5).flatten()
    assert rows.numel() == k
    mesh.shortrange_sparse(pos_r_t, rows, got, pos_s_t, table, 4095/maxr2, rng_**2, 0.0,
                           (factors, jumped_t))
    kick = (ref - base).abs().max()
    assert float(kick) > 0
    assert float((got - ref).abs().max()) <= 1e-12*float(kick)
    others = torch.ones(n_r, dtype=torch.bool, device='cuda')
    others[rows] = False
    assert bool((got[others] == base[others]).all())   # nobody else was touched
    # without rungs: one factor for all
    ref2, got2 = torch.zeros_like(base), torch.zeros_like(base)
    mesh.shortrange_sweep_cells(cr, ref2, cs, nt, table, 4095/maxr2, rng_**2, 1.7)
    mesh.shortrange_sparse(pos_r_t, rows, got2, pos_s_t, table, 4095/maxr2, rng_**2, 1.7)
    assert float((got2[rows] - ref2[rows]).abs().max()) <= 1e-12*float(ref2[rows].abs().max())
    from concept_amd.lib import ConceptGPUError
    with pytest.raises(ConceptGPUError, match='active receivers'):
        mesh.shortrange_sparse(pos_r_t, torch.arange(9, device='cuda'), got2, pos_s_t, table,
                               4095/maxr2, rng_**2, 1.0)
    mesh.close()


@pytest.mark.parametrize('nt_box', [(64, 64.0), (24, 24.0)])
def test_active_first_cell_list_and_its_sweep(nt_box):
    """cg_shortrange_cells_rungs / cg_shortrange_sweep_cells_active (the list of a sub-step:
    the particles on active rungs first in every cell — the reference's lists by tile and rung,
    species.py tiles_rungs_N, interactions.py:1688-1761): the list's invariants, and its sweep
    against cg_shortrange_sweep_cells_rungs on a plain list for every lowest active rung —
    receivers that are not the suppliers, receivers on the box faces and corners (the blocks
    that reach across the faces), a box of 4 tiles a side (every tile on a face), factors by
    the jumped rung index, momentum buffers accumulated, inactive particles untouched."""
    import torch
    from concept_amd import commons, shortrange
    from concept_amd.lib import ConceptGPUError
    from concept_amd.mesh import PotentialMesh
    N, L = nt_box
    mesh = PotentialMesh(N, L)
    rng = np.random.default_rng(606)
    n_r, n_s = 30000 if N == 64 else 4000, 41003 if N == 64 else 5003
    pos_r = rng.uniform(0, L, (n_r, 3))
    pos_s = rng.uniform(0, L, (n_s, 3))
    pos_s[:3000] = np.mod(L - 0.5 + rng.normal(0, 0.7, (3000, 3)), L)   # a clump on the corner
    pos_r[:6] = [[1e-9, L - 1e-9, 0.3], [L - 0.2, 0.1, L/2], [L/3, L/2, L - 1e-7],
                 [L - 1e-9, L - 1e-9, L - 1e-9], [0.0, 0.0, 0.0], [L/2, 1e-8, L - 1e-8]]
    pos_r_t = torch.as_tensor(pos_r, device='cuda')
    pos_s_t = torch.as_tensor(pos_s, device='cuda')
    scale = 1.25*L/N
    rng_ = 4.5*scale
    nt = int(L/rng_*(1 + commons.machine_ϵ))
    table, maxr2 = shortrange.get_shortrange_table(0.03*L/27, scale, rng_, 4096, 'spline',
                                                   pos_r_t.device)
    N_rungs = 8
    rung = rng.choice(5, n_r, p=[0.5, 0.25, 0.13, 0.1, 0.02]).astype(np.int8)
    rung[:6] = 4
    jumped = rung.copy()
    flag = rng.random(n_r) < 0.05
    jumped[flag & (rung < 4)] += 2*N_rungs       # up
    jumped[flag & (rung == 4)] += N_rungs        # down
    factors = torch.as_tensor(rng.uniform(0.5, 2.0, 3*N_rungs - 1), device='cuda')
    rung_t, jumped_t = torch.as_tensor(rung, device='cuda'), torch.as_tensor(jumped, device='cuda')
    base = torch.as_tensor(rng.normal(0, 1e-3, (n_r, 3)), device='cuda')
    plain_r = mesh.shortrange_cells(pos_r_t, nt, L/nt)
    cs = mesh.shortrange_cells(pos_s_t, nt, L/nt)
    assert len(mesh.shortrange_cells(pos_r_t, nt, L/nt, (rung_t, jumped_t, 0))) == 3
    for la in (1, 2, 3, 4, 5):
        act_r = mesh.shortrange_cells(pos_r_t, nt, L/nt, (rung_t, jumped_t, la))
        order, offset, pos_sorted, nact = (t.cpu().numpy() for t in act_r[:4])
        # the same cells with the same members as the plain list
        assert np.array_equal(offset, plain_r[1].cpu().numpy())
        assert np.array_equal(np.sort(order[:n_r]), np.arange(n_r))
        assert np.array_equal(pos_sorted[:n_r], pos_r[order[:n_r]])
        cell_of_row = np.repeat(np.arange(offset.size - 1), np.diff(offset))
        rank = np.arange(n_r) - offset[cell_of_row]
        active_row = rung[order[:n_r]] >= la
        assert np.array_equal(active_row, rank < nact[cell_of_row])   # the active ones first
        assert nact.sum() == (rung >= la).sum()
        ref, got = base.clone(), base.clone()
        mesh.shortrange_sweep_cells(plain_r, ref, cs, nt, table, 4095/maxr2, rng_**2, 0.0,
                                    (factors, rung_t, jumped_t, la))
        mesh.shortrange_sweep_cells(act_r, got, cs, nt, table, 4095/maxr2, rng_**2, 0.0,
                                    (factors, rung_t, jumped_t, la))
        # ... and cell by cell (one wavefront per cell that holds an active receiver)
        got_c = base.clone()
        mesh.shortrange_sweep_cells(act_r, got_c, cs, nt, table, 4095/maxr2, rng_**2, 0.0,
                                    (factors, rung_t, jumped_t, la), int((rung >= la).sum()))
        inactive = rung_t < la
        assert bool((got[inactive] == base[inactive]).all())
        assert bool((got_c[inactive] == base[inactive]).all())
        if la == 5:
            assert bool((got == base).all()) and bool((got_c == base).all())
            continue
        kick = (ref - base)[~inactive]
        assert float(kick.abs().max()) > 0
        assert float((got - ref).abs().max()) <= 1e-12*float(kick.pow(2).mean().sqrt())
        assert float((got_c - ref).abs().max()) <= 1e-12*float(kick.pow(2).mean().sqrt())
        # as a suppliers' list it is a plain list
        got2 = base.clone()
        mesh.shortrange_sweep_cells(plain_r, got2, act_r, nt, table, 4095/maxr2, rng_**2, 1.3)
        ref2 = base.clone()
        mesh.shortrange_sweep_cells(plain_r, ref2, plain_r, nt, table, 4095/maxr2, rng_**2, 1.3)
        assert float((got2 - ref2).abs().max()) <= 1e-12*float((ref2 - base).pow(2).mean().sqrt())
        # a bound below the number of active receivers: the flag is raised, not a silent loss
        if la == 1:
            short = base.clone()
            mesh.shortrange_sweep_cells(act_r, short, cs, nt, table, 4095/maxr2, rng_**2, 0.0,
                                        (factors, rung_t, jumped_t, la), int((rung >= la).sum()) - 5)
            with pytest.raises(ConceptGPUError, match='more receivers on active rungs'):
                mesh.check_errors()
            assert mesh.error_flags() == 0          # (read and cleared)
        # a list made for other rungs is refused
        with pytest.raises(ConceptGPUError, match='other rungs'):
            mesh.shortrange_sweep_cells(act_r, got, cs, nt, table, 4095/maxr2, rng_**2, 0.0,
                                        (factors, rung_t, jumped_t, la + 1))
    mesh.close()


def test_dense_sweep_on_the_randomised_and_multi_component_cases(monkeypatch):
    """The dense tiles' sweep (Hilbert sub-cell order, supplier quads culled by their boxes;
    here from 3 particles per tile on instead of 64) through the cases the cells sweep is fuzzed
    with — the random parameter draws, receivers that are not suppliers, the knot with adaptive
    rungs across domains, whole random P³M time loops with rungs — and, with the default
    threshold, a box of 1576 particles per tile against the cells sweep by itself."""
    import torch
    from concept_amd import commons, shortrange
    from concept_amd.mesh import PotentialMesh
    dense_everywhere(monkeypatch)
    for seed in range(16):
        test_random_shortrange_vs_oracle(seed)
    for cell_centered in (True, False):
        test_shortrange_two_components_receivers_not_suppliers(cell_centered)
    test_adaptive_rungs_knot_across_domains()
    for seed in range(2):
        test_random_p3m_timeloops_across_domains(seed)
    monkeypatch.delenv('CONCEPT_GPU_SR_DENSE_MIN')
    N, n = 64, 100**3
    mesh = PotentialMesh(N, float(N))
    gen = torch.Generator(device='cuda').manual_seed(11)
    pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen)*(N*(1 - 1e-13))
    scale = 1.25
    rng_ = 4.5*scale
    nt = int(N/rng_*(1 + commons.machine_ϵ))
    table, maxr2 = shortrange.get_shortrange_table(0.01, scale, rng_, 4096, 'spline', pos.device)
    cells = mesh.shortrange_cells(pos, nt, N/nt)
    out = []
    for dense in ('0', '64'):
        monkeypatch.setenv('CONCEPT_GPU_SR_DENSE_MIN', dense)
        dm = torch.zeros_like(pos)
        mesh.shortrange_sweep_cells(cells, dm, cells, nt, table, 4095/maxr2, rng_**2, 1.0)
        out.append(dm)
    rms = float(out[0].pow(2).mean().sqrt())
    assert float((out[1] - out[0]).abs().max()) <= 1e-12*rms
    assert float(out[1].sum(0).abs().max()) <= 1e-9*rms*n**0.5   # Newton's third law
    mesh.close()


@pytest.mark.gpu
@pytest.mark.parametrize('lowest', [0, 1, 3])
def test_dense_sweep_with_active_rungs_equals_the_cells_sweep(lowest, monkeypatch):
    """cg_shortrange_sweep_cells_rungs on a clustered box (blobs of several hundred particles per
    tile; rungs assigned so that the upper ones sit in the blobs, some flagged to jump; receivers
    and suppliers two different components): the sub-step for the rungs >= lowest with the
    dense tiles' sweep — tiles dense with ACTIVE receivers, by the default threshold and cost
    model — against the half-tile cells everywhere (CONCEPT_GPU_SR_DENSE_MIN=0).  Same sums; the
    inactive receivers untouched."""
    import torch
    from concept_amd import commons, shortrange
    from concept_amd.mesh import PotentialMesh
    N, L = 96, 96.0
    mesh = PotentialMesh(N, L)
    rng = np.random.default_rng(5)
    n_r, n_s = 160000, 150000

    def box(n):
        pos = rng.uniform(0, L, (n, 3))
        centres = np.array([[20.0, 20.0, 20.0], [0.5, 48.0, 95.5], [70.0, 1.0, 40.0]])
        k = int(0.7*n)
        pos[:k] = centres[rng.integers(0, 3, k)] + rng.normal(0, 2.5, (k, 3))
        return np.mod(pos, L), k
    pos_r, k_r = box(n_r)
    pos_s, _ = box(n_s)
    N_rungs = 8
    rung = rng.integers(0, 2, n_r).astype(np.int8)
    rung[:k_r] = rng.integers(0, 5, k_r)              # the blobs: rungs 0-4
    jumped = rung.copy()
    flag = rng.choice(n_r, 500, replace=False)
    jumped[flag] = rung[flag] + N_rungs                # flagged to jump up: another factor
    pos_r_t, pos_s_t = torch.as_tensor(pos_r, device='cuda'), torch.as_tensor(pos_s, device='cuda')
    rung_t, jumped_t = torch.as_tensor(rung, device='cuda'), torch.as_tensor(jumped, device='cuda')
    factors = torch.as_tensor(rng.uniform(0.5, 2.0, 3*N_rungs - 1), device='cuda')
    scale = 1.25*L/N
    rng_ = 4.5*scale
    nt = int(L/rng_*(1 + commons.machine_ϵ))
    table, maxr2 = shortrange.get_shortrange_table(0.02, scale, rng_, 4096, 'spline', 'cuda')
    cr = mesh.shortrange_cells(pos_r_t, nt, L/nt)
    cs = mesh.shortrange_cells(pos_s_t, nt, L/nt)
    base = torch.as_tensor(rng.normal(0, 1e-3, (n_r, 3)), device='cuda')
    out = []
    for dense in ('0', '64'):
        monkeypatch.setenv('CONCEPT_GPU_SR_DENSE_MIN', dense)
        dm = base.clone()
        mesh.shortrange_sweep_cells(cr, dm, cs, nt, table, 4095/maxr2, rng_**2, 0.0,
                                    (factors, rung_t, jumped_t, lowest))
        out.append(dm)
    kick = float((out[0] - base).abs().max())
    assert kick > 0
    assert float((out[1] - out[0]).abs().max()) <= 1e-12*kick
    idle = rung_t < lowest
    assert bool((out[1][idle] == base[idle]).all())
    mesh.close()
