#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the output .npz
files are data (inputs + the reference's outputs and intermediates) and are
what travels.  One reference import per process (its parameters are module
globals), so every case is generated in its own subprocess:

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py pm_n8_g16  # one case (child mode)

Reference entry points exercised (file:line in /root/reference/src):
  interactions.gravity            interactions.py:2854
  interactions.particle_mesh      interactions.py:1985
  mesh.interpolate_particles      mesh.py:1512      (CIC deposit)
  mesh.interpolate_domaingrid_to_particles mesh.py:376 (CIC gather-kick)
  mesh.diff_domaingrid            mesh.py:4874
  species.Component.drift         species.py:2179
  gravity.gravity_pairwise_shortrange gravity.py:263 (P3M cases)
  main.timeloop                   main.py:102       (traj_* cases: whole runs a_begin -> 1 with
                                                     the matter + Λ clock, every
                                                     get_time_step_integrals call recorded)
  integration.init_time / scalefactor_integral  integration.py:864, 712 (through the above)
Intermediates are captured by rebinding module globals of the imported
pure-Python modules (commons.py:1268-1274 binds cimported names as plain
globals), never by editing reference files.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

# name -> dict(method, n (particles per dim or explicit), gridsize, boxsize, ...)
CASES = {
    # PM, default differentiation order 2, uniform random particles + random momenta
    'pm_n8_g16': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=1, dist='uniform',
                      diff=2, full=True),
    'pm_n16_g32': dict(method='pm', n=16, gridsize=32, boxsize=100.0, seed=2, dist='uniform',
                       diff=2, full=True),
    # boundary stress: particles exactly at 0, at cell centres/edges, next below boxsize
    'pm_edge_g16': dict(method='pm', n=0, gridsize=16, boxsize=64.0, seed=3, dist='edge',
                        diff=2, full=True),
    # differentiation order 4 (the p3m default, commons.py:3209-3237)
    'pm_n8_g16_d4': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=4, dist='uniform',
                         diff=4, full=True),
    # the other differentiation orders of diff_domaingrid (mesh.py:4874-5030): 6 and 8 (the
    # reference raises nghosts to 3 and 4 for them, commons.py:4428-4430) and the one-sided
    # order 1
    'pm_n8_g16_d6': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=36, dist='uniform',
                         diff=6, full=True),
    'pm_n8_g16_d8': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=37, dist='clustered',
                         diff=8, full=True),
    'pm_n8_g16_d1': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=38, dist='uniform',
                         diff=1, full=True),
    # other user parameters of the path: only the upstream deconvolution; Plummer softening
    # with a non-default short-range scale / range / table size
    'pm_n8_g16_deconv_up': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=41,
                                dist='uniform', diff=2, full=True,
                                extra="potential_options['deconvolve'] = "
                                      "{'gravity': {'pm': (True, False)}}\n"),
    'p3m_n8_g32_plummer': dict(method='p3m', n=8, gridsize=32, boxsize=32.0, seed=42,
                               dist='uniform', diff=4, full=True,
                               extra="softening_kernel = 'plummer'\n"
                                     "shortrange_params = {'gravity': {'scale': "
                                     "'1.1*boxsize/gridsize', 'range': '4.2*scale', "
                                     "'tablesize': 2048, 'subtiling': 2}}\n"),
    # vertex-centred grids (the user parameter cell_centered = False, commons.py)
    'pm_n8_g16_vertex': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=39,
                             dist='uniform', diff=2, full=True, vertex=True),
    # clustered ("Zel'dovich-like" displaced lattice), in/out only
    'pm_n32_g64': dict(method='pm', n=32, gridsize=64, boxsize=256.0, seed=5, dist='lattice',
                       diff=2, full=False),
    # P3M long-range (Gaussian cut-off in the Poisson kernel) + short-range
    'p3m_n8_g32': dict(method='p3m', n=8, gridsize=32, boxsize=32.0, seed=6, dist='uniform',
                       diff=4, full=True),
    'p3m_n12_g36_lattice': dict(method='p3m', n=12, gridsize=36, boxsize=36.0, seed=7,
                                dist='perfect_lattice', diff=4, full=False),
    'p3m_n16_g48_clustered': dict(method='p3m', n=16, gridsize=48, boxsize=48.0, seed=8,
                                  dist='clustered', diff=4, full=False),
    # the callers (A18): init half kicks, then two base steps of drift -> kicks in the order
    # of main.timeloop (main.py:255-361), with plain scalars
    'steps_pm_n8_g16': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=11,
                            dist='uniform', diff=2, steps=2),
    'steps_p3m_n8_g32': dict(method='p3m', n=8, gridsize=32, boxsize=32.0, seed=12,
                             dist='uniform', diff=4, steps=2),
    # adaptive rungs (A14/A16): the reference's own initialize_rung_populations / kick_long /
    # kick_short / driftkick_short (main.py) with N_rungs = 4 on a clustered set
    'rungs_p3m_n8_g32': dict(method='p3m', n=8, gridsize=32, boxsize=32.0, seed=13,
                             dist='clustered', diff=4, rungs=4, steps=2),
    # fluid coupling on the shared mesh.  One PM interaction with
    # receivers = suppliers = [particles, fluid] (equal grid sizes): fluid ϱ added to the
    # upstream grid, per-representation deconvolution, fluid kick J += -ᔑdt (ϱ + c⁻²𝒫) ∇φ
    'fluid_pm_n8_g16': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=21,
                            dist='uniform', diff=2, fluid=dict(gridsize=16)),
    # two fluids + two particle components, differentiation order 4 for the particles
    # the example_nonlinnu shape (param/example_nonlinnu:36-45): particles default P³M on
    # mesh 32, fluid on grid 16, global PM grid 16 -> three long-range interactions per kick:
    # (p3m, [particles], [particles]), (pm, [particles], [fluid]), (pm, [fluid], [particles, fluid])
    'nonlinnu_like_n8': dict(method='pm', n=8, gridsize=16, boxsize=32.0, seed=23,
                             dist='uniform', diff=2, fluid=dict(gridsize=16), nonlinnu=True),
    # unequal upstream / global / downstream grid sizes (copy_modes
    # up- and down-scaling with the cell-centring phase, mesh.py:1018-1326): global 16,
    # particles0 (up 24, down 12), particles1 (up 16, down 32), fluid on its own grid 8
    'multigrid_n8_g16': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=24,
                             dist='clustered', diff=2, fluid=dict(gridsize=8),
                             particle_components=2,
                             component_gridsizes={'particles0': (24, 12), 'particles1': (16, 32)}),
    # the same shape on power-of-two grids (the multi-GPU FFT's sizes), for the runs over
    # 2 and 4 x-slab domains: global 16, particles0 (up 32, down 16), particles1 (up 16,
    # down 32), fluid on its own grid 32
    'multigrid_n8_pow2': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=34,
                              dist='clustered', diff=2, fluid=dict(gridsize=32),
                              particle_components=2,
                              component_gridsizes={'particles0': (32, 16), 'particles1': (16, 32)}),
    # ... with interlacing and Fourier-space differentiation on top
    'cic_fcc_multigrid_pow2': dict(method='pm', n=8, gridsize=32, boxsize=64.0, seed=35,
                                   dist='uniform', diff=0, fluid=dict(gridsize=16),
                                   interpolation='CIC', interlace=('bcc', 'fcc'),
                                   component_gridsizes={'particles0': (16, 32)}),
    # the same shape on vertex-centred grids (cell_centered = False: lattice shifts of the
    # other sign, interpolation offsets, 3 ghost layers)
    'cic_fcc_multigrid_vertex_pow2': dict(method='pm', n=8, gridsize=32, boxsize=64.0, seed=40,
                                          dist='uniform', diff=2, fluid=dict(gridsize=16),
                                          interpolation='CIC', interlace=('bcc', 'fcc'),
                                          component_gridsizes={'particles0': (16, 32)},
                                          vertex=True),
    # particles only, one component: upstream 32 -> global 16 -> downstream 24, order 4
    'multigrid_n8_up32_down24': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=25,
                                     dist='uniform', diff=4, fluid=dict(gridsize=8, count=0),
                                     component_gridsizes={'particles0': (32, 24)}),
    # interlacing, TSC / PCS / NGP, Fourier-space differentiation
    'tsc_bcc_n8_g16': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=26, dist='uniform',
                           diff=2, fluid=dict(gridsize=16, count=0),
                           interpolation='TSC', interlace=('bcc', 'bcc')),
    # ... with only the downstream deconvolution (general path: the two are applied apart)
    'tsc_bcc_deconv_down_n8_g16': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=43,
                                       dist='uniform', diff=2, fluid=dict(gridsize=16),
                                       interpolation='TSC', interlace=('bcc', 'bcc'),
                                       extra="potential_options['deconvolve'] = "
                                             "{'gravity': {'pm': (False, True)}}\n"),
    'pcs_fcc_fourier_n8_g16': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=27,
                                   dist='clustered', diff=0, fluid=dict(gridsize=16, count=0),
                                   interpolation='PCS', interlace=('fcc', 'sc')),
    'ngp_fluid_n8_g16': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=28,
                             dist='uniform', diff=4, fluid=dict(gridsize=16),
                             interpolation='NGP', interlace=('sc', 'bcc')),
    'cic_fcc_multigrid_n8': dict(method='pm', n=8, gridsize=16, boxsize=64.0, seed=29,
                                 dist='uniform', diff=0, fluid=dict(gridsize=8),
                                 interpolation='CIC', interlace=('bcc', 'fcc'),
                                 component_gridsizes={'particles0': (24, 12)}),
    # direct summation with the Ewald correction (gravity 'pp':
    # gravity_pairwise gravity.py:121-206, ewald.py) and without (ppnonperiodic); the Ewald
    # table is tabulated by the reference's own summation() on a small grid
    'pp_ewald_n4': dict(method='pp', n=4, gridsize=8, boxsize=20.0, seed=31, dist='uniform',
                        diff=2, pp=True, ewald_gridsize=6),
    'ppnonperiodic_n4': dict(method='ppnonperiodic', n=4, gridsize=8, boxsize=20.0, seed=32,
                             dist='clustered', diff=2, pp=True, ewald_gridsize=6),
    # GADGET-2 snapshots (snapshot.py:640-2640) written by the
    # reference (the binary file is the fixture) and read back by it (expected arrays)
    'gadget_sf2_32': dict(method='pm', n=4, gridsize=8, boxsize=64.0, seed=41, dist='uniform',
                          diff=2, gadget=dict(snapformat=2, bits=32, types=['halo'])),
    'gadget_sf1_64': dict(method='pm', n=4, gridsize=8, boxsize=48.0, seed=42, dist='clustered',
                          diff=2, gadget=dict(snapformat=1, bits=64, types=['disk'])),
    # two components (running identifiers continue over them), POS in double and VEL in single
    # precision, 64-bit identifiers: for the writer (snapshot.save)
    # a snapshot split over several files (gadget_snapshot_params['particles per file']): two
    # components, 100 + 100 particles, at most 70 per file -> three files; for the reader
    'gadget_files3': dict(method='pm', n=5, gridsize=8, boxsize=56.0, seed=44, dist='uniform',
                          diff=2, gadget=dict(snapformat=2, bits=64, id_bits=32, per_file=70,
                                              types=['halo', 'bndry'], n_each=100)),
    'gadget_sf2_multi': dict(method='pm', n=4, gridsize=8, boxsize=40.0, seed=43, dist='uniform',
                             diff=2, gadget=dict(snapformat=2, bits=64, vel_bits=32, id_bits=64,
                                                 types=['halo', 'stars'])),
    'fluid2_pm_n6_g12': dict(method='pm', n=6, gridsize=12, boxsize=48.0, seed=22,
                             dist='clustered', diff=4, fluid=dict(gridsize=12, count=2),
                             particle_components=2),
    # A18, trajectory level: the reference's own main.timeloop() (main.py:102-471) from a_begin
    # to a = 1 on the shapes of test/pure_python_pm/param (8^3 particles, PM mesh 8, box 8 Mpc,
    # a: 0.02 -> 1, dumps at 0.1, 0.5, 1) and test/pure_python_p3m/param (P3M mesh 24, box
    # 4 Mpc, range 3.1 scale, a: 0.1 -> 1, dumps at 0.3, 0.5, 1, adaptive rungs) with the matter
    # + Λ background (enable_class_background = False).  Records every call of
    # get_time_step_integrals (t_start, t_end, all integrals), every base step's (t, a, Δt) and
    # the particles at every dump.
    'traj_pm_n8_g8': dict(method='pm', n=8, gridsize=8, boxsize=8.0, seed=51, traj=dict(
        a_begin=0.02, outputs=(0.1, 0.5, 1))),
    'traj_p3m_n8_g24': dict(method='p3m', n=8, gridsize=24, boxsize=4.0, seed=52, traj=dict(
        a_begin=0.1, outputs=(0.3, 0.5, 1),
        extra="shortrange_params = {'gravity': {'scale': '1.25*boxsize/gridsize', "
              "'range': '3.1*scale', 'subtiling': 2}}\n")),
    # on a power-of-two mesh (the multi-GPU FFT's sizes) for the runs over 2 and 4 domains
    'traj_p3m_n8_g32': dict(method='p3m', n=8, gridsize=32, boxsize=4.0, seed=53, traj=dict(
        a_begin=0.1, outputs=(0.3, 1),
        extra="shortrange_params = {'gravity': {'scale': '1.25*boxsize/gridsize', "
              "'range': '3.1*scale', 'subtiling': 2}}\n")),
    'traj_pm_n8_g16': dict(method='pm', n=8, gridsize=16, boxsize=8.0, seed=54, traj=dict(
        a_begin=0.05, outputs=(0.2, 1))),
    # a larger P3M run with adaptive rungs on a CLUSTERED box: the shape of
    # test/concept_vs_gadget_p3m/param:9-46 — box 8 Mpc, P3M mesh 32, CIC, differentiation
    # order 4, range 4.5 scale, softening 0.03 boxsize/cbrt(N) — with 16^3 particles, 60 % of
    # them in three Gaussian clumps (tiles of several hundred particles, rungs that differ)
    'traj_p3m_n16_g32_clustered': dict(method='p3m', n=16, gridsize=32, boxsize=8.0, seed=55,
                                       traj=dict(
        a_begin=0.1, outputs=(0.11, 0.122), clustered=0.6,
        extra="shortrange_params = {'gravity': {'scale': '1.25*boxsize/gridsize', "
              "'range': '4.5*scale'}}\n"
              "potential_options['differentiation'] = {'matter': {'gravity': {'p3m': 4}}}\n"
              "select_softening_length = {'matter': '0.03*boxsize/cbrt(N)'}\n")),
    # the same P3M run without adaptive rungs (N_rungs = 1)
    'traj_p3m_n8_g24_r1': dict(method='p3m', n=8, gridsize=24, boxsize=4.0, seed=52, traj=dict(
        a_begin=0.1, outputs=(0.3, 0.5, 1),
        extra="shortrange_params = {'gravity': {'scale': '1.25*boxsize/gridsize', "
              "'range': '3.1*scale', 'subtiling': 2}}\nN_rungs = 1\n")),
}


def make_positions(np, cfg):
    rng = np.random.default_rng(cfg['seed'])
    L = cfg['boxsize']
    n = cfg['n']
    dist = cfg['dist']
    if dist == 'uniform':
        pos = rng.uniform(0, L, size=(n**3, 3))
    elif dist == 'lattice':
        g = (np.arange(n) + 0.5)*(L/n)
        pos = np.stack(np.meshgrid(g, g, g, indexing='ij'), axis=-1).reshape(-1, 3)
        pos = pos + rng.normal(0, 1.5*L/cfg['gridsize'], size=pos.shape)
        pos = np.mod(pos, L)
    elif dist == 'perfect_lattice':
        g = (np.arange(n) + 0.5)*(L/n)
        pos = np.stack(np.meshgrid(g, g, g, indexing='ij'), axis=-1).reshape(-1, 3)
    elif dist == 'clustered':
        nc = 6
        centres = rng.uniform(0, L, size=(nc, 3))
        which = rng.integers(0, nc, size=n**3)
        pos = centres[which] + rng.normal(0, 0.03*L, size=(n**3, 3))
        pos = np.mod(pos, L)
    elif dist == 'edge':
        N = cfg['gridsize']
        c = L/N
        specials = [0.0, np.nextafter(L, 0), c, np.nextafter(c, 0), 0.5*c, np.nextafter(0.5*c, 0),
                    np.nextafter(0.5*c, L), L - 0.5*c, np.nextafter(L - 0.5*c, L), L/2,
                    np.nextafter(L/2, 0), 5e-324, 1e-300, 7*c, 7.5*c]
        pts = []
        for x in specials:
            for y in (0.0, np.nextafter(L, 0), 3.3*c):
                for z in (np.nextafter(L, 0), 0.5*c, 9.99*c):
                    pts.append((x, y, z))
                    pts.append((z, x, y))
                    pts.append((y, z, x))
        pos = np.array(pts, dtype=np.float64)
    else:
        raise ValueError(dist)
    pos[pos >= L] = 0.0
    return np.ascontiguousarray(pos, dtype=np.float64)


def param_text(cfg):
    method = cfg['method']
    txt = f"""
boxsize = {cfg['boxsize']!r}*Mpc
potential_options = {{
    'gridsize': {{'gravity': {{'{method}': {cfg['gridsize']}}}}},
    'differentiation': {{'matter': {{'gravity': {{'{method}': {cfg['diff']}}}}}}},
}}
H0 = 70*km/s/Mpc
Ωcdm = 0.25
Ωb = 0.05
a_begin = 0.5
enable_class_background = False
select_forces = {{'matter': {{'gravity': '{method}'}}}}
"""
    if cfg.get('vertex'):
        txt += "cell_centered = False\n"
    if method == 'p3m':
        txt += "shortrange_params = {'gravity': {'subtiling': %r}}\n" % (cfg.get('subtiling', 2),)
        txt += "select_softening_length = {'matter': '0.03*boxsize/cbrt(N)'}\n"
    if 'rungs' in cfg:
        txt += f"N_rungs = {cfg['rungs']}\nenable_Hubble = False\na_begin = 1\n"
        txt += "particle_reordering = False\n"
    txt += cfg.get('extra', '')
    return txt


def child_steps(name):
    """kick / drift sequence of main.timeloop (main.py:255-361) driven with plain
    scalars: init = half long kick (+ half short kick for p3m); each base step =
    drift, (short kick,) long kick.  The state after the init and after every step is
    stored (rows sorted by x: the reference's tile_sort reorders particle memory)."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, 'oracle', 'refharness'))
    from ref_import import load_reference
    cfg = CASES[name]
    ref = load_reference(param_text(cfg), f'/tmp/concept_golden_work/{name}')
    commons, interactions, species = ref.commons, ref.interactions, ref.species
    L = commons.boxsize
    pos = make_positions(np, cfg)
    N = pos.shape[0]
    rng = np.random.default_rng(1000 + cfg['seed'])
    mass = commons.ρ_mbar*L**3/N
    mom = rng.normal(0, 1.0, size=(N, 3))*mass*0.01
    comp = species.Component('matter', 'matter', N=N, mass=mass)
    for d, s_ in enumerate('xyz'):
        comp.populate(np.ascontiguousarray(pos[:, d]), 'pos' + s_)
        comp.populate(np.ascontiguousarray(mom[:, d]), 'mom' + s_)
    method = cfg['method']
    nr = commons.N_rungs
    key2 = ('a**(-3*w_eff₀-3*w_eff₁-1)', 'matter', 'matter')

    def scalars(dt):
        return {'1': dt, 'a**(-2)': dt*1.3, ('a**(-3*w_eff)', 'matter'): dt*1.1,
                ('a**(-3*w_eff-1)', 'matter'): dt*0.9}

    def rung_scalars(dt):
        return {key2: np.full(3*nr - 1, dt*0.8)}

    def state():
        p = np.array(comp.pos_mv3[:N])
        m = np.array(comp.mom_mv3[:N])
        o = np.argsort(p[:, 0], kind='stable')
        return p[o].copy(), m[o].copy()

    def kick_long(dt):
        interactions.gravity(method, [comp], [comp], scalars(dt), 'long-range', False)

    def kick_short(dt):
        comp.nullify_Δ('mom')
        comp.lowest_active_rung = 0
        comp.lowest_populated_rung = 0
        comp.highest_populated_rung = 0
        interactions.gravity(method, [comp], [comp], rung_scalars(dt), 'short-range', False)
        comp.apply_Δmom()

    dt = 0.02
    out = dict(boxsize=L, gridsize=cfg['gridsize'], G_Newton=commons.G_Newton, mass=mass, N=N,
               diff_order=cfg['diff'], dt=dt, N_rungs=nr, method=method,
               softening_length=comp.softening_length,
               pos_in=np.array(comp.pos_mv3[:N]).copy(), mom_in=np.array(comp.mom_mv3[:N]).copy())
    if method == 'p3m':
        out['shortrange_scale'] = commons.shortrange_params['gravity']['scale']
        out['shortrange_range'] = commons.shortrange_params['gravity']['range']
    kick_long(dt/2)
    if method == 'p3m':
        kick_short(dt/2)
    out['pos_init'], out['mom_init'] = state()
    for step in range(cfg['steps']):
        comp.drift(scalars(dt))
        if method == 'p3m':
            kick_short(dt)
        kick_long(dt)
        out[f'pos_step{step + 1}'], out[f'mom_step{step + 1}'] = state()
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, {k: getattr(v, 'shape', v) for k, v in out.items()})


def child_rungs(name):
    """The reference's adaptive-rung machinery driven through its own main.py functions
    (imported with jobid = -1 so that nothing runs at import): initialize_rung_populations
    (main.py:1639), kick_long (:1104), kick_short (:1173), driftkick_short (:1347).  The
    time-step integrals are replaced by t_end - t_start (a = 1, enable_Hubble = False
    semantics) so that they are plain, bit-exact inputs."""
    import importlib
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, 'oracle', 'refharness'))
    from ref_import import load_reference
    cfg = CASES[name]
    ref = load_reference(param_text(cfg), f'/tmp/concept_golden_work/{name}')
    commons, species = ref.commons, ref.species
    commons.jobid = -1
    main = importlib.import_module('main')
    L = commons.boxsize
    nr = commons.N_rungs
    pos = make_positions(np, cfg)
    N = pos.shape[0]
    rng = np.random.default_rng(1000 + cfg['seed'])
    mass = commons.ρ_mbar*L**3/N
    mom = rng.normal(0, 1.0, size=(N, 3))*mass*0.002
    comp = species.Component('matter', 'matter', N=N, mass=mass)
    for d, s_ in enumerate('xyz'):
        comp.populate(np.ascontiguousarray(pos[:, d]), 'pos' + s_)
        comp.populate(np.ascontiguousarray(mom[:, d]), 'mom' + s_)
    assert comp.use_rungs
    calls = []

    def integrals(t_start, t_end, components):
        keys = ['1', 'a**2', 'a**(-1)', 'a**(-2)', 'ȧ/a']
        for c in components:
            for k in ('a**(-3*w_eff)', 'a**(-3*(1+w_eff))', 'a**(-3*w_eff-1)', 'a**(3*w_eff-2)',
                      'a**(-3*w_eff)*Γ/H'):
                keys.append((k, c.name))
        for c0 in components:
            for c1 in components:
                keys.append(('a**(-3*w_eff₀-3*w_eff₁-1)', c0.name, c1.name))
        if not main.ᔑdt_rungs:
            for k in keys:
                main.ᔑdt_rungs[k] = np.zeros(3*nr - 1)
        calls.append((t_start, t_end))
        return {k: (t_end - t_start) for k in keys}
    main.get_time_step_integrals = integrals
    uni = commons.universals
    uni.t = 0.0
    uni.a = 1.0
    sync_time = float('inf')
    dt = cfg.get('dt', 0.6)

    def state():
        p = np.array(comp.pos_mv3[:N])
        m = np.array(comp.mom_mv3[:N])
        r = np.array(comp.rung_indices_mv[:N]).astype(np.int8)
        o = np.argsort(p[:, 0], kind='stable')
        return p[o].copy(), m[o].copy(), r[o].copy()

    out = dict(boxsize=L, gridsize=cfg['gridsize'], G_Newton=commons.G_Newton, mass=mass, N=N,
               diff_order=cfg['diff'], dt=dt, N_rungs=nr, method='p3m',
               softening_length=comp.softening_length, fac_softening=main.fac_softening,
               dt_jump_fac=main.Δt_jump_fac, dt_reltol=main.Δt_reltol,
               shortrange_scale=commons.shortrange_params['gravity']['scale'],
               shortrange_range=commons.shortrange_params['gravity']['range'],
               pos_in=np.array(comp.pos_mv3[:N]).copy(), mom_in=np.array(comp.mom_mv3[:N]).copy())
    main.initialize_rung_populations([comp], dt)
    out['rung_init'] = np.array(comp.rung_indices_mv[:N]).astype(np.int8)  # input order
    out['acc_init'] = np.array(comp.Δmom_mv3[:N]).copy()  # accelerations of the fake kick
    out['rungs_N_init'] = np.array([comp.rungs_N[i] for i in range(nr)], dtype=np.int64)
    main.kick_long([comp], dt, sync_time, 'init')
    main.kick_short([comp], dt)
    out['pos_init'], out['mom_init'], out['rungs_after_init'] = state()
    for step in range(cfg['steps']):
        main.driftkick_short([comp], dt, sync_time)
        uni.t += 0.5*dt
        main.kick_long([comp], dt, sync_time, 'full')
        uni.t += 0.5*dt
        p_, m_, r_ = state()
        out[f'pos_step{step + 1}'], out[f'mom_step{step + 1}'] = p_, m_
        out[f'rungs_step{step + 1}'] = r_
        out[f'rungs_N_step{step + 1}'] = np.array([comp.rungs_N[i] for i in range(nr)],
                                                  dtype=np.int64)
    out['n_integral_calls'] = len(calls)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, 'rungs_N init', out['rungs_N_init'], 'step1', out['rungs_N_step1'],
          'step2', out['rungs_N_step2'], 'integral calls', len(calls))


def fluid_param_text(cfg):
    g = cfg['gridsize']
    if cfg.get('nonlinnu'):
        return f"""
boxsize = {cfg['boxsize']!r}*Mpc
potential_options = {{
    'gridsize': {{'global': {{'gravity': {{'pm': {g}, 'p3m': {2*g}}}}}}},
}}
H0 = 70*km/s/Mpc
Ωcdm = 0.25
Ωb = 0.05
a_begin = 0.5
enable_class_background = False
select_forces = {{'particles': {{'gravity': 'p3m'}}, 'fluid': {{'gravity': 'pm'}}}}
select_boltzmann_closure = {{'all': 'truncate'}}
select_approximations = {{'all': {{'P=wρ': True}}}}
select_softening_length = {{'particles': '0.03*boxsize/cbrt(N)'}}
"""
    own = ''.join(f"'{k}': {{'gravity': {{'pm': {v!r}}}}}, "
                  for k, v in cfg.get('component_gridsizes', {}).items())
    return f"""
boxsize = {cfg['boxsize']!r}*Mpc
potential_options = {{
    'gridsize': {{'global': {{'gravity': {{'pm': {g}}}}}, 'particles': {{'gravity': {{'pm': {g}}}}}, {own}}},
    'differentiation': {{'particles': {{'gravity': {{'pm': {cfg['diff']}}}}},
                        'fluid': {{'gravity': {{'pm': 2}}}}}},
    'interpolation': {{'gravity': {{'pm': {cfg.get('interpolation', 'CIC')!r}}}}},
    'interlace': {{'gravity': {{'pm': {cfg.get('interlace', ('sc', 'sc'))!r}}}}},
}}
H0 = 70*km/s/Mpc
Ωcdm = 0.25
Ωb = 0.05
a_begin = 0.5
enable_class_background = False
select_forces = {{'all': {{'gravity': 'pm'}}}}
select_boltzmann_closure = {{'all': 'truncate'}}
select_approximations = {{'all': {{'P=wρ': True}}}}
""" + ("cell_centered = False\n" if cfg.get('vertex') else '') + cfg.get('extra', '')


def child_fluid(name):
    """gravity('pm', receivers, suppliers, ...) with particle AND fluid components on equal
    grid sizes (interactions.py:1985-2402): inputs, the momenta of every particle component
    and the J grids of every fluid component afterwards, plus the k-space potential."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, 'oracle', 'refharness'))
    from ref_import import load_reference
    cfg = CASES[name]
    ref = load_reference(fluid_param_text(cfg), f'/tmp/concept_golden_work/{name}')
    commons, interactions, species = ref.commons, ref.interactions, ref.species
    L = commons.boxsize
    rng = np.random.default_rng(1000 + cfg['seed'])
    out = dict(boxsize=L, gridsize=cfg['gridsize'], nghosts=commons.nghosts,
               cell_centered=int(commons.cell_centered),
               deconvolve=np.array(commons.potential_options['deconvolve']['gravity']['pm'],
                                   dtype=np.int64),
               G_Newton=commons.G_Newton, light_speed=commons.light_speed, diff_order=cfg['diff'])
    comps = []
    npc = cfg.get('particle_components', 1)
    pos_all = make_positions(np, cfg)
    per = pos_all.shape[0]//npc
    dt = 0.017
    sdt = {'1': dt}
    for c in range(npc):
        pos = pos_all[c*per:(c + 1)*per]
        N = pos.shape[0]
        mass = commons.ρ_mbar*L**3/pos_all.shape[0]*(1.0 + 0.5*c)
        mom = rng.normal(0, 1.0, size=(N, 3))*mass*0.01
        nm = f'particles{c}'
        comp = species.Component(nm, 'matter', N=N, mass=mass)
        for d, s_ in enumerate('xyz'):
            comp.populate(np.ascontiguousarray(pos[:, d]), 'pos' + s_)
            comp.populate(np.ascontiguousarray(mom[:, d]), 'mom' + s_)
        comps.append(comp)
        sdt['a**(-3*w_eff)', nm] = dt*(1.1 + 0.1*c)
        sdt['a**(-3*w_eff-1)', nm] = dt*(1.9 + 0.2*c)
        out[f'p{c}_pos'] = np.array(comp.pos_mv3[:N]).copy()
        out[f'p{c}_mom_in'] = np.array(comp.mom_mv3[:N]).copy()
        out[f'p{c}_mass'] = mass
        out[f'p{c}_dt_kick'] = sdt['a**(-3*w_eff)', nm]
        out[f'p{c}_dt_dens'] = sdt['a**(-3*w_eff-1)', nm]
    gs = cfg['fluid']['gridsize']
    nfl = cfg['fluid'].get('count', 1)
    for c in range(nfl):
        nm = f'fluid{c}'
        fl = species.Component(nm, 'matter', gridsize=gs, boltzmann_order=1)
        rho = commons.ρ_mbar*(0.3 + 0.1*c)*(1 + 0.2*rng.normal(size=(gs, gs, gs)))
        fl.populate(np.ascontiguousarray(rho), 'ϱ')
        J = [0.01*commons.ρ_mbar*rng.normal(size=(gs, gs, gs)) for _ in range(3)]
        for d in range(3):
            fl.populate(np.ascontiguousarray(J[d]), 'J', d)
        comps.append(fl)
        sdt['a**(-3*w_eff)', nm] = dt*(0.7 + 0.1*c)
        sdt['a**(-3*w_eff-1)', nm] = dt*(1.3 + 0.2*c)
        out[f'f{c}_rho'] = np.array(fl.ϱ.grid_noghosts[:gs, :gs, :gs]).copy()
        out[f'f{c}_P'] = np.array(fl.𝒫.grid_noghosts[:gs, :gs, :gs]).copy()
        out[f'f{c}_J_in'] = np.stack([np.array(fl.J[d].grid_noghosts[:gs, :gs, :gs]).copy()
                                      for d in range(3)])
        out[f'f{c}_dt_kick'] = sdt['a**(-3*w_eff)', nm]
        out[f'f{c}_dt_dens'] = sdt['a**(-3*w_eff-1)', nm]
        out[f'f{c}_w_eff'] = fl.w_eff(a=commons.universals.a)
    out['dt_1'] = dt
    out['interpolation'] = cfg.get('interpolation', 'CIC')
    out['interlace'] = np.array(cfg.get('interlace', ('sc', 'sc')))
    out['n_particle_components'] = npc
    out['n_fluid_components'] = nfl
    inter = interactions.find_interactions(comps, 'long-range')
    out['n_interactions'] = len(inter)
    out['interactions'] = np.array(
        [f'{f}|{m}|' + ','.join(r.name for r in rec) + '|' + ','.join(u.name for u in sup)
         for f, m, rec, sup in inter])
    for comp in comps:
        for force, dd in comp.potential_gridsizes.items():
            for m, gsz in dd.items():
                out[f'gridsizes_{comp.name}_{m}'] = np.array([gsz.upstream, gsz.downstream])
        for force, dd in comp.potential_differentiations.items():
            for m, od in dd.items():
                out[f'differentiation_{comp.name}_{m}'] = od
    if cfg.get('nonlinnu'):
        out['shortrange_scale'] = commons.shortrange_params['gravity']['scale']
        out['shortrange_range'] = commons.shortrange_params['gravity']['range']
    # capture the k-space potential: fft(slab, 'backward') is called once per downstream
    # subgroup; the first call's input is the (possibly deconvolved) downstream potential
    cap = {}
    orig_fft = interactions.fft

    def fft_hook(slab, direction, *a, **k):
        if direction == 'backward':
            cap.setdefault('slab_k', []).append(np.array(slab).copy())
        return orig_fft(slab, direction, *a, **k)
    interactions.fft = fft_hook
    for force, method, receivers, suppliers in inter:
        getattr(interactions, force)(method, receivers, suppliers, sdt, 'long-range', False)
    interactions.fft = orig_fft
    for i, sk in enumerate(cap.get('slab_k', [])):
        out[f'slab_k_before_backward_{i}'] = sk
    ci = fi = 0
    for comp in comps:
        if comp.representation == 'particles':
            out[f'p{ci}_mom_out'] = np.array(comp.mom_mv3[:comp.N]).copy()
            ci += 1
        else:
            out[f'f{fi}_J_out'] = np.stack([np.array(comp.J[d].grid_noghosts[:gs, :gs, :gs]).copy()
                                           for d in range(3)])
            out[f'f{fi}_rho_out'] = np.array(comp.ϱ.grid_noghosts[:gs, :gs, :gs]).copy()
            fi += 1
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, {k: getattr(v, 'shape', v) for k, v in out.items()})


def child_gadget(name):
    """snapshot.save(..., snapshot_type 'gadget') then snapshot.load() of the same file:
    tests/golden/<name>.gadget is what the reference wrote, the .npz what it reads back
    (positions, momenta, masses, parameters) — the reader under test must agree."""
    import importlib
    import shutil
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, 'oracle', 'refharness'))
    from ref_import import load_reference
    cfg = CASES[name]
    gd = cfg['gadget']
    text = f"""
boxsize = {cfg['boxsize']!r}*Mpc
H0 = 70*km/s/Mpc
Ωcdm = 0.25
Ωb = 0.05
a_begin = 0.5
enable_class_background = False
select_forces = {{'all': {{'gravity': 'pm'}}}}
snapshot_type = 'gadget'
gadget_snapshot_params = {{'snapformat': {gd['snapformat']},
                          'dataformat': {{'POS': {gd['bits']}, 'VEL': {gd.get('vel_bits', gd['bits'])},
                                         'ID': {gd.get('id_bits', 'automatic')!r}}},
                          'particles per file': {gd.get('per_file', 'automatic')!r}}}
"""
    work = f'/tmp/concept_golden_work/{name}'
    ref = load_reference(text, work)
    commons, species = ref.commons, ref.species
    snapshot = importlib.import_module('snapshot')
    L = commons.boxsize
    rng = np.random.default_rng(1000 + cfg['seed'])
    pos_all = make_positions(np, cfg)
    if gd.get('n_each'):
        pos_all = np.random.default_rng(cfg['seed']).uniform(0, L, (gd['n_each']*len(gd['types']), 3))
    per = pos_all.shape[0]//len(gd['types'])
    comps = []
    for i, typ in enumerate(gd['types']):
        pos = pos_all[i*per:(i + 1)*per]
        N = pos.shape[0]
        mass = commons.ρ_mbar*L**3/pos_all.shape[0]*(1 + i)
        mom = rng.normal(0, 1, size=(N, 3))*mass
        comp = species.Component(f'GADGET {typ}', 'matter', N=N, mass=mass)
        for d, s_ in enumerate('xyz'):
            comp.populate(np.ascontiguousarray(pos[:, d]), 'pos' + s_)
            comp.populate(np.ascontiguousarray(mom[:, d]), 'mom' + s_)
        comps.append(comp)
    originals = [(c.name, int(c.N), float(c.mass), np.array(c.pos_mv3[:c.N]).copy(),
                  np.array(c.mom_mv3[:c.N]).copy()) for c in comps]
    fn = snapshot.save(comps, f'{work}/out/snap', save_all=True)
    if os.path.isdir(fn):   # several files: the directory as it is
        dst = os.path.join(HERE, name + '.gadget')
        shutil.rmtree(dst, ignore_errors=True)
        shutil.copytree(fn, dst)
    else:
        shutil.copyfile(fn, os.path.join(HERE, name + '.gadget'))
    # (the reference's GADGET *loader* does not fill the particle arrays in pure-Python
    # mode, so the expected values are what its writer was given: the reader under test
    # must invert the writer's unit conversions, snapshot.py:1520-1553)
    probe = snapshot.GadgetSnapshot()
    probe.populate(comps, {})
    out = dict(snapformat=gd['snapformat'], bits=gd['bits'], vel_bits=gd.get('vel_bits', gd['bits']),
               id_bits=gd.get('id_bits', 0), boxsize=L,
               a=commons.universals.a, H0=commons.H0, n_components=len(comps),
               unit_length=probe.unit_length, unit_velocity=probe.unit_velocity,
               unit_mass=probe.unit_mass, h=probe.h,
               names=np.array([o[0] for o in originals]))
    for i, (nm, n_c, mass, pos, mom) in enumerate(originals):
        out[f'c{i}_N'] = n_c
        out[f'c{i}_mass'] = mass
        out[f'c{i}_pos'] = pos
        out[f'c{i}_mom'] = mom
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, (sorted(os.listdir(os.path.join(HERE, name + '.gadget')))
                         if os.path.isdir(os.path.join(HERE, name + '.gadget'))
                         else os.path.getsize(os.path.join(HERE, name + '.gadget'))), 'bytes;',
          {k: getattr(v, 'shape', v) for k, v in out.items()})


def child_pp(name):
    """gravity('pp' | 'ppnonperiodic', [c], [c], ᔑdt_rungs, 'any', False): Δmom of every
    particle with all particles on rung 0, the Ewald grid the reference tabulated
    (ewald.tabulate -> summation, ewald.py:62-118) and sample look-ups ewald(x, y, z)."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, 'oracle', 'refharness'))
    from ref_import import load_reference
    cfg = CASES[name]
    method = cfg['method']
    text = f"""
boxsize = {cfg['boxsize']!r}*Mpc
H0 = 70*km/s/Mpc
Ωcdm = 0.25
Ωb = 0.05
a_begin = 0.5
enable_class_background = False
select_forces = {{'matter': {{'gravity': '{method}'}}}}
select_softening_length = {{'matter': '0.03*boxsize/cbrt(N)'}}
ewald_gridsize = {cfg['ewald_gridsize']}
"""
    ref = load_reference(text, f'/tmp/concept_golden_work/{name}')
    commons, interactions, species = ref.commons, ref.interactions, ref.species
    import importlib
    ewald = importlib.import_module('ewald')
    L = commons.boxsize
    pos = make_positions(np, cfg)
    N = pos.shape[0]
    mass = commons.ρ_mbar*L**3/N
    comp = species.Component('matter', 'matter', N=N, mass=mass)
    for d, s_ in enumerate('xyz'):
        comp.populate(np.ascontiguousarray(pos[:, d]), 'pos' + s_)
        comp.populate(np.zeros(N), 'mom' + s_)
    nr = commons.N_rungs
    key2 = ('a**(-3*w_eff₀-3*w_eff₁-1)', 'matter', 'matter')
    sdt_rungs = {key2: 0.02*2.3*(1 + 0.1*np.arange(3*nr - 1))}
    out = dict(boxsize=L, N=N, mass=mass, G_Newton=commons.G_Newton, method=method,
               softening_length=comp.softening_length, N_rungs=nr,
               ewald_gridsize=commons.ewald_gridsize, dt_rungs_pair=sdt_rungs[key2].copy(),
               pos_in=np.array(comp.pos_mv3[:N]).copy())
    if method == 'pp':
        # ewald.tabulate() with filename '' (h5py is absent: the grid is not cached on disk)
        ewald.grid = ref.mesh.tabulate_vectorgrid(
            commons.ewald_gridsize, ewald.summation, 0.5/(commons.ewald_gridsize - 1), '')
    comp.nullify_Δ('mom')
    comp.lowest_active_rung = 0
    comp.lowest_populated_rung = 0
    comp.highest_populated_rung = 0
    interactions.gravity(method, [comp], [comp], sdt_rungs, 'any', False)
    out['dmom'] = np.array(comp.Δmom_mv3[:N]).copy()
    out['pos_after'] = np.array(comp.pos_mv3[:N]).copy()
    if method == 'pp':
        out['ewald_grid'] = np.array(ewald.grid).copy()
        rng = np.random.default_rng(7)
        pts = rng.uniform(-0.5*L, 0.5*L, size=(40, 3))
        pts[0] = (0.0, 0.1*L, -0.2*L)
        pts[1] = (0.25*L, 0.0, 0.0)
        out['ewald_points'] = pts
        out['ewald_values'] = np.array([[ewald.ewald(*p)[d] for d in range(3)] for p in pts])
        out['ewald_constants'] = np.array([ewald.rs, ewald.maxdist, ewald.maxh2, ewald.h_lower,
                                           ewald.h_upper, ewald.n_lower, ewald.n_upper], float)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, {k: getattr(v, 'shape', v) for k, v in out.items()})


def traj_param_text(cfg):
    tr = cfg['traj']
    method = cfg['method']
    return f"""
output_dirs = {{'snapshot': f'{{param.dir}}/output'}}
output_bases = {{'snapshot': 'snapshot'}}
output_times = {{'snapshot': {tuple(tr['outputs'])!r}}}
boxsize = {cfg['boxsize']!r}*Mpc
potential_options = {{'gridsize': {{'gravity': {{'{method}': {cfg['gridsize']}}}}}}}
H0 = 70*km/s/Mpc
Ωcdm = 0.25
Ωb = 0.05
a_begin = {tr['a_begin']!r}
enable_class_background = False
select_forces = {{'matter': {{'gravity': '{method}'}}}}
particle_reordering = False
print_load_imbalance = False
""" + tr.get('extra', '')


def child_traj(name):
    """The reference's main.timeloop() itself (imported with jobid = -1 so that nothing runs at
    import), fed in-memory initial conditions through main.get_initial_conditions, its dumps
    captured through main.dump, every get_time_step_integrals call recorded."""
    import importlib
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, 'oracle', 'refharness'))
    from ref_import import load_reference
    cfg = CASES[name]
    text = traj_param_text(cfg)
    work = f'/tmp/concept_golden_work/{name}'
    ref = load_reference(text, work)
    commons, species, integration = ref.commons, ref.species, ref.integration
    commons.jobid = -1
    main = importlib.import_module('main')
    L = commons.boxsize
    n = cfg['n']
    N = n**3
    rng = np.random.default_rng(1000 + cfg['seed'])
    # a displaced lattice moving in the growing mode: mom = m a^2 H f ψ with f = 1
    lat = (np.stack(np.meshgrid(*[np.arange(n)]*3, indexing='ij'), -1).reshape(-1, 3) + 0.5)*(L/n)
    psi = rng.normal(0, 0.08*L/n, (N, 3))
    pos = np.ascontiguousarray((lat + psi) % L)
    if cfg['traj'].get('clustered'):
        # that fraction of the particles in three Gaussian clumps of sigma = L/25
        k = int(cfg['traj']['clustered']*N)
        which = rng.permutation(N)[:k]
        centres = rng.uniform(0, L, (3, 3))
        pos[which] = (centres[rng.integers(0, 3, k)] + rng.normal(0, L/25, (k, 3))) % L
    pos[pos >= L] = 0.0
    mass = commons.ρ_mbar*L**3/N
    integration.init_time()
    uni = commons.universals
    mom = np.ascontiguousarray(psi*mass*uni.a**2*integration.hubble(uni.a))
    comp = species.Component('matter', 'matter', N=N, mass=mass)
    for d, s_ in enumerate('xyz'):
        comp.populate(np.ascontiguousarray(pos[:, d]), 'pos' + s_)
        comp.populate(np.ascontiguousarray(mom[:, d]), 'mom' + s_)
    ids = np.arange(N)
    out = dict(param_text=text, boxsize=L, gridsize=cfg['gridsize'], N=N, mass=mass,
               method=cfg['method'], G_Newton=commons.G_Newton, H0=commons.H0, Ωm=commons.Ωm,
               N_rungs=commons.N_rungs, softening_length=comp.softening_length,
               a_begin=uni.a, t_begin=uni.t, pos_in=pos.copy(), mom_in=mom.copy(),
               bg_a=np.array(integration.temporal_splines.a_t.x)[::25].copy(),
               bg_t=np.array(integration.temporal_splines.a_t.y)[::25].copy())
    main.get_initial_conditions = lambda *a, **k: [comp]
    main.check_autosave = lambda: (0, 0.0, 0.0, {})
    main.autosave_subdir = work + '/no_autosave'
    dumps = []

    def dump(components, output_filenames, dump_time, Δt=0):
        c = components[0]
        # particle memory may have been reordered (tile_sort): rows are identified by the
        # lattice site nearest to where each particle started — instead, keep the order
        # fixed with particle_reordering = False and check it
        dumps.append((uni.a, uni.t, np.array(c.pos_mv3[:N]).copy(), np.array(c.mom_mv3[:N]).copy()))
        return False
    main.dump = dump
    calls = []
    orig_integrals = main.get_time_step_integrals

    def integrals(t_start, t_end, components):
        res = orig_integrals(t_start, t_end, components)
        if t_start != t_end:
            calls.append((float(t_start), float(t_end), {k: float(v) for k, v in res.items()}))
            contexts.append(context[0])
        return res
    main.get_time_step_integrals = integrals
    # who asked: 'L0' / 'L1' kick_long init / full, 'S' kick_short, 'D' driftkick_short (rung
    # kicks), 'Dd' the drift inside driftkick_short
    contexts, context = [], ['']

    def tagged(tag, func):
        def wrapper(*a, **k):
            context[0] = tag(*a, **k) if callable(tag) else tag
            try:
                return func(*a, **k)
            finally:
                context[0] = ''
        return wrapper
    main.kick_short = tagged('S', main.kick_short)
    main.driftkick_short = tagged('D', main.driftkick_short)
    orig_drift = species.Component.drift

    def drift(self, *a, **k):
        contexts[-1] = 'Dd'   # the call just made was for this drift
        return orig_drift(self, *a, **k)
    species.Component.drift = drift
    steps = []
    orig_heading = main.print_timestep_heading

    def heading(time_step, Δt, bottleneck, components, end=False):
        steps.append((int(time_step), float(uni.t), float(uni.a), float(Δt), str(bottleneck),
                      int(end)))
        return orig_heading(time_step, Δt, bottleneck, components, end)
    main.print_timestep_heading = heading
    kicks = []
    orig_kick_long = main.kick_long

    def kick_long(components, Δt, sync_time, step_type):
        kicks.append((float(uni.t), float(Δt), float(sync_time), step_type == 'full'))
        context[0] = 'L1' if step_type == 'full' else 'L0'
        try:
            return orig_kick_long(components, Δt, sync_time, step_type)
        finally:
            context[0] = ''
    main.kick_long = kick_long
    main.timeloop()
    keys = list(calls[0][2])
    out['integral_keys'] = np.array(['|'.join(k) if isinstance(k, tuple) else k for k in keys])
    out['integral_t'] = np.array([(c[0], c[1]) for c in calls])
    out['integral_values'] = np.array([[c[2][k] for k in keys] for c in calls])
    out['integral_context'] = np.array(contexts)
    out['step_number'] = np.array([s_[0] for s_ in steps])
    out['step_t'] = np.array([s_[1] for s_ in steps])
    out['step_a'] = np.array([s_[2] for s_ in steps])
    out['step_dt'] = np.array([s_[3] for s_ in steps])
    out['step_bottleneck'] = np.array([s_[4] for s_ in steps])
    out['kick_t'] = np.array([k[0] for k in kicks])
    out['kick_dt'] = np.array([k[1] for k in kicks])
    out['kick_sync'] = np.array([k[2] for k in kicks])
    out['kick_full'] = np.array([k[3] for k in kicks])
    out['dump_a'] = np.array([d[0] for d in dumps])
    out['dump_t'] = np.array([d[1] for d in dumps])
    out['dump_pos'] = np.array([d[2] for d in dumps])
    out['dump_mom'] = np.array([d[3] for d in dumps])
    if cfg['method'] == 'p3m':
        out['shortrange_scale'] = commons.shortrange_params['gravity']['scale']
        out['shortrange_range'] = commons.shortrange_params['gravity']['range']
        out['shortrange_tilesize'] = commons.shortrange_params['gravity']['tilesize']
        out['shortrange_tablesize'] = commons.shortrange_params['gravity']['tablesize']
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, 'steps', len(steps), 'integral calls', len(calls), 'dumps at a =',
          out['dump_a'])


def child(name):
    import numpy as np
    if 'traj' in CASES[name]:
        return child_traj(name)
    if CASES[name].get('gadget'):
        return child_gadget(name)
    if CASES[name].get('pp'):
        return child_pp(name)
    if 'fluid' in CASES[name]:
        return child_fluid(name)
    if 'rungs' in CASES[name]:
        return child_rungs(name)
    if 'steps' in CASES[name]:
        return child_steps(name)
    sys.path.insert(0, os.path.join(REPO, 'oracle', 'refharness'))
    from ref_import import load_reference
    cfg = CASES[name]
    ref = load_reference(param_text(cfg), f'/tmp/concept_golden_work/{name}')
    commons, mesh, interactions, species = ref.commons, ref.mesh, ref.interactions, ref.species
    L = commons.boxsize
    assert L == cfg['boxsize']
    pos = make_positions(np, cfg)
    N = pos.shape[0]
    rng = np.random.default_rng(1000 + cfg['seed'])
    mass = commons.ρ_mbar*L**3/N
    mom = rng.normal(0, 1.0, size=(N, 3))*mass*0.01
    comp = species.Component('matter', 'matter', N=N, mass=mass)
    for d, s in enumerate('xyz'):
        comp.populate(np.ascontiguousarray(pos[:, d]), 'pos' + s)
        comp.populate(np.ascontiguousarray(mom[:, d]), 'mom' + s)
    out = dict(
        boxsize=L, gridsize=cfg['gridsize'], nghosts=commons.nghosts, G_Newton=commons.G_Newton,
        mass=mass, N=N, diff_order=cfg['diff'], cell_centered=int(commons.cell_centered),
        softening_kernel=str(commons.softening_kernel),
        deconvolve=np.array(commons.potential_options['deconvolve']['gravity'][cfg['method']],
                            dtype=np.int64),
        softening_length=comp.softening_length,
        pos_in=np.array(comp.pos_mv3[:N]).copy(), mom_in=np.array(comp.mom_mv3[:N]).copy(),
    )
    # Time-step integrals: plain, distinct numbers so a mixed-up key shows
    dt = 0.013
    sdt = {
        '1': dt,
        'a**(-2)': dt*3.7,
        ('a**(-3*w_eff)', 'matter'): dt*1.1,
        ('a**(-3*w_eff-1)', 'matter'): dt*1.9,
    }
    out.update(dt_1=sdt['1'], dt_am2=sdt['a**(-2)'], dt_kick=sdt['a**(-3*w_eff)', 'matter'],
               dt_dens=sdt['a**(-3*w_eff-1)', 'matter'])
    # ---- capture hooks (module-global rebinding) ----
    cap = {}
    idx_log = []
    orig_set_weights = mesh.set_weights_CIC

    def set_weights_CIC(x, weights):
        index = orig_set_weights(x, weights)
        idx_log.append(index)
        return index
    mesh.set_weights_CIC = set_weights_CIC

    orig_interp_particles = mesh.interpolate_particles

    def interpolate_particles(component, gridsize, grid, *a, **k):
        orig_interp_particles(component, gridsize, grid, *a, **k)
        cap['grid_deposit'] = np.array(grid).copy()
        cap['cic_index_deposit'] = np.array(idx_log, dtype=np.int64).reshape(-1, 3)
        idx_log.clear()
    mesh.interpolate_particles = interpolate_particles

    orig_upstream = interactions.interpolate_upstream

    def interpolate_upstream(*a, **k):
        slab = orig_upstream(*a, **k)
        cap['slab_density_k'] = np.array(slab).copy()
        return slab
    interactions.interpolate_upstream = interpolate_upstream

    orig_fft = interactions.fft

    def fft(slab, direction, *a, **k):
        if direction == 'backward':
            cap['slab_potential_k'] = np.array(slab).copy()
        return orig_fft(slab, direction, *a, **k)
    interactions.fft = fft

    orig_dd = interactions.domain_decompose

    def domain_decompose(*a, **k):
        grid = orig_dd(*a, **k)
        cap['grid_potential'] = np.array(grid).copy()
        return grid
    interactions.domain_decompose = domain_decompose

    orig_diff = interactions.diff_domaingrid
    forces = []

    def diff_domaingrid(*a, **k):
        g = orig_diff(*a, **k)
        forces.append(np.array(g).copy())
        return g
    interactions.diff_domaingrid = diff_domaingrid

    orig_apply = interactions.apply_particle_mesh_force
    gather_idx = []

    def apply_particle_mesh_force(*a, **k):
        idx_log.clear()
        orig_apply(*a, **k)
        gather_idx.append(np.array(idx_log, dtype=np.int64).reshape(-1, 3))
        idx_log.clear()
    interactions.apply_particle_mesh_force = apply_particle_mesh_force

    method = cfg['method']
    interactions.gravity(method, [comp], [comp], sdt, 'long-range', False)
    out['mom_after_long'] = np.array(comp.mom_mv3[:N]).copy()
    out['cic_index_deposit'] = cap['cic_index_deposit']
    out['cic_index_gather'] = gather_idx[0]
    assert all(np.array_equal(gather_idx[0], g) for g in gather_idx)
    if cfg['full']:
        out['grid_deposit'] = cap['grid_deposit']
        out['slab_density_k'] = cap['slab_density_k']
        out['slab_potential_k'] = cap['slab_potential_k']
        out['grid_potential'] = cap['grid_potential']
        out['grid_force'] = np.stack(forces)
    else:
        # checksums + strided sample of the big grids
        for key in ('grid_deposit', 'slab_density_k', 'slab_potential_k', 'grid_potential'):
            a = cap[key]
            out[key + '_sum'] = a.sum()
            out[key + '_abssum'] = np.abs(a).sum()
            out[key + '_sample'] = a.ravel()[::97].copy()
    mesh.set_weights_CIC = orig_set_weights
    if method == 'p3m':
        out['shortrange_scale'] = commons.shortrange_params['gravity']['scale']
        out['shortrange_range'] = commons.shortrange_params['gravity']['range']
        out['shortrange_tilesize'] = commons.shortrange_params['gravity']['tilesize']
        out['shortrange_tablesize'] = commons.shortrange_params['gravity']['tablesize']
        out['N_rungs'] = commons.N_rungs
        # Short-range kick: all particles on rung 0, Δmom zeroed
        # (what main.kick_short() sets up, main.py:1173-1262)
        nr = commons.N_rungs
        key2 = ('a**(-3*w_eff₀-3*w_eff₁-1)', 'matter', 'matter')
        sdt_rungs = {key2: np.zeros(3*nr - 1)}
        sdt_rungs[key2][:] = dt*2.3*(1 + 0.1*np.arange(3*nr - 1))
        out['dt_rungs_pair'] = sdt_rungs[key2].copy()
        comp.nullify_Δ('mom')
        comp.lowest_active_rung = 0
        comp.lowest_populated_rung = 0
        comp.highest_populated_rung = 0
        interactions.gravity(method, [comp], [comp], sdt_rungs, 'short-range', False)
        out['dmom_short'] = np.array(comp.Δmom_mv3[:N]).copy()
        out['pos_after_short'] = np.array(comp.pos_mv3[:N]).copy()  # tile_sort may reorder
        out['mom_after_short_sorted'] = np.array(comp.mom_mv3[:N]).copy()
        gravity_mod = ref.gravity
        (table,) = gravity_mod.shortrange_tables.values()  # one softening -> one cached table
        out['shortrange_table'] = np.array(table).copy()
        out['shortrange_table_maxr2'] = gravity_mod.shortrange_table_maxr2
        tiling = comp.tilings['gravity (tiles)']
        out['tiling_shape'] = np.array(tiling.shape, dtype=np.int64)
        sub = comp.tilings['gravity (subtiles)']
        out['subtiling_shape'] = np.array(sub.shape, dtype=np.int64)
    # Drift (species.py:2179-2199); universals.a is a_begin
    pos_before = np.array(comp.pos_mv3[:N]).copy()
    mom_before = np.array(comp.mom_mv3[:N]).copy()
    comp.drift(sdt)
    if method == 'p3m':  # tile_sort reordered the particles
        out['drift_pos_in'] = pos_before
        out['drift_mom_in'] = mom_before
    else:  # drift input is (pos_in, mom_after_long)
        assert np.array_equal(pos_before, out['pos_in'])
        assert np.array_equal(mom_before, out['mom_after_long'])
    out['drift_pos_out'] = np.array(comp.pos_mv3[:N]).copy()
    out['drift_dt_over_mass'] = sdt['a**(-2)']*commons.universals.a**(
        3*comp.w_eff(a=commons.universals.a))/comp.mass
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, {k: getattr(v, 'shape', v) for k, v in out.items()})


def main():
    if len(sys.argv) > 1 and sys.argv[1] in CASES:
        child(sys.argv[1])
        return
    names = list(CASES)
    for name in names:
        print('===', name, flush=True)
        log = f'/tmp/concept_golden_{name}.log'
        with open(log, 'w') as f:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), name], stdout=f,
                               stderr=subprocess.STDOUT)
        tail = open(log).read().splitlines()[-3:]
        print('\n'.join(tail))
        if r.returncode:
            sys.exit(f'case {name} failed, see {log}')


if __name__ == '__main__':
    main()
