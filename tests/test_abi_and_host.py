"""CPU-side checks: the C-ABI library loads here (no GPU) and exports every
symbol include/concept_gpu.h declares; host-side parameter logic mirrors the
reference's names and defaults.  No compute call is made."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, 'include', 'concept_gpu.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cg_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_exported_and_bound():
    from concept_amd import build
    if not os.path.exists(build.LIB):
        build.build(verbose=False)
    lib = ctypes.CDLL(build.LIB)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f'{n} declared in concept_gpu.h but not exported'
    from concept_amd import lib as binding
    assert sorted(binding.SYMBOLS) == names, 'ctypes binding out of sync with the header'
    assert binding.raw().cg_abi_version() == 2


def test_no_product_module_touches_the_oracle():
    """The oracle is test infrastructure: nothing under concept_amd/ may import it."""
    for root, _, files in os.walk(os.path.join(REPO, 'concept_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(root, f), encoding='utf-8').read()
                assert 'oracle' not in text.lower(), f'{f} mentions the oracle'


def test_params_names_and_defaults():
    from concept_amd import commons
    p = commons.load_params("""
boxsize = 64*Mpc
potential_options = {'gridsize': {'gravity': {'p3m': 32}},
                     'differentiation': {'matter': {'gravity': {'p3m': 4}}}}
select_forces = {'matter': {'gravity': 'p3m'}}
select_softening_length = {'matter': '0.03*boxsize/cbrt(N)'}
""")
    assert p.boxsize == 64.0
    assert p.G_Newton == 4.4985024439973154e-05  # reference value in (Mpc, Gyr, 1e10 m_sun)
    assert p.nghosts == 2 and p.cell_centered and p.N_rungs == 8
    assert p.potential_options['interpolation']['gravity']['p3m'] == 2
    assert p.potential_options['deconvolve']['gravity']['pm'] == (True, True)
    sr = commons.resolve_shortrange(p, 32)
    assert sr['scale'] == 1.25*64/32 and sr['range'] == 4.5*sr['scale'] and sr['tablesize'] == 4096
    assert abs(commons.softening_length(p, 'matter', 512) - 0.03*64/8) < 1e-15
    # default differentiation orders: pm 2, p3m 4 (commons.py:3209-3237)
    assert p.potential_options['differentiation']['default']['gravity'] == {'pm': 2, 'p3m': 4}


def test_deposit_scalar_matches_oracle_expression():
    from oracle import oracle
    # the host code in interactions.particle_mesh and the oracle evaluate the same expression
    mass, dt_dens, dt_1, N, L = 3.7, 0.0247, 0.013, 32, 100.0
    contribution = dt_dens/dt_1
    contribution *= mass
    contribution *= float(N)**(-3)*(N/L)**3
    assert contribution == oracle.deposit_contribution(mass, dt_dens, dt_1, N, L)


class _Fake:
    """what find_interactions() reads of a component"""

    def __init__(self, name, representation, forces):
        self.name, self.representation, self.forces = name, representation, forces

    def __repr__(self):
        return self.name


def _describe(found):
    return [f'{f}|{m}|' + ','.join(r.name for r in rec) + '|' + ','.join(s.name for s in sup)
            for f, m, rec, sup in found]


def test_find_interactions_matches_reference(golden):
    """interactions.py:2456-2636 — against the lists the reference itself produced
    (stored in the fluid goldens) and its documented splitting / merging rules."""
    from concept_amd import interactions
    g = golden('nonlinnu_like_n8')
    part = _Fake('particles0', 'particles', {'gravity': 'p3m'})
    fluid = _Fake('fluid0', 'fluid', {'gravity': 'pm'})
    assert _describe(interactions.find_interactions([part, fluid], 'long-range')) == \
        list(g['interactions'])
    # short-range: only the P³M interaction survives, with the particle supplier
    assert _describe(interactions.find_interactions([part, fluid], 'short-range')) == \
        ['gravity|p3m|particles0|particles0']
    g = golden('fluid2_pm_n6_g12')
    comps = [_Fake(f'particles{i}', 'particles', {'gravity': 'pm'}) for i in range(2)]
    comps += [_Fake(f'fluid{i}', 'fluid', {'gravity': 'pm'}) for i in range(2)]
    assert _describe(interactions.find_interactions(comps, 'long-range')) == \
        list(g['interactions'])
    assert interactions.find_interactions(comps, 'short-range') == []
    # two particle species on different methods supply each other
    a = _Fake('a', 'particles', {'gravity': 'p3m'})
    b = _Fake('b', 'particles', {'gravity': 'pm'})
    assert _describe(interactions.find_interactions([a, b])) == \
        ['gravity|p3m|a|a,b', 'gravity|pm|b|a,b']
    from concept_amd.lib import ConceptGPUError
    with pytest.raises(ConceptGPUError):
        interactions.find_interactions([_Fake('c', 'particles', {'lapse': 'pm'})])
    with pytest.raises(ConceptGPUError):
        interactions.find_interactions([a], 'sideways')


def test_group_components_ordering():
    from concept_amd import interactions
    cs = [_Fake('p', 'particles', {}), _Fake('f', 'fluid', {}), _Fake('q', 'particles', {})]
    groups = interactions.group_components(cs, [32, 16, 16], [16, ...])
    assert list(groups) == [16, 32]
    assert [c.name for c in groups[16]['fluid']] == ['f']
    assert [c.name for c in groups[16]['particles']] == ['q']
    groups = interactions.group_components(cs, [32, 16, 64], [..., 32])
    assert list(groups) == [16, 64, 32]
    flat = interactions.group_components(cs, [4, 2, 4], [4, 2], split_representations=False)
    assert [c.name for c in flat[4]] == ['p', 'q'] and list(flat) == [4, 2]


def test_is_selected_precedence():
    from concept_amd import commons
    c = _Fake('Nu One', 'fluid', {})
    c.species = 'neutrino'
    d = {'default': 1, 'all': 2, 'fluid': 3, 'neutrino': 4, 'nu one': 5}
    assert commons.is_selected(c, d) == 5
    del d['nu one']
    assert commons.is_selected(c, d) == 4
    assert commons.is_selected(c, {'particles': 9}, default='none') == 'none'
    merged = commons.is_selected(c, {'all': {'gravity': 'pm', 'x': 1}, 'fluid': {'x': 2}},
                                 accumulate=True)
    assert merged == {'gravity': 'pm', 'x': 2}


def test_select_forces_forms_and_defaults():
    """select_forces as the reference reads it (commons.py:3664-3700): a dict per selector, a
    bare force name (its default method), or nothing at all — then the methods follow the
    global potential grid sizes."""
    import pytest
    from concept_amd import commons
    p = commons.load_params({'potential_options': {'gridsize': {'global': {'gravity': {'pm': 64}}}}})
    assert p.select_forces == {'particles': {'gravity': 'pm'}, 'fluid': {'gravity': 'pm'}}
    p = commons.load_params({'potential_options': {'gridsize': {'global': {'gravity': {
        'pm': 64, 'p3m': 128}}}}})
    assert p.select_forces == {'particles': {'gravity': 'p3m'}, 'fluid': {'gravity': 'pm'}}
    p = commons.load_params({'select_forces': {'Matter': 'gravity'}})
    assert p.select_forces == {'matter': {'gravity': 'p3m'}}
    p = commons.load_params({'select_forces': {'all': {'Gravity': 'PM'}}})
    assert p.select_forces == {'all': {'gravity': 'pm'}}
    with pytest.raises(ValueError):
        commons.load_params({'select_forces': {'all': {'gravity': 'tree'}}})
    # vertex-centred grids keep one more ghost layer (commons.py:4411-4419)
    assert commons.load_params({'cell_centered': False}).nghosts == 3
    assert commons.load_params({}).nghosts == 2


def test_parameter_file_with_foreign_names_and_late_definitions():
    """A parameter file in the style of the reference's param/example_*: names of other
    subsystems (`path`, `param`, CLASS settings), multi-line dict literals, `h` used before H0
    is defined, Fourier-space differentiation by name.  What the gravity path reads must come
    out; the rest is ignored."""
    from concept_amd import commons
    text = '''
_n = 48
initial_conditions = {'species': 'matter', 'N': _n**3}
output_dirs = {'powerspec': f'{path.output_dir}/{param}'}
output_times = {'powerspec': logspace(log10(a_begin), log10(1), 3)}
boxsize = 300*Mpc/h
potential_options = {
    'gridsize': {
        'global': {
            'gravity': {
                'pm' : _n//2,
                'p3m': 2*_n,
            },
        },
    },
    'differentiation': {
        'all': {'gravity': {'pm': 'Fourier', 'p3m': 4}},
    },
}
H0 = 75*km/(s*Mpc)
a_begin = 0.02
class_params = {'N_ncdm': 1, 'm_ncdm': 0.1}
shortrange_params = {'gravity': {'scale': '1.1*boxsize/gridsize', 'range': '4.8*scale'}}
'''
    p = commons.load_params(text)
    assert abs(p.boxsize - 400.0) < 1e-9                      # h = 0.75 from the later H0
    assert p.potential_options['gridsize']['global']['gravity'] == {'pm': 24, 'p3m': 96}
    assert p.potential_options['differentiation']['all']['gravity'] == {'pm': 0, 'p3m': 4}
    assert p.select_forces == {'particles': {'gravity': 'p3m'}, 'fluid': {'gravity': 'pm'}}
    sr = commons.resolve_shortrange(p, 96)
    assert abs(sr['scale'] - 1.1*400.0/96) < 1e-12 and abs(sr['range'] - 4.8*sr['scale']) < 1e-12
    assert p.nghosts == 2


def test_cosmology_and_timestep_parameters():
    """The parameters of the time loop and its clock keep the reference's names and defaults
    (commons.py:3630-3638, 3875-3890, 4313-4319, 4435-4479)."""
    from concept_amd import commons
    p = commons.load_params({})
    u = p.units
    assert p.H0 == 67*u.km/(u.s*u.Mpc) and p.Ωb == 0.049 and p.Ωcdm == 0.27
    assert p.a_begin == 1 and p.t_begin == 0 and p.enable_Hubble is True
    assert p.Δa_max_early == 0.00153 and p.Δa_max_late == 0.022
    assert p.Δt_base_background_factor == p.Δt_base_nonlinear_factor == p.Δt_rung_factor == 1
    assert p.Δt_increase_max_factor == float('inf') and p.static_timestepping is None
    assert p.ρ_crit == 3*p.H0**2/(8*commons.π*p.G_Newton) and p.ρ_mbar == p.Ωm*p.ρ_crit
    assert p.output_times == {'a': (), 't': ()}
    # output_times: {kind: times} are scale factors with the Hubble expansion, cosmic times
    # without; {'a': ..., 't': ...} with {kind: times} or plain times inside
    p = commons.load_params("output_times = {'snapshot': (0.1, 0.5, 1), 'powerspec': 1}\n")
    assert sorted(p.output_times['a']) == [0.1, 0.5, 1.0, 1.0] and p.output_times['t'] == ()
    p = commons.load_params({'enable_Hubble': False, 'output_times': {'snapshot': (2, 3)}})
    assert p.output_times == {'a': (), 't': (2.0, 3.0)}
    p = commons.load_params({'output_times': {'a': {'snapshot': 0.5}, 't': (13.0,)}})
    assert p.output_times == {'a': (0.5,), 't': (13.0,)}
    for bad in ({'Δt_increase_max_factor': 1.0}, {'Δa_max_late': 0},
                {'enable_Hubble': False, 'static_timestepping': (lambda a: 0.01)}):
        with pytest.raises(ValueError):
            commons.load_params(bad)


def test_parameter_file_forward_references_and_skipped_path_parameters():
    """statements are re-executed until nothing new resolves (later definitions
    reach earlier uses, with and without H0); a statement assigning one of THIS path's
    parameters that never runs is warned about instead of silently defaulting."""
    import warnings
    from concept_amd import commons
    p = commons.load_params("boxsize = _L*Mpc\n_L = 100\nfoo = CLASS_thing\n")
    assert p.boxsize == 100.0
    p = commons.load_params("boxsize = _L*Mpc/h\n_L = 100\nH0 = 50*km/(s*Mpc)\nbar = _undefined\n")
    assert p.boxsize == 200.0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        p = commons.load_params("boxsize = 3*_never_defined\nN_rungs = 4\n")
    assert p.N_rungs == 4 and any('boxsize' in str(x.message) for x in w)


def test_library_is_built_from_the_sources_in_the_tree():
    """libconcept_gpu.so.srchash (written by concept_amd.build) names the sources the library was
    built from: a library older than an edit of csrc/ would run — and be profiled — as something
    the tree no longer describes (bench.py quotes the committed counter run only for a matching
    hash).  __graft_entry__.build() rebuilds a stale library; this catches one that was not."""
    from concept_amd import build
    built = open(build.LIB + '.srchash').read().strip()
    assert built == build.source_hash(), 'run `python -m concept_amd.build`'


def test_output_parameters():
    """load_params: the input / output names of a parameter file (commons.py:2547-2572,
    2787-2830) — output_dirs as a dict or one directory, output_bases, snapshot_type,
    gadget_snapshot_params with the reference's loose key matching, output_times by kind (all of
    them are dumps of the time loop, the 'snapshot' ones write files), initial_conditions"""
    from concept_amd import commons
    p = commons.load_params("""
boxsize = 100*Mpc
output_dirs = {'snapshot': '/tmp/out', 'powerspec': '/tmp/ps'}
output_times = {'a': {'snapshot': (0.5, 1), 'powerspec': 0.7}, 't': {'snapshot': 13*Gyr}}
snapshot_type = 'GADGET'
gadget_snapshot_params = {'SnapFormat': 1, 'dataformat': {'POS': 64}, 'Particles per file': 1000}
initial_conditions = '/tmp/ic'
""")
    assert p.output_times == {'a': (0.5, 1.0, 0.7), 't': (13.0,)}
    assert p.snapshot_times == {'a': (0.5, 1.0), 't': (13.0,)}
    assert p.output_dirs['snapshot'] == '/tmp/out' and p.output_bases['snapshot'] == 'snapshot'
    assert p.snapshot_type == 'gadget' and p.initial_conditions == '/tmp/ic'
    gsp = p.gadget_snapshot_params
    assert gsp['snapformat'] == 1 and gsp['particles per file'] == 1000
    assert gsp['dataformat'] == {'POS': 64, 'VEL': 32, 'ID': 'automatic'}
    p = commons.load_params({'output_dirs': '/tmp/all', 'output_times': {'snapshot': 1.0},
                             'output_bases': {'snapshot': 'snap'}})
    assert p.output_dirs == {'snapshot': '/tmp/all'} and p.output_bases['snapshot'] == 'snap'
    assert p.snapshot_times['a'] == (1.0,) and p.snapshot_type == 'concept'
    with pytest.raises(ValueError, match='snapformat'):
        commons.load_params({'gadget_snapshot_params': {'snapformat': 3}})
    p = commons.load_params({})
    assert p.output_dirs == {} and p.snapshot_times == {'a': (), 't': ()}


def test_bench_box_is_a_function_of_seed_and_identifier(tmp_path, monkeypatch):
    """bench.py's particles: whichever ranks make the chunks of identifiers,
    the union is the 1-rank box, row by row; and the 1-rank values an N-rank run compares its
    sample with survive the round trip through their file bit for bit."""
    import argparse
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    monkeypatch.setattr(bench, 'ID_CHUNK', 4096)
    args = argparse.Namespace(seed=3, thermal=0.2)
    n_p, L = 5*4096 + 17, 64.0
    dev = torch.device('cpu')
    one = bench.global_particles(torch, args, n_p, L, 1.0, 1.0, 1e-4, dev)
    assert one[0].shape == (n_p, 3) and torch.equal(one[2], torch.arange(n_p))
    assert float(one[0].min()) >= 0 and float(one[0].max()) < L and float(one[1].abs().max()) > 0
    for world in (2, 3, 8):
        parts = [bench.global_particles(torch, args, n_p, L, 1.0, 1.0, 1e-4, dev, r, world)
                 for r in range(world)]
        ids = torch.cat([p[2] for p in parts])
        order = torch.argsort(ids)
        assert torch.equal(ids[order], one[2])
        assert torch.equal(torch.cat([p[0] for p in parts])[order], one[0])
        assert torch.equal(torch.cat([p[1] for p in parts])[order], one[1])
    other = bench.global_particles(torch, argparse.Namespace(seed=4, thermal=0.2), n_p, L, 1.0, 1.0,
                                   1e-4, dev)
    assert not torch.equal(other[0], one[0])
    monkeypatch.setattr(bench, 'VERIFY_DIR', str(tmp_path))
    sel = slice(0, n_p, 41)
    path = bench.verify_save('k', one[2][sel].numpy(), one[0][sel].numpy(), one[1][sel].numpy(),
                             n_p, 1.25, {'made_by': 'test'})
    got = bench.verify_load('k')
    assert os.path.exists(path) and got['particles'] == n_p and got['sum_mom2'] == 1.25
    assert np.array_equal(got['ids'], one[2][sel].numpy())
    assert np.array_equal(got['pos'], one[0][sel].numpy())
    assert np.array_equal(got['mom'], one[1][sel].numpy())
    assert bench.verify_load('no such key') is None
