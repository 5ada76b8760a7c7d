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
    assert binding.raw().cg_abi_version() == 1


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
