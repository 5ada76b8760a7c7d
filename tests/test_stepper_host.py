"""Host logic of the rung loop (no GPU): the step integrals worked out when read and the
ᔑdt_rungs arrays filled in when read (stepper._LazyIntegrals, _RungIntegrals) give what the
eager forms give (get_time_step_integrals, main.py:998-1073; the arrays of main.py:1480-1552);
the integrals' cache; bench.py's link model."""
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lazy_integrals_equal_the_eager_ones():
    from concept_amd import stepper
    keys = ['1', 'a**2', 'a**(-2)', ('a**(-3*w_eff)', 'matter'), ('pair', 'matter', 'matter')]
    calls = []

    def one(key, t0, t1):
        calls.append(key)
        return (hash(key) % 97 + 1)*(t1 - t0)
    eager = {k: one(k, 0.25, 0.75) for k in keys}
    calls.clear()
    lazy = stepper._LazyIntegrals(one, keys, 0.25, 0.75)
    assert calls == [] and len(lazy) == len(keys) and list(lazy) == keys and '1' in lazy
    assert 'b' not in lazy and lazy.get('b', 7) == 7
    assert lazy['a**2'] == eager['a**2'] and calls == ['a**2']
    assert lazy['a**2'] == eager['a**2'] and calls == ['a**2']          # worked out once
    assert dict(lazy.items()) == eager and sorted(map(str, lazy.keys())) == sorted(map(str, keys))
    try:
        lazy['b']
    except KeyError:
        pass
    else:
        raise AssertionError('an integrand outside the run\'s keys must raise')


def test_rung_integral_arrays_fill_in_when_read():
    from concept_amd import stepper
    keys = ['1', 'a**2']
    one = lambda key, t0, t1: (2.0 if key == '1' else 3.0)*(t1 - t0)   # noqa: E731
    nr = 4
    eager = {k: np.zeros(3*nr - 1) for k in keys}
    R = stepper._RungIntegrals(3*nr - 1)
    script = [(0, 0.0, 1.0), (nr, None, None), (2*nr, 0.0, 0.5), (1, 0.5, 1.0), (0, 0.25, 0.5),
              (nr + 1, 0.0, 0.125), (nr, None, None)]
    for index, t0, t1 in script:
        if t0 is None:
            R.set(-1, index)
            for arr in eager.values():
                arr[index] = -1
        else:
            R.set(stepper._LazyIntegrals(one, keys, t0, t1), index)
            for k, arr in eager.items():
                arr[index] = one(k, t0, t1)
        if index == 1:
            assert np.array_equal(R['1'], eager['1'])       # read in between: later sets still land
    for k in keys:
        assert k in R and np.array_equal(R[k], eager[k]) and np.array_equal(R.get(k), eager[k])
    assert 'x' not in R and R.get('x') is None
    # a plain dictionary of numbers (what a caller's own integrals give) works the same
    R2 = stepper._RungIntegrals(5)
    R2.set({'1': 0.5, 'a**2': 0.25}, 3)
    assert R2['1'][3] == 0.5 and R2['a**2'][3] == 0.25 and set(R2.keys()) == {'1', 'a**2'}
    assert all(a.shape == (5,) for a in R2.values())


def test_integrals_cache_prefetch():
    from concept_amd import stepper
    n = []

    def f(t0, t1):
        n.append((t0, t1))
        return {'1': t1 - t0}
    c = stepper._CachedIntegrals(f)
    assert c(0.0, 1.0) == {'1': 1.0} and len(n) == 1
    c.prefetch(1.0, 2.0)
    c.prefetch(1.0, 2.0)
    assert len(n) == 2
    assert c(1.0, 2.0) == {'1': 1.0} and len(n) == 2      # from the cache
    assert c(2.0, 3.0) == {'1': 1.0} and len(n) == 3


def test_link_model_prediction():
    """bench.py --link-model: the prediction a first SCALE run is compared with;
    one xGMI link per peer bounds the two transposes of a step."""
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--link-model'],
                         stdout=subprocess.PIPE, check=True, timeout=300).stdout.decode()
    d = json.loads(out)
    assert d['particles'] == 2**28 and d['gridsize'] == 1024
    rows = d['by_gpus']
    assert set(rows) == {'1', '2', '4', '8'}
    assert rows['1']['transpose_ms_at_link_rate'] == 0 and rows['1']['predicted_speedup'] == 1
    for P in (2, 4, 8):
        r = rows[str(P)]
        per_peer = 1024*1024*1026*8/P/P
        assert abs(r['bytes_per_peer_per_transpose'] - per_peer) <= 8
        assert abs(r['transpose_ms_at_link_rate'] - per_peer/76.8e9*1e3) < 1e-2
        assert r['predicted_step_ms'] >= 2*r['transpose_ms_at_link_rate']
    assert rows['8']['predicted_speedup'] > rows['4']['predicted_speedup'] > rows['2']['predicted_speedup']
    committed = json.load(open(os.path.join(REPO, 'profiles', 'r06_link_model_ns.json')))
    assert committed['by_gpus'] == rows
