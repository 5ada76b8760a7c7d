"""bench.py keeps its contract under every flag it documents: one JSON line on stdout with the
driver's keys, `roofline` and (unless switched off) `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
        'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'}


def run(*flags, env=None):
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--workload', 'tiny',
                        '--steps', '2', '--warmup', '1'] + list(flags),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout
    d = json.loads(lines[0])
    assert KEYS <= set(d), KEYS - set(d)
    assert d['steps'] == 2 and d['warmup'] == 1 and d['value'] > 0 and d['dtype'] == 'f64'
    r = d['roofline']
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(r)
    assert 0 < r['frac'] <= 1
    for k in d.get('kernels', {}).values():
        assert k['frac_hbm'] <= 1
    return d


@pytest.mark.parametrize('flags', [
    (), ('--no-fused',), ('--no-sort',), ('--no-prepare',), ('--split-poisson',),
    ('--dist', 'lattice'), ('--dist', 'clustered'), ('--thermal', '0'), ('--p3m',),
    ('--p3m', '--sr-tiles'), ('--p3m', '--dist', 'clustered')])
def test_bench_flags(flags):
    d = run('--no-cpu-baseline', *flags)
    assert d['n_gpus'] == 1 and 'workload' in d['config']


def test_bench_cpu_baseline_and_sharded_paths():
    d = run()  # with the CPU baseline leg
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['cores'] >= 1 and cb['sample']
    # the N > 1 code path with one rank (RCCL), and two self-spawned ranks sharing the GPU (gloo)
    d = run('--no-cpu-baseline', env={'CONCEPT_BENCH_FORCE_DIST': '1'})
    assert d['n_gpus'] == 1 and 'stages_ms_rank0' in d
    d = run('--no-cpu-baseline', '--gpus', '2')
    assert d['n_gpus'] == 2 and d['transport']['bound'] == 'xgmi'
    lm = d['link_model']   # the prediction beside the measured stages
    assert lm['predicted_step_ms'] > 0 and lm['transpose_ms_at_link_rate'] > 0
    assert lm['measured_step_ms'] == pytest.approx(d['ms_per_step'], rel=1e-3)
    assert d['config']['particles'] == 32**3
