#!/usr/bin/env python3
"""One-off soak of the P³M time loop with adaptive rungs at BASELINE configs[2]'s size (256^3
particles / 512^3 mesh, 8 rungs): a white-noise field at rest from a = 0.02 to a_end (default
0.1), through stepper.Timeloop.  Prints steps, rung populations at the end, checks identifiers,
bounds and the net momentum."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from concept_amd import commons, stepper  # noqa: E402
from concept_amd.species import Component  # noqa: E402

a1 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
a0 = float(os.environ.get('SOAK_A0', '0.02'))   # SOAK_DIST=clustered: the bench's clustered box
n_side, N = 256, 512
n = n_side**3
p = commons.load_params({
    'boxsize': 512.0, 'H0': 0.07, 'Ωb': 0.05, 'Ωcdm': 0.25, 'a_begin': a0,
    'output_times': {'a': (a1,)},
    'potential_options': {'gridsize': {'gravity': {'p3m': N}}},
    'select_forces': {'all': {'gravity': 'p3m'}}})
mass = p.ρ_mbar*p.boxsize**3/n
c = Component('matter', 'matter', N=n, mass=mass)
gen = torch.Generator(device='cuda').manual_seed(13)
torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen, out=c.pos)
c.pos.mul_(p.boxsize*(1 - 1e-13))
if os.environ.get('SOAK_DIST') == 'clustered':
    from tools.sr_positions import positions
    c.pos.copy_(positions('clustered', n, p.boxsize, gen))
c.mom.zero_()
stamps = []
def on_step(lp):
    torch.cuda.synchronize()
    stamps.append(time.perf_counter())
loop = stepper.Timeloop([c], on_step=on_step)
t0 = time.perf_counter()
loop.run()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
d = np.diff(np.array(stamps))
print(f'a {a0} -> {loop.cosmo.a}: {loop.time_step} base steps in {wall:.2f} s; s per step: first 5 '
      + ' '.join(f'{v:.2f}' for v in d[:5]) + ' | last 5 ' + ' '.join(f'{v:.2f}' for v in d[-5:]))
from concept_amd import shortrange  # noqa: E402
print('sweeps without a cell list:', shortrange.sparse_sweeps)
pops = torch.bincount(c.rung_indices.long(), minlength=8).tolist()
print('rung populations at the end:', pops)
assert c.N_local == n
ids = c.ids.sort().values
assert bool((ids == torch.arange(n, device='cuda')).all()), 'identifiers are not a permutation'
assert bool(((c.pos >= 0) & (c.pos < p.boxsize)).all()) and bool(torch.isfinite(c.mom).all())
print(f'net momentum / sum |mom| = {float(c.mom.sum(0).abs().max()/c.mom.abs().sum()):.2e}; '
      f'rms peculiar velocity {float(((c.mom/(mass*loop.cosmo.a))**2).sum(1).mean().sqrt())/(p.units.km/p.units.s):.1f} km/s')
