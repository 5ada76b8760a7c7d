#!/usr/bin/env python3
"""One-off soak of the streaming time loop at the metric's size: 2^28 particles / 1024^3 PM mesh
through stepper.Timeloop from a_begin to a_end (default 0.02 -> 0.3: ~150 base steps in which
the white-noise initial field clusters at the mesh scale, so tile populations change and the
predicted region capacities are exercised).  Prints passes, replays, wrong guesses, the spread
of the tile populations at the end, and checks that every particle is still there."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from concept_amd import commons, stepper  # noqa: E402
from concept_amd.species import Component  # noqa: E402

a0, a1 = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (0.02, 0.3)
n, N = 2**28, 1024
p = commons.load_params({
    'boxsize': 1024.0, 'H0': 0.07, 'Ωb': 0.05, 'Ωcdm': 0.25, 'a_begin': a0,
    'output_times': {'a': (a1,)},
    'potential_options': {'gridsize': {'gravity': {'pm': N}}},
    'select_forces': {'all': {'gravity': 'pm'}}})
mass = p.ρ_mbar*p.boxsize**3/n
c = Component('matter', 'matter', N=n, mass=mass)
gen = torch.Generator(device='cuda').manual_seed(7)
torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen, out=c.pos)
c.pos.mul_(p.boxsize*(1 - 1e-13))
c.mom.zero_()
stamps = []
def on_step(lp):
    torch.cuda.synchronize()
    stamps.append((time.perf_counter(), lp.cosmo.a))
loop = stepper.Timeloop([c], on_step=on_step)
replays = stepper.stream_replays
torch.cuda.synchronize()
t0 = time.perf_counter()
loop.run()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f'a {a0} -> {loop.cosmo.a}: {loop.time_step} base steps, {loop.stream_passes} passes, '
      f'{stepper.stream_replays - replays} replays, {loop.stream_wrong_guesses} wrong guesses, '
      f'{wall:.2f} s = {wall/max(loop.time_step, 1)*1e3:.1f} ms per base step')
import numpy as np  # noqa: E402
d = np.diff(np.array([t for t, _ in stamps]))*1e3
print('ms between base steps: first 10', ' '.join(f'{v:.0f}' for v in d[:10]), '| last 10',
      ' '.join(f'{v:.0f}' for v in d[-10:]), f'| median {np.median(d):.1f}, max {d.max():.0f}')
assert c.N_local == n
ids = c.ids.sort().values
assert bool((ids == torch.arange(n, device='cuda')).all()), 'identifiers are not a permutation'
del ids
assert bool(((c.pos >= 0) & (c.pos < p.boxsize)).all())
mesh = c._mesh()
c.tile_sort(mesh)
tab = c.tile_table.long()
pop = (tab[8::8] - tab[:-8:8]) if tab.numel() % 8 == 1 else None
if pop is not None:
    print(f'tile populations at the end: mean {pop.float().mean():.0f}, min {int(pop.min())}, '
          f'max {int(pop.max())}, rms/mean {float(pop.float().std()/pop.float().mean()):.2f}')
v = (c.mom/(mass*loop.cosmo.a)).norm(dim=1)
print(f'rms peculiar velocity {float((v**2).mean().sqrt())/(p.units.km/p.units.s):.1f} km/s, '
      f'net momentum / sum |mom| = {float(c.mom.sum(0).abs().max()/c.mom.abs().sum()):.2e}')
