"""Is the cells sweep held back by the unequal loads of its four wavefronts per tile?  The same
number of particles (3 per half-tile cell: every wavefront has 6 receivers and every range the
same length) against the uniform random box (Poisson counts), pair tests per second of each."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concept_amd import commons, shortrange  # noqa: E402
from concept_amd.mesh import PotentialMesh  # noqa: E402

N, L = 512, 512.0
mesh = PotentialMesh(N, L)
scale = 1.25*L/N
rng_ = 4.5*scale
nt = int(L/rng_*(1 + commons.machine_ϵ))
nc = 2*nt
gen = torch.Generator(device='cuda').manual_seed(3)
table, maxr2 = shortrange.get_shortrange_table(0.025*L/256, scale, rng_, 4096, 'spline', 'cuda')
for per_cell in (3, 0):
    if per_cell:
        idx = torch.arange(nc**3, device='cuda').repeat_interleave(per_cell)
        cell = torch.stack((idx // (nc*nc), (idx // nc) % nc, idx % nc), 1).double()
        pos = (cell + 0.02 + 0.96*torch.rand((idx.numel(), 3), dtype=torch.float64, device='cuda',
                                             generator=gen))*(L/nc)
        pos = pos[torch.randperm(pos.shape[0], device='cuda', generator=gen)].contiguous()
    else:
        pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen)*(L*(1 - 1e-13))
    n = pos.shape[0]
    lst = mesh.shortrange_cells(pos, nt, L/nt)
    off = lst[1].long()
    pop = (off[1:] - off[:-1]).reshape(nc, nc, nc).double()
    box = sum(torch.roll(pop, s_, 0) for s_ in range(-2, 3))
    box = sum(torch.roll(box, s_, 1) for s_ in range(-2, 3))
    colz = box.reshape(nc, nc, nc//2, 2).sum(3)
    win = sum(torch.roll(colz, s_, 2) for s_ in (-1, 0, 1))
    tests = float((pop.reshape(nc, nc, nc//2, 2).sum(3)*win).sum())
    dm = torch.zeros_like(pos)
    mesh.shortrange_sweep_cells(lst, dm, lst, nt, table, 4095/maxr2, rng_**2, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        mesh.shortrange_sweep_cells(lst, dm, lst, nt, table, 4095/maxr2, rng_**2, 1.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0)/5*1e3
    print(f'{"3 per cell" if per_cell else "random"}: n = {n}, {tests/n:.0f} tests per particle, '
          f'sweep {ms:.2f} ms, {tests/ms*1e-9:.3f}e12 tests/s', flush=True)
