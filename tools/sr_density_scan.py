"""Both short-range sweeps on uniform boxes of rising density (particles per short-range tile):
where, if anywhere, the tile sweep with sub-cell order and block culling overtakes the half-tile
cells sweep.  `python tools/sr_density_scan.py`"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concept_amd import commons, shortrange  # noqa: E402
from concept_amd.mesh import PotentialMesh  # noqa: E402


def run(N, npart, reps=3):
    L = float(N)
    mesh = PotentialMesh(N, L)
    gen = torch.Generator(device='cuda').manual_seed(5)
    pos = torch.rand((npart, 3), dtype=torch.float64, device='cuda', generator=gen)*(L*(1 - 1e-13))
    scale = 1.25*L/N
    rng_ = 4.5*scale
    nt = int(L/rng_*(1 + commons.machine_ϵ))
    table, maxr2 = shortrange.get_shortrange_table(0.025*L/max(round(npart**(1/3)), 1), scale, rng_,
                                                   4096, 'spline', pos.device)
    line = f'N={N} n={npart} nt={nt} per tile {npart/nt**3:8.1f}:'
    for name, build, sweep in (('cells', mesh.shortrange_cells, mesh.shortrange_sweep_cells),
                               ('tiles', mesh.shortrange_tiles, mesh.shortrange_sweep_tiles)):
        dm = torch.zeros_like(pos)
        lst = build(pos, nt, L/nt)
        sweep(lst, dm, lst, nt, table, 4095/maxr2, rng_**2, 1.0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            sweep(lst, dm, lst, nt, table, 4095/maxr2, rng_**2, 1.0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        line += f'  {name} {(t2 - t1)/reps*1e3:8.2f} ms'
    print(line, flush=True)
    mesh.close()


if __name__ == '__main__':
    # nt = N / 5.625: N = 90 -> 16 tiles per edge = 4096 tiles
    for per_tile in (25, 50, 100, 200, 400, 800, 1600, 3200):
        run(90, 4096*per_tile)
