"""A/B of the two tile sweeps on the GPU: cg_shortrange_sweep_tiles (matrix-core pre-filter)
against cg_shortrange_sweep_cells (half-tile cells) on the same particles — max |Δ| relative to
the largest kick, and the time of each.  `python tools/sr_mfma_check.py [small] [big] [time]`"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concept_amd import commons, shortrange  # noqa: E402
from concept_amd.mesh import PotentialMesh  # noqa: E402


def positions(dist, n, L, gen):
    pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen)
    if dist == 'clustered':
        centres = torch.rand((64, 3), dtype=torch.float64, device='cuda', generator=gen)*L
        which = torch.randint(0, 64, (n,), device='cuda', generator=gen)
        blob = centres[which] + torch.randn((n, 3), dtype=torch.float64, device='cuda',
                                            generator=gen)*(L/40)
        keep = torch.rand(n, dtype=torch.float64, device='cuda', generator=gen) < 0.2
        pos = torch.where(keep[:, None], pos*L, torch.remainder(blob, L))
        return pos.clamp_(0.0, L*(1 - 1e-13)).contiguous()
    if dist == 'void':  # a few particles in a big box: chunks spanning many tiles
        return pos*(L*(1 - 1e-13))
    return pos*(L*(1 - 1e-13))


def run(N, npart, dist, seed=3, reps=0, L=None):
    L = float(N) if L is None else L
    mesh = PotentialMesh(N, L)
    gen = torch.Generator(device='cuda').manual_seed(seed)
    pos = positions(dist, npart, L, gen)
    scale = 1.25*L/N
    rng_ = 4.5*scale
    nt = int(L/rng_*(1 + commons.machine_ϵ))
    table, maxr2 = shortrange.get_shortrange_table(0.025*L/max(round(npart**(1/3)), 1), scale, rng_,
                                                   4096, 'spline', pos.device)
    dm_c = torch.zeros_like(pos)
    cells = mesh.shortrange_cells(pos, nt, L/nt)
    mesh.shortrange_sweep_cells(cells, dm_c, cells, nt, table, 4095/maxr2, rng_**2, 1.0)
    dm_t = torch.zeros_like(pos)
    tiles = mesh.shortrange_tiles(pos, nt, L/nt)
    mesh.shortrange_sweep_tiles(tiles, dm_t, tiles, nt, table, 4095/maxr2, rng_**2, 1.0)
    torch.cuda.synchronize()
    ref = float(dm_c.abs().max())
    err = float((dm_t - dm_c).abs().max())
    bad = int(((dm_t - dm_c).abs().max(1).values > 1e-11*ref).sum())
    line = f'N={N} n={npart} {dist} nt={nt}: max|Δ|/max = {err/ref if ref else err:.3e}  rows off: {bad}'
    if reps:
        for name, build, sweep in (('cells', mesh.shortrange_cells, mesh.shortrange_sweep_cells),
                                   ('tiles', mesh.shortrange_tiles, mesh.shortrange_sweep_tiles)):
            dm = torch.zeros_like(pos)
            lst = build(pos, nt, L/nt)
            sweep(lst, dm, lst, nt, table, 4095/maxr2, rng_**2, 1.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                lst = build(pos, nt, L/nt)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                sweep(lst, dm, lst, nt, table, 4095/maxr2, rng_**2, 1.0)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            line += f'  | {name}: build {(t1 - t0)/reps*1e3:.2f} ms, sweep {(t2 - t1)/reps*1e3:.2f} ms'
    print(line, flush=True)
    mesh.close()
    return err <= 1e-11*ref


if __name__ == '__main__':
    what = sys.argv[1:] or ['small']
    ok = True
    if 'small' in what:
        for N, n, dist in ((32, 8**3, 'uniform'), (32, 20**3, 'uniform'), (48, 16**3, 'clustered'),
                           (64, 32**3, 'uniform'), (64, 32**3, 'clustered'), (128, 4000, 'void'),
                           (128, 64**3, 'uniform'), (128, 64**3, 'clustered')):
            ok &= run(N, n, dist)
    if 'dense' in what:  # every tile dense: 128^3 particles on a 64^3 mesh (~1600 per tile)
        ok &= run(64, 128**3, 'uniform', reps=2)
        ok &= run(128, 128**3, 'uniform', reps=3)   # ~200 per tile
    if 'big' in what:
        for dist in ('uniform', 'clustered'):
            ok &= run(512, 256**3, dist, reps=3 if 'time' in what else 0)
    print('OK' if ok else 'MISMATCH')
    sys.exit(0 if ok else 1)
