"""The cells sweep by itself (CONCEPT_GPU_SR_DENSE_MIN=0) against the cells sweep that hands its
densely populated tiles to cg_shortrange_dense.hip (the default), on the same particles: the
largest difference of the kicks relative to the largest kick, and the time of each.
`python tools/sr_dense_check.py [small] [scan] [big] [time]`"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concept_amd import commons, shortrange  # noqa: E402
from concept_amd.mesh import PotentialMesh  # noqa: E402
from tools.sr_positions import positions  # noqa: E402


def run(N, npart, dist, seed=3, reps=0, min_pops=(96,), L=None):
    L = float(N) if L is None else L
    mesh = PotentialMesh(N, L)
    gen = torch.Generator(device='cuda').manual_seed(seed)
    pos = positions(dist, npart, L, gen)
    scale = 1.25*L/N
    rng_ = 4.5*scale
    nt = int(L/rng_*(1 + commons.machine_ϵ))
    table, maxr2 = shortrange.get_shortrange_table(0.025*L/max(round(npart**(1/3)), 1), scale, rng_,
                                                   4096, 'spline', pos.device)
    cells = mesh.shortrange_cells(pos, nt, L/nt)

    def sweep(dm):
        mesh.shortrange_sweep_cells(cells, dm, cells, nt, table, 4095/maxr2, rng_**2, 1.0)

    def timed():
        dm = torch.zeros_like(pos)
        sweep(dm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(reps, 1)):
            sweep(dm)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0)/max(reps, 1)*1e3

    os.environ['CONCEPT_GPU_SR_DENSE_MIN'] = '0'
    ref = torch.zeros_like(pos)
    sweep(ref)
    torch.cuda.synchronize()
    big = float(ref.abs().max())
    line = f'N={N} n={npart} {dist} nt={nt} ({npart/nt**3:.0f} per tile):'
    if reps:
        line += f' cells only {timed():.2f} ms;'
    ok = True
    for mp in min_pops:
        os.environ['CONCEPT_GPU_SR_DENSE_MIN'] = str(mp)
        dm = torch.zeros_like(pos)
        sweep(dm)
        torch.cuda.synchronize()
        err = float((dm - ref).abs().max())
        ok &= err <= 1e-11*big
        line += f' min {mp}: max|Δ|/max {err/big if big else err:.2e}'
        if reps:
            line += f' {timed():.2f} ms;'
    os.environ.pop('CONCEPT_GPU_SR_DENSE_MIN', None)
    print(line, flush=True)
    mesh.close()
    return ok


if __name__ == '__main__':
    what = sys.argv[1:] or ['small']
    ok = True
    if 'small' in what:
        for N, n, dist in ((32, 8**3, 'uniform'), (32, 20**3, 'uniform'), (48, 16**3, 'clustered'),
                           (64, 32**3, 'uniform'), (64, 32**3, 'clustered'), (128, 4000, 'void'),
                           (128, 64**3, 'uniform'), (128, 64**3, 'clustered'), (64, 128**3, 'uniform')):
            ok &= run(N, n, dist, min_pops=(1, 8, 96))
    if 'scan' in what:  # uniform boxes of rising density: where the dense sweep overtakes
        for per_tile in (50, 100, 200, 400, 800, 1600):
            ok &= run(90, 4096*per_tile, 'uniform', reps=3, min_pops=(32,))
    if 'big' in what:
        ok &= run(512, 256**3, 'uniform', reps=3 if 'time' in what else 0)
        ok &= run(512, 256**3, 'clustered', reps=3 if 'time' in what else 0,
                  min_pops=(48, 64, 96, 128, 192) if 'time' in what else (96,))
    print('OK' if ok else 'MISMATCH')
    sys.exit(0 if ok else 1)
