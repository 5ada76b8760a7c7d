"""What a sub-step of the P3M rung loop costs as a function of its lowest active rung: 256^3
particles / 512^3 mesh, rungs dealt out at random with the populations the soak ends with
(tools/soak_p3m.py: 52 % / 23 % / 13 % / 11 % / 0.9 %).  Times the cell list and the sweep with
rungs per lowest active rung; `python tools/sr_rung_cost.py [uniform|clustered]`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concept_amd import commons, shortrange  # noqa: E402
from concept_amd.mesh import PotentialMesh  # noqa: E402
from tools.sr_positions import positions  # noqa: E402

N = 512
L, n = float(N), 256**3
mesh = PotentialMesh(N, L)
dist = (sys.argv[1:] or ['uniform'])[0]
gen = torch.Generator(device='cuda').manual_seed(3)
pos = positions(dist, n, L, gen)
scale = 1.25*L/N
rng_ = 4.5*scale
nt = int(L/rng_*(1 + commons.machine_ϵ))
table, maxr2 = shortrange.get_shortrange_table(0.025*L/round(n**(1/3)), scale, rng_, 4096, 'spline',
                                               pos.device)
u = torch.rand(n, device='cuda', generator=gen)
edges = torch.tensor([0.519, 0.751, 0.883, 0.9913], device='cuda')
rung = torch.bucketize(u, edges).to(torch.int8)
factors = torch.ones(23, dtype=torch.float64, device='cuda')
dm = torch.zeros_like(pos)


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0)/reps*1e3


# (particle memory in the order of the mesh tiles, as the time loop keeps it)
key = (pos/16).floor().long()
pos = pos[torch.argsort((key[:, 0]*64 + key[:, 1])*64 + key[:, 2])].contiguous()
lst = mesh.shortrange_cells(pos, nt, L/nt)
print(f'{dist}: cell list {timed(lambda: mesh.shortrange_cells(pos, nt, L/nt)):.3f} ms', flush=True)
for la in range(5):
    act = int((rung >= la).sum())
    for kind in ('plain', 'active first', 'by cell'):
        if kind != 'plain':
            if la == 0:
                continue
            blocks = kind == 'active first'
            t_l = timed(lambda: mesh.shortrange_cells(pos, nt, L/nt, (rung, rung, la), blocks))
            rl = mesh.shortrange_cells(pos, nt, L/nt, (rung, rung, la), blocks)
        else:
            t_l, rl = float('nan'), lst
        f = lambda: mesh.shortrange_sweep_cells(rl, dm, lst, nt, table, 4095/maxr2, rng_**2, 0.0,  # noqa: E731
                                                (factors, rung, rung, la),
                                                act if kind == 'by cell' else None)
        t = timed(f)
        mesh.shortrange_stats(True)
        f()
        torch.cuda.synchronize()
        st = mesh.shortrange_stats(False)
        tests, hits, trips = (a + b for a, b in zip(st['cells'], st['dense']))
        print(f'{dist}: lowest active rung {la} ({kind} list{"" if t_l != t_l else f", made in {t_l:.3f} ms"}): '
              f'{act} receivers, sweep {t:.3f} ms; {tests:.3e} tests '
              f'({tests/max(act, 1):.0f} per receiver), {trips:.3e} trips, lane use '
              f'{tests/max(64*trips, 1):.3f}', flush=True)
