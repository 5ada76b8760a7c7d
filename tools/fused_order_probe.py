#!/usr/bin/env python3
"""A/B of the workgroup -> tile order (cgk_tile_order, CONCEPT_GPU_TILE_ORDER_MIN) in ONE process:
the same particle arrays, a context per setting, alternating — deposit and fused pass of the
bench's north-star box, uniform and clustered.  (Process-to-process the fused pass moves by
+-2 % with the placement of its pages, which is more than what the table costs a uniform box.)
    python tools/fused_order_probe.py [uniform|clustered ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from concept_amd.mesh import PotentialMesh
n_p, N = 2**28, 1024
L = float(N)
dev = torch.device('cuda')
dt = 1e-4


def particles(dist):
    gen = torch.Generator(device=dev).manual_seed(1)
    pos = torch.rand((n_p, 3), dtype=torch.float64, device=dev, generator=gen)*(L*(1 - 1e-13))
    if dist == 'clustered':  # bench.py's clustered box
        centres = torch.rand((64, 3), dtype=torch.float64, device=dev, generator=gen)*L
        which = torch.randint(0, 64, (n_p,), device=dev, generator=gen)
        blob = centres[which] + torch.randn((n_p, 3), dtype=torch.float64, device=dev, generator=gen)*(L/40)
        keep = torch.rand(n_p, dtype=torch.float64, device=dev, generator=gen) < 0.2
        pos = torch.where(keep[:, None], pos, torch.remainder(blob, L)).clamp_(0.0, L*(1 - 1e-13))
    mom = torch.randn((n_p, 3), dtype=torch.float64, device=dev, generator=gen)*(0.2/3**0.5/dt)
    return pos, mom


def timed(f, reps=6):
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return ms


for dist in (sys.argv[1:] or ['uniform', 'clustered']):
    pos, mom = particles(dist)
    mesh = PotentialMesh(N, L)
    cap = mesh.region_capacity(n_p)
    pa = torch.empty((cap, 3), dtype=torch.float64, device=dev)
    ma = torch.empty((cap, 3), dtype=torch.float64, device=dev)
    table = mesh.sort_particles(pos, mom, None, pa[:n_p], ma[:n_p], None)
    mesh.close()
    del pos, mom
    pb, mb = torch.empty_like(pa), torch.empty_like(ma)
    # PROBE_MINS=1536,3072,...: the order on with these thresholds (CONCEPT_GPU_TILE_ORDER_MIN) instead
    # of off against on
    mins = [m for m in os.environ.get('PROBE_MINS', '').split(',') if m]
    for rep in range(3 if mins else 2):
        for mode in (mins or ('0', '1')):
            # (off against on: thresholds 0 = the plain walk and 1536 = the default)
            os.environ['CONCEPT_GPU_TILE_ORDER_MIN'] = mode if mins else ('0', '1536')[int(mode)]
            mesh = PotentialMesh(N, L)
            # (each deposit makes the order anew: its cost is inside these times)
            dep = timed(lambda: mesh.deposit_tiled(pa[:n_p], table, 1.0/N**3))
            mesh.poisson_solve(4, -L**2/3.141592653589793, False, 0.0)
            start_out, count_out = mesh.new_region_table()
            mesh.predict_regions(table, None, start_out)
            fu = timed(lambda: mesh.gather_kick_drift_scatter(
                pa, ma, None, table, None, pb, mb, None, start_out, count_out, 2, -dt, dt))
            nh = mesh.tile_order()
            nh = -1 if nh is None else len(nh)
            print(f'{dist:9s} {"min" if mins else "order"} {mode} heavy tiles {nh:6d}  deposit',
                  ' '.join(f'{v:.3f}' for v in dep[1:]), ' fused', ' '.join(f'{v:.3f}' for v in fu[1:]),
                  ' flags', mesh.error_flags(), flush=True)
            mesh.close()
    del pa, ma, pb, mb
    torch.cuda.empty_cache()
