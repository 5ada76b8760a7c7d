#!/usr/bin/env python3
"""Does it matter where the four particle arrays of the fused pass start relative to each other?
(pos, mom, pos_out, mom_out are read and written at the same row index by a workgroup; allocated
by themselves they all start on a 2 MB boundary.)  One process, a context per setting,
alternating: the arrays as allocated against the same arrays entered SKEW rows further each
(PROBE_SKEW_ROWS, default 1365 rows = 32,760 bytes: mom +1, pos_out +2, mom_out +3 times that)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from concept_amd.mesh import PotentialMesh
n_p, N = 2**28, 1024
L = float(N)
dev = torch.device('cuda')
dt = 1e-4
SK = int(os.environ.get('PROBE_SKEW_ROWS', '1365'))
gen = torch.Generator(device=dev).manual_seed(1)
pos = torch.rand((n_p, 3), dtype=torch.float64, device=dev, generator=gen)*(L*(1 - 1e-13))
mom = torch.randn((n_p, 3), dtype=torch.float64, device=dev, generator=gen)*(0.2/3**0.5/dt)
mesh = PotentialMesh(N, L)
cap = mesh.region_capacity(n_p)
mesh.close()
big = [torch.empty((cap + 4*SK, 3), dtype=torch.float64, device=dev) for _ in range(4)]
for rep in range(3):
    for skew in (0, 1):
        pa, ma, pb, mb = (big[i][i*SK*skew:i*SK*skew + cap] for i in range(4))
        mesh = PotentialMesh(N, L)
        table = mesh.sort_particles(pos, mom, None, pa[:n_p], ma[:n_p], None)
        mesh.deposit_tiled(pa[:n_p], table, 1.0/N**3)
        mesh.poisson_solve(4, -L**2/3.141592653589793, False, 0.0)
        s1, c1 = mesh.new_region_table()
        mesh.predict_regions(table, None, s1)
        mesh.gather_kick_drift_scatter(pa, ma, None, table, None, pb, mb, None, s1, c1, 2, -dt, dt)
        s2, c2 = mesh.new_region_table()
        mesh.predict_regions(s1, c1, s2)
        ms = []
        for i in range(6):   # the steady state: from regions with gaps
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            mesh.gather_kick_drift_scatter(pb, mb, None, s1, c1, pa, ma, None, s2, c2, 2, -dt, dt)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        print(f'skew {skew*SK:5d} rows: fused pass', ' '.join(f'{v:.3f}' for v in ms[1:]), 'flags',
              mesh.error_flags(), flush=True)
        mesh.close()
