"""Time cg_shortrange_sweep_cells (cells sweep + the dense tiles' sweep) alone, 256^3 particles /
512^3 mesh: `python tools/sr_dense_time.py [uniform] [clustered]`; CONCEPT_GPU_LIB selects a
variant build, CONCEPT_GPU_SR_DENSE_MIN=0 the cells sweep by itself."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concept_amd import commons, shortrange  # noqa: E402
from concept_amd.mesh import PotentialMesh  # noqa: E402
from tools.sr_positions import positions  # noqa: E402

N = int(os.environ.get('SR_N', '512'))
L, n = float(N), int(os.environ.get('SR_NP', str(256**3)))
mesh = PotentialMesh(N, L)
for dist in (sys.argv[1:] or ['clustered']):
    gen = torch.Generator(device='cuda').manual_seed(3)
    pos = positions(dist, n, L, gen)
    scale = 1.25*L/N
    rng_ = 4.5*scale
    nt = int(L/rng_*(1 + commons.machine_ϵ))
    table, maxr2 = shortrange.get_shortrange_table(0.025*L/round(n**(1/3)), scale, rng_, 4096, 'spline',
                                                   pos.device)
    dm = torch.zeros_like(pos)
    lst = mesh.shortrange_cells(pos, nt, L/nt)
    mesh.shortrange_sweep_cells(lst, dm, lst, nt, table, 4095/maxr2, rng_**2, 1.0)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        mesh.shortrange_sweep_cells(lst, dm, lst, nt, table, 4095/maxr2, rng_**2, 1.0)
    torch.cuda.synchronize()
    print(f'{os.environ.get("CONCEPT_GPU_LIB", "default")} {dist}: sweep '
          f'{(time.perf_counter() - t0)/reps*1e3:.2f} ms', flush=True)
