"""Time cg_shortrange_sweep_tiles alone (256^3 particles / 512^3 mesh, uniform and clustered):
`python tools/sr_mfma_time.py [uniform] [clustered]`; CONCEPT_GPU_LIB selects a variant build."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concept_amd import commons, shortrange  # noqa: E402
from concept_amd.mesh import PotentialMesh  # noqa: E402
from tools.sr_mfma_check import positions  # noqa: E402

N, L, n = 512, 512.0, 256**3
mesh = PotentialMesh(N, L)
for dist in (sys.argv[1:] or ['uniform', 'clustered']):
    gen = torch.Generator(device='cuda').manual_seed(3)
    pos = positions(dist, n, L, gen)
    scale = 1.25*L/N
    rng_ = 4.5*scale
    nt = int(L/rng_*(1 + commons.machine_ϵ))
    table, maxr2 = shortrange.get_shortrange_table(0.025*L/256, scale, rng_, 4096, 'spline',
                                                   pos.device)
    dm = torch.zeros_like(pos)
    lst = mesh.shortrange_tiles(pos, nt, L/nt)
    mesh.shortrange_sweep_tiles(lst, dm, lst, nt, table, 4095/maxr2, rng_**2, 1.0)
    torch.cuda.synchronize()
    reps = 5 if dist == 'uniform' else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        mesh.shortrange_sweep_tiles(lst, dm, lst, nt, table, 4095/maxr2, rng_**2, 1.0)
    torch.cuda.synchronize()
    print(f'{os.environ.get("CONCEPT_GPU_LIB", "default")} {dist}: sweep '
          f'{(time.perf_counter() - t0)/reps*1e3:.2f} ms', flush=True)
    from concept_amd import lib as _lib
    if hasattr(_lib.raw(), 'cg_srm_debug_counters'):
        import ctypes
        out = (ctypes.c_ulonglong*8)()
        _lib.raw().cg_srm_debug_counters(out, 1)
        k = reps + 1
        print(f'   per sweep: wave trips {out[0]/k:.4g}, lane candidates {out[1]/k:.4g} (in range of '
              f'the wave {out[4]/k:.4g}), hits {out[2]/k:.4g}, products {out[3]/k:.4g} '
              f'-> lanes busy per trip {out[1]/max(out[0], 1)/64:.2f}', flush=True)
