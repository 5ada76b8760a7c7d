#!/usr/bin/env python3
"""Time cg_gather_kick_drift_scatter by itself on the bench workload (2^28 particles / 1024^3,
thermal momenta): the same input every call, HIP events around each.  Used with
CONCEPT_GPU_LIB=<variant> (tools/variant.py) for A/B probes of the kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from concept_amd.mesh import PotentialMesh
n_p, N = (2**28, 1024) if len(sys.argv) < 2 else (int(sys.argv[1])**3, int(sys.argv[2]))
L = float(N)
dev = torch.device('cuda')
mesh = PotentialMesh(N, L)
gen = torch.Generator(device=dev).manual_seed(1)
pos = torch.rand((n_p, 3), dtype=torch.float64, device=dev, generator=gen)*(L*(1 - 1e-13))
if os.environ.get('PROBE_SUBBOX'):   # all particles in the sub-box [0, L/k)^3: k^-3 of the tiles, heavy
    pos /= float(os.environ['PROBE_SUBBOX'])
if os.environ.get('PROBE_CLUSTERED'):  # bench.py's clustered box
    centres = torch.rand((64, 3), dtype=torch.float64, device=dev, generator=gen)*L
    which = torch.randint(0, 64, (n_p,), device=dev, generator=gen)
    blob = centres[which] + torch.randn((n_p, 3), dtype=torch.float64, device=dev, generator=gen)*(L/40)
    keep = torch.rand(n_p, dtype=torch.float64, device=dev, generator=gen) < 0.2
    pos = torch.where(keep[:, None], pos, torch.remainder(blob, L)).clamp_(0.0, L*(1 - 1e-13))
    del centres, which, blob, keep
dt = 1e-4
mom = torch.randn((n_p, 3), dtype=torch.float64, device=dev, generator=gen)*(0.2/3**0.5/dt)
cap = mesh.region_capacity(n_p)
pa = torch.empty((cap, 3), dtype=torch.float64, device=dev)
ma = torch.empty((cap, 3), dtype=torch.float64, device=dev)
table = mesh.sort_particles(pos, mom, None, pa[:n_p], ma[:n_p], None)
del pos, mom
mesh.deposit_tiled(pa[:n_p], table, 1.0/N**3)
mesh.poisson_solve(4, -L**2/3.141592653589793, False, 0.0)
pb, mb = torch.empty_like(pa), torch.empty_like(ma)
start_out, count_out = mesh.new_region_table()
mesh.predict_regions(table, None, start_out)
ms = []
for i in range(int(os.environ.get("PROBE_ITERS", "6"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    mesh.gather_kick_drift_scatter(pa, ma, None, table, None, pb, mb, None, start_out, count_out,
                                   2, -dt, dt)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print('fused ms per call:', ' '.join(f'{v:.3f}' for v in ms), 'flags', mesh.error_flags(),
      'placed', int(count_out.long().sum()))
# the steady state: the input itself in regions with gaps (the output of the pass above)
start2, count2 = mesh.new_region_table()
mesh.predict_regions(start_out, count_out, start2)
ms = []
for i in range(int(os.environ.get("PROBE_ITERS", "6"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    mesh.gather_kick_drift_scatter(pb, mb, None, start_out, count_out, pa, ma, None, start2, count2,
                                   2, -dt, dt)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print('from regions with gaps:', ' '.join(f'{v:.3f}' for v in ms), 'flags', mesh.error_flags(),
      'placed', int(count2.long().sum()))

# probe build with -DCG_GK_TIMING: where wave 0 of every workgroup spent its cycles
import ctypes
try:
    h = ctypes.CDLL(os.environ.get('CONCEPT_GPU_LIB', ''))
    buf = (ctypes.c_ulonglong*16)()
    if h.cg_debug_gk_timing(buf, 1) == 0:
        tot = sum(buf[i] for i in range(5))
        names = ['staging', 'particle loads', 'gather/kick/drift/key', 'reservation', 'stores']
        print('workgroups', buf[8], ' cycles per workgroup:', ' '.join(f'{n} {buf[i]/max(buf[8],1):.0f} ({100*buf[i]/max(tot,1):.0f}%)' for i, n in enumerate(names)))
except (OSError, AttributeError):
    pass
