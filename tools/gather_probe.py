"""Timing probe for the tiled gather-kick at the north-star size: differentiation order 2 vs 4
(48 vs 96 LDS reads per particle) with and without the fused histogram."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concept_amd.mesh import PotentialMesh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2**28
g = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
box = 1000.0
m = PotentialMesh(g, box)
torch.manual_seed(1)
pos = torch.rand((n, 3), dtype=torch.float64, device='cuda')*box
mom = torch.zeros_like(pos)
ids = torch.arange(n, dtype=torch.int64, device='cuda')
po, mo, io = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
tab = m.new_tile_table()
m.sort_particles(pos, mom, ids, po, mo, io, tab)
del pos, mom, ids
m.zero(); m.deposit_tiled(po, tab, 1.0); m.poisson_solve(4, -1.0)
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
out = {}
for order in (2, 4):
    out[f'order{order}'] = t(lambda: m.gather_kick_tiled(po, mo, tab, order, 1e-9))
    out[f'order{order}_prepare'] = t(lambda: m.gather_kick_tiled_prepare(po, mo, tab, order, 1e-9, 1e-9))
print(json.dumps(out))
